// Device helpers for kernels that walk text (read files, FASTA letters): byte streams read as aligned words, and
// lane groups that copy / filter a span into an output at a group-uniform cursor.  Used by trim.hip and seqdb.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pgx {

// f(byte, index) for the bytes [off, off + n) of the text, in order, read as aligned 16-byte words: one lane per
// record walking single bytes would pull every 64-byte line from L2 once per byte (the lanes of a wavefront are a
// record apart, so nothing is shared in L1).  `text` must be 16-byte aligned and readable up to the next 16-byte boundary past the span (the buffers that hold
// text carry 16 bytes of padding).
template <typename F> __device__ __forceinline__ void for_bytes(const uint8_t *__restrict__ text, uint64_t off, uint64_t n, F f)
{
	const uint64_t hi = off + n;
	for (uint64_t w = off & ~15ull; w < hi; w += 16) {
		const uint4 v = *reinterpret_cast<const uint4 *>(text + w);
		const uint32_t q[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const uint64_t i = w + j;
			if (i >= off && i < hi)
				f((uint8_t)(q[j >> 2] >> (8 * (j & 3))), i - off);
		}
	}
}

// the same for a span given by a pointer of any alignment
template <typename F> __device__ __forceinline__ void for_bytes_at(const uint8_t *__restrict__ p, uint64_t n, F f)
{
	if (n == 0)
		return;
	const uint64_t mis = (uint64_t)(reinterpret_cast<uintptr_t>(p) & 15u);
	for_bytes(p - mis, mis, n, f);
}

// A record is written by a group of kGroup = 16 lanes (four records per wavefront in flight: the writers wait on a
// chain of dependent loads per record, more records per wavefront hide it).  `pos` is uniform within a group; the
// lanes of a group call these together (groups of one wavefront may be in different calls or iterations).
constexpr int kGroup = 16;

template <typename Keep, typename Map>
__device__ __forceinline__ void w_copy(char *__restrict__ out, uint64_t &pos, const uint8_t *__restrict__ src, uint64_t n, Keep keep, Map map)
{
	const int lane = threadIdx.x & (kGroup - 1), shift = (threadIdx.x & 63) & ~(kGroup - 1);
	const uint32_t lt = (1u << lane) - 1u;
	for (uint64_t base = 0; base < n; base += kGroup) {
		const uint64_t i = base + lane;
		uint8_t c = 0;
		bool k = false;
		if (i < n) {
			c = src[i];
			k = keep(c);
		}
		const uint32_t m = (uint32_t)(__ballot(k) >> shift) & ((1u << kGroup) - 1u);
		if (k)
			out[pos + __popc(m & lt)] = (char)map(c);
		pos += (uint64_t)__popc(m);
	}
}

__device__ __forceinline__ void w_fill(char *__restrict__ out, uint64_t &pos, char c, uint64_t n)
{
	for (uint64_t i = threadIdx.x & (kGroup - 1); i < n; i += kGroup)
		out[pos + i] = c;
	pos += n;
}

__device__ __forceinline__ void w_lit(char *__restrict__ out, uint64_t &pos, const char *lit, int n)
{
	const int lane = threadIdx.x & (kGroup - 1);
	if (lane < n)
		out[pos + lane] = lit[lane];
	pos += (uint64_t)n;
}

} // namespace pgx
