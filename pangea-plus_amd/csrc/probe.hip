// Diagnostics: the ceiling of the access shape the seed kernel lives on.  k_seed_extend fetches 8-56 useful bytes
// from a different 64-byte line with almost every lane-load (bucket table, posting records, block records, database
// windows), so the streaming figure of the memory (8 TB/s) is not its roof.  This probe measures what the chip
// delivers for exactly that shape with nothing else to do: every lane loads 8 bytes from uniformly random 64-byte
// lines of a table far larger than the caches, `kInFlight` independent loads per lane per step, full occupancy.
#include "engine.hpp"

namespace pgx {

constexpr int kInFlight = 8;

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_probe_gather(const uint64_t *__restrict__ table, uint64_t line_mask, uint32_t steps, int stream,
						      uint64_t *__restrict__ sink)
{
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t acc = 0, x = mix64(tid);
	for (uint32_t s = 0; s < steps; s++) {
		uint64_t v[kInFlight];
#pragma unroll
		for (int k = 0; k < kInFlight; k++) {
			x = mix64(x + k);
			const uint64_t *p = table + (x & line_mask) * 8; // one 8-byte word of a random 64-byte line
			v[k] = stream ? __builtin_nontemporal_load(p) : *p;
		}
#pragma unroll
		for (int k = 0; k < kInFlight; k++)
			acc += v[k];
	}
	if (acc == 0x123456789ull) // never: the table is zero-filled; keeps the loads alive
		sink[0] = acc;
}

} // namespace pgx

using namespace pgx;

// lines_per_s: 64-byte lines fetched per second by random 8-byte lane loads over a zero-filled table of `table_bytes`
// (rounded down to a power of two); stream != 0 uses non-temporal loads
extern "C" int pgx_probe_gather(uint64_t table_bytes, int stream, double *lines_per_s, double *ms)
{
	if (!lines_per_s || !ms || table_bytes < (1u << 20))
		return fail(PGX_E_ARG, "pgx_probe_gather: bad argument");
	PGX_TRY(require_device());
	uint64_t lines = 1;
	while (lines * 2 * 64 <= table_bytes)
		lines *= 2;
	DevBuf<uint64_t> table, sink;
	PGX_TRY(table.alloc(lines * 8, 0, 0, true));
	PGX_TRY(sink.alloc(1, 0, 0, true));
	const unsigned blocks = 256 * 32;
	const uint32_t steps = 64;
	hipEvent_t e0, e1;
	PGX_HIP(hipEventCreate(&e0));
	PGX_HIP(hipEventCreate(&e1));
	float best = 0;
	for (int rep = 0; rep < 4; rep++) { // the first launch warms up clocks and TLBs
		PGX_HIP(hipEventRecord(e0, 0));
		hipLaunchKernelGGL(k_probe_gather, dim3(blocks), dim3(256), 0, 0, table.data(), lines - 1, steps, stream, sink.data());
		PGX_HIP(hipEventRecord(e1, 0));
		PGX_HIP(hipEventSynchronize(e1));
		float t = 0;
		PGX_HIP(hipEventElapsedTime(&t, e0, e1));
		if (rep && (best == 0 || t < best))
			best = t;
	}
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	*ms = best;
	*lines_per_s = (double)blocks * 256 * steps * kInFlight / (best * 1e-3);
	return 0;
}
