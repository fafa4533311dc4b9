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

// ---------------------------------------------------------------------------------------------------------------------
// The issue roof of the gapped stage's instruction mix.  k_gapped_fast is integer VALU (v_max3_i32, adds, compares,
// v_alignbit, v_ffbl), scalar bookkeeping and LDS reads on letters held per lane.  What a SIMD delivers for each such
// instruction at 1 .. 8 resident wavefronts is measured here, not assumed: every wavefront runs `iters` trips of 64
// vector instructions of ONE kind over 8 independent registers (instruction j works on register j mod 8, so a dependent
// instruction is 8 issues away).  One workgroup of 256 x W threads per CU (100 KB of dynamic LDS keeps a second one off
// the CU), so W wavefronts sit on each SIMD.  Cycles come from s_memtime of wave 0.
namespace pgx {

static const char *const kIssueKinds[] = {
	"v_add_u32",
	"v_max3_i32",
	"v_alignbit_b32",
	"v_ffbl_b32",
	"v_and_b32",
	"v_or_b32",
	"v_xor_b32",
	"v_lshlrev_b32",
	"v_lshrrev_b32",
	"v_ashrrev_i32",
	"v_max_i32",
	"v_min_u32",
	"v_sub_u32",
	"v_sub_u32 const",
	"v_add_u32 sgpr",
	"v_mov_b32",
	"v_cndmask vcc",
	"v_cmp_gt_i32 vcc",
	"v_cmp_gt_i32 e64",
	"v_bfe_u32",
	"v_bfe_u32 consts",
	"v_and_or_b32",
	"v_add3_u32",
	"v_lshl_add_u32",
	"v_lshl_or_b32",
	"v_add_u32 e64",
	"v_med3_i32",
	"v_max3 2 consts",
	"v_bfi_b32",
	"v_perm_b32",
	"v_mul_u32_u24",
	"v_mad_u32_u24",
	"v_mul_lo_u32",
	"v_bcnt_u32_b32",
	"v_ffbh_u32",
	"v_bfrev_b32",
	"v_pk_max_i16",
	"v_pk_add_u16",
	"v_pk_sub_i16",
	"v_cndmask e64 sgpr",
	"v_mov_b32 dpp shr",
	"v_add_u32 dpp",
	"v_add_u32 sdwa",
	"v_max_i32 + s_add per 2",
	"v_add_u32 + ds_read per 4",
	"max3+add dependent pair"
};
constexpr int kIssueKindCount = 46;

template <int KIND>
__global__ __launch_bounds__(1024) void k_probe_issue(uint32_t iters, unsigned long long *__restrict__ cycles, uint32_t *__restrict__ sink)
{
	extern __shared__ uint32_t pad[];
	int a[8];
#pragma unroll
	for (int k = 0; k < 8; k++)
		a[k] = threadIdx.x + k;
	const int b = (int)blockIdx.x, c = 3;
	int s0 = (int)blockIdx.x;
	uint32_t l0 = 0;
	const uint32_t laddr = (threadIdx.x & 63) * 4u;
	pad[threadIdx.x & 63] = threadIdx.x;
	__syncthreads();
	asm volatile("s_mov_b64 s[22:23], 0x5555" ::: "s22", "s23");
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
		for (int j = 0; j < 64; j++) {
			int &x = a[j & 7];
			if constexpr (KIND == 0)
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 1)
				asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
			if constexpr (KIND == 2)
				asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
			if constexpr (KIND == 3)
				asm volatile("v_ffbl_b32 %0, %0" : "+v"(x));
			if constexpr (KIND == 4)
				asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 5)
				asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 6)
				asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 7)
				asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x) : "v"(c));
			if constexpr (KIND == 8)
				asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x) : "v"(c));
			if constexpr (KIND == 9)
				asm volatile("v_ashrrev_i32 %0, 16, %0" : "+v"(x));
			if constexpr (KIND == 10)
				asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 11)
				asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 12)
				asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 13)
				asm volatile("v_sub_u32 %0, 5, %0" : "+v"(x));
			if constexpr (KIND == 14)
				asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "s"(s0));
			if constexpr (KIND == 15)
				asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 16)
				asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c) : "vcc");
			if constexpr (KIND == 17)
				asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(x), "v"(c) : "vcc");
			if constexpr (KIND == 18)
				asm volatile("v_cmp_gt_i32 s[20:21], %0, %1" : : "v"(x), "v"(c) : "s20", "s21");
			if constexpr (KIND == 19)
				asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
			if constexpr (KIND == 20)
				asm volatile("v_bfe_u32 %0, %0, 8, 6" : "+v"(x));
			if constexpr (KIND == 21)
				asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
			if constexpr (KIND == 22)
				asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
			if constexpr (KIND == 23)
				asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 24)
				asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 25)
				asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 26)
				asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
			if constexpr (KIND == 27)
				asm volatile("v_max3_i32 %0, %0, 0, 1" : "+v"(x));
			if constexpr (KIND == 28)
				asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
			if constexpr (KIND == 29)
				asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
			if constexpr (KIND == 30)
				asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 31)
				asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
			if constexpr (KIND == 32)
				asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 33)
				asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 34)
				asm volatile("v_ffbh_u32 %0, %0" : "+v"(x));
			if constexpr (KIND == 35)
				asm volatile("v_bfrev_b32 %0, %0" : "+v"(x));
			if constexpr (KIND == 36)
				asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 37)
				asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 38)
				asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(x) : "v"(c));
			if constexpr (KIND == 39)
				asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(x) : "v"(c));
			if constexpr (KIND == 40)
				asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(c));
			if constexpr (KIND == 41)
				asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(c));
			if constexpr (KIND == 42)
				asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(x) : "v"(c));
			if constexpr (KIND == 43) {
				asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(c));
				if (j & 1)
					asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");
			}
			if constexpr (KIND == 44) {
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(c));
				if ((j & 3) == 3)
					asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(3)" : "=v"(l0) : "v"(laddr + 256u * ((j >> 2) & 3)));
			}
			if constexpr (KIND == 45) {
				int &y = a[(j >> 1) & 7];
				if (j & 1)
					asm volatile("v_add_u32 %0, %0, %1" : "+v"(y) : "v"(c));
				else
					asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(y) : "v"(b), "v"(c));
			}
		}
	}
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if (blockIdx.x == 0 && threadIdx.x == 0)
		cycles[0] = t1 - t0;
	if ((a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7] ^ s0 ^ (int)l0) == 0x7FEDCBA9)
		sink[0] = 1;
}

template <int KIND> static int launch_probe_issue(int kind, unsigned blocks, unsigned threads, size_t lds, uint32_t iters, unsigned long long *cyc, uint32_t *sink)
{
	if constexpr (KIND < kIssueKindCount) {
		if (kind != KIND)
			return launch_probe_issue<KIND + 1>(kind, blocks, threads, lds, iters, cyc, sink);
		PGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe_issue<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		hipLaunchKernelGGL(k_probe_issue<KIND>, dim3(blocks), dim3(threads), lds, 0, iters, cyc, sink);
		PGX_HIP(hipGetLastError());
		return 0;
	} else {
		return fail(PGX_E_ARG, "pgx_probe_issue: no such kind");
	}
}

} // namespace pgx

extern "C" const char *pgx_probe_issue_name(int kind)
{
	return kind >= 0 && kind < kIssueKindCount ? kIssueKinds[kind] : nullptr;
}

// waves_per_simd in 1..8, kind 0 .. (pgx_probe_issue_name(kind) != NULL).
// out[0] = vector wave-instructions per second per SIMD, out[1] = shader cycles per vector instruction of ONE wavefront
// (s_memtime), out[2] = kernel milliseconds, out[3] = shader clock in Hz (wave 0's cycles over its kernel's time; exact
// when every wavefront runs the whole time, i.e. at 1 and 2 wavefronts per SIMD)
extern "C" int pgx_probe_issue(int waves_per_simd, int kind, double *out)
{
	if (!out || waves_per_simd < 1 || waves_per_simd > 8 || kind < 0 || kind >= kIssueKindCount)
		return fail(PGX_E_ARG, "pgx_probe_issue: bad argument");
	PGX_TRY(require_device());
	DevBuf<unsigned long long> cyc;
	DevBuf<uint32_t> sink;
	PGX_TRY(cyc.alloc(1, 0, 0, true));
	PGX_TRY(sink.alloc(1, 0, 0, true));
	// 4 W wavefronts per CU: one workgroup of 256 W threads up to W = 4, two of 128 W threads beyond (a workgroup's
	// wavefronts go round the CU's four SIMDs)
	const int per_cu = waves_per_simd > 4 ? 2 : 1;
	const unsigned threads = per_cu == 1 ? 256u * waves_per_simd : 128u * waves_per_simd;
	const size_t lds = per_cu == 1 ? 100 * 1024 : 60 * 1024;
	const uint32_t iters = 10000;
	const unsigned blocks = 256 * per_cu;
	hipEvent_t e0, e1;
	PGX_HIP(hipEventCreate(&e0));
	PGX_HIP(hipEventCreate(&e1));
	float best = 0;
	for (int rep = 0; rep < 3; rep++) {
		PGX_HIP(hipEventRecord(e0, 0));
		PGX_TRY(launch_probe_issue<0>(kind, blocks, threads, lds, iters, cyc.data(), sink.data()));
		PGX_HIP(hipEventRecord(e1, 0));
		PGX_HIP(hipEventSynchronize(e1));
		float t = 0;
		PGX_HIP(hipEventElapsedTime(&t, e0, e1));
		if (rep && (best == 0 || t < best))
			best = t;
	}
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	unsigned long long c = 0;
	PGX_HIP(hipMemcpy(&c, cyc.data(), sizeof c, hipMemcpyDeviceToHost));
	const double valu_per_wave = (double)iters * 64;
	out[0] = valu_per_wave * waves_per_simd / (best * 1e-3);
	out[1] = (double)c / valu_per_wave;
	out[2] = best;
	out[3] = (double)c / (best * 1e-3);
	return 0;
}
