// Spec pgx-blastn v2, S3d: low-complexity masking of the reads for seeding -- `blastn -dust "20 64 1"`, the default of the
// program the reference calls (README.md:96; Scripts/run_multi_blastn.pl:56).  The definition of symmetric DUST (Morgulis,
// Gertz, Schaffer, Agarwala, J Comput Biol 13 (2006)), restated by the checker in oracle/o_dust.c:
//   a read is overlapping triplets; an interval of l >= 2 consecutive triplets scores sum_t c_t (c_t - 1) / 2 over its
//   triplet values, divided by l - 1; an interval of at most 62 triplets (window 64) is PERFECT when its score exceeds 2.0
//   (level 20) and no sub-interval scores higher; every base of a perfect interval is masked.  Triplets with a letter other
//   than A C G T belong to no interval.  Scores are compared as exact fractions.
// The mask only removes SEEDS: what the seed stage reads is, per strand, one bit per read position i that says "the 28
// bases from i on touch no masked base" (`d_dustwin_f`, `d_dustwin_r`; 64 positions per word at the read's word offset).
//
// k_dust_mask     one lane per read: the dynamic programme over (first triplet a descending, last triplet b ascending)
//                 with the triplet counts, the row below and the current row (best sub-interval scores) in LDS
// k_dust_windows  one lane per (read, 64 positions): the window bits of both strands from the mask
#include "bitops.hpp"
#include "engine.hpp"

namespace pgx {

constexpr int kDustMaxT = 62, kDustLevel = 20;

struct DustLane {
	uint8_t cnt[64];
	uint32_t row[2][kDustMaxT + 2]; // score r | (triplets - 1) << 16; 0 = no score
};

__device__ __forceinline__ bool frac_gt(uint32_t a, uint32_t b) // a > b; "no score" is below every score
{
	const uint32_t aq = a >> 16, bq = b >> 16;
	if (aq == 0)
		return false;
	if (bq == 0)
		return true;
	return (a & 0xFFFFu) * bq > (b & 0xFFFFu) * aq;
}

__global__ __launch_bounds__(64) void k_dust_mask(const uint64_t *__restrict__ fwd, const uint64_t *__restrict__ amb,
						   const uint32_t *__restrict__ len, const uint32_t *__restrict__ woff, uint32_t n,
						   uint64_t *__restrict__ mask, uint8_t *__restrict__ any)
{
	__shared__ DustLane s_lane[64];
	DustLane &ld = s_lane[threadIdx.x];
	const uint32_t r = blockIdx.x * 64u + threadIdx.x;
	if (r >= n)
		return;
	const int L = (int)len[r], nt = L - 2;
	const uint64_t *rw = fwd + woff[r], *ra = amb ? amb + woff[r] : nullptr;
	uint64_t *mw = mask + woff[r];
	bool marked = false;
	if (nt < 2) {
		any[r] = 0;
		return;
	}
	// triplet value at position i, or -1 (6 bits of the packed read; ambiguity flags of its three letters)
	auto triplet = [&](int i) -> int {
		const uint64_t w = window64(rw, i);
		if (ra && (window64(ra, i) & 0x15ull))
			return -1;
		return (int)(((w & 3ull) << 4) | (((w >> 2) & 3ull) << 2) | ((w >> 4) & 3ull));
	};
	int below = 0;
	for (int k = 0; k < kDustMaxT + 2; k++)
		ld.row[0][k] = ld.row[1][k] = 0u;
	for (int a = nt - 1; a >= 0; a--) {
		for (int k = 0; k < 16; k++)
			reinterpret_cast<uint32_t *>(ld.cnt)[k] = 0u;
		uint32_t *rb = ld.row[below], *rc = ld.row[below ^ 1];
		uint32_t rsum = 0, left = 0u;
		int b = a;
		for (; b < nt && b - a < kDustMaxT; b++) {
			const int t = triplet(b);
			if (t < 0)
				break;
			rsum += ld.cnt[t]++;
			const uint32_t q = (uint32_t)(b - a);
			const uint32_t s = q ? (rsum | (q << 16)) : 0u;
			uint32_t sub = left;
			if (b > a && frac_gt(rb[b - a - 1], sub))
				sub = rb[b - a - 1];
			if (q && rsum * 10u > (uint32_t)kDustLevel * q && !frac_gt(sub, s)) {
				// mask bases a .. b + 2 (this lane's own words)
				marked = true;
				for (int k = a; k <= b + 2;) {
					const int w = k >> 6, lo = k & 63;
					const int hi = (b + 2 - (w << 6)) < 63 ? b + 2 - (w << 6) : 63;
					const uint64_t bits = (hi == 63 ? ~0ull : ((2ull << hi) - 1ull)) & (~0ull << lo);
					mw[w] |= bits;
					k = (w + 1) << 6;
				}
			}
			const uint32_t best = frac_gt(s, sub) ? s : sub;
			rc[b - a] = best;
			left = best;
		}
		for (int k = b - a; k <= kDustMaxT; k++)
			rc[k] = 0u;
		below ^= 1;
	}
	any[r] = marked ? 1 : 0;
}

// window bits of both strands: bit i of dustwin_f[read]: bases i .. i + 27 of the read hold no masked base (and i + 28 <= L);
// dustwin_r the same for the reverse-complement strand (its mask is the forward mask reversed)
__global__ void k_dust_windows(const uint64_t *__restrict__ mask, const uint8_t *__restrict__ any, const uint32_t *__restrict__ len,
			       const uint32_t *__restrict__ woff, uint32_t n, uint64_t *__restrict__ win_f, uint64_t *__restrict__ win_r)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n)
		return;
	const int L = (int)len[r];
	const uint32_t w0 = woff[r];
	const int nw = (L + 63) >> 6;
	const bool dirty = any[r] != 0;
	auto masked = [&](int i) { return (mask[w0 + (i >> 6)] >> (i & 63)) & 1ull; };
	for (int w = 0; w < nw; w++) {
		uint64_t f = 0, rv = 0;
		for (int k = 0; k < 64; k++) {
			const int i = (w << 6) + k;
			if (i + kWord > L)
				break;
			if (!dirty) {
				f |= 1ull << k;
				rv |= 1ull << k;
				continue;
			}
			bool cf = true, cr = true;
			for (int j = 0; j < kWord && (cf || cr); j++) {
				cf = cf && !masked(i + j);
				cr = cr && !masked(L - 1 - (i + j));
			}
			f |= cf ? 1ull << k : 0ull;
			rv |= cr ? 1ull << k : 0ull;
		}
		win_f[w0 + w] = f;
		win_r[w0 + w] = rv;
	}
}

// the DUST window bits of a batch (reads_finish); the batch keeps them whether or not a search uses them (`-dust no`)
int reads_dust(pgx_reads *rd)
{
	const size_t n = (size_t)rd->n;
	rd->has_dust = false;
	if (n == 0)
		return 0;
	DevBuf<uint64_t> d_mask;
	DevBuf<uint8_t> &d_any = rd->d_dust_any;
	PGX_TRY(d_mask.alloc((size_t)rd->n_words + 24, 0, 0, true));
	PGX_TRY(d_any.alloc(n));
	hipLaunchKernelGGL(k_dust_mask, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, rd->d_fwd.data(),
			   rd->has_amb ? rd->d_fwd_amb.data() : (const uint64_t *)nullptr, rd->d_len.data(), rd->d_woff.data(), (uint32_t)n,
			   d_mask.data(), d_any.data());
	PGX_HIP(hipGetLastError());
	std::vector<uint8_t> &h_any = rd->h_read_dust;
	h_any.resize(n);
	PGX_TRY(d_any.download(h_any.data(), n));
	bool some = false;
	for (size_t i = 0; i < n && !some; i++)
		some = h_any[i] != 0;
	if (!some) {
		h_any.clear();
		return 0; // no read of the batch has a masked base: the seed stage runs as without DUST
	}
	PGX_TRY(rd->d_dustwin_f.alloc((size_t)rd->n_words + 24, 0, 0, true));
	PGX_TRY(rd->d_dustwin_r.alloc((size_t)rd->n_words + 24, 0, 0, true));
	hipLaunchKernelGGL(k_dust_windows, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, 0, d_mask.data(), d_any.data(), rd->d_len.data(),
			   rd->d_woff.data(), (uint32_t)n, rd->d_dustwin_f.data(), rd->d_dustwin_r.data());
	PGX_HIP(hipGetLastError());
	PGX_HIP(hipDeviceSynchronize());
	rd->has_dust = true;
	return 0;
}

} // namespace pgx
