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
// k_dust_trigger<false>  one lane per read, linear: the published algorithm's own bookkeeping (window of the last 62
//                 triplets with its pair count r_w; its longest suffix in which no triplet occurs more than 4 times, of L
//                 triplets) and its test "10 r_w > 20 L", without which that algorithm never looks for a perfect interval
//                 ending at the position.  93 % of random 150-base reads never pass it.  Triplets come from 64-bit
//                 registers that move along the read; the suffix of ONE lane at a time is shrunk by the whole wavefront
//                 (ballot for the earliest copy, a 64-bin histogram for the counters).
// k_dust_trigger<true>   the listed reads again, packed: at every position that passes, the algorithm's walk over the
//                 suffixes longer than that suffix; a read is kept when one scores above the level -- an interval above
//                 the level exists exactly when a perfect one does (its best sub-interval), so these are the reads with a
//                 masked base (the checker's fuzz of both tests against the definition: oracle/fuzz_dust.c) -- with the
//                 first and last such position.
// k_dust_mask     one lane per KEPT read: the definition itself, a dynamic programme over (first triplet a descending,
//                 last triplet b ascending) restricted to intervals that end at or before the last such position and
//                 start at most 61 triplets before the first; triplet counts and one row of best sub-interval scores in LDS
// k_dust_windows  one lane per (read, 64 positions): the window bits of both strands from the mask
#include "bitops.hpp"
#include "engine.hpp"

namespace pgx {

constexpr int kDustMaxT = 62, kDustLevel = 20;

struct DustLane {
	uint8_t cnt[64];
	uint32_t row[kDustMaxT + 2]; // score r | (triplets - 1) << 16; 0 = no score
	uint32_t pad;                // 81 words per lane (odd stride: see TrigLane)
};
static_assert(sizeof(DustLane) / 4 % 2 == 1, "odd word stride");

__device__ __forceinline__ bool frac_gt(uint32_t a, uint32_t b) // a > b; "no score" is below every score
{
	const uint32_t aq = a >> 16, bq = b >> 16;
	if (aq == 0)
		return false;
	if (bq == 0)
		return true;
	return (a & 0xFFFFu) * bq > (b & 0xFFFFu) * aq;
}

// triplet value at position i of a packed read, or -1 (6 bits of the packed read; ambiguity flags of its three letters)
__device__ __forceinline__ int dust_triplet(const uint64_t *rw, const uint64_t *ra, int i)
{
	const uint64_t w = window64(rw, i);
	if (ra && (window64(ra, i) & 0x15ull))
		return -1;
	return (int)(((w & 3ull) << 4) | (((w >> 2) & 3ull) << 2) | ((w >> 4) & 3ull));
}

// LDS written by one lane, read by another of the same wavefront (LDS operations of a wavefront complete in order)
__device__ __forceinline__ void dust_wave_sync()
{
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
}

// (the first pass never walks the longer suffixes: without `ct` its wavefront needs 8.4 KB of LDS instead of 12.5 -- 4 to a
// SIMD instead of 3 for a kernel that waits on its own LDS counters)
template <bool CONFIRM> struct TrigLaneT {
	uint8_t cw[64], cv[64], ct[CONFIRM ? 64 : 4];
	uint32_t pad[CONFIRM ? 1 : 2]; // an odd word stride per lane: an even one put the 64 lanes' counters on two LDS banks (32-way conflicts)
};
static_assert(sizeof(TrigLaneT<true>) / 4 % 2 == 1 && sizeof(TrigLaneT<false>) / 4 % 2 == 1, "odd word stride");

// the trigger works on raw 6-bit triplet codes (any one-to-one naming of the 64 triplets counts the same pairs)
__device__ __forceinline__ int dust_tid(const uint64_t *rw, const uint64_t *ra, int i)
{
	if (ra && (window64(ra, i) & 0x15ull))
		return -1;
	return (int)(window64(rw, i) & 63ull);
}

// CONFIRM = false: every read of the batch, the algorithm's own test only (a read that never passes it is done);
// CONFIRM = true: the reads the first pass listed, packed 64 to a wavefront, with the walk over the longer suffixes at every
// position that passes (kept out of the first pass: one lane walking there held up the other 63)
template <bool CONFIRM>
__global__ __launch_bounds__(64) void k_dust_trigger(const uint64_t *__restrict__ fwd, const uint64_t *__restrict__ amb,
						      const uint32_t *__restrict__ len, const uint32_t *__restrict__ woff, uint32_t n,
						      const uint32_t *__restrict__ in_list, const uint32_t *__restrict__ n_in,
						      uint32_t *__restrict__ list, uint2 *__restrict__ range, uint32_t *__restrict__ n_list)
{
	using TrigLane = TrigLaneT<CONFIRM>;
	__shared__ TrigLane s_lane[64];
	__shared__ uint32_t s_hist[64];
	TrigLane &ld = s_lane[threadIdx.x];
	const uint32_t at0 = blockIdx.x * 64u + threadIdx.x;
	// (every lane runs the loop: the wavefront shrinks suffixes together; a lane past the end has no read)
	const bool has_read = at0 < (CONFIRM ? *n_in : n);
	const uint32_t r = has_read ? (CONFIRM ? in_list[at0] : at0) : 0u;
	const int nt = (int)len[r] - 2;
	const uint64_t *rw = fwd + woff[r], *ra = amb ? amb + woff[r] : nullptr;
	uint32_t *c32 = reinterpret_cast<uint32_t *>(&ld);
	for (int k = 0; k < 32; k++)
		c32[k] = 0u;
	int first = -1, last = -1;
	int size = 0, L = 0, rw_pairs = 0, rv_pairs = 0; // the window is the `size` triplets that end at the current one
	// the entering and the leaving triplet come from two 64-bit registers that hold the next 32 letters each and move on
	// by one letter per position (refilled every 16 positions); only the rare steps below go back to memory
	uint64_t in_w = 0, out_w = 0;
	int out_pos = -1;
	int nt_wave = has_read ? nt : 0; // the loop is the wavefront's: lanes past their read's end idle but still help
	for (int sh = 1; sh < 64; sh <<= 1) {
		const int o = __shfl_xor(nt_wave, sh);
		nt_wave = o > nt_wave ? o : nt_wave;
	}
	for (int b = 0; b < nt_wave; b++) {
		int cv_now = 0; // the suffix's count of the entering triplet, after it entered
		int t = -1;
		const bool mine = has_read && b < nt;
		if (mine) {
			if ((b & 15) == 0)
				in_w = window64(rw, b);
			t = (int)(in_w & 63ull);
			in_w >>= 2;
			if (ra && (window64(ra, b) & 0x15ull))
				t = -1;
			if (t < 0) { // a letter that is no base: no interval crosses it
				for (int k = 0; k < 32; k++)
					c32[k] = 0u;
				size = L = rw_pairs = rv_pairs = 0;
			}
		}
		if (t >= 0) {
		if (size >= kDustMaxT) {
			// the oldest triplet leaves: position b - 62 (the window was full, so it is 62 triplets behind)
			const int ob = b - kDustMaxT;
			if (ob != out_pos || (ob & 15) == 0) {
				out_w = window64(rw, ob);
				out_pos = ob;
			}
			const int s0 = (int)(out_w & 63ull);
			out_w >>= 2;
			out_pos++;
			// the leaving and the entering triplet in one go: four counters read together, then written (four dependent
			// read-modify-writes of LDS bytes per position were the rest of this kernel's time).  Same triplet leaving and
			// entering: the counts and both pair sums end where they were.
			const bool in_suffix = L > size - 1;
			if (s0 != t) {
				const int cw_s = ld.cw[s0], cv_s = ld.cv[s0], cw_t = ld.cw[t], cv_t = ld.cv[t];
				ld.cw[s0] = (uint8_t)(cw_s - 1);
				rw_pairs += cw_t - (cw_s - 1);
				ld.cw[t] = (uint8_t)(cw_t + 1);
				if (in_suffix) {
					ld.cv[s0] = (uint8_t)(cv_s - 1);
					rv_pairs -= cv_s - 1;
				}
				rv_pairs += cv_t;
				cv_now = cv_t + 1;
				ld.cv[t] = (uint8_t)cv_now;
			} else if (in_suffix) {
				cv_now = ld.cv[t]; // (one out, one in)
			} else {
				// the leaving copy lies before the suffix: only the window's count is a wash, the suffix gains one
				const int cv_t = ld.cv[t];
				rv_pairs += cv_t;
				cv_now = cv_t + 1;
				ld.cv[t] = (uint8_t)cv_now;
			}
			if (!in_suffix)
				L++; // (size stays: one left, one entered)
		} else {
			size++;
			L++;
			const int cw_t = ld.cw[t], cv_t = ld.cv[t];
			rw_pairs += cw_t;
			rv_pairs += cv_t;
			ld.cw[t] = (uint8_t)(cw_t + 1);
			cv_now = cv_t + 1;
			ld.cv[t] = (uint8_t)cv_now;
		}
		}
		dust_wave_sync();
		// The suffix shrinks past the earliest copy of t when t now occurs more than 4 times in it.  Some lane of the 64 needs
		// that at most positions, and a lane's own loop over up to 61 triplets (a counter read-modify-write each) held up
		// the other 63: the WAVEFRONT shrinks one lane's suffix at a time -- lane j looks at the suffix's j-th triplet, a
		// ballot finds the earliest copy, the removed triplets go into a 64-bin histogram, lane v settles triplet v's
		// counter and its share of the pair sum (v leaves m times from a count of c: m c - m (m + 1) / 2 pairs fewer).
		for (unsigned long long need = __ballot(cv_now * 10 > 2 * kDustLevel); need; need &= need - 1ull) {
			const int x = __ffsll((unsigned long long)need) - 1, lane = (int)(threadIdx.x & 63);
			const int xL = __shfl(L, x), xt = __shfl(t, x), xsp = b - xL + 1;
			const uint64_t *xrw = reinterpret_cast<const uint64_t *>(
				((unsigned long long)(uint32_t)__shfl((int)((uintptr_t)rw >> 32), x) << 32) | (uint32_t)__shfl((int)(uintptr_t)rw, x));
			const int val = lane < xL ? (int)(window64(xrw, xsp + lane) & 63ull) : -1;
			const unsigned long long hit = __ballot(val == xt);
			const int k = __ffsll((unsigned long long)hit) - 1; // the earliest copy (the entering one is in the suffix: there is one)
			s_hist[lane] = 0u;
			dust_wave_sync();
			if (lane <= k)
				atomicAdd(&s_hist[val], 1u);
			dust_wave_sync();
			const int m = (int)s_hist[lane];
			int term = 0;
			if (m) {
				const int c = s_lane[x].cv[lane];
				term = m * c - m * (m + 1) / 2;
				s_lane[x].cv[lane] = (uint8_t)(c - m);
			}
			for (int sh = 1; sh < 64; sh <<= 1)
				term += __shfl_xor(term, sh);
			if (lane == x) {
				rv_pairs -= term;
				L -= k + 1;
			}
			dust_wave_sync();
		}
		if (!CONFIRM && t >= 0 && rw_pairs * 10 > L * kDustLevel) {
			first = first < 0 ? b : first;
			last = b;
		}
		if constexpr (CONFIRM) {
		if (t >= 0 && rw_pairs * 10 > L * kDustLevel) {
			// the algorithm would now look at the suffixes LONGER than that suffix, longest last; an interval that scores
			// above the level exists in the read exactly when one of these does somewhere (its best sub-interval is
			// perfect), so this decides whether the read has a masked base at all
			uint32_t *t32 = reinterpret_cast<uint32_t *>(ld.ct);
			const uint32_t *v32 = reinterpret_cast<const uint32_t *>(ld.cv);
			for (int k = 0; k < 16; k++)
				t32[k] = v32[k];
			int rr = rv_pairs;
			int q0 = -64; // the walk goes DOWN the read: 30 triplets per 64-bit window of letters, taken with a shift
			uint64_t qw = 0;
			for (int k = size - L - 1; k >= 0; k--) {
				const int q = b - size + 1 + k;
				if (q < q0 || q > q0 + 29) {
					q0 = q >= 29 ? q - 29 : 0;
					qw = window64(rw, q0);
				}
				const int tt = (int)((qw >> (2 * (q - q0))) & 63ull); // (inside the window every triplet is one of bases)
				rr += ld.ct[tt]++;
				if (rr * 10 > kDustLevel * (size - k - 1)) {
					first = first < 0 ? b : first;
					last = b;
					break;
				}
			}
		}
		}
	}
	// one atomic per wavefront (a single counter takes ~90 M atomics a second: one per listed read was most of this kernel)
	const unsigned long long listed = __ballot(first >= 0);
	if (listed) {
		uint32_t base = 0;
		const int leader = __ffsll((unsigned long long)listed) - 1;
		if ((int)(threadIdx.x & 63) == leader)
			base = atomicAdd(n_list, (uint32_t)__popcll(listed));
		base = __shfl(base, leader);
		if (first >= 0) {
			const uint32_t at = base + (uint32_t)__popcll(listed & ((1ull << (threadIdx.x & 63)) - 1ull));
			list[at] = r;
			range[at] = make_uint2((uint32_t)first, (uint32_t)last);
		}
	}
}

__global__ __launch_bounds__(64) void k_dust_mask(const uint64_t *__restrict__ fwd, const uint64_t *__restrict__ amb,
						   const uint32_t *__restrict__ len, const uint32_t *__restrict__ woff,
						   const uint32_t *__restrict__ list, const uint2 *__restrict__ range, const uint32_t *__restrict__ n_list,
						   uint64_t *__restrict__ mask, uint8_t *__restrict__ any)
{
	__shared__ DustLane s_lane[64];
	DustLane &ld = s_lane[threadIdx.x];
	const uint32_t at = blockIdx.x * 64u + threadIdx.x;
	if (at >= *n_list)
		return;
	const uint32_t r = list[at];
	const int L = (int)len[r], nt = L - 2;
	const uint64_t *rw = fwd + woff[r], *ra = amb ? amb + woff[r] : nullptr;
	uint64_t *mw = mask + woff[r];
	bool marked = false;
	// perfect intervals end at a position that passed the trigger: [b_first, b_last]; they hold at most 62 triplets
	const int b_hi = (int)range[at].y, a_lo = (int)range[at].x - (kDustMaxT - 1) > 0 ? (int)range[at].x - (kDustMaxT - 1) : 0;
	for (int k = 0; k < kDustMaxT + 2; k++)
		ld.row[k] = 0u;
	// one row, updated in place: before the step for (a, b) row[b - a] holds the best of [a + 1, b + 1] and row[b - a - 1]
	// the best of [a + 1, b] (the row below); the step leaves the best of [a, b] in row[b - a]
	for (int a = b_hi < nt - 1 ? b_hi : nt - 1; a >= a_lo; a--) {
		for (int k = 0; k < 16; k++)
			reinterpret_cast<uint32_t *>(ld.cnt)[k] = 0u;
		uint32_t rsum = 0, left = 0u, below_prev = 0u; // below_prev = row below at index b - a - 1 (saved before it is overwritten)
		int b = a;
		uint64_t fw = 0; // letters from b on, one letter further per step, refilled every 16 (any one-to-one naming of the
				 // triplets serves the counters)
		for (; b <= b_hi && b < nt && b - a < kDustMaxT; b++) {
			if (b == a || ((b - a) & 15) == 0)
				fw = window64(rw, b);
			const int t = ra && (window64(ra, b) & 0x15ull) ? -1 : (int)(fw & 63ull);
			fw >>= 2;
			if (t < 0)
				break;
			rsum += ld.cnt[t]++;
			const uint32_t q = (uint32_t)(b - a);
			const uint32_t s = q ? (rsum | (q << 16)) : 0u;
			const uint32_t old_here = ld.row[b - a]; // row below at index b - a: the next step's `below_prev`
			uint32_t sub = left;
			if (b > a && frac_gt(below_prev, sub))
				sub = below_prev;
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
			ld.row[b - a] = best;
			left = best;
			below_prev = old_here;
		}
		for (int k = b - a; k <= kDustMaxT; k++)
			ld.row[k] = 0u;
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
	const int n_valid = L - kWord + 1; // window positions 0 .. n_valid - 1
	const bool dirty = any[r] != 0;
	auto valid_bits = [&](int w) -> uint64_t { // positions of word w below n_valid
		const int left = n_valid - (w << 6);
		return left <= 0 ? 0ull : (left >= 64 ? ~0ull : ((1ull << left) - 1ull));
	};
	for (int w = 0; w < nw; w++) {
		uint64_t f = valid_bits(w);
		if (dirty) {
			// a window is spoilt by a masked base at any of its 28 positions: OR of the mask moved down by 0 .. 27, in log
			// steps on the 128 bits that start at this word (2, 4, 8, 16 wide, then 16 + 8 + 4)
			uint64_t lo = mask[w0 + w], hi = w + 1 < nw ? mask[w0 + w + 1] : 0ull;
			auto shr = [](uint64_t &a, uint64_t &b, int k) { // (a, b) |= (a, b) >> k
				a |= (a >> k) | (b << (64 - k));
				b |= b >> k;
			};
			shr(lo, hi, 1);
			const uint64_t l2 = lo, h2 = hi; // covers 2
			shr(lo, hi, 2);
			const uint64_t l4 = lo, h4 = hi; // covers 4
			shr(lo, hi, 4);
			const uint64_t l8 = lo, h8 = hi; // covers 8
			shr(lo, hi, 8); // covers 16
			(void)l2;
			(void)h2;
			const uint64_t d = lo | ((l8 >> 16) | (h8 << 48)) | ((l4 >> 24) | (h4 << 40)); // 16 + 8 + 4 = 28
			f &= ~d;
		}
		win_f[w0 + w] = f;
	}
	// the reverse-complement strand: its window i is the forward window n_valid - 1 - i
	for (int w = 0; w < nw; w++) {
		uint64_t rv = valid_bits(w);
		if (dirty) {
			// forward positions n_valid - 64 w - 64 .. n_valid - 64 w - 1, reversed
			const int s0 = n_valid - (w << 6) - 64;
			uint64_t span;
			if (s0 >= 0) {
				const int wi = s0 >> 6, sh = s0 & 63;
				const uint64_t a = win_f[w0 + wi], b = (sh && wi + 1 < nw) ? win_f[w0 + wi + 1] : 0ull;
				span = sh ? (a >> sh) | (b << (64 - sh)) : a;
			} else if (s0 > -64) {
				span = win_f[w0] << (-s0);
			} else {
				span = 0ull;
			}
			rv = __brevll(span);
		}
		win_r[w0 + w] = rv;
	}
}

// The second trigger pass and the mask kernel run one lane per LISTED read, and a lane's work grows with the stretch of
// positions that passed the pass before (last - first): the suffix walks there, the rows of the dynamic programme here.  A
// wavefront runs as long as its longest lane, so the lists are ordered by that length first, longest first (a counting sort,
// 256 buckets, order inside a bucket left open: the kernels behind treat every read on its own).  k_dust_mask: 6.1 -> see
// DESIGN section 7.
__global__ __launch_bounds__(256) void k_dust_order_hist(const uint2 *__restrict__ range, const uint32_t *__restrict__ n_ptr, uint32_t *__restrict__ hist)
{
	__shared__ uint32_t h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t n = *n_ptr;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
		const uint2 r = range[i];
		const uint32_t k = r.y - r.x;
		atomicAdd(&h[k < 255u ? k : 255u], 1u);
	}
	__syncthreads();
	if (h[threadIdx.x])
		atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// hist[0 .. 255] -> hist[256 + k] = entries with a longer range than k (where bucket k starts)
__global__ __launch_bounds__(256) void k_dust_order_bases(uint32_t *__restrict__ hist)
{
	if (threadIdx.x == 0) {
		uint32_t run = 0;
		for (int k = 255; k >= 0; k--) {
			hist[256 + k] = run;
			run += hist[k];
		}
	}
}

__global__ __launch_bounds__(256) void k_dust_order_scatter(const uint32_t *__restrict__ list, const uint2 *__restrict__ range,
							    const uint32_t *__restrict__ n_ptr, uint32_t *__restrict__ hist, uint32_t *__restrict__ list_out,
							    uint2 *__restrict__ range_out)
{
	// per tile of 2 048 entries: bucket counts in LDS, ONE global atomic per bucket and tile (most entries share a handful of
	// buckets, and a single address takes ~90 M atomics a second), places inside the tile from LDS atomics
	__shared__ uint32_t h[256], at[256];
	const uint32_t n = *n_ptr;
	for (uint32_t t0 = blockIdx.x * 2048u; t0 < n; t0 += gridDim.x * 2048u) {
		h[threadIdx.x] = 0;
		__syncthreads();
		uint2 r[8];
		uint32_t k[8];
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const uint32_t i = t0 + q * 256u + threadIdx.x;
			r[q] = i < n ? range[i] : make_uint2(0u, 0u);
			k[q] = r[q].y - r[q].x;
			k[q] = k[q] < 255u ? k[q] : 255u;
			if (i < n)
				atomicAdd(&h[k[q]], 1u);
		}
		__syncthreads();
		at[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&hist[256 + threadIdx.x], h[threadIdx.x]) : 0u;
		__syncthreads();
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const uint32_t i = t0 + q * 256u + threadIdx.x;
			if (i < n) {
				const uint32_t o = atomicAdd(&at[k[q]], 1u);
				list_out[o] = list[i];
				range_out[o] = r[q];
			}
		}
		__syncthreads();
	}
}

// list / range (n entries, count on the device) -> list_s / range_s in the order above
static int dust_order(DustBufs &b, const uint32_t *list, const uint2 *range, const uint32_t *n_ptr, size_t n_max, hipStream_t stream)
{
	if (n_max == 0)
		return 0;
	PGX_HIP(hipMemsetAsync(b.hist.data(), 0, 512 * sizeof(uint32_t), stream));
	const unsigned grid = (unsigned)std::min<size_t>((n_max + 2047) / 2048, 256 * 8);
	hipLaunchKernelGGL(k_dust_order_hist, dim3(grid), dim3(256), 0, stream, range, n_ptr, b.hist.data());
	hipLaunchKernelGGL(k_dust_order_bases, dim3(1), dim3(64), 0, stream, b.hist.data());
	hipLaunchKernelGGL(k_dust_order_scatter, dim3(grid), dim3(256), 0, stream, list, range, n_ptr, b.hist.data(), b.list_s.data(),
			   b.range_s.data());
	PGX_HIP(hipGetLastError());
	return 0;
}

// buffers of a pass over a batch of n reads, n_mask window words; `listed` = what the second list needs (0: not known yet)
static int dust_bufs_ensure(DustBufs &b, size_t n, size_t n_mask, size_t listed)
{
	PGX_TRY(b.mask.ensure(n_mask));
	PGX_TRY(b.any.ensure(n));
	PGX_TRY(b.list.ensure(n));
	PGX_TRY(b.range.ensure(n));
	PGX_TRY(b.list_s.ensure(n));
	PGX_TRY(b.range_s.ensure(n));
	PGX_TRY(b.hist.ensure(512));
	PGX_TRY(b.n.ensure(2));
	if (listed)
		PGX_TRY(b.list2.ensure(listed));
	return 0;
}

// the DUST window bits of a batch (reads_finish); the batch keeps them whether or not a search uses them (`-dust no`)
int reads_dust(pgx_reads *rd)
{
	const size_t n = (size_t)rd->n;
	rd->has_dust = false;
	if (n == 0)
		return 0;
	DustBufs &b = rd->dustb;
	const size_t n_mask = (size_t)rd->n_words + 24;
	PGX_TRY(dust_bufs_ensure(b, n, n_mask, 0));
	PGX_HIP(hipMemsetAsync(b.mask.data(), 0, n_mask * sizeof(uint64_t), 0));
	PGX_HIP(hipMemsetAsync(b.any.data(), 0, n, 0));
	PGX_HIP(hipMemsetAsync(b.n.data(), 0, 2 * sizeof(uint32_t), 0));
	const uint64_t *amb = rd->has_amb ? rd->d_fwd_amb.data() : (const uint64_t *)nullptr;
	hipLaunchKernelGGL(k_dust_trigger<false>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, rd->d_fwd.data(), amb, rd->d_len.data(),
			   rd->d_woff.data(), (uint32_t)n, (const uint32_t *)nullptr, (const uint32_t *)nullptr, b.list.data(), b.range.data(),
			   b.n.data());
	PGX_HIP(hipGetLastError());
	uint32_t n_listed[2] = { 0, 0 };
	PGX_TRY(b.n.download(n_listed, 1));
	if (n_listed[0]) {
		PGX_TRY(b.list2.ensure(n_listed[0]));
		PGX_TRY(dust_order(b, b.list.data(), b.range.data(), b.n.data(), n_listed[0], 0));
		hipLaunchKernelGGL(k_dust_trigger<true>, dim3((unsigned)((n_listed[0] + 63) / 64)), dim3(64), 0, 0, rd->d_fwd.data(), amb,
				   rd->d_len.data(), rd->d_woff.data(), (uint32_t)n, b.list_s.data(), b.n.data(), b.list2.data(), b.range.data(),
				   b.n.data() + 1);
		PGX_HIP(hipGetLastError());
		PGX_TRY(b.n.download(n_listed, 2));
	}
	if (n_listed[1]) {
		PGX_TRY(dust_order(b, b.list2.data(), b.range.data(), b.n.data() + 1, n_listed[1], 0));
		hipLaunchKernelGGL(k_dust_mask, dim3((unsigned)((n_listed[1] + 63) / 64)), dim3(64), 0, 0, rd->d_fwd.data(), amb, rd->d_len.data(),
				   rd->d_woff.data(), b.list_s.data(), b.range_s.data(), b.n.data() + 1, b.mask.data(), b.any.data());
		PGX_HIP(hipGetLastError());
	}
	std::vector<uint8_t> &h_any = rd->h_read_dust;
	h_any.resize(n);
	PGX_TRY(b.any.download(h_any.data(), n));
	bool some = false;
	for (size_t i = 0; i < n && !some; i++)
		some = h_any[i] != 0;
	if (!some) {
		h_any.clear();
		return 0; // no read of the batch has a masked base: the seed stage runs as without DUST
	}
	PGX_TRY(b.win_f.ensure(n_mask));
	PGX_TRY(b.win_r.ensure(n_mask));
	PGX_HIP(hipMemsetAsync(b.win_f.data(), 0, n_mask * sizeof(uint64_t), 0));
	PGX_HIP(hipMemsetAsync(b.win_r.data(), 0, n_mask * sizeof(uint64_t), 0));
	hipLaunchKernelGGL(k_dust_windows, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, 0, b.mask.data(), b.any.data(), rd->d_len.data(),
			   rd->d_woff.data(), (uint32_t)n, b.win_f.data(), b.win_r.data());
	PGX_HIP(hipGetLastError());
	PGX_HIP(hipDeviceSynchronize());
	rd->has_dust = true;
	rd->dust_listed[0] = n_listed[0];
	rd->dust_listed[1] = n_listed[1];
	return 0;
}

// The same passes again over a resident batch, on `stream`, without a host wait, INTO THE CALLER'S BUFFERS: S3d as part of a
// search (BLAST masks its queries inside every search; `pgx_db_set_dust_each_search`).  The batch is only read: the buffers
// belong to the searching handle's workspace, so searches of one batch through two handles do not meet.  The
// masks depend on the reads alone, so the launch sizes the import's pass found (and the search classes made from its
// per-read flags) hold; the kernels take their counts on the device.
int reads_dust_again(const pgx_reads *rd, DustBufs &b, hipStream_t stream)
{
	const size_t n = (size_t)rd->n;
	if (n == 0 || !rd->dustb.mask.base)
		return 0;
	const size_t n_mask = (size_t)rd->n_words + 24;
	PGX_TRY(dust_bufs_ensure(b, n, n_mask, rd->dust_listed[0]));
	if (rd->has_dust) {
		PGX_TRY(b.win_f.ensure(n_mask));
		PGX_TRY(b.win_r.ensure(n_mask));
	}
	PGX_HIP(hipMemsetAsync(b.mask.data(), 0, n_mask * sizeof(uint64_t), stream));
	PGX_HIP(hipMemsetAsync(b.any.data(), 0, n, stream));
	PGX_HIP(hipMemsetAsync(b.n.data(), 0, 2 * sizeof(uint32_t), stream));
	const uint64_t *amb = rd->has_amb ? rd->d_fwd_amb.data() : (const uint64_t *)nullptr;
	hipLaunchKernelGGL(k_dust_trigger<false>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, rd->d_fwd.data(), amb, rd->d_len.data(),
			   rd->d_woff.data(), (uint32_t)n, (const uint32_t *)nullptr, (const uint32_t *)nullptr, b.list.data(),
			   b.range.data(), b.n.data());
	if (rd->dust_listed[0]) {
		PGX_TRY(dust_order(b, b.list.data(), b.range.data(), b.n.data(), rd->dust_listed[0], stream));
		hipLaunchKernelGGL(k_dust_trigger<true>, dim3((unsigned)((rd->dust_listed[0] + 63) / 64)), dim3(64), 0, stream, rd->d_fwd.data(), amb,
				   rd->d_len.data(), rd->d_woff.data(), (uint32_t)n, b.list_s.data(), b.n.data(), b.list2.data(),
				   b.range.data(), b.n.data() + 1);
	}
	if (rd->dust_listed[1]) {
		PGX_TRY(dust_order(b, b.list2.data(), b.range.data(), b.n.data() + 1, rd->dust_listed[1], stream));
		hipLaunchKernelGGL(k_dust_mask, dim3((unsigned)((rd->dust_listed[1] + 63) / 64)), dim3(64), 0, stream, rd->d_fwd.data(), amb, rd->d_len.data(),
				   rd->d_woff.data(), b.list_s.data(), b.range_s.data(), b.n.data() + 1, b.mask.data(), b.any.data());
	}
	if (rd->has_dust) {
		PGX_HIP(hipMemsetAsync(b.win_f.data(), 0, n_mask * sizeof(uint64_t), stream));
		PGX_HIP(hipMemsetAsync(b.win_r.data(), 0, n_mask * sizeof(uint64_t), stream));
		hipLaunchKernelGGL(k_dust_windows, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, b.mask.data(), b.any.data(),
				   rd->d_len.data(), rd->d_woff.data(), (uint32_t)n, b.win_f.data(), b.win_r.data());
	}
	PGX_HIP(hipGetLastError());
	return 0;
}

} // namespace pgx

// Recompute the DUST window bits of a resident batch (the same bits: the masks depend on the reads alone).  For callers who
// count query masking as part of every search -- BLAST runs it per search -- and for bench.py's `dust_in_step` figure.
extern "C" int pgx_reads_redo_dust(pgx_reads *r)
{
	if (!r)
		return pgx::fail(PGX_E_ARG, "pgx_reads_redo_dust: null argument");
	if (int rc = pgx::require_device())
		return rc;
	return pgx::guard("pgx_reads_redo_dust", [&]() -> int {
		const bool had = r->has_dust;
		if (int rc = pgx::reads_dust(r))
			return rc;
		if (r->has_dust != had)
			return pgx::fail(PGX_E_NODEVICE, "pgx_reads_redo_dust: the masks of the batch changed between two passes");
		return 0;
	});
}

