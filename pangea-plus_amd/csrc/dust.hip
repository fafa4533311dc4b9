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
// k_dust_scan<NW>   every read, one lane per read, linear: the published algorithm's own bookkeeping (the window of the last 62
//                 triplets with its pair count r_w; L, the length of its longest suffix in which no triplet occurs more than
//                 4 times) and its test "10 r_w > 20 L", without which that algorithm never looks for a perfect interval
//                 ending at the position.  93 % of random 150-base reads never pass it.  The read sits in NW 64-bit
//                 registers; the state is ONE LDS word per triplet value (window count + its last four positions), so the
//                 suffix start moves with the entering triplet alone -- no walk.  Lists the reads with a position that
//                 passes, with the first and last such position.
// k_dust_perfect  ONE WAVEFRONT per listed read: the definition itself on the stretch the first pass marked, a dynamic
//                 programme over the interval LENGTH with lane = interval start, neighbours by DPP wave shifts; writes the
//                 masked bases and whether there are any (an interval above the level exists exactly when a perfect one
//                 does; the checker's fuzz of the first pass's test against the definition: oracle/fuzz_dust.c).
// k_dust_windows  one lane per (listed read with a masked base, 64 positions): the window bits of both strands from the mask
#include <type_traits>

#include "bitops.hpp"
#include "engine.hpp"

namespace pgx {

constexpr int kDustMaxT = 62, kDustLevel = 20;

// LDS written by one lane, read by another of the same wavefront (LDS operations of a wavefront complete in order)
__device__ __forceinline__ void dust_wave_sync()
{
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
}
// One 32-bit word per triplet value and lane: bits 24-29 the value's count in the window of the last 62 triplets, bits
// 0-23 the positions (mod 64) of its last four occurrences, newest in the low six bits.  65 words per lane: an odd stride
// (an even one puts the 64 lanes' words on two LDS banks).
struct ScanLane {
	uint32_t w[64];
	uint32_t pad;
};
static_assert(sizeof(ScanLane) / 4 % 2 == 1, "odd word stride");

// 64 bits of a read held in registers (R[k] = its k-th word, R[NW] .. = 0), from base `pos`; `pos` is the same in every lane
template <int NR> __device__ __forceinline__ uint64_t reg_window64(const uint64_t (&R)[NR], int pos)
{
	const int wi = __builtin_amdgcn_readfirstlane(pos >> 5), sh = (pos & 31) * 2;
	uint64_t lo = 0, hi = 0;
#pragma unroll
	for (int k = 0; k + 1 < NR; k++)
		if (wi == k) {
			lo = R[k];
			hi = R[k + 1];
		}
	return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

// First pass, every read of the batch, one lane per read: the published algorithm's window bookkeeping and its test
// "10 r_w > 20 L" (header), without which it never looks for a perfect interval ending at a position.  Lists the reads with
// a position that passes, with the first and last such position.
//   r_w  pairs of equal triplets in the window of the last (at most) 62 triplets: the entering triplet adds its count, the
//        leaving one takes its count - 1 away.
//   L    length of the window's longest suffix in which no triplet occurs more than 4 times.  Only the ENTERING triplet can
//        move that suffix's start: to just behind its own fifth-most-recent occurrence, if that lies inside the window --
//        which the triplet's window count says (5 or more with the entering one) and the triplet's last four positions
//        give.  Both live in ONE LDS word per triplet value (ScanLane): a position costs two reads and two writes.
//        Round 3 kept a second set of counters for the suffix and, when a count there reached 5, walked the suffix for its
//        earliest copy -- the whole wavefront for one lane's walk, some lane at most positions: ~1 500 cycles a position.
// NW > 0: reads of at most 32 NW bases without ambiguity letters, the whole read in NW 64-bit registers per lane -- no
// memory access inside the loop; NW = 0: any length, ambiguity flags, letters from memory every 16 positions.
template <int NW>
__global__ __launch_bounds__(64) void k_dust_scan(const uint64_t *__restrict__ fwd, const uint64_t *__restrict__ amb,
						   const uint32_t *__restrict__ len, const uint32_t *__restrict__ woff, uint32_t n,
						   uint32_t *__restrict__ list, uint2 *__restrict__ range, uint32_t *__restrict__ n_list)
{
	constexpr bool REGS = NW > 0;
	constexpr int NR = REGS ? NW + 1 : 1;
	__shared__ ScanLane s_lane[64];
	uint32_t *tab = s_lane[threadIdx.x].w;
	const uint32_t at0 = blockIdx.x * 64u + threadIdx.x;
	const bool has_read = at0 < n;
	const uint32_t r = has_read ? at0 : 0u;
	const int nt = has_read ? (int)len[r] - 2 : 0;
	const uint64_t *rw = fwd + woff[r], *ra = (!REGS && amb) ? amb + woff[r] : nullptr;
	uint64_t R[NR];
	if constexpr (REGS) {
#pragma unroll
		for (int k = 0; k < NR; k++)
			R[k] = k < NW ? rw[k] : 0ull; // (the packed reads carry spare words behind every read)
	}
	for (int k = 0; k < 64; k++)
		tab[k] = 0u;
	int first = -1, last = -1;
	int size = 0, start = 0, rw_pairs = 0; // the window is the `size` triplets that end at the current one; the suffix starts at `start`
	// the entering and the leaving triplet come from two 64-bit registers that hold 32 letters each (refilled every 16 positions)
	uint64_t in_w = 0, out_w = 0;
	int out_pos = -1;
	int nt_wave = nt; // (REGS: the refills are the wavefront's, every lane runs to the longest read's end)
	if constexpr (REGS) {
		for (int sh = 1; sh < 64; sh <<= 1) {
			const int o = __shfl_xor(nt_wave, sh);
			nt_wave = o > nt_wave ? o : nt_wave;
		}
	}
	for (int b = 0; b < nt_wave; b++) {
		int t = -1;
		if constexpr (REGS) {
			if ((b & 15) == 0)
				in_w = reg_window64(R, b);
			if (b >= kDustMaxT && ((b - kDustMaxT) & 15) == 0)
				out_w = reg_window64(R, b - kDustMaxT); // (every lane whose window is full is at the same position b - 62)
		}
		if (b < nt) {
			if constexpr (REGS) {
				t = (int)((in_w >> (2 * (b & 15))) & 63ull);
			} else {
				if ((b & 15) == 0)
					in_w = window64(rw, b);
				t = (int)(in_w & 63ull);
				in_w >>= 2;
				if (ra && (window64(ra, b) & 0x15ull))
					t = -1;
				if (t < 0) { // a letter that is no base: no interval crosses it
					for (int k = 0; k < 64; k++)
						tab[k] = 0u;
					size = rw_pairs = 0;
					start = b + 1;
				}
			}
		}
		if (t >= 0) {
			uint32_t wt = tab[t];
			int cnt = (int)(wt >> 24); // the entering triplet's count in the window, itself included (below)
			if (size >= kDustMaxT) {
				// the oldest triplet leaves: position b - 62 (the window was full, so it is 62 triplets behind)
				const int ob = b - kDustMaxT;
				int s0;
				if constexpr (REGS) {
					s0 = (int)((out_w >> (2 * (ob & 15))) & 63ull);
				} else {
					if (ob != out_pos || (ob & 15) == 0) {
						out_w = window64(rw, ob);
						out_pos = ob;
					}
					s0 = (int)(out_w & 63ull);
					out_w >>= 2;
					out_pos++;
				}
				if (s0 != t) {
					const uint32_t ws = tab[s0];
					tab[s0] = ws - (1u << 24);
					rw_pairs += cnt - ((int)(ws >> 24) - 1);
					cnt++;
				} // (the same triplet leaving and entering: its count and the pair sum end where they were)
			} else {
				size++;
				rw_pairs += cnt;
				cnt++;
			}
			// its fifth-most-recent occurrence = the oldest of the four positions kept, valid when the window holds five
			if (cnt >= 5) {
				const int d = (b - (int)((wt >> 18) & 63u)) & 63; // 4 .. 61 positions back
				const int cand = b - d + 1;
				start = cand > start ? cand : start;
			}
			tab[t] = (((wt << 6) | (uint32_t)(b & 63)) & 0xFFFFFFu) | ((uint32_t)cnt << 24);
			int L = b - start + 1;
			L = L < size ? L : size;
			if (rw_pairs * 10 > L * kDustLevel) {
				first = first < 0 ? b : first;
				last = b;
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

// Second pass, ONE WAVEFRONT per listed read: the definition itself (header) on the stretch where the first pass's test
// passed -- intervals [a, b] of at most 62 triplets with a >= first - 61 and b <= last -- as a dynamic programme over the
// interval LENGTH: lane j holds the start a = a0 + j, step l every lane's interval [a, a + l - 1].
//   pairs(a, b) = pairs(a + 1, b) + pairs(a, b - 1) - pairs(a + 1, b - 1) + [t_a = t_b]
//   best(a, b)  = max(score(a, b), best(a + 1, b), best(a, b - 1))      (exact fractions, a missing score below all)
// so a step needs lane j + 1's values of the two steps before (two DPP wave shifts) and the triplet at the interval's end
// (a third); perfect = score above the level and no sub-interval higher; lane j keeps the furthest base its perfect
// intervals reach.  More than 64 starts: chunks of 64 from the top, a chunk's lane 0 leaving its two sequences in LDS
// for lane 63 of the chunk below.  Round 3 ran this stretch twice with ONE LANE per read -- a walk over the longer
// suffixes at every position that passed (3.9 ms for 0.7 M reads), then the dynamic programme row by row on the reads
// that survived (3.0 ms for 0.26 M) -- eleven and four thousand wavefronts of a hundred thousand instructions each on
// a chip that holds sixteen thousand: 6.9 ms of a nearly empty machine.
__device__ __forceinline__ int dust_dpp_up(int v, int last_lane_value) // lane j <- lane j + 1; lane 63 <- last_lane_value
{
	return __builtin_amdgcn_update_dpp(last_lane_value, v, 0x130, 0xf, 0xf, false);
}

__global__ __launch_bounds__(64) void k_dust_perfect(const uint64_t *__restrict__ fwd, const uint64_t *__restrict__ amb,
						      const uint32_t *__restrict__ len, const uint32_t *__restrict__ woff,
						      const uint32_t *__restrict__ list, const uint2 *__restrict__ range, const uint32_t *__restrict__ n_list,
						      uint64_t *__restrict__ mask, uint8_t *__restrict__ any)
{
	// lane 0's sequences (pairs, best score as numerator and denominator; index = interval length) of the chunk above
	__shared__ uint32_t s_up[3][64];
	const int lane = (int)(threadIdx.x & 63);
	const uint32_t n_in = *n_list;
	for (uint32_t at = blockIdx.x; at < n_in; at += gridDim.x) {
		const uint32_t r = list[at];
		const int L = (int)len[r], nt = L - 2;
		const uint64_t *rw = fwd + woff[r], *ra = amb ? amb + woff[r] : nullptr;
		uint64_t *mw = mask + woff[r];
		const int nw = (L + 63) >> 6;
		for (int w = lane; w < nw; w += 64)
			mw[w] = 0ull;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the words are zero before any lane ORs into them)
		const int b_hi = (int)range[at].y < nt - 1 ? (int)range[at].y : nt - 1;
		const int a_lo = (int)range[at].x - (kDustMaxT - 1) > 0 ? (int)range[at].x - (kDustMaxT - 1) : 0;
		const int n_chunks = (b_hi - a_lo + 64) / 64;
		bool marked = false;
		for (int c = n_chunks - 1; c >= 0; c--) {
			const int a0 = a_lo + 64 * c;
			// this lane's triplet and the one 64 positions on (what lane 63's intervals reach), -1: none
			auto triplet = [&](int p) -> int {
				if (p >= nt || (ra && (window64(ra, p) & 0x15ull)))
					return -1;
				return (int)(window64(rw, p) & 63ull);
			};
			const int a = a0 + lane;
			const int ta = triplet(a), t_far = triplet(a + 64);
			const bool has_up = c + 1 < n_chunks; // (the top chunk's lane 63 has no neighbour: its longer intervals end past b_hi anyway)
			const uint32_t upP = has_up ? s_up[0][lane] : 0u, upN = has_up ? s_up[1][lane] : 0u, upQ = has_up ? s_up[2][lane] : 1u;
			dust_wave_sync(); // (read before this chunk's lane 0 replaces them)
			bool live = a <= b_hi && ta >= 0;
			int tb = ta;
			// pairs of the two steps before; best score of the step before as a fraction Bn / Bq (no score yet = 0 / 1: below or
			// equal to every score, and the level test never passes with zero pairs)
			uint32_t P1 = 0, P2 = 0, Bn = 0, Bq = 1;
			uint32_t myP = 0, myN = 0, myQ = 1; // lane l: lane 0's values at length l
			int end = -1;                       // furthest base a perfect interval of this start covers
			const int l_max = b_hi - a0 + 1 < kDustMaxT ? b_hi - a0 + 1 : kDustMaxT; // (lane 0's reach; the other lanes die earlier)
			// (two copies of the loop: without a chunk above -- the only chunk of most reads -- lane 63's neighbour values are
			// constants, which lets the compiler fold the wave shifts into the instructions that use them)
			auto steps = [&](auto UP) {
			constexpr bool HAS_UP = decltype(UP)::value;
			// (lane 63's constants as opaque registers: as literals the compiler folds the shift into the instruction that
			// uses it with bound_ctrl, and the results were wrong on gfx950)
			int k_zero = 0, k_one = 1, k_none = -1;
			asm volatile("" : "+v"(k_zero), "+v"(k_one), "+v"(k_none));
			for (int l = 2; l <= l_max; l++) {
				// lane j + 1's registers before this step: its pairs at l - 1 and l - 2, its best at l - 1, its end triplet;
				// lane 63's neighbour is lane 0 of the chunk above
				const uint32_t nP1 = (uint32_t)dust_dpp_up((int)P1, HAS_UP ? __builtin_amdgcn_readlane((int)upP, l - 1) : k_zero);
				const uint32_t nP2 = (uint32_t)dust_dpp_up((int)P2, HAS_UP && l > 2 ? __builtin_amdgcn_readlane((int)upP, l - 2) : k_zero);
				const uint32_t nBn = (uint32_t)dust_dpp_up((int)Bn, HAS_UP ? __builtin_amdgcn_readlane((int)upN, l - 1) : k_zero);
				const uint32_t nBq = (uint32_t)dust_dpp_up((int)Bq, HAS_UP ? __builtin_amdgcn_readlane((int)upQ, l - 1) : k_one);
				tb = dust_dpp_up(tb, HAS_UP ? __builtin_amdgcn_readlane(t_far, l - 2) : k_none); // position a0 + 63 + l - 1
				live = live && tb >= 0 && a + l - 1 <= b_hi;
				const uint32_t P = P1 + nP1 - nP2 + (tb == ta ? 1u : 0u);
				const uint32_t q = (uint32_t)(l - 1);
				// the better of the two sub-interval bests: [a, b - 1] (this lane, the step before) and [a + 1, b] (lane j + 1)
				const bool up_better = __umul24(nBn, Bq) > __umul24(Bn, nBq);
				const uint32_t sn = up_better ? nBn : Bn, sq = up_better ? nBq : Bq;
				const uint32_t lhs = __umul24(sn, q), rhs = __umul24(P, sq); // sub > score <=> lhs > rhs
				if (live && P * 10u > (uint32_t)kDustLevel * q && lhs <= rhs)
					end = a + l + 1; // bases a .. b + 2, b = a + l - 1
				const bool sc_better = rhs > lhs;
				P2 = P1;
				P1 = live ? P : 0u;
				Bn = live ? (sc_better ? P : sn) : 0u;
				Bq = live ? (sc_better ? q : sq) : 1u;
				if (c > 0) { // (somebody below will ask)
					// (lane 0's values first, in statements of their own: inside the conditional operand they would be
					// read under the branch's execution mask -- from lane l itself)
					const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)P1, 0), n0 = (uint32_t)__builtin_amdgcn_readlane((int)Bn, 0),
						       q0 = (uint32_t)__builtin_amdgcn_readlane((int)Bq, 0);
					const bool me = lane == l;
					myP = me ? p0 : myP;
					myN = me ? n0 : myN;
					myQ = me ? q0 : myQ;
				}
			}
			};
			if (has_up)
				steps(std::true_type{});
			else
				steps(std::false_type{});
			if (c > 0) {
				s_up[0][lane] = myP;
				s_up[1][lane] = myN;
				s_up[2][lane] = myQ;
				dust_wave_sync();
			}
			if (end >= 0) {
				marked = true;
				for (int k = a; k <= end;) {
					const int w = k >> 6, lo = k & 63;
					const int hi = (end - (w << 6)) < 63 ? end - (w << 6) : 63;
					const uint64_t bits = (hi == 63 ? ~0ull : ((2ull << hi) - 1ull)) & (~0ull << lo);
					atomicOr(reinterpret_cast<unsigned long long *>(mw + w), (unsigned long long)bits);
					k = (w + 1) << 6;
				}
			}
		}
		if (__ballot(marked) && lane == 0)
			any[r] = 1;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
}

// window bits of both strands of the LISTED reads that hold a masked base: bit i of dustwin_f[read]: bases i .. i + 27 of
// the read hold no masked base (and i + 28 <= L); dustwin_r the same for the reverse-complement strand (its mask is the
// forward mask reversed).  The seed stage looks at the words of reads with any[read] != 0 only, so the words of the other
// reads are never written (round 3 wrote -- and cleared -- all of them: 2 GB per 10 M reads).
__global__ void k_dust_windows(const uint64_t *__restrict__ mask, const uint8_t *__restrict__ any, const uint32_t *__restrict__ len,
			       const uint32_t *__restrict__ woff, const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_list,
			       uint64_t *__restrict__ win_f, uint64_t *__restrict__ win_r)
{
	const uint32_t n_in = *n_list;
	for (uint32_t at = blockIdx.x * blockDim.x + threadIdx.x; at < n_in; at += gridDim.x * blockDim.x) {
		const uint32_t r = list[at];
		if (!any[r])
			continue;
		const int L = (int)len[r];
		const uint32_t w0 = woff[r];
		const int nw = (L + 63) >> 6;
		const int n_valid = L - kWord + 1; // window positions 0 .. n_valid - 1
		auto valid_bits = [&](int w) -> uint64_t { // positions of word w below n_valid
			const int left = n_valid - (w << 6);
			return left <= 0 ? 0ull : (left >= 64 ? ~0ull : ((1ull << left) - 1ull));
		};
		for (int w = 0; w < nw; w++) {
			uint64_t f = valid_bits(w);
			// a window is spoilt by a masked base at any of its 28 positions: OR of the mask moved down by 0 .. 27, in log
			// steps on the 128 bits that start at this word (2, 4, 8, 16 wide, then 16 + 8 + 4)
			uint64_t lo = mask[w0 + w], hi = w + 1 < nw ? mask[w0 + w + 1] : 0ull;
			auto shr = [](uint64_t &a, uint64_t &b, int k) { // (a, b) |= (a, b) >> k
				a |= (a >> k) | (b << (64 - k));
				b |= b >> k;
			};
			shr(lo, hi, 1); // covers 2
			shr(lo, hi, 2); // covers 4
			const uint64_t l4 = lo, h4 = hi;
			shr(lo, hi, 4); // covers 8
			const uint64_t l8 = lo, h8 = hi;
			shr(lo, hi, 8); // covers 16
			const uint64_t d = lo | ((l8 >> 16) | (h8 << 48)) | ((l4 >> 24) | (h4 << 40)); // 16 + 8 + 4 = 28
			f &= ~d;
			win_f[w0 + w] = f;
		}
		// the reverse-complement strand: its window i is the forward window n_valid - 1 - i
		for (int w = 0; w < nw; w++) {
			uint64_t rv = valid_bits(w);
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
			win_r[w0 + w] = rv;
		}
	}
}

// One DUST pass over a batch on `stream`, into `b`: the first pass over every read, the definition on the listed reads, the
// window bits of the reads with a masked base.  No host wait: the kernels behind the first pass take the list's length
// on the device.
static int dust_pass(const pgx_reads *rd, DustBufs &b, hipStream_t stream)
{
	const size_t n = (size_t)rd->n;
	const size_t n_mask = (size_t)rd->n_words + 24;
	PGX_TRY(b.mask.ensure(n_mask));
	PGX_TRY(b.win_f.ensure(n_mask));
	PGX_TRY(b.win_r.ensure(n_mask));
	PGX_TRY(b.any.ensure(n));
	PGX_TRY(b.list.ensure(n));
	PGX_TRY(b.range.ensure(n));
	PGX_TRY(b.n.ensure(2));
	PGX_HIP(hipMemsetAsync(b.any.data(), 0, n, stream));
	PGX_HIP(hipMemsetAsync(b.n.data(), 0, 2 * sizeof(uint32_t), stream));
	const uint64_t *amb = rd->has_amb ? rd->d_fwd_amb.data() : (const uint64_t *)nullptr;
	const dim3 g((unsigned)((n + 63) / 64)), blk(64);
#define PGX_DUST_SCAN(NW)                                                                                                                \
	hipLaunchKernelGGL(k_dust_scan<NW>, g, blk, 0, stream, rd->d_fwd.data(), amb, rd->d_len.data(), rd->d_woff.data(), (uint32_t)n, \
			   b.list.data(), b.range.data(), b.n.data())
	if (amb || rd->max_len > 512)
		PGX_DUST_SCAN(0);
	else if (rd->max_len <= 192)
		PGX_DUST_SCAN(6);
	else if (rd->max_len <= 320)
		PGX_DUST_SCAN(10);
	else
		PGX_DUST_SCAN(16);
#undef PGX_DUST_SCAN
	const unsigned grid2 = (unsigned)std::min<size_t>(n, 256 * 32);
	hipLaunchKernelGGL(k_dust_perfect, dim3(grid2), dim3(64), 0, stream, rd->d_fwd.data(), amb, rd->d_len.data(), rd->d_woff.data(), b.list.data(),
			   b.range.data(), b.n.data(), b.mask.data(), b.any.data());
	hipLaunchKernelGGL(k_dust_windows, dim3((unsigned)std::min<size_t>((n + 127) / 128, 256 * 8)), dim3(128), 0, stream, b.mask.data(), b.any.data(),
			   rd->d_len.data(), rd->d_woff.data(), b.list.data(), b.n.data(), b.win_f.data(), b.win_r.data());
	PGX_HIP(hipGetLastError());
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
	PGX_TRY(dust_pass(rd, b, 0));
	std::vector<uint8_t> &h_any = rd->h_read_dust;
	h_any.resize(n);
	PGX_TRY(b.any.download(h_any.data(), n)); // (waits for the pass)
	bool some = false;
	for (size_t i = 0; i < n && !some; i++)
		some = h_any[i] != 0;
	if (!some) {
		h_any.clear();
		return 0; // no read of the batch has a masked base: the seed stage runs as without DUST
	}
	rd->has_dust = true;
	return 0;
}

// The same pass again over a resident batch, on `stream`, without a host wait, INTO THE CALLER'S BUFFERS: S3d as part of a
// search (BLAST masks its queries inside every search; `pgx_db_set_dust_each_search`).  The batch is only read: the buffers
// belong to the searching handle's workspace, so searches of one batch through two handles do not meet.  The
// masks depend on the reads alone, so the search classes made from the import's per-read flags hold.
int reads_dust_again(const pgx_reads *rd, DustBufs &b, hipStream_t stream)
{
	if (rd->n == 0)
		return 0;
	return dust_pass(rd, b, stream);
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

