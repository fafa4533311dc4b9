// Gapped stage of BLAST mode (spec pgx-blastn v2, S3b): every initial HSP the seed stage found is extended with gaps
// from the first base of its seed run, to the left and to the right.
//
// Replaces the gapped half of `blastn -query F -db DB -outfmt 6` (reference README.md:96; Scripts/run_multi_blastn.pl:56;
// Scripts/submit_MPI-blast.job:24); the reference's record of that program's output, validation_dataset/
// Data-set_2_consensus.xlsx, has gapopen > 0 in 10 933 of 10 992 rows.  The algorithm is the published one megablast
// uses (Zhang, Schwartz, Wagner, Miller, J Comput Biol 7 (2000), fig. 4): match +1, mismatch -2, gap column -2.5,
// X = 54; an extension that reaches (i, j) with d differences scores (i + j) / 2 - 3 d, kept doubled here
// (S2 = i + j - 6 d).  R(d, k) = furthest i on diagonal k = i - j with d differences; the rules the paper leaves open
// are stated in DESIGN.md (spec v2) and restated by the checker in oracle/o_gapped.c.
//
// Tiers, the same results from each; a tier lists what it cannot hold for the next:
//   k_gapped_rows<MAXL, WAVES, 18>   the main table of reads of <= 512 bases: one LANE per HSP, both sides in one round of 64
//                  HSPs (gap_round); the rounds follow the slot list that the counting sorts k_reg_* / k_seg_* make -- by
//                  database region, inside a region by the seed stage's level estimates -- and are handed out in order from
//                  one counter, so the wavefronts in flight share about a megabyte of database words (L2).  The lean cell
//                  (greedy_rows_lean): the row R(d, .) of a lane in registers, letters in transposed LDS rows, one
//                  `v_max3` per parent choice, 18 differences a side (no X-drop test is due below 19).
//   k_gapped_rows<512, 2, 40>        reads of 321-512 bases: the same rounds with rows for 40 differences a side, the
//                  X-drop history in a register ring (greedy_rows_deep)
//   k_gapped_pool<FLAT, ...>         the overflow table and batches of long reads: a wavefront orders a pool of HSPs by the
//                  key byte itself, then gap_round
//   k_gapped_fast<2, MAXL, WAVES, 18> / <2, 512, 2, 40>   the LISTS behind the lean tier: rows with the full statistics in
//                  the cell (mismatches, gap openings, open-gap kind) for 18, then 40 differences with the X-drop history
//   k_gapped_diag  one WAVEFRONT per listed HSP, one lane per diagonal, the level in one register per lane, neighbours by
//                  DPP wave shifts: reads above 512 bases and what the lane-per-HSP tiers pass on
//   k_gapped_big<62>, <1000>   one wavefront per listed HSP, rows double-buffered in LDS, ambiguity flags: what
//                  k_gapped_diag hands on (drift beyond its 64 lanes, ambiguity letters, reads above 2 048 bases)
// (k_gapped_fast<0|1>, round 2's two-pass pools over the tables, is compiled in measurement builds only: PGX_STAGE_PROBES.)
// Every tier cuts a cell whose score could not pass the best one even if every remaining letter matched.  The cut cannot
// change the result (a child's bound is below its parent's, so no surviving cell has a cut parent; a cut cell never holds
// the best score), which is why the sequential kernel, the parallel ones (bound taken one step late) and the checker (no
// cut at all) agree.
#include <type_traits>
#include "bitops.hpp"
#include "engine.hpp"

namespace pgx {

constexpr int kGX2 = 108;     // 2 X
constexpr int kGLag = 19;     // floor((X + 1/2) / 3) + 1: the X-drop test looks at the best score 19 differences earlier
constexpr int kGFastD = 18;   // differences per side of the lane-per-HSP kernel's first tier: below kGLag, so it never makes an X-drop test
constexpr int kGFastD2 = 40;  // second tier (the HSPs the first one lists): keeps the score history the X-drop test needs
constexpr int kGDmax = 1000;  // differences per side, spec
constexpr int kGGroup = 4;    // diagonals per group of the unrolled row (lane kernel)
constexpr int kGUnrollLevels = 7; // levels of the lane kernel compiled as straight code
constexpr uint32_t kCellNone = 0x80000000u; // lane kernel: a dead cell holds i = -32768
constexpr uint32_t kBigNone = 0xFFFFFFFFu;  // wide kernel: a dead cell
static_assert(kGFastD < kGLag, "the first tier keeps no score history");

#ifdef PGX_STAGE_PROBES
// measurement builds: [0] cell steps of a wavefront, [1] cells evaluated by lanes, [2] slide rounds (wavefront) inside cell
// steps, [3] levels, [4] rounds (sides x 64), [5] lanes with a side, [6] walk rounds of B0, [7] lane-cells alive at level start
__device__ unsigned long long g_gap_stats[8];
#define GAP_STAT(i, n)                                          \
	do {                                                    \
		const unsigned long long n_ = (unsigned long long)(n); /* (ballots are taken by the whole wavefront) */ \
		if ((threadIdx.x & 63) == 0)                    \
			atomicAdd(&g_gap_stats[i], n_);         \
	} while (0)
#else
#define GAP_STAT(i, n) ((void)0)
#endif

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F> __device__ __forceinline__ void static_for(F &&f)
{
	if constexpr (B < E) {
		f(std::integral_constant<int, B>{});
		static_for<B + 1, E>(f);
	}
}

__device__ __forceinline__ void lds_sync()
{
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
}

struct GapSeqs {
	const uint64_t *rw, *ra;   // read strand: words, spaced ambiguity flags (or null)
	const uint64_t *dbw, *dba; // database words, flags (or null)
};

struct Side {
	int i, j, s2, mism, gopen;
};

// Letters that match from read position qp / database position dp on (DIR = +1: ascending, DIR = -1: descending), at
// most `cap`.  16 letters per round: one 8-byte load per sequence.
template <int DIR> __device__ __forceinline__ int lcp(const GapSeqs &s, int qp, int64_t dp, int cap)
{
	int n = 0;
	while (n < cap) {
		const int take = cap - n < 16 ? cap - n : 16;
		// ascending: the window starts at the next letter; descending: it ENDS there (it never reaches below the first
		// letter of the sequence: take <= letters left)
		const int q0 = DIR > 0 ? qp + n : qp - n - (take - 1);
		const int64_t d0 = DIR > 0 ? dp + n : dp - n - (take - 1);
		const uint32_t x = window16(s.rw, q0) ^ window16(s.dbw, d0);
		uint32_t y = (x | (x >> 1)) & 0x55555555u;
		if (s.ra)
			y |= window16(s.ra, q0) & 0x55555555u;
		if (s.dba)
			y |= window16(s.dba, d0) & 0x55555555u;
		if (take < 16)
			y &= (1u << (2 * take)) - 1u;
		int run;
		if (DIR > 0) {
			run = y ? (__ffs((int)y) - 1) >> 1 : take;
		} else {
			y <<= 2 * (16 - take);
			run = y ? __clz((int)y) >> 1 : take;
		}
		n += run;
		if (run < take)
			break;
	}
	return n;
}

// ------------------------------------------------------------------------------------------ one lane per HSP
// Work per HSP is quadratic in its differences and varies a lot (0 to ~500 cells).  History of this kernel, 10 M reads
// (282 M HSPs) per launch (DESIGN.md section 7 has the table): nested per-lane loops, letters through the caches: 1 080 ms;
// a per-lane state machine with dynamic refill: 340 ms (470 vector instructions per trip: every rare per-lane event
// happens in some lane on every trip); WAVE-UNIFORM loops over d and k with the HSPs of a pool ordered by cost, the row in
// registers, letters staged in LDS: 198 -> 124 ms; then the finding that a wavefront of this kernel is bound by its OWN
// instruction issue (one instruction per 4 cycles whatever its type): scalar skip tests grouped and then compiled away for
// the first levels, one forward-only copy of the row code (left sides staged reversed), parents by one signed maximum,
// the bound as two compares, B0 and the level count handed over by the seed stage, 4 wavefronts per SIMD: 56 ms.
// HSPs that touch an ambiguity letter (read or database window) go to the wide kernel, which applies the flag words.

// the hardware's bit scans as they are: -1 when no bit is set (the C forms add a select for that case)
__device__ __forceinline__ uint32_t scan_low(uint32_t y)
{
	uint32_t f;
	asm("v_ffbl_b32 %0, %1" : "=v"(f) : "v"(y));
	return f;
}

// Letters that match from read position qp / window position dp on (positions as BIT offsets, 2 per letter: the funnel
// shift takes its amount from the low five bits as they are), at most min(16, cap).  Letters past `cap` are whatever the
// window holds: the unsigned minimum drops them, and turns "no mismatch" (-1 from the bit scan) into the cap.
// Only ascending: the side LEFT of the anchor is staged reversed (see k_gapped_fast), so one copy of the row code serves
// both sides -- half the instruction footprint of the kernel.
// (the list tiers' letters lie in TRANSPOSED rows too since round 3: word i of a lane at w[i * 64], its own LDS bank)
constexpr int kLaneWordStride = 64;
__device__ __forceinline__ uint32_t lds_window16_bits(const uint32_t *w, int bit)
{
	const int i = bit >> 5;
	return __builtin_amdgcn_alignbit(w[(i + 1) * kLaneWordStride], w[i * kLaneWordStride], (uint32_t)bit);
}

__device__ __forceinline__ int lcp16(const uint32_t *rd, const uint32_t *db, int qbit, int dbit, int cap)
{
	const uint32_t x = lds_window16_bits(rd, qbit) ^ lds_window16_bits(db, dbit);
	const uint32_t y = (x | (x >> 1)) & 0x55555555u;
	const uint32_t f = scan_low(y) >> 1;
	const uint32_t r = f < 16u ? f : 16u;
	return (int)(r < (uint32_t)cap ? r : (uint32_t)cap);
}

struct GapView {
	const uint64_t *fwd, *rc, *fwd_amb, *rc_amb;
	const uint32_t *len, *woff;
	const uint64_t *dbw, *dba;
	const uint32_t *seq_off;
	int64_t db_bases;   // letters in the database (its packed words carry kDbPadBases zero letters in front and behind)
	const uint8_t *key; // per slot of the table this launch works on: the seed stage's work estimate
	const uint32_t *amb_blk; // one bit per 512-base block of the database that holds an ambiguity letter (or null)
};

// the seed record of the seed stage (classify.hip: pack_seed): everything a round needs except the subject's ends
struct Anchor {
	int strand, L, qa, sa, slen;
	int b0l, b0r; // B0 of the two sides as the seed stage found them on the diagonal's mismatch flags (0: none given)
	int64_t S0;
	uint32_t gpos; // database position of the anchor
	GapSeqs s;
};

__device__ __forceinline__ Anchor anchor_of(const GapView &v, const pgx_hit &h)
{
	Anchor a;
	a.strand = h.send & 1;
	a.b0l = (h.send >> 1) & 0x7FF;
	a.b0r = (h.send >> 12) & 0x7FF;
	a.L = h.score;
	a.qa = h.qend;
	a.gpos = (uint32_t)h.sstart;
	a.S0 = (int64_t)v.seq_off[h.subject];
	a.slen = (int)(v.seq_off[h.subject + 1] - v.seq_off[h.subject]);
	a.sa = (int)((int64_t)a.gpos - a.S0);
	const uint32_t w0 = (uint32_t)h.qstart;
	a.s.rw = (a.strand ? v.rc : v.fwd) + w0;
	const uint64_t *ra = a.strand ? v.rc_amb : v.fwd_amb;
	a.s.ra = ra ? ra + w0 : nullptr;
	a.s.dbw = v.dbw;
	a.s.dba = v.dba;
	return a;
}

__device__ __forceinline__ void write_gapped(pgx_hit *hp, const pgx_hit &h, const Anchor &a, const Side &l, const Side &r)
{
	const int bl = a.qa - l.i, br = a.qa + r.i - 1, sl = a.sa - l.j, sr = a.sa + r.j - 1;
	pgx_hit o;
	o.read = h.read;
	o.subject = h.subject;
	o.score = (l.s2 + r.s2) >> 1;
	o.mismatch = (uint16_t)(l.mism + r.mism);
	o.gapopen = (uint16_t)(l.gopen + r.gopen);
	if (!a.strand) {
		o.qstart = bl + 1;
		o.qend = br + 1;
		o.sstart = sl + 1;
		o.send = sr + 1;
	} else {
		o.qstart = a.L - br;
		o.qend = a.L - bl;
		o.sstart = sr + 1;
		o.send = sl + 1;
	}
	*hp = o;
}

// One side of a lane's HSP, all 64 lanes in step: the loops over d and k are wave-uniform, so the row R(d, .) of a lane
// lives in REGISTERS (kGFastCells words, indexed by the unrolled k) and is updated in place, k ascending.  Lanes whose
// cell is dead, or that have finished, idle for that step; the caller groups sides of similar cost to keep that rare.
// cell: bits 16-31 i (signed; kCellNone holds -32768 there, so a dead parent loses every maximum), 14-15 zero (the
// parents' priority goes there while they are compared), 8-13 mismatches, 2-7 gap openings, 0-1 the OPEN gap: the kind of
// the path's last column (1 gap in the subject row, 2 gap in the query row) if that column is a gap and no letter matched
// after it, else 0 -- a gap column opens a gap unless it continues that one.  With i on top, ONE signed maximum over the
// three parent words (each moved to the row it would reach, its priority in bits 14-15) yields the furthest row, the
// parent that wins ties (this diagonal, then k - 1, then k + 1) and that parent's statistics together.
//
// A second cut, also unable to change the result while no X-drop test is made (d < kGLag: every score then passes it):
// B0 = the best score of the path that never leaves the anchor's diagonal (at most 18 mismatches).  The SEED stage
// finds it on the diagonal's mismatch flags, which it holds anyway, and hands it over in the seed record together with
// the number of levels the side will take, floor((2 M - B0) / 5), by which the rounds are ordered (a walk here cost a
// twentieth of the kernel; ordering by the diagonal's mismatch count left the level counts of a round uneven).
// That path is among the cells of the literal algorithm, so the final best is >= B0; a cell whose bound is BELOW B0 can
// neither reach the final best nor tie it, its children's bounds are lower still, and no surviving cell has such a
// parent.  A side that is still alive at d = kGLag goes to the wide kernel, which makes the X-drop tests.
// Returns false for a lane whose cells are still alive after kGFastD differences.
// D = differences per side this instance holds; `xring` (D >= kGLag only) = this lane's T[0 .. D - kGLag] in LDS: the best
// score seen with at most d' differences, for the X-drop test of level d' + kGLag
template <int D>
__device__ __forceinline__ bool greedy_rows(const uint32_t *rdw, const uint32_t *dbwin, int *xring, bool on, int q0, int d0, int M, int N, int b0, Side &out)
{
	int slide_rounds = 0;
	(void)slide_rounds;
	const int q0b = 2 * q0, d0b = 2 * d0; // bit offsets of the side's first letters
	auto slide = [&](int &ii, int &jj) {
		for (;;) {
#ifdef PGX_STAGE_PROBES
			slide_rounds++;
#endif
			const int cap = M - ii < N - jj ? M - ii : N - jj;
			if (cap <= 0)
				break;
			const int run = lcp16(rdw, dbwin, q0b + 2 * ii, d0b + 2 * jj, cap);
			ii += run;
			jj += run;
			if (run < 16)
				break;
		}
	};
	out.i = out.j = out.s2 = out.mism = out.gopen = 0;
	int i0 = 0, j0 = 0;
	GAP_STAT(4, 1);
	GAP_STAT(5, __popcll(__ballot(on)));
	if (on)
		slide(i0, j0);
	out.i = out.j = i0;
	out.s2 = 2 * i0;
	bool live = on && !(i0 == M || i0 == N); // this lane still has cells to explore
	constexpr int kCells = 2 * D + 3;
	uint32_t R[kCells];
#pragma unroll
	for (int c = 0; c < kCells; c++)
		R[c] = kCellNone;
	constexpr int C = D + 1;
	int tbest = 2 * i0, tcmp = -(1 << 29); // the true best so far (the X-drop history; `best` below may start above it)
	if constexpr (D >= kGLag)
		xring[0] = tbest;
	R[C] = live ? (uint32_t)i0 << 16 : kCellNone;
	// `best` starts one below B0 when the anchor's diagonal does better than its first run: no cell below B0 can be the
	// answer (the diagonal's own cells reach B0), so the bound test needs one number, not two
	int best = 2 * i0 > b0 - 1 ? 2 * i0 : b0 - 1, best_k = 0;
	uint32_t best_cell = (uint32_t)i0 << 16; // the cell that holds the best score (its statistics travel in it)
	const int A2 = 2 * M, B2 = 2 * N;
	int slack_q = A2 - best, slack_s = B2 - best;
	bool over = false;
	uint32_t prev = kCellNone; // the old value of the cell left of the one being written
	bool any = false;
	// one cell of level d (six_d = 6 d), c a compile-time index into the row
	// (`reach` = the widest diagonal the level before could have written, when that is known while compiling: parents
	// beyond it are dead without looking)
	auto cell = [&](auto cc, int six_d, auto reach_c) {
		constexpr int c = decltype(cc)::value;
		constexpr int k = c - C;
		constexpr int reach = decltype(reach_c)::value;
		const uint32_t cur = R[c], nxt = R[c + 1];
		// the bound min(2 M - k, 2 N + k) - 6 d > best, as two tests of per-lane slacks against wave-uniform numbers; when
		// no lane passes it (the dead half of a side's last levels) the cell is dead for the whole wavefront
		const bool can = (slack_q > six_d + k) & (slack_s > six_d - k);
		if (__ballot(can) == 0ull) {
			prev = cur;
			R[c] = kCellNone;
			return;
		}
		int m3;
		if constexpr (k - 1 < -reach && k > reach) // (level 1, k = +1 handled below; this is never true)
			m3 = (int)kCellNone;
		else if constexpr (k < -reach)
			m3 = (int)nxt; // the left edge of the level: only diagonal k + 1 can lead here
		else if constexpr (k > reach)
			m3 = (int)(prev + 0x00014000u); // the right edge: only diagonal k - 1
		else if constexpr (k - 1 < -reach && k + 1 > reach)
			m3 = (int)(cur + 0x00018000u); // level 1, k = 0
		else if constexpr (k - 1 < -reach)
			m3 = max((int)(cur + 0x00018000u), (int)nxt);
		else if constexpr (k + 1 > reach)
			m3 = max((int)(cur + 0x00018000u), (int)(prev + 0x00014000u));
		else
			m3 = max((int)(cur + 0x00018000u), max((int)(prev + 0x00014000u), (int)nxt));
		const int v = m3 >> 16;
		const uint32_t kind = 2u - (((uint32_t)m3 >> 14) & 3u); // 0: mismatch on this diagonal, 1: from k - 1, 2: from k + 1
		const int jj0 = v - k;
		// (a lane that is not live holds a row of dead cells -- every cell of its last level was written dead -- so it
		// needs no test of its own: v is -32768 there)
		bool alive = ((uint32_t)v <= (uint32_t)M) & ((uint32_t)jj0 <= (uint32_t)N) & can;
		if constexpr (D >= kGLag)
			alive = alive & (v + jj0 - six_d >= tcmp); // the X-drop test on the score before sliding (tcmp = T[d - 19] - 2 X)
		uint32_t nc = kCellNone;
		GAP_STAT(0, 1);
		GAP_STAT(1, __popcll(__ballot(alive)));
#ifdef PGX_STAGE_PROBES
		slide_rounds = 0;
#endif
		if (alive) {
			// the first 16 letters without the loop's bookkeeping: off the anchor's diagonal a run rarely goes further
			int ii = v, jj = jj0;
			{
				const int cap = M - ii < N - jj ? M - ii : N - jj; // >= 0 for a live cell
				const int run = lcp16(rdw, dbwin, q0b + 2 * ii, d0b + 2 * jj, cap);
				ii += run;
				jj += run;
				if (run == 16)
					slide(ii, jj);
			}
			// one more mismatch, or one more gap opening unless the column continues the parent's open gap
			const uint32_t pst = (uint32_t)m3;
			const uint32_t inc = kind == 0u ? 1u << 8 : ((pst & 3u) != kind ? 1u << 2 : 0u);
			nc = ((uint32_t)ii << 16) | (((pst & 0x3FFCu) + inc) | (ii > v ? 0u : kind));
			const int s2 = ii + jj - six_d;
			if constexpr (D >= kGLag)
				tbest = s2 > tbest ? s2 : tbest;
			if (s2 > best) {
				best = s2;
				best_cell = nc;
				best_k = k;
			}
			slack_q = A2 - best;
			slack_s = B2 - best;
			any = true;
		}
#ifdef PGX_STAGE_PROBES
		{
			int mx = alive ? slide_rounds : 0;
			for (int sh = 1; sh < 64; sh <<= 1) {
				const int o = __shfl_xor(mx, sh);
				mx = o > mx ? o : mx;
			}
			GAP_STAT(2, mx);
		}
#endif
		prev = cur;
		R[c] = nc;
	};
	// The first kGUnrollLevels levels as straight code: level d is the diagonals -d .. d, known when compiling, so they
	// carry no tests at all (most sides of a 150-base read end within them); the later levels share one unrolled row
	// whose diagonals are tested, in groups of kGGroup first, for "level d does not reach this one".
	// After a level: can any cell of the next one pass the bound?  Its diagonals k must lie in
	// (6 (d + 1) - slack_s, slack_q - 6 (d + 1)) and in [-(d + 1), d + 1]; when no lane has such a k the side ends here,
	// without a level of dead cells to find that out (`best` only grows, so a lane that fails this fails it for good).
	auto more = [&](int d) {
		const int n6 = 6 * (d + 1);
		const int lo = max(n6 - slack_s + 1, -(d + 1)), hi = min(slack_q - n6 - 1, d + 1);
		return lo <= hi;
	};
	bool done = false;
	static_for<1, kGUnrollLevels + 1>([&](auto dc) {
		constexpr int d = decltype(dc)::value;
		if (!done) {
			if (__ballot(live) == 0ull) {
				done = true;
			} else {
				prev = kCellNone;
				any = false;
				GAP_STAT(3, 1);
				GAP_STAT(7, __popcll(__ballot(live)));
				static_for<C - d, C + d + 1>([&](auto cc) { cell(cc, 6 * d, std::integral_constant<int, d - 1>{}); });
				live = any && more(d);
				if constexpr (D >= kGLag)
					xring[d] = tbest; // (d <= kGUnrollLevels <= D - kGLag)
			}
		}
	});
	for (int d = kGUnrollLevels + 1; !done && __ballot(live) != 0ull; d++) {
		if (d > D) {
			over = live;
			break;
		}
		if constexpr (D >= kGLag)
			tcmp = d >= kGLag ? xring[d - kGLag] - kGX2 : -(1 << 29);
		prev = kCellNone;
		any = false;
		GAP_STAT(3, 1);
		GAP_STAT(7, __popcll(__ballot(live)));
		const int six_d = 6 * d;
		static_for<0, (kCells - 2 + kGGroup - 1) / kGGroup>([&](auto gc) {
			constexpr int c0 = 1 + decltype(gc)::value * kGGroup;
			if (!(c0 + kGGroup - 1 - C < -d || c0 - C > d)) {
				static_for<c0, (c0 + kGGroup < kCells - 1 ? c0 + kGGroup : kCells - 1)>([&](auto cc) {
					constexpr int k = decltype(cc)::value - C;
					if (!(k < -d || k > d))
						cell(cc, six_d, std::integral_constant<int, D>{});
				});
			}
		});
		live = any && more(d);
		if constexpr (D >= kGLag)
			if (d <= D - kGLag)
				xring[d] = tbest;
	}
	out.i = (int)best_cell >> 16;
	out.j = out.i - best_k;
	out.s2 = best > 2 * i0 ? best : 2 * i0; // (no cell passed the first run: B0 was that run)
	out.mism = (int)((best_cell >> 8) & 63u);
	out.gopen = (int)((best_cell >> 2) & 63u);
	return !over;
}

// ------------------------------------------------------------------------------------------ the lean rows (first tier)
// Round 3.  tools/probe_issue.py measured what a SIMD of this chip issues per instruction KIND (profiles/r03_issue_table.txt):
// v_add / v_sub / v_and / v_or / v_xor / v_lshrrev / v_ashrrev / v_mov on registers and inline constants run at one
// wavefront-instruction per ~2.2 cycles, EVERYTHING else (v_max3, v_min, v_cmp, v_cndmask, v_alignbit, v_ffbl, v_bfe,
// v_lshlrev, any three-operand form, anything with a scalar-register or SDWA operand) at one per ~4.1.  The cell step above
// is ~55 vector instructions, most of the second kind: ~200 SIMD cycles per cell step of the 281 measured, so the stage
// IS vector-bound, and what helps is fewer and cheaper instructions per cell.  This form of the cell:
//   * the cell word is  i << 17 | 0 << 16 | priority << 14 | G1  (G1 = gap columns in the subject row on the path).  The
//     other statistics FOLLOW: on diagonal k the path holds G1 - k gap columns in the query row, a level is one difference,
//     so mismatches = d - gap columns; gap OPENINGS are the gap columns while there is at most one of each kind (the
//     greedy rule picks an insertion + deletion pair over two mismatches in 9 % of the bench's HSPs: the furthest row
//     counts, not the score).  A side whose best cell holds two or more gap columns of ONE kind (1 %) is handed to the next tier, which carries the full statistics (the cells, their
//     order and the best cell are the same with or without the statistics: nothing below the priority takes part in a
//     comparison that can tie).  No traceback, no per-cell statistics update.
//   * with bit 16 clear, (word + Q) >> 16 IS the bit offset of read letter i in the lane's staged letters, (word + Q) >> 13
//     its byte offset in the TRANSPOSED letter rows (row r of lane l at r * 256 + l * 4: a lane always reads its own LDS
//     bank, whatever the other lanes' rows are -- the per-lane rows of the first form conflicted 2.4 x per read);
//     the database letter's offsets are the same word plus a per-diagonal constant.
//   * a dead cell is i = -4 (not -32768): its candidates stay negative, and every address a dead or out-of-range cell forms
//     stays inside the lane's rows (two spare rows in front), so no cell is branched around: the letters are fetched and
//     compared for all 64 lanes and a select writes the dead ones -- straight code the compiler interleaves across cells,
//     the LDS latency of one cell behind the arithmetic of the next.
//   * the best cell is a running MAXIMUM of  score << 16 | (2047 - cell order) << 5 | G1  (first cell in (d, k) order among
//     equal scores), one add and one max per cell; the bound is taken once per level from it (the cut never changes a
//     result, see the header; the oracle has no cut at all).
typedef __attribute__((address_space(3))) uint32_t lds_word;
typedef __attribute__((address_space(3))) char lds_byte;
// (the rows' base stays a link-time constant that folds into the instruction's offset field; `off` = row * 256 | lane * 4)
__device__ __forceinline__ uint32_t lds_ld(const lds_word *seq0, uint32_t off) { return *(const lds_word *)((const lds_byte *)seq0 + off); }
__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t r;
	asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}

#ifndef PGX_LEAN_GROUP
#define PGX_LEAN_GROUP 2
#endif
#ifndef PGX_LEAN_WAVES
#define PGX_LEAN_WAVES 4
#endif
constexpr int kLeanGroup = PGX_LEAN_GROUP; // cells computed as one piece of straight code (their LDS reads in flight together)
constexpr uint32_t kLeanDead = 0xFFF80000u; // i = -4
constexpr uint32_t kLeanFromCur = 0x28000u;  // i + 1, priority 2 (a mismatch on this diagonal)
constexpr uint32_t kLeanFromPrev = 0x24001u; // i + 1, priority 1, one more gap column in the subject row (from k - 1)
constexpr int kLeanFrontRows = 2;            // spare rows in front of a lane's letters

// matching letters from bit offsets qb / db of the lane's staged letters on, as BITS (2 per letter): min(32, cap2, 2 run)
__device__ __forceinline__ uint32_t lean_lcp(const lds_word *seq0, uint32_t lane4, uint32_t qb, uint32_t db, uint32_t cap2)
{
	const uint32_t qa = ((qb << 3) & 0xFFFFFF00u) | lane4, da = ((db << 3) & 0xFFFFFF00u) | lane4;
	const uint32_t x = __builtin_amdgcn_alignbit(lds_ld(seq0, qa + 256), lds_ld(seq0, qa), qb) ^ __builtin_amdgcn_alignbit(lds_ld(seq0, da + 256), lds_ld(seq0, da), db);
	const uint32_t y = (x | (x >> 1)) & 0x55555555u;
	return min3u(scan_low(y), 32u, cap2);
}

// status: 0 done, 1 cells alive after D differences (next tier), 2 the best cell holds two or more gap columns of a kind (next tier)
// seq0 = the letter rows (row r of lane l at byte r * 256 + l * 4), lane4 = l * 4; QB / DB0 = bit offsets of the side's first read / database
// letter there; M, N = letters of the read / of the subject on the side; b0 as for greedy_rows
template <int D>
__device__ __forceinline__ int greedy_rows_lean(const lds_word *seq0, uint32_t lane4, bool on, int QB, int DB0, int M, int N, int b0, Side &out)
{
	static_assert(D < kGLag && D < 32, "no X-drop history, 5-bit fields");
	constexpr int kCells = 2 * D + 3, C = D + 1;
	const int A2 = 2 * M + QB, B2 = 2 * N + QB; // ends of the two sequences as read bit offsets (B2 + 2 k on diagonal k)
	// ---- the first run
	int i2 = 0; // 2 i
	if (on) {
		for (;;) {
			const int cap2 = min(2 * M, 2 * N) - i2;
			if (cap2 <= 0)
				break;
			const uint32_t r2 = lean_lcp(seq0, lane4, (uint32_t)(QB + i2), (uint32_t)(DB0 + i2), (uint32_t)cap2);
			i2 += (int)r2;
			if (r2 < 32u)
				break;
		}
	}
	out.i = out.j = i2 >> 1;
	out.s2 = i2;
	out.mism = out.gopen = 0;
	bool live = on && !(i2 == 2 * M || i2 == 2 * N);
	uint32_t R[kCells];
#pragma unroll
	for (int c = 0; c < kCells; c++)
		R[c] = kLeanDead;
	R[C] = live ? (uint32_t)i2 << 16 : kLeanDead;
	// order of a cell: d * 64 + k + 32 (the first run: 32); key = s2 << 16 | (2047 - order) << 5 | G1
	int best_key = (i2 << 16) | ((2047 - 32) << 5);
	const uint32_t Qc = (uint32_t)QB << 16;
	const uint32_t D0 = (uint32_t)(DB0 - QB) << 16;
	const uint32_t c17 = 0x20000u;
	uint32_t prev = kLeanDead;
	// per level, per lane: the diagonals k the bound lets live are lo1 .. lo1 + width (none: lo1 = 1 << 20)
	int lo1 = 0;
	uint32_t width = 0;
	uint32_t alldead = 0xFFFFFFFFu; // AND of the level's new cells: negative while every one of them is dead
	auto set_range = [&](int d) { // for level d, from the best score so far
		const int best = max(best_key >> 16, b0 - 1);
		const int lo = 6 * d - (2 * N - best) + 1, hi = (2 * M - best) - 6 * d - 1; // lo <= k <= hi
		const bool some = lo <= hi;
		lo1 = some ? lo : (1 << 20);
		width = some ? (uint32_t)(hi - lo) : 0u;
		return some && lo <= d && hi >= -d;
	};
	// running per-cell values of a level (k ascending): Dk = D0 - k << 17, Bk = B2 + 2 k, rk = k - lo1
	uint32_t Dk = 0, rk = 0;
	int Bk = 0;
	auto level_start = [&](int d) {
		Dk = D0 + ((uint32_t)d << 17);
		Bk = B2 - 2 * d;
		rk = (uint32_t)(-d - lo1);
		prev = kLeanDead;
		alldead = 0xFFFFFFFFu;
	};
	// one cell; c compile-time; ckd = -(k + 6 d) << 16 | (2047 - order) << 5, wave-uniform
	auto cell = [&](auto cc, auto reach_c, int ckd, uint32_t &r2_out) {
		constexpr int c = decltype(cc)::value;
		constexpr int k = c - C;
		constexpr int reach = decltype(reach_c)::value;
		const uint32_t cur = R[c], nxt = R[c + 1];
		int m3;
		if constexpr (k < -reach)
			m3 = (int)nxt; // the left edge of the level: only diagonal k + 1 can lead here
		else if constexpr (k > reach)
			m3 = (int)(prev + kLeanFromPrev); // the right edge: only diagonal k - 1
		else if constexpr (k - 1 < -reach && k + 1 > reach)
			m3 = (int)(cur + kLeanFromCur); // level 1, k = 0
		else if constexpr (k - 1 < -reach)
			m3 = max((int)(cur + kLeanFromCur), (int)nxt);
		else if constexpr (k + 1 > reach)
			m3 = max((int)(cur + kLeanFromCur), (int)(prev + kLeanFromPrev));
		else
			m3 = max((int)(cur + kLeanFromCur), max((int)(prev + kLeanFromPrev), (int)nxt));
		const uint32_t t = (uint32_t)m3 + Qc, u = t + Dk;
		const uint32_t qbit = (uint32_t)((int)t >> 16), dbit = (uint32_t)((int)u >> 16);
		const uint32_t qa = ((t >> 13) & 0xFFFFFF00u) | lane4, da = ((u >> 13) & 0xFFFFFF00u) | lane4;
		const int cap2 = min(A2, Bk) - (int)qbit;
		const bool ok = ((m3 | cap2) >= 0) & (rk <= width);
		const uint32_t x = __builtin_amdgcn_alignbit(lds_ld(seq0, qa + 256), lds_ld(seq0, qa), qbit) ^ __builtin_amdgcn_alignbit(lds_ld(seq0, da + 256), lds_ld(seq0, da), dbit);
		const uint32_t y = (x | (x >> 1)) & 0x55555555u;
		const uint32_t r2 = min3u(scan_low(y), 32u, (uint32_t)cap2);
		const uint32_t nc = ((uint32_t)m3 & 0xFFFF3FFFu) + (r2 << 16);
		const uint32_t nv = ok ? nc : kLeanDead;
		best_key = max(best_key, (int)(nv + (uint32_t)ckd));
		alldead &= nv;
		r2_out = r2;
		prev = cur;
		R[c] = nv;
		Dk -= c17;
		Bk += 2;
		rk += 1u;
	};
	// a cell whose first 16 letters all matched and that has more to compare: slide on (rare off the alignment's own diagonal)
	auto slide_on = [&](auto cc, int ckd, uint32_t r2) {
		constexpr int c = decltype(cc)::value;
		constexpr int k = c - C;
		const uint32_t w = R[c];
		bool more = (int)w >= 0 && r2 == 32u;
		if (__ballot(more) == 0ull)
			return;
		int q2 = (int)w >> 16; // 2 i
		while (more) {
			const int cap2 = min(A2, B2 + 2 * k) - (QB + q2);
			if (cap2 <= 0)
				break;
			const uint32_t rr = lean_lcp(seq0, lane4, (uint32_t)(QB + q2), (uint32_t)(DB0 + q2 - 2 * k), (uint32_t)cap2);
			q2 += (int)rr;
			more = rr == 32u;
		}
		const uint32_t nv = (int)w >= 0 ? ((uint32_t)q2 << 16) | (w & 0xFFFFu) : w;
		R[c] = nv;
		best_key = max(best_key, (int)(nv + (uint32_t)ckd));
	};
	// cells c0 .. c0 + n - 1 are dead for every lane: written so, the running values stepped past them
	auto skip_group = [&](auto c0c, auto nc) {
		constexpr int c0 = decltype(c0c)::value, n = decltype(nc)::value;
		prev = R[c0 + n - 1];
		static_for<0, n>([&](auto jc) { R[c0 + decltype(jc)::value] = kLeanDead; });
		Dk -= c17 * (uint32_t)n;
		Bk += 2 * n;
		rk += (uint32_t)n;
	};
	auto ckd_of = [](int d, int k) { return (int)((uint32_t)(-(k + 6 * d)) << 16) | ((2047 - (d * 64 + k + 32)) << 5); };
	bool done = false, over = false;
	bool in_range = set_range(1);
	live = live && in_range;
	static_for<1, kGUnrollLevels + 1>([&](auto dc) {
		constexpr int d = decltype(dc)::value;
		if (!done) {
			if (__ballot(live) == 0ull) {
				done = true;
			} else {
				level_start(d);
				// groups of cells: straight code, then the (rare) longer slides of the group; a group no lane's bound lets
				// live (the dead half of a side's last levels) is written dead without looking
				static_for<0, (2 * d + 1 + kLeanGroup - 1) / kLeanGroup>([&](auto gc) {
					constexpr int c0 = C - d + kLeanGroup * decltype(gc)::value;
					constexpr int n = (C + d + 1 - c0) < kLeanGroup ? (C + d + 1 - c0) : kLeanGroup;
					if (__ballot(lo1 <= c0 + n - 1 - C && lo1 + (int)width >= c0 - C) == 0ull) {
						skip_group(std::integral_constant<int, c0>{}, std::integral_constant<int, n>{});
					} else {
						uint32_t r2[kLeanGroup] = {};
						uint32_t any32 = 0;
						static_for<0, n>([&](auto jc) {
							constexpr int c = c0 + decltype(jc)::value;
							cell(std::integral_constant<int, c>{}, std::integral_constant<int, d - 1>{}, ckd_of(d, c - C), r2[decltype(jc)::value]);
							any32 |= r2[decltype(jc)::value];
						});
						if (__ballot((any32 & 32u) != 0u) != 0ull) {
							static_for<0, n>([&](auto jc) {
								constexpr int c = c0 + decltype(jc)::value;
								slide_on(std::integral_constant<int, c>{}, ckd_of(d, c - C), r2[decltype(jc)::value]);
							});
						}
					}
				});
				const bool some = set_range(d + 1);
				live = (int)alldead >= 0 && some;
			}
		}
	});
	for (int d = kGUnrollLevels + 1; !done && __ballot(live) != 0ull; d++) {
		if (d > D) {
			over = live;
			break;
		}
		level_start(d);
		// (the running values start at k = -d: cells left of it are skipped without a step)
		static_for<0, (kCells - 2 + kLeanGroup - 1) / kLeanGroup>([&](auto gc) {
			constexpr int c0 = 1 + decltype(gc)::value * kLeanGroup;
			constexpr int n = (kCells - 1 - c0) < kLeanGroup ? (kCells - 1 - c0) : kLeanGroup;
			if (!(c0 + n - 1 - C < -d || c0 - C > d)) {
				if (c0 - C >= -d && c0 + n - 1 - C <= d) {
					// the whole group lies inside the level: the same straight code as above
					if (__ballot(lo1 <= c0 + n - 1 - C && lo1 + (int)width >= c0 - C) == 0ull) {
						skip_group(std::integral_constant<int, c0>{}, std::integral_constant<int, n>{});
					} else {
						uint32_t r2[kLeanGroup] = {};
						uint32_t any32 = 0;
						static_for<0, n>([&](auto jc) {
							constexpr int c = c0 + decltype(jc)::value;
							cell(std::integral_constant<int, c>{}, std::integral_constant<int, D>{}, ckd_of(d, c - C), r2[decltype(jc)::value]);
							any32 |= r2[decltype(jc)::value];
						});
						if (__ballot((any32 & 32u) != 0u) != 0ull) {
							static_for<0, n>([&](auto jc) {
								constexpr int c = c0 + decltype(jc)::value;
								slide_on(std::integral_constant<int, c>{}, ckd_of(d, c - C), r2[decltype(jc)::value]);
							});
						}
					}
				} else {
					// a group the level's ends cut through: cell by cell
					static_for<0, n>([&](auto jc) {
						constexpr int c = c0 + decltype(jc)::value;
						constexpr int k = c - C;
						if (!(k < -d || k > d)) {
							uint32_t r2 = 0;
							cell(std::integral_constant<int, c>{}, std::integral_constant<int, D>{}, ckd_of(d, k), r2);
							if (__ballot((r2 & 32u) != 0u) != 0ull)
								slide_on(std::integral_constant<int, c>{}, ckd_of(d, k), r2);
						}
					});
				}
			}
		});
		const bool some = set_range(d + 1);
		live = (int)alldead >= 0 && some;
	}
	// the best cell: score, where, its gap columns
	const int s2 = best_key >> 16;
	const int order = 2047 - ((best_key >> 5) & 2047);
	const int bd = order >> 6, bk = (order & 63) - 32;
	const int g1 = best_key & 31, g2 = g1 - bk;
	out.i = (s2 + bk + 6 * bd) >> 1;
	out.j = out.i - bk;
	out.s2 = s2;
	out.gopen = g1 + g2;
	out.mism = bd - (g1 + g2);
	if (over)
		return 1;
	// (a gap column in the subject row and one in the query row are two gaps; only two columns of one kind can be one gap or two)
	return g1 >= 2 || g2 >= 2 ? 2 : 0;
}

// The lean rows for D >= kGLag differences a side (reads of 320-512 bases: at 7 % divergence most sides of a 500-base read
// need 19-40 levels; round 2's 40-difference tier with the per-cell statistics took 73 of the 145 ms per 1 M such reads).
// Differences to greedy_rows_lean: the X-drop test (T[d - 19] in a 19-register ring that shifts once per level); the
// cell's low 14 bits carry gap openings [5:0] | open-gap kind [7:6] | G1 [13:8] as in k_gapped_diag (with 19-40 levels
// most best paths hold two gap columns of a kind, which the G1-only word cannot count), the best cell's word is kept
// beside the running maximum of  score << 16 | 65535 - (d * 128 + k + 64).
// status: 0 done, 1 cells alive after D differences (next tier)
// seq0 = the letter rows (row r of lane l at byte r * 256 + l * 4), lane4 = l * 4; QB / DB0 = bit offsets of the side's first read / database
// letter there; M, N = letters of the read / of the subject on the side; b0 as for greedy_rows
constexpr uint32_t kDeepFromPrev = 0x24100u; // i + 1, priority 1, G1 + 1 (G1 at bit 8)
template <int D>
__device__ __forceinline__ int greedy_rows_deep(const lds_word *seq0, uint32_t lane4, bool on, int QB, int DB0, int M, int N, int b0, Side &out)
{
	static_assert(D >= kGLag && D < 62, "X-drop history kept, 6-bit fields");
	constexpr int kCells = 2 * D + 3, C = D + 1;
	const int A2 = 2 * M + QB, B2 = 2 * N + QB; // ends of the two sequences as read bit offsets (B2 + 2 k on diagonal k)
	// ---- the first run
	int i2 = 0; // 2 i
	if (on) {
		for (;;) {
			const int cap2 = min(2 * M, 2 * N) - i2;
			if (cap2 <= 0)
				break;
			const uint32_t r2 = lean_lcp(seq0, lane4, (uint32_t)(QB + i2), (uint32_t)(DB0 + i2), (uint32_t)cap2);
			i2 += (int)r2;
			if (r2 < 32u)
				break;
		}
	}
	out.i = out.j = i2 >> 1;
	out.s2 = i2;
	out.mism = out.gopen = 0;
	bool live = on && !(i2 == 2 * M || i2 == 2 * N);
	uint32_t R[kCells];
#pragma unroll
	for (int c = 0; c < kCells; c++)
		R[c] = kLeanDead;
	R[C] = live ? ((uint32_t)i2 << 16) | 0x80u : kLeanDead;
	// order of a cell: d * 128 + k + 64 (the first run: 64); key = s2 << 16 | 65535 - order
	int best_key = (i2 << 16) | (65535 - 64);
	uint32_t best_word = ((uint32_t)i2 << 16) | 0x80u;
	int Tr[kGLag]; // T[d - 19 .. d - 1] while level d runs (T[x] = best score within x differences; x < 0: no test)
#pragma unroll
	for (int x = 0; x < kGLag; x++)
		Tr[x] = -(1 << 14);
	Tr[kGLag - 1] = i2;
	int thrk = 0; // running per cell: (T[d - 19] - 2 X + 6 d + k) << 16, the X-drop threshold on the word's upper half
	const uint32_t Qc = (uint32_t)QB << 16;
	const uint32_t D0 = (uint32_t)(DB0 - QB) << 16;
	const uint32_t c17 = 0x20000u;
	uint32_t prev = kLeanDead;
	// per level, per lane: the diagonals k the bound lets live are lo1 .. lo1 + width (none: lo1 = 1 << 20)
	int lo1 = 0;
	uint32_t width = 0;
	uint32_t alldead = 0xFFFFFFFFu; // AND of the level's new cells: negative while every one of them is dead
	auto set_range = [&](int d) { // for level d, from the best score so far
		const int best = max(best_key >> 16, b0 - 1);
		const int lo = 6 * d - (2 * N - best) + 1, hi = (2 * M - best) - 6 * d - 1; // lo <= k <= hi
		const bool some = lo <= hi;
		lo1 = some ? lo : (1 << 20);
		width = some ? (uint32_t)(hi - lo) : 0u;
		return some && lo <= d && hi >= -d;
	};
	// running per-cell values of a level (k ascending): Dk = D0 - k << 17, Bk = B2 + 2 k, rk = k - lo1
	uint32_t Dk = 0, rk = 0;
	int Bk = 0;
	auto level_start = [&](int d) {
		Dk = D0 + ((uint32_t)d << 17);
		Bk = B2 - 2 * d;
		rk = (uint32_t)(-d - lo1);
		prev = kLeanDead;
		alldead = 0xFFFFFFFFu;
		thrk = (int)((uint32_t)(Tr[0] - kGX2 + 6 * d - d) << 16);
	};
	// one cell; c compile-time; ckd = -(k + 6 d) << 16 | (2047 - order) << 5, wave-uniform
	auto cell = [&](auto cc, auto reach_c, int ckd, uint32_t &r2_out) {
		constexpr int c = decltype(cc)::value;
		constexpr int k = c - C;
		constexpr int reach = decltype(reach_c)::value;
		const uint32_t cur = R[c], nxt = R[c + 1];
		int m3;
		if constexpr (k < -reach)
			m3 = (int)nxt; // the left edge of the level: only diagonal k + 1 can lead here
		else if constexpr (k > reach)
			m3 = (int)(prev + kDeepFromPrev); // the right edge: only diagonal k - 1
		else if constexpr (k - 1 < -reach && k + 1 > reach)
			m3 = (int)(cur + kLeanFromCur); // level 1, k = 0
		else if constexpr (k - 1 < -reach)
			m3 = max((int)(cur + kLeanFromCur), (int)nxt);
		else if constexpr (k + 1 > reach)
			m3 = max((int)(cur + kLeanFromCur), (int)(prev + kDeepFromPrev));
		else
			m3 = max((int)(cur + kLeanFromCur), max((int)(prev + kDeepFromPrev), (int)nxt));
		const uint32_t t = (uint32_t)m3 + Qc, u = t + Dk;
		const uint32_t qbit = (uint32_t)((int)t >> 16), dbit = (uint32_t)((int)u >> 16);
		const uint32_t qa = ((t >> 13) & 0xFFFFFF00u) | lane4, da = ((u >> 13) & 0xFFFFFF00u) | lane4;
		const int cap2 = min(A2, Bk) - (int)qbit;
		bool ok = ((m3 | cap2) >= 0) & (rk <= width);
		if constexpr (reach >= kGLag - 1) // (levels below kGLag pass every X-drop test)
			ok = ok & ((int)((uint32_t)m3 & 0xFFFF0000u) >= thrk);
		const uint32_t x = __builtin_amdgcn_alignbit(lds_ld(seq0, qa + 256), lds_ld(seq0, qa), qbit) ^ __builtin_amdgcn_alignbit(lds_ld(seq0, da + 256), lds_ld(seq0, da), dbit);
		const uint32_t y = (x | (x >> 1)) & 0x55555555u;
		const uint32_t r2 = min3u(scan_low(y), 32u, (uint32_t)cap2);
		// the winning parent's statistics -> this cell's (see k_gapped_diag)
		const uint32_t w = (uint32_t)m3;
		const uint32_t pk = (w >> 8) & 0xC0u;
		const uint32_t differs = ((w ^ pk) & 0xC0u) != 0u ? 1u : 0u;
		const uint32_t inc = (w & 0x8000u) ? 0u : differs;
		const uint32_t nc = ((w & 0xFFFF0000u) + (r2 << 16)) | (((w & 0x3F3Fu) + inc) | (r2 != 0u ? 0x80u : pk));
		const uint32_t nv = ok ? nc : kLeanDead;
		{
			const int key = (int)((nv & 0xFFFF0000u) + (uint32_t)ckd);
			best_word = key > best_key ? nv : best_word;
			best_key = max(best_key, key);
		}
		alldead &= nv;
		r2_out = r2;
		prev = cur;
		R[c] = nv;
		Dk -= c17;
		Bk += 2;
		rk += 1u;
		thrk += 0x10000;
	};
	// a cell whose first 16 letters all matched and that has more to compare: slide on (rare off the alignment's own diagonal)
	auto slide_on = [&](auto cc, int ckd, uint32_t r2) {
		constexpr int c = decltype(cc)::value;
		constexpr int k = c - C;
		const uint32_t w = R[c];
		bool more = (int)w >= 0 && r2 == 32u;
		if (__ballot(more) == 0ull)
			return;
		int q2 = (int)w >> 16; // 2 i
		while (more) {
			const int cap2 = min(A2, B2 + 2 * k) - (QB + q2);
			if (cap2 <= 0)
				break;
			const uint32_t rr = lean_lcp(seq0, lane4, (uint32_t)(QB + q2), (uint32_t)(DB0 + q2 - 2 * k), (uint32_t)cap2);
			q2 += (int)rr;
			more = rr == 32u;
		}
		const uint32_t nv = (int)w >= 0 ? ((uint32_t)q2 << 16) | (w & 0xFFFFu) : w;
		R[c] = nv;
		{
			const int key = (int)((nv & 0xFFFF0000u) + (uint32_t)ckd);
			best_word = key > best_key ? nv : best_word;
			best_key = max(best_key, key);
		}
	};
	// cells c0 .. c0 + n - 1 are dead for every lane: written so, the running values stepped past them
	auto skip_group = [&](auto c0c, auto nc) {
		constexpr int c0 = decltype(c0c)::value, n = decltype(nc)::value;
		prev = R[c0 + n - 1];
		static_for<0, n>([&](auto jc) { R[c0 + decltype(jc)::value] = kLeanDead; });
		Dk -= c17 * (uint32_t)n;
		Bk += 2 * n;
		rk += (uint32_t)n;
		thrk += 0x10000 * n;
	};
	auto ckd_of = [](int d, int k) { return (int)((uint32_t)(-(k + 6 * d)) << 16) | (65535 - (d * 128 + k + 64)); };
	auto level_end = [&]() { // T[d] = the best score so far; the ring moves on
#pragma unroll
		for (int x = 0; x + 1 < kGLag; x++)
			Tr[x] = Tr[x + 1];
		Tr[kGLag - 1] = best_key >> 16;
	};
	bool done = false, over = false;
	bool in_range = set_range(1);
	live = live && in_range;
	static_for<1, kGUnrollLevels + 1>([&](auto dc) {
		constexpr int d = decltype(dc)::value;
		if (!done) {
			if (__ballot(live) == 0ull) {
				done = true;
			} else {
				level_start(d);
				// groups of cells: straight code, then the (rare) longer slides of the group; a group no lane's bound lets
				// live (the dead half of a side's last levels) is written dead without looking
				static_for<0, (2 * d + 1 + kLeanGroup - 1) / kLeanGroup>([&](auto gc) {
					constexpr int c0 = C - d + kLeanGroup * decltype(gc)::value;
					constexpr int n = (C + d + 1 - c0) < kLeanGroup ? (C + d + 1 - c0) : kLeanGroup;
					if (__ballot(lo1 <= c0 + n - 1 - C && lo1 + (int)width >= c0 - C) == 0ull) {
						skip_group(std::integral_constant<int, c0>{}, std::integral_constant<int, n>{});
					} else {
						uint32_t r2[kLeanGroup] = {};
						uint32_t any32 = 0;
						static_for<0, n>([&](auto jc) {
							constexpr int c = c0 + decltype(jc)::value;
							cell(std::integral_constant<int, c>{}, std::integral_constant<int, d - 1>{}, ckd_of(d, c - C), r2[decltype(jc)::value]);
							any32 |= r2[decltype(jc)::value];
						});
						if (__ballot((any32 & 32u) != 0u) != 0ull) {
							static_for<0, n>([&](auto jc) {
								constexpr int c = c0 + decltype(jc)::value;
								slide_on(std::integral_constant<int, c>{}, ckd_of(d, c - C), r2[decltype(jc)::value]);
							});
						}
					}
				});
				const bool some = set_range(d + 1);
				live = (int)alldead >= 0 && some;
				level_end();
			}
		}
	});
	for (int d = kGUnrollLevels + 1; !done && __ballot(live) != 0ull; d++) {
		if (d > D) {
			over = live;
			break;
		}
		level_start(d);
		// (the running values start at k = -d: cells left of it are skipped without a step)
		static_for<0, (kCells - 2 + kLeanGroup - 1) / kLeanGroup>([&](auto gc) {
			constexpr int c0 = 1 + decltype(gc)::value * kLeanGroup;
			constexpr int n = (kCells - 1 - c0) < kLeanGroup ? (kCells - 1 - c0) : kLeanGroup;
			if (!(c0 + n - 1 - C < -d || c0 - C > d)) {
				if (c0 - C >= -d && c0 + n - 1 - C <= d) {
					// the whole group lies inside the level: the same straight code as above
					if (__ballot(lo1 <= c0 + n - 1 - C && lo1 + (int)width >= c0 - C) == 0ull) {
						skip_group(std::integral_constant<int, c0>{}, std::integral_constant<int, n>{});
					} else {
						uint32_t r2[kLeanGroup] = {};
						uint32_t any32 = 0;
						static_for<0, n>([&](auto jc) {
							constexpr int c = c0 + decltype(jc)::value;
							cell(std::integral_constant<int, c>{}, std::integral_constant<int, D>{}, ckd_of(d, c - C), r2[decltype(jc)::value]);
							any32 |= r2[decltype(jc)::value];
						});
						if (__ballot((any32 & 32u) != 0u) != 0ull) {
							static_for<0, n>([&](auto jc) {
								constexpr int c = c0 + decltype(jc)::value;
								slide_on(std::integral_constant<int, c>{}, ckd_of(d, c - C), r2[decltype(jc)::value]);
							});
						}
					}
				} else {
					// a group the level's ends cut through: cell by cell
					static_for<0, n>([&](auto jc) {
						constexpr int c = c0 + decltype(jc)::value;
						constexpr int k = c - C;
						if (!(k < -d || k > d)) {
							uint32_t r2 = 0;
							cell(std::integral_constant<int, c>{}, std::integral_constant<int, D>{}, ckd_of(d, k), r2);
							if (__ballot((r2 & 32u) != 0u) != 0ull)
								slide_on(std::integral_constant<int, c>{}, ckd_of(d, k), r2);
						}
					});
				}
			}
		});
		const bool some = set_range(d + 1);
		live = (int)alldead >= 0 && some;
		level_end();
	}
	// the best cell: score, where, its statistics
	const int s2 = best_key >> 16;
	const int order = 65535 - (best_key & 65535);
	const int bd = order >> 7, bk = (order & 127) - 64;
	const int g1 = (int)((best_word >> 8) & 63u), g2 = g1 - bk;
	out.i = (s2 + bk + 6 * bd) >> 1;
	out.j = out.i - bk;
	out.s2 = s2;
	out.gopen = (int)(best_word & 63u);
	out.mism = bd - (g1 + g2);
	return over ? 1 : 0;
}

constexpr int kBlkItems = 2048; // HSPs a wavefront orders at a time
constexpr int kKeyBuckets = 16;  // mismatches of the diagonal on one side of the seed run, capped at 15

template <int MAXL, int D, bool LEAN> struct FastLds {
	static constexpr int kRd = MAXL / 16 + 2;                    // read strand, 16 bases per word
	static constexpr int kDb = (MAXL + 2 * D + 48 + 15) / 16 + 1; // database window
	static constexpr int kSeq = (kRd + kDb) | 1; // odd stride: lanes that use the same index hit different banks
	static constexpr int kRing = D >= kGLag ? ((D - kGLag + 2) | 1) : 1; // the X-drop history of a lane (second tier)
	static constexpr bool kLean = LEAN; // first tier: TRANSPOSED rows (row r of lane l at seq[r * 64 + l]), spare rows in front
	static constexpr int kRows = kLeanFrontRows + kRd + kDb;
	uint32_t seq[kLean ? kRows * 64 : 64 * kSeq];
	int xring[D >= kGLag ? 64 : 0][kRing]; // (nothing in the first tier: 256 bytes more there cost a fifth wavefront its LDS)
	uint32_t bucket[kKeyBuckets];
	// (the pool's HSPs in cost order live in GLOBAL scratch, one array per block: with them here the kernel held 12.6 KB of
	// LDS per wavefront = 3 wavefronts per SIMD; without, 8.5 KB = 4, and a wavefront of this kernel is bound by its own
	// instruction issue, so the fourth one is worth a fifth of the stage: 74.6 -> 64.8 ms per 10 M reads)
	// (no copy of the work keys: 2 KB more LDS costs a wavefront per SIMD, and the rows are bound by vector issue:
	// 12.6 KB -> 14.7 KB per wavefront ran 24 -> 31 ms per 2 M reads; 1 024-HSP chunks fill the buckets too thinly: 31 ms)
};

// An HSP this tier cannot finish goes on the next tier's list.  Entries past the list's capacity are dropped -- the host
// sees the count and repeats the step with a larger list -- but their records are still SEED records (word offsets and
// work estimates where a hit has coordinates), which the stages behind must never read as hits: those are made harmless.
__device__ __forceinline__ void neutral_hit(pgx_hit *hp)
{
	pgx_hit o = *hp;
	o.qstart = o.qend = o.sstart = o.send = 1;
	o.score = 0;
	o.mismatch = o.gapopen = 0;
	*hp = o;
}

// one atomic per call (a single address takes ~90 M atomics a second: one per listed HSP would cost more than the rows)
// `also`: a second counter for the same entries (the first tier's own appends to a list that a later tier appends to as well)
__device__ __forceinline__ void list_append(bool fail, pgx_hit *hp, unsigned long long *__restrict__ list, uint32_t *__restrict__ count, uint32_t cap,
					    uint32_t *__restrict__ also = nullptr)
{
	const unsigned long long m = __ballot(fail);
	if (m == 0ull)
		return;
	const int lane = threadIdx.x & 63, first = __ffsll((unsigned long long)m) - 1;
	uint32_t base = 0;
	if (lane == first) {
		base = atomicAdd(count, (uint32_t)__popcll(m));
		if (also)
			atomicAdd(also, (uint32_t)__popcll(m));
	}
	base = __shfl(base, first);
	if (fail) {
		const uint32_t w = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
		if (w < cap)
			list[w] = (unsigned long long)(uintptr_t)hp;
		else
			neutral_hit(hp);
	}
}

// FLAT: the table is a flat array of *count hits (overflow table); otherwise the hits of read r are the read_cnt[r] records
// from read_start[r] of the seed stage's main table (kFragmented: they are in the overflow table).
// A wavefront takes the HSPs of 64 reads (at most kBlkItems at a time), orders them by the seed stage's work estimate
// (a counting sort in LDS), and runs them 64 at a time: lanes of one round have about the same number of rows.
// MODE 0: the reads' pools of the main table; 1 (FLAT): the overflow table; 2 (LIST): the HSPs the first tier listed, as
// pointers to their seed records (`table` = the list as pgx_hit **; a side's result is parked by list position).
// D: differences per side (18 first tier, 40 second tier, which makes the X-drop tests).
template <int MODE, int MAXL, int WAVES, int D>
// (WAVES per SIMD: 4 = 128 registers -- the row is 39 of them; the compiler parks three values in scratch around the rows;
// 5 = 96 registers, 24 values in scratch, for reads of <= 160 bases whose staged letters fit 7.5 KB of LDS: worth 6 % once
// the grid is two full rounds of resident wavefronts -- with 8 192 blocks on 5 120 slots the second round ran at 60 %)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES))) void k_gapped_fast(GapView v, pgx_hit *__restrict__ table, unsigned long long table_cap,
						     const uint32_t *__restrict__ read_start, const uint32_t *__restrict__ read_cnt,
						     uint32_t n_reads, const unsigned long long *__restrict__ flat_count,
						     unsigned long long *__restrict__ big_list, uint32_t *__restrict__ big_count, uint32_t big_cap,
						     uint2 *__restrict__ side_res, int dbg, uint32_t *__restrict__ order_all)
{
	constexpr bool FLAT = MODE != 0, LIST = MODE == 2;
	constexpr bool LEAN = MODE != 2 && D < kGLag; // the first tier over the tables; over a list: the rows with the full statistics
	using Lds = FastLds<MAXL, D, LEAN>;
	__shared__ Lds lds;
	uint32_t *order = order_all + (size_t)blockIdx.x * kBlkItems; // written and read by this wavefront only, through L2
	const int lane = threadIdx.x & 63;
	// word w of the lane's staged read letters / database window
	uint32_t *rdw = LEAN ? &lds.seq[kLeanFrontRows * 64 + lane] : &lds.seq[lane];
	uint32_t *dbwin = rdw + Lds::kRd * 64;
	constexpr int WS = kLaneWordStride; // stride of a lane's words (row r of lane l at seq[r * 64 + l])
	const unsigned long long n_flat_raw = FLAT ? (LIST ? (unsigned long long)*reinterpret_cast<const uint32_t *>(flat_count) : *flat_count) : 0ull;
	const unsigned long long n_flat = n_flat_raw < table_cap ? n_flat_raw : table_cap;
	// (with 2 048 entries per block the few listed HSPs of a short-read batch all fell to one or two wavefronts)
	// (LIST: 64 entries, one round a side, per block while the list is short, so that its HSPs spread over the chip; 512 --
	// eight rounds ordered by the level estimate -- once there are enough of them to fill it several times over)
	const unsigned long long kItems = LIST ? (n_flat > 64ull * 4ull * gridDim.x ? 512ull : 64ull) : (unsigned long long)kBlkItems;
	const unsigned long long n_blocks = FLAT ? (n_flat + kItems - 1) / kItems : ((unsigned long long)n_reads + 63ull) / 64ull;

	// LEAN: the HSPs to hand on wait in the two spare rows in front of the letters (128 table slots; what a dead cell reads
	// there is never used) and go to the list 64-128 at a time
	uint32_t n_pend = 0;
	auto flush_pend = [&]() {
		if constexpr (LEAN) {
			for (uint32_t e0 = 0; e0 < n_pend; e0 += 64) {
				const bool mine = e0 + lane < n_pend;
				pgx_hit *hp = table + (mine ? lds.seq[e0 + lane] : 0u);
				list_append(mine, hp, big_list, big_count, big_cap);
			}
			n_pend = 0;
			lds_sync();
		}
	};
	for (unsigned long long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
		uint32_t excl = 0, st = 0, T;
		if (FLAT) {
			const unsigned long long left = n_flat - blk * kItems;
			T = (uint32_t)(left < kItems ? left : kItems);
		} else {
			const uint32_t r = (uint32_t)blk * 64u + lane;
			uint32_t cnt = 0;
			if (r < n_reads) {
				st = read_start[r];
				cnt = st == kFragmented ? 0u : read_cnt[r];
				if ((unsigned long long)st + cnt > table_cap)
					cnt = 0; // the table was too small for this read: the host repeats the step with a larger one
			}
			uint32_t incl = cnt;
#pragma unroll
			for (int dd = 1; dd < 64; dd <<= 1) {
				const uint32_t t = __shfl_up(incl, dd);
				if (lane >= dd)
					incl += t;
			}
			excl = incl - cnt;
			T = __shfl(incl, 63);
		}
		// the bucket of an HSP's side: the seed stage's level estimate (first tier: the key byte of its slot; second tier:
		// the record's own columns, whose estimates are not cut at 15, moved down so that 16 .. 31 spread over the buckets)
		auto key_of = [&](const pgx_hit *p, int side) -> uint32_t {
			if (LIST && D < kGLag)
				return (uint32_t)(side ? p->gapopen : p->mismatch) & 15u; // (the HSPs the lean tier listed: its own estimates)
			if (LIST) {
				// the record's own estimates are cut at 15, where this tier's sides begin: the levels again from the record's
				// B0 and the side's letters, floor((2 M - B0) / 5), 16 .. 31 spread over the buckets
				const int anchor = p->qend, L = p->score, b0 = side ? (p->send >> 12) & 0x7FF : (p->send >> 1) & 0x7FF;
				const int lv = (2 * (side ? L - anchor : anchor) - b0) / 5 - 16;
				return (uint32_t)(b0 == 0 || lv > 15 ? 15 : (lv < 0 ? 0 : lv));
			}
			return (uint32_t)(v.key[p - table] >> (4 * side)) & 15u;
		};
		// item -> its record
		auto locate = [&](uint32_t item) -> pgx_hit * {
			if (LIST)
				return reinterpret_cast<pgx_hit *const *>(table)[blk * kItems + item];
			if (FLAT)
				return table + blk * kItems + item;
			int o = 0;
#pragma unroll
			for (int step = 32; step >= 1; step >>= 1) {
				const int cand = o + step;
				const uint32_t e = __shfl(excl, cand & 63);
				if (cand < 64 && e <= item)
					o = cand;
			}
			const uint32_t base = __shfl(st, o), ex = __shfl(excl, o);
			return table + base + (item - ex);
		};
		for (uint32_t chunk = 0; chunk < T; chunk += kBlkItems) {
			const uint32_t n_it = T - chunk < (uint32_t)kBlkItems ? T - chunk : (uint32_t)kBlkItems;
			// The two sides of an HSP cost differently (rows ~ 1.2 x the mismatches of the diagonal on that side), so they
			// are run as separate passes, each ordered by its own side's estimate: side 0 = left of the anchor (its result is
			// parked in `side_res`), side 1 = right of it (reads the parked result and writes the finished hit).
			for (int side = 0; side < 2; side++) {
				// ---- counting sort of the chunk's HSPs by the side's mismatch estimate (16 buckets)
				if (lane < kKeyBuckets)
					lds.bucket[lane] = 0;
				lds_sync();
				for (uint32_t it = 0; it < n_it; it += 64) {
					const uint32_t item = it + lane;
					const pgx_hit *p = locate(chunk + (item < n_it ? item : n_it - 1));
					if (item < n_it)
						atomicAdd(&lds.bucket[key_of(p, side)], 1u);
				}
				lds_sync();
				{
					const uint32_t c = lane < kKeyBuckets ? lds.bucket[lane] : 0u;
					uint32_t incl = c;
#pragma unroll
					for (int dd = 1; dd < kKeyBuckets; dd <<= 1) {
						const uint32_t t = __shfl_up(incl, dd);
						if (lane >= dd)
							incl += t;
					}
					if (lane < kKeyBuckets)
						lds.bucket[lane] = incl - c;
				}
				lds_sync();
				for (uint32_t it = 0; it < n_it; it += 64) {
					const uint32_t item = it + lane;
					const pgx_hit *p = locate(chunk + (item < n_it ? item : n_it - 1));
					if (item < n_it) {
						const uint32_t slot = atomicAdd(&lds.bucket[key_of(p, side)], 1u);
						__hip_atomic_store(&order[slot], item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
				}
				lds_sync();
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the order array is in L2 before any lane reads it back
				// ---- 64 sides of similar cost per round
				for (uint32_t it = 0; it < n_it; it += 64) {
					const bool mine = it + lane < n_it;
					const uint32_t item = __hip_atomic_load(&order[mine ? it + lane : n_it - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					pgx_hit *hp = locate(chunk + item);
					const size_t slot = LIST ? (size_t)(blk * kItems + chunk + item) : (size_t)(hp - table);
					pgx_hit h;
					h.read = h.subject = h.qstart = h.qend = h.sstart = h.send = h.score = 0;
					h.mismatch = h.gapopen = 0;
					Anchor a;
					a.strand = a.L = a.qa = a.sa = a.slen = a.b0l = a.b0r = 0;
					a.S0 = 0;
					a.gpos = 0;
					a.s.rw = a.s.ra = a.s.dbw = a.s.dba = nullptr;
					bool on = false;
					int awin = 0, sq0 = 0, sd0 = 0; // the anchor's position in the window; where the side starts in the staged rows
					uint2 parked = make_uint2(0u, 0u);
					if (mine) {
						h = *hp;
						a = anchor_of(v, h);
						// the database window of this HSP: from D + 16 bases left of where the read's first base
						// would lie, in whole 16-base words
						const int64_t lo = ((int64_t)a.gpos - a.qa - D - 16) >> 4; // word index (may be negative: front padding)
						bool wide = a.L > MAXL;
						if (side == 1) {
							parked = side_res[slot];
							wide = (parked.y >> 31) != 0u; // the left pass sent this HSP to the wide kernel
						} else {
							if (!wide && a.s.ra) {
								uint64_t any = 0;
								for (int w = 0; w < (a.L + 31) / 32; w++)
									any |= a.s.ra[w];
								wide = any != 0;
							}
							if (!wide && a.s.dba) {
								const int64_t b0 = (lo * 16) >> kBlkShift, b1 = (lo * 16 + Lds::kDb * 16) >> kBlkShift;
								if (v.amb_blk) {
									for (int64_t bb = b0 < 0 ? 0 : b0; bb <= b1; bb++)
										wide = wide || ((v.amb_blk[bb >> 5] >> (bb & 31)) & 1u);
								} else {
									wide = true;
								}
							}
						}
						if (!wide) {
							const uint32_t *gr = reinterpret_cast<const uint32_t *>(a.s.rw);
							const uint32_t *gd = reinterpret_cast<const uint32_t *>(a.s.dbw) + lo;
							awin = (int)((int64_t)a.gpos - lo * 16); // the anchor's position in the window (>= qa + 34)
							if (side) {
								// right of the anchor: the read from the anchor's word on, the window likewise
								for (int w = a.qa >> 4; w < Lds::kRd; w++)
									rdw[w * WS] = gr[w];
								for (int w = awin >> 4; w < Lds::kDb; w++)
									dbwin[w * WS] = gd[w];
								sq0 = a.qa;
								sd0 = awin;
							} else {
								// left of the anchor, REVERSED: whole words in reverse order, each with its bits reversed (the two
								// bits of a letter swap in both sequences alike, and letters are only ever compared); letter y of W
								// words lands at 16 W - 1 - y, so the side starts at 16 W - (anchor position)
								const int wr = (a.qa + 15) >> 4, wd = (awin + 15) >> 4;
								for (int w = 0; w < wr; w++)
									rdw[w * WS] = __builtin_bitreverse32(gr[wr - 1 - w]);
								rdw[wr * WS] = 0u; // (the window of the last letters reads one word further)
								for (int w = 0; w < wd; w++)
									dbwin[w * WS] = __builtin_bitreverse32(gd[wd - 1 - w]);
								dbwin[wd * WS] = 0u;
								sq0 = 16 * wr - a.qa;
								sd0 = 16 * wd - awin;
							}
							on = true;
						}
					}
					lds_sync();
					Side sd;
					bool ok;
					if constexpr (LEAN) {
						// (a side whose best cell holds two or more gap columns goes on: the next tier carries the statistics)
						ok = greedy_rows_lean<D>((const lds_word *)&lds.seq[0], (uint32_t)lane * 4u, on, 32 * kLeanFrontRows + 2 * sq0, 32 * (kLeanFrontRows + Lds::kRd) + 2 * sd0,
									  side ? a.L - a.qa : a.qa, side ? a.slen - a.sa : a.sa, side ? a.b0r : a.b0l, sd) == 0;
					} else {
						ok = greedy_rows<D>(rdw, dbwin, D >= kGLag ? &lds.xring[D >= kGLag ? lane : 0][0] : nullptr, on, sq0, sd0, side ? a.L - a.qa : a.qa, side ? a.slen - a.sa : a.sa, side ? a.b0r : a.b0l, sd);
					}
					if (mine) {
						if (side == 0) {
							// parked: i | j << 10 | mismatches << 20 | gap openings << 26 ; gap columns | wide << 31
							const uint32_t gaps = (uint32_t)((sd.i + sd.j - sd.s2) / 6 - sd.mism);
							side_res[slot] = (on && ok) ? make_uint2((uint32_t)sd.i | ((uint32_t)sd.j << 10) | ((uint32_t)sd.mism << 20) | ((uint32_t)sd.gopen << 26), gaps)
										    : make_uint2(0u, 1u << 31);
						} else if (on && ok) {
							Side l;
							l.i = (int)(parked.x & 1023u);
							l.j = (int)((parked.x >> 10) & 1023u);
							l.mism = (int)((parked.x >> 20) & 63u);
							l.gopen = (int)(parked.x >> 26);
							l.s2 = l.i + l.j - 6 * (l.mism + (int)parked.y);
							write_gapped(hp, h, a, l, sd);
						}
					}
					if (side == 1) {
						const bool fail = mine && !(on && ok);
						if constexpr (LEAN) {
							const unsigned long long fm = __ballot(fail);
							if (fm != 0ull) {
								lds_sync(); // (the rows' last reads of the front rows are done)
								if (fail)
									lds.seq[n_pend + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = (uint32_t)(hp - table);
								n_pend += (uint32_t)__popcll(fm);
								lds_sync();
								if (n_pend > 64u)
									flush_pend();
							}
						} else {
							list_append(fail, hp, big_list, big_count, big_cap);
						}
					}
					lds_sync();
				}
			}
		}
		flush_pend();
	}
}

// ------------------------------------------------------------------------------------------ the main table, binned
// Round 3, second finding.  With the lean rows the stage did NOT get faster: PMC (profiles/r03_*) shows what holds it --
// 209 L2 misses per read (7.4 per HSP) at 39 G misses a second, 0.78 of what the chip delivers for random 64-byte lines
// (pgx_probe_gather: 50 G/s).  The pool-per-wavefront form above touches every record three times (ordering, left pass,
// right pass), parks the left result, and fetches an HSP's letters twice.  This form touches them ONCE:
//   1. k_gap_hist / k_gap_bins / k_gap_scatter: a counting sort of the main table's SLOT NUMBERS by the seed stage's key
//      byte (levels left | levels right << 4; 0xFF = no record there), streaming over the 1-byte keys only: 256 bins, the
//      costliest first.  Every HSP of a bin runs the same number of levels on BOTH sides.
//   2. k_gapped_rows: a wavefront takes 64 consecutive entries of the binned list, fetches each record once, stages the
//      letters of both sides once (right part forward, left part reversed, each lane's areas packed behind one another
//      in its column of the transposed rows), runs the lean rows left then right -- the lanes of a round finish both
//      sides together because the bin says so -- and writes the finished hit over the record.  No ordering pass per
//      wavefront, no cost-order scratch, no parked sides.
//   3. (the step that paid most) the list is ordered by database REGION first (at most 240 regions of ~4 Mbases, a byte the
//      seed stage writes beside the key), by key byte within a region: the wavefronts in flight work on one region at a
//      time, ~1 MB of database words that stay in L2 -- every database line is fetched from memory once per step and
//      XCD instead of once per HSP and side (2 of the 5.6 misses per HSP).
// The overflow table (a handful of reads per batch) and the lists keep the pool form.
constexpr int kGapBins = 256;
// words of the sorting passes' table: [r] entries of region r, [kBaseAt + r] where they start, [kCurAt + r] scatter cursor,
// [kTotalAt] entries in all, [kTileAt + r] tiles before region r (one more: all tiles), [kSegAt + r * 256 + b] entries
// (then cursor) of key byte b within region r
constexpr int kBaseAt = 256, kCurAt = 512, kTotalAt = 768, kTileAt = 1024, kSegAt = 2048, kBinsWords = kSegAt + 256 * kGapBins + 16;
constexpr int kSegTile = 8192; // entries a block sorts at a time in the second pass (its LDS copy: 40 KB)

// valid slots per database region (key byte 0xFF: no record in the slot)
__global__ __launch_bounds__(256) void k_reg_hist(const uint8_t *__restrict__ key, const uint8_t *__restrict__ reg, const unsigned long long *__restrict__ used,
						  unsigned long long cap, uint32_t *__restrict__ bins)
{
	__shared__ uint32_t h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	const unsigned long long n = *used < cap ? *used : cap;
	for (unsigned long long base = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 16; base < n; base += (unsigned long long)gridDim.x * 256 * 16) {
		const uint4 kk = *reinterpret_cast<const uint4 *>(key + base), rr = *reinterpret_cast<const uint4 *>(reg + base); // (padded to 16-byte words)
		const uint32_t kw[4] = { kk.x, kk.y, kk.z, kk.w }, rw[4] = { rr.x, rr.y, rr.z, rr.w };
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const uint32_t b = (kw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
			if (b != 0xFFu && base + j < n)
				atomicAdd(&h[(rw[j >> 2] >> (8 * (j & 3))) & 0xFFu], 1u);
		}
	}
	__syncthreads();
	if (h[threadIdx.x])
		atomicAdd(&bins[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_reg_bases(uint32_t *__restrict__ bins)
{
	if (threadIdx.x == 0) {
		uint32_t run = 0, tiles = 0;
		for (int r = 0; r < 256; r++) {
			bins[kBaseAt + r] = run;
			bins[kCurAt + r] = run;
			bins[kTileAt + r] = tiles;
			run += bins[r];
			tiles += (bins[r] + kSegTile - 1) / kSegTile;
		}
		bins[kTotalAt] = run;
		bins[kTileAt + 256] = tiles;
	}
}

// exclusive prefix sums of the 256 counts of a block's histogram (256 threads: thread t owns h[t]); wsum: 4 words of LDS
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *wsum)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t inc = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(inc, o);
		if (lane >= o)
			inc += t;
	}
	if (lane == 63)
		wsum[wave] = inc;
	__syncthreads();
	uint32_t off = 0;
	for (int w = 0; w < wave; w++)
		off += wsum[w];
	return off + inc - v;
}

// first pass: slot numbers (and their key bytes) by region.  A tile of 8 192 slots is ordered by region IN LDS first (the
// rank inside (tile, region) is what the counting atomic returns) and written out run by run: a region's ~34 entries of a
// tile go out as 136 + 34 contiguous bytes.  Scattered straight from the registers every entry was two L2 requests of its
// own -- 31 requests per read, 0.77 of what the L2s take per second: the pass was bound by its request count, 1.49 ms for
// 2.4 GB (profiles/r03_c_pmc.txt).
constexpr int kRegTile = 8192;
__global__ __launch_bounds__(256) void k_reg_scatter(const uint8_t *__restrict__ key, const uint8_t *__restrict__ reg, const unsigned long long *__restrict__ used,
						     unsigned long long cap, uint32_t *__restrict__ bins, uint32_t *__restrict__ items1, uint8_t *__restrict__ keys1)
{
	__shared__ uint32_t h[256], loc[256], at[256], wsum[4];
	__shared__ uint32_t s_item[kRegTile];
	__shared__ uint8_t s_key[kRegTile], s_reg[kRegTile];
	const unsigned long long n = *used < cap ? *used : cap;
	constexpr int kQ = kRegTile / (256 * 16); // 16-byte words of keys per thread
	for (unsigned long long t0 = (unsigned long long)blockIdx.x * kRegTile; t0 < n; t0 += (unsigned long long)gridDim.x * kRegTile) {
		h[threadIdx.x] = 0;
		__syncthreads();
		uint4 kk[kQ], rr[kQ];
		uint32_t rk[kQ][8]; // ranks inside (tile, region), two to a word (< 8 192)
#pragma unroll
		for (int q = 0; q < kQ; q++) {
			const unsigned long long base = t0 + ((unsigned long long)q * 256 + threadIdx.x) * 16;
			kk[q] = base < n ? *reinterpret_cast<const uint4 *>(key + base) : make_uint4(~0u, ~0u, ~0u, ~0u);
			rr[q] = base < n ? *reinterpret_cast<const uint4 *>(reg + base) : make_uint4(0u, 0u, 0u, 0u);
			const uint32_t kw[4] = { kk[q].x, kk[q].y, kk[q].z, kk[q].w }, rw[4] = { rr[q].x, rr[q].y, rr[q].z, rr[q].w };
#pragma unroll
			for (int j = 0; j < 16; j++) {
				const uint32_t b = (kw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
				uint32_t rank = 0;
				if (b != 0xFFu && base + j < n)
					rank = atomicAdd(&h[(rw[j >> 2] >> (8 * (j & 3))) & 0xFFu], 1u);
				if (j & 1)
					rk[q][j >> 1] |= rank << 16;
				else
					rk[q][j >> 1] = rank;
			}
		}
		__syncthreads();
		const uint32_t mine = h[threadIdx.x];
		loc[threadIdx.x] = block_excl_scan_256(mine, wsum);
		at[threadIdx.x] = mine ? atomicAdd(&bins[kCurAt + threadIdx.x], mine) : 0u;
		__syncthreads();
#pragma unroll
		for (int q = 0; q < kQ; q++) {
			const unsigned long long base = t0 + ((unsigned long long)q * 256 + threadIdx.x) * 16;
			const uint32_t kw[4] = { kk[q].x, kk[q].y, kk[q].z, kk[q].w }, rw[4] = { rr[q].x, rr[q].y, rr[q].z, rr[q].w };
#pragma unroll
			for (int j = 0; j < 16; j++) {
				const uint32_t b = (kw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
				if (b != 0xFFu && base + j < n) {
					const uint32_t r = (rw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
					const uint32_t p = loc[r] + ((rk[q][j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
					s_item[p] = (uint32_t)(base + j);
					s_key[p] = (uint8_t)b;
					s_reg[p] = (uint8_t)r;
				}
			}
		}
		__syncthreads();
		const uint32_t total = loc[255] + h[255];
		for (uint32_t p = threadIdx.x; p < total; p += 256) {
			const uint32_t r = s_reg[p], d = at[r] + (p - loc[r]);
			items1[d] = s_item[p];
			keys1[d] = s_key[p];
		}
		__syncthreads();
	}
}

// a tile of the second pass: up to kSegTile consecutive entries of ONE region
__device__ __forceinline__ bool seg_tile(const uint32_t *__restrict__ bins, uint32_t tile, uint32_t &r, uint32_t &lo, uint32_t &hi)
{
	if (tile >= bins[kTileAt + 256])
		return false;
	uint32_t a = 0, b = 256; // the last region whose first tile is <= tile
	while (b - a > 1) {
		const uint32_t m = (a + b) / 2;
		if (bins[kTileAt + m] <= tile)
			a = m;
		else
			b = m;
	}
	r = a;
	lo = bins[kBaseAt + r] + (tile - bins[kTileAt + r]) * (uint32_t)kSegTile;
	const uint32_t end = bins[kBaseAt + r] + bins[r];
	hi = lo + (uint32_t)kSegTile < end ? lo + (uint32_t)kSegTile : end;
	return true;
}

__global__ __launch_bounds__(256) void k_seg_hist(const uint8_t *__restrict__ keys1, uint32_t *__restrict__ bins)
{
	__shared__ uint32_t h[256];
	for (uint32_t tile = blockIdx.x;; tile += gridDim.x) {
		uint32_t r, lo, hi;
		if (!seg_tile(bins, tile, r, lo, hi))
			break;
		h[threadIdx.x] = 0;
		__syncthreads();
		// four entries per thread and step (a 4-byte word of keys; the list is padded to whole words)
		for (uint32_t i0 = (lo & ~3u) + 4u * threadIdx.x; i0 < hi; i0 += 1024u) {
			const uint32_t kw = *reinterpret_cast<const uint32_t *>(keys1 + i0);
#pragma unroll
			for (int j = 0; j < 4; j++)
				if (i0 + j >= lo && i0 + j < hi)
					atomicAdd(&h[(kw >> (8 * j)) & 0xFFu], 1u);
		}
		__syncthreads();
		if (h[threadIdx.x])
			atomicAdd(&bins[kSegAt + r * 256 + threadIdx.x], h[threadIdx.x]);
		__syncthreads();
	}
}

// one block per region: where each key byte's entries start, the costliest (levels left + levels right) first
__global__ __launch_bounds__(256) void k_seg_bases(uint32_t *__restrict__ bins)
{
	__shared__ uint32_t cnt[kGapBins], at_rank[kGapBins];
	const int b = threadIdx.x, r = blockIdx.x;
	cnt[b] = bins[kSegAt + r * 256 + b];
	const int cost = (b & 15) + (b >> 4);
	int rank = 0;
	for (int o = 0; o < kGapBins; o++) {
		const int co = (o & 15) + (o >> 4);
		rank += (co > cost) || (co == cost && o < b);
	}
	at_rank[rank] = (uint32_t)b;
	__syncthreads();
	if (b == 0) {
		uint32_t run = bins[kBaseAt + r];
		for (int q = 0; q < kGapBins; q++) {
			const uint32_t bb = at_rank[q];
			bins[kSegAt + r * 256 + bb] = run;
			run += cnt[bb];
		}
	}
}

__global__ __launch_bounds__(256) void k_seg_scatter(const uint8_t *__restrict__ keys1, const uint32_t *__restrict__ items1, uint32_t *__restrict__ bins,
						     uint32_t *__restrict__ items2)
{
	// (as k_reg_scatter: the tile ordered by key byte in LDS, written out run by run)
	__shared__ uint32_t h[256], loc[256], at[256], wsum[4];
	__shared__ uint32_t s_item[kSegTile];
	__shared__ uint8_t s_b[kSegTile];
	for (uint32_t tile = blockIdx.x;; tile += gridDim.x) {
		uint32_t r, lo, hi;
		if (!seg_tile(bins, tile, r, lo, hi))
			break;
		h[threadIdx.x] = 0;
		__syncthreads();
		constexpr int kSteps = kSegTile / 1024 + 1; // (a tile that starts inside a word reaches one step further)
		uint32_t kw[kSteps], rk[kSteps][2];
#pragma unroll
		for (int q = 0; q < kSteps; q++) {
			const uint32_t i0 = (lo & ~3u) + 4u * threadIdx.x + 1024u * q;
			kw[q] = i0 < hi ? *reinterpret_cast<const uint32_t *>(keys1 + i0) : 0u;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				uint32_t rank = 0;
				if (i0 + j >= lo && i0 + j < hi)
					rank = atomicAdd(&h[(kw[q] >> (8 * j)) & 0xFFu], 1u);
				if (j & 1)
					rk[q][j >> 1] |= rank << 16;
				else
					rk[q][j >> 1] = rank;
			}
		}
		__syncthreads();
		const uint32_t mine = h[threadIdx.x];
		loc[threadIdx.x] = block_excl_scan_256(mine, wsum);
		at[threadIdx.x] = mine ? atomicAdd(&bins[kSegAt + r * 256 + threadIdx.x], mine) : 0u;
		__syncthreads();
#pragma unroll
		for (int q = 0; q < kSteps; q++) {
			const uint32_t i0 = (lo & ~3u) + 4u * threadIdx.x + 1024u * q;
			if (i0 < hi) {
				const uint4 it = *reinterpret_cast<const uint4 *>(items1 + i0);
				const uint32_t iv[4] = { it.x, it.y, it.z, it.w };
#pragma unroll
				for (int j = 0; j < 4; j++)
					if (i0 + j >= lo && i0 + j < hi) {
						const uint32_t b = (kw[q] >> (8 * j)) & 0xFFu;
						const uint32_t p = loc[b] + ((rk[q][j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
						s_item[p] = iv[j];
						s_b[p] = (uint8_t)b;
					}
			}
		}
		__syncthreads();
		const uint32_t total = hi - lo;
		for (uint32_t p = threadIdx.x; p < total; p += 256) {
			const uint32_t b = s_b[p];
			items2[at[b] + (p - loc[b])] = s_item[p];
		}
		__syncthreads();
	}
}

template <int MAXL, int DD = kGFastD> struct RowsLds {
	static constexpr int D = DD;
	static constexpr int kQ = MAXL / 16 + 4;                     // rows of a lane's read letters: right part, then left part reversed
	static constexpr int kD = (MAXL + 2 * D + 32 + 60) / 16 + 3; // rows of its database letters, likewise
	static constexpr int kRows = kLeanFrontRows + kQ + kD + 1;   // (one spare row behind: the last word's neighbour)
	uint32_t seq[kRows * 64];
};

// Where the lean tier hands on what it cannot finish: list A = HSPs whose best cell holds two gap columns of a kind (the
// rows with the full statistics over 18 differences decide them), list B = HSPs alive after 18 differences, HSPs the seed
// stage's own estimate puts beyond 18 levels, HSPs that touch an ambiguity letter (the 40-difference tier and the wide
// kernels: sending them through list A's tier first ran reads of 400-500 bases 10-15 % slower than round 2).
struct TierLists {
	unsigned long long *a, *b;
	uint32_t *a_count, *b_count;
	uint32_t cap;
	uint32_t *b_first; // counts the first tier's own appends to list B (the tier behind list A appends there too), or null
};
struct Pending {
	uint32_t na = 0, nb = 0; // table slots waiting in front row 0 (for list A) and front row 1 (for list B), <= 64 each
};

// One round of 64 HSPs, one per lane, both sides: the record once, the letters once (right part forward, left part
// reversed, packed behind one another in the lane's column of the transposed rows), the lean rows left then right, the
// finished hit over the record.  HSPs this tier cannot finish wait in the two front rows (`n_pend` table slots).
template <int MAXL, int D = kGFastD>
__device__ __forceinline__ void gap_round(RowsLds<MAXL, D> &lds, const GapView &v, pgx_hit *__restrict__ table, pgx_hit *hp, bool mine, Pending &pend,
					  const TierLists &tl)
{
	using Lds = RowsLds<MAXL, D>;
	const int lane = threadIdx.x & 63;
	const lds_word *seq0 = (const lds_word *)&lds.seq[0];
	uint32_t *col = &lds.seq[lane]; // row r of this lane: col[r * 64]
	const pgx_hit h = *hp;
	const Anchor a = anchor_of(v, h);
	bool on = mine && a.L <= MAXL;
	// (the seed stage's own estimate of the levels a side will run, floor((2 letters - B0) / 5): a side it puts two levels
	// beyond this tier's D would run all D levels here only to be handed on)
	// (not for the 40-difference rows: the seed stage follows the diagonal for 18 mismatches only, its estimate says
	// nothing about sides beyond that -- half of a 500-base read's HSPs were handed on unseen)
	if constexpr (D < kGLag)
		on = on && (2 * a.qa - a.b0l) / 5 <= D + 1 && (2 * (a.L - a.qa) - a.b0r) / 5 <= D + 1;
	if (on && a.s.ra) {
		uint64_t any = 0;
		for (int w = 0; w < (a.L + 31) / 32; w++)
			any |= a.s.ra[w];
		on = any == 0;
	}
	// the database window: from D + 16 bases left of where the read's first base would lie, in whole 16-base words
	const int64_t lo = ((int64_t)a.gpos - a.qa - D - 16) >> 4; // (may be negative: front padding)
	const int awin = (int)((int64_t)a.gpos - lo * 16);        // the anchor's position in it
	if (on && a.s.dba) {
		const int64_t b0 = (lo * 16) >> kBlkShift, b1 = (lo * 16 + (a.L + 2 * D + 64)) >> kBlkShift;
		if (v.amb_blk) {
			for (int64_t bb = b0 < 0 ? 0 : b0; bb <= b1; bb++)
				on = on && !((v.amb_blk[bb >> 5] >> (bb & 31)) & 1u);
		} else {
			on = false;
		}
	}
	int QB[2] = { 0, 0 }, DB0[2] = { 0, 0 };
	{
		// Letters come in 16-byte loads, the same words for every lane (so they stay in registers), and go to rows that
		// differ per lane: right of the anchor forward from the anchor's word, left of it REVERSED (words in reverse order,
		// bits reversed: the two bits of a letter swap in both sequences alike).  Word by word, the loads of round 3's
		// first form asked L2 for every line eight times over (20 requests per HSP, 13 of them hits).
		constexpr int NQ = (MAXL / 16 + 2 + 3) / 4, ND = ((MAXL + 2 * D + 62) / 16 + 2 + 3) / 4;
		const uint32_t *gr = reinterpret_cast<const uint32_t *>(a.s.rw);
		const uint32_t *gd = reinterpret_cast<const uint32_t *>(a.s.dbw) + lo;
		uint4 q4[NQ], d4[ND];
		if (on) {
#pragma unroll
			for (int t = 0; t < NQ; t++)
				q4[t] = *reinterpret_cast<const uint4 *>(gr + 4 * t);
#pragma unroll
			for (int t = 0; t < ND; t++)
				d4[t] = *reinterpret_cast<const uint4 *>(gd + 4 * t);
			const int wq0 = a.qa >> 4, wq1 = (a.L + 15) >> 4, wr = (a.qa + 15) >> 4;
			const int rowR = kLeanFrontRows, rowL = rowR + (wq1 - wq0 + 1);
			QB[1] = 32 * rowR + 2 * (a.qa & 15);
			QB[0] = 32 * rowL + 2 * (16 * wr - a.qa);
#pragma unroll
			for (int w = 0; w < 4 * NQ; w++) {
				const uint32_t val = w & 2 ? (w & 1 ? q4[w >> 2].w : q4[w >> 2].z) : (w & 1 ? q4[w >> 2].y : q4[w >> 2].x);
				if (w >= wq0 && w <= wq1)
					col[(rowR + w - wq0) * 64] = val;
				if (w < wr)
					col[(rowL + wr - 1 - w) * 64] = __builtin_bitreverse32(val);
			}
			col[(rowL + wr) * 64] = 0u; // (the window of the last letters reads one word further)
			const int wd0 = awin >> 4, wd1 = ((awin + (a.L - a.qa) + D + 16) >> 4) + 1, wd = (awin + 15) >> 4;
			const int rowDR = kLeanFrontRows + Lds::kQ, rowDL = rowDR + (wd1 - wd0 + 1);
			DB0[1] = 32 * rowDR + 2 * (awin & 15);
			DB0[0] = 32 * rowDL + 2 * (16 * wd - awin);
#pragma unroll
			for (int w = 0; w < 4 * ND; w++) {
				const uint32_t val = w & 2 ? (w & 1 ? d4[w >> 2].w : d4[w >> 2].z) : (w & 1 ? d4[w >> 2].y : d4[w >> 2].x);
				if (w >= wd0 && w <= wd1)
					col[(rowDR + w - wd0) * 64] = val;
				if (w < wd)
					col[(rowDL + wd - 1 - w) * 64] = __builtin_bitreverse32(val);
			}
			col[(rowDL + wd) * 64] = 0u;
		}
	}
	lds_sync();
	// What has to outlive the rows is kept SMALL and packed: in this kernel a spilled register is memory traffic (136 bytes
	// of scratch per lane were ~1.3 of the 4.6 L2 misses per HSP).  Slot number, anchor | length, the record's `send` word
	// (strand | B0 left | B0 right), the anchor's place in the subject and the subject's length, the two sides' bit offsets,
	// the left side's result in two words.  `read` and `subject` of the record stay where they are: the hit is written
	// over the other 24 bytes.
	const uint32_t slot = (uint32_t)(hp - table);
	const uint32_t geo = (uint32_t)a.qa | ((uint32_t)a.L << 10), send_w = (uint32_t)h.send;
	const int sa = a.sa, slen = a.slen;
	const uint32_t off_l = (uint32_t)QB[0] | ((uint32_t)DB0[0] << 16), off_r = (uint32_t)QB[1] | ((uint32_t)DB0[1] << 16);
	uint32_t left_pk = 0;
	int left_s2 = 0;
	Side rr;
	rr.i = rr.j = rr.s2 = rr.mism = rr.gopen = 0;
	int worst = 0; // 0 both sides done, 2 a side needs the full statistics, 1 a side is alive after D differences
#pragma unroll 1
	for (int side = 0; side < 2; side++) {
		const int qa_ = (int)(geo & 1023u), L_ = (int)(geo >> 10);
		const uint32_t off = side ? off_r : off_l;
		Side r;
		int st;
		if constexpr (D >= kGLag)
			st = greedy_rows_deep<D>(seq0, (uint32_t)lane * 4u, on, (int)(off & 0xFFFFu), (int)(off >> 16), side ? L_ - qa_ : qa_,
						 side ? slen - sa : sa, (int)(side ? (send_w >> 12) & 0x7FFu : (send_w >> 1) & 0x7FFu), r);
		else
			st = greedy_rows_lean<D>(seq0, (uint32_t)lane * 4u, on, (int)(off & 0xFFFFu), (int)(off >> 16), side ? L_ - qa_ : qa_,
						 side ? slen - sa : sa, (int)(side ? (send_w >> 12) & 0x7FFu : (send_w >> 1) & 0x7FFu), r);
		worst = st == 1 || worst == 1 ? 1 : (st == 2 || worst == 2 ? 2 : 0);
		if (side) {
			rr = r;
		} else {
			left_pk = (uint32_t)r.i | ((uint32_t)r.j << 10) | ((uint32_t)r.mism << 20) | ((uint32_t)r.gopen << 26);
			left_s2 = r.s2;
		}
	}
	if (on && worst == 0) {
		const int qa_ = (int)(geo & 1023u), L_ = (int)(geo >> 10);
		const int li = (int)(left_pk & 1023u), lj = (int)((left_pk >> 10) & 1023u);
		const int bl = qa_ - li, br = qa_ + rr.i - 1, sl = sa - lj, sr = sa + rr.j - 1;
		int4 c; // qstart, qend, sstart, send
		if (!(send_w & 1u))
			c = make_int4(bl + 1, br + 1, sl + 1, sr + 1);
		else
			c = make_int4(L_ - br, L_ - bl, sr + 1, sl + 1);
		pgx_hit *dst = table + slot;
		*reinterpret_cast<int4 *>(&dst->qstart) = c;
		const uint32_t mg = ((left_pk >> 20) & 63u) + (uint32_t)rr.mism + ((((left_pk >> 26) & 63u) + (uint32_t)rr.gopen) << 16);
		*reinterpret_cast<int2 *>(&dst->score) = make_int2((left_s2 + rr.s2) >> 1, (int)mg);
	}
	const bool fail_a = mine && on && worst == 2, fail_b = mine && (!on || worst == 1);
	lds_sync(); // (the rows' reads are done before the letters are replaced)
	auto park = [&](bool fail, uint32_t &n, int row, unsigned long long *list, uint32_t *count, uint32_t *also) {
		const unsigned long long fm = __ballot(fail);
		if (fm == 0ull)
			return;
		const uint32_t cnt = (uint32_t)__popcll(fm);
		if (n + cnt > 64u) { // the row is full: its slots go to the list, one atomic for all of them
			list_append(lane < (int)n, table + (lane < (int)n ? lds.seq[row * 64 + lane] : 0u), list, count, tl.cap, also);
			n = 0;
			lds_sync();
		}
		if (fail)
			lds.seq[row * 64 + n + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = slot;
		n += cnt;
		lds_sync();
	};
	park(fail_a, pend.na, 0, tl.a, tl.a_count, nullptr);
	park(fail_b, pend.nb, 1, tl.b, tl.b_count, tl.b_first);
}

template <int MAXL, int D = kGFastD>
__device__ __forceinline__ void gap_flush(RowsLds<MAXL, D> &lds, pgx_hit *__restrict__ table, Pending &pend, const TierLists &tl)
{
	const int lane = threadIdx.x & 63;
	list_append(lane < (int)pend.na, table + (lane < (int)pend.na ? lds.seq[lane] : 0u), tl.a, tl.a_count, tl.cap);
	list_append(lane < (int)pend.nb, table + (lane < (int)pend.nb ? lds.seq[64 + lane] : 0u), tl.b, tl.b_count, tl.cap, tl.b_first);
	pend.na = pend.nb = 0;
	lds_sync();
}

// the main table: rounds of 64 consecutive entries of the sorted slot list
template <int MAXL, int WAVES, int D = kGFastD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES))) void k_gapped_rows(GapView v, pgx_hit *__restrict__ table, const uint32_t *__restrict__ items,
													  uint32_t *__restrict__ bins, TierLists tl)
{
	__shared__ RowsLds<MAXL, D> lds;
	const int lane = threadIdx.x & 63;
	const uint32_t n_items = bins[kTotalAt];
	uint32_t *next_round = bins + kTotalAt + 1; // (zero at launch)
	const uint32_t n_rounds = (n_items + 63u) / 64u;
	Pending pend;
	// rounds are handed out in order, kRoundGrab at a time, from one counter: the wavefronts in flight then work on
	// neighbouring rounds -- one or two database regions -- however unevenly they progress (with a fixed stride per
	// wavefront they drifted tens of regions apart and the region order bought nothing)
	constexpr uint32_t kRoundGrab = 8;
	for (;;) {
		uint32_t first = 0;
		if (lane == 0)
			first = atomicAdd(next_round, kRoundGrab);
		first = __shfl(first, 0);
		if (first >= n_rounds)
			break;
		const uint32_t last = first + kRoundGrab < n_rounds ? first + kRoundGrab : n_rounds;
		for (uint32_t round = first; round < last; round++) {
			const uint32_t idx = round * 64u + lane;
			const bool mine = idx < n_items;
			gap_round<MAXL, D>(lds, v, table, table + items[mine ? idx : n_items - 1u], mine, pend, tl);
		}
	}
	gap_flush<MAXL, D>(lds, table, pend, tl);
}

// The pool form, ONE pass (round 3): a wavefront takes the HSPs of 64 reads (FLAT: 2 048 entries of the overflow table),
// orders them by the seed stage's two estimates -- levels left of the anchor, then levels right of it: a counting sort
// over the 1-byte keys, 256 buckets that borrow the letter rows before the first round needs them -- and runs them 64 at
// a time with gap_round.  What the binned form gave up is kept: the pool's records are neighbours (two to a line, the
// second one an L2 hit) and its 64 reads' letters (5 KB) are fetched from memory once for their ~1 800 HSPs; what the
// two-pass pools of round 2 paid is gone: the second fetch of every letter and record, the parked left sides.
// 7.4 L2 misses per HSP there (and 7.2 binned: every record and read line a first touch) -> see profiles/r03_*.
template <bool FLAT, int MAXL, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES))) void k_gapped_pool(GapView v, pgx_hit *__restrict__ table, unsigned long long table_cap,
													  const uint32_t *__restrict__ read_start, const uint32_t *__restrict__ read_cnt,
													  uint32_t n_reads, const unsigned long long *__restrict__ flat_count,
													  TierLists tl, uint32_t *__restrict__ order_all)
{
	__shared__ RowsLds<MAXL> lds;
	static_assert(RowsLds<MAXL>::kRows * 64 >= 128 + kGapBins, "the buckets borrow the letter rows");
	uint32_t *bucket = &lds.seq[128]; // (behind the two front rows, which hold the pending list)
	uint32_t *order = order_all + (size_t)blockIdx.x * kBlkItems; // written and read by this wavefront only, through L2
	const int lane = threadIdx.x & 63;
	const unsigned long long n_flat_raw = FLAT ? *flat_count : 0ull;
	const unsigned long long n_flat = n_flat_raw < table_cap ? n_flat_raw : table_cap;
	const unsigned long long n_blocks = FLAT ? (n_flat + kBlkItems - 1) / kBlkItems : ((unsigned long long)n_reads + 63ull) / 64ull;
	Pending pend;
	for (unsigned long long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
		uint32_t excl = 0, st = 0, T;
		if (FLAT) {
			const unsigned long long left = n_flat - blk * kBlkItems;
			T = (uint32_t)(left < (unsigned long long)kBlkItems ? left : (unsigned long long)kBlkItems);
		} else {
			const uint32_t r = (uint32_t)blk * 64u + lane;
			uint32_t cnt = 0;
			if (r < n_reads) {
				st = read_start[r];
				cnt = st == kFragmented ? 0u : read_cnt[r];
				if ((unsigned long long)st + cnt > table_cap)
					cnt = 0; // the table was too small for this read: the host repeats the step with a larger one
			}
			uint32_t incl = cnt;
#pragma unroll
			for (int dd = 1; dd < 64; dd <<= 1) {
				const uint32_t t = __shfl_up(incl, dd);
				if (lane >= dd)
					incl += t;
			}
			excl = incl - cnt;
			T = __shfl(incl, 63);
		}
		auto locate = [&](uint32_t item) -> pgx_hit * {
			if (FLAT)
				return table + blk * kBlkItems + item;
			int o = 0;
#pragma unroll
			for (int step = 32; step >= 1; step >>= 1) {
				const int cand = o + step;
				const uint32_t e = __shfl(excl, cand & 63);
				if (cand < 64 && e <= item)
					o = cand;
			}
			const uint32_t base = __shfl(st, o), ex = __shfl(excl, o);
			return table + base + (item - ex);
		};
		// bucket of an HSP: levels left of the anchor first, then levels right of it (the key byte with its halves swapped)
		auto bucket_of = [&](const pgx_hit *p) -> uint32_t {
			const uint32_t k = v.key[p - table];
			return ((k & 15u) << 4) | (k >> 4);
		};
		for (uint32_t chunk = 0; chunk < T; chunk += kBlkItems) {
			const uint32_t n_it = T - chunk < (uint32_t)kBlkItems ? T - chunk : (uint32_t)kBlkItems;
			for (int b = lane; b < kGapBins; b += 64)
				bucket[b] = 0;
			lds_sync();
			for (uint32_t it = 0; it < n_it; it += 64) {
				const uint32_t item = it + lane;
				const pgx_hit *p = locate(chunk + (item < n_it ? item : n_it - 1));
				if (item < n_it)
					atomicAdd(&bucket[bucket_of(p)], 1u);
			}
			lds_sync();
			{
				// exclusive scan of the 256 counts: four consecutive buckets per lane
				const uint32_t c0 = bucket[4 * lane], c1 = bucket[4 * lane + 1], c2 = bucket[4 * lane + 2], c3 = bucket[4 * lane + 3];
				const uint32_t sum = c0 + c1 + c2 + c3;
				uint32_t incl = sum;
#pragma unroll
				for (int dd = 1; dd < 64; dd <<= 1) {
					const uint32_t t = __shfl_up(incl, dd);
					if (lane >= dd)
						incl += t;
				}
				const uint32_t e = incl - sum;
				lds_sync();
				bucket[4 * lane] = e;
				bucket[4 * lane + 1] = e + c0;
				bucket[4 * lane + 2] = e + c0 + c1;
				bucket[4 * lane + 3] = e + c0 + c1 + c2;
			}
			lds_sync();
			for (uint32_t it = 0; it < n_it; it += 64) {
				const uint32_t item = it + lane;
				const pgx_hit *p = locate(chunk + (item < n_it ? item : n_it - 1));
				if (item < n_it) {
					const uint32_t slot = atomicAdd(&bucket[bucket_of(p)], 1u);
					__hip_atomic_store(&order[slot], item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
			lds_sync();
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the order array is in L2 before any lane reads it back
			for (uint32_t it = 0; it < n_it; it += 64) {
				const bool mine = it + lane < n_it;
				const uint32_t item = __hip_atomic_load(&order[mine ? it + lane : n_it - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				gap_round<MAXL>(lds, v, table, locate(chunk + item), mine, pend, tl);
			}
		}
		gap_flush<MAXL>(lds, table, pend, tl);
	}
}

#ifdef PGX_STAGE_PROBES
} // namespace pgx
extern "C" int pgx_gap_stats(unsigned long long *out, int reset) // measurement builds only (tools/probe_gapstats.py)
{
	unsigned long long z[8] = { 0 };
	if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(pgx::g_gap_stats), sizeof z) != hipSuccess)
		return -1;
	if (reset && hipMemcpyToSymbol(HIP_SYMBOL(pgx::g_gap_stats), z, sizeof z) != hipSuccess)
		return -1;
	return 0;
}
namespace pgx {
#endif

// ------------------------------------------------------------------------------------------ one wavefront per HSP
// cell: x = i, y = mismatches | gap openings << 12 | kind << 24 | matched-after << 26
// DMAX = differences per side this instance can hold: 62 (the diagonals of a level fit one wavefront, 2 KB of LDS: 32
// wavefronts per CU) for nearly every listed HSP, kGDmax = 1 000 (32 KB: 5 per CU) for the few sides that need more --
// with the large rows only, reads of 300-500 bases, most of whose HSPs are listed, ran at 5 wavefronts per CU
constexpr int kBigStageL = 2048; // reads up to this length have their letters staged in LDS by the small tier
constexpr int kDbPadBases = 768; // zero letters in front of and behind the database's packed words (seqdb.hip: 24 words each)
template <int DMAX> struct BigLds {
	uint2 row[2][2 * DMAX + 3];
	int ring[kGLag + 1]; // best score with at most d differences, for the last kGLag + 1 values of d
	// the small tier's staged letters (16 per word): the read strand, and the database window from DMAX + 16 bases left of
	// where the read's first base would lie
	static constexpr int kRd = DMAX < kGDmax ? kBigStageL / 16 + 2 : 1, kDb = DMAX < kGDmax ? (kBigStageL + 2 * DMAX + 48 + 15) / 16 + 2 : 1;
	uint32_t rd[kRd], db[kDb];
};

// lcp<DIR> on letters staged in LDS (no ambiguity flags: HSPs that touch one read the sequences in memory)
__device__ __forceinline__ uint32_t window16_lds(const uint32_t *w, int pos)
{
	const int i = pos >> 4;
	const uint64_t v = (uint64_t)w[i] | ((uint64_t)w[i + 1] << 32);
	return (uint32_t)(v >> ((pos & 15) * 2));
}
template <int DIR> __device__ __forceinline__ int lcp_lds(const uint32_t *rd, const uint32_t *db, int qp, int dp, int cap)
{
	int n = 0;
	while (n < cap) {
		const int take = cap - n < 16 ? cap - n : 16;
		const int q0 = DIR > 0 ? qp + n : qp - n - (take - 1), d0 = DIR > 0 ? dp + n : dp - n - (take - 1);
		const uint32_t x = window16_lds(rd, q0) ^ window16_lds(db, d0);
		uint32_t y = (x | (x >> 1)) & 0x55555555u;
		if (take < 16)
			y &= (1u << (2 * take)) - 1u;
		int run;
		if (DIR > 0) {
			run = y ? (__ffs((int)y) - 1) >> 1 : take;
		} else {
			y <<= 2 * (16 - take);
			run = y ? __clz((int)y) >> 1 : take;
		}
		n += run;
		if (run < take)
			break;
	}
	return n;
}

// returns false when cells were still alive after DMAX < kGDmax differences (the side needs the larger instance)
// staged: the letters are in lds->rd / lds->db and `dwin` is d0's position in that window
template <int DIR, int DMAX>
__device__ bool greedy_big(BigLds<DMAX> *lds, const GapSeqs &s, bool staged, int q0, int64_t d0, int dwin, int M, int N, Side &out)
{
	const int lane = threadIdx.x & 63;
	auto slide = [&](int i, int j) {
		const int cap = M - i < N - j ? M - i : N - j;
		if (DMAX < kGDmax && staged)
			return lcp_lds<DIR>(lds->rd, lds->db, q0 + DIR * i, dwin + DIR * j, cap);
		return lcp<DIR>(s, q0 + DIR * i, d0 + DIR * j, cap);
	};
	const int i0 = slide(0, 0);
	out.i = out.j = i0;
	out.s2 = 2 * i0;
	out.mism = out.gopen = 0;
	if (i0 == M || i0 == N)
		return true;
	constexpr int C = DMAX + 1;
	int cur_row = 0;
	if (lane == 0) {
		lds->row[0][C] = make_uint2((uint32_t)i0, i0 > 0 ? 1u << 26 : 0u);
		lds->ring[0] = 2 * i0;
	}
	lds_sync();
	int best = 2 * i0, L = 0, U = 0;
	for (int d = 1; d <= DMAX; d++) {
		const int tcmp = d >= kGLag ? lds->ring[(d - kGLag) % (kGLag + 1)] : 0;
		const uint2 *pa = lds->row[cur_row];
		uint2 *pb = lds->row[cur_row ^ 1];
		int nl = 1 << 20, nu = -(1 << 20);
		for (int kb = L - 1; kb <= U + 1; kb += 64) {
			const int k = kb + lane;
			const bool in = k <= U + 1;
			// rows are read through [L, U] only, and every cell of [L - 1, U + 1] is written each round (dead ones
			// as kBigNone), so no stale cell of an earlier round is ever read
			auto old = [&](int kk, uint2 &c) {
				bool ok = kk >= L && kk <= U;
				if (ok) {
					c = pa[C + kk];
					ok = c.x != kBigNone;
				}
				return ok;
			};
			uint2 cp, cc, cn, p = make_uint2(0u, 0u);
			int v = -1, par = 0;
			if (in) {
				if (old(k, cc)) {
					v = (int)cc.x + 1;
					p = cc;
				}
				if (old(k - 1, cp) && (int)cp.x + 1 > v) {
					v = (int)cp.x + 1;
					par = 1;
					p = cp;
				}
				if (old(k + 1, cn) && (int)cn.x > v) {
					v = (int)cn.x;
					par = 2;
					p = cn;
				}
			}
			int ii = v, jj = v - k, s2 = -(1 << 30);
			uint32_t stats = 0;
			const int ub = (2 * M - k < 2 * N + k ? 2 * M - k : 2 * N + k) - 6 * d;
			const bool alive = in && v >= 0 && ii <= M && jj <= N && jj >= 0 && ii + jj - 6 * d >= tcmp - kGX2 && ub > best;
			if (alive) {
				const int run = slide(ii, jj);
				ii += run;
				jj += run;
				const uint32_t pk = (p.y >> 24) & 3u, pslid = (p.y >> 26) & 1u;
				const uint32_t mism = (p.y & 0xFFFu) + (par == 0 ? 1u : 0u);
				const uint32_t gopen = ((p.y >> 12) & 0xFFFu) + ((par != 0 && !(pk == (uint32_t)par && !pslid)) ? 1u : 0u);
				stats = mism | (gopen << 12) | ((uint32_t)par << 24) | (run > 0 ? 1u << 26 : 0u);
				s2 = ii + jj - 6 * d;
			}
			if (in)
				pb[C + k] = alive ? make_uint2((uint32_t)ii, stats) : make_uint2(kBigNone, 0u);
			const unsigned long long am = __ballot(alive);
			if (am) {
				const int lo = __ffsll((unsigned long long)am) - 1, hi = 63 - __clzll((long long)am);
				nl = kb + lo < nl ? kb + lo : nl;
				nu = kb + hi > nu ? kb + hi : nu;
				// the first cell (k ascending) that reaches the largest score of this chunk
				int m = s2;
#pragma unroll
				for (int sh = 32; sh >= 1; sh >>= 1) {
					const int t = __shfl_xor(m, sh);
					m = t > m ? t : m;
				}
				if (m > best) {
					const int w = __ffsll((unsigned long long)__ballot(alive && s2 == m)) - 1;
					best = m;
					out.i = __shfl(ii, w);
					out.j = __shfl(jj, w);
					out.s2 = m;
					const uint32_t st = __shfl(stats, w);
					out.mism = (int)(st & 0xFFFu);
					out.gopen = (int)((st >> 12) & 0xFFFu);
				}
			}
		}
		if (lane == 0)
			lds->ring[d % (kGLag + 1)] = best;
		lds_sync();
		if (nl > nu)
			return true;
		cur_row ^= 1;
		L = nl;
		U = nu;
	}
	return DMAX == kGDmax; // (the spec's limit: what was found within 1 000 differences stands)
}

template <int DMAX>
__global__ __launch_bounds__(64) void k_gapped_big(GapView v, const unsigned long long *__restrict__ list, const uint32_t *__restrict__ count,
						    uint32_t cap, unsigned long long *__restrict__ next_list, uint32_t *__restrict__ next_count)
{
	__shared__ BigLds<DMAX> lds;
	const uint32_t n = *count < cap ? *count : cap;
	for (uint32_t idx = blockIdx.x; idx < n; idx += gridDim.x) {
		pgx_hit *hp = reinterpret_cast<pgx_hit *>((uintptr_t)list[idx]);
		const pgx_hit h = *hp;
		const Anchor a = anchor_of(v, h);
		Side l, r;
		bool staged = false;
		int awin = 0;
		if (DMAX < kGDmax && a.L <= kBigStageL) {
			// letters into LDS unless the read or the window holds an ambiguity letter (then the flag words are needed)
			const int lane = threadIdx.x & 63;
			const int64_t lo = ((int64_t)a.gpos - a.qa - DMAX - 16) >> 4; // window start, in 16-base words (may be negative: padding)
			bool amb = false;
			if (a.s.ra)
				for (int w = lane; w < (a.L + 31) / 32; w += 64)
					amb = amb || a.s.ra[w] != 0;
			const int n_db = (a.L + 2 * DMAX + 48 + 15) / 16 + 1;
			if (a.s.dba) {
				if (v.amb_blk) {
					const int64_t b0 = (lo * 16) >> kBlkShift, b1 = (lo * 16 + n_db * 16) >> kBlkShift;
					for (int64_t bb = (b0 < 0 ? 0 : b0) + lane; bb <= b1; bb += 64)
						amb = amb || ((v.amb_blk[bb >> 5] >> (bb & 31)) & 1u);
				} else {
					amb = true;
				}
			}
			// (a long query that overhangs the first or the last subject by more than the padding: the window would begin
			// before / end behind the packed words -- such an HSP reads its letters in memory, position by position, where
			// the caps of the slides keep every access inside)
			const bool inside = lo * 16 >= -(int64_t)kDbPadBases && (lo + n_db + 1) * 16 <= v.db_bases + kDbPadBases;
			if (__ballot(amb) == 0ull && inside) {
				const uint32_t *gr = reinterpret_cast<const uint32_t *>(a.s.rw), *gd = reinterpret_cast<const uint32_t *>(a.s.dbw) + lo;
				for (int w = lane; w < a.L / 16 + 2; w += 64)
					lds.rd[w] = gr[w];
				for (int w = lane; w <= n_db; w += 64)
					lds.db[w] = gd[w];
				staged = true;
				awin = (int)((int64_t)a.gpos - lo * 16);
			}
			lds_sync();
		}
		const bool okl = greedy_big<-1, DMAX>(&lds, a.s, staged, a.qa - 1, a.S0 + a.sa - 1, awin - 1, a.qa, a.sa, l);
		lds_sync();
		const bool okr = okl && greedy_big<+1, DMAX>(&lds, a.s, staged, a.qa, a.S0 + a.sa, awin, a.L - a.qa, a.slen - a.sa, r);
		lds_sync();
		if ((threadIdx.x & 63) == 0) {
			if (okl && okr) {
				write_gapped(hp, h, a, l, r);
			} else if (next_list) {
				const uint32_t w = atomicAdd(next_count, 1u);
				if (w < cap)
					next_list[w] = list[idx];
			}
		}
	}
}

// ------------------------------------------------------------------------------------------ one wavefront per HSP, one LANE per diagonal
// Round 3.  k_gapped_big keeps a level's cells in LDS, two rows of (i, statistics), read through three
// guarded loads per cell, and finds the level's best with six shuffles: ~850 SIMD cycles per level, the whole of config 2's
// 14.6 ms.  Here the level IS the wavefront: lane l holds the cell of diagonal k = l - 32 in ONE register (the lean word of
// the first tier: i << 17 | priority << 14 | statistics), its neighbours come by DPP wave shifts, no LDS row, no barrier.
// The live diagonals of an X-drop alignment stay within 2 X / 6 + 1 = 19 of the best one, so 64 lanes hold them unless the
// path itself drifts by > ~12 net gap columns; an HSP whose live cells reach lane 0 or 63 is passed on to k_gapped_big.
//   statistics (14 bits): gap openings [5:0] | open-gap kind [7:6] | G1 = gap columns in the subject row [13:8].  The kind
//   is stored AS THE PRIORITY of the column that made it (1: from k - 1, a gap in the subject row; 0: from k + 1, a gap in
//   the query row; 2: none -- a mismatch column, or letters matched after the gap), so "this gap column continues the
//   parent's open gap" is one XOR of two fields of the winning parent's word.  Mismatches = d - gap columns, gap columns in
//   the query row = G1 - k.  A count that reaches 56 (looked at every eighth level) hands the HSP on.
//   best cell: every lane keeps the best of ITS diagonal (score << 10 | 1023 - d: first d among equal scores) and the word
//   of that cell; one reduction at the end of the side picks the first lane among the best.  The X-drop history T[d] (best
//   score within d differences) needs the wave's maximum per level: six DPP steps; T itself lies across the lanes of one
//   register (lane d & 63 holds T[d]: the test of level d reads T[d - 19]).
// Letters: the side's part of the read and the subject window, forward in the direction of extension (the left side
// reversed, as in the lean rows), 1.1 KB of LDS per wavefront.
constexpr int kDiagL = 2048;                       // longest read this kernel stages
constexpr int kDiagFront = 3;                      // spare words in front of each sequence (dead cells and diagonals k > i read there)
constexpr int kDiagQW = kDiagFront + kDiagL / 16 + 5;
constexpr int kDiagDW = kDiagFront + (kDiagL + 64) / 16 + 6;
constexpr uint32_t kDiagDead = 0xFFF80000u;        // i = -4
constexpr uint32_t kDiagFromCur = 0x28000u;        // i + 1, priority 2
constexpr uint32_t kDiagFromPrev = 0x24100u;       // i + 1, priority 1, G1 + 1
struct DiagLds {
	uint32_t seq[kDiagQW + kDiagDW];
};

__device__ __forceinline__ int dpp_from_lower(int v, int dead) { return __builtin_amdgcn_update_dpp(dead, v, 0x138, 0xf, 0xf, false); } // lane l <- lane l - 1
__device__ __forceinline__ int dpp_from_upper(int v, int dead) { return __builtin_amdgcn_update_dpp(dead, v, 0x130, 0xf, 0xf, false); } // lane l <- lane l + 1
// the wavefront's maximum, in every lane.  v_max_i32 with a DPP operand (dst = src1 = v: a lane without a source, or in a
// masked row, keeps v); a VALU write needs two wait states before a DPP read of the register
__device__ __forceinline__ int wave_max_i32(int v)
{
	asm volatile("s_nop 1\n\t"
		     "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
		     "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
		     "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
		     "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" // lane 15 of each row: the row's maximum
		     "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t" // into rows 1 and 3
		     "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" // into rows 2 and 3: lane 63 holds the maximum
		     : "+v"(v));
	return __builtin_amdgcn_readlane(v, 63);
}

// 16 letters at letter position p of packed words g (16 letters per word), words outside [w_lo, w_hi] read as zero
__device__ __forceinline__ uint32_t diag_window(const uint32_t *g, int64_t p, int64_t w_lo, int64_t w_hi)
{
	const int64_t w = p >> 4;
	const uint32_t lo = w >= w_lo && w <= w_hi ? g[w] : 0u, hi = w + 1 >= w_lo && w + 1 <= w_hi ? g[w + 1] : 0u;
	return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(p & 15) * 2u);
}

// one side.  seq0: the staged letters (read part at word 0, subject window at word kDiagQW, letter 0 of each kDiagFront
// words in); M, N = letters of the read / of the subject on the side.  Returns 0 done, 1 hand the HSP on.
__device__ __forceinline__ int diag_side(const lds_word *seq0, int M, int N, Side &out)
{
	const int lane = threadIdx.x & 63, k = lane - 32;
	constexpr int QB = 32 * kDiagFront, DBb = 32 * (kDiagQW + kDiagFront);
	auto ld = [&](uint32_t bit) { // the 16 letters from bit offset `bit` of the staged words on
		const uint32_t a = (bit >> 3) & 0xFFFFFFFCu;
		return __builtin_amdgcn_alignbit(lds_ld(seq0, a + 4), lds_ld(seq0, a), bit);
	};
	auto lcp2 = [&](uint32_t qb, uint32_t db, uint32_t cap2) { // matching letters as bits: min(32, cap2, 2 run)
		const uint32_t x = ld(qb) ^ ld(db);
		const uint32_t y = (x | (x >> 1)) & 0x55555555u;
		return min3u(scan_low(y), 32u, cap2);
	};
	// ---- the first run (the same in every lane)
	int i2 = 0;
	for (;;) {
		const int cap2 = min(2 * M, 2 * N) - i2;
		if (cap2 <= 0)
			break;
		const uint32_t r2 = lcp2((uint32_t)(QB + i2), (uint32_t)(DBb + i2), (uint32_t)cap2);
		i2 += (int)r2;
		if (r2 < 32u)
			break;
	}
	out.i = out.j = i2 >> 1;
	out.s2 = i2;
	out.mism = out.gopen = 0;
	if (i2 == 2 * M || i2 == 2 * N)
		return 0;
	// per-lane constants of the diagonal
	const int E = min(2 * M, 2 * N + 2 * k) + QB;   // end of the shorter sequence on this diagonal, as a read bit offset
	const int F = min(2 * M - k, 2 * N + k);        // the bound of a cell of this diagonal, before the 6 d
	const uint32_t Qc = (uint32_t)QB << 16, Dl = (uint32_t)(DBb - QB - 2 * k) << 16;
	const int k16 = (int)((uint32_t)k << 16);
	int R = lane == 32 ? (int)(((uint32_t)i2 << 16) | 0x80u) : (int)kDiagDead;
	int lane_key = lane == 32 ? (i2 << 10) | 1023 : (int)0x80000000u;
	int lane_word = R;
	int best = i2;   // wave-uniform
	int T = i2;      // lane (d & 63) holds T[d]; lane 0: T[0]
	int status = 0;
	for (int d = 1; d <= kGDmax; d++) {
		const int tcmp = d >= kGLag ? __builtin_amdgcn_readlane(T, (d - kGLag) & 63) : 0;
		const int thr16 = (int)((uint32_t)(tcmp - kGX2 + 6 * d) << 16); // 2 i - k >= tcmp - 2 X + 6 d, on the word's upper half
		const int a = R + (int)kDiagFromCur, b = dpp_from_lower(R, (int)kDiagDead) + (int)kDiagFromPrev, c = dpp_from_upper(R, (int)kDiagDead);
		const int m3 = max(a, max(b, c));
		const uint32_t t = (uint32_t)m3 + Qc, u = t + Dl;
		const uint32_t qbit = (uint32_t)((int)t >> 16), dbit = (uint32_t)((int)u >> 16);
		const int cap2 = E - (int)qbit;
		// (m3 - k16 compares the word's upper half with the threshold: the low half is below 2^16 and k16, thr16 have none)
		const bool ok = ((m3 | cap2) >= 0) & (F - 6 * d > best) & ((m3 & (int)0xFFFF0000u) - k16 >= thr16);
		uint32_t r2 = lcp2(qbit, dbit, (uint32_t)cap2);
		// statistics of the winning parent -> of this cell
		const uint32_t w = (uint32_t)m3;
		const uint32_t pk = (w >> 8) & 0xC0u;                         // the priority, where the kind lies
		const uint32_t differs = ((w ^ pk) & 0xC0u) != 0u ? 1u : 0u;  // the parent's open gap is not of this column's kind
		const uint32_t inc = (w & 0x8000u) ? 0u : differs;            // a gap column (priority 0 or 1) that opens a gap
		// letters past the first 16
		int q2 = (int)(w >> 16) + (int)r2; // 2 i (m3 >= 0 where it counts)
		bool more = ok && r2 == 32u;
		const bool slid = r2 != 0u;
		while (__builtin_amdgcn_ballot_w64(more) != 0ull) { // (the builtin: __ballot went through a select and a compare here)
			if (more) {
				const int c2 = E - (QB + q2);
				if (c2 <= 0) {
					more = false;
				} else {
					const uint32_t rr = lcp2((uint32_t)(QB + q2), (uint32_t)((int)(Dl >> 16) + QB + q2), (uint32_t)c2);
					q2 += (int)rr;
					more = rr == 32u;
				}
			}
		}
		const uint32_t nc = ((uint32_t)q2 << 16) | (((w & 0x3F3Fu) + inc) | (slid ? 0x80u : pk));
		const int nv = ok ? (int)nc : (int)kDiagDead;
		const int s2 = (nv >> 16) - k - 6 * d;
		const int key = nv >= 0 ? (int)((uint32_t)s2 << 10) + (1023 - d) : (int)0x80000000u;
		if (key > lane_key) {
			lane_key = key;
			lane_word = nv;
		}
		R = nv;
		const unsigned long long am = __builtin_amdgcn_ballot_w64(nv >= 0);
		if (am == 0ull)
			break;
		// live cells at the wavefront's ends, or a statistics field about to overflow: the wide kernel takes the HSP
		// (the counts grow by at most one a level: looked at every eighth level, handed on from 56)
		if ((am & 0x8000000000000001ull) != 0ull) {
			status = 1;
			break;
		}
		if ((d & 7) == 0) {
			const bool near_full = nv >= 0 && ((((uint32_t)nv & 0x3F3Fu) + 0x0808u) & 0x4040u) != 0u;
			if (__builtin_amdgcn_ballot_w64(near_full) != 0ull) {
				status = 1;
				break;
			}
		}
		// (every live key of the level carries the same 1023 - d below the score: the largest key holds the largest score)
		best = max(best, wave_max_i32(key) >> 10);
		T = lane == (d & 63) ? best : T;
	}
	if (status)
		return status;
	const int top = wave_max_i32(lane_key);
	const int wl = __ffsll((unsigned long long)__ballot(lane_key == top)) - 1;
	const uint32_t st = (uint32_t)__builtin_amdgcn_readlane(lane_word, wl);
	const int bk = wl - 32, bd = 1023 - (top & 1023);
	const int g1 = (int)((st >> 8) & 63u), g2 = g1 - bk;
	out.s2 = top >> 10;
	out.i = (int)(st >> 17);
	out.j = out.i - bk;
	out.gopen = (int)(st & 63u);
	out.mism = bd - (g1 + g2);
	return 0;
}

__global__ __launch_bounds__(64) void k_gapped_diag(GapView v, const unsigned long long *__restrict__ list, const uint32_t *__restrict__ count, uint32_t cap,
						     unsigned long long *__restrict__ next_list, uint32_t *__restrict__ next_count)
{
	__shared__ DiagLds lds;
	const lds_word *seq0 = (const lds_word *)&lds.seq[0];
	const int lane = threadIdx.x & 63;
	const uint32_t n = *count < cap ? *count : cap;
	for (uint32_t idx = blockIdx.x; idx < n; idx += gridDim.x) {
		pgx_hit *hp = reinterpret_cast<pgx_hit *>((uintptr_t)list[idx]);
		const pgx_hit h = *hp;
		const Anchor a = anchor_of(v, h);
		bool fits = a.L <= kDiagL;
		// ambiguity letters (read or window): the wide kernel reads those HSPs in memory, flag words and all
		const int64_t g_lo = (int64_t)a.gpos - a.qa - 160, g_hi = (int64_t)a.gpos + (a.L - a.qa) + 160; // every letter either side may stage
		if (fits) {
			bool amb = false;
			if (a.s.ra)
				for (int w = lane; w < (a.L + 31) / 32; w += 64)
					amb = amb || a.s.ra[w] != 0;
			if (a.s.dba) {
				if (v.amb_blk) {
					// (letters outside the database read as zero below and lie beyond every cap)
					const int64_t b0 = g_lo >> kBlkShift, b1 = (g_hi < v.db_bases ? g_hi : v.db_bases - 1) >> kBlkShift;
					for (int64_t bb = (b0 < 0 ? 0 : b0) + lane; bb <= b1; bb += 64)
						amb = amb || ((v.amb_blk[bb >> 5] >> (bb & 31)) & 1u);
				} else {
					amb = true;
				}
			}
			fits = __ballot(amb) == 0ull;
		}
		Side sd[2];
		int status = fits ? 0 : 1;
		const uint32_t *gr = reinterpret_cast<const uint32_t *>(a.s.rw), *gd = reinterpret_cast<const uint32_t *>(a.s.dbw);
		const int64_t rd_hi = a.L / 16 + 1, db_lo = -(int64_t)(kDbPadBases / 16), db_hi = (v.db_bases + kDbPadBases) / 16 - 1;
		for (int side = 0; side < 2 && status == 0; side++) {
			// side 0: left of the anchor, reversed (letter y of the staged read part = read letter qa - 1 - y); side 1: right
			const int M = side ? a.L - a.qa : a.qa, N = side ? a.slen - a.sa : a.sa;
			const int nq = kDiagFront + (M + 34) / 16 + 2, nd = kDiagFront + (M + 32 + 34) / 16 + 3;
			for (int x = lane; x < nq; x += 64) {
				const int y0 = 16 * (x - kDiagFront); // first staged letter of the word
				lds.seq[x] = side ? diag_window(gr, (int64_t)a.qa + y0, 0, rd_hi) : __builtin_bitreverse32(diag_window(gr, (int64_t)a.qa - 1 - y0 - 15, 0, rd_hi));
			}
			for (int x = lane; x < nd; x += 64) {
				const int y0 = 16 * (x - kDiagFront);
				lds.seq[kDiagQW + x] = side ? diag_window(gd, (int64_t)a.gpos + y0, db_lo, db_hi)
							     : __builtin_bitreverse32(diag_window(gd, (int64_t)a.gpos - 1 - y0 - 15, db_lo, db_hi));
			}
			lds_sync();
			status = diag_side(seq0, M, N, sd[side]);
			lds_sync();
		}
		if (lane == 0) {
			if (status == 0) {
				write_gapped(hp, h, a, sd[0], sd[1]);
			} else if (next_list) {
				const uint32_t w = atomicAdd(next_count, 1u);
				if (w < cap)
					next_list[w] = list[idx];
			}
		}
	}
}

int gapped_stage(const DbView &dv, const ReadsView &rv, pgx_hit *main_table, const uint8_t *main_key, const uint32_t *read_start,
		 const uint32_t *read_cnt, pgx_hit *ovf_table, const uint8_t *ovf_key, const unsigned long long *ovf_count, unsigned long long ovf_cap, bool long_reads,
		 unsigned long long hit_cap, int max_len, GappedWork &gw, hipStream_t stream, const unsigned long long *main_used, const uint8_t *main_reg)
{
	GapView v;
	v.fwd = rv.fwd;
	v.rc = rv.rc;
	v.fwd_amb = rv.fwd_amb;
	v.rc_amb = rv.rc_amb;
	v.len = rv.len;
	v.woff = rv.woff;
	v.dbw = dv.words;
	v.dba = dv.amb;
	v.seq_off = dv.seq_off;
	v.db_bases = dv.n_bases;
	v.amb_blk = dv.amb ? dv.amb_blk : nullptr;
	// HSPs the first tier passes on: those whose best cell holds two or more gap columns and those still alive at 18
	// differences (about one in a hundred for substitution-only reads, more for reads with indels); every one for long queries
	const unsigned long long want = long_reads ? hit_cap + ovf_cap : std::max<unsigned long long>(1ull << 20, (hit_cap + ovf_cap) / 16);
	const uint32_t big_cap = (uint32_t)std::min<unsigned long long>(want, 0xFFFFFFF0ull);
	PGX_TRY(gw.big_list.ensure(big_cap));
	PGX_TRY(gw.big_count.ensure(8));
#ifdef PGX_STAGE_PROBES
	// measurement builds keep round 3's comparison forms of the main-table tier (read once per process)
	static const bool pools1_env = getenv("PGX_GAP_POOLS1") != nullptr; // the one-pass pools
	static const bool pools2 = getenv("PGX_GAP_POOLS2") != nullptr;     // round 2's two-pass pools
	const bool binned = !long_reads && !pools1_env;
#else
	constexpr bool pools2 = false;
	const bool binned = !long_reads;
#endif
	if (binned) {
		PGX_TRY(gw.items.ensure(hit_cap)); // the main table's slots in (region, bin) order
		PGX_TRY(gw.items1.ensure(hit_cap + 16)); // (read as 16-byte words / 4-byte words of keys)
		PGX_TRY(gw.keys1.ensure(hit_cap + 16));
		PGX_TRY(gw.bins.ensure(kBinsWords));
	}
	if (pools2) {
		PGX_TRY(gw.side_main.ensure(hit_cap)); // the left side's result of every HSP, parked between the two passes
		PGX_TRY(gw.side_ovf.ensure(ovf_cap));
	}
	PGX_TRY(gw.order.ensure((size_t)10240 * kBlkItems));
	// (the list may have been grown by the caller after a step that overflowed it: the later tiers' lists and the parked
	// sides of the list tiers follow ITS size, not the first guess -- they are indexed by its entries)
	const uint32_t cap = (uint32_t)std::min<size_t>(gw.big_list.n, 0xFFFFFFF0ull);
	PGX_TRY(gw.big_list2.ensure(cap));
	PGX_TRY(gw.side_list.ensure(long_reads ? 1 : cap));
	PGX_HIP(hipMemsetAsync(gw.big_count.data(), 0, 8 * sizeof(uint32_t), stream));
	const uint32_t n = rv.n;
	const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)n + 63) / 64, 256ull * 40);
	const int dbg = 0; // (round 2's truncation probes are gone: a truncated stage leaves seed records where the stages behind expect hits)
	if (binned) {
		// the main table's slots by database region, within a region by key byte, costliest first: counting sorts that
		// stream over one byte (then five) per slot
		PGX_HIP(hipMemsetAsync(gw.bins.data(), 0, kBinsWords * sizeof(uint32_t), stream));
		hipLaunchKernelGGL(k_reg_hist, dim3(256 * 8), dim3(256), 0, stream, main_key, main_reg, main_used, hit_cap, gw.bins.data());
		hipLaunchKernelGGL(k_reg_bases, dim3(1), dim3(256), 0, stream, gw.bins.data());
		hipLaunchKernelGGL(k_reg_scatter, dim3(256 * 8), dim3(256), 0, stream, main_key, main_reg, main_used, hit_cap, gw.bins.data(), gw.items1.data(),
				   gw.keys1.data());
		hipLaunchKernelGGL(k_seg_hist, dim3(256 * 8), dim3(256), 0, stream, gw.keys1.data(), gw.bins.data());
		hipLaunchKernelGGL(k_seg_bases, dim3(256), dim3(256), 0, stream, gw.bins.data());
		hipLaunchKernelGGL(k_seg_scatter, dim3(256 * 8), dim3(256), 0, stream, gw.keys1.data(), gw.items1.data(), gw.bins.data(), gw.items.data());
	}
	unsigned long long *listA = gw.big_list.data(), *listB = gw.big_list2.data();
	uint32_t *cnt = gw.big_count.data(); // [0] list A, [1] B (both checked by the caller), [2] C (in A's buffer), [3] D (in B's), [4] E (in A's), [5] the first tier's own appends to B
	// (long reads: every HSP goes straight to the wide kernels, which read list A; reads of 321-512 bases: the first tier
	// holds 40 differences a side with the full statistics, what it cannot finish goes to list A and the wide kernels too)
	static const bool no_deep = getenv("PGX_GAP_NODEEP") != nullptr; // (measurement aid, read once per process)
	const bool deep = !long_reads && max_len > gapped_deep_from() && !no_deep;
	const bool one_list = long_reads || deep;
	const TierLists tl = { listA, one_list ? listA : listB, cnt, one_list ? cnt : cnt + 1, cap, one_list ? nullptr : cnt + 5 };
	// staged sequences sized for the batch's longest read (the LDS footprint decides the occupancy).  The lean rows over
	// the two tables; then the rows with the full statistics over what they listed (list A), which lists for the wider tiers (B)
#ifdef PGX_STAGE_PROBES
#define PGX_POOLS2_MAIN(ML, WV)                                                                                                             \
	hipLaunchKernelGGL((k_gapped_fast<0, ML, WV, kGFastD>), dim3(grid_wv ? grid_wv : 1), dim3(64), 0, stream, v, main_table, hit_cap, read_start, \
			   read_cnt, n, (const unsigned long long *)nullptr, listA, cnt, cap, gw.side_main.data(), dbg, gw.order.data())
#define PGX_POOLS2_OVF(ML, WV)                                                                                                              \
	hipLaunchKernelGGL((k_gapped_fast<1, ML, WV, kGFastD>), dim3(256), dim3(64), 0, stream, v, ovf_table, ovf_cap, (const uint32_t *)nullptr, \
			   (const uint32_t *)nullptr, 0u, ovf_count, listA, cnt, cap, gw.side_ovf.data(), dbg, gw.order.data())
#else
#define PGX_POOLS2_MAIN(ML, WV) ((void)0)
#define PGX_POOLS2_OVF(ML, WV) ((void)0)
#endif
#define PGX_GAPPED_LAUNCH(ML, WV, DD)                                                                                                          \
	do {                                                                                                                                 \
		v.key = main_key;                                                                                                            \
		const unsigned grid_wv = std::min<unsigned>(grid, 256u * 4u * WV * 2u); /* two full rounds of resident wavefronts */  \
		if (binned)                                                                                                                  \
			hipLaunchKernelGGL((k_gapped_rows<ML, (WV > 4 ? 4 : WV), DD>), dim3(256u * 4u * (WV > 4 ? 4 : WV) * 2u), dim3(64), 0, stream, v, main_table, \
					   gw.items.data(), gw.bins.data(), tl);                                                                  \
		else if (pools2)                                                                                                             \
			PGX_POOLS2_MAIN(ML, WV);                                                                                                 \
		else                                                                                                                         \
			hipLaunchKernelGGL((k_gapped_pool<false, ML, (WV > 4 ? 4 : WV)>), dim3(std::max(1u, std::min<unsigned>(grid, 256u * 4u * (WV > 4 ? 4 : WV) * 2u))), dim3(64), 0, stream, \
					   v, main_table, hit_cap, read_start, read_cnt, n, (const unsigned long long *)nullptr, tl, gw.order.data()); \
		v.key = ovf_key;                                                                                                             \
		if (pools2)                                                                                                                  \
			PGX_POOLS2_OVF(ML, WV);                                                                                                  \
		else                                                                                                                         \
			hipLaunchKernelGGL((k_gapped_pool<true, ML, (WV > 4 ? 4 : WV)>), dim3(256), dim3(64), 0, stream, v, ovf_table, ovf_cap,       \
					   (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, ovf_count, tl, gw.order.data()); \
		if (!one_list)                                                                                                               \
			hipLaunchKernelGGL((k_gapped_fast<2, ML, WV, kGFastD>), dim3(256 * 4 * WV), dim3(64), 0, stream, v,                      \
					   reinterpret_cast<pgx_hit *>(listA), (unsigned long long)cap, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, \
					   reinterpret_cast<const unsigned long long *>(cnt), listB, cnt + 1, cap, gw.side_list.data(), 0, gw.order.data()); \
	} while (0)
	if (max_len <= 160)
		PGX_GAPPED_LAUNCH(160, PGX_LEAN_WAVES, kGFastD);
	else if (max_len <= 192)
		PGX_GAPPED_LAUNCH(192, 4, kGFastD);
	else if (max_len <= 320 && deep)
		PGX_GAPPED_LAUNCH(320, 2, kGFastD2);
	else if (max_len <= 320)
		PGX_GAPPED_LAUNCH(320, 3, kGFastD);
	else if (deep)
		PGX_GAPPED_LAUNCH(512, 2, kGFastD2);
	else
		PGX_GAPPED_LAUNCH(512, 2, kGFastD);
#undef PGX_GAPPED_LAUNCH
#undef PGX_POOLS2_MAIN
#undef PGX_POOLS2_OVF
	// the wider tiers, each passing on what it cannot hold: the lane-per-HSP kernel with rows for 40 differences a side
	// and the X-drop history (reads of <= 512 bases without ambiguity letters: most of what reads of 300-500 bases
	// list), then one wavefront per HSP with rows for 62 differences, then for the spec's 1 000
	if (!one_list)
		hipLaunchKernelGGL((k_gapped_fast<2, 512, 2, kGFastD2>), dim3(256 * 6 * 2), dim3(64), 0, stream, v,
				   reinterpret_cast<pgx_hit *>(listB), (unsigned long long)cap, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u,
				   reinterpret_cast<const unsigned long long *>(cnt + 1), listA, cnt + 2, cap,
				   gw.side_list.data(), 0, gw.order.data());
	// (list C, or list A itself for long reads) -> one lane per diagonal -> D -> the LDS rows for 62 differences -> E -> for 1 000
	const uint32_t *c_diag = one_list ? cnt : cnt + 2;
	static const bool no_diag = getenv("PGX_GAP_NODIAG") != nullptr; // (measurement aid: the wide kernels alone, as in round 2)
	if (no_diag) {
		hipLaunchKernelGGL(k_gapped_big<62>, dim3(256 * 32), dim3(64), 0, stream, v, (const unsigned long long *)listA, c_diag, cap, listB, cnt + 3);
		hipLaunchKernelGGL(k_gapped_big<kGDmax>, dim3(256 * 8), dim3(64), 0, stream, v, (const unsigned long long *)listB, cnt + 3, cap,
				   (unsigned long long *)nullptr, (uint32_t *)nullptr);
	} else {
		hipLaunchKernelGGL(k_gapped_diag, dim3(256 * 32), dim3(64), 0, stream, v, (const unsigned long long *)listA, c_diag, cap, listB, cnt + 3);
		hipLaunchKernelGGL(k_gapped_big<62>, dim3(256 * 32), dim3(64), 0, stream, v, (const unsigned long long *)listB, cnt + 3, cap, listA, cnt + 4);
		hipLaunchKernelGGL(k_gapped_big<kGDmax>, dim3(256 * 8), dim3(64), 0, stream, v, (const unsigned long long *)listA, cnt + 4, cap,
				   (unsigned long long *)nullptr, (uint32_t *)nullptr);
	}
	PGX_HIP(hipGetLastError());
	return 0;
}

} // namespace pgx
