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
// Two kernels, the same results:
//   k_gapped_fast  one LANE per HSP (reads of <= 512 bases, <= 15 differences per side): the R row of the lane lives in
//                  33 LDS words and is updated in place (k ascending, the two neighbours carried in registers); a cell
//                  carries its own statistics (mismatches, gap openings, kind of the last column), so there is no
//                  traceback.  HSPs that need more go on a list.
//   k_gapped_big   one WAVEFRONT per listed HSP, one lane per diagonal (64 at a time, <= 1000 differences per side), rows
//                  double-buffered in LDS.
// Both cut a cell whose score could not pass the best one even if every remaining letter matched.  The cut cannot change
// the result (a child's bound is below its parent's, so no surviving cell has a cut parent; a cut cell never holds the
// best score), which is why the sequential kernel, the parallel one (bound taken one step late) and the checker (no
// cut at all) agree.
#include "bitops.hpp"
#include "engine.hpp"

namespace pgx {

constexpr int kGX2 = 108;     // 2 X
constexpr int kGLag = 19;     // floor((X + 1/2) / 3) + 1: the X-drop test looks at the best score 19 differences earlier
constexpr int kGFastD = 15;   // differences per side of the lane-per-HSP kernel (< kGLag: its X-drop reference is 0)
constexpr int kGFastLen = 512; // read length up to which a lane's 16-bit positions are enough
constexpr int kGDmax = 1000;  // differences per side, spec
constexpr int kGFastCells = 2 * kGFastD + 3;
constexpr uint32_t kCellNone = 0xFFFFFFFFu;
static_assert(kGFastD < kGLag, "the lane-per-HSP kernel keeps no score history");

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
// cell: bits 0-15 i, 16-20 mismatches, 21-25 gap openings, 26-27 kind of the last difference (0 mismatch, 1 gap in the
// subject row, 2 gap in the query row), 28: letters matched after it
template <int DIR>
__device__ __forceinline__ bool greedy_fast(uint32_t *__restrict__ cells, const GapSeqs &s, int q0, int64_t d0, int M, int N, Side &out)
{
	auto slide = [&](int i, int j) {
		const int cap = M - i < N - j ? M - i : N - j;
		return lcp<DIR>(s, q0 + DIR * i, d0 + DIR * j, cap);
	};
	const int i0 = slide(0, 0);
	out.i = out.j = i0;
	out.s2 = 2 * i0;
	out.mism = out.gopen = 0;
	if (i0 == M || i0 == N)
		return true;
	constexpr int C = kGFastD + 1;
#pragma unroll
	for (int c = 0; c < kGFastCells; c++)
		cells[c] = kCellNone;
	cells[C] = (uint32_t)i0 | (i0 > 0 ? 1u << 28 : 0u);
	int best = 2 * i0, L = 0, U = 0;
	for (int d = 1; d <= kGFastD; d++) {
		int nl = 1 << 20, nu = -(1 << 20);
		uint32_t prev = kCellNone, cur = kCellNone; // cells outside [L, U] are kCellNone
		for (int k = L - 1; k <= U + 1; k++) {
			const uint32_t nxt = cells[C + k + 1];
			int v = -1, par = 0;
			uint32_t p = 0;
			if (cur != kCellNone) {
				v = (int)(cur & 0xFFFFu) + 1;
				p = cur;
			}
			if (prev != kCellNone && (int)(prev & 0xFFFFu) + 1 > v) {
				v = (int)(prev & 0xFFFFu) + 1;
				par = 1;
				p = prev;
			}
			if (nxt != kCellNone && (int)(nxt & 0xFFFFu) > v) {
				v = (int)(nxt & 0xFFFFu);
				par = 2;
				p = nxt;
			}
			int ii = v, jj = v - k;
			const int ub = (2 * M - k < 2 * N + k ? 2 * M - k : 2 * N + k) - 6 * d;
			uint32_t nc = kCellNone;
			if (v >= 0 && ii <= M && jj <= N && jj >= 0 && ii + jj - 6 * d >= -kGX2 && ub > best) {
				const int run = slide(ii, jj);
				ii += run;
				jj += run;
				const uint32_t pk = (p >> 26) & 3u, pslid = (p >> 28) & 1u;
				const uint32_t mism = ((p >> 16) & 31u) + (par == 0 ? 1u : 0u);
				const uint32_t gopen = ((p >> 21) & 31u) + ((par != 0 && !(pk == (uint32_t)par && !pslid)) ? 1u : 0u);
				nc = (uint32_t)ii | (mism << 16) | (gopen << 21) | ((uint32_t)par << 26) | (run > 0 ? 1u << 28 : 0u);
				const int s2 = ii + jj - 6 * d;
				if (s2 > best) {
					best = s2;
					out.i = ii;
					out.j = jj;
					out.s2 = s2;
					out.mism = (int)mism;
					out.gopen = (int)gopen;
				}
				nl = k < nl ? k : nl;
				nu = k > nu ? k : nu;
			}
			cells[C + k] = nc;
			prev = cur;
			cur = nxt;
		}
		if (nl > nu)
			return true;
		L = nl;
		U = nu;
	}
	return false; // cells still alive after kGFastD differences: the wide kernel takes this HSP
}

struct GapView {
	const uint64_t *fwd, *rc, *fwd_amb, *rc_amb;
	const uint32_t *len, *woff;
	const uint64_t *dbw, *dba;
	const uint32_t *seq_off;
};

// the initial HSP (seed stage, gapped mode): score = offset of the first base of the seed run from the HSP's start, on
// the strand of the hit
struct Anchor {
	int strand, L, qa, sa, slen;
	int64_t S0;
	GapSeqs s;
};

__device__ __forceinline__ Anchor anchor_of(const GapView &v, const pgx_hit &h)
{
	Anchor a;
	const uint32_t r = (uint32_t)h.read;
	a.L = (int)v.len[r];
	a.strand = h.sstart > h.send;
	const int bl = a.strand ? a.L - h.qend : h.qstart - 1;
	const int sl = (a.strand ? h.send : h.sstart) - 1;
	a.qa = bl + h.score;
	a.sa = sl + h.score;
	a.S0 = (int64_t)v.seq_off[h.subject];
	a.slen = (int)(v.seq_off[h.subject + 1] - v.seq_off[h.subject]);
	const uint32_t w0 = v.woff[r];
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

__device__ __forceinline__ void gapped_one(const GapView &v, pgx_hit *hp, uint32_t *cells, unsigned long long *big_list,
					   uint32_t *big_count, uint32_t big_cap)
{
	const pgx_hit h = *hp;
	const Anchor a = anchor_of(v, h);
	bool done = false;
	if (a.L <= kGFastLen) {
		Side l, r;
		done = greedy_fast<-1>(cells, a.s, a.qa - 1, a.S0 + a.sa - 1, a.qa, a.sa, l);
		if (done)
			done = greedy_fast<+1>(cells, a.s, a.qa, a.S0 + a.sa, a.L - a.qa, a.slen - a.sa, r);
		if (done)
			write_gapped(hp, h, a, l, r);
	}
	if (!done) {
		const uint32_t slot = atomicAdd(big_count, 1u);
		if (slot < big_cap)
			big_list[slot] = (unsigned long long)(uintptr_t)hp;
	}
}

constexpr int kGWaves = 4;

// main table: hits of read r are the read_cnt[r] records from read_start[r] (reads whose hits went to the overflow table
// carry kFragmented); a wavefront takes 64 reads at a time and deals their hits to its lanes
__global__ __launch_bounds__(64 * kGWaves) void k_gapped_fast(GapView v, pgx_hit *__restrict__ table, unsigned long long table_cap,
							       const uint32_t *__restrict__ read_start,
							       const uint32_t *__restrict__ read_cnt, uint32_t n_reads,
							       unsigned long long *__restrict__ big_list, uint32_t *__restrict__ big_count,
							       uint32_t big_cap)
{
	__shared__ uint32_t s_cells[kGWaves][64][kGFastCells];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	uint32_t *cells = s_cells[wave][lane];
	for (uint32_t rb = (blockIdx.x * kGWaves + wave) * 64u; rb < n_reads; rb += gridDim.x * kGWaves * 64u) {
		const uint32_t r = rb + lane;
		uint32_t cnt = 0, st = 0;
		if (r < n_reads) {
			st = read_start[r];
			cnt = st == kFragmented ? 0u : read_cnt[r];
			if ((unsigned long long)st + cnt > table_cap)
				cnt = 0; // the table was too small for this read: the host repeats the step with a larger one
		}
		uint32_t incl = cnt;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t = __shfl_up(incl, d);
			if (lane >= d)
				incl += t;
		}
		const uint32_t excl = incl - cnt, T = __shfl(incl, 63);
		for (uint32_t it = 0; it < T; it += 64) {
			const uint32_t item = it + lane;
			const bool active = item < T;
			const uint32_t key = active ? item : T - 1;
			int o = 0;
#pragma unroll
			for (int step = 32; step >= 1; step >>= 1) {
				const int cand = o + step;
				const uint32_t e = __shfl(excl, cand & 63);
				if (cand < 64 && e <= key)
					o = cand;
			}
			const uint32_t base = __shfl(st, o), ex = __shfl(excl, o);
			if (active)
				gapped_one(v, table + base + (key - ex), cells, big_list, big_count, big_cap);
		}
	}
}

// overflow table: flat
__global__ __launch_bounds__(64 * kGWaves) void k_gapped_flat(GapView v, pgx_hit *__restrict__ table, const unsigned long long *__restrict__ count,
							       unsigned long long cap, unsigned long long *__restrict__ big_list,
							       uint32_t *__restrict__ big_count, uint32_t big_cap)
{
	__shared__ uint32_t s_cells[kGWaves][64][kGFastCells];
	uint32_t *cells = s_cells[threadIdx.x >> 6][threadIdx.x & 63];
	const unsigned long long n = *count < cap ? *count : cap;
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
		gapped_one(v, table + i, cells, big_list, big_count, big_cap);
}

// ------------------------------------------------------------------------------------------ one wavefront per HSP
// cell: x = i, y = mismatches | gap openings << 12 | kind << 24 | matched-after << 26
constexpr int kBigCells = 2 * kGDmax + 3;
struct BigLds {
	uint2 row[2][kBigCells];
	int ring[kGLag + 1]; // best score with at most d differences, for the last kGLag + 1 values of d
};

__device__ __forceinline__ void lds_sync()
{
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
}

template <int DIR> __device__ void greedy_big(BigLds *lds, const GapSeqs &s, int q0, int64_t d0, int M, int N, Side &out)
{
	const int lane = threadIdx.x & 63;
	auto slide = [&](int i, int j) {
		const int cap = M - i < N - j ? M - i : N - j;
		return lcp<DIR>(s, q0 + DIR * i, d0 + DIR * j, cap);
	};
	const int i0 = slide(0, 0);
	out.i = out.j = i0;
	out.s2 = 2 * i0;
	out.mism = out.gopen = 0;
	if (i0 == M || i0 == N)
		return;
	constexpr int C = kGDmax + 1;
	int cur_row = 0;
	if (lane == 0) {
		lds->row[0][C] = make_uint2((uint32_t)i0, i0 > 0 ? 1u << 26 : 0u);
		lds->ring[0] = 2 * i0;
	}
	lds_sync();
	int best = 2 * i0, L = 0, U = 0;
	for (int d = 1; d <= kGDmax; d++) {
		const int tcmp = d >= kGLag ? lds->ring[(d - kGLag) % (kGLag + 1)] : 0;
		const uint2 *pa = lds->row[cur_row];
		uint2 *pb = lds->row[cur_row ^ 1];
		int nl = 1 << 20, nu = -(1 << 20);
		for (int kb = L - 1; kb <= U + 1; kb += 64) {
			const int k = kb + lane;
			const bool in = k <= U + 1;
			// rows are read through [L, U] only, and every cell of [L - 1, U + 1] is written each round (dead ones
			// as kCellNone), so no stale cell of an earlier round is ever read
			auto old = [&](int kk, uint2 &c) {
				bool ok = kk >= L && kk <= U;
				if (ok) {
					c = pa[C + kk];
					ok = c.x != kCellNone;
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
				pb[C + k] = alive ? make_uint2((uint32_t)ii, stats) : make_uint2(kCellNone, 0u);
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
			return;
		cur_row ^= 1;
		L = nl;
		U = nu;
	}
}

__global__ __launch_bounds__(64) void k_gapped_big(GapView v, const unsigned long long *__restrict__ list, const uint32_t *__restrict__ count,
						    uint32_t cap)
{
	__shared__ BigLds lds;
	const uint32_t n = *count < cap ? *count : cap;
	for (uint32_t idx = blockIdx.x; idx < n; idx += gridDim.x) {
		pgx_hit *hp = reinterpret_cast<pgx_hit *>((uintptr_t)list[idx]);
		const pgx_hit h = *hp;
		const Anchor a = anchor_of(v, h);
		Side l, r;
		greedy_big<-1>(&lds, a.s, a.qa - 1, a.S0 + a.sa - 1, a.qa, a.sa, l);
		lds_sync();
		greedy_big<+1>(&lds, a.s, a.qa, a.S0 + a.sa, a.L - a.qa, a.slen - a.sa, r);
		lds_sync();
		if ((threadIdx.x & 63) == 0)
			write_gapped(hp, h, a, l, r);
	}
}

int gapped_stage(const DbView &dv, const ReadsView &rv, pgx_hit *main_table, const uint32_t *read_start, const uint32_t *read_cnt,
		 pgx_hit *ovf_table, const unsigned long long *ovf_count, unsigned long long ovf_cap, bool long_reads,
		 unsigned long long hit_cap, GappedWork &gw, hipStream_t stream)
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
	// HSPs the lane-per-HSP kernel passes on: a handful for sequencing reads, every one for long queries
	const unsigned long long want = long_reads ? hit_cap + ovf_cap : (1ull << 20);
	const uint32_t big_cap = (uint32_t)std::min<unsigned long long>(want, 0xFFFFFFF0ull);
	PGX_TRY(gw.big_list.ensure(big_cap));
	PGX_TRY(gw.big_count.ensure(1));
	const uint32_t cap = (uint32_t)std::min<size_t>(gw.big_list.n, 0xFFFFFFF0ull);
	PGX_HIP(hipMemsetAsync(gw.big_count.data(), 0, sizeof(uint32_t), stream));
	const uint32_t n = rv.n;
	const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)n + 64 * kGWaves - 1) / (64 * kGWaves), 256ull * 16);
	hipLaunchKernelGGL(k_gapped_fast, dim3(grid ? grid : 1), dim3(64 * kGWaves), 0, stream, v, main_table, hit_cap, read_start, read_cnt, n,
			   gw.big_list.data(), gw.big_count.data(), cap);
	hipLaunchKernelGGL(k_gapped_flat, dim3(256), dim3(64 * kGWaves), 0, stream, v, ovf_table, ovf_count, ovf_cap, gw.big_list.data(),
			   gw.big_count.data(), cap);
	hipLaunchKernelGGL(k_gapped_big, dim3(long_reads ? 256 * 8 : 256), dim3(64), 0, stream, v, gw.big_list.data(), gw.big_count.data(), cap);
	PGX_HIP(hipGetLastError());
	return 0;
}

} // namespace pgx
