// Classify kernels, BLAST mode (spec pgx-blastn v2): the seed + ungapped stage, the grouping of a search's
// hits by read, and the per-read ordering + consensus.  The gapped stage between them is gapped.hip, query masking dust.hip.
//
// Replaces the arithmetic of `blastn -query F -db DB -outfmt 6` (reference README.md:96,
// Scripts/run_multi_blastn.pl:56, Scripts/submit_MPI-blast.job:24) and the per-read arg-max of
// Consensus/Consensus_BLAST_SOAP_RDP-1.1.pl:141-234.
//
// k_seed_extend<AMB, NW, LISTED, DUST>
//                 one wavefront per TWO reads of up to NW x 64 bases (NW = 0: one read of any length).  Lanes first act as
//                 PROBES: the 16-mers at stride 13 of both strands are looked up in the direct-address bucket table (one
//                 dependent gather each).  The posting counts are prefix-summed across the wave and the postings -- 12-byte
//                 records that carry the 13 database bases left and the 12 right of the 16-mer -- are dealt to the lanes
//                 128 at a time: a posting whose flanks say the previous probe reports the same run, or that the exact run
//                 is shorter than 28, is dropped before anything else is fetched.  Survivors get their block record
//                 (subject, bounds), one candidate per (read, strand, diagonal) stays (an LDS set), and a queue hands 64
//                 CANDIDATES at a time to the lanes: mismatch flags of the whole diagonal in registers (packed read XOR
//                 funnel-shifted database window), DUST window bits (S3d), every exact run of >= 28, the X-drop walks of
//                 the ungapped extension, and for the gapped stage the anchor and the level estimate B0.  Hits are staged
//                 in LDS per read and leave contiguously into the wave's chunk of the table (one atomic per chunk).
//                 search_pipeline (below) measures it stage by stage in PGX_STAGE_PROBES builds.
// k_scan_*, k_scatter_hits, k_merge_pieces
//                 per-read offsets of the table; only fragmented reads are moved.
// k_sort_consensus<32 / 64>
//                 one wavefront per two reads of up to 32 hits, then per read of 33-64: one hit per lane, keys in LDS, the
//                 best score of a hit's subject, spec S3c (duplicate alignments of one subject), the rank by counting, the
//                 ordered rows written out, then the (rank,name) agreement with the read's RDP assignment per hit and the
//                 order-dependent arg-max of the Perl with its string comparisons as two reductions (a chain walk when
//                 lineages differ in depth).  Reads with more hits: bigreads.hip.
#include <algorithm>
#include <chrono>

#include "bitops.hpp"
#include "consensus_core.hpp"
#include "engine.hpp"

namespace pgx {

// ------------------------------------------------------------------------------------------ diagonal masks
struct Diag {
	const uint64_t *rw, *ra;   // read words / spaced ambiguity flags of the strand (ra may be null)
	const uint64_t *dbw, *dba; // database words / flags (dba may be null)
	int64_t dstart;            // database position aligned with read position 0
	int lo, hi;                // read positions [lo, hi) lie inside the subject
};

// spaced mismatch flags of read bases [32w, 32w+32); bases outside [lo, hi) are flagged
template <bool AMB> __device__ __forceinline__ uint64_t mmw(const Diag &D, int w)
{
	int b0 = 32 * w;
	if (b0 + 32 <= D.lo || b0 >= D.hi)
		return kEven;
	uint64_t x = D.rw[w] ^ window64(D.dbw, D.dstart + b0);
	uint64_t m = (x | (x >> 1)) & kEven;
	if (AMB) {
		if (D.ra)
			m |= D.ra[w];
		if (D.dba)
			m |= window64(D.dba, D.dstart + b0);
	}
	m |= kEven & ~spaced_range(D.lo - b0, D.hi - b0);
	return m;
}

// even bits of x packed into the low 32 bits
__device__ __forceinline__ uint64_t compress_even(uint64_t x)
{
	x = (x | (x >> 1)) & 0x3333333333333333ull;
	x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
	x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
	x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
	x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
	return x;
}

// Mismatch flags of one diagonal, two forms with the same interface:
//   DenseMask  reads of <= 64 NW bases (NW = 3 or 5): the whole diagonal as NW x 64 one-bit flags in registers, built
//              once (5 read words + 5 database windows for a 150-bp read); every scan is bit logic
//   LazyMask   any length: 32-base words rebuilt on demand (loads served by L1/L2)
template <bool AMB, int NW> struct DenseMask {
	static constexpr bool kHasWindows = true;
	static constexpr int kBits = 64 * NW; // bases the flags cover
	uint64_t m[NW];
	int lo, hi;
	// register arrays are only ever indexed by unrolled loop counters; a run-time word index goes through pick()
	__device__ __forceinline__ static uint64_t pick(const uint64_t (&a)[NW], int wi)
	{
		uint64_t v = 0;
#pragma unroll
		for (int k = 0; k < NW; k++)
			v = wi == k ? a[k] : v;
		return v;
	}
	__device__ __forceinline__ static uint64_t below(int n) { return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1)); }
	__device__ __forceinline__ static uint64_t from(int n) { return n <= 0 ? ~0ull : (n >= 64 ? 0ull : (~0ull << n)); } // bits at positions >= n
	// The database and read arrays carry 24 spare words on either side, so the 2 NW read words and 2 NW + 1
	// database words of the diagonal are fetched without per-word range checks (all loads independent); the
	// subject and read bounds are applied afterwards on the dense flags.
	__device__ __forceinline__ void build(const Diag &D)
	{
		lo = D.lo;
		hi = D.hi;
		const int64_t wi = D.dstart >> 5;
		const int sh = (int)(D.dstart & 31) * 2;
		uint64_t dbv[2 * NW + 1], rwv[2 * NW];
#pragma unroll
		for (int w = 0; w < 2 * NW + 1; w++)
			dbv[w] = D.dbw[wi + w];
#pragma unroll
		for (int w = 0; w < 2 * NW; w++)
			rwv[w] = D.rw[w];
		uint64_t d[2 * NW];
#pragma unroll
		for (int w = 0; w < 2 * NW; w++) {
			const uint64_t win = sh ? (dbv[w] >> sh) | (dbv[w + 1] << (64 - sh)) : dbv[w];
			const uint64_t x = rwv[w] ^ win;
			uint64_t mm = (x | (x >> 1)) & kEven;
			if (AMB) {
				if (D.ra)
					mm |= D.ra[w];
				if (D.dba)
					mm |= window64(D.dba, D.dstart + 32 * w);
			}
			d[w] = compress_even(mm);
		}
		// flag everything outside [lo, hi)
#pragma unroll
		for (int k = 0; k < NW; k++)
			m[k] = (d[2 * k] | (d[2 * k + 1] << 32)) | below(lo - 64 * k) | from(hi - 64 * k);
	}
	// w[i] set iff positions i .. i+27 all match (a 28-window of matches starts at i): log-step doubling
	// 2,4,8,16 then 16+8 and 24+4; every exact run >= 28 shows up as a run of set bits from its start
	struct Win {
		uint64_t w[NW];
	};
	template <int K> __device__ __forceinline__ static void and_shr(const uint64_t (&a)[NW], const uint64_t (&src)[NW], uint64_t (&o)[NW])
	{
		// o = a & (src >> K) across the NW words
#pragma unroll
		for (int k = 0; k < NW; k++) {
			const uint64_t up = k + 1 < NW ? src[k + 1] : 0ull;
			o[k] = a[k] & ((src[k] >> K) | (up << (64 - K)));
		}
	}
	__device__ __forceinline__ Win seed_windows() const
	{
		uint64_t r[NW], a2[NW], a4[NW], a8[NW], b16[NW], c24[NW];
#pragma unroll
		for (int k = 0; k < NW; k++)
			r[k] = ~m[k];
		and_shr<1>(r, r, a2);
		and_shr<2>(a2, a2, a4);
		and_shr<4>(a4, a4, a8);
		and_shr<8>(a8, a8, b16);
		and_shr<16>(b16, a8, c24);
		Win w;
		and_shr<24>(c24, a4, w.w);
		return w;
	}
	// is any window bit set below position n?
	__device__ __forceinline__ static bool win_any_below(const Win &w, int n)
	{
		uint64_t any = 0;
#pragma unroll
		for (int k = 0; k < NW; k++)
			any |= w.w[k] & below(n - 64 * k);
		return any != 0;
	}
	// lowest set bit of a[] at position >= n, or kBits
	__device__ __forceinline__ static int first_set_ge(const uint64_t (&a)[NW], int n)
	{
		int r = kBits;
#pragma unroll
		for (int k = NW - 1; k >= 0; k--) {
			const uint64_t v = a[k] & from(n - 64 * k);
			r = v ? 64 * k + __ffsll((unsigned long long)v) - 1 : r;
		}
		return r;
	}
	__device__ __forceinline__ static int win_first_ge(const Win &w, int n) { return first_set_ge(w.w, n); }
	// is position p (inside [lo, hi)) flagged?
	__device__ __forceinline__ bool flagged(int p) const { return (pick(m, p >> 6) >> (p & 63)) & 1ull; }
	// flagged positions of [a, b) that lie inside [lo, hi)
	__device__ __forceinline__ int count_range(int a, int b) const
	{
		a = a > lo ? a : lo;
		b = b < hi ? b : hi;
		int c = 0;
#pragma unroll
		for (int k = 0; k < NW; k++)
			c += __popcll(m[k] & from(a - 64 * k) & below(b - 64 * k));
		return c;
	}
	// smallest flagged position >= pos, or hi
	__device__ __forceinline__ int first_ge(int pos) const
	{
		if (pos >= hi)
			return hi;
		const int r = first_set_ge(m, pos);
		return r < hi ? r : hi;
	}
	// largest flagged position < pos, or lo-1
	__device__ __forceinline__ int last_lt(int pos) const
	{
		if (pos <= lo)
			return lo - 1;
		int r = -1;
#pragma unroll
		for (int k = 0; k < NW; k++) {
			const uint64_t v = m[k] & below(pos - 64 * k);
			r = v ? 64 * k + 63 - __clzll((long long)v) : r;
		}
		return r >= lo ? r : lo - 1;
	}
	// Cursors over the flags for the X-drop walks: the current 64-flag word is consumed bit by bit
	// (one count-zeros and one clear per mismatch); a new word is picked only when it runs empty.
	// (32-bit half words -- count-zeros, single-bit mask and clear are one instruction each there, three to four on 64 bits --
	// were tried in round 4 and ran SLOWER, 32.9 against 31.8 ms: the register allocation of this kernel decides, not its
	// instruction count.)
	struct Fwd {
		uint64_t w;
		int wi;
	};
	// flagged positions > k, ascending (k in [-1, kBits - 1])
	__device__ __forceinline__ Fwd fwd_from(int k) const
	{
		const int pos = k + 1;
		Fwd c;
		c.wi = pos >> 6;
		c.w = pick(m, c.wi) & (~0ull << (pos & 63));
		return c;
	}
	__device__ __forceinline__ int fwd_next(Fwd &c) const
	{
		while (!c.w) {
			c.wi++;
			if (c.wi >= NW)
				return hi;
			c.w = pick(m, c.wi);
		}
		const int r = c.wi * 64 + __ffsll((unsigned long long)c.w) - 1;
		c.w &= c.w - 1;
		return r < hi ? r : hi;
	}
	struct Bwd {
		uint64_t w;
		int wi;
	};
	// flagged positions < k, descending (k in [0, kBits])
	__device__ __forceinline__ Bwd bwd_from(int k) const
	{
		Bwd c;
		if (k <= 0) {
			c.w = 0;
			c.wi = 0;
			return c;
		}
		const int q = k - 1, bb = q & 63;
		c.wi = q >> 6;
		c.w = pick(m, c.wi) & (bb == 63 ? ~0ull : ((2ull << bb) - 1));
		return c;
	}
	__device__ __forceinline__ int bwd_next(Bwd &c) const
	{
		while (!c.w) {
			if (c.wi == 0)
				return lo - 1;
			c.wi--;
			c.w = pick(m, c.wi);
		}
		const int bb = 63 - __clzll((long long)c.w);
		c.w ^= 1ull << bb;
		const int r = c.wi * 64 + bb;
		return r >= lo ? r : lo - 1;
	}
};

template <bool AMB> struct LazyMask {
	static constexpr bool kHasWindows = false;
	struct Win {
	};
	Diag D;
	int lo, hi;
	__device__ __forceinline__ void build(const Diag &d)
	{
		D = d;
		lo = d.lo;
		hi = d.hi;
	}
	__device__ __forceinline__ int count_range(int, int) const { return 15; } // (long reads go to the wide gapped kernel anyway)
	__device__ __forceinline__ bool flagged(int p) const { return first_ge(p) == p; }
	__device__ __forceinline__ int first_ge(int pos) const
	{
		if (pos >= hi)
			return hi;
		int w = pos >> 5;
		uint64_t m = mmw<AMB>(D, w) & ~((1ull << (2 * (pos & 31))) - 1);
		while (!m) {
			w++;
			if (32 * w >= hi)
				return hi;
			m = mmw<AMB>(D, w);
		}
		int r = 32 * w + ((__ffsll((unsigned long long)m) - 1) >> 1);
		return r < hi ? r : hi;
	}
	__device__ __forceinline__ int last_lt(int pos) const
	{
		if (pos <= lo)
			return lo - 1;
		int q = pos - 1, w = q >> 5, bit = q & 31;
		uint64_t keep = bit == 31 ? ~0ull : ((1ull << (2 * bit + 2)) - 1);
		uint64_t m = mmw<AMB>(D, w) & keep;
		while (!m) {
			if (32 * w <= lo)
				return lo - 1;
			w--;
			m = mmw<AMB>(D, w);
		}
		int r = 32 * w + ((63 - __clzll((long long)m)) >> 1);
		return r >= lo ? r : lo - 1;
	}
	struct Fwd {
		int k;
	};
	__device__ __forceinline__ Fwd fwd_from(int k) const { return Fwd{ k }; }
	__device__ __forceinline__ int fwd_next(Fwd &c) const { return c.k = first_ge(c.k + 1); }
	struct Bwd {
		int k;
	};
	__device__ __forceinline__ Bwd bwd_from(int k) const { return Bwd{ k }; }
	__device__ __forceinline__ int bwd_next(Bwd &c) const { return c.k = last_lt(c.k); }
};

constexpr int kStage = 128; // hits staged in LDS per wave between flushes
constexpr int kDeal = 2;     // postings dealt per lane per round (independent loads in flight)
constexpr int kQueue = 64 * (kDeal + 1); // candidate queue per wave (filled 64*kDeal at a time, drained at >= 64)
constexpr int kWavesPerBlock = 4;

// the ordered hit table is written once and read by a later call: streamed past the caches (the database words,
// block records and subject records are what should stay in them)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
union HitWords {
	pgx_hit h;
	u32x4 v[2];
};
__device__ __forceinline__ void store_hit_stream(pgx_hit *p, const pgx_hit &h)
{
	HitWords w;
	w.h = h;
	u32x4 *d = reinterpret_cast<u32x4 *>(p);
	__builtin_nontemporal_store(w.v[0], d);
	__builtin_nontemporal_store(w.v[1], d + 1);
}


constexpr int kDiagSlots = 128;       // open-addressed set of (diagonal, strand) keys per read
constexpr int kDiagProbes = 8;
constexpr uint32_t kNoDiag = 0xFFFFFFFFu;

struct WaveLds {
	// one read per wave: kStage full records.  Two reads per wave (reads of <= 192 bases): kStage 16-byte
	// records per read in the same bytes: subject, sstart, qstart | qend << 16 | minus << 31, score | mismatch << 16
	union {
		pgx_hit hit[kStage];
		uint4 chit[2 * kStage];
	};
	// queued candidates: posting, strand << 31 | tested << 30 | read slot << 29 | owns-diagonal << 28 | qpos, and the
	// subject found while filtering
	uint32_t qp[kQueue], qmeta[kQueue], qsubj[kQueue], qs0[kQueue], qs1[kQueue];
	uint32_t diag[2][2][kDiagSlots / 2]; // per read of the pair and strand: diagonals already owned by a queued candidate
	uint64_t uwin[2][2][8];              // per read of the pair and strand: S3d window bits (positions whose 28 bases touch no masked base)
	unsigned int n[2];                   // staged hits per read slot
	unsigned int direct[2];              // hits that found the stage full and went straight to the overflow table
};

struct OutView {
	pgx_hit *main, *ovf;
	uint8_t *main_key, *ovf_key; // gapped mode: per slot the work estimate of the gapped stage (mismatches left | right << 4)
	uint8_t *main_reg;           // gapped mode: per slot of the main table the REGION of the database the anchor lies in (position >> reg_shift)
	int reg_shift;
	unsigned long long main_cap, ovf_cap;
	// [0] main-table slots reserved (chunks), [1] probes, [2] postings, [3] seed runs, [4] overflow hits,
	// [5] hits stored in the main table, [6] candidates that survive the duplicate filter
	unsigned long long *counters;
};
constexpr unsigned int kChunk = 2048; // main-table slots a wave reserves with one atomic (a single hot counter
				      // sustains only ~90 M atomics/s: one per read would cost more than the search)

__device__ __forceinline__ uint4 pack_hit(const pgx_hit &h)
{
	const uint32_t minus = h.sstart > h.send;
	return make_uint4((uint32_t)h.subject, (uint32_t)h.sstart, (uint32_t)h.qstart | ((uint32_t)h.qend << 16) | (minus << 31),
			  (uint32_t)h.score | ((uint32_t)h.mismatch << 16) | ((uint32_t)h.gapopen << 24));
}

__device__ __forceinline__ pgx_hit unpack_hit(const uint4 c, uint32_t read)
{
	pgx_hit h;
	h.read = (int32_t)read;
	h.subject = (int32_t)c.x;
	h.sstart = (int32_t)c.y;
	h.qstart = (int32_t)(c.z & 0xFFFFu);
	h.qend = (int32_t)((c.z >> 16) & 0x7FFFu);
	const int span = h.qend - h.qstart;
	h.send = (c.z >> 31) ? h.sstart - span : h.sstart + span;
	h.score = (int32_t)(c.w & 0xFFFFu);
	h.mismatch = (uint16_t)((c.w >> 16) & 0xFFu); // (ungapped hits of <= 512-base reads: at most 170 mismatches)
	h.gapopen = (uint16_t)(c.w >> 24);
	return h;
}

// Spec v2: what the seed stage hands to the gapped stage is not a hit but a SEED RECORD in the same 32 bytes, holding
// what a round of the gapped kernel needs without a second look-up: read, subject, qstart = word offset of the read,
// qend = anchor (read position on the hit's strand), sstart = database position of the anchor, send = strand | B0 of
// the left side << 1 | B0 of the right side << 12 (11 bits each, see process_candidate), score = read length,
// mismatch / gapopen = levels the gapped stage will run left / right of the anchor (capped at 15).
// Staged 16-byte form (two short reads per wavefront): subject, database position, anchor | B0 left << 15 | strand << 31,
// estimates | B0 right << 8.
__device__ __forceinline__ uint4 pack_seed(const pgx_hit &h)
{
	const uint32_t sd = (uint32_t)h.send; // strand | B0 left << 1 | B0 right << 12
	return make_uint4((uint32_t)h.subject, (uint32_t)h.sstart, (uint32_t)h.qend | (((sd >> 1) & 0x7FFu) << 15) | (sd << 31),
			  (uint32_t)h.mismatch | ((uint32_t)h.gapopen << 4) | ((sd >> 12) << 8));
}

__device__ __forceinline__ pgx_hit unpack_seed(const uint4 c, uint32_t read, uint32_t woff, int L)
{
	pgx_hit h;
	h.read = (int32_t)read;
	h.subject = (int32_t)c.x;
	h.qstart = (int32_t)woff;
	h.qend = (int32_t)(c.z & 0x7FFFu);
	h.sstart = (int32_t)c.y;
	h.send = (int32_t)((c.z >> 31) | (((c.z >> 15) & 0x7FFu) << 1) | ((c.w >> 8) << 12));
	h.score = L;
	h.mismatch = (uint16_t)(c.w & 15u);
	h.gapopen = (uint16_t)((c.w >> 4) & 15u);
	return h;
}

__device__ __forceinline__ uint8_t seed_key(const pgx_hit &h) { return (uint8_t)((h.mismatch & 15u) | ((h.gapopen & 15u) << 4)); }

// PACKED: two reads share the wavefront, read slot `rs` stages 16-byte records
template <bool PACKED>
__device__ __forceinline__ void emit_hit(WaveLds *st, int rs, const OutView &ov, const pgx_hit &h, bool gapped)
{
	unsigned int slot = atomicAdd(&st->n[rs], 1u);
	if (slot < (unsigned)kStage) {
		if (PACKED)
			st->chit[rs * kStage + slot] = gapped ? pack_seed(h) : pack_hit(h);
		else
			st->hit[slot] = h;
	} else {
		atomicAdd(&st->direct[rs], 1u);
		unsigned long long g = atomicAdd(&ov.counters[4], 1ull);
		if (g < ov.ovf_cap) {
			ov.ovf[g] = h;
			if (gapped)
				ov.ovf_key[g] = seed_key(h);
		}
	}
}

// S3d on dense window bits in memory: is any of the positions a .. b a window free of masked bases?
__device__ __forceinline__ bool uwin_any(const uint64_t *uw, int a, int b)
{
	if (b < a)
		return false;
	for (int w = a >> 6; w <= (b >> 6); w++) {
		uint64_t v = uw[w];
		if (w == (a >> 6))
			v &= ~0ull << (a & 63);
		if (w == (b >> 6) && (b & 63) != 63)
			v &= (2ull << (b & 63)) - 1ull;
		if (v)
			return true;
	}
	return false;
}

// One candidate = one (strand, probe position, posting).  `tested` says the index already proved that
// this probe is the left-most one of its exact run (test 1 below).
template <bool AMB, class Mask, class Emit>
__device__ __forceinline__ void process_candidate(const DbView &db, const uint64_t *rw, const uint64_t *ra, const uint64_t *uw, int L,
						   uint32_t read, uint32_t woff, int strand, int qp, uint32_t p, bool tested, bool claimed, uint32_t s,
						   uint32_t s_start, uint32_t s_end, Emit &emit, unsigned long long &n_runs)
{
	if (p + (uint32_t)kSeedK > s_end)
		return; // seed straddles two subjects
	Diag D;
	D.rw = rw;
	D.ra = ra;
	D.dbw = db.words;
	D.dba = db.amb;

	D.dstart = (int64_t)p - qp;
	if (AMB && db.amb && db.amb_blk) {
		// most diagonals touch no block with an ambiguity letter: they skip the flag words of the database
		const int64_t a = D.dstart > 0 ? D.dstart : 0, b = D.dstart + L - 1;
		uint32_t any = 0;
		for (int64_t blk = a >> kBlkShift; blk <= (b >> kBlkShift); blk++)
			any |= (db.amb_blk[blk >> 5] >> (blk & 31)) & 1u;
		if (!any)
			D.dba = nullptr;
	}
	int64_t lo64 = (int64_t)s_start - D.dstart, hi64 = (int64_t)s_end - D.dstart;
	D.lo = lo64 > 0 ? (int)lo64 : 0;
	D.hi = hi64 < L ? (int)hi64 : L;
	Mask M;
	M.build(D);
	if (PGX_DBG_STOP(db) == 4) {
		n_runs += (unsigned long long)(M.first_ge(qp) & 1);
		return;
	}

	int pos, run_start = 0; // (run_start is set by the branch that owns it: whole diagonals only exist for masks with windows)
	typename Mask::Win W;
	bool whole = false;
	if constexpr (Mask::kHasWindows)
		whole = claimed;
	if (whole) {
		// this candidate owns its diagonal (claimed in the dealing stage, see k_seed_extend): every >= 28 run of
		// the diagonal is taken from the window bitmap, whichever probe got here first
		if constexpr (Mask::kHasWindows) {
			W = M.seed_windows();
			if (uw) {
#pragma unroll
				for (int k = 0; k < Mask::kBits / 64; k++)
					W.w[k] &= uw[k]; // S3d: windows that touch a masked base of the read seed nothing
			}
			run_start = Mask::win_first_ge(W, 0);
		}
		if (run_start >= D.hi)
			return; // no exact run of 28 on this diagonal (the dealing-stage filter only bounds it)
		if (uw)
			run_start = M.last_lt(run_start) + 1; // (the first clean window may lie inside its run)
		n_runs++;
		if (PGX_DBG_STOP(db) == 5 || PGX_DBG_STOP(db) == 6)
			return;
	} else {
		// (1) only the left-most probe inside an exact run reports that run
		const int lm = M.last_lt(qp);
		if (!tested && qp >= kProbeStride && lm < qp - kProbeStride)
			return;
		// (2) the probe 16-mer itself must match (bucket collisions, ambiguity, boundaries)
		const int re = M.first_ge(qp);
		if (re < qp + kSeedK)
			return;
		run_start = lm + 1;
		if (re - run_start < kWord)
			return;
		// (3) only the first seed run of a diagonal generates the diagonal's HSPs; S3d: a run seeds only through a window
		// that touches no masked base of the read
		if constexpr (Mask::kHasWindows) {
			W = M.seed_windows();
			if (uw) {
#pragma unroll
				for (int k = 0; k < Mask::kBits / 64; k++)
					W.w[k] &= uw[k];
				if (Mask::win_first_ge(W, run_start) > re - kWord)
					return;
			}
			n_runs++; // a seed run reached through its left-most probe
			if (PGX_DBG_STOP(db) == 5)
				return;
			if (Mask::win_any_below(W, run_start))
				return;
		} else {
			if (uw && !uwin_any(uw, run_start, re - kWord))
				return;
			n_runs++;
			if (PGX_DBG_STOP(db) == 5)
				return;
			pos = D.lo;
			while (pos < run_start) {
				int m1 = M.first_ge(pos);
				if (m1 - pos >= kWord && (!uw || uwin_any(uw, pos, m1 - kWord)))
					return;
				pos = m1 + 1;
			}
		}
		if (PGX_DBG_STOP(db) == 6)
			return;
	}
	// (4) spec S3: seeds left to right, X-drop extension on the mismatch flags
	int covered = D.lo;
	pos = run_start;
	while (pos < D.hi) {
		const int e = M.first_ge(pos);
		const int len = e - pos;
		bool seeds = len >= kWord && pos >= covered;
		if constexpr (!Mask::kHasWindows)
			seeds = seeds && (!uw || uwin_any(uw, pos, e - kWord));
		if (seeds) {
			int best = 0, cur = 0, bl = pos, nmm = 0, mm_best = 0;
			int k = pos - 1; // a flagged position, or lo-1
			typename Mask::Bwd cl = M.bwd_from(k);
			while (k >= D.lo) {
				cur -= 2;
				nmm++;
				if (best - cur > kXdrop)
					break;
				const int p2 = M.bwd_next(cl);
				// (selects, not branches: a skipped branch still costs the wavefront its mask bookkeeping, and some lane takes
				// each of them at most steps -- the walks are a third of this kernel)
				cur += k - 1 - p2; // the matches between the two flagged positions (0 when they are neighbours)
				const bool up = cur > best;
				best = up ? cur : best;
				bl = up ? p2 + 1 : bl;
				mm_best = up ? nmm : mm_best;
				k = p2;
			}
			int bestr = 0, br = e - 1, mmr_best = 0;
			cur = 0;
			nmm = 0;
			k = e; // a flagged position, or hi
			typename Mask::Fwd cr = M.fwd_from(k);
			while (k < D.hi) {
				cur -= 2;
				nmm++;
				if (bestr - cur > kXdrop)
					break;
				const int n2 = M.fwd_next(cr);
				cur += n2 - (k + 1);
				const bool up = cur > bestr;
				bestr = up ? cur : bestr;
				br = up ? n2 - 1 : br;
				mmr_best = up ? nmm : mmr_best;
				k = n2;
			}
			pgx_hit h;
			h.read = (int32_t)read;
			h.subject = (int32_t)s;
			h.score = len + best + bestr;
			h.mismatch = (uint16_t)(mm_best + mmr_best);
			h.gapopen = 0;
			const int64_t sl = D.dstart + bl - (int64_t)s_start + 1, sr = D.dstart + br - (int64_t)s_start + 1;
			if (!strand) {
				h.qstart = bl + 1;
				h.qend = br + 1;
				h.sstart = (int32_t)sl;
				h.send = (int32_t)sr;
			} else {
				h.qstart = L - br;
				h.qend = L - bl;
				h.sstart = (int32_t)sr;
				h.send = (int32_t)sl;
			}
			if (db.gapped) {
				// spec v2: this is an INITIAL HSP; the gapped stage (gapped.hip) extends it from its ANCHOR: the first base of
				// the run of matches that holds the last matching position at or before the HSP's middle column (S3b; an
				// extension from the middle costs the least: its work grows with the square of the differences on a side).
				// What leaves here is a seed record (see pack_seed), not a hit.
				int q = bl + (br - bl) / 2;
				while (M.flagged(q))
					q--; // (bl is a match)
				int anchor = M.last_lt(q) + 1;
				anchor = anchor > bl ? anchor : bl;
				// B0 of either side: a doubled score, 2 (letters) - 6 (mismatches), that the gapped stage is sure to reach on
				// the anchor's own diagonal with at most 18 mismatches, which it prunes with from its first cell on.  The
				// ungapped extension above scores a letter +1 / -2, i.e. +2 / -4 doubled -- the same as a matched / mismatched
				// letter there -- so its ends bl, br ARE the best prefixes unless the drop-off of 10 stopped it early (then the
				// bound is only lower, never wrong).  From it the number of LEVELS that stage will run on the side,
				// floor((2 letters of the read on the side - B0) / 5): the work estimate its rounds are ordered by.
				int b0l = 0, b0r = 0, kl = 14, kr = 14; // (estimates stop at 14: the key byte 0xFF marks a slot without a record)
				if constexpr (Mask::kHasWindows) {
					// (the walks counted the HSP's mismatches: the right part's follow from the left part's)
					const int ml = M.count_range(bl, anchor), mr = (int)h.mismatch - ml;
					b0l = ml <= 18 ? 2 * (anchor - bl) - 6 * ml : 0;
					b0r = mr <= 18 ? 2 * (br + 1 - anchor) - 6 * mr : 0;
					b0l = b0l > 0 && b0l < 2047 ? b0l : 0; // (11 bits in the record; 0 = no bound)
					b0r = b0r > 0 && b0r < 2047 ? b0r : 0;
					kl = (2 * anchor - b0l) / 5;
					kr = (2 * (L - anchor) - b0r) / 5;
					kl = kl < 14 ? kl : 14;
					kr = kr < 14 ? kr : 14;
					if (L > db.deep_from) {
						// reads the gapped stage runs with 40 differences a side (gapped.hip: greedy_rows_deep): B0 says nothing
						// beyond 18 mismatches, and half the sides of a 500-base read at 7 % hold more.  Its rounds are ordered by
						// the diagonal's own mismatch count, three levels to a step, plus the ~10 levels a side runs on past an
						// extension that the drop-off ended inside the read
						kl = (ml + (bl > 0 ? 10 : 0)) / 3;
						kr = (mr + (br + 1 < L ? 10 : 0)) / 3;
						kl = kl < 14 ? kl : 14;
						kr = kr < 14 ? kr : 14;
					}
				}
				h.qstart = (int32_t)woff;
				h.qend = anchor;
				h.sstart = (int32_t)(uint32_t)(D.dstart + anchor);
				h.send = strand | (b0l << 1) | (b0r << 12);
				h.score = L;
				h.mismatch = (uint16_t)kl;
				h.gapopen = (uint16_t)kr;
			}
			if (PGX_DBG_STOP(db) != 7)
				emit(h);
			covered = br + 1;
		}
		if constexpr (Mask::kHasWindows) {
			pos = Mask::win_first_ge(W, covered > e ? covered : e + 1); // next seed run (a run start)
			if (uw && pos < D.hi)
				pos = M.last_lt(pos) + 1; // (with S3d the first clean window may lie inside its run)
		} else
			pos = covered > e + 1 ? covered : e + 1;
	}
}

// All LDS traffic of these kernels is wave-private (each wave owns its slice), and a wave's LDS operations
// execute in issue order.  The fence therefore only has to stop the compiler from moving memory operations
// across it and to drain the wave's own LDS queue; it deliberately does NOT wait for outstanding global
// loads/stores (a workgroup-scope release would: `s_waitcnt vmcnt(0)` after every flush exposes the full
// store latency once per read).
__device__ __forceinline__ void lds_fence()
{
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
}

// Seed + extend.  Per read (one wavefront):
//   probes      lanes look the stride-13 16-mers of both strands up in the bucket table
//   deal        posting counts are prefix-summed over the wave and postings dealt to lanes 64 at a time
//   filter      with a direct-address index (bits == 32) a posting p of probe q is dropped at once when p-13 is
//               a posting of probe q-13 (binary search in the neighbour lane's list): the earlier probe
//               reports that run, and the database is never touched for ~2/3 of the postings
//   queue       survivors are compacted (ballot + popcount prefix) into an LDS queue and drained 64 at a time,
//               so the expensive diagonal work always runs with full lanes
//   output      hits are staged in LDS and leave with ONE atomic per read, contiguously (read_start/read_cnt);
//               a read that overflows the stage is marked fragmented and goes through the overflow table
// RPW = reads per wavefront.  Reads of <= 192 bases need at most 28 probes, so two of them share a wavefront:
// lanes 0-31 probe the first, lanes 32-63 the second, their postings are dealt together and their candidates
// drained together (160 candidates fill 64-lane drains far better than 80), each read staging its hits in
// its own half of the stage.  Longer reads keep the wavefront to themselves.
// LISTED: the launch works on the reads named by rd.list (one search class of a mixed batch) instead of 0 .. n-1
template <bool AMB, int NW, bool LISTED, bool DUST>
__global__ __launch_bounds__(64 * kWavesPerBlock, NW == 8 ? 3 : 4) void k_seed_extend(DbView db, ReadsView rd, OutView ov,
								      uint32_t *__restrict__ read_cnt,
								      uint32_t *__restrict__ read_start)
{
	constexpr bool DENSE = NW > 0; // NW x 64 one-bit flags per diagonal in registers (0: lazy 32-base words, any length)
	constexpr int RPW = DENSE ? 2 : 1;
	constexpr int LPR = 64 / RPW;                     // probe lanes per read
	constexpr unsigned int SLOT_CAP = kStage;         // staged hits per read (16-byte records when two reads share)
	__shared__ WaveLds s_lds[kWavesPerBlock];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int my_rs = lane / LPR, my_li = lane % LPR; // read slot / probe lane of this lane
	WaveLds *st = &s_lds[wave];
	unsigned long long n_probe = 0, n_post = 0, n_runs = 0, n_main = 0, n_surv = 0;
	unsigned long long chunk_base = 0;
	unsigned int chunk_used = kChunk; // forces a reservation at the first flush
	const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;

	for (uint32_t rbase = (blockIdx.x * kWavesPerBlock + wave) * RPW; rbase < rd.n; rbase += gridDim.x * kWavesPerBlock * RPW) {
		// wave-uniform description of the (up to) two reads
		const bool hasB = RPW == 2 && rbase + 1 < rd.n;
		const uint32_t rA = LISTED ? rd.list[rbase] : rbase, rB = LISTED ? (hasB ? rd.list[rbase + 1] : 0u) : rbase + 1;
		const int LA = (int)rd.len[rA], LB = hasB ? (int)rd.len[rB] : 0;
		const uint32_t wA = rd.woff[rA], wB = hasB ? rd.woff[rB] : 0u;
		// this lane's own read (for the probe phase)
		const int L = my_rs ? LB : LA;
		const uint32_t w0 = my_rs ? wB : wA;
		const int nps = L >= kSeedK ? (L - kSeedK) / kProbeStride + 1 : 0;
		const int P = 2 * nps; // probes of this lane's read
		// longest probe list among the reads of the wave (one chunk of LPR lanes per round)
		const int PA = LA >= kSeedK ? 2 * ((LA - kSeedK) / kProbeStride + 1) : 0;
		const int PB = LB >= kSeedK ? 2 * ((LB - kSeedK) / kProbeStride + 1) : 0;
		const int Pmax = PA > PB ? PA : PB;
		unsigned int emitted[2] = { 0, 0 }, q_n = 0;
		bool frag[2] = { false, false };
		if (lane == 0) {
			st->n[0] = st->n[1] = 0;
			st->direct[0] = st->direct[1] = 0;
		}
		if (RPW == 2) {
#pragma unroll
			for (int k = 0; k < 2 * kDiagSlots / 64; k++)
				(&st->diag[0][0][0])[k * 64 + lane] = kNoDiag;
		}
		// S3d: only reads with a masked base (a few per cent of random reads: a homopolymer of seven) carry window bits
		// (DUST = the launch works on the class of reads that have one; a listed pair may still mix in its last wave)
		const bool dustA = DUST && rd.dustwin_f && rd.dust_any[rA] != 0, dustB = DUST && rd.dustwin_f && hasB && rd.dust_any[rB] != 0;
		if (DUST && DENSE && (dustA || dustB) && lane < 32) {
			const int us = lane >> 4, ustr = (lane >> 3) & 1, uk = lane & 7;
			const int uL = us ? LB : LA;
			uint64_t v = 0;
			if ((us ? dustB : dustA) && uk < (uL + 63) / 64)
				v = (ustr ? rd.dustwin_r : rd.dustwin_f)[(us ? wB : wA) + uk];
			st->uwin[us][ustr][uk] = v;
		}
		lds_fence();

		// spill one read's staged hits to the overflow table in the middle of the work: it becomes fragmented
		auto spill = [&](int rs) {
			unsigned int n = st->n[rs];
			if (n > SLOT_CAP)
				n = SLOT_CAP;
			unsigned long long base = 0;
			if (lane == 0)
				base = atomicAdd(&ov.counters[4], (unsigned long long)n);
			base = __shfl(base, 0);
			for (unsigned int i = lane; i < n; i += 64)
				if (base + i < ov.ovf_cap) {
					const pgx_hit hh = RPW == 2 ? (db.gapped ? unpack_seed(st->chit[rs * SLOT_CAP + i], rs ? rB : rA, rs ? wB : wA, rs ? LB : LA)
										  : unpack_hit(st->chit[rs * SLOT_CAP + i], rs ? rB : rA))
								     : st->hit[i];
					ov.ovf[base + i] = hh;
					if (db.gapped)
						ov.ovf_key[base + i] = seed_key(hh);
				}
			if (rs)
				emitted[1] += n, frag[1] = true;
			else
				emitted[0] += n, frag[0] = true;
			lds_fence();
			if (lane == 0)
				st->n[rs] = 0;
			lds_fence();
		};

		auto drain = [&](unsigned int cnt) {
			// the last `cnt` queue entries, one per lane
			q_n -= cnt;
			if ((unsigned)lane < cnt && PGX_DBG_STOP(db) != 3) {
				const uint32_t p = st->qp[q_n + lane], meta = st->qmeta[q_n + lane];
				const uint32_t sj = st->qsubj[q_n + lane], s0 = st->qs0[q_n + lane], s1 = st->qs1[q_n + lane];
				const int strand = (int)(meta >> 31), rs = (int)((meta >> 29) & 1), qp = (int)(meta & 0x0FFFFFFFu);
				const bool tested = (meta >> 30) & 1, claimed = (meta >> 28) & 1;
				const uint32_t cw0 = rs ? wB : wA, cr = rs ? rB : rA;
				const int cL = rs ? LB : LA;
				const uint64_t *rw = (strand ? rd.rc : rd.fwd) + cw0;
				const uint64_t *ra = nullptr;
				if (AMB) {
					const uint64_t *a = strand ? rd.rc_amb : rd.fwd_amb;
					ra = a ? a + cw0 : nullptr;
				}
				// S3d window bits of this candidate's read and strand: LDS copy (dense flags), or the words in memory
				const uint64_t *uw = nullptr;
				if (DUST && (rs ? dustB : dustA))
					uw = DENSE ? st->uwin[rs][strand] : (strand ? rd.dustwin_r : rd.dustwin_f) + cw0;
				auto emit = [&](const pgx_hit &hh) { emit_hit<RPW == 2>(st, rs, ov, hh, db.gapped != 0); };
				if (DENSE)
					process_candidate<AMB, DenseMask<AMB, (NW > 0 ? NW : 3)>>(db, rw, ra, uw, cL, cr, cw0, strand, qp, p, tested, claimed, sj, s0, s1, emit, n_runs);
				else
					process_candidate<AMB, LazyMask<AMB>>(db, rw, ra, uw, cL, cr, cw0, strand, qp, p, tested, claimed, sj, s0, s1, emit, n_runs);
			}
			lds_fence();
			// a read's stage more than half full in the middle of the work: the read becomes fragmented
			if (st->n[0] > SLOT_CAP / 2)
				spill(0);
			if (RPW == 2 && st->n[1] > SLOT_CAP / 2)
				spill(1);
		};

		for (int pbase = 0; pbase < Pmax; pbase += LPR) {
			const int pid = pbase + my_li;
			uint32_t cnt = 0, lo = 0;
			int strand = 0, qpos = 0;
			uint32_t wl = 0, wr = 0; // the read's 16 bases from 13 left of the probe, and the 16 right of it
			uint32_t flank_amb = 0;  // bit 0 / 1: an ambiguity letter in the read's left / right flank
			if (pid < P && (my_rs == 0 || hasB)) {
				strand = pid >= nps;
				qpos = (pid - strand * nps) * kProbeStride;
				const uint64_t *rw = (strand ? rd.rc : rd.fwd) + w0;
				wl = qpos >= kProbeStride ? window16(rw, qpos - kProbeStride) : 0u;
				wr = window16(rw, qpos + kSeedK);
				if (AMB) {
					// ambiguity letters in the read's own flanks switch the corresponding filter off for this probe
					const uint64_t *ra0 = strand ? rd.rc_amb : rd.fwd_amb;
					if (ra0) {
						if (qpos >= kProbeStride && (window64(ra0 + w0, qpos - kProbeStride) & ((1ull << 26) - 1)))
							flank_amb |= 1;
						if (window64(ra0 + w0, qpos + kSeedK) & ((1ull << 24) - 1))
							flank_amb |= 2;
					}
				}
				bool ok = true;
				if (AMB) {
					const uint64_t *ra = strand ? rd.rc_amb : rd.fwd_amb;
					if (ra && (uint32_t)window64(ra + w0, qpos))
						ok = false;
				}
				if (ok) {
					uint32_t b = seed_bucket(window16(rw, qpos), db.bits);
					lo = __builtin_nontemporal_load(&db.bucket_off[b]);
					cnt = __builtin_nontemporal_load(&db.bucket_off[(uint64_t)b + 1]) - lo;
				}
				n_probe++;
			}
			// wave prefix sum of the posting counts
			uint32_t incl = cnt;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				uint32_t t = __shfl_up(incl, d);
				if (lane >= d)
					incl += t;
			}
			const uint32_t excl = incl - cnt;
			const uint32_t T = __shfl(incl, 63);
			const uint32_t pbase_idx = lo - excl;                               // posting index of item k of this probe: pbase_idx + k
			const uint32_t pmeta = (uint32_t)qpos | (flank_amb << 28) | ((uint32_t)strand << 31); // what an item needs to know of its probe
			n_post += cnt;
			if (PGX_DBG_STOP(db) == 1)
				continue;
			for (uint32_t it = 0; it < T; it += 64 * kDeal) {
				// kDeal postings per lane, every load of a stage issued before the first use
				bool active[kDeal], tested[kDeal], keep[kDeal];
				uint32_t pidx[kDeal], p[kDeal], sj[kDeal], s0[kDeal], s1[kDeal];
				int o_strand[kDeal], o_qpos[kDeal], o_rs[kDeal];
				uint32_t o_wl[kDeal], o_wr[kDeal], o_famb[kDeal];
#pragma unroll
				for (int u = 0; u < kDeal; u++) {
					const uint32_t item = it + 64 * u + lane;
					active[u] = item < T;
					const uint32_t key = active[u] ? item : T - 1;
					// owner probe: the last lane whose exclusive prefix is <= item
					int o = 0;
#pragma unroll
					for (int step = 32; step >= 1; step >>= 1) {
						int cand = o + step;
						uint32_t e = __shfl(excl, cand & 63);
						if (cand < 64 && e <= key)
							o = cand;
					}
					pidx[u] = __shfl(pbase_idx, o) + key;
					const uint32_t m = __shfl(pmeta, o);
					o_strand[u] = (int)(m >> 31);
					o_qpos[u] = (int)(m & 0x0FFFFFFFu);
					o_famb[u] = AMB ? (m >> 28) & 3u : 0u;
					o_rs[u] = o / LPR;
					o_wl[u] = __shfl(wl, o);
					o_wr[u] = __shfl(wr, o);
				}
				uint32_t raw[kDeal];
				uint2 ctx[kDeal];
#pragma unroll
				for (int u = 0; u < kDeal; u++) {
					raw[u] = 0u;
					ctx[u] = make_uint2(0u, 0u);
					if (active[u]) {
						// posting and its context in one 12-byte record
						const uint32_t *rp = reinterpret_cast<const uint32_t *>(db.post_ctx + pidx[u]);
						raw[u] = __builtin_nontemporal_load(rp);
						ctx[u] = make_uint2(__builtin_nontemporal_load(rp + 1), __builtin_nontemporal_load(rp + 2));
					}
				}
				// stage 2: the 13 database bases left and the 12 right of the 16-mer (post_ctx, fetched beside
				// the posting) against the read's (extracted once per probe, handed over by its lane)
				uint32_t xl[kDeal], xr[kDeal];
				bool lknown[kDeal], rknown[kDeal];
#pragma unroll
				for (int u = 0; u < kDeal; u++) {
					p[u] = raw[u];
					keep[u] = active[u];
					tested[u] = false;
					xl[u] = xr[u] = 0;
					lknown[u] = rknown[u] = false;
					if (active[u]) {
						// flanks with an ambiguity letter (database: bit 31 of the context; read: the probe's flags)
						// are unknown: the filters then assume the most they could match
						const bool db_clean = !AMB || !(ctx[u].x >> 31);
						const uint32_t fa = AMB ? o_famb[u] : 0u; // reads of the unambiguous classes have clean flanks
						lknown[u] = db_clean && !(fa & 1u) && o_qpos[u] >= kProbeStride;
						rknown[u] = db_clean && !(fa & 2u);
						if (lknown[u]) {
							// a posting within 13 bases of its sequence's start (bit 30 of the context) is never "tested"
							tested[u] = !((ctx[u].x >> 30) & 1u);
							xl[u] = ctx[u].x ^ o_wl[u];
						}
						if (rknown[u])
							xr[u] = ctx[u].y ^ o_wr[u];
					}
				}
				// stage 3: la / ra = matching bases immediately left / right of the 16-mer (capped at 13 / 12,
				// and by the read's ends).  la == 13: the previous probe reports this run: drop.
				// la + 16 + ra < 28: the exact run around the 16-mer is shorter than a word: drop.
				// (Subject boundaries are ignored here, so this only ever keeps too much; the exact tests
				// follow on the diagonal's flags.)
#pragma unroll
				for (int u = 0; u < kDeal; u++) {
					if (keep[u]) {
						const uint32_t ml = (xl[u] | (xl[u] >> 1)) & 0x01555555u; // bases 0..12
						const uint32_t mr = (xr[u] | (xr[u] >> 1)) & 0x00555555u; // bases 0..11
						int la = o_qpos[u] >= kProbeStride ? (lknown[u] ? (ml ? 12 - ((31 - __clz((int)ml)) >> 1) : kProbeStride) : kProbeStride) : 0;
						int ra = rknown[u] ? (mr ? (__ffs((int)mr) - 1) >> 1 : kWord - kSeedK) : kWord - kSeedK;
						if (tested[u] && la == kProbeStride)
							keep[u] = false;
						const int room = (o_rs[u] ? LB : LA) - o_qpos[u] - kSeedK;
						ra = ra < room ? ra : room;
						if (la + ra < kWord - kSeedK)
							keep[u] = false;
					}
				}
				if (PGX_DBG_STOP(db) == 8) { // (measurement builds: postings fetched and filtered, no block record yet)
					for (int u = 0; u < kDeal; u++)
						n_runs += keep[u];
					continue;
				}
				// stage 4: subject and bounds of the survivors: the block's first subject, or (a boundary inside
				// the block) the next one; only blocks holding three or more subjects walk seq_off
				uint4 bi[kDeal];
#pragma unroll
				for (int u = 0; u < kDeal; u++)
					bi[u] = keep[u] ? db.blk_info[p[u] >> kBlkShift] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
				for (int u = 0; u < kDeal; u++) {
					sj[u] = bi[u].x;
					s0[u] = bi[u].y;
					s1[u] = bi[u].z;
					if (keep[u] && s1[u] <= p[u]) {
						sj[u]++;
						s0[u] = s1[u];
						s1[u] = bi[u].w;
						while (s1[u] <= p[u]) {
							sj[u]++;
							s0[u] = s1[u];
							s1[u] = db.seq_off[sj[u] + 1];
						}
					}
				}
				if (PGX_DBG_STOP(db) == 9) { // (measurement builds: block records fetched, no ownership set yet)
					for (int u = 0; u < kDeal; u++)
						n_runs += keep[u] + s1[u];
					continue;
				}
				// stage 5 (two short reads per wave): one candidate per diagonal.  The first survivor of a
				// (read, strand, diagonal) enters the set and will take ALL exact runs of the diagonal from its
				// window bitmap; later survivors of the same diagonal (one per further >= 28 run) are dropped
				// here, before any flag build.  A diagonal that leaves its subject inside the read's span, or a
				// full probe window of the set, falls back to the left-most-probe-of-the-first-run rule, which
				// needs no bookkeeping; the two rules never mix on one diagonal (a key that is absent once the
				// window is full was never inserted), so each diagonal is still reported exactly once.
				bool claimed[kDeal];
#pragma unroll
				for (int u = 0; u < kDeal; u++) {
					claimed[u] = false;
					if (RPW == 2 && keep[u]) {
						const int64_t d0 = (int64_t)p[u] - o_qpos[u];
						const int Lr = o_rs[u] ? LB : LA;
						if (d0 >= (int64_t)s0[u] && d0 + Lr <= (int64_t)s1[u]) {
							const uint32_t key = (uint32_t)d0; // positions are 32-bit: one table per strand
							const uint32_t h0 = (key * 0x9E3779B1u) >> 26;
							for (int t = 0; t < kDiagProbes; t++) {
								const uint32_t old = atomicCAS(&st->diag[o_rs[u]][o_strand[u]][(h0 + t) & (kDiagSlots / 2 - 1)], kNoDiag, key);
								if (old == kNoDiag) {
									claimed[u] = true;
									break;
								}
								if (old == key) {
									keep[u] = false;
									break;
								}
							}
						}
					}
				}
				if (PGX_DBG_STOP(db) == 2) {
					for (int u = 0; u < kDeal; u++)
						n_runs += keep[u] + s1[u];
					continue;
				}
				// compact the survivors into the queue
#pragma unroll
				for (int u = 0; u < kDeal; u++) {
					const unsigned long long km = __ballot(keep[u]);
					if (keep[u]) {
						const unsigned int slot = q_n + (unsigned)__popcll(km & lt_mask);
						st->qp[slot] = p[u];
						st->qmeta[slot] = ((uint32_t)o_strand[u] << 31) | ((uint32_t)tested[u] << 30) | ((uint32_t)o_rs[u] << 29) |
								  ((uint32_t)claimed[u] << 28) | (uint32_t)o_qpos[u];
						st->qsubj[slot] = sj[u];
						st->qs0[slot] = s0[u];
						st->qs1[slot] = s1[u];
					}
					q_n += (unsigned)__popcll(km);
					n_surv += lane == 0 ? (unsigned long long)__popcll(km) : 0ull;
				}
				lds_fence();
				while (q_n >= 64)
					drain(64);
			}
		}
		if (q_n)
			drain(q_n);
		lds_fence();
		// close the reads: staged hits leave contiguously into the wave's chunk of the hit table
#pragma unroll
		for (int rs = 0; rs < RPW; rs++) {
			if (rs == 1 && !hasB)
				break;
			unsigned int n = st->n[rs];
			if (n > SLOT_CAP)
				n = SLOT_CAP;
			bool fr = rs ? frag[1] : frag[0];
			if (st->direct[rs])
				fr = true;
			uint32_t start = kFragmented;
			unsigned int em = rs ? emitted[1] : emitted[0];
			if (n) {
				unsigned long long base = 0;
				if (fr) {
					if (lane == 0)
						base = atomicAdd(&ov.counters[4], (unsigned long long)n);
					base = __shfl(base, 0);
				} else {
					// sub-allocate from the wave's chunk of the main table
					if (chunk_used + n > kChunk) {
						if (lane == 0)
							chunk_base = atomicAdd(&ov.counters[0], (unsigned long long)kChunk);
						chunk_base = __shfl(chunk_base, 0);
						chunk_used = 0;
					}
					base = chunk_base + chunk_used;
					chunk_used += n;
					n_main += lane == 0 ? n : 0;
				}
				pgx_hit *dst = fr ? ov.ovf : ov.main;
				uint8_t *dkey = fr ? ov.ovf_key : ov.main_key;
				const unsigned long long dcap = fr ? ov.ovf_cap : ov.main_cap;
				for (unsigned int i = lane; i < n; i += 64)
					if (base + i < dcap) {
						const pgx_hit hh = RPW == 2 ? (db.gapped ? unpack_seed(st->chit[rs * SLOT_CAP + i], rs ? rB : rA, rs ? wB : wA, rs ? LB : LA)
											  : unpack_hit(st->chit[rs * SLOT_CAP + i], rs ? rB : rA))
									     : st->hit[i];
						dst[base + i] = hh;
						if (db.gapped) {
							dkey[base + i] = seed_key(hh);
							if (!fr)
								ov.main_reg[base + i] = (uint8_t)((uint32_t)hh.sstart >> ov.reg_shift);
						}
					}
				em += n;
				if (!fr)
					start = (uint32_t)base;
			}
			em += st->direct[rs];
			if (lane == 0) {
				read_cnt[rs ? rB : rA] = em;
				read_start[rs ? rB : rA] = start;
			}
		}
		lds_fence();
	}
	// per-wave statistics (a handful of atomics per wave, not per read)
	for (int d = 32; d >= 1; d >>= 1) {
		n_probe += __shfl_down(n_probe, d);
		n_post += __shfl_down(n_post, d);
		n_runs += __shfl_down(n_runs, d);
	}
	if (lane == 0) {
		atomicAdd(&ov.counters[5], n_main);
		atomicAdd(&ov.counters[6], n_surv);
		atomicAdd(&ov.counters[1], n_probe);
		atomicAdd(&ov.counters[2], n_post);
		atomicAdd(&ov.counters[3], n_runs);
	}
}

// ------------------------------------------------------------------------------------------ grouping by read
constexpr int kScanBlock = 256, kScanItems = 8; // 2048 counts per block

__global__ __launch_bounds__(kScanBlock) void k_scan_partials(const uint32_t *__restrict__ cnt, uint64_t n,
							      uint32_t *__restrict__ partial)
{
	__shared__ uint32_t s[kScanBlock];
	uint64_t base = (uint64_t)blockIdx.x * kScanBlock * kScanItems;
	uint32_t sum = 0;
	for (int k = 0; k < kScanItems; k++) {
		uint64_t i = base + (uint64_t)threadIdx.x * kScanItems + k;
		if (i < n)
			sum += cnt[i];
	}
	s[threadIdx.x] = sum;
	__syncthreads();
	for (int d = kScanBlock / 2; d >= 1; d >>= 1) {
		if ((int)threadIdx.x < d)
			s[threadIdx.x] += s[threadIdx.x + d];
		__syncthreads();
	}
	if (threadIdx.x == 0)
		partial[blockIdx.x] = s[0];
}

// exclusive scan of the block partials by one block (chunks of 256 with a running carry)
__global__ __launch_bounds__(kScanBlock) void k_scan_top(uint32_t *__restrict__ partial, uint32_t n_part)
{
	__shared__ uint32_t s[kScanBlock];
	__shared__ uint32_t carry;
	if (threadIdx.x == 0)
		carry = 0;
	__syncthreads();
	for (uint32_t base = 0; base < n_part; base += kScanBlock) {
		uint32_t i = base + threadIdx.x;
		uint32_t v = i < n_part ? partial[i] : 0;
		s[threadIdx.x] = v;
		__syncthreads();
		for (int d = 1; d < kScanBlock; d <<= 1) {
			uint32_t t = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0;
			__syncthreads();
			s[threadIdx.x] += t;
			__syncthreads();
		}
		uint32_t incl = s[threadIdx.x];
		if (i < n_part)
			partial[i] = carry + incl - v;
		__syncthreads();
		if (threadIdx.x == kScanBlock - 1)
			carry += incl;
		__syncthreads();
	}
}

__global__ __launch_bounds__(kScanBlock) void k_scan_final(const uint32_t *__restrict__ cnt, uint64_t n,
							   const uint32_t *__restrict__ partial, uint32_t *__restrict__ off)
{
	__shared__ uint32_t s[kScanBlock];
	uint64_t base = (uint64_t)blockIdx.x * kScanBlock * kScanItems;
	uint32_t v[kScanItems], sum = 0;
	for (int k = 0; k < kScanItems; k++) {
		uint64_t i = base + (uint64_t)threadIdx.x * kScanItems + k;
		v[k] = i < n ? cnt[i] : 0;
		sum += v[k];
	}
	s[threadIdx.x] = sum;
	__syncthreads();
	for (int d = 1; d < kScanBlock; d <<= 1) {
		uint32_t t = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0;
		__syncthreads();
		s[threadIdx.x] += t;
		__syncthreads();
	}
	uint32_t run = partial[blockIdx.x] + s[threadIdx.x] - sum;
	for (int k = 0; k < kScanItems; k++) {
		uint64_t i = base + (uint64_t)threadIdx.x * kScanItems + k;
		if (i < n)
			off[i] = run;
		run += v[k];
	}
	if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kScanBlock - 1)
		off[n] = run;
}

// the overflow table's hits into their reads' slots; the count is read on the device (the host does not wait for it)
__global__ void k_scatter_hits(const pgx_hit *__restrict__ in, const unsigned long long *__restrict__ n_ptr, unsigned long long in_cap,
			       const uint32_t *__restrict__ off, uint32_t *__restrict__ cursor, pgx_hit *__restrict__ out,
			       unsigned long long out_cap)
{
	const uint64_t n_hits = *n_ptr < in_cap ? *n_ptr : in_cap;
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < n_hits; i += stride) {
		pgx_hit h = in[i];
		uint32_t slot = off[h.read] + atomicAdd(&cursor[h.read], 1u);
		if (slot < out_cap)
			out[slot] = h;
	}
}

// ------------------------------------------------------------------------------------------ per-read order + consensus
struct ConsView {
	const uint32_t *subj_tok_off, *subj_tok; // per subject token ids
	const int8_t *tok_rank;                  // token id -> index in "0".."6" or -1
	const uint32_t *simrank_lut;             // pident hundredths -> string-order rank
	const uint32_t *simrank_len;             // [length * 256 + mismatches] for alignments < 256 long
	uint32_t simrank_undef, simrank_zero;
	const uint32_t *rdp_off, *rdp_name;
	const int8_t *rdp_rank;
	const uint8_t *rdp_present;
	const uint32_t *subj_pairs; // pair_words (8 or 16) words per subject: ntok | npairs << 16, then name << 3 | rank + 1 per pair
	int pair_words;
	const uint32_t *rdp_code;   // name << 3 | rank + 1 per RDP triplet
	int dbg;                    // profiling aid (PGX_SORT_STOP): truncate k_sort_consensus after a stage
	int np_max, nr_max;         // most pairs of any subject record / triplets of any read: the compare grid is np_max x nr_max
};

// string-order rank of the hit's pident text
__device__ __forceinline__ uint32_t hit_simrank(const ConsView &cv, const pgx_hit &h)
{
	const int len = hit_length(h), diffs = hit_diffs(h); // columns, columns that are not identities
	if (len < 256)
		return cv.simrank_len[len * 256 + diffs];
	return cv.simrank_lut[pident_hundredths(len - diffs, len)];
}

constexpr int kRdpRegs = 8; // RDP triplets of a read kept in registers by the fast agreement count

template <int NP, int NR>
__device__ __forceinline__ uint32_t pair_grid(const uint32_t (&pr)[15], const uint32_t (&rc)[kRdpRegs], uint32_t np)
{
	uint32_t rm = 0;
#pragma unroll
	for (int a = 0; a < NP; a++) {
		if ((uint32_t)a < np) {
#pragma unroll
			for (int b = 0; b < NR; b++)
				rm += pr[a] == rc[b];
		}
	}
	return rm;
}

// fast (rank,name) agreement: one 64-byte record per subject against the read's RDP codes in registers
__device__ __forceinline__ uint32_t pair_matches(const ConsView &cv, uint32_t subject, const uint32_t (&rc)[kRdpRegs],
						  uint32_t r0, uint32_t r1, uint32_t *ntok_out)
{
#ifdef PGX_STAGE_PROBES
	if (cv.dbg == 7)
		subject &= 0x3FFFu; // (what the stage would cost if every subject's record were in L2: 16 384 records)
#endif
	// (both halves of a 32-byte record are asked for at once: the second used to be asked for only after the first had
	// arrived and said how many pairs there are -- two trips to memory in the middle of the ordering kernel's chain)
	const uint4 *rec = reinterpret_cast<const uint4 *>(cv.subj_pairs + (unsigned long long)cv.pair_words * subject);
	const uint4 q0 = rec[0];
	const uint4 q1 = rec[1];
	const uint32_t nt = q0.x & 0xFFFFu, np = q0.x >> 16;
	*ntok_out = nt;
	if (np == 0xFFFFu || r1 - r0 > (uint32_t)kRdpRegs) {
		// more pairs / triplets than the record or the registers hold: the general count
		const uint32_t t0 = cv.subj_tok_off[subject], n2 = cv.subj_tok_off[subject + 1] - t0;
		*ntok_out = n2;
		return rank_matches(cv.subj_tok + t0, n2, cv.tok_rank, cv.rdp_name, cv.rdp_rank, r0, r1);
	}
	uint4 q2 = make_uint4(0u, 0u, 0u, 0u), q3 = q2;
	if (cv.pair_words > 8) { // (kernel-uniform: records of 32 bytes hold up to 7 pairs)
		q2 = rec[2];
		q3 = rec[3];
	}
	const uint32_t pr[15] = { q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w };
	// the compare grid: 7 pairs x 6 triplets for the usual seven-rank lineages (one kernel-uniform branch), 15 x 8 otherwise
	if (cv.np_max <= 7 && cv.nr_max <= 6)
		return pair_grid<7, 6>(pr, rc, np);
	return pair_grid<15, kRdpRegs>(pr, rc, np);
}

// (rank,name) agreement of one hit with the read's RDP triplets
__device__ __forceinline__ uint32_t hit_rank_matches(const ConsView &cv, uint32_t subject, uint32_t r0, uint32_t r1,
						      uint32_t *ntok_out)
{
	const uint32_t t0 = cv.subj_tok_off[subject], nt = cv.subj_tok_off[subject + 1] - t0;
	*ntok_out = nt;
	return rank_matches(cv.subj_tok + t0, nt, cv.tok_rank, cv.rdp_name, cv.rdp_rank, r0, r1);
}

constexpr uint32_t kNoRead = 0xFFFFFFFFu; // hole in a chunked read list
constexpr int kSortCap = 64; // hits of one read ordered by its wave: one hit per lane

// S5 order as two integers compared lexicographically (reads up to 65 535 bases; longer reads take the
// field-by-field comparison of the big-read path).  `send` needs one bit: with equal qstart, qend and sstart
// the minus-strand hit (send = sstart - span) precedes the plus-strand one (send = sstart + span);
// sstart < 2^31 because database positions are.
struct SortKey {
	uint64_t k1, k2;
	uint32_t send, mg; // third level (gapped hits: send no longer follows from the other columns): send, then mismatch, gap openings
};
__device__ __forceinline__ SortKey make_key(const pgx_hit &h, int best)
{
	SortKey k;
	k.k1 = ((uint64_t)(0xFFFF - best) << 48) | ((uint64_t)(uint32_t)h.subject << 16) | (uint64_t)(0xFFFF - h.score);
	k.k2 = ((uint64_t)(uint32_t)h.qstart << 48) | ((uint64_t)(uint32_t)h.qend << 32) |
	       (uint64_t)(((uint32_t)h.sstart << 1) | (h.send > h.sstart ? 1u : 0u));
	k.send = (uint32_t)h.send;
	k.mg = ((uint32_t)h.mismatch << 16) | (uint32_t)h.gapopen;
	return k;
}

// LDS of one wave.  Phase A holds (subject, score) pairs then the sort keys; phase B (after every lane has
// its rank) reuses the same bytes for the per-rank consensus inputs.  3 KB per wave keeps the kernel at the
// register-limited occupancy: it is bound by the latency of its gathers, not by arithmetic.
struct SortWave {
	union {
		struct {
			uint64_t k1[kSortCap], k2[kSortCap];
			int subj[kSortCap], score[kSortCap];
			uint32_t send[kSortCap], mg[kSortCap];
		} a;
		struct {
			uint64_t krm[kSortCap], kcnt[kSortCap]; // text-order keys of the agreement and token counts
			uint32_t rm[kSortCap], sim[kSortCap];
		} b;
		struct { // the same per-rank inputs with 32-bit text-order keys (counts below 10^8), plus the chain pointers
			uint32_t kcnt[kSortCap], sim_e[kSortCap], krm[kSortCap], rm[kSortCap], nxt[kSortCap];
		} c;
	};
	uint32_t rcode[2][kRdpRegs]; // the RDP codes of the wave's (up to) two reads: fetched one per lane at the top of a round, read
				     // back when the ranks are known -- eight registers per lane less through the ranking loops
};

// G lanes per read: G = 32 orders two reads per wavefront (most reads have <= 32 hits), G = 64 one.  A read with
// more than G hits is passed on through `next_list` (to the G = 64 launch, then to the big-read path).
// `list`/`n_list_ptr`: optional indirection (read ids and their count in device memory); null = all reads.
template <int G>
__global__ __launch_bounds__(64 * kWavesPerBlock, 8) void k_sort_consensus(pgx_hit *__restrict__ hits,
									 const pgx_hit *__restrict__ scratch,
									 const uint32_t *__restrict__ read_start,
									 const uint32_t *__restrict__ off, uint32_t *__restrict__ read_cnt,
									 unsigned long long hits_cap, unsigned long long scratch_cap, uint32_t n_reads,
									 const uint32_t *__restrict__ list,
									 const uint32_t *__restrict__ n_list_ptr, ConsView cv,
									 int do_consensus, int lds_ok,
									 pgx_consensus_rec *__restrict__ recs,
									 uint32_t *__restrict__ next_list,
									 uint32_t *__restrict__ next_count)
{
	constexpr int RPW = 64 / G; // reads per wavefront
	__shared__ SortWave s_sw[kWavesPerBlock];
	const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform: scalar registers
	SortWave *sw = &s_sw[wave_id];
	const int lane = threadIdx.x & 63;
	const int g = lane / G, li = lane % G, slot0 = g * G;
	const uint32_t total = list ? *n_list_ptr : n_reads;
	uint32_t chunk_base = 0, chunk_used = 64; // this wave's chunk of next_list (G == 32)
	for (uint32_t base = (blockIdx.x * kWavesPerBlock + (uint32_t)wave_id) * RPW; base < total;
	     base += gridDim.x * kWavesPerBlock * RPW) {
		const uint32_t idx = base + (uint32_t)g;
		bool valid = idx < total;
		uint32_t r = valid ? (list ? list[idx] : idx) : 0u;
		if (r == kNoRead) { // unused slot of a chunked list
			valid = false;
			r = 0;
		}
		const uint32_t o = valid ? off[r] : 0u;
		uint32_t n = valid ? off[r + 1] - o : 0u;
		if ((unsigned long long)o + n > hits_cap)
			n = 0; // the table was sized on a guess that was too small: the host grows it and repeats the step
		if (valid && n == 0 && do_consensus && li == 0) {
			recs[r].hit = -2;
			recs[r].matches = 0;
		}
		const bool pass_on = valid && (n > (uint32_t)G || !lds_ok);
		if (G == 64) {
			if (pass_on && li == 0)
				next_list[atomicAdd(next_count, 1u)] = r; // rare (> 64 hits): one atomic each is fine
		} else {
			// a third of the reads take this exit: list slots come in chunks of 64 per wave (a single hot
			// counter sustains only ~90 M atomics/s); unused slots of a chunk are closed with kNoRead
			const unsigned long long pm = __ballot(pass_on && li == 0);
			if (pm) {
				const uint32_t cnt = (uint32_t)__popcll(pm);
				if (chunk_used + cnt > 64u) {
					if (lane == 0) {
						for (uint32_t k = chunk_used; k < 64u; k++)
							next_list[chunk_base + k] = kNoRead;
						chunk_base = atomicAdd(next_count, 64u);
					}
					chunk_base = __shfl(chunk_base, 0);
					chunk_used = 0;
				}
				if (pass_on && li == 0)
					next_list[chunk_base + chunk_used + (uint32_t)__popcll(pm & ((1ull << lane) - 1))] = r;
				chunk_used += cnt;
			}
		}
		if (pass_on)
			n = 0; // not ours
		// the read's RDP codes (slots past the end match nothing): lane b of the read's lanes fetches code b
		// (asked for before the rows: the two chains of dependent loads run side by side)
		uint32_t rdp0 = 0, rdp1 = 0, my_code = 0xFFFFFFFEu;
		bool present = true;
		if (do_consensus && n) {
			rdp0 = cv.rdp_off[r];
			rdp1 = cv.rdp_off[r + 1];
			if (li < kRdpRegs && rdp0 + (uint32_t)li < rdp1)
				my_code = cv.rdp_code[rdp0 + (uint32_t)li];
			if (cv.rdp_present)
				present = cv.rdp_present[r] != 0;
		}
		// unfragmented reads still sit contiguously in the seed kernel's table; fragmented ones were scattered
		uint32_t st0 = n ? read_start[r] : 0u;
		if (n && st0 != kFragmented && (unsigned long long)st0 + n > scratch_cap) {
			n = 0; // the seed stage's table was too small: the step is repeated
			st0 = 0;
		}
		const pgx_hit *src = st0 == kFragmented ? hits + o : scratch + st0;
		const bool mine = (uint32_t)li < n;
		pgx_hit h;
		h.subject = h.score = h.qstart = h.qend = h.sstart = h.send = h.read = 0;
		h.mismatch = h.gapopen = 0;
		if (mine)
			h = src[li];
		// (rows past the read's end hold a subject no hit has, so the loops below need no "j < n" of their own)
		sw->a.subj[slot0 + li] = mine ? h.subject : -1;
		sw->a.score[slot0 + li] = h.score;
		if (li < kRdpRegs)
			sw->rcode[g][li] = my_code;
		lds_fence();
		if (PGX_SORT_DBG(cv) == 1) {
			if (mine)
				hits[o + li] = h;
			continue;
		}
		// best score of the hit's subject.  Loops run a wave-uniform number of rounds, four LDS rows in
		// flight per round; rows past a read's end (stale bytes) are masked by j < n.
		const uint32_t nmax = RPW == 2 ? max(__shfl(n, 0), __shfl(n, 32)) : n;
		int best = h.score;
		uint32_t same_subj = 0; // hits of this read on the hit's subject (itself included)
		for (uint32_t j0 = 0; j0 < nmax; j0 += 4) {
			int sj[4], sc[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				sj[u] = sw->a.subj[slot0 + ((j0 + u) & (G - 1))];
				sc[u] = sw->a.score[slot0 + ((j0 + u) & (G - 1))];
			}
#pragma unroll
			for (int u = 0; u < 4; u++)
				if (sj[u] == h.subject) {
					same_subj++;
					if (sc[u] > best)
						best = sc[u];
				}
		}
		const SortKey kx = make_key(h, best);
		lds_fence();
		if (PGX_SORT_DBG(cv) == 2) {
			if (mine) {
				h.score = best;
				hits[o + li] = h;
			}
			continue;
		}
		sw->a.k1[slot0 + li] = mine ? kx.k1 : ~0ull; // (past the end: a key that precedes none and equals none)
		if (mine) {
			sw->a.k2[slot0 + li] = kx.k2;
			sw->a.send[slot0 + li] = kx.send;
			sw->a.mg[slot0 + li] = kx.mg;
		}
		lds_fence();
		// Spec v2, S3c: hits of one subject that describe the same alignment (the seeds on either side of a gap all grow
		// into it).  A hit is dropped when a hit BEFORE it in the S5 order, on the same subject and strand, starts at the
		// same point, ends at the same point, or holds it.  Only reads with two hits on one subject look (wave-uniform).
		bool dropped = false;
		if (__ballot(mine && same_subj > 1u)) {
			const uint32_t my_s0 = (uint32_t)(h.send > h.sstart ? h.sstart : h.send), my_s1 = (uint32_t)(h.send > h.sstart ? h.send : h.sstart);
			for (uint32_t j = 0; j < nmax; j++) {
				const uint32_t sl = slot0 + (j & (G - 1));
				const uint64_t b1 = sw->a.k1[sl], b2 = sw->a.k2[sl];
				const uint32_t bsend = sw->a.send[sl], bmg = sw->a.mg[sl];
				const bool before = (b1 < kx.k1) | ((b1 == kx.k1) & ((b2 < kx.k2) | ((b2 == kx.k2) & ((bsend < kx.send) | ((bsend == kx.send) &
						    ((bmg < kx.mg) | ((bmg == kx.mg) & (j < (uint32_t)li))))))));
				const bool same = (uint32_t)(b1 >> 16) == (uint32_t)h.subject && ((b2 ^ kx.k2) & 1ull) == 0ull;
				const uint32_t bq0 = (uint32_t)(b2 >> 48), bq1 = (uint32_t)(b2 >> 32) & 0xFFFFu, bsst = (uint32_t)(b2 >> 1) & 0x7FFFFFFFu;
				const uint32_t bs0 = bsend > bsst ? bsst : bsend, bs1 = bsend > bsst ? bsend : bsst;
				const bool dup = (bq0 == (uint32_t)h.qstart && bsst == (uint32_t)h.sstart) || (bq1 == (uint32_t)h.qend && bsend == (uint32_t)h.send) ||
						 ((uint32_t)h.qstart >= bq0 && (uint32_t)h.qend <= bq1 && my_s0 >= bs0 && my_s1 <= bs1);
				dropped |= (j < n) & before & same & dup;
			}
		}
		const unsigned long long drop_mask = __ballot(mine && dropped);
		const unsigned long long grp_mask = G == 64 ? ~0ull : (g ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull);
		const uint32_t n_drop = (uint32_t)__popcll(drop_mask & grp_mask);
		const uint32_t n_all = n; // slots of the read; n becomes the hits that are kept
		const bool kept = mine && !dropped;
		if (n_drop) {
			n -= n_drop;
			if (li == 0)
				read_cnt[r] = n;
		}
		// rank = number of hits that precede this one.  Nearly every pair differs in the first key word (best
		// score, subject, score): count on that word alone, and compare second words only inside groups of equal
		// first words (several HSPs of one subject with one score), which most reads do not have.
		uint32_t rank = 0, same = 0;
		if (drop_mask == 0ull) { // the usual wavefront: nothing was dropped, rows past a read's end hold the last key
			for (uint32_t j0 = 0; j0 < nmax; j0 += 4) {
				uint64_t a1[4];
#pragma unroll
				for (int u = 0; u < 4; u++)
					a1[u] = sw->a.k1[slot0 + ((j0 + u) & (G - 1))];
#pragma unroll
				for (int u = 0; u < 4; u++) {
					rank += a1[u] < kx.k1 ? 1u : 0u;
					same += a1[u] == kx.k1 ? 1u : 0u;
				}
			}
		} else {
			for (uint32_t j0 = 0; j0 < nmax; j0 += 4) {
				uint64_t a1[4];
#pragma unroll
				for (int u = 0; u < 4; u++)
					a1[u] = sw->a.k1[slot0 + ((j0 + u) & (G - 1))];
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const bool in = (j0 + u < n_all) & !((drop_mask >> (slot0 + ((j0 + u) & (G - 1)))) & 1ull);
					rank += (in & (a1[u] < kx.k1)) ? 1u : 0u;
					same += (in & (a1[u] == kx.k1)) ? 1u : 0u;
				}
			}
		}
		if (__ballot(kept && same > 1u)) {
			for (uint32_t j = 0; j < nmax; j++) {
				const uint32_t sl = slot0 + (j & (G - 1));
				const uint64_t b1 = sw->a.k1[sl], b2 = sw->a.k2[sl];
				const uint32_t bsend = sw->a.send[sl], bmg = sw->a.mg[sl];
				const bool before = (b1 == kx.k1) & ((b2 < kx.k2) | ((b2 == kx.k2) & ((bsend < kx.send) | ((bsend == kx.send) &
						    ((bmg < kx.mg) | ((bmg == kx.mg) & (j < (uint32_t)li)))))));
				rank += (before & (j < n_all) & !((drop_mask >> sl) & 1ull)) ? 1u : 0u;
			}
		}
		if (dropped) // dropped hits keep the slots behind the kept ones (the table stays initialised)
			rank = n + (uint32_t)__popcll(drop_mask & grp_mask & ((1ull << lane) - 1ull));
		if (mine)
			store_hit_stream(hits + o + rank, h);
		if (!do_consensus || PGX_SORT_DBG(cv) == 3)
			continue;
		uint32_t rmv = 0, ntok = 0, sim = 0;
		if (kept) {
			uint32_t rcode[kRdpRegs];
#pragma unroll
			for (int b = 0; b < kRdpRegs; b++)
				rcode[b] = sw->rcode[g][b];
			sim = hit_simrank(cv, h); // (asked for first: its gather and the record's are in flight together)
			rmv = pair_matches(cv, (uint32_t)h.subject, rcode, rdp0, rdp1, &ntok);
		}
		if (PGX_SORT_DBG(cv) == 5) { // (probe: the agreement counts alone)
			if (__ballot(rmv + sim + ntok == 0xFFFFFFF0u) && li == 0)
				recs[r].hit = 0;
			continue;
		}
		// Consensus:186-204 is an order-dependent selection (ArgmaxState).  When every hit of the read has
		// the same lineage token count c >= 1 -- the normal case -- it has a closed form: after the first
		// step maxcount stays c, so the winner is, among the hits whose agreement count is the text-order
		// maximum, the first one in table order whose pident text is greatest.  That is two reductions
		// over the read's lanes; anything else takes the literal walk below.
		const unsigned long long kept_mask = __ballot(kept) & grp_mask;
		const uint32_t c0 = __shfl(ntok, kept_mask ? __ffsll((unsigned long long)kept_mask) - 1 : slot0);
		const bool odd = kept && (ntok != c0 || ntok == 0u || rmv >= 100000000u || sim >= (1u << 25));
		const unsigned long long odd_mask = __ballot(odd);
		const unsigned long long gmask = G == 64 ? ~0ull : (g ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull);
		const bool slow = (odd_mask & gmask) != 0ull;
		if (n && !present && li == 0) {
			recs[r].hit = -2;
			recs[r].matches = 0;
		}
		const uint32_t k32 = kept ? dec_str_key32(rmv) : 0u; // >= 1 for any value
		uint32_t kmax = k32;
#pragma unroll
		for (int m = 1; m < G; m <<= 1)
			kmax = max(kmax, (uint32_t)__shfl_xor(kmax, m));
		const bool elig = kept && k32 == kmax;
		if (PGX_SORT_DBG(cv) == 6) { // (probe: + the first reduction)
			if (__ballot(elig && slow && !present) && li == 0)
				recs[r].hit = 0;
			continue;
		}
		if (PGX_SORT_DBG(cv) != 4) {
			const uint32_t key2 = elig ? ((sim << 6) | (63u - rank)) + 1u : 0u;
			uint32_t top = key2;
#pragma unroll
			for (int m = 1; m < G; m <<= 1)
				top = max(top, (uint32_t)__shfl_xor(top, m));
			if (elig && key2 == top && present && !slow) {
				recs[r].hit = (int32_t)(o + rank);
				recs[r].matches = (int32_t)rmv;
			}
		}
		if (odd_mask) {
			const bool any_huge = __ballot(kept && (rmv >= 100000000u || ntok >= 100000000u || sim >= (1u << 25))) != 0ull;
			lds_fence(); // every lane is done with the keys: the bytes become the per-rank arrays
			if (any_huge) {
				// counts of nine digits and more: 64-bit text keys, the literal walk
				if (kept) {
					sw->b.rm[slot0 + rank] = rmv;
					sw->b.sim[slot0 + rank] = sim;
					sw->b.krm[slot0 + rank] = dec_str_key(rmv);
					sw->b.kcnt[slot0 + rank] = dec_str_key(ntok);
				}
				lds_fence();
				if (li == 0 && n && slow && present) {
					ArgmaxState am;
					am.cursim = r == 0 ? cv.simrank_undef : cv.simrank_zero;
					for (uint32_t k = 0; k < n; k++)
						am.step_keys((int32_t)(o + k), sw->b.rm[slot0 + k], sw->b.krm[slot0 + k], sw->b.kcnt[slot0 + k],
							     sw->b.sim[slot0 + k]);
					recs[r].hit = am.win;
					recs[r].matches = (int32_t)am.maxrm;
				}
			} else {
				// Lineages of different depth.  When the first hit of the table already holds the largest agreement
				// count, the walk of Consensus:186-204 reduces to a chain: the current choice h is replaced by the
				// next hit j with the largest count that is not dominated by it (token-count text greater, or pident
				// text greater), and after a replacement the remembered token count is j's own.  Every lane finds
				// the successor of its own hit (one pass over the per-rank arrays), lane 0 follows the chain.
				const uint32_t c32 = dec_str_key32(ntok);
				if (kept) {
					sw->c.kcnt[slot0 + rank] = c32;
					sw->c.sim_e[slot0 + rank] = sim | (elig ? 0x80000000u : 0u);
					sw->c.krm[slot0 + rank] = k32;
					sw->c.rm[slot0 + rank] = rmv;
				}
				lds_fence();
				if (n && slow && present) {
					const bool top_is_max = (sw->c.sim_e[slot0] >> 31) != 0u;
					const uint32_t init_sim = r == 0 ? cv.simrank_undef : cv.simrank_zero;
					if (top_is_max) {
						uint32_t nx = kNoRead;
						for (uint32_t t = 0; t < n; t++) {
							const uint32_t a = sw->c.kcnt[slot0 + t], b2 = sw->c.sim_e[slot0 + t];
							const bool cand = kept && t > rank && (b2 >> 31) && (a > c32 || (b2 & 0x7FFFFFFFu) > sim);
							nx = (cand && nx == kNoRead) ? t : nx;
						}
						if (kept)
							sw->c.nxt[slot0 + rank] = nx;
						lds_fence();
						if (li == 0) {
							uint32_t w = kNoRead;
							if (sw->c.rm[slot0] > 0u) {
								w = 0; // the first step takes hit 0 and its token count
							} else {
								// no agreement anywhere: the first hit that beats the initial (count "0", similarity) pair
								for (uint32_t t = 0; t < n && w == kNoRead; t++)
									if (sw->c.kcnt[slot0 + t] > 1u /* key of "0" */ || (sw->c.sim_e[slot0 + t] & 0x7FFFFFFFu) > init_sim)
										w = t;
							}
							if (w != kNoRead) {
								for (uint32_t nxw = sw->c.nxt[slot0 + w]; nxw != kNoRead; nxw = sw->c.nxt[slot0 + w])
									w = nxw;
								recs[r].hit = (int32_t)(o + w);
								recs[r].matches = (int32_t)sw->c.rm[slot0 + w];
							} else {
								recs[r].hit = -1;
								recs[r].matches = 0;
							}
						}
					} else if (li == 0) {
						// the literal walk on the 32-bit keys
						uint32_t kr = 1u /* key of "0" */, mrm = 0, kc = 1u, cs = init_sim;
						int32_t win = -1;
						for (uint32_t k = 0; k < n; k++) {
							const uint32_t vkrm = sw->c.krm[slot0 + k], vrm = sw->c.rm[slot0 + k], vkc = sw->c.kcnt[slot0 + k],
								       vs = sw->c.sim_e[slot0 + k] & 0x7FFFFFFFu;
							if (vkrm > kr) {
								kr = vkrm;
								mrm = vrm;
								win = (int32_t)(o + k);
								cs = vs;
							}
							if ((vkc > kc || cs < vs) && vrm == mrm) {
								kc = vkc;
								win = (int32_t)(o + k);
								cs = vs;
							}
						}
						recs[r].hit = win;
						recs[r].matches = (int32_t)mrm;
					}
				}
			}
		}
		lds_fence();
	}
	if (G != 64 && lane == 0 && chunk_used < 64u && chunk_used > 0)
		for (uint32_t k = chunk_used; k < 64u; k++)
			next_list[chunk_base + k] = kNoRead;
}

// consensus for the big reads and for file mode (hits already in order): one lane per read
__global__ void k_consensus_serial(const pgx_hit *__restrict__ hits, const uint32_t *__restrict__ off,
				   const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ list, uint32_t n_list,
				   ConsView cv, pgx_consensus_rec *__restrict__ recs)
{
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_list)
		return;
	const uint32_t r = list ? list[t] : t;
	const uint32_t o = off[r], n = cnt[r];
	pgx_consensus_rec rec;
	rec.hit = n ? -1 : -2;
	rec.matches = 0;
	if (cv.rdp_present && !cv.rdp_present[r])
		rec.hit = -2;
	if (rec.hit == -1) {
		const uint32_t r0 = cv.rdp_off[r], r1 = cv.rdp_off[r + 1];
		ArgmaxState am;
		am.cursim = r == 0 ? cv.simrank_undef : cv.simrank_zero;
		for (uint32_t k = 0; k < n; k++) {
			const pgx_hit h = hits[o + k];
			uint32_t c;
			const uint32_t rm = hit_rank_matches(cv, (uint32_t)h.subject, r0, r1, &c);
			am.step((int32_t)(o + k), rm, c, hit_simrank(cv, h));
		}
		rec.hit = am.win;
		rec.matches = (int32_t)am.maxrm;
	}
	recs[r] = rec;
}

// ------------------------------------------------------------------------------------------ host drivers
// a window of a hit table as a table of its own (pgx_hits_slice): offsets from 0, read numbers from 0
__global__ void k_slice_offsets(const uint32_t *__restrict__ off, const uint32_t *__restrict__ cnt, uint32_t first, uint32_t n,
				 uint32_t *__restrict__ off_out, uint32_t *__restrict__ cnt_out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i <= n)
		off_out[i] = off[first + i] - off[first];
	if (i < n)
		cnt_out[i] = cnt[first + i];
}

__global__ void k_slice_hits(const pgx_hit *__restrict__ in, uint64_t base, uint64_t n_slots, uint32_t first, pgx_hit *__restrict__ out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * blockDim.x) {
		pgx_hit h = in[base + i];
		h.read -= (int32_t)first;
		out[i] = h;
	}
}

static thread_local pgx_stage_times t_times; // stage times of the calling thread's last pipeline call

static DbView db_view(const pgx_db *db)
{
	DbView v;
	v.words = db->d_words.data();
	v.amb = db->has_amb ? db->d_amb.data() : nullptr;
	v.seq_off = db->d_seq_off.data();
	v.blk_subj = db->d_blk_subj.data();
	v.blk_info = db->d_blk_info.data();
	v.post_ctx = db->d_post_ctx.data();
	static const bool no_amb_blk = getenv("PGX_NO_AMB_BLK") != nullptr; // (measurement aid, read once per process)
	v.amb_blk = db->has_amb && !no_amb_blk ? db->d_amb_blk.data() : nullptr;
	v.bucket_off = db->d_bucket_off.data();
	v.postings = db->d_postings.data();
	v.n_seq = (uint32_t)db->n_seq;
	v.n_bases = db->n_bases;
	v.bits = db->index_bits;
	v.gapped = db->ungapped ? 0 : 1;
	v.deep_from = gapped_deep_from();
#ifdef PGX_STAGE_PROBES
	v.dbg_stop = getenv("PGX_SEED_STOP") ? atoi(getenv("PGX_SEED_STOP")) : 0; // (measurement builds: the kernel truncated after a stage)
#else
	v.dbg_stop = 0;
#endif
	return v;
}

static ReadsView reads_view(const pgx_reads *rd)
{
	ReadsView v;
	v.fwd = rd->d_fwd.data();
	v.rc = rd->d_rc.data();
	v.fwd_amb = rd->has_amb ? rd->d_fwd_amb.data() : nullptr;
	v.rc_amb = rd->has_amb ? rd->d_rc_amb.data() : nullptr;
	v.len = rd->d_len.data();
	v.woff = rd->d_woff.data();
	v.n = (uint32_t)rd->n;
	v.list = nullptr;
	v.dustwin_f = rd->has_dust ? rd->dustb.win_f.data() : nullptr;
	v.dustwin_r = rd->has_dust ? rd->dustb.win_r.data() : nullptr;
	v.dust_any = rd->has_dust ? rd->dustb.any.data() : nullptr;
	return v;
}

static ConsView cons_view(const pgx_db *db, const pgx_rdp *rdp)
{
	ConsView cv;
	memset(&cv, 0, sizeof cv);
#ifdef PGX_STAGE_PROBES
	cv.dbg = getenv("PGX_SORT_STOP") ? atoi(getenv("PGX_SORT_STOP")) : 0;
#else
	cv.dbg = 0;
#endif
	if (db && db->bound) {
		cv.subj_tok_off = db->d_subj_tok_off.data();
		cv.subj_tok = db->d_subj_tok.data();
		cv.tok_rank = db->d_tok_rank.data();
		cv.subj_pairs = db->d_subj_pairs.data();
		cv.pair_words = db->pair_words;
		cv.simrank_lut = db->d_simrank_lut.data();
		cv.simrank_len = db->d_simrank_len.data();
		cv.simrank_undef = db->simrank_undef;
		cv.simrank_zero = db->simrank_zero;
	}
	cv.np_max = db ? db->max_pairs : 15;
	cv.nr_max = rdp ? rdp->max_trip : kRdpRegs;
	if (rdp) {
		cv.rdp_off = rdp->d_off.data();
		cv.rdp_name = rdp->d_name.data();
		cv.rdp_rank = rdp->d_rank.data();
		cv.rdp_present = rdp->d_present.data();
		cv.rdp_code = rdp->d_code.data();
	}
	return cv;
}

// stage boundaries on the pipeline's stream; read only after the one synchronisation at the end of the step
struct StageEvents {
	static constexpr int kN = 8;
	hipEvent_t e[kN] = {};
	bool ok = false;
	int init()
	{
		if (ok)
			return 0;
		for (auto &x : e)
			PGX_HIP(hipEventCreate(&x));
		ok = true;
		return 0;
	}
	void mark(int i, hipStream_t s) { (void)hipEventRecord(e[i], s); }
	float ms(int a, int b) const
	{
		float t = 0;
		(void)hipEventElapsedTime(&t, e[a], e[b]);
		return t;
	}
};

// Buffers that persist across pipeline calls: the steady state of a batch loop allocates nothing.
// ------------------------------------------------------------------------------------------ pieces back to reads
// A batch whose reads hold long N runs was searched piece by piece (seqdb.hip: reads_build_pieces).  Pieces of a read
// are consecutive, so the exclusive scan of the per-piece counts is already the layout of the per-read table: the
// hits of every piece are moved there with the read's number and the piece's offset on the query coordinates.
__global__ __launch_bounds__(256) void k_merge_pieces(const pgx_hit *__restrict__ scratch, const uint32_t *__restrict__ piece_start,
						      const uint32_t *__restrict__ piece_off, const uint32_t *__restrict__ parent,
						      const uint32_t *__restrict__ qoff, uint32_t n_pieces, pgx_hit *__restrict__ hits,
						      unsigned long long hits_cap, unsigned long long scratch_cap)
{
	const uint32_t waves = gridDim.x * (blockDim.x / 64), lane = threadIdx.x & 63;
	for (uint32_t s = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); s < n_pieces; s += waves) {
		const uint32_t o = piece_off[s], cnt = piece_off[s + 1] - o;
		if (cnt == 0 || (unsigned long long)o + cnt > hits_cap)
			continue;
		const uint32_t st = piece_start[s];
		if (st != kFragmented && (unsigned long long)st + cnt > scratch_cap)
			continue;
		const pgx_hit *src = st == kFragmented ? hits + o : scratch + st; // fragmented pieces were scattered in place
		const int32_t read = (int32_t)parent[s], shift = (int32_t)qoff[s];
		for (uint32_t i = lane; i < cnt; i += 64) {
			pgx_hit h = src[i];
			h.read = read;
			h.qstart += shift;
			h.qend += shift;
			hits[o + i] = h;
		}
	}
}

__global__ void k_piece_ranges(const uint32_t *__restrict__ piece_first, const uint32_t *__restrict__ piece_off, uint32_t n_reads,
			       uint32_t *__restrict__ read_off, uint32_t *__restrict__ read_cnt, uint32_t *__restrict__ read_start)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_reads)
		return;
	const uint32_t o = piece_off[piece_first[r]];
	read_off[r] = o;
	if (r < n_reads) {
		read_cnt[r] = piece_off[piece_first[r + 1]] - o;
		read_start[r] = kFragmented; // the ordering kernels take every read from the grouped table
	}
}

// Everything a search through one database handle needs between and during calls: owned by the handle (pgx_db::work),
// so two handles searched from two host threads share nothing; searches through ONE handle are serialised by its mutex.
// The steady state of a batch loop allocates nothing, and the host waits for the device once per step.
struct Workspace {
	hipStream_t stream = nullptr;
	StageEvents ev;
	DevBuf<unsigned long long> counters; // [0..7] seed stage (OutView), [8] reads passed to the big-read path, [9] 33..64-hit list
	unsigned long long *h_counters = nullptr; // pinned mirror of counters + the gapped stage's list count
	DevBuf<pgx_hit> scratch, ovf;
	DevBuf<uint8_t> scratch_key, ovf_key; // gapped mode: the gapped stage's work estimate per slot of the two tables
	DevBuf<uint8_t> scratch_reg;          // gapped mode: database region of the anchor, per slot of the main table
	DevBuf<uint32_t> partial, cursor, big_list, read_start, mid_list;
	DevBuf<uint32_t> piece_cnt, piece_off, parent_start; // batches searched piece by piece
	DevBuf<pgx_consensus_rec> recs;
	GappedWork gapped;
	DustBufs dust; // S3d computed inside a search (pgx_db_set_dust_each_search) lands here, never in the caller's batch
	pgx_hits hits; // used when the caller does not keep the hit table
	uint64_t hit_cap_hint = 0, ovf_cap_hint = 0;
	double table_per_read_hint = 0.0; // the hit table belongs to the caller and is made per call: sized per read, not by the largest batch seen
	                                  // (a 2 M-read batch after 10 M-read ones used to allocate -- and map -- the 10 M table: 25 ms)
	pgx_stage_times times;
	int device = -1;
	~Workspace()
	{
		// (the handle is closed by its owner while the runtime is alive; nothing here runs from a static destructor)
		if (h_counters)
			(void)hipHostFree(h_counters);
		if (stream)
			(void)hipStreamDestroy(stream);
	}
};
constexpr int kNCounters = 12;

static int workspace_of(pgx_db *db, Workspace **out)
{
	if (!db->work) {
		auto w = std::make_shared<Workspace>();
		PGX_HIP(hipGetDevice(&w->device));
		PGX_HIP(hipStreamCreate(&w->stream));
		PGX_HIP(hipHostMalloc((void **)&w->h_counters, (kNCounters + 3) * sizeof(unsigned long long)));
		PGX_TRY(w->ev.init());
		PGX_TRY(w->counters.alloc(kNCounters, 0, 0, true));
		db->work = w;
	}
	*out = static_cast<Workspace *>(db->work.get());
	return 0;
}

// search + gapped stage + group + order (+ consensus when rdp != null). d_recs: device array of n_reads records.
int search_pipeline(pgx_db *db, pgx_reads *rd, const pgx_rdp *rdp, pgx_hits *out, pgx_consensus_rec *d_recs)
{
	PGX_TRY(require_device());
	if (rdp && !db->bound)
		return fail(PGX_E_ARG, "consensus needs pgx_db_bind_taxonomy() first");
	if (rdp && rdp->n != rd->n)
		return fail(PGX_E_ARG, "RDP stream holds %lld reads, batch holds %lld", (long long)rdp->n, (long long)rd->n);
	std::lock_guard<std::mutex> lock(db->search_mu);
	static const bool laps_on = getenv("PGX_CALL_LAPS") != nullptr; // host time of the call's phases, nothing synchronised for it
	const auto lap_t0 = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (laps_on)
			fprintf(stderr, "[pgx lap] %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - lap_t0).count(), what);
	};
	trace_point("search_pipeline: entered");
	Workspace *wsp = nullptr;
	PGX_TRY(workspace_of(db, &wsp));
	Workspace &ws = *wsp;
	hipStream_t st = ws.stream;
	pgx_stage_times &tm = ws.times;
	memset(&tm, 0, sizeof tm);
	const uint64_t n = (uint64_t)rd->n;
	out->n_reads = rd->n;
	PGX_TRY(out->d_read_cnt.ensure(n + 1));
	PGX_TRY(out->d_read_off.ensure(n + 1));
	if (n == 0) {
		out->n_hits = 0;
		t_times = tm;
		return 0;
	}
	index_check(db, "search_pipeline");
	const DbView dv = db_view(db);
	// the batch the seed stage runs on: the reads, or their pieces (reads with long N runs, engine.hpp)
	const bool split = rd->pieces && rd->pieces->n > 0; // (no piece at all: every read is N's; search the reads as they are)
	const pgx_reads *sr = split ? rd->pieces.get() : rd;
	const uint64_t ns = (uint64_t)sr->n;
	if (split) {
		PGX_TRY(ws.piece_cnt.ensure(ns + 1));
		PGX_TRY(ws.piece_off.ensure(ns + 1));
		PGX_TRY(ws.parent_start.ensure(n));
	}
	ReadsView rv = reads_view(sr);
	if (!db->dust)
		rv.dustwin_f = rv.dustwin_r = nullptr; // `-dust no`
	const int grid = (int)std::min<uint64_t>((n + kWavesPerBlock - 1) / kWavesPerBlock, 256ull * 8);

	// capacities are guesses kept from earlier calls; every kernel checks them, the counters say at the end of the
	// step whether one was too small, and the step is then repeated with larger tables
	uint64_t cap = std::max<uint64_t>(std::max<uint64_t>(std::max(n, ns) * 40, 1 << 16) + 256ull * 8 * kWavesPerBlock * kChunk, ws.hit_cap_hint);
	uint64_t ovf_cap = std::max<uint64_t>(std::max<uint64_t>(std::max(n, ns) / 4, 1 << 16), ws.ovf_cap_hint);
	uint64_t table_cap = std::max<uint64_t>(std::max<uint64_t>(std::max(n, ns) * 36, 1 << 16), (uint64_t)((double)std::max(n, ns) * ws.table_per_read_hint) + 1);
	DevBuf<pgx_hit> &scratch = ws.scratch, &ovf = ws.ovf;
	DevBuf<uint32_t> &read_start = ws.read_start;
	PGX_TRY(read_start.ensure(std::max(n, ns)));
	unsigned long long *h_cnt = ws.h_counters;
	// reads are searched class by class (engine.hpp: pgx_reads::classes); a database with ambiguity letters makes every
	// class ambiguity-aware
	const std::vector<pgx_reads::SearchClass> &classes = sr->classes;
	bool long_reads = false; // a class whose flags do not fit registers (reads above 512 bases)
	for (auto &c : classes)
		long_reads = long_reads || c.words == 0;
	const uint32_t n_part = (uint32_t)((ns + kScanBlock * kScanItems - 1) / (kScanBlock * kScanItems));
	PGX_TRY(ws.partial.ensure(n_part));
	PGX_TRY(ws.cursor.ensure(ns));
	PGX_TRY(ws.big_list.ensure(n));
	PGX_TRY(ws.mid_list.ensure(n + 64ull * kWavesPerBlock * 256 * 8)); // + one open chunk per wave
	trace_point("search_pipeline: lists ready");
	const ConsView cv = cons_view(db, rdp);
	const int lds_ok = rd->max_len <= 65535 ? 1 : 0;
	uint64_t H = 0, H_ovf = 0;
	// S3d inside the search (pgx_db_set_dust_each_search): the window bits of the batch computed again, on this stream
	ws.ev.mark(6, st);
	const bool dust_now = db->dust && db->dust_each_search;
	if (dust_now) {
		PGX_TRY(reads_dust_again(sr, ws.dust, st));
		if (sr->has_dust) {
			rv.dustwin_f = ws.dust.win_f.data();
			rv.dustwin_r = ws.dust.win_r.data();
			rv.dust_any = ws.dust.any.data();
		}
	}
	ws.ev.mark(7, st);
	for (int attempt = 0;; attempt++) {
		if (attempt > 8)
			return fail(PGX_E_LIMIT, "hit tables did not settle after %d attempts", attempt);
		PGX_TRY(scratch.ensure(cap));
		PGX_TRY(ovf.ensure(ovf_cap));
		trace_point("search_pipeline: seed tables ready");
		PGX_TRY(out->d_hits.ensure(table_cap));
		trace_point("search_pipeline: hit table ready");
		cap = scratch.n;
		ovf_cap = ovf.n;
		if (dv.gapped) {
			// (16 bytes more: the binning kernels of the gapped stage read the keys as 16-byte words; 0xFF = no record in the slot --
			// the tail of a wavefront's chunk of the table)
			PGX_TRY(ws.scratch_key.ensure(cap + 16));
			PGX_TRY(ws.scratch_reg.ensure(cap + 16));
			PGX_TRY(ws.ovf_key.ensure(ovf_cap));
			PGX_HIP(hipMemsetAsync(ws.scratch_key.data(), 0xFF, cap + 16, st));
		}
		table_cap = out->d_hits.n;
		PGX_HIP(hipMemsetAsync(ws.counters.data(), 0, kNCounters * sizeof(unsigned long long), st));
		OutView ov;
		ov.main = scratch.data();
		ov.ovf = ovf.data();
		ov.main_key = ws.scratch_key.data();
		ov.ovf_key = ws.ovf_key.data();
		ov.main_reg = ws.scratch_reg.data();
		ov.reg_shift = gapped_region_shift(db->n_bases);
		ov.main_cap = cap;
		ov.ovf_cap = ovf_cap;
		ov.counters = ws.counters.data();
		uint32_t *big_count = reinterpret_cast<uint32_t *>(ws.counters.data() + 8), *mid_count = reinterpret_cast<uint32_t *>(ws.counters.data() + 9);
		trace_point("search_pipeline: tables ready");
		lap("tables ready");
		ws.ev.mark(0, st);
		uint32_t *rc_ptr = split ? ws.piece_cnt.data() : out->d_read_cnt.data(), *rs_ptr = read_start.data();
		for (const pgx_reads::SearchClass &c0 : classes) {
			pgx_reads::SearchClass c = c0;
			c.amb = c.amb || db->has_amb;
			ReadsView cvw = rv;
			const uint64_t cn = c.count;
			cvw.n = (uint32_t)cn;
			cvw.list = c.listed ? sr->d_class_list.data() + c.off : nullptr;
			const int per_block = kWavesPerBlock * (c.words > 0 ? 2 : 1);
			const dim3 g((unsigned)std::min<uint64_t>((cn + per_block - 1) / per_block, 256ull * 8)), b(64 * kWavesPerBlock);
#define PGX_SEED_LAUNCH2(A, W, D)                                                                                                  \
	do {                                                                                                                       \
		if (c.listed)                                                                                                      \
			hipLaunchKernelGGL((k_seed_extend<A, W, true, D>), g, b, 0, st, dv, cvw, ov, rc_ptr, rs_ptr);             \
		else                                                                                                               \
			hipLaunchKernelGGL((k_seed_extend<A, W, false, D>), g, b, 0, st, dv, cvw, ov, rc_ptr, rs_ptr);            \
	} while (0)
#define PGX_SEED_LAUNCH(A, W)                                                                                                      \
	do {                                                                                                                       \
		if (c.dust && cvw.dustwin_f)                                                                                       \
			PGX_SEED_LAUNCH2(A, W, true);                                                                              \
		else                                                                                                               \
			PGX_SEED_LAUNCH2(A, W, false);                                                                             \
	} while (0)
			if (c.amb && c.words == 3)
				PGX_SEED_LAUNCH(true, 3);
			else if (c.amb && c.words == 5)
				PGX_SEED_LAUNCH(true, 5);
			else if (c.amb && c.words == 8)
				PGX_SEED_LAUNCH(true, 8);
			else if (c.amb)
				PGX_SEED_LAUNCH(true, 0);
			else if (c.words == 3)
				PGX_SEED_LAUNCH(false, 3);
			else if (c.words == 5)
				PGX_SEED_LAUNCH(false, 5);
			else if (c.words == 8)
				PGX_SEED_LAUNCH(false, 8);
			else
				PGX_SEED_LAUNCH(false, 0);
#undef PGX_SEED_LAUNCH
#undef PGX_SEED_LAUNCH2
		}
		PGX_HIP(hipGetLastError());
		trace_point("k_seed_extend");
		ws.ev.mark(1, st);

		// spec v2: the initial HSPs become gapped alignments, in place (gapped.hip)
		if (dv.gapped) {
			ReadsView all = rv;
			all.n = (uint32_t)ns;
			PGX_TRY(gapped_stage(dv, all, scratch.data(), ws.scratch_key.data(), rs_ptr, rc_ptr, ovf.data(), ws.ovf_key.data(),
					     ws.counters.data() + 4, ovf_cap, long_reads, cap, (int)sr->max_len, ws.gapped, st, ws.counters.data(), ws.scratch_reg.data()));
			trace_point("gapped_stage");
		}
		ws.ev.mark(2, st);

		// group by read: exclusive scan of the per-read counts; only the overflow hits need a scatter
		// (per piece when the batch was searched piece by piece: the scan of the piece counts is the per-read layout too)
		const uint32_t *unit_cnt = split ? ws.piece_cnt.data() : out->d_read_cnt.data();
		uint32_t *unit_off = split ? ws.piece_off.data() : out->d_read_off.data();
		hipLaunchKernelGGL(k_scan_partials, dim3(n_part), dim3(kScanBlock), 0, st, unit_cnt, ns, ws.partial.data());
		hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kScanBlock), 0, st, ws.partial.data(), n_part);
		hipLaunchKernelGGL(k_scan_final, dim3(n_part), dim3(kScanBlock), 0, st, unit_cnt, ns, ws.partial.data(), unit_off);
		// (overflow hits are rare; their count is only known on the device, so the scatter always runs)
		PGX_HIP(hipMemsetAsync(ws.cursor.data(), 0, ns * sizeof(uint32_t), st));
		hipLaunchKernelGGL(k_scatter_hits, dim3(256), dim3(256), 0, st, ovf.data(), ws.counters.data() + 4, ovf_cap, unit_off,
				   ws.cursor.data(), out->d_hits.data(), table_cap);
		PGX_HIP(hipGetLastError());
		const uint32_t *sort_start = read_start.data();
		if (split) {
			hipLaunchKernelGGL(k_merge_pieces, dim3((unsigned)std::min<uint64_t>((ns + 3) / 4, 256ull * 32)), dim3(256), 0, st, scratch.data(),
					   read_start.data(), ws.piece_off.data(), rd->d_piece_parent.data(), rd->d_piece_qoff.data(), (uint32_t)ns,
					   out->d_hits.data(), table_cap, cap);
			hipLaunchKernelGGL(k_piece_ranges, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, rd->d_piece_first.data(),
					   ws.piece_off.data(), (uint32_t)n, out->d_read_off.data(), out->d_read_cnt.data(), ws.parent_start.data());
			PGX_HIP(hipGetLastError());
			sort_start = ws.parent_start.data();
		}
		trace_point("group");
		ws.ev.mark(3, st);

		// per-read order (+ consensus): two reads per wavefront first; reads with 33..64 hits go through `mid_list` to
		// the one-read-per-wave launch, which passes reads with more than 64 hits on to `big_list`
		const int grid2 = (int)std::min<uint64_t>((n + 2 * kWavesPerBlock - 1) / (2 * kWavesPerBlock), 256ull * 8);
		hipLaunchKernelGGL(k_sort_consensus<32>, dim3(grid2), dim3(64 * kWavesPerBlock), 0, st, out->d_hits.data(), scratch.data(),
				   sort_start, out->d_read_off.data(), out->d_read_cnt.data(), (unsigned long long)table_cap, (unsigned long long)cap, (uint32_t)n,
				   (const uint32_t *)nullptr, (const uint32_t *)nullptr, cv, rdp ? 1 : 0, lds_ok, d_recs, ws.mid_list.data(), mid_count);
		trace_point("k_sort_consensus<32>");
		hipLaunchKernelGGL(k_sort_consensus<64>, dim3(grid), dim3(64 * kWavesPerBlock), 0, st, out->d_hits.data(), scratch.data(),
				   sort_start, out->d_read_off.data(), out->d_read_cnt.data(), (unsigned long long)table_cap, (unsigned long long)cap, (uint32_t)n,
				   ws.mid_list.data(), mid_count, cv, rdp ? 1 : 0, lds_ok, d_recs, ws.big_list.data(), big_count);
		PGX_HIP(hipGetLastError());
		trace_point("k_sort_consensus<64>");
		ws.ev.mark(4, st);

		// the one wait of the step: counters (+ the gapped stage's list count, + the last read offset = slots used)
		PGX_HIP(hipMemcpyAsync(h_cnt, ws.counters.data(), kNCounters * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
		h_cnt[kNCounters] = 0;
		h_cnt[kNCounters + 2] = 0;
		if (dv.gapped) {
			PGX_HIP(hipMemcpyAsync(h_cnt + kNCounters, ws.gapped.big_count.data(), 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st)); // lists A and B
			PGX_HIP(hipMemcpyAsync(h_cnt + kNCounters + 2, ws.gapped.big_count.data() + 4, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st)); // ([5]: below)
		}
		PGX_HIP(hipMemcpyAsync(h_cnt + kNCounters + 1, out->d_read_off.data() + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
		lap("step enqueued");
		PGX_HIP(hipStreamSynchronize(st));
		lap("step finished");
		H_ovf = h_cnt[4];
		H = h_cnt[5] + H_ovf;
		// (either list of the gapped stage's first tier may have outgrown its buffer: both have the capacity of list A's)
		const uint64_t gap_listed = std::max<uint64_t>((uint32_t)h_cnt[kNCounters], (uint32_t)(h_cnt[kNCounters] >> 32)), gap_cap = ws.gapped.big_list.n;
		bool again = false;
		if (h_cnt[0] > cap) {
			cap = h_cnt[0] + h_cnt[0] / 8;
			again = true;
		}
		if (H_ovf > ovf_cap) {
			ovf_cap = H_ovf + H_ovf / 8;
			again = true;
		}
		if (dv.gapped && gap_listed > gap_cap) {
			PGX_TRY(ws.gapped.big_list.ensure(gap_listed + gap_listed / 8));
			again = true;
		}
		if (!again && H > table_cap) { // (H is only exact once the seed tables held everything)
			table_cap = H + H / 16;
			again = true;
		}
		tm.attempts = attempt + 1;
		if (!again)
			break;
	}
	// h_cnt[0] counts reserved slots (chunks), h_cnt[5] the hits actually stored there
	ws.hit_cap_hint = std::max<uint64_t>(ws.hit_cap_hint, h_cnt[0] + h_cnt[0] / 16);
	ws.ovf_cap_hint = std::max<uint64_t>(ws.ovf_cap_hint, H_ovf + H_ovf / 16);
	ws.table_per_read_hint = std::max(ws.table_per_read_hint, (double)(H + H / 16) / (double)std::max(n, ns));
	out->n_hits = (int64_t)H;
	out->gapped = dv.gapped != 0;
	tm.hits = (int64_t)H;
	tm.probes = (int64_t)h_cnt[1];
	tm.postings = (int64_t)h_cnt[2];
	tm.candidates = (int64_t)h_cnt[3];
	tm.survivors = (int64_t)h_cnt[6];
	// HSPs the first tier listed: list A, plus ITS OWN appends to list B (list B's total also holds what the tier behind list A
	// passed on: an HSP that went A -> B was counted twice); batches with one list: list A is the count
	tm.gapped_wide = (int64_t)(uint32_t)h_cnt[kNCounters] + (int64_t)(uint32_t)(h_cnt[kNCounters + 2] >> 32);
	tm.seed_extend_ms = ws.ev.ms(0, 1);
	tm.gapped_ms = ws.ev.ms(1, 2);
	tm.group_ms = ws.ev.ms(2, 3);
	tm.sort_ms = ws.ev.ms(3, 4);
	tm.total_ms = ws.ev.ms(0, 4);
	tm.dust_ms = dust_now ? ws.ev.ms(6, 7) : 0.0f; // (events 6 and 7 bracket the DUST passes alone: a repeated attempt is not in it)
	tm.total_ms += tm.dust_ms;
	// (PGX_HIT_LIMIT lowers the limit: tests use it to exercise the callers' batch halving)
	const unsigned long long hit_limit = getenv("PGX_HIT_LIMIT") ? strtoull(getenv("PGX_HIT_LIMIT"), nullptr, 10) : (1ull << 32);
	if (H >= hit_limit)
		return fail(PGX_E_LIMIT, "%llu hits in one batch exceed the 32-bit slot limit: use smaller batches",
			    (unsigned long long)H);
	const uint32_t n_big = (uint32_t)h_cnt[8];
	if (n_big) {
		// more than 64 hits (or reads too long for the packed keys): segmented radix sorts, any size (bigreads.hip)
		const uint32_t *sort_start = split ? ws.parent_start.data() : read_start.data();
		ws.ev.mark(4, st);
		PGX_TRY(sort_big_reads(out->d_hits.data(), scratch.data(), sort_start, out->d_read_off.data(), out->d_read_cnt.data(),
				       ws.big_list.data(), n_big, dv.gapped != 0));
		if (rdp) {
			hipLaunchKernelGGL(k_consensus_serial, dim3((n_big + 63) / 64), dim3(64), 0, 0, out->d_hits.data(), out->d_read_off.data(),
					   out->d_read_cnt.data(), ws.big_list.data(), n_big, cv, d_recs);
			PGX_HIP(hipGetLastError());
		}
		ws.ev.mark(5, 0);
		PGX_HIP(hipDeviceSynchronize());
		const float extra = ws.ev.ms(4, 5);
		tm.sort_ms += extra;
		tm.total_ms += extra;
	}
	t_times = tm;
	lap("done");
	trace_point("search_pipeline: done");
	return 0;
}

// consensus over an existing hit table (pgx_consensus_batch): one lane per read
int consensus_device(const pgx_db *db, const pgx_hits *hits, const pgx_rdp *rdp, pgx_consensus_rec *d_out, pgx_stage_times *)
{
	if (!db->bound)
		return fail(PGX_E_ARG, "consensus needs pgx_db_bind_taxonomy() first");
	if (rdp->n != hits->n_reads)
		return fail(PGX_E_ARG, "RDP stream and hit table cover different read counts");
	const uint32_t n = (uint32_t)hits->n_reads;
	if (n == 0)
		return 0;
	const ConsView cv = cons_view(db, rdp);
	hipLaunchKernelGGL(k_consensus_serial, dim3((n + 63) / 64), dim3(64), 0, 0, hits->d_hits.data(), hits->d_read_off.data(),
			   hits->d_read_cnt.data(), (const uint32_t *)nullptr, n, cv, d_out);
	PGX_HIP(hipGetLastError());
	PGX_HIP(hipDeviceSynchronize());
	return 0;
}

} // namespace pgx

using namespace pgx;

extern "C" {

int pgx_blast_search(pgx_db *db, pgx_reads *reads, pgx_hits **out)
{
	if (!db || !reads || !out)
		return fail(PGX_E_ARG, "pgx_blast_search: null argument");
	pgx_hits *h = new pgx_hits();
	int rc = search_pipeline(db, reads, nullptr, h, nullptr);
	if (rc < 0) {
		delete h;
		return rc;
	}
	*out = h;
	return 0;
}

int pgx_classify_consensus(pgx_db *db, pgx_reads *reads, const pgx_rdp *rdp, pgx_hits **hits_out, pgx_consensus_rec *out,
			   int64_t cap)
{
	if (!db || !reads || !rdp)
		return fail(PGX_E_ARG, "pgx_classify_consensus: null argument");
	Workspace *ws = nullptr;
	{
		std::lock_guard<std::mutex> lock(db->search_mu);
		PGX_TRY(workspace_of(db, &ws));
		PGX_TRY(ws->recs.ensure((size_t)reads->n + 1));
	}
	pgx_hits *h = hits_out ? new pgx_hits() : &ws->hits;
	int rc = search_pipeline(db, reads, rdp, h, ws->recs.data());
	if (rc == 0 && out) {
		if (cap < reads->n)
			rc = fail(PGX_E_ARG, "record buffer too small");
		else
			rc = ws->recs.download(out, (size_t)reads->n);
		trace_point("pgx_classify_consensus: records downloaded");
	}
	if (hits_out) {
		if (rc < 0)
			delete h;
		else
			*hits_out = h;
	}
	return rc;
}

// BASELINE config 5 (BLAST + SOAP + RDP): the reference's Consensus takes the SOAP classification as a third stream,
// opens it, and never reads it (Consensus_BLAST_SOAP_RDP-1.1.pl:40-46; an option value that Perl holds false -- "" or
// "0" -- counts as not given).  Same here: the stream must be openable, and the records do not depend on it.
int pgx_classify_consensus_tri(pgx_db *db, pgx_reads *reads, const pgx_rdp *rdp, const char *soap_stream_path, pgx_hits **hits_out,
			       pgx_consensus_rec *out, int64_t cap)
{
	if (soap_stream_path && *soap_stream_path && strcmp(soap_stream_path, "0") != 0) {
		FILE *f = fopen(soap_stream_path, "r");
		if (!f)
			return fail(PGX_E_IO, "Error: Unable to open %s file.", soap_stream_path);
		fclose(f);
	}
	return pgx_classify_consensus(db, reads, rdp, hits_out, out, cap);
}

int pgx_db_set_dust_each_search(pgx_db *db, int on)
{
	if (!db)
		return fail(PGX_E_ARG, "pgx_db_set_dust_each_search: null argument");
	db->dust_each_search = on != 0;
	return 0;
}

int pgx_db_set_ungapped(pgx_db *db, int ungapped)
{
	if (!db)
		return fail(PGX_E_ARG, "pgx_db_set_ungapped: null argument");
	std::lock_guard<std::mutex> lock(db->search_mu);
	db->ungapped = ungapped != 0;
	return 0;
}

int pgx_db_set_dust(pgx_db *db, int dust)
{
	if (!db)
		return fail(PGX_E_ARG, "pgx_db_set_dust: null argument");
	std::lock_guard<std::mutex> lock(db->search_mu);
	db->dust = dust != 0;
	return 0;
}

void pgx_hits_close(pgx_hits *h) { delete h; }
int64_t pgx_hits_count(const pgx_hits *h) { return h ? h->n_hits : 0; }

int pgx_hits_copy(const pgx_hits *h, pgx_hit *out, int64_t cap)
{
	if (!h || !out || cap < h->n_hits)
		return fail(PGX_E_ARG, "pgx_hits_copy: bad argument");
	return h->d_hits.download(out, (size_t)h->n_hits);
}

int pgx_hits_slice(const pgx_hits *h, int64_t first_read, int64_t n_reads, pgx_hits **out)
{
	if (!h || !out || first_read < 0 || n_reads < 0 || first_read + n_reads > h->n_reads)
		return fail(PGX_E_ARG, "pgx_hits_slice: reads [%lld, +%lld) are not inside a table of %lld reads", (long long)first_read,
			    (long long)n_reads, (long long)(h ? h->n_reads : 0));
	PGX_TRY(require_device());
	return guard("pgx_hits_slice", [&]() -> int {
		std::unique_ptr<pgx_hits> s(new pgx_hits);
		s->n_reads = n_reads;
		s->gapped = h->gapped;
		PGX_TRY(s->d_read_off.alloc((size_t)n_reads + 1));
		PGX_TRY(s->d_read_cnt.alloc((size_t)n_reads + 1));
		uint32_t ends[2] = { 0, 0 };
		PGX_TRY(h->d_read_off.download(&ends[0], 1, (size_t)first_read));
		PGX_TRY(h->d_read_off.download(&ends[1], 1, (size_t)(first_read + n_reads)));
		const uint64_t n_slots = (uint64_t)ends[1] - ends[0];
		s->n_hits = (int64_t)n_slots;
		PGX_TRY(s->d_hits.alloc(n_slots ? n_slots : 1));
		hipLaunchKernelGGL(k_slice_offsets, dim3((unsigned)((n_reads + 1 + 255) / 256)), dim3(256), 0, 0, h->d_read_off.data(), h->d_read_cnt.data(),
				   (uint32_t)first_read, (uint32_t)n_reads, s->d_read_off.data(), s->d_read_cnt.data());
		if (n_slots)
			hipLaunchKernelGGL(k_slice_hits, dim3((unsigned)std::min<uint64_t>((n_slots + 255) / 256, 256ull * 32)), dim3(256), 0, 0, h->d_hits.data(),
					   (uint64_t)ends[0], n_slots, (uint32_t)first_read, s->d_hits.data());
		PGX_HIP(hipGetLastError());
		PGX_HIP(hipDeviceSynchronize());
		*out = s.release();
		return 0;
	});
}

int pgx_hits_read_offsets(const pgx_hits *h, int64_t *out, int64_t cap)
{
	if (!h || !out || cap < h->n_reads + 1)
		return fail(PGX_E_ARG, "pgx_hits_read_offsets: bad argument");
	std::vector<uint32_t> tmp((size_t)h->n_reads + 1);
	PGX_TRY(h->d_read_off.download(tmp.data(), tmp.size()));
	for (size_t i = 0; i < tmp.size(); i++)
		out[i] = tmp[i];
	return 0;
}

int pgx_hits_read_counts(const pgx_hits *h, int64_t *out, int64_t cap)
{
	if (!h || !out || cap < h->n_reads)
		return fail(PGX_E_ARG, "pgx_hits_read_counts: bad argument");
	std::vector<uint32_t> tmp((size_t)h->n_reads);
	PGX_TRY(h->d_read_cnt.download(tmp.data(), tmp.size()));
	for (size_t i = 0; i < tmp.size(); i++)
		out[i] = tmp[i];
	return 0;
}

int pgx_last_stage_times(pgx_stage_times *out)
{
	if (!out)
		return fail(PGX_E_ARG, "pgx_last_stage_times: null argument");
	*out = t_times;
	return 0;
}
}
