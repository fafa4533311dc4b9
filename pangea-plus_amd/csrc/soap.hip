// Classify, SOAP verb: `2bwt-builder ref.fa` + `soap -a reads -D ref.fa.index -o out -M 4 [-r] [-n] [-u]`
// (reference README.md:130-134, soap.man:29-83).  The reference ships soap only as a closed ELF;
// every rule here was observed from that binary and is pinned by tests/golden/soap/:
//   letters other than ACGT read as G (reference and reads); runs of >= 10 such letters in the
//   reference are cut out (no hit may overlap one); a hit must end before the last base of its
//   segment; reads shorter than 27 or with more than -n non-ACGT letters are not aligned; -M 4
//   keeps the full-length ungapped hits with the fewest (<= 2) mismatches on either strand.
//
// Paired-end runs (`-b B -2 unpaired -m MIN -x MAX`, soap.man:29-50): the rules observed on the ELF are listed at
// soap_run_paired below and pinned by tests/golden/soap/pe_*.
//
// k_soap_search  one wavefront per read, two passes over the same candidates (no per-read storage):
//   pass 0 finds the minimum mismatch count and how many placements reach it; the wavefront then reserves the read's rows
//   in the hit list with ONE atomic (a read's rows are contiguous: first_hit[r] .. + n_best[r]) and pass 1 writes them.
//   mode 5 (paired-end) keeps every placement with <= 2 mismatches.
// k_soap_pair  one wavefront per read pair: counts the valid pairs of placements at mismatch levels 0, 1, 2.
//   Seeds: three disjoint exact 16-mers per strand (pigeonhole for <= 2 mismatches) when the read
//   has >= 48 bases, else the 1 129 variants of the first 16-mer with <= 2 substitutions.  Seeds hit
//   the same direct-address 16-mer index as the BLAST verb; verification is XOR + popcount of the
//   packed read against the funnel-shifted reference window.
#include <algorithm>

#include "bitops.hpp"
#include "engine.hpp"

namespace pgx {

struct SoapHit {
	uint32_t read, subject, pos; // pos: 0-based leftmost position in the subject
	int32_t mis0, mis1;          // reference-oriented mismatch offsets, ascending (-1 = none)
	uint32_t strand_nmis;        // strand << 8 | nmis
};

struct SoapView {
	const uint64_t *words;
	const uint32_t *seq_off, *blk_subj, *bucket_off, *postings;
	const uint32_t *seg_lo, *seg_hi; // sorted segments (global base coordinates), n_seg of them
	uint32_t n_seg;
	int bits;
};

constexpr int kVariants = 1 + 16 * 3 + 120 * 9; // 16-mer with <= 2 substitutions

// v-th variant of a 16-mer (v = 0 is the k-mer itself)
__device__ __forceinline__ uint32_t kmer_variant(uint32_t kmer, int v)
{
	if (v == 0)
		return kmer;
	if (v <= 48) {
		int pos = (v - 1) / 3, delta = (v - 1) % 3 + 1;
		uint32_t b = (kmer >> (2 * pos)) & 3;
		return (kmer & ~(3u << (2 * pos))) | (((b + delta) & 3) << (2 * pos));
	}
	int u = v - 49, pi = u / 9, combo = u % 9;
	int i = 0, rem = pi; // pi-th pair i < j in lexicographic order
	while (rem >= 15 - i) {
		rem -= 15 - i;
		i++;
	}
	int j = i + 1 + rem;
	uint32_t bi = (kmer >> (2 * i)) & 3, bj = (kmer >> (2 * j)) & 3;
	uint32_t out = kmer & ~(3u << (2 * i)) & ~(3u << (2 * j));
	out |= ((bi + combo / 3 + 1) & 3) << (2 * i);
	out |= ((bj + combo % 3 + 1) & 3) << (2 * j);
	return out;
}

// mismatches of the read (nw words, L bases) against the reference at global position gp; stops
// counting above 2. Returns the count (3 = more than 2) and the first two offsets.
__device__ __forceinline__ int soap_mismatches(const uint64_t *rw, int L, const uint64_t *dbw, int64_t gp, int &m0, int &m1)
{
	int n = 0;
	m0 = m1 = -1;
	const int nw = (L + 31) >> 5;
	for (int w = 0; w < nw; w++) {
		uint64_t x = rw[w] ^ window64(dbw, gp + 32 * w);
		uint64_t m = (x | (x >> 1)) & kEven;
		if (w == nw - 1 && (L & 31))
			m &= (1ull << (2 * (L & 31))) - 1;
		while (m) {
			int pos = 32 * w + ((__ffsll((unsigned long long)m) - 1) >> 1);
			m &= m - 1;
			if (n == 0)
				m0 = pos;
			else if (n == 1)
				m1 = pos;
			if (++n > 2)
				return 3;
		}
	}
	return n;
}

// are read bases [a, a+16) free of mismatches at placement gp?
__device__ __forceinline__ bool seed_exact(const uint64_t *rw, const uint64_t *dbw, int64_t gp, int a)
{
	return (uint32_t)window64(rw, a) == (uint32_t)window64(dbw, gp + a);
}

__global__ __launch_bounds__(256) void k_soap_search(SoapView db, const uint64_t *__restrict__ fwd,
						     const uint64_t *__restrict__ rc, const uint32_t *__restrict__ len,
						     const uint32_t *__restrict__ woff, const uint8_t *__restrict__ skip,
						     uint32_t n_reads, SoapHit *__restrict__ hits, unsigned long long cap,
						     unsigned long long *__restrict__ hit_count, uint32_t *__restrict__ best_nmis,
						     uint32_t *__restrict__ n_best, unsigned long long *__restrict__ first_hit, int mode)
{
	const int lane = threadIdx.x & 63;
	for (uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n_reads; r += gridDim.x * 4) {
		const int L = (int)len[r];
		const uint32_t w0 = woff[r];
		if (skip[r] || L < 27) {
			if (lane == 0) {
				best_nmis[r] = 3;
				n_best[r] = 0;
				first_hit[r] = 0;
			}
			continue;
		}
		const bool exact_mode = L >= 48;
		const int so[3] = { 0, L / 3, 2 * (L / 3) };
		const int per_strand = exact_mode ? 3 : kVariants;
		const int P = 2 * per_strand;
		uint32_t best = 3;
		unsigned long long slot = 0; // pass 1: where the read's next row goes
		for (int pass = 0; pass < 2; pass++) {
			uint32_t c0 = 0, c1 = 0, c2 = 0;
			for (int pbase = 0; pbase < P; pbase += 64) {
				const int pid = pbase + lane;
				uint32_t cnt = 0, lo = 0, kmer = 0;
				int strand = 0, sidx = 0;
				if (pid < P) {
					strand = pid >= per_strand;
					sidx = pid - strand * per_strand;
					const uint64_t *rw = (strand ? rc : fwd) + w0;
					kmer = exact_mode ? kmer16(rw, so[sidx]) : kmer_variant(kmer16(rw, 0), sidx);
					uint32_t b = seed_bucket(kmer, db.bits);
					lo = db.bucket_off[b];
					cnt = db.bucket_off[(uint64_t)b + 1] - lo;
				}
				uint32_t incl = cnt;
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) {
					uint32_t t = __shfl_up(incl, d);
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
						int cand = o + step;
						uint32_t e = __shfl(excl, cand & 63);
						if (cand < 64 && e <= key)
							o = cand;
					}
					const uint32_t o_excl = __shfl(excl, o), o_lo = __shfl(lo, o), o_kmer = __shfl(kmer, o);
					const int o_strand = __shfl(strand, o), o_sidx = __shfl(sidx, o);
					int nm = 3, m0 = -1, m1 = -1;
					int64_t gp = 0;
					do {
						if (!active)
							break;
						const uint32_t p = db.postings[o_lo + (key - o_excl)];
						const uint64_t *rw = (o_strand ? rc : fwd) + w0;
						const int seed_off = exact_mode ? so[o_sidx] : 0;
						if (p < (uint32_t)seed_off)
							break;
						gp = (int64_t)p - seed_off;
						// the hit must lie inside one segment and end before its last base
						uint32_t sl = 0, sh = db.n_seg;
						while (sh - sl > 1) {
							uint32_t mid = sl + (sh - sl) / 2;
							if (db.seg_lo[mid] <= gp)
								sl = mid;
							else
								sh = mid;
						}
						if (db.n_seg == 0 || gp < db.seg_lo[sl] || gp + L >= (int64_t)db.seg_hi[sl])
							break;
						if (exact_mode) {
							// the seed itself must match (bucket collisions) and no earlier seed may:
							// the placement is reported through its first exact seed
							if (!seed_exact(rw, db.words, gp, seed_off))
								break;
							bool earlier = false;
							for (int j = 0; j < o_sidx; j++)
								earlier |= seed_exact(rw, db.words, gp, so[j]);
							if (earlier)
								break;
						} else if (kmer16(db.words, gp) != o_kmer) {
							break;
						}
						nm = soap_mismatches(rw, L, db.words, gp, m0, m1);
						// observed on the ELF, paired-end runs only: a read of exactly 32 bases is not placed where it has
						// two mismatches that both lie in its first 20 bases
						if (mode == 5 && L == 32 && nm == 2 && m1 < 20)
							nm = 3;
					} while (false);
					if (pass == 0) {
						c0 += nm == 0;
						c1 += nm == 1;
						c2 += nm == 2;
					} else {
						const bool emit = mode == 5 ? nm <= 2 : (uint32_t)nm == best;
						const unsigned long long vote = __ballot(emit);
						if (emit) {
							uint32_t s = db.blk_subj[(uint64_t)gp >> kBlkShift];
							while (db.seq_off[s + 1] <= (uint64_t)gp)
								s++;
							SoapHit h;
							h.read = r;
							h.subject = s;
							h.pos = (uint32_t)(gp - db.seq_off[s]);
							h.mis0 = m0;
							h.mis1 = m1;
							h.strand_nmis = ((uint32_t)o_strand << 8) | (uint32_t)nm;
							const unsigned long long g = slot + (unsigned long long)__popcll(vote & ((1ull << lane) - 1));
							if (g < cap)
								hits[g] = h;
						}
						slot += (unsigned long long)__popcll(vote);
					}
				}
			}
			if (pass == 0) {
				for (int d = 32; d >= 1; d >>= 1) {
					c0 += __shfl_down(c0, d);
					c1 += __shfl_down(c1, d);
					c2 += __shfl_down(c2, d);
				}
				c0 = __shfl(c0, 0);
				c1 = __shfl(c1, 0);
				c2 = __shfl(c2, 0);
				// -M 4: the fewest mismatches any placement has; -M 0 / 1 / 2 (soap.man:73-82, observed on the ELF):
				// the placements with exactly that many, whether or not a better one exists
				// mode 5 (paired-end): every placement with at most two mismatches (`best` = the fewest any has)
				if (mode == 4 || mode == 5)
					best = c0 ? 0u : (c1 ? 1u : (c2 ? 2u : 3u));
				else
					best = (mode == 0 ? c0 : (mode == 1 ? c1 : c2)) ? (uint32_t)mode : 3u;
				const uint32_t rows = mode == 5 ? c0 + c1 + c2 : (best == 0 ? c0 : (best == 1 ? c1 : (best == 2 ? c2 : 0u)));
				if (lane == 0) {
					best_nmis[r] = best;
					n_best[r] = rows;
					slot = rows ? atomicAdd(hit_count, (unsigned long long)rows) : 0ull;
					first_hit[r] = slot;
				}
				slot = __shfl(slot, 0);
				if (best == 3)
					break;
			}
		}
	}
}

// a valid pair of placements: one subject, opposite strands, (position of the '-' mate - position of the '+' mate) + the A
// mate's length inside [min_ins, max_ins]
__host__ __device__ inline bool soap_pair_geometry(const SoapHit &a, const SoapHit &b, int len_a, int min_ins, int max_ins)
{
	const uint32_t sa = a.strand_nmis >> 8, sb = b.strand_nmis >> 8;
	if (a.subject != b.subject || sa == sb)
		return false;
	const int64_t ins = (sa ? (int64_t)a.pos - (int64_t)b.pos : (int64_t)b.pos - (int64_t)a.pos) + len_a;
	return ins >= min_ins && ins <= max_ins;
}
__host__ __device__ inline uint32_t soap_pair_level(const SoapHit &a, const SoapHit &b)
{
	const uint32_t na = a.strand_nmis & 255, nb = b.strand_nmis & 255;
	return na > nb ? na : nb;
}

// one wavefront per read pair: how many valid pairs of placements there are with both mates at <= 0, <= 1, <= 2 mismatches
// -> the pair's level (the smallest with any; 3 = none) and its number of valid pairs at that level
__global__ __launch_bounds__(256) void k_soap_pair(const SoapHit *__restrict__ ha, const unsigned long long *__restrict__ first_a,
						   const uint32_t *__restrict__ n_a, const SoapHit *__restrict__ hb,
						   const unsigned long long *__restrict__ first_b, const uint32_t *__restrict__ n_b,
						   const uint32_t *__restrict__ len_a, uint32_t n_pairs, int min_ins, int max_ins,
						   uint32_t *__restrict__ level, unsigned long long *__restrict__ count)
{
	const int lane = threadIdx.x & 63;
	for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n_pairs; i += gridDim.x * 4) {
		const uint32_t na = n_a[i], nb = n_b[i];
		const SoapHit *a = ha + first_a[i], *b = hb + first_b[i];
		const int la = (int)len_a[i];
		unsigned long long c[3] = { 0, 0, 0 };
		const unsigned long long combos = (unsigned long long)na * nb;
		for (unsigned long long k = lane; k < combos; k += 64) {
			const SoapHit x = a[k / nb], y = b[k % nb];
			if (!soap_pair_geometry(x, y, la, min_ins, max_ins))
				continue;
			const uint32_t lv = soap_pair_level(x, y);
			c[0] += lv == 0;
			c[1] += lv <= 1;
			c[2]++;
		}
#pragma unroll
		for (int j = 0; j < 3; j++)
			for (int d = 32; d >= 1; d >>= 1)
				c[j] += __shfl_down(c[j], d);
		if (lane == 0) {
			const uint32_t lv = c[0] ? 0u : (c[1] ? 1u : (c[2] ? 2u : 3u));
			level[i] = lv;
			count[i] = lv < 3 ? c[lv] : 0ull;
		}
	}
}

static const char kLetters[5] = "ACGT";

static inline int host_base(const std::vector<uint64_t> &w, uint64_t p) { return (int)((w[p >> 5] >> (2 * (p & 31))) & 3); }

// one output row of soap (format observed from the ELF, see file header)
// paired-end runs order the two entries of a two-mismatch row differently (observed on the ELF, tests/golden/soap/pe_sweep_*):
// descending when one lies at or behind offset 2 s, s = 7 for reads under 32 bases, 10 under 39, else a third of the length
static inline int pe_descending_from(int L) { return 2 * (L < 32 ? 7 : (L < 39 ? 10 : L / 3)); }

static void soap_row(std::string &out, const std::string &name, const std::vector<int> &rd, const SoapHit &h, uint64_t nbest,
		     const pgx_db *db, int repeat, char mate = 'a', bool paired_run = false)
{
	const int L = (int)rd.size();
	const int strand = (int)(h.strand_nmis >> 8), nmis = (int)(h.strand_nmis & 255);
	out += name;
	out += '\t';
	for (int k = 0; k < L; k++)
		out += kLetters[rd[(size_t)k]];
	out += '\t';
	out.append((size_t)L, 'h');
	char buf[256];
	// (the subject id is appended as it is: ids are accepted up to 1 MiB, a fixed buffer would cut the row)
	snprintf(buf, sizeof buf, "\t%llu\t%c\t%d\t%c\t", (unsigned long long)nbest, mate, L, strand ? '-' : '+');
	out += buf;
	out += db->ids[h.subject];
	snprintf(buf, sizeof buf, "\t%u\t%d", h.pos + 1, nmis);
	out += buf;
	const uint64_t g0 = (uint64_t)db->h_seq_off[h.subject] + h.pos;
	int m[2] = { h.mis0, h.mis1 };
	if (nmis == 2 && m[1] >= (paired_run ? pe_descending_from(L) : L - 13))
		std::swap(m[0], m[1]); // single-end: entries descend when one lies in the last 13 bases
	for (int k = 0; k < nmis; k++) {
		const int q = (nmis == 1 && strand && m[k] == 0) ? -64 : 40;
		// the ELF keeps the offset in 8 bits and fetches the read base through it
		snprintf(buf, sizeof buf, "\t%c->%d%c%d", kLetters[host_base(db->h_words, g0 + (uint64_t)m[k])], m[k] & 255,
			 kLetters[rd[(size_t)(m[k] & 255)]], q);
		out += buf;
	}
	snprintf(buf, sizeof buf, "\t%dM\t", L);
	out += buf;
	int run = 0;
	bool first = true;
	for (int k = 0; k < L; k++) {
		const int rb = host_base(db->h_words, g0 + (uint64_t)k);
		if (rd[(size_t)k] == rb) {
			run++;
			continue;
		}
		if (first || run > 0)
			out += std::to_string(run);
		out += kLetters[rb];
		run = 0;
		first = false;
	}
	if (run > 0 || repeat != 1 || first)
		out += std::to_string(run);
	out += '\n';
}

} // namespace pgx

using namespace pgx;

extern "C" {

int pgx_soap_index(const char *fasta_path)
{
	if (!fasta_path)
		return fail(PGX_E_ARG, "2bwt-builder: reference FASTA required");
	std::string prefix = std::string(fasta_path) + ".index";
	return pgx_db_build(fasta_path, prefix.c_str());
}

} // extern "C"

namespace pgx {

// the reference as the ELF's index sees it: ambiguity codes folded to G, the segments between runs of >= 10 of them
struct SoapDb {
	pgx_db *db = nullptr;
	DevBuf<uint32_t> d_seg_lo, d_seg_hi;
	SoapView v{};
	~SoapDb() { pgx_db_close(db); }
};

static int soap_db_open(const char *prefix, SoapDb &sd)
{
	pgx_db *src = nullptr;
	PGX_TRY(db_read_host(prefix, &src));
	// segments between runs of >= 10 ambiguity codes (what 2bwt-builder cuts out)
	std::vector<uint32_t> seg_lo, seg_hi;
	for (int64_t s = 0; s < src->n_seq; s++) {
		const uint64_t a = src->h_seq_off[(size_t)s], e = src->h_seq_off[(size_t)s + 1];
		uint64_t start = a, k = a;
		while (k <= e) {
			uint64_t r = k;
			if (src->has_amb)
				while (r < e && ((src->h_amb[r >> 5] >> (2 * (r & 31))) & 1))
					r++;
			if (k == e || r - k >= 10) {
				if (k > start) {
					seg_lo.push_back((uint32_t)start);
					seg_hi.push_back((uint32_t)k);
				}
				start = r;
			}
			k = r > k ? r : k + 1;
		}
	}
	int rc = db_fold_amb_to_g(src, &sd.db);
	delete src;
	if (rc < 0)
		return rc;
	PGX_TRY(sd.d_seg_lo.alloc(seg_lo.size() ? seg_lo.size() : 1));
	PGX_TRY(sd.d_seg_lo.upload(seg_lo.data(), seg_lo.size()));
	PGX_TRY(sd.d_seg_hi.alloc(seg_hi.size() ? seg_hi.size() : 1));
	PGX_TRY(sd.d_seg_hi.upload(seg_hi.data(), seg_hi.size()));
	const pgx_db *db = sd.db;
	sd.v.words = db->d_words.data();
	sd.v.seq_off = db->d_seq_off.data();
	sd.v.blk_subj = db->d_blk_subj.data();
	sd.v.bucket_off = db->d_bucket_off.data();
	sd.v.postings = db->d_postings.data();
	sd.v.seg_lo = sd.d_seg_lo.data();
	sd.v.seg_hi = sd.d_seg_hi.data();
	sd.v.n_seg = (uint32_t)seg_lo.size();
	sd.v.bits = db->index_bits;
	return 0;
}

// one file of reads searched: the rows of read r are hits[first[r] .. first[r] + n_best[r]), in (subject, position, strand) order
struct SoapSearch {
	pgx_reads *rd = nullptr;
	std::vector<uint32_t> nn, best, n_best;
	std::vector<unsigned long long> first;
	std::vector<SoapHit> hits;
	DevBuf<SoapHit> d_hits;
	DevBuf<uint32_t> d_best, d_nbest;
	DevBuf<unsigned long long> d_first;
	std::vector<int> fw, rv; // bases of the read last unpacked
	~SoapSearch() { pgx_reads_close(rd); }
	void unpack(size_t r)
	{
		const int L = (int)rd->h_len[r];
		fw.resize((size_t)L);
		rv.resize((size_t)L);
		for (int k = 0; k < L; k++)
			fw[(size_t)k] = host_base(rd->h_fwd, (uint64_t)rd->h_woff[r] * 32 + (uint64_t)k);
		for (int k = 0; k < L; k++)
			rv[(size_t)k] = 3 - fw[(size_t)(L - 1 - k)];
	}
	void unmapped(std::string &unm, size_t r) const
	{
		unm += ">" + rd->name_of(r) + "\n";
		for (int b : fw)
			unm += kLetters[b];
		unm += "\n";
	}
};

static int soap_search(const SoapDb &sd, const char *reads_path, int max_n, int mode, SoapSearch &s)
{
	PGX_TRY(reads_from_fasta_ex(reads_path, 0, -1, true, &s.nn, &s.rd));
	const size_t n = (size_t)s.rd->n;
	s.best.assign(n, 3);
	s.n_best.assign(n, 0);
	s.first.assign(n, 0);
	PGX_TRY(s.d_best.alloc(n ? n : 1));
	PGX_TRY(s.d_nbest.alloc(n ? n : 1));
	PGX_TRY(s.d_first.alloc(n ? n : 1));
	PGX_TRY(s.d_hits.alloc(1));
	if (!n)
		return 0;
	std::vector<uint8_t> skip(n, 0);
	for (size_t i = 0; i < n; i++)
		skip[i] = s.nn[i] > (uint32_t)max_n;
	DevBuf<uint8_t> d_skip;
	DevBuf<unsigned long long> d_count;
	PGX_TRY(d_skip.alloc(n));
	PGX_TRY(d_skip.upload(skip.data(), n));
	PGX_TRY(d_count.alloc(1, 0, 0, true));
	uint64_t cap = std::max<uint64_t>(n * 8, 1024);
	unsigned long long total = 0;
	for (;;) {
		PGX_TRY(s.d_hits.alloc(cap));
		if (hipMemset(d_count.data(), 0, sizeof(unsigned long long)) != hipSuccess)
			return fail(PGX_E_NODEVICE, "hipMemset failed");
		const int grid = (int)std::min<uint64_t>((n + 3) / 4, 2048);
		hipLaunchKernelGGL(k_soap_search, dim3(grid), dim3(256), 0, 0, sd.v, s.rd->d_fwd.data(), s.rd->d_rc.data(), s.rd->d_len.data(),
				   s.rd->d_woff.data(), d_skip.data(), (uint32_t)n, s.d_hits.data(), (unsigned long long)cap, d_count.data(),
				   s.d_best.data(), s.d_nbest.data(), s.d_first.data(), mode);
		if (hipGetLastError() != hipSuccess)
			return fail(PGX_E_NODEVICE, "k_soap_search launch failed");
		PGX_TRY(d_count.download(&total, 1));
		if (total <= cap)
			break;
		cap = total; // the list was too short for this file: once more with the size it asked for
	}
	s.hits.resize((size_t)total);
	PGX_TRY(s.d_hits.download(s.hits.data(), s.hits.size()));
	PGX_TRY(s.d_best.download(s.best.data(), n));
	PGX_TRY(s.d_nbest.download(s.n_best.data(), n));
	PGX_TRY(s.d_first.download(s.first.data(), n));
	// a read's rows in (subject, position, strand) order (the ELF's order is its suffix array's: a documented deviation)
	for (size_t r = 0; r < n; r++)
		if (s.n_best[r] > 1)
			std::sort(s.hits.begin() + (ptrdiff_t)s.first[r], s.hits.begin() + (ptrdiff_t)(s.first[r] + s.n_best[r]),
				  [](const SoapHit &a, const SoapHit &b) {
					  if (a.subject != b.subject) return a.subject < b.subject;
					  if (a.pos != b.pos) return a.pos < b.pos;
					  return (a.strand_nmis >> 8) < (b.strand_nmis >> 8);
				  });
	return 0;
}

static int soap_run_single(const pgx_soap_opts *o, const SoapDb &sd)
{
	SoapSearch s;
	PGX_TRY(soap_search(sd, o->reads_path, o->max_n, o->match_mode, s));
	if (o->match_mode != 4 && s.rd->max_len > 256)
		// for reads above -l (256) the ELF applies -M 0 / 1 / 2 to the first 256 bases only: not restated
		return fail(PGX_E_LIMIT, "soap: -M %d is implemented for reads of at most 256 bases (this file holds one of %d)", o->match_mode,
			    (int)s.rd->max_len);
	std::string out, unm;
	const size_t n = (size_t)s.rd->n;
	for (size_t r = 0; r < n; r++) {
		s.unpack(r);
		const uint32_t nb = s.n_best[r];
		const SoapHit *h = s.hits.data() + s.first[r];
		bool printed = false;
		if (nb > 0 && !(o->repeat_mode == 0 && nb > 1)) {
			const uint32_t lim = o->repeat_mode == 2 ? nb : 1;
			for (uint32_t x = 0; x < lim; x++)
				soap_row(out, o->report_id ? std::to_string(r) : s.rd->name_of(r), // -t: the read's 0-based ordinal in the file
					 (h[x].strand_nmis >> 8) ? s.rv : s.fw, h[x], nb, sd.db, o->repeat_mode);
			printed = true;
		}
		if (!printed && nb <= 1)
			s.unmapped(unm, r);
	}
	PGX_TRY(write_text_file(o->out_path, out));
	if (o->unmapped_path && *o->unmapped_path)
		PGX_TRY(write_text_file(o->unmapped_path, unm));
	return 0;
}

// Paired-end run, `soap -a A -b B -D ref.index -o paired -2 unpaired [-u unmapped] -m MIN -x MAX` (soap.man:29-50).  The
// reference's pipeline calls soap single-ended (README.md:134); the rules were observed on the ELF and are pinned by
// tests/golden/soap/pe_*:
//   pair i = read i of A with read i of B; pairs are looked for among the placements with at most k mismatches of either
//   mate, k = 0, then 1, then 2 -- the first k that gives a valid pair decides and all valid pairs at that k are the result;
//   valid: one subject, opposite strands, (position of the '-' mate - position of the '+' mate) + LENGTH OF THE A MATE in
//   [MIN, MAX]; -r 2: all valid pairs (A rows, then B rows in the same order; column 4 = their number), -r 1: one, -r 0:
//   only a unique one (else both mates to the unmapped file); no valid pair: every placement of either mate to the -2 file
//   as single-end rows (column 4 = the mate's placements; -r 1: one; -r 0: a unique one, else unmapped).
// Outside what was observed, and refused: mates under 27 bases (the ELF crashes on some), mates above 256 bases, -M other
// than 4, -t.
static int soap_run_paired(const pgx_soap_opts *o, const SoapDb &sd)
{
	if (o->match_mode != 4)
		return fail(PGX_E_ARG, "soap: paired-end runs are implemented for -M 4");
	if (o->report_id)
		return fail(PGX_E_ARG, "soap: -t is not implemented for paired-end runs");
	if (!o->unpaired_path || !*o->unpaired_path)
		return fail(PGX_E_ARG, "soap: -2 (the file of unpaired placements) is required with -b");
	const int min_ins = o->min_insert == 0 && o->max_insert == 0 ? 400 : o->min_insert;
	const int max_ins = o->min_insert == 0 && o->max_insert == 0 ? 600 : o->max_insert;
	SoapSearch a, b;
	PGX_TRY(soap_search(sd, o->reads_path, o->max_n, 5, a));
	PGX_TRY(soap_search(sd, o->reads_b_path, o->max_n, 5, b));
	const size_t n = (size_t)std::min(a.rd->n, b.rd->n);
	for (size_t i = 0; i < n; i++) {
		const uint32_t la = a.rd->h_len[i], lb = b.rd->h_len[i];
		if (la < 27 || lb < 27 || la > 256 || lb > 256)
			return fail(PGX_E_LIMIT, "soap: paired-end runs are implemented for mates of 27 to 256 bases (pair %zu: %u and %u)", i + 1, la,
				    lb);
	}
	std::vector<uint32_t> level(n, 3);
	std::vector<unsigned long long> pairs(n, 0);
	if (n) {
		// the device lists in the order the host sorted them
		PGX_TRY(a.d_hits.upload(a.hits.data(), a.hits.size()));
		PGX_TRY(b.d_hits.upload(b.hits.data(), b.hits.size()));
		DevBuf<uint32_t> d_level;
		DevBuf<unsigned long long> d_pairs;
		PGX_TRY(d_level.alloc(n));
		PGX_TRY(d_pairs.alloc(n));
		const int grid = (int)std::min<uint64_t>((n + 3) / 4, 2048);
		hipLaunchKernelGGL(k_soap_pair, dim3(grid), dim3(256), 0, 0, a.d_hits.data(), a.d_first.data(), a.d_nbest.data(), b.d_hits.data(),
				   b.d_first.data(), b.d_nbest.data(), a.rd->d_len.data(), (uint32_t)n, min_ins, max_ins, d_level.data(),
				   d_pairs.data());
		if (hipGetLastError() != hipSuccess)
			return fail(PGX_E_NODEVICE, "k_soap_pair launch failed");
		PGX_TRY(d_level.download(level.data(), n));
		PGX_TRY(d_pairs.download(pairs.data(), n));
	}
	std::string out, un2, unm;
	for (size_t i = 0; i < n; i++) {
		a.unpack(i);
		b.unpack(i);
		const SoapHit *ha = a.hits.data() + a.first[i], *hb = b.hits.data() + b.first[i];
		const uint32_t na = a.n_best[i], nb = b.n_best[i];
		const int la = (int)a.rd->h_len[i];
		const unsigned long long np = pairs[i];
		const bool ambiguous = o->repeat_mode == 0 && np > 1;
		for (int mate = 0; mate < 2 && np > 0 && !ambiguous; mate++) {
			unsigned long long seen = 0; // the A rows first, then the B rows of the same pairs in the same order
			for (uint32_t x = 0; x < na; x++)
				for (uint32_t y = 0; y < nb; y++) {
					if (!soap_pair_geometry(ha[x], hb[y], la, min_ins, max_ins) || soap_pair_level(ha[x], hb[y]) > level[i])
						continue;
					if (o->repeat_mode != 2 && seen++ > 0)
						continue;
					if (mate == 0)
						soap_row(out, a.rd->name_of(i), (ha[x].strand_nmis >> 8) ? a.rv : a.fw, ha[x], np, sd.db, o->repeat_mode, 'a', true);
					else
						soap_row(out, b.rd->name_of(i), (hb[y].strand_nmis >> 8) ? b.rv : b.fw, hb[y], np, sd.db, o->repeat_mode, 'b', true);
				}
		}
		if (np > 0 && !ambiguous)
			continue;
		for (int mate = 0; mate < 2; mate++) { // the mates on their own
			SoapSearch &m = mate ? b : a;
			const SoapHit *h = mate ? hb : ha;
			const uint32_t nh = mate ? nb : na;
			if (np == 0 && nh > 0 && !(o->repeat_mode == 0 && nh > 1)) {
				const uint32_t lim = o->repeat_mode == 2 ? nh : 1;
				for (uint32_t x = 0; x < lim; x++)
					soap_row(un2, m.rd->name_of(i), (h[x].strand_nmis >> 8) ? m.rv : m.fw, h[x], nh, sd.db, o->repeat_mode, mate ? 'b' : 'a',
						 true);
			} else {
				m.unmapped(unm, i);
			}
		}
	}
	PGX_TRY(write_text_file(o->out_path, out));
	PGX_TRY(write_text_file(o->unpaired_path, un2));
	if (o->unmapped_path && *o->unmapped_path)
		PGX_TRY(write_text_file(o->unmapped_path, unm));
	return 0;
}

} // namespace pgx

extern "C" {

int pgx_soap_run(const pgx_soap_opts *o)
{
	if (!o || !o->reads_path || !o->db_prefix || !o->out_path)
		return fail(PGX_E_ARG, "soap: -a, -D and -o are required");
	if (o->match_mode != 4 && (o->match_mode < 0 || o->match_mode > 2))
		return fail(PGX_E_ARG, "soap: -M must be 0, 1, 2 or 4");
	if (o->repeat_mode < 0 || o->repeat_mode > 2)
		return fail(PGX_E_ARG, "soap: -r must be 0, 1 or 2");
	PGX_TRY(require_device());
	SoapDb sd;
	PGX_TRY(soap_db_open(o->db_prefix, sd));
	return o->reads_b_path && *o->reads_b_path ? soap_run_paired(o, sd) : soap_run_single(o, sd);
}
}
