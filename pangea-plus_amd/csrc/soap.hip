// Classify, SOAP verb: `2bwt-builder ref.fa` + `soap -a reads -D ref.fa.index -o out -M 4 [-r] [-n] [-u]`
// (reference README.md:130-134, soap.man:29-83).  The reference ships soap only as a closed ELF;
// every rule here was observed from that binary and is pinned by tests/golden/soap/:
//   letters other than ACGT read as G (reference and reads); runs of >= 10 such letters in the
//   reference are cut out (no hit may overlap one); a hit must end before the last base of its
//   segment; reads shorter than 27 or with more than -n non-ACGT letters are not aligned; -M 4
//   keeps the full-length ungapped hits with the fewest (<= 2) mismatches on either strand.
//
// k_soap_search  one wavefront per read, two passes over the same candidates (no per-read storage):
//   pass 0 finds the minimum mismatch count and how many placements reach it, pass 1 emits them.
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
						     uint32_t *__restrict__ n_best, int mode)
{
	const int lane = threadIdx.x & 63;
	for (uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n_reads; r += gridDim.x * 4) {
		const int L = (int)len[r];
		const uint32_t w0 = woff[r];
		if (skip[r] || L < 27) {
			if (lane == 0) {
				best_nmis[r] = 3;
				n_best[r] = 0;
			}
			continue;
		}
		const bool exact_mode = L >= 48;
		const int so[3] = { 0, L / 3, 2 * (L / 3) };
		const int per_strand = exact_mode ? 3 : kVariants;
		const int P = 2 * per_strand;
		uint32_t best = 3;
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
					if (!active)
						continue;
					const uint32_t p = db.postings[o_lo + (key - o_excl)];
					const uint64_t *rw = (o_strand ? rc : fwd) + w0;
					const int seed_off = exact_mode ? so[o_sidx] : 0;
					if (p < (uint32_t)seed_off)
						continue;
					const int64_t gp = (int64_t)p - seed_off;
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
						continue;
					if (exact_mode) {
						// the seed itself must match (bucket collisions) and no earlier seed may:
						// the placement is reported through its first exact seed
						if (!seed_exact(rw, db.words, gp, seed_off))
							continue;
						bool earlier = false;
						for (int j = 0; j < o_sidx; j++)
							earlier |= seed_exact(rw, db.words, gp, so[j]);
						if (earlier)
							continue;
					} else if (kmer16(db.words, gp) != o_kmer) {
						continue;
					}
					int m0, m1;
					const int nm = soap_mismatches(rw, L, db.words, gp, m0, m1);
					if (nm > 2)
						continue;
					if (pass == 0) {
						c0 += nm == 0;
						c1 += nm == 1;
						c2 += nm == 2;
					} else if ((uint32_t)nm == best) {
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
						unsigned long long g = atomicAdd(hit_count, 1ull);
						if (g < cap)
							hits[g] = h;
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
				if (mode == 4)
					best = c0 ? 0u : (c1 ? 1u : (c2 ? 2u : 3u));
				else
					best = (mode == 0 ? c0 : (mode == 1 ? c1 : c2)) ? (uint32_t)mode : 3u;
				if (lane == 0) {
					best_nmis[r] = best;
					n_best[r] = best == 0 ? c0 : (best == 1 ? c1 : (best == 2 ? c2 : 0u));
				}
				if (best == 3)
					break;
			}
		}
	}
}

static const char kLetters[5] = "ACGT";

static inline int host_base(const std::vector<uint64_t> &w, uint64_t p) { return (int)((w[p >> 5] >> (2 * (p & 31))) & 3); }

// one output row of soap (format observed from the ELF, see file header)
static void soap_row(std::string &out, const std::string &name, const std::vector<int> &rd, const SoapHit &h, uint32_t nbest,
		     const pgx_db *db, int repeat)
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
	snprintf(buf, sizeof buf, "\t%u\ta\t%d\t%c\t", nbest, L, strand ? '-' : '+');
	out += buf;
	out += db->ids[h.subject];
	snprintf(buf, sizeof buf, "\t%u\t%d", h.pos + 1, nmis);
	out += buf;
	const uint64_t g0 = (uint64_t)db->h_seq_off[h.subject] + h.pos;
	int m[2] = { h.mis0, h.mis1 };
	if (nmis == 2 && m[1] >= L - 13)
		std::swap(m[0], m[1]); // entries descend when one lies in the last 13 bases
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

int pgx_soap_run(const pgx_soap_opts *o)
{
	if (!o || !o->reads_path || !o->db_prefix || !o->out_path)
		return fail(PGX_E_ARG, "soap: -a, -D and -o are required");
	if (o->match_mode != 4 && (o->match_mode < 0 || o->match_mode > 2))
		return fail(PGX_E_ARG, "soap: -M must be 0, 1, 2 or 4");
	if (o->repeat_mode < 0 || o->repeat_mode > 2)
		return fail(PGX_E_ARG, "soap: -r must be 0, 1 or 2");
	PGX_TRY(require_device());
	pgx_db *src = nullptr;
	PGX_TRY(db_read_host(o->db_prefix, &src));
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
	pgx_db *db = nullptr;
	int rc = db_fold_amb_to_g(src, &db);
	delete src;
	if (rc < 0)
		return rc;
	pgx_reads *rd = nullptr;
	std::vector<uint32_t> nn;
	rc = reads_from_fasta_ex(o->reads_path, 0, -1, true, &nn, &rd);
	if (rc == 0 && o->match_mode != 4 && rd->max_len > 256) {
		// for reads above -l (256) the ELF applies -M 0 / 1 / 2 to the first 256 bases only: not restated
		const int longest = (int)rd->max_len;
		pgx_reads_close(rd);
		pgx_db_close(db);
		return fail(PGX_E_LIMIT, "soap: -M %d is implemented for reads of at most 256 bases (this file holds one of %d)", o->match_mode,
			    longest);
	}
	std::string out, unm;
	if (rc == 0) {
		const size_t n = (size_t)rd->n;
		std::vector<uint8_t> skip(n ? n : 1, 0);
		for (size_t i = 0; i < n; i++)
			skip[i] = nn[i] > (uint32_t)o->max_n;
		DevBuf<uint32_t> d_seg_lo, d_seg_hi, d_best, d_nbest;
		DevBuf<uint8_t> d_skip;
		DevBuf<unsigned long long> d_count;
		DevBuf<SoapHit> d_hits;
		std::vector<SoapHit> hv;
		std::vector<uint32_t> best(n), nbest(n);
		if (rc == 0) rc = d_seg_lo.alloc(seg_lo.size() ? seg_lo.size() : 1);
		if (rc == 0) rc = d_seg_lo.upload(seg_lo.data(), seg_lo.size());
		if (rc == 0) rc = d_seg_hi.alloc(seg_hi.size() ? seg_hi.size() : 1);
		if (rc == 0) rc = d_seg_hi.upload(seg_hi.data(), seg_hi.size());
		if (rc == 0) rc = d_skip.alloc(skip.size());
		if (rc == 0) rc = d_skip.upload(skip.data(), skip.size());
		if (rc == 0) rc = d_best.alloc(n ? n : 1);
		if (rc == 0) rc = d_nbest.alloc(n ? n : 1);
		if (rc == 0) rc = d_count.alloc(1, 0, 0, true);
		SoapView v;
		v.words = db->d_words.data();
		v.seq_off = db->d_seq_off.data();
		v.blk_subj = db->d_blk_subj.data();
		v.bucket_off = db->d_bucket_off.data();
		v.postings = db->d_postings.data();
		v.seg_lo = d_seg_lo.data();
		v.seg_hi = d_seg_hi.data();
		v.n_seg = (uint32_t)seg_lo.size();
		v.bits = db->index_bits;
		uint64_t cap = std::max<uint64_t>(n * 8, 1024);
		unsigned long long total = 0;
		while (rc == 0 && n) {
			rc = d_hits.alloc(cap);
			if (rc == 0 && hipMemset(d_count.data(), 0, sizeof(unsigned long long)) != hipSuccess)
				rc = fail(PGX_E_NODEVICE, "hipMemset failed");
			if (rc < 0)
				break;
			const int grid = (int)std::min<uint64_t>((n + 3) / 4, 2048);
			hipLaunchKernelGGL(k_soap_search, dim3(grid), dim3(256), 0, 0, v, rd->d_fwd.data(), rd->d_rc.data(),
					   rd->d_len.data(), rd->d_woff.data(), d_skip.data(), (uint32_t)n, d_hits.data(),
					   (unsigned long long)cap, d_count.data(), d_best.data(), d_nbest.data(), o->match_mode);
			if (hipGetLastError() != hipSuccess) {
				rc = fail(PGX_E_NODEVICE, "k_soap_search launch failed");
				break;
			}
			rc = d_count.download(&total, 1);
			if (rc < 0 || total <= cap)
				break;
			cap = total;
		}
		if (rc == 0 && n) {
			hv.resize((size_t)total);
			rc = d_hits.download(hv.data(), hv.size());
			if (rc == 0) rc = d_best.download(best.data(), n);
			if (rc == 0) rc = d_nbest.download(nbest.data(), n);
		}
		if (rc == 0) {
			// equal-best rows of a read in (subject, position, strand) order
			std::sort(hv.begin(), hv.end(), [](const SoapHit &a, const SoapHit &b) {
				if (a.read != b.read) return a.read < b.read;
				if (a.subject != b.subject) return a.subject < b.subject;
				if (a.pos != b.pos) return a.pos < b.pos;
				return (a.strand_nmis >> 8) < (b.strand_nmis >> 8);
			});
			size_t hp = 0;
			std::vector<int> fw, rv;
			for (size_t r = 0; r < n; r++) {
				const int L = (int)rd->h_len[r];
				fw.resize((size_t)L);
				rv.resize((size_t)L);
				for (int k = 0; k < L; k++)
					fw[(size_t)k] = host_base(rd->h_fwd, (uint64_t)rd->h_woff[r] * 32 + (uint64_t)k);
				for (int k = 0; k < L; k++)
					rv[(size_t)k] = 3 - fw[(size_t)(L - 1 - k)];
				size_t he = hp;
				while (he < hv.size() && hv[he].read == r)
					he++;
				const uint32_t nb = nbest[r];
				bool printed = false;
				if (nb > 0 && !(o->repeat_mode == 0 && nb > 1)) {
					const size_t lim = o->repeat_mode == 2 ? he - hp : 1;
					for (size_t x = 0; x < lim && hp + x < he; x++)
						soap_row(out, o->report_id ? std::to_string(r) : rd->name_of(r), // -t: the read's 0-based ordinal in the file
							 (hv[hp + x].strand_nmis >> 8) ? rv : fw, hv[hp + x], nb, db, o->repeat_mode);
					printed = true;
				}
				if (!printed && nb <= 1) {
					unm += ">" + rd->name_of(r) + "\n";
					for (int k = 0; k < L; k++)
						unm += kLetters[fw[(size_t)k]];
					unm += "\n";
				}
				hp = he;
			}
		}
	}
	if (rc == 0)
		rc = write_text_file(o->out_path, out);
	if (rc == 0 && o->unmapped_path && *o->unmapped_path)
		rc = write_text_file(o->unmapped_path, unm);
	pgx_reads_close(rd);
	pgx_db_close(db);
	return rc;
}
}
