// Megaclust/megaclust2.pl and Megaclustable/megaclustable.pl: the two steps that turn
// the Consensus output into per-lineage counts and a rank-level abundance table.
//
//   pgx_megaclust_file    the reference command line on a text file: the host splits lines and columns (the
//                         script's own regex) and interns the OTU / query texts; thresholds, pair
//                         de-duplication and counting run on the device
//   pgx_megaclust_batch   the same table straight from the consensus records of a batch in HBM: thresholds
//                         become integer tests (hundredths of pident, minimum raw score per read length) that
//                         are exact images of the text comparisons the script would make on the printed line
//   pgx_megaclustable     rank-level pivot of several such tables: one device lane per taxon adds its
//                         lines in file order (the Perl's `+=`, same order, same doubles)
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>

#include "engine.hpp"

namespace pgx {

void format_score_columns(int score, int64_t qlen, int64_t db_len, int64_t db_nseq, bool gapped, std::string &evalue, std::string &bits);

// ------------------------------------------------------------------------------------------ Perl semantics (host)
static inline bool p_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

// what `<`, `>` and `+=` make of a string (perlnumber): optional blanks and sign, Inf/NaN, decimal digits
// with optional fraction and exponent; anything else counts as 0, trailing text is ignored
double perl_num(const char *s, size_t n)
{
	size_t i = 0;
	while (i < n && p_space(s[i]))
		i++;
	const size_t st = i;
	if (i < n && (s[i] == '+' || s[i] == '-'))
		i++;
	auto low = [&](size_t k) { return k < n ? (char)tolower((unsigned char)s[k]) : '\0'; };
	if (low(i) == 'i' && low(i + 1) == 'n' && low(i + 2) == 'f')
		return s[st] == '-' ? -INFINITY : INFINITY;
	if (low(i) == 'n' && low(i + 1) == 'a' && low(i + 2) == 'n')
		return NAN;
	size_t nd = 0;
	while (i < n && isdigit((unsigned char)s[i]))
		i++, nd++;
	if (i < n && s[i] == '.') {
		i++;
		while (i < n && isdigit((unsigned char)s[i]))
			i++, nd++;
	}
	if (nd == 0)
		return 0.0;
	if (i < n && (s[i] == 'e' || s[i] == 'E')) {
		size_t j = i + 1;
		if (j < n && (s[j] == '+' || s[j] == '-'))
			j++;
		if (j < n && isdigit((unsigned char)s[j])) {
			while (j < n && isdigit((unsigned char)s[j]))
				j++;
			i = j;
		}
	}
	std::string t(s + st, i - st);
	return strtod(t.c_str(), nullptr);
}
static double perl_num(const std::string &s) { return perl_num(s.data(), s.size()); }

// Perl truth of an option value: undef, "" and "0" are false
bool perl_true(const char *v) { return v && v[0] && !(v[0] == '0' && v[1] == 0); }

struct Field {
	const char *p;
	size_t n;
	bool defined;
};

// split /\t\t|\t\s|\s\t|\t/ (megaclust2.pl:96): leftmost match, alternatives in that order; trailing empty
// fields are dropped; the first `max` fields are returned (missing ones undefined)
static void mc_split(const char *s, size_t n, Field *f, int max)
{
	std::vector<Field> all;
	size_t start = 0, i = 0;
	while (i < n) {
		size_t sep = 0;
		if (s[i] == '\t')
			sep = (i + 1 < n && p_space(s[i + 1])) ? 2 : 1;
		else if (p_space(s[i]) && i + 1 < n && s[i + 1] == '\t')
			sep = 2;
		if (sep) {
			all.push_back({ s + start, i - start, true });
			i += sep;
			start = i;
		} else {
			i++;
		}
	}
	all.push_back({ s + start, n - start, true });
	while (!all.empty() && all.back().n == 0)
		all.pop_back();
	for (int k = 0; k < max; k++)
		f[k] = (size_t)k < all.size() ? all[(size_t)k] : Field{ "", 0, false };
}

static const char *const kUsage = // megaclust2.pl:166-185
	"Usage:\n"
	"\t\t   cluster-blast-output.pl -i infile -o outfile [options]\n"
	"\t\t   \n"
	"\t\t   Required options:\n"
	"\t\t   -i input BLAST tabular results file (megablast or blastall -m 8)\n"
	"\t\t   -o output file name\n"
	"\n"
	"\t\t   Optional parameters:\n"
	"\t\t   -s similarity lower threshold (percent, between 0-100) (default 95)\n"
	"\t\t   -e e-value upper threshold (default 1e-20)\n"
	"\t\t   -b bitscore lower threshold (default 200)\n"
	"\t\t   -d delimiter (default to comma)\n"
	"\t\t   \n"
	"\t\t   Optional switches:\n"
	"\t\t   -c count every query hit (if -c not given, then only count\n"
	"\t\t\t\t\t     any query-genome pair as one genome hit)\n"
	"\t\t   -h print usage summary\n"
	"\t\t   ";

struct McParams {
	double sim = 95, ev = 1e-20, bits = 200;
	std::string delim = ",";
	bool count_all = false;
};

// megaclust2.pl:37-73.  Returns 1 when the script would have printed a message and exited (status 0).
static int mc_params(const pgx_megaclust_opts *o, bool need_paths, McParams &p, Text &log)
{
	if (o->help) {
		log.s += kUsage;
		return 1;
	}
	if (need_paths && !(perl_true(o->in_path) && perl_true(o->out_path))) {
		log.s += "Must specify both an input and output filename\n";
		log.s += kUsage;
		return 1;
	}
	if (perl_true(o->s)) {
		const double v = perl_num(o->s, strlen(o->s));
		if (v < 0 || v > 100) {
			log.s += "similarity threshold must be between 0 and 100\n";
			log.s += kUsage;
			return 1;
		}
		p.sim = v;
	}
	if (perl_true(o->e))
		p.ev = perl_num(o->e, strlen(o->e));
	if (perl_true(o->b))
		p.bits = perl_num(o->b, strlen(o->b));
	if (perl_true(o->d))
		p.delim = o->d;
	p.count_all = perl_true(o->c);
	return 0;
}

// ------------------------------------------------------------------------------------------ device: counting
constexpr unsigned long long kNoKey = ~0ull;

// Text-table form: the three numeric columns of every line against the thresholds (IEEE comparisons, the
// same the Perl makes on its NVs; NaN compares false on both sides)
__global__ void k_mc_filter_lines(const double *__restrict__ pid, const double *__restrict__ ev, const double *__restrict__ bits,
				  uint64_t n, double sim, double ev_max, double bits_min, uint8_t *__restrict__ pass)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		pass[i] = !(pid[i] < sim || ev[i] > ev_max || bits[i] < bits_min);
}

// Batch form: one consensus record per read.  A winner's printed columns are functions of integers:
// pident text = hundredths / 100, e-value and bit-score texts depend on (raw score, read length) only, so the
// script's three text comparisons are `hundredths >= h_min` and `score >= s_min[read length]`.
__global__ void k_mc_filter_batch(const pgx_consensus_rec *__restrict__ recs, const pgx_hit *__restrict__ hits,
				  const uint32_t *__restrict__ read_len, const uint32_t *__restrict__ subj_lin, uint64_t n,
				  int h_min, const uint32_t *__restrict__ s_min, uint32_t max_len, int empty_pass,
				  uint32_t empty_lin, uint8_t *__restrict__ pass, uint8_t *__restrict__ line,
				  uint32_t *__restrict__ lin)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const int32_t w = recs[i].hit;
	uint8_t ps = 0, ln = 0;
	uint32_t l = empty_lin;
	if (w == -1) { // the Consensus prints an empty line for this read: examined, and all three columns undefined
		ln = 1;
		ps = (uint8_t)empty_pass;
	} else if (w >= 0) {
		ln = 1;
		const pgx_hit h = hits[w];
		const int len = hit_length(h);
		const int hund = pident_hundredths(len - hit_diffs(h), len);
		const uint32_t L = read_len[i];
		const uint32_t smin = L <= max_len ? s_min[L] : 0xFFFFFFFFu;
		ps = hund >= h_min && (uint32_t)h.score >= smin;
		l = subj_lin[h.subject];
	}
	pass[i] = ps;
	line[i] = ln;
	lin[i] = l;
}

// counts per OTU (count-every-hit mode), first counted line per OTU, keys for the pair de-duplication, and the
// number of examined / rejected lines
__global__ void k_mc_tally(const uint8_t *__restrict__ pass, const uint8_t *__restrict__ line, const uint32_t *__restrict__ otu,
			   const uint32_t *__restrict__ query, uint64_t n, int count_all, int want_keys,
			   unsigned long long *__restrict__ cnt, unsigned long long *__restrict__ first,
			   unsigned long long *__restrict__ keys, unsigned long long *__restrict__ totals)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const bool in = i < n;
	const bool is_line = in && (!line || line[i]);
	const bool ok = is_line && pass[i];
	const unsigned long long lm = __ballot(is_line), rm = __ballot(is_line && !ok);
	if ((threadIdx.x & 63) == 0) {
		if (lm)
			atomicAdd(&totals[0], (unsigned long long)__popcll(lm));
		if (rm)
			atomicAdd(&totals[1], (unsigned long long)__popcll(rm));
	}
	if (in && want_keys)
		keys[i] = ok ? ((unsigned long long)otu[i] << 32) | (unsigned long long)query[i] : kNoKey;
	if (!ok)
		return;
	// lanes that count the same OTU combine: one atomic per distinct OTU of the wavefront (a popular
	// lineage would otherwise serialise on one address)
	const uint32_t mine = otu[i];
	unsigned long long todo = __ballot(1);
	const int lane = threadIdx.x & 63;
	while (todo) {
		const int leader = __ffsll(todo) - 1;
		const uint32_t v = __shfl(mine, leader);
		const unsigned long long same = __ballot(mine == v) & todo;
		if (lane == leader) {
			if (count_all || !want_keys)
				atomicAdd(&cnt[v], (unsigned long long)__popcll(same));
			// first counted line of the OTU: the lowest lane of the group has the lowest line index
			atomicMin(&first[v], (unsigned long long)i);
		}
		todo &= ~same;
	}
}

// sorted keys: the head of every run of equal (OTU, query) keys counts once
__global__ void k_mc_heads(const unsigned long long *__restrict__ keys, uint64_t n, unsigned long long *__restrict__ cnt)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const unsigned long long k = keys[i];
	if (k == kNoKey || (i > 0 && keys[i - 1] == k))
		return;
	atomicAdd(&cnt[k >> 32], 1ull);
}

struct McResult {
	std::vector<unsigned long long> cnt, first;
	unsigned long long examined = 0, rejected = 0;
};

// pass/line/otu/query are device arrays of n entries (line may be null: every entry is a line)
static int mc_count(const uint8_t *d_pass, const uint8_t *d_line, const uint32_t *d_otu, const uint32_t *d_query, uint64_t n,
		    size_t n_otu, bool count_all, bool unique_queries, McResult &res)
{
	DevBuf<unsigned long long> d_cnt, d_first, d_tot, d_keys, d_sorted;
	PGX_TRY(d_cnt.alloc(n_otu ? n_otu : 1, 0, 0, true));
	PGX_TRY(d_first.alloc(n_otu ? n_otu : 1));
	PGX_HIP(hipMemset(d_first.data(), 0xFF, (n_otu ? n_otu : 1) * sizeof(unsigned long long)));
	PGX_TRY(d_tot.alloc(2, 0, 0, true));
	// without -c a (OTU, query) pair counts once: when every query text occurs once the pairs are distinct
	// by construction; otherwise sort the pair keys and count run heads
	const bool want_keys = !count_all && !unique_queries;
	if (want_keys) {
		PGX_TRY(d_keys.alloc(n ? n : 1));
		PGX_TRY(d_sorted.alloc(n ? n : 1));
	}
	if (n) {
		const unsigned grid = (unsigned)((n + 255) / 256);
		hipLaunchKernelGGL(k_mc_tally, dim3(grid), dim3(256), 0, 0, d_pass, d_line, d_otu, d_query, n, count_all ? 1 : 0,
				   want_keys ? 1 : 0, d_cnt.data(), d_first.data(), d_keys.data(), d_tot.data());
		PGX_HIP(hipGetLastError());
		if (want_keys) {
			size_t tmp_bytes = 0;
			PGX_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys.data(), d_sorted.data(), n));
			DevBuf<uint8_t> tmp;
			PGX_TRY(tmp.alloc(tmp_bytes ? tmp_bytes : 1));
			PGX_HIP(rocprim::radix_sort_keys(tmp.data(), tmp_bytes, d_keys.data(), d_sorted.data(), n));
			hipLaunchKernelGGL(k_mc_heads, dim3(grid), dim3(256), 0, 0, d_sorted.data(), n, d_cnt.data());
			PGX_HIP(hipGetLastError());
		}
	}
	res.cnt.resize(n_otu);
	res.first.resize(n_otu);
	PGX_TRY(d_cnt.download(res.cnt.data(), n_otu));
	PGX_TRY(d_first.download(res.first.data(), n_otu));
	unsigned long long tot[2];
	PGX_TRY(d_tot.download(tot, 2));
	res.examined = tot[0];
	res.rejected = tot[1];
	return 0;
}

// "OTU,times_hit" table in first-counted order (the reference prints `keys %h`: Perl hash order, undefined)
static void mc_render(const McResult &res, const std::vector<std::string> &otu_text, const McParams &p, std::string &csv, Text &log)
{
	std::vector<uint32_t> order;
	for (size_t i = 0; i < res.cnt.size(); i++)
		if (res.cnt[i])
			order.push_back((uint32_t)i);
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return res.first[a] < res.first[b]; });
	csv = "OTU" + p.delim + "times_hit\n"; // megaclust2.pl:152-153
	for (uint32_t i : order) {
		csv += otu_text[i];
		csv += p.delim;
		csv += std::to_string(res.cnt[i]);
		csv += '\n';
	}
	log.printf("Run complete:\n%llu hits examined\n%llu hits beyond thresholds and therefore not counted.\n", res.examined,
		   res.rejected); // :161-163
}

static uint32_t intern_text(std::unordered_map<std::string, uint32_t> &map, std::vector<std::string> &text, const char *p, size_t n,
			    bool *is_new = nullptr)
{
	std::string s(p, n);
	auto it = map.find(s);
	if (is_new)
		*is_new = it == map.end();
	if (it != map.end())
		return it->second;
	const uint32_t id = (uint32_t)text.size();
	text.push_back(s);
	map.emplace(std::move(s), id);
	return id;
}

// ------------------------------------------------------------------------------------------ megaclustable device part
// one lane per (file, taxon) cell: the cell's lines are added in file order, exactly as the Perl's `+=` does
__global__ void k_pivot_cells(const uint32_t *__restrict__ cell_off, const double *__restrict__ val, uint32_t n_cells,
			      double *__restrict__ sum)
{
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_cells)
		return;
	double s = 0.0; // the cell starts as "0" (or as its first line's text, which is val[first])
	for (uint32_t k = cell_off[c]; k < cell_off[c + 1]; k++)
		s += val[k];
	sum[c] = s;
}

static long p_index(const std::string &s, const std::string &sub, long pos)
{
	if (pos < 0)
		pos = 0;
	if ((size_t)pos > s.size())
		pos = (long)s.size();
	const size_t r = s.find(sub, (size_t)pos);
	return r == std::string::npos ? -1 : (long)r;
}
// substr(str, off, len), off >= 0; a negative len leaves that many characters off the end
static std::string p_substr(const std::string &s, long off, long len)
{
	if (off < 0 || (size_t)off > s.size())
		return std::string();
	long end = len >= 0 ? off + len : (long)s.size() + len;
	if (end > (long)s.size())
		end = (long)s.size();
	if (end <= off)
		return std::string();
	return s.substr((size_t)off, (size_t)(end - off));
}

static std::string perl_number_text(double v)
{
	char buf[64];
	snprintf(buf, sizeof buf, "%.15g", v);
	return buf;
}

} // namespace pgx

using namespace pgx;

extern "C" {

int pgx_megaclust_file(const pgx_megaclust_opts *o, char **log_text)
{
	Text log;
	auto done = [&](int rc) {
		if (log_text)
			*log_text = log.release_malloc(nullptr);
		return rc;
	};
	if (!o)
		return done(fail(PGX_E_ARG, "pgx_megaclust_file: null options"));
	McParams p;
	if (mc_params(o, true, p, log)) {
		log.s += "\n";
		return done(0);
	}
	int rc = require_device();
	if (rc < 0)
		return done(rc);
	bool ok;
	const std::string text = read_text_file(o->in_path, &ok);
	if (!ok)
		return done(fail(PGX_E_IO, "couldn't open infile %s", o->in_path)); // megaclust2.pl:75 dies
	FILE *probe = fopen(o->out_path, "w");
	if (!probe)
		return done(fail(PGX_E_IO, "couldn't open outfile %s", o->out_path)); // :76
	fclose(probe);
	// host: lines, columns, interned OTU / query texts, numeric columns as the Perl would numify them
	std::vector<double> pid, ev, bits;
	std::vector<uint32_t> otu, query;
	std::unordered_map<std::string, uint32_t> otu_id, query_id;
	std::vector<std::string> otu_text, query_text;
	bool unique_queries = true;
	for (size_t s = 0; s < text.size();) {
		const size_t nl = text.find('\n', s);
		const size_t e = nl == std::string::npos ? text.size() : nl; // chomp: the newline only
		const char *line = text.data() + s;
		const size_t n = e - s;
		s = nl == std::string::npos ? text.size() : e + 1;
		if (n > 0 && line[0] == '#') // :81
			continue;
		Field f[13];
		mc_split(line, n, f, 13);
		pid.push_back(f[2].defined ? perl_num(f[2].p, f[2].n) : 0.0);
		ev.push_back(f[10].defined ? perl_num(f[10].p, f[10].n) : 0.0);
		bits.push_back(f[11].defined ? perl_num(f[11].p, f[11].n) : 0.0);
		otu.push_back(intern_text(otu_id, otu_text, f[1].p, f[1].n));
		bool is_new;
		query.push_back(intern_text(query_id, query_text, f[0].p, f[0].n, &is_new));
		if (!is_new)
			unique_queries = false;
	}
	const uint64_t n = pid.size();
	DevBuf<double> d_pid, d_ev, d_bits;
	DevBuf<uint32_t> d_otu, d_query;
	DevBuf<uint8_t> d_pass;
	rc = d_pid.alloc(n ? n : 1);
	if (rc == 0) rc = d_ev.alloc(n ? n : 1);
	if (rc == 0) rc = d_bits.alloc(n ? n : 1);
	if (rc == 0) rc = d_otu.alloc(n ? n : 1);
	if (rc == 0) rc = d_query.alloc(n ? n : 1);
	if (rc == 0) rc = d_pass.alloc(n ? n : 1);
	if (rc == 0 && n) {
		rc = d_pid.upload(pid.data(), n);
		if (rc == 0) rc = d_ev.upload(ev.data(), n);
		if (rc == 0) rc = d_bits.upload(bits.data(), n);
		if (rc == 0) rc = d_otu.upload(otu.data(), n);
		if (rc == 0) rc = d_query.upload(query.data(), n);
	}
	if (rc < 0)
		return done(rc);
	if (n) {
		hipLaunchKernelGGL(k_mc_filter_lines, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_pid.data(), d_ev.data(),
				   d_bits.data(), n, p.sim, p.ev, p.bits, d_pass.data());
		if (hipGetLastError() != hipSuccess)
			return done(fail(PGX_E_NODEVICE, "k_mc_filter_lines launch failed"));
	}
	McResult res;
	rc = mc_count(d_pass.data(), nullptr, d_otu.data(), d_query.data(), n, otu_text.size(), p.count_all, unique_queries, res);
	if (rc < 0)
		return done(rc);
	std::string csv;
	mc_render(res, otu_text, p, csv, log);
	rc = write_text_file(o->out_path, csv);
	return done(rc);
}

int pgx_megaclust_batch(const pgx_db *db, const pgx_reads *reads, const pgx_hits *hits, const pgx_consensus_rec *recs, int64_t n,
			const pgx_megaclust_opts *o, char **csv_text, size_t *csv_len, char **log_text)
{
	Text log;
	auto done = [&](int rc) {
		if (log_text)
			*log_text = log.release_malloc(nullptr);
		return rc;
	};
	if (csv_text)
		*csv_text = nullptr;
	if (!db || !reads || !hits || !recs || !o || !csv_text || n < 0 || n > reads->n)
		return done(fail(PGX_E_ARG, "pgx_megaclust_batch: bad argument"));
	if (!db->bound)
		return done(fail(PGX_E_ARG, "pgx_megaclust_batch: database is not bound to a taxonomy"));
	McParams p;
	if (mc_params(o, false, p, log)) {
		log.s += "\n";
		return done(0);
	}
	int rc = require_device();
	if (rc < 0)
		return done(rc);
	// OTU ids come with the taxonomy binding: one per distinct lineage text, plus one for the empty text
	const std::vector<std::string> &lin_text = db->lin_text;
	const uint32_t empty_lin = db->empty_lin;
	// thresholds as integers.  pident: the text is "%.2f" of hundredths / 100
	int h_min = 10001;
	for (int h = 0; h <= 10000; h++) {
		char buf[16];
		snprintf(buf, sizeof buf, "%d.%02d", h / 100, h % 100);
		if (!(perl_num(buf, strlen(buf)) < p.sim)) {
			h_min = h;
			break;
		}
	}
	// e-value and bit score: both texts are functions of (raw score, read length); accepted scores form an
	// upper range (E falls and the bit score rises with the score, the formats are monotone)
	const uint32_t max_len = (uint32_t)reads->max_len;
	std::vector<uint32_t> s_min((size_t)max_len + 1, 0xFFFFFFFFu);
	{
		std::vector<uint8_t> seen((size_t)max_len + 1, 0);
		for (int64_t r = 0; r < n; r++)
			seen[reads->h_len[(size_t)r]] = 1;
		std::string evt, bst;
		auto ok_score = [&](int score, uint32_t L) {
			format_score_columns(score, L, db->n_bases, db->n_seq, hits->gapped, evt, bst);
			return !(perl_num(evt) > p.ev) && !(perl_num(bst) < p.bits);
		};
		for (uint32_t L = 0; L <= max_len; L++) {
			if (!seen[L] || L == 0)
				continue;
			if (!ok_score((int)L, L))
				continue; // not even a full-length perfect match passes
			int lo = 0, hi = (int)L; // hi passes; lowest passing score by bisection
			while (lo < hi) {
				const int mid = lo + (hi - lo) / 2;
				if (ok_score(mid, L))
					hi = mid;
				else
					lo = mid + 1;
			}
			s_min[L] = (uint32_t)hi;
		}
	}
	const int empty_pass = !(0.0 < p.sim || 0.0 > p.ev || 0.0 < p.bits);
	// query texts: a pair (OTU, query) counts once; only repeated read names need the key sort
	bool unique_queries = true;
	std::vector<uint32_t> qid;
	if (!reads->synthetic && !p.count_all && !ReadNameIndex(*reads).unique) {
		std::unordered_map<std::string, uint32_t> qmap;
		std::vector<std::string> qtext;
		qid.resize((size_t)n);
		for (int64_t r = 0; r < n; r++) {
			const std::string nm = reads->name_of(r);
			bool is_new;
			qid[(size_t)r] = intern_text(qmap, qtext, nm.data(), nm.size(), &is_new);
			if (!is_new)
				unique_queries = false;
		}
	}
	DevBuf<pgx_consensus_rec> d_recs;
	DevBuf<uint32_t> d_smin, d_lin, d_qid;
	DevBuf<uint8_t> d_pass, d_line;
	const size_t nn = n ? (size_t)n : 1;
	rc = d_recs.alloc(nn);
	if (rc == 0) rc = d_smin.alloc(s_min.size());
	if (rc == 0) rc = d_lin.alloc(nn);
	if (rc == 0) rc = d_pass.alloc(nn);
	if (rc == 0) rc = d_line.alloc(nn);
	if (rc == 0) rc = d_qid.alloc(nn);
	if (rc == 0 && n) rc = d_recs.upload(recs, (size_t)n);
	if (rc == 0) rc = d_smin.upload(s_min.data(), s_min.size());
	if (rc == 0 && !unique_queries) rc = d_qid.upload(qid.data(), (size_t)n);
	if (rc < 0)
		return done(rc);
	if (n) {
		hipLaunchKernelGGL(k_mc_filter_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_recs.data(), hits->d_hits.data(),
				   reads->d_len.data(), db->d_subj_lin.data(), (uint64_t)n, h_min, d_smin.data(), max_len, empty_pass, empty_lin,
				   d_pass.data(), d_line.data(), d_lin.data());
		if (hipGetLastError() != hipSuccess)
			return done(fail(PGX_E_NODEVICE, "k_mc_filter_batch launch failed"));
	}
	McResult res;
	rc = mc_count(d_pass.data(), d_line.data(), d_lin.data(), d_qid.data(), (uint64_t)n, lin_text.size(), p.count_all, unique_queries,
		      res);
	if (rc < 0)
		return done(rc);
	Text csv;
	mc_render(res, lin_text, p, csv.s, log);
	*csv_text = csv.release_malloc(csv_len);
	return done(*csv_text ? 0 : fail(PGX_E_NOMEM, "out of memory"));
}

int pgx_megaclustable(int argc, const char *const *argv, char **log_text)
{
	Text log;
	auto done = [&](int rc) {
		if (log_text)
			*log_text = log.release_malloc(nullptr);
		return rc;
	};
	if (argc < 0 || (argc > 0 && !argv))
		return done(fail(PGX_E_ARG, "pgx_megaclustable: bad argument vector"));
	if (argc - 1 < 5) { // megaclustable.pl:17-21
		log.s += "Please enter the correct parameters.\n";
		return done(0);
	}
	const char *output = nullptr;
	std::string level;
	std::vector<std::string> files;
	bool m_in = false;
	for (int a = 0; a < argc; a++) { // :25-52; `$mIN` (:38) is a typo in the script: -t never leaves the file-list mode
		const std::string arg = argv[a];
		if (arg == "-m") {
			m_in = true;
		} else if (arg == "-o") {
			m_in = false;
			a++;
			output = a < argc ? argv[a] : nullptr;
		} else if (arg == "-t") {
			a++;
			const std::string t = a < argc ? argv[a] : "";
			const double v = perl_num(t);
			if (v > 6 || v < 0) {
				log.s += "You must enter a number between 0 and 6 for taxonomy level where 0 = domain and 6 = species.\n";
				return done(0);
			}
			level = "[" + t + "]";
		} else if (m_in) {
			files.push_back(arg);
		}
	}
	int rc = require_device();
	if (rc < 0)
		return done(rc);
	// host pass: the script's index/substr arithmetic per line; every line becomes (file, taxon, number)
	std::vector<std::string> taxa;
	std::unordered_map<std::string, uint32_t> taxon_id;
	struct Item {
		uint32_t file, taxon;
		double val;
	};
	std::vector<Item> items;
	std::vector<size_t> size_at_file;                 // taxa known when each file was opened (cells start as "0")
	std::map<std::pair<uint32_t, uint32_t>, std::string> raw; // (file, taxon) -> text of the line that introduced the taxon
	for (size_t b = 0; b < files.size(); b++) {
		bool ok;
		const std::string text = read_text_file(files[b].c_str(), &ok);
		if (!ok) { // :63-67
			log.printf("Unable to open %s\nMake sure you entered the extension when entering the file name.\n", files[b].c_str());
			return done(0);
		}
		size_at_file.push_back(taxa.size());
		for (size_t s = 0; s < text.size();) {
			const size_t nl = text.find('\n', s);
			const size_t e = nl == std::string::npos ? text.size() : nl;
			const std::string line = text.substr(s, e - s);
			s = nl == std::string::npos ? text.size() : e + 1;
			long loc = p_index(line, level, 0); // :76
			if (loc < 0)
				continue;
			loc += 3;
			long end = p_index(line, ";", loc);
			if (end == -1)
				end = p_index(line, ",", loc);
			const std::string name = p_substr(line, loc, end - loc);
			const long num_start = p_index(line, ",", end) + 1;
			const std::string num = p_substr(line, num_start, (long)line.size() - num_start);
			auto it = taxon_id.find(name);
			if (it == taxon_id.end()) { // :98-103: a new taxon keeps the text of its first number
				const uint32_t id = (uint32_t)taxa.size();
				taxa.push_back(name);
				taxon_id.emplace(name, id);
				raw[{ (uint32_t)b, id }] = num;
				items.push_back({ (uint32_t)b, id, perl_num(num) });
			} else {
				items.push_back({ (uint32_t)b, it->second, perl_num(num) }); // :93 `+=`
			}
		}
	}
	// device: cells in (file, taxon) order, each summed in line order
	const size_t n_files = files.size(), n_taxa = taxa.size(), n_cells = n_files * n_taxa;
	std::vector<uint32_t> cell_cnt(n_cells + 1, 0), cell_off(n_cells + 1, 0);
	for (const Item &it : items)
		cell_cnt[(size_t)it.file * n_taxa + it.taxon]++;
	for (size_t c = 0; c < n_cells; c++)
		cell_off[c + 1] = cell_off[c] + cell_cnt[c];
	std::vector<double> vals(items.size());
	{
		std::vector<uint32_t> cur(cell_off.begin(), cell_off.end() - 1);
		for (const Item &it : items) // stable: line order inside a cell
			vals[cur[(size_t)it.file * n_taxa + it.taxon]++] = it.val;
	}
	std::vector<double> sums(n_cells, 0.0);
	if (n_cells) {
		DevBuf<uint32_t> d_off;
		DevBuf<double> d_val, d_sum;
		PGX_TRY(d_off.alloc(n_cells + 1));
		PGX_TRY(d_val.alloc(vals.size() ? vals.size() : 1));
		PGX_TRY(d_sum.alloc(n_cells));
		PGX_TRY(d_off.upload(cell_off.data(), n_cells + 1));
		PGX_TRY(d_val.upload(vals.data(), vals.size()));
		hipLaunchKernelGGL(k_pivot_cells, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, 0, d_off.data(), d_val.data(),
				   (uint32_t)n_cells, d_sum.data());
		PGX_HIP(hipGetLastError());
		PGX_TRY(d_sum.download(sums.data(), n_cells));
	}
	// :111-128
	if (!output)
		return done(fail(PGX_E_IO, "megaclustable: no output file"));
	std::string out;
	for (size_t a = 1; a <= n_files; a++)
		out += "\t" + std::to_string(a);
	for (size_t a = 0; a < n_taxa; a++) {
		out += "\n";
		out += taxa[a].empty() ? "0" : taxa[a]; // `eq ""` turns an empty name into 0 (:120-123)
		out += "\t";
		for (size_t b = 0; b < n_files; b++) {
			const size_t c = b * n_taxa + a;
			const uint32_t k = cell_cnt[c];
			auto r = raw.find({ (uint32_t)b, (uint32_t)a });
			std::string cell;
			if (r != raw.end() && k == 1)
				cell = r->second.empty() ? "0" : r->second; // pushed as text, never added to
			else if (k == 0)
				cell = "0"; // "0" (taxon known when the file was opened) or undef (added by a later file): both print 0
			else
				cell = perl_number_text(sums[c]);
			out += cell;
			out += "\t";
		}
	}
	rc = write_text_file(output, out);
	return done(rc);
}
}
