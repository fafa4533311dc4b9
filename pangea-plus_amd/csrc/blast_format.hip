// Host side of the BLAST verb: Karlin-Altschul statistics and the -outfmt 6 formatter
// (12 tab-separated columns qseqid sseqid pident length mismatch gapopen qstart qend sstart send
// evalue bitscore — the layout the reference's downstream tools parse: Megaclust/megaclust2.pl:84-96,
// NCBI-taxcollector-0.01.pl:75-77), plus the `blastn` entry point (reference README.md:96).
#include <cmath>

#include "engine.hpp"

namespace pgx {

struct BlastStats {
	double lambda = 1.28, K = 0.46, H = 0.85; // ungapped blastn, reward 1 / penalty -2
	int64_t db_len = 0, db_nseq = 0;
};

// BLAST's length adjustment: largest ell with ell <= (ln K + ln((m-ell)(n-N ell)))/H, found by the
// published bracketing iteration (20 rounds at most)
static int64_t length_adjust(const BlastStats &st, int64_t qlen)
{
	const double logK = std::log(st.K), adl = 1.0 / st.H;
	const double m = (double)qlen, n = (double)st.db_len, N = (double)st.db_nseq;
	const double c = n * m - std::max(m, n) / st.K;
	if (c < 0)
		return 0;
	const double mb = m * N + n;
	double ell_min = 0, ell_max = 2 * c / (mb + std::sqrt(mb * mb - 4 * N * c)), ell_next = 0;
	bool converged = false;
	for (int i = 1; i <= 20; i++) {
		const double ell = ell_next;
		const double ell_bar = adl * (logK + std::log((m - ell) * (n - N * ell)));
		if (ell_bar >= ell) {
			ell_min = ell;
			if (ell_bar - ell_min <= 1.0) {
				converged = true;
				break;
			}
			if (ell_min == ell_max)
				break;
		} else {
			ell_max = ell;
		}
		if (ell_min <= ell_bar && ell_bar <= ell_max)
			ell_next = ell_bar;
		else
			ell_next = i == 1 ? ell_max : (ell_min + ell_max) / 2;
	}
	int64_t adj = (int64_t)ell_min;
	if (converged) {
		const double up = std::ceil(ell_min);
		if (up <= ell_max && adl * (logK + std::log((m - up) * (n - N * up))) >= up)
			adj = (int64_t)up;
	}
	return adj;
}

struct QueryStats {
	double searchsp;
};

static double search_space(const BlastStats &st, int64_t qlen)
{
	const int64_t adj = length_adjust(st, qlen);
	int64_t eff_db = st.db_len - st.db_nseq * adj, eff_q = qlen - adj;
	if (eff_db <= 0)
		eff_db = 1;
	if (eff_q <= 0)
		eff_q = 1;
	return (double)eff_db * (double)eff_q;
}

static void format_evalue(double e, char out[32])
{
	if (e < 1.0e-180)
		snprintf(out, 32, "0.0");
	else if (e < 1.0e-99)
		snprintf(out, 32, "%2.0le", e);
	else if (e < 0.0009)
		snprintf(out, 32, "%3.0le", e);
	else if (e < 0.1)
		snprintf(out, 32, "%4.3lf", e);
	else if (e < 1.0)
		snprintf(out, 32, "%3.2lf", e);
	else if (e < 10.0)
		snprintf(out, 32, "%2.1lf", e);
	else
		snprintf(out, 32, "%5.0lf", e);
}

static void format_bitscore(double b, char out[32])
{
	if (b > 9999)
		snprintf(out, 32, "%4.3le", b);
	else if (b > 99.9)
		snprintf(out, 32, "%4ld", (long)b);
	else
		snprintf(out, 32, "%4.1lf", b);
}

// the 10 numeric columns of one hit ("pident\tlength\t...\tbitscore"), shared with the consensus text
void format_hit_columns(const pgx_hit &h, int64_t qlen, int64_t db_len, int64_t db_nseq, Text &out)
{
	BlastStats st;
	st.db_len = db_len;
	st.db_nseq = db_nseq;
	static thread_local int64_t cached_qlen = -1, cached_len = -1, cached_nseq = -1;
	static thread_local double cached_sp = 0;
	if (cached_qlen != qlen || cached_len != db_len || cached_nseq != db_nseq) {
		cached_sp = search_space(st, qlen);
		cached_qlen = qlen;
		cached_len = db_len;
		cached_nseq = db_nseq;
	}
	const int length = h.qend - h.qstart + 1;
	const double evalue = cached_sp * std::exp(-st.lambda * (double)h.score + std::log(st.K));
	const double bits = (st.lambda * (double)h.score - std::log(st.K)) / std::log(2.0);
	char ev[32], bs[32];
	format_evalue(evalue, ev);
	format_bitscore(bits, bs);
	const double pident = 100.0 * (double)(length - h.mismatch) / (double)length;
	out.printf("%.2f\t%d\t%d\t0\t%d\t%d\t%d\t%d\t%s\t%s", pident, length, h.mismatch, h.qstart, h.qend, h.sstart,
		   h.send, ev, bs);
}

// the e-value and bit-score columns of a hit of raw score `score` for a query of `qlen` bases
void format_score_columns(int score, int64_t qlen, int64_t db_len, int64_t db_nseq, std::string &evalue, std::string &bits)
{
	BlastStats st;
	st.db_len = db_len;
	st.db_nseq = db_nseq;
	const double sp = search_space(st, qlen);
	const double e = sp * std::exp(-st.lambda * (double)score + std::log(st.K));
	const double b = (st.lambda * (double)score - std::log(st.K)) / std::log(2.0);
	char ev[32], bs[32];
	format_evalue(e, ev);
	format_bitscore(b, bs);
	evalue = ev;
	bits = bs;
}

int format_hits_text(const pgx_hits *h, const pgx_db *db, const pgx_reads *reads, Text &out)
{
	std::vector<pgx_hit> hv((size_t)h->n_hits);
	std::vector<uint32_t> off((size_t)h->n_reads + 1), cnt((size_t)h->n_reads + 1);
	PGX_TRY(h->d_hits.download(hv.data(), hv.size()));
	PGX_TRY(h->d_read_off.download(off.data(), off.size()));
	PGX_TRY(h->d_read_cnt.download(cnt.data(), (size_t)h->n_reads));
	out.s.reserve(hv.size() * 80);
	for (int64_t r = 0; r < h->n_reads; r++) {
		const std::string qid = reads->name_of(r);
		const int64_t qlen = reads->h_len[(size_t)r];
		for (uint32_t k = 0; k < cnt[(size_t)r]; k++) {
			const pgx_hit &x = hv[off[(size_t)r] + k];
			out.s += qid;
			out.s += '\t';
			out.s += db->ids[(size_t)x.subject];
			out.s += '\t';
			format_hit_columns(x, qlen, db->n_bases, db->n_seq, out);
			out.s += '\n';
		}
	}
	return 0;
}

} // namespace pgx

using namespace pgx;

extern "C" {

int pgx_hits_format(const pgx_hits *h, const pgx_db *db, const pgx_reads *reads, char **text, size_t *len)
{
	if (!h || !db || !reads || !text)
		return fail(PGX_E_ARG, "pgx_hits_format: null argument");
	Text t;
	PGX_TRY(format_hits_text(h, db, reads, t));
	*text = t.release_malloc(len);
	return *text ? 0 : fail(PGX_E_NOMEM, "out of memory");
}

int pgx_blastn_run(const pgx_blastn_opts *o)
{
	if (!o || !o->query_path || !o->db_prefix || !o->out_path)
		return fail(PGX_E_ARG, "blastn: -query, -db and -out are required");
	if (o->outfmt != 6)
		return fail(PGX_E_ARG, "blastn: only -outfmt 6 is implemented");
	PGX_TRY(require_device());
	pgx_db *db = nullptr;
	PGX_TRY(pgx_db_open(o->db_prefix, &db));
	// read sharding (mpiblastn's static query partition, Scripts/submit_MPI-blast.job:24): this
	// process takes block `rank` of `world_size` contiguous blocks
	int ws = o->world_size > 0 ? o->world_size : 1, rk = o->rank;
	int rc = 0;
	const int64_t total = fasta_count_records(o->query_path);
	if (total < 0)
		rc = fail(PGX_E_IO, "cannot open query file %s", o->query_path);
	pgx_reads *rd = nullptr;
	if (rc == 0) {
		int64_t lo = total * rk / ws, hi = total * (rk + 1) / ws;
		rc = pgx_reads_from_fasta(o->query_path, lo, hi - lo, &rd);
	}
	pgx_hits *h = nullptr;
	if (rc == 0)
		rc = pgx_blast_search(db, rd, &h);
	Text t;
	if (rc == 0)
		rc = format_hits_text(h, db, rd, t);
	if (rc == 0)
		rc = write_text_file(o->out_path, t.s);
	pgx_hits_close(h);
	pgx_reads_close(rd);
	pgx_db_close(db);
	return rc;
}
}
