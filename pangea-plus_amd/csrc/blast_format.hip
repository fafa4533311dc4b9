// Host side of the BLAST verb: Karlin-Altschul statistics and the -outfmt 6 formatter
// (12 tab-separated columns qseqid sseqid pident length mismatch gapopen qstart qend sstart send
// evalue bitscore — the layout the reference's downstream tools parse: Megaclust/megaclust2.pl:84-96,
// NCBI-taxcollector-0.01.pl:75-77), plus the `blastn` entry point (reference README.md:96).
#include <rocprim/device/device_scan.hpp>

#include <cmath>
#include <functional>

#include <functional>
#include <future>
#include "engine.hpp"

namespace pgx {

struct BlastStats {
	// blastn, reward 1 / penalty -2: lambda 1.28, K 0.46, H 0.85 with or without (linear) gaps; the length adjustment
	// uses alpha 1.5, beta -2 for the gapped search (spec v2) and alpha = lambda / H, beta 0 for `-ungapped` (the
	// published blastn parameter table)
	double lambda = 1.28, K = 0.46, H = 0.85, alpha = 1.5, beta = -2.0;
	int64_t db_len = 0, db_nseq = 0;
	void set(int64_t len, int64_t nseq, bool gapped)
	{
		db_len = len;
		db_nseq = nseq;
		alpha = gapped ? 1.5 : lambda / H;
		beta = gapped ? -2.0 : 0.0;
	}
};

// BLAST's length adjustment: largest ell with ell <= alpha / lambda * (ln K + ln((m-ell)(n-N ell))) + beta, found by
// the published bracketing iteration (20 rounds at most)
static int64_t length_adjust(const BlastStats &st, int64_t qlen)
{
	const double logK = std::log(st.K), adl = st.alpha / st.lambda, beta = st.beta;
	const double m = (double)qlen, n = (double)st.db_len, N = (double)st.db_nseq;
	const double c = n * m - std::max(m, n) / st.K;
	if (c < 0)
		return 0;
	const double mb = m * N + n;
	double ell_min = 0, ell_max = 2 * c / (mb + std::sqrt(mb * mb - 4 * N * c)), ell_next = 0;
	bool converged = false;
	for (int i = 1; i <= 20; i++) {
		const double ell = ell_next;
		const double ell_bar = adl * (logK + std::log((m - ell) * (n - N * ell))) + beta;
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
		if (up <= ell_max && adl * (logK + std::log((m - up) * (n - N * up))) + beta >= up)
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
void format_hit_columns(const pgx_hit &h, int64_t qlen, int64_t db_len, int64_t db_nseq, bool gapped, Text &out)
{
	BlastStats st;
	st.set(db_len, db_nseq, gapped);
	static thread_local int64_t cached_qlen = -1, cached_len = -1, cached_nseq = -1;
	static thread_local int cached_gapped = -1;
	static thread_local double cached_sp = 0;
	if (cached_qlen != qlen || cached_len != db_len || cached_nseq != db_nseq || cached_gapped != (int)gapped) {
		cached_sp = search_space(st, qlen);
		cached_qlen = qlen;
		cached_len = db_len;
		cached_nseq = db_nseq;
		cached_gapped = (int)gapped;
	}
	const int length = hit_length(h), diffs = hit_diffs(h);
	const double evalue = cached_sp * std::exp(-st.lambda * (double)h.score + std::log(st.K));
	const double bits = (st.lambda * (double)h.score - std::log(st.K)) / std::log(2.0);
	char ev[32], bs[32];
	format_evalue(evalue, ev);
	format_bitscore(bits, bs);
	const double pident = 100.0 * (double)(length - diffs) / (double)length;
	out.printf("%.2f\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%s\t%s", pident, length, (int)h.mismatch, (int)h.gapopen, h.qstart, h.qend,
		   h.sstart, h.send, ev, bs);
}

// the e-value and bit-score columns of a hit of raw score `score` for a query of `qlen` bases
void format_score_columns(int score, int64_t qlen, int64_t db_len, int64_t db_nseq, bool gapped, std::string &evalue, std::string &bits)
{
	BlastStats st;
	st.set(db_len, db_nseq, gapped);
	const double sp = search_space(st, qlen);
	const double e = sp * std::exp(-st.lambda * (double)score + std::log(st.K));
	const double b = (st.lambda * (double)score - std::log(st.K)) / std::log(2.0);
	char ev[32], bs[32];
	format_evalue(e, ev);
	format_bitscore(b, bs);
	evalue = ev;
	bits = bs;
}

} // namespace pgx

// columns 11-12 of an `-outfmt 6` row as the row formatter prints them (pure arithmetic on the host: the table the
// device formatter indexes is filled by this function); exported so that the known BLAST+ rows of the reference's
// validation spreadsheet can be checked against the product itself (tests/test_gpu_blast_rows.py)
extern "C" int pgx_blast_score_columns(int32_t score, int64_t qlen, int64_t db_len, int64_t db_nseq, char evalue[32], char bits[32])
{
	return pgx_blast_score_columns_v(score, qlen, db_len, db_nseq, 1, evalue, bits);
}

// the same with the statistics named: gapped != 0 = spec S4 (a table of a gapped search), 0 = S4u (`-ungapped`)
extern "C" int pgx_blast_score_columns_v(int32_t score, int64_t qlen, int64_t db_len, int64_t db_nseq, int gapped, char evalue[32], char bits[32])
{
	if (!evalue || !bits || qlen <= 0 || db_len <= 0 || db_nseq <= 0)
		return pgx::fail(PGX_E_ARG, "pgx_blast_score_columns: bad argument");
	std::string e, b;
	pgx::format_score_columns(score, qlen, db_len, db_nseq, gapped != 0, e, b);
	snprintf(evalue, 32, "%s", e.c_str());
	snprintf(bits, 32, "%s", b.c_str());
	return 0;
}

namespace pgx {

// host rendering, one snprintf per column: kept for batches whose (read length x score) table would be huge
static int format_hits_text_host(const pgx_hits *h, const pgx_db *db, const pgx_reads *reads, Text &out)
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
			format_hit_columns(x, qlen, db->n_bases, db->n_seq, h->gapped, out);
			out.s += '\n';
		}
	}
	return 0;
}

// ------------------------------------------------------------------------------------------ -outfmt 6 on the device
// A search of 10 M reads yields ~280 M rows, 20 GB of text: the table is rendered by kernels.  Everything in a
// row is an integer, a name, or a function of (raw score, query length): the e-value and bit-score columns
// come from a host-made table of their printed texts per (length, score), pident is hundredths / 100 (exactly
// what "%.2f" prints, engine.hpp), the rest is decimal integers.  Pass 1 sizes the rows, a scan places them,
// pass 2 writes them; chunks of rows go to the host as they are ready.
struct FmtView {
	const pgx_hit *hits;
	const uint32_t *read_off, *read_cnt, *read_len;
	const unsigned char *name_blob; // read names back to back (null: synthetic names r<first + i>)
	const uint32_t *name_off;
	unsigned long long first;
	const unsigned char *id_blob; // subject ids back to back
	const uint32_t *id_off;
	const uint32_t *len_slot;  // read length -> slot of the score table (or ~0)
	const uint32_t *slot_base; // slot -> first entry of its scores
	const uint32_t *score_off; // entry -> offset of "evalue\tbitscore" in score_blob; entry + 1 ends it
	const unsigned char *score_blob;
};

__device__ __forceinline__ uint32_t dec_len(unsigned long long v)
{
	uint32_t n = 1;
	while (v >= 10ull) {
		v /= 10ull;
		n++;
	}
	return n;
}
__device__ __forceinline__ unsigned char *put_dec(unsigned char *p, unsigned long long v)
{
	const uint32_t n = dec_len(v);
	for (uint32_t k = n; k-- > 0;) {
		p[k] = (unsigned char)('0' + (uint32_t)(v % 10ull));
		v /= 10ull;
	}
	return p + n;
}
__device__ __forceinline__ unsigned char *put_bytes(unsigned char *p, const unsigned char *s, uint32_t n)
{
	for (uint32_t k = 0; k < n; k++)
		p[k] = s[k];
	return p + n;
}

// WRITE = false: row lengths; WRITE = true: the rows at their offsets
template <bool WRITE>
__global__ void k_fmt_rows(FmtView v, uint64_t j0, uint64_t j1, unsigned long long *__restrict__ len_or_off,
			   unsigned char *__restrict__ out)
{
	const uint64_t j = j0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= j1)
		return;
	const pgx_hit h = v.hits[j];
	const uint32_t r = (uint32_t)h.read;
	const bool valid = j - v.read_off[r] < v.read_cnt[r]; // slots past the 500-subject cut print nothing
	if (!WRITE && !valid) {
		len_or_off[j - j0] = 0;
		return;
	}
	if (WRITE && !valid)
		return;
	const int alen = hit_length(h);
	const int hund = pident_hundredths(alen - hit_diffs(h), alen);
	const uint32_t qlen = v.read_len[r];
	const uint32_t e = v.slot_base[v.len_slot[qlen]] + (uint32_t)h.score;
	const uint32_t so = v.score_off[e], sn = v.score_off[e + 1] - so;
	const uint32_t io = v.id_off[h.subject], in = v.id_off[h.subject + 1] - io;
	uint32_t no = 0, nn = 0;
	if (v.name_blob) {
		no = v.name_off[r];
		nn = v.name_off[r + 1] - no;
	} else {
		nn = 1 + dec_len(v.first + r);
	}
	if (!WRITE) {
		len_or_off[j - j0] = (unsigned long long)nn + in + (dec_len((unsigned long long)(hund / 100)) + 3) + dec_len((unsigned long long)alen) +
				     dec_len((unsigned long long)h.mismatch) + dec_len((unsigned long long)h.gapopen) + dec_len((unsigned long long)h.qstart) +
				     dec_len((unsigned long long)h.qend) + dec_len((unsigned long long)h.sstart) +
				     dec_len((unsigned long long)h.send) + sn + 10 /* tabs (the 11th is inside the score text) */ + 1 /* newline */;
		return;
	}
	unsigned char *p = out + len_or_off[j - j0];
	if (v.name_blob) {
		p = put_bytes(p, v.name_blob + no, nn);
	} else {
		*p++ = 'r';
		p = put_dec(p, v.first + r);
	}
	*p++ = '\t';
	p = put_bytes(p, v.id_blob + io, in);
	*p++ = '\t';
	p = put_dec(p, (unsigned long long)(hund / 100));
	*p++ = '.';
	*p++ = (unsigned char)('0' + (hund % 100) / 10);
	*p++ = (unsigned char)('0' + hund % 10);
	*p++ = '\t';
	p = put_dec(p, (unsigned long long)alen);
	*p++ = '\t';
	p = put_dec(p, (unsigned long long)h.mismatch);
	*p++ = '\t';
	p = put_dec(p, (unsigned long long)h.gapopen);
	*p++ = '\t';
	p = put_dec(p, (unsigned long long)h.qstart);
	*p++ = '\t';
	p = put_dec(p, (unsigned long long)h.qend);
	*p++ = '\t';
	p = put_dec(p, (unsigned long long)h.sstart);
	*p++ = '\t';
	p = put_dec(p, (unsigned long long)h.send);
	*p++ = '\t';
	p = put_bytes(p, v.score_blob + so, sn);
	*p++ = '\n';
}

static bool build_score_table(const pgx_db *db, const pgx_reads *reads, int64_t n, bool trim_bits, bool gapped, std::vector<uint32_t> &len_slot,
			      std::vector<uint32_t> &slot_base, std::vector<uint32_t> &score_off, std::string &score_blob);

// one text per id, back to back, in HBM: made once per handle (engine.hpp: pgx_db::fmt_mu)
static int text_table_once(const std::vector<std::string> &texts, std::mutex &mu, bool &ready, DevBuf<unsigned char> &d_blob, DevBuf<uint32_t> &d_off)
{
	std::lock_guard<std::mutex> lock(mu);
	if (ready)
		return 0;
	std::string blob;
	std::vector<uint32_t> off(texts.size() + 1, 0);
	size_t total = 0;
	for (auto &t : texts)
		total += t.size();
	if (total > 0xFFFFFFFFull)
		return fail(PGX_E_LIMIT, "the texts of a formatter table exceed 4 GB");
	blob.reserve(total);
	for (size_t i = 0; i < texts.size(); i++) {
		blob += texts[i];
		off[i + 1] = (uint32_t)blob.size();
	}
	PGX_TRY(d_blob.alloc(blob.size() ? blob.size() : 1));
	PGX_TRY(d_blob.upload((const unsigned char *)blob.data(), blob.size()));
	PGX_TRY(d_off.alloc(off.size()));
	PGX_TRY(d_off.upload(off.data(), off.size()));
	ready = true;
	return 0;
}

// the lineage texts of a bound handle (pgx_db_bind_taxonomy calls this: binding is set-up, a consensus file is not)
int formatter_tables(const pgx_db *db) { return text_table_once(db->lin_text, db->fmt_mu, db->lin_blob_ready, db->d_lin_blob, db->d_lin_off); }

// renders the whole table; `sink` receives consecutive pieces of text
int format_hits_stream(const pgx_hits *h, const pgx_db *db, const pgx_reads *reads,
		       const std::function<int(const char *, size_t)> &sink)
{
	PGX_TRY(require_device());
	const uint64_t H = (uint64_t)h->n_hits;
	if (H == 0 || h->n_reads == 0)
		return 0;
	PGX_TRY(text_table_once(db->ids, db->fmt_mu, db->id_blob_ready, db->d_id_blob, db->d_id_off)); // subject ids
	// read names: the batch's own compact copy in HBM (seqdb.hip); a batch without one (older callers) gets it made here
	std::string name_blob;
	std::vector<uint32_t> name_off;
	const bool dev_names = !reads->synthetic && reads->d_name_at.base != nullptr;
	if (!reads->synthetic && !dev_names) {
		name_off.assign((size_t)reads->n + 1, 0);
		for (int64_t r = 0; r < reads->n; r++) {
			name_blob.append(reads->h_text->data() + reads->name_off[(size_t)r], reads->name_len[(size_t)r]);
			name_off[(size_t)r + 1] = (uint32_t)name_blob.size();
		}
	}
	// score columns per (read length, raw score): a dense table, fine for sequencing reads (a few hundred lengths); a
	// batch of long, all-different queries would need hundreds of millions of entries and is rendered by the host
	std::vector<uint32_t> len_slot, slot_base, score_off;
	std::string score_blob;
	if (!build_score_table(db, reads, reads->n, false, h->gapped, len_slot, slot_base, score_off, score_blob)) {
		Text t;
		PGX_TRY(format_hits_text_host(h, db, reads, t));
		return sink(t.s.data(), t.s.size());
	}
	DevBuf<unsigned char> d_name_blob, d_score_blob, d_out;
	DevBuf<uint32_t> d_name_off, d_len_slot, d_slot_base, d_score_off;
	auto up_bytes = [](DevBuf<unsigned char> &d, const std::string &s) -> int {
		PGX_TRY(d.alloc(s.size() ? s.size() : 1));
		return d.upload((const unsigned char *)s.data(), s.size());
	};
	auto up_u32 = [](DevBuf<uint32_t> &d, const std::vector<uint32_t> &v) -> int {
		PGX_TRY(d.alloc(v.size() ? v.size() : 1));
		return d.upload(v.data(), v.size());
	};
	if (!reads->synthetic && !dev_names) {
		PGX_TRY(up_bytes(d_name_blob, name_blob));
		PGX_TRY(up_u32(d_name_off, name_off));
	}
	PGX_TRY(up_u32(d_len_slot, len_slot));
	PGX_TRY(up_u32(d_slot_base, slot_base));
	PGX_TRY(up_u32(d_score_off, score_off));
	PGX_TRY(up_bytes(d_score_blob, score_blob));
	FmtView v;
	v.hits = h->d_hits.data();
	v.read_off = h->d_read_off.data();
	v.read_cnt = h->d_read_cnt.data();
	v.read_len = reads->d_len.data();
	v.name_blob = reads->synthetic ? nullptr : (dev_names ? reads->d_names.data() : d_name_blob.data());
	v.name_off = dev_names ? reads->d_name_at.data() : d_name_off.data();
	v.first = (unsigned long long)reads->first;
	v.id_blob = db->d_id_blob.data();
	v.id_off = db->d_id_off.data();
	v.len_slot = d_len_slot.data();
	v.slot_base = d_slot_base.data();
	v.score_off = d_score_off.data();
	v.score_blob = d_score_blob.data();
	const uint64_t chunk = 16ull << 20; // rows per piece (~1.2 GB of text)
	DevBuf<unsigned long long> d_len, d_off;
	PGX_TRY(d_len.alloc(std::min(H, chunk) + 1));
	PGX_TRY(d_off.alloc(std::min(H, chunk) + 1));
	size_t scan_bytes = 0;
	PGX_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, d_len.data(), d_off.data(), 0ull, (size_t)std::min(H, chunk) + 1,
					rocprim::plus<unsigned long long>()));
	DevBuf<uint8_t> scan_tmp;
	PGX_TRY(scan_tmp.alloc(scan_bytes ? scan_bytes : 1));
	std::vector<char> host;
	for (uint64_t j0 = 0; j0 < H; j0 += chunk) {
		const uint64_t j1 = std::min(H, j0 + chunk), m = j1 - j0;
		const unsigned grid = (unsigned)((m + 255) / 256);
		hipLaunchKernelGGL(k_fmt_rows<false>, dim3(grid), dim3(256), 0, 0, v, j0, j1, d_len.data(), (unsigned char *)nullptr);
		PGX_HIP(hipMemsetAsync(d_len.data() + m, 0, sizeof(unsigned long long), 0));
		PGX_HIP(rocprim::exclusive_scan(scan_tmp.data(), scan_bytes, d_len.data(), d_off.data(), 0ull, (size_t)m + 1,
						rocprim::plus<unsigned long long>()));
		unsigned long long bytes = 0;
		PGX_TRY(d_off.download(&bytes, 1, (size_t)m));
		if (bytes == 0)
			continue;
		if (d_out.n < bytes)
			PGX_TRY(d_out.alloc(bytes + bytes / 8));
		hipLaunchKernelGGL(k_fmt_rows<true>, dim3(grid), dim3(256), 0, 0, v, j0, j1, d_off.data(), d_out.data());
		PGX_HIP(hipGetLastError());
		host.resize(bytes);
		PGX_HIP(hipMemcpy(host.data(), d_out.data(), bytes, hipMemcpyDeviceToHost));
		PGX_TRY(sink(host.data(), bytes));
	}
	return 0;
}

// ------------------------------------------------------------------------------------------ Consensus text on the device
// One block of text per read (Consensus:223-234): the taxcollector line of the winning hit -- id, lineage, then the
// non-empty numeric columns (taxcollector:148-153; its split on blanks drops the blank in front of a 3-digit bit
// score) -- and "#Matches found: N".  Same two-pass scheme as the hit table.
struct ConsFmtView {
	FmtView f;                    // id_blob / id_off hold the lineage texts per lineage id here
	const pgx_consensus_rec *recs;
	const uint32_t *subj_lin;
};

template <bool WRITE>
__global__ void k_fmt_consensus(ConsFmtView c, uint64_t r0, uint64_t r1, unsigned long long *__restrict__ len_or_off,
				unsigned char *__restrict__ out)
{
	const uint64_t r = r0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= r1)
		return;
	const FmtView &v = c.f;
	const pgx_consensus_rec rec = c.recs[r];
	if (rec.hit == -2) { // no BLAST lines for this read: the Perl prints nothing for it
		if (!WRITE)
			len_or_off[r - r0] = 0;
		return;
	}
	const uint32_t mlen = 16 /* "#Matches found: " */ + dec_len((unsigned long long)rec.matches) + 1;
	uint32_t row = 1; // the newline of an empty line
	pgx_hit h;
	int alen = 0, hund = 0;
	uint32_t so = 0, sn = 0, io = 0, in = 0, no = 0, nn = 0;
	if (rec.hit >= 0) {
		h = v.hits[rec.hit];
		alen = hit_length(h);
		hund = pident_hundredths(alen - hit_diffs(h), alen);
		const uint32_t e = v.slot_base[v.len_slot[v.read_len[r]]] + (uint32_t)h.score;
		so = v.score_off[e];
		sn = v.score_off[e + 1] - so;
		const uint32_t lin = c.subj_lin[h.subject];
		io = v.id_off[lin];
		in = v.id_off[lin + 1] - io;
		if (v.name_blob) {
			no = v.name_off[r];
			nn = v.name_off[r + 1] - no;
		} else {
			nn = 1 + dec_len(v.first + r);
		}
		row = nn + in + (dec_len((unsigned long long)(hund / 100)) + 3) + dec_len((unsigned long long)alen) +
		      dec_len((unsigned long long)h.mismatch) + dec_len((unsigned long long)h.gapopen) + dec_len((unsigned long long)h.qstart) + dec_len((unsigned long long)h.qend) +
		      dec_len((unsigned long long)h.sstart) + dec_len((unsigned long long)h.send) + sn + 10 + 1;
	}
	if (!WRITE) {
		len_or_off[r - r0] = (unsigned long long)row + mlen;
		return;
	}
	unsigned char *p = out + len_or_off[r - r0];
	if (rec.hit >= 0) {
		if (v.name_blob) {
			p = put_bytes(p, v.name_blob + no, nn);
		} else {
			*p++ = 'r';
			p = put_dec(p, v.first + r);
		}
		*p++ = '\t';
		p = put_bytes(p, v.id_blob + io, in);
		*p++ = '\t';
		p = put_dec(p, (unsigned long long)(hund / 100));
		*p++ = '.';
		*p++ = (unsigned char)('0' + (hund % 100) / 10);
		*p++ = (unsigned char)('0' + hund % 10);
		*p++ = '\t';
		p = put_dec(p, (unsigned long long)alen);
		*p++ = '\t';
		p = put_dec(p, (unsigned long long)h.mismatch);
		*p++ = '\t';
		p = put_dec(p, (unsigned long long)h.gapopen);
		*p++ = '\t';
		p = put_dec(p, (unsigned long long)h.qstart);
		*p++ = '\t';
		p = put_dec(p, (unsigned long long)h.qend);
		*p++ = '\t';
		p = put_dec(p, (unsigned long long)h.sstart);
		*p++ = '\t';
		p = put_dec(p, (unsigned long long)h.send);
		*p++ = '\t';
		p = put_bytes(p, v.score_blob + so, sn);
	}
	*p++ = '\n';
	const char tag[] = "#Matches found: ";
	for (int k = 0; k < 16; k++)
		*p++ = (unsigned char)tag[k];
	p = put_dec(p, (unsigned long long)rec.matches);
	*p++ = '\n';
}

// score-column table shared by the two renderers; false when it would be too large (see format_hits_stream)
static bool build_score_table(const pgx_db *db, const pgx_reads *reads, int64_t n, bool trim_bits, bool gapped, std::vector<uint32_t> &len_slot,
			      std::vector<uint32_t> &slot_base, std::vector<uint32_t> &score_off, std::string &score_blob)
{
	const uint32_t max_len = (uint32_t)reads->max_len;
	std::vector<uint8_t> seen((size_t)max_len + 1, 0);
	uint64_t entries = 0;
	for (int64_t r = 0; r < n; r++)
		if (!seen[reads->h_len[(size_t)r]]) {
			seen[reads->h_len[(size_t)r]] = 1;
			entries += reads->h_len[(size_t)r] + 1;
		}
	if (entries > (16ull << 20))
		return false;
	len_slot.assign((size_t)max_len + 1, 0xFFFFFFFFu);
	std::string ev, bs;
	for (uint32_t L = 0; L <= max_len; L++) {
		if (!seen[L])
			continue;
		len_slot[L] = (uint32_t)slot_base.size();
		slot_base.push_back((uint32_t)score_off.size());
		for (uint32_t sc = 0; sc <= L; sc++) {
			score_off.push_back((uint32_t)score_blob.size());
			format_score_columns((int)sc, L, db->n_bases, db->n_seq, gapped, ev, bs);
			score_blob += ev;
			score_blob += '\t';
			size_t b0 = 0;
			while (trim_bits && b0 < bs.size() && bs[b0] == ' ')
				b0++;
			score_blob.append(bs, b0, std::string::npos);
		}
	}
	score_off.push_back((uint32_t)score_blob.size());
	return true;
}

// The Consensus text of a batch, rendered by kernels in pieces of `piece` reads; every piece is handed to `sink` as host
// memory (a pinned buffer, valid until the sink returns).  false: not rendered (score table too large), the caller uses its
// host loop.
bool consensus_format_pieces(const pgx_db *db, const pgx_reads *reads, const pgx_hits *hits, const pgx_consensus_rec *recs, int64_t n,
			     uint64_t piece, const std::function<int(const char *, size_t)> &sink, int *rc_out)
{
	*rc_out = 0;
	std::vector<uint32_t> len_slot, slot_base, score_off;
	std::string score_blob;
	if (!build_score_table(db, reads, n, true, hits->gapped, len_slot, slot_base, score_off, score_blob))
		return false;
	auto run = [&]() -> int {
		std::string name_blob;
		std::vector<uint32_t> name_off;
		PGX_TRY(text_table_once(db->lin_text, db->fmt_mu, db->lin_blob_ready, db->d_lin_blob, db->d_lin_off));
		const bool dev_names = !reads->synthetic && reads->d_name_at.base != nullptr;
		if (!reads->synthetic && !dev_names) {
			name_off.assign((size_t)n + 1, 0);
			for (int64_t r = 0; r < n; r++) {
				name_blob.append(reads->h_text->data() + reads->name_off[(size_t)r], reads->name_len[(size_t)r]);
				name_off[(size_t)r + 1] = (uint32_t)name_blob.size();
			}
		}
		DevBuf<unsigned char> d_name_blob, d_score_blob, d_out;
		DevBuf<uint32_t> d_name_off, d_len_slot, d_slot_base, d_score_off;
		DevBuf<pgx_consensus_rec> d_recs;
		auto up_bytes = [](DevBuf<unsigned char> &d, const std::string &s) -> int {
			PGX_TRY(d.alloc(s.size() ? s.size() : 1));
			return d.upload((const unsigned char *)s.data(), s.size());
		};
		auto up_u32 = [](DevBuf<uint32_t> &d, const std::vector<uint32_t> &v) -> int {
			PGX_TRY(d.alloc(v.size() ? v.size() : 1));
			return d.upload(v.data(), v.size());
		};
		if (!reads->synthetic && !dev_names) {
			PGX_TRY(up_bytes(d_name_blob, name_blob));
			PGX_TRY(up_u32(d_name_off, name_off));
		}
		PGX_TRY(up_u32(d_len_slot, len_slot));
		PGX_TRY(up_u32(d_slot_base, slot_base));
		PGX_TRY(up_u32(d_score_off, score_off));
		PGX_TRY(up_bytes(d_score_blob, score_blob));
		PGX_TRY(d_recs.alloc(n ? (size_t)n : 1));
		PGX_TRY(d_recs.upload(recs, (size_t)n));
		ConsFmtView c;
		c.f.hits = hits->d_hits.data();
		c.f.read_off = hits->d_read_off.data();
		c.f.read_cnt = hits->d_read_cnt.data();
		c.f.read_len = reads->d_len.data();
		c.f.name_blob = reads->synthetic ? nullptr : (dev_names ? reads->d_names.data() : d_name_blob.data());
		c.f.name_off = dev_names ? reads->d_name_at.data() : d_name_off.data();
		c.f.first = (unsigned long long)reads->first;
		c.f.id_blob = db->d_lin_blob.data();
		c.f.id_off = db->d_lin_off.data();
		c.f.len_slot = d_len_slot.data();
		c.f.slot_base = d_slot_base.data();
		c.f.score_off = d_score_off.data();
		c.f.score_blob = d_score_blob.data();
		c.recs = d_recs.data();
		c.subj_lin = db->d_subj_lin.data();
		const uint64_t N = (uint64_t)n, chunk = piece;
		// two pinned buffers: the sink (a file writer) works on one piece while the next is rendered and copied.  They are
		// the process's (pinning 90 MB costs as much as writing a piece): kept for the next table, one table at a time
		struct Pinned {
			char *p = nullptr;
			size_t cap = 0;
			int ensure(size_t n)
			{
				if (n <= cap)
					return 0;
				if (p)
					(void)hipHostFree(p);
				p = nullptr;
				cap = 0;
				PGX_HIP(hipHostMalloc((void **)&p, n + n / 8, hipHostMallocDefault));
				cap = n + n / 8;
				return 0;
			}
		};
		static std::mutex pin_mu;
		static Pinned pin[2];
		std::lock_guard<std::mutex> pin_lock(pin_mu);
		std::future<int> pending; // the sink's call on the previous piece
		uint64_t k_piece = 0;
		DevBuf<unsigned long long> d_len, d_off;
		PGX_TRY(d_len.alloc(std::min(N, chunk) + 1));
		PGX_TRY(d_off.alloc(std::min(N, chunk) + 1));
		size_t scan_bytes = 0;
		PGX_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, d_len.data(), d_off.data(), 0ull, (size_t)std::min(N, chunk) + 1,
						rocprim::plus<unsigned long long>()));
		DevBuf<uint8_t> scan_tmp;
		PGX_TRY(scan_tmp.alloc(scan_bytes ? scan_bytes : 1));
		for (uint64_t r0 = 0; r0 < N; r0 += chunk) {
			const uint64_t r1 = std::min(N, r0 + chunk), m = r1 - r0;
			const unsigned grid = (unsigned)((m + 255) / 256);
			hipLaunchKernelGGL(k_fmt_consensus<false>, dim3(grid), dim3(256), 0, 0, c, r0, r1, d_len.data(), (unsigned char *)nullptr);
			PGX_HIP(hipMemsetAsync(d_len.data() + m, 0, sizeof(unsigned long long), 0));
			PGX_HIP(rocprim::exclusive_scan(scan_tmp.data(), scan_bytes, d_len.data(), d_off.data(), 0ull, (size_t)m + 1,
							rocprim::plus<unsigned long long>()));
			unsigned long long bytes = 0;
			PGX_TRY(d_off.download(&bytes, 1, (size_t)m));
			if (bytes == 0)
				continue;
			if (d_out.n < bytes)
				PGX_TRY(d_out.alloc(bytes + bytes / 8));
			hipLaunchKernelGGL(k_fmt_consensus<true>, dim3(grid), dim3(256), 0, 0, c, r0, r1, d_off.data(), d_out.data());
			PGX_HIP(hipGetLastError());
			Pinned &pb = pin[k_piece & 1];
			PGX_TRY(pb.ensure((size_t)bytes));
			PGX_HIP(hipMemcpy(pb.p, d_out.data(), bytes, hipMemcpyDeviceToHost));
			if (pending.valid())
				PGX_TRY(pending.get()); // (the piece before: its buffer is the one the NEXT piece will be copied into)
			const char *hp = pb.p;
			const size_t hb = (size_t)bytes;
			pending = std::async(std::launch::async, [&sink, hp, hb]() { return sink(hp, hb); });
			k_piece++;
		}
		if (pending.valid())
			PGX_TRY(pending.get());
		return 0;
	};
	*rc_out = run();
	return true;
}

// the whole text in memory
bool consensus_format_device(const pgx_db *db, const pgx_reads *reads, const pgx_hits *hits, const pgx_consensus_rec *recs, int64_t n,
			     std::string &out, int *rc_out)
{
	return consensus_format_pieces(db, reads, hits, recs, n, 4ull << 20, [&](const char *p, size_t b) {
		out.append(p, b);
		return 0;
	}, rc_out);
}

int format_hits_text(const pgx_hits *h, const pgx_db *db, const pgx_reads *reads, Text &out)
{
	return format_hits_stream(h, db, reads, [&](const char *p, size_t n) {
		out.s.append(p, n);
		return 0;
	});
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
	db->ungapped = o->ungapped != 0;
	db->dust = o->no_dust == 0;
	// (a rank outside [0, world_size) used to write an empty file and succeed)
	if (o->world_size > 0 && (o->rank < 0 || o->rank >= o->world_size)) {
		pgx_db_close(db);
		return fail(PGX_E_ARG, "blastn: rank %d is outside [0, %d)", o->rank, o->world_size);
	}
	// read sharding (mpiblastn's static query partition, Scripts/submit_MPI-blast.job:24): this
	// process takes block `rank` of `world_size` contiguous blocks
	int ws = o->world_size > 0 ? o->world_size : 1, rk = o->rank;
	int rc = 0;
	const int64_t total = fasta_count_records(o->query_path);
	if (total < 0)
		rc = fail(PGX_E_IO, "cannot open query file %s", o->query_path);
	const int64_t lo = total > 0 ? total * rk / ws : 0, hi = total > 0 ? total * (rk + 1) / ws : 0;
	FILE *fo = nullptr, *fq = nullptr;
	if (rc == 0 && !(fo = fopen(o->out_path, "wb")))
		rc = fail(PGX_E_IO, "cannot open %s for writing", o->out_path);
	if (rc == 0 && !(fq = fopen(o->query_path, "rb")))
		rc = fail(PGX_E_IO, "cannot open query file %s", o->query_path);
	// The query file is streamed in pieces of about 1 GiB that end at a record boundary: each piece is one resident
	// batch (split + packed on the device, searched, its table rendered on the device and appended to the output),
	// so neither the file nor the hit table has to fit anywhere at once.
	const size_t piece_bytes = getenv("PGX_BLASTN_PIECE") ? (size_t)atoll(getenv("PGX_BLASTN_PIECE")) : ((size_t)1 << 30);
	std::string carry;
	int64_t seen = 0; // records in front of the current piece
	bool eof = false;
	while (rc == 0 && !eof && seen < hi) {
		std::string piece = std::move(carry);
		carry.clear();
		const size_t old = piece.size();
		piece.resize(old + piece_bytes);
		const size_t got = fread(&piece[old], 1, piece_bytes, fq);
		piece.resize(old + got);
		eof = got < piece_bytes;
		if (!eof) {
			// cut at the last record start; the rest joins the next piece
			size_t cut = piece.rfind("\n>");
			if (cut == std::string::npos || cut == 0) {
				carry = std::move(piece); // one record larger than a piece: keep reading
				continue;
			}
			carry.assign(piece, cut + 1, std::string::npos);
			piece.resize(cut + 1);
		}
		bool bol = true;
		const int64_t here = fasta_count_records_text(piece.data(), piece.size(), &bol);
		const int64_t a = std::max(lo, seen), b = std::min(hi, seen + here);
		if (a < b) {
			// a batch whose hit table does not fit (2^32 slots, or HBM) is halved until it does: reads that hit
			// tens of thousands of subjects each need small batches, ordinary reads take the whole piece
			const auto text = std::make_shared<const TextBlob>(std::move(piece));
			std::function<int(int64_t, int64_t)> run = [&](int64_t first, int64_t count) -> int {
				pgx_reads *rd = nullptr;
				pgx_hits *h = nullptr;
				int q = reads_from_fasta_text(text, first, count, false, nullptr, &rd);
				if (q == 0)
					q = pgx_blast_search(db, rd, &h);
				const bool searched = q == 0; // nothing of this batch has been written yet if the search failed
				if (q == 0)
					q = format_hits_stream(h, db, rd, [&](const char *p, size_t n) {
						return fwrite(p, 1, n, fo) == n ? 0 : fail(PGX_E_IO, "short write to %s", o->out_path);
					});
				pgx_hits_close(h);
				pgx_reads_close(rd);
				if (!searched && (q == PGX_E_LIMIT || q == PGX_E_NOMEM) && count > 1) {
					q = run(first, count / 2);
					if (q == 0)
						q = run(first + count / 2, count - count / 2);
				}
				return q;
			};
			rc = run(a - seen, b - a);
		}
		seen += here;
	}
	if (fq)
		fclose(fq);
	if (fo && fclose(fo) != 0 && rc == 0)
		rc = fail(PGX_E_IO, "cannot close %s", o->out_path);
	pgx_db_close(db);
	return rc;
}
}
