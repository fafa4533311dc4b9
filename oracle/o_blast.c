/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * BLAST mode, spec "pgx-blastn v1" — PARITY UNPINNED against NCBI BLAST+ 2.2.26 (not
 * vendored: Classify/Runblast/install_blast.sh:67; the reference only calls it:
 * README.md:96 `blastn -query F -db DB -outfmt 6 -out O`).  This restatement follows the
 * public megablast description the way BLAST itself is organised on a CPU:
 * a lookup table over the QUERY 12-mers, the database scanned at stride 17 (so that every
 * exact 28-mer contains one scanned word), exact-run confirmation, ungapped X-drop
 * extension, Karlin-Altschul statistics and the -outfmt 6 formatter.
 *
 * Spec (also DESIGN.md section "pgx-blastn v1"):
 *   S1 letters A C G T/U (any case) are 0..3; every other letter matches nothing.
 *   S2 both query strands; minus-strand hits are reported with sstart > send.
 *   S3 on every (query strand, subject, diagonal): maximal exact runs of >= 28 are seeds,
 *      taken left to right; a seed whose start lies inside the previous HSP of that
 *      diagonal is skipped; otherwise it is extended to both sides with +1/-2 scoring,
 *      stopping when the running score falls more than X = 10 below its best, and the
 *      HSP is the best-scoring extent (ties: shorter).
 *   S4 statistics: lambda 1.28, K 0.46, H 0.85 (ungapped 1/-2), BLAST length adjustment,
 *      E = searchsp * exp(-lambda*S + ln K); hits with E > 10 are dropped.
 *   S5 per query: subjects by (best score desc, subject ordinal asc), at most 500
 *      subjects; HSPs of a subject by (score desc, qstart, qend, sstart, send asc).
 *   S6 no gapped stage (gapopen column is 0), no DUST: both are stated deviations.
 */
#include "o_classify.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LUT_K 12
#define LUT_STRIDE (O_BLAST_W - LUT_K + 1) /* 17 */

static inline int is_match(const uint8_t *q, const uint8_t *s, int64_t d, int32_t k)
{
	return q[k] < 4 && q[k] == s[k + d];
}

/* S3d: an exact run [a, b) of the query strand is a seed only if one of its 28-base windows touches no masked base */
int o_blast_run_is_seed(const uint8_t *qmask, int32_t a, int32_t b)
{
	if (b - a < O_BLAST_W)
		return 0;
	if (!qmask)
		return 1;
	int32_t clean = 0;
	for (int32_t k = a; k < b; k++) {
		clean = qmask[k] ? 0 : clean + 1;
		if (clean >= O_BLAST_W)
			return 1;
	}
	return 0;
}

void o_blast_diag_hsps(const uint8_t *q, const uint8_t *qmask, int32_t qlen, const uint8_t *s, int32_t slen, int64_t d,
		       void (*emit)(void *, int32_t, int32_t, int32_t, int32_t, int32_t), void *ctx)
{
	int32_t lo = d < 0 ? (int32_t)(-d) : 0;
	int32_t hi = (int64_t)qlen < (int64_t)slen - d ? qlen : (int32_t)((int64_t)slen - d);
	if (hi - lo < O_BLAST_W)
		return;
	int32_t covered = lo;
	int32_t i = lo;
	while (i < hi) {
		if (!is_match(q, s, d, i)) {
			i++;
			continue;
		}
		int32_t j = i;
		while (j < hi && is_match(q, s, d, j))
			j++;
		if (o_blast_run_is_seed(qmask, i, j) && i >= covered) {
			int32_t best = 0, cur = 0, bl = i;
			for (int32_t k = i - 1; k >= lo; k--) {
				cur += is_match(q, s, d, k) ? O_BLAST_REWARD : O_BLAST_PENALTY;
				if (cur > best) {
					best = cur;
					bl = k;
				} else if (best - cur > O_BLAST_XDROP) {
					break;
				}
			}
			int32_t bestr = 0, br = j - 1;
			cur = 0;
			for (int32_t k = j; k < hi; k++) {
				cur += is_match(q, s, d, k) ? O_BLAST_REWARD : O_BLAST_PENALTY;
				if (cur > bestr) {
					bestr = cur;
					br = k;
				} else if (bestr - cur > O_BLAST_XDROP) {
					break;
				}
			}
			int32_t mism = 0;
			for (int32_t k = bl; k <= br; k++)
				mism += !is_match(q, s, d, k);
			/* S3b anchor: the exact run around the middle of the HSP -- the first base of the run of matches that holds
			 * the last matching position at or before the middle column (an extension from the middle costs the least:
			 * its work grows with the square of the differences on a side) */
			int32_t anchor = bl + (br - bl) / 2;
			while (!is_match(q, s, d, anchor))
				anchor--; /* (bl is a match) */
			while (anchor > bl && is_match(q, s, d, anchor - 1))
				anchor--;
			emit(ctx, bl, br, (j - i) + best + bestr, mism, anchor);
			covered = br + 1;
		}
		i = j;
	}
}

/* ---------------- statistics (S4) ---------------- */
int o_blast_gapped = 1, o_blast_prune = 1, o_blast_dust = 1;

void o_blast_stats_init(o_blast_stats *st, int64_t db_len, int64_t db_nseq, int gapped)
{
	/* Karlin-Altschul parameters of blastn for reward 1 / penalty -2: lambda 1.28, K 0.46, H 0.85 with or without
	 * (linear) gaps; the length adjustment uses alpha 1.5, beta -2 for the gapped search and alpha = lambda / H,
	 * beta 0 for the ungapped one (the published blastn parameter table) */
	st->lambda = 1.28;
	st->K = 0.46;
	st->H = 0.85;
	st->db_len = db_len;
	st->db_nseq = db_nseq;
	st->alpha = gapped ? 1.5 : st->lambda / st->H;
	st->beta = gapped ? -2.0 : 0.0;
}

int64_t o_blast_length_adjust(const o_blast_stats *st, int64_t qlen)
{
	/* the published BLAST_ComputeLengthAdjustment iteration; ungapped blastn: alpha/lambda = 1/H, beta = 0 */
	const double K = st->K, logK = log(st->K), adl = st->alpha / st->lambda, beta = st->beta;
	double m = (double)qlen, n = (double)st->db_len, N = (double)st->db_nseq;
	double ell, ss, ell_min = 0, ell_max, ell_next = 0;
	int converged = 0;
	double a = N, mb = m * N + n, c = n * m - (m > n ? m : n) / K;
	if (c < 0)
		return 0;
	ell_max = 2 * c / (mb + sqrt(mb * mb - 4 * a * c));
	for (int i = 1; i <= 20; i++) {
		double ell_bar;
		ell = ell_next;
		ss = (m - ell) * (n - N * ell);
		ell_bar = adl * (logK + log(ss)) + beta;
		if (ell_bar >= ell) {
			ell_min = ell;
			if (ell_bar - ell_min <= 1.0) {
				converged = 1;
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
			ell_next = (i == 1) ? ell_max : (ell_min + ell_max) / 2;
	}
	int64_t adj = (int64_t)ell_min;
	if (converged) {
		ell = ceil(ell_min);
		if (ell <= ell_max) {
			ss = (m - ell) * (n - N * ell);
			if (adl * (logK + log(ss)) + beta >= ell)
				adj = (int64_t)ell;
		}
	}
	return adj;
}

static double search_space(const o_blast_stats *st, int64_t qlen)
{
	int64_t adj = o_blast_length_adjust(st, qlen);
	int64_t eff_db = st->db_len - st->db_nseq * adj;
	if (eff_db <= 0)
		eff_db = 1;
	int64_t eff_q = qlen - adj;
	if (eff_q <= 0)
		eff_q = 1;
	return (double)eff_db * (double)eff_q;
}

double o_blast_evalue(const o_blast_stats *st, int64_t qlen, int32_t score)
{
	return search_space(st, qlen) * exp(-st->lambda * (double)score + log(st->K));
}

double o_blast_bitscore(const o_blast_stats *st, int32_t score)
{
	return (st->lambda * (double)score - log(st->K)) / log(2.0);
}

void o_blast_format_evalue(double e, char out[32])
{
	/* BLAST+ tabular e-value rules */
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

void o_blast_format_bitscore(double b, char out[32])
{
	/* BLAST+ 2.2.2x tabular bit-score rules: width-4 integer above 99.9 (" 937") */
	if (b > 9999)
		snprintf(out, 32, "%4.3le", b);
	else if (b > 99.9)
		snprintf(out, 32, "%4ld", (long)b);
	else
		snprintf(out, 32, "%4.1lf", b);
}

void o_blast_format_hit(const o_hit *h, const o_seqset *queries, const o_seqset *db, const o_blast_stats *st, obuf *out)
{
	char qid[512], sid[512], ev[32], bs[32];
	o_seq_id(queries->header[h->query], qid, sizeof qid);
	o_seq_id(db->header[h->subject], sid, sizeof sid);
	int64_t qlen = queries->off[h->query + 1] - queries->off[h->query];
	o_blast_format_evalue(o_blast_evalue(st, qlen, h->score), ev);
	o_blast_format_bitscore(o_blast_bitscore(st, h->score), bs);
	double pident = 100.0 * (double)(h->length - h->mismatch - h->gaps) / (double)h->length;
	obuf_printf(out, "%s\t%s\t%.2f\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%s\t%s\n", qid, sid, pident, h->length, h->mismatch,
		    h->gapopen, h->qstart, h->qend, h->sstart, h->send, ev, bs);
}

/* ---------------- search ---------------- */
typedef struct {
	int32_t *head;  /* 4^12 heads, -1 = empty */
	int32_t *next;  /* per entry */
	int32_t *qs;    /* per entry: query strand index = 2*query + strand */
	int32_t *qpos;  /* per entry */
	int64_t n;
} lut_t;

static void hv_push(o_hitvec *v, const o_hit *h)
{
	if (v->n == v->cap) {
		v->cap = v->cap ? v->cap * 2 : 1024;
		v->h = (o_hit *)realloc(v->h, v->cap * sizeof(o_hit));
	}
	v->h[v->n++] = *h;
}

typedef struct {
	o_hitvec *out;
	int32_t query, subject, strand, qlen, slen;
	int64_t d;
	const uint8_t *q, *s; /* the query strand and the subject as base arrays */
} emit_ctx;

/* S3b: every initial HSP is extended with gaps from the first base of its seed run, to the left and to the right
 * (o_gapped.c); without the gapped stage (`-ungapped`, spec v1) the initial HSP is the hit */
static void emit_hit(void *vctx, int32_t bl, int32_t br, int32_t score, int32_t mism, int32_t seed)
{
	emit_ctx *c = (emit_ctx *)vctx;
	o_hit h;
	h.query = c->query;
	h.subject = c->subject;
	h.score = score;
	h.mismatch = mism;
	h.gapopen = h.gaps = 0;
	int32_t sl = (int32_t)(bl + c->d), sr = (int32_t)(br + c->d); /* 0-based subject extent */
	if (o_blast_gapped) {
		const int32_t qa = seed, sa = (int32_t)(seed + c->d);
		o_gext L, R;
		o_greedy_extend(c->q + qa - 1, qa, c->s + sa - 1, sa, -1, o_blast_prune, &L);
		o_greedy_extend(c->q + qa, c->qlen - qa, c->s + sa, c->slen - sa, +1, o_blast_prune, &R);
		bl = qa - L.i;
		br = qa + R.i - 1;
		sl = sa - L.j;
		sr = sa + R.j - 1;
		h.score = (L.s2 + R.s2) >> 1; /* floor of the half-integral score */
		h.mismatch = L.mism + R.mism;
		h.gapopen = L.gapopen + R.gapopen;
		h.gaps = L.gap_s + L.gap_q + R.gap_s + R.gap_q;
	}
	h.length = ((br - bl + 1) + (sr - sl + 1) + h.gaps) / 2;
	if (!c->strand) {
		h.qstart = bl + 1;
		h.qend = br + 1;
		h.sstart = sl + 1;
		h.send = sr + 1;
	} else {
		h.qstart = c->qlen - br;
		h.qend = c->qlen - bl;
		h.sstart = sr + 1;
		h.send = sl + 1;
	}
	hv_push(c->out, &h);
}

static int cmp_final(const void *a, const void *b);

typedef struct {
	o_hit h;
	int32_t best; /* best score of (query, subject) */
} hit_key;

static int cmp_group(const void *a, const void *b)
{
	const hit_key *x = (const hit_key *)a, *y = (const hit_key *)b;
	if (x->h.query != y->h.query)
		return x->h.query < y->h.query ? -1 : 1;
	if (x->h.subject != y->h.subject)
		return x->h.subject < y->h.subject ? -1 : 1;
	return 0;
}

static int cmp_final(const void *a, const void *b)
{
	const hit_key *x = (const hit_key *)a, *y = (const hit_key *)b;
#define CMP(f, desc)                                                                                                   \
	if (x->f != y->f)                                                                                              \
		return ((x->f < y->f) ? -1 : 1) * ((desc) ? -1 : 1);
	CMP(h.query, 0)
	CMP(best, 1)
	CMP(h.subject, 0)
	CMP(h.score, 1)
	CMP(h.qstart, 0)
	CMP(h.qend, 0)
	CMP(h.sstart, 0)
	CMP(h.send, 0)
	CMP(h.mismatch, 0)
	CMP(h.gapopen, 0)
#undef CMP
	return 0;
}

int o_blast_search(const o_seqset *queries, const o_seqset *db, o_hitvec *out, int threads)
{
	memset(out, 0, sizeof *out);
	const int64_t nq = queries->nseq;
	/* both strands of every query as base arrays (S2) */
	uint8_t *rcb = (uint8_t *)malloc((size_t)queries->total + 1);
	for (int64_t qi = 0; qi < nq; qi++) {
		int64_t o = queries->off[qi], L = queries->off[qi + 1] - o;
		for (int64_t k = 0; k < L; k++) {
			uint8_t b = queries->base[o + L - 1 - k];
			rcb[o + k] = b < 4 ? (uint8_t)(3 - b) : O_AMB;
		}
	}
	/* S3d: low-complexity mask of every query, forward and (the same mask reversed) reverse-complement strand */
	uint8_t *maskf = NULL, *maskr = NULL;
	if (o_blast_dust) {
		maskf = (uint8_t *)calloc((size_t)queries->total + 1, 1);
		maskr = (uint8_t *)calloc((size_t)queries->total + 1, 1);
		for (int64_t qi = 0; qi < nq; qi++) {
			int64_t o = queries->off[qi], L = queries->off[qi + 1] - o;
			o_dust_mask(queries->base + o, (int32_t)L, maskf + o);
			for (int64_t k = 0; k < L; k++)
				maskr[o + k] = maskf[o + L - 1 - k];
		}
	}
	/* lookup table over all query 12-mers of both strands */
	lut_t lut;
	lut.head = (int32_t *)malloc(sizeof(int32_t) << (2 * LUT_K));
	memset(lut.head, 0xFF, sizeof(int32_t) << (2 * LUT_K));
	int64_t cap = 2 * queries->total + 16;
	lut.next = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
	lut.qs = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
	lut.qpos = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
	lut.n = 0;
	for (int64_t qi = 0; qi < nq; qi++) {
		int64_t o = queries->off[qi], L = queries->off[qi + 1] - o;
		for (int st = 0; st < 2; st++) {
			const uint8_t *b = st ? rcb + o : queries->base + o;
			uint32_t w = 0;
			int valid = 0;
			for (int64_t k = 0; k < L; k++) {
				if (b[k] < 4) {
					w = ((w << 2) | b[k]) & ((1u << (2 * LUT_K)) - 1);
					valid++;
				} else {
					valid = 0;
				}
				if (valid >= LUT_K) {
					int64_t e = lut.n++;
					lut.qs[e] = (int32_t)(2 * qi + st);
					lut.qpos[e] = (int32_t)(k - LUT_K + 1);
					lut.next[e] = lut.head[w];
					lut.head[w] = (int32_t)e;
				}
			}
		}
	}

	if (threads < 1)
		threads = 1;
	o_hitvec *tv = (o_hitvec *)calloc((size_t)threads, sizeof(o_hitvec));
	const int64_t nwords = db->total >= LUT_K ? (db->total - LUT_K) / LUT_STRIDE + 1 : 0;
#pragma omp parallel num_threads(threads)
	{
#ifdef _OPENMP
		int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
		int tid = 0, nt = 1;
#endif
		int64_t w0 = nwords * tid / nt, w1 = nwords * (tid + 1) / nt;
		int64_t subj = 0;
		/* subject containing the first scanned position of this chunk */
		if (w0 < w1) {
			int64_t p0 = w0 * LUT_STRIDE, lo = 0, hi = db->nseq;
			while (hi - lo > 1) {
				int64_t mid = (lo + hi) / 2;
				if (db->off[mid] <= p0)
					lo = mid;
				else
					hi = mid;
			}
			subj = lo;
		}
		for (int64_t wi = w0; wi < w1; wi++) {
			int64_t p = wi * LUT_STRIDE;
			while (subj + 1 < db->nseq && db->off[subj + 1] <= p)
				subj++;
			if (p + LUT_K > db->off[subj + 1])
				continue; /* word crosses a subject boundary */
			uint32_t w = 0;
			int ok = 1;
			for (int k = 0; k < LUT_K; k++) {
				uint8_t b = db->base[p + k];
				if (b >= 4) {
					ok = 0;
					break;
				}
				w = (w << 2) | b;
			}
			if (!ok)
				continue;
			const uint8_t *s = db->base + db->off[subj];
			int32_t slen = (int32_t)(db->off[subj + 1] - db->off[subj]);
			int32_t spos = (int32_t)(p - db->off[subj]);
			for (int32_t e = lut.head[w]; e >= 0; e = lut.next[e]) {
				int32_t qsi = lut.qs[e], qi = qsi >> 1, st = qsi & 1, qp = lut.qpos[e];
				int64_t o = queries->off[qi];
				int32_t qlen = (int32_t)(queries->off[qi + 1] - o);
				const uint8_t *q = st ? rcb + o : queries->base + o;
				const uint8_t *qm = maskf ? (st ? maskr + o : maskf + o) : NULL;
				int64_t d = (int64_t)spos - qp;
				int32_t lo = d < 0 ? (int32_t)(-d) : 0;
				int32_t hi = (int64_t)qlen < (int64_t)slen - d ? qlen : (int32_t)((int64_t)slen - d);
				int32_t a = qp, b = qp + LUT_K;
				while (a - 1 >= lo && is_match(q, s, d, a - 1))
					a--;
				while (b < hi && is_match(q, s, d, b))
					b++;
				if (!o_blast_run_is_seed(qm, a, b))
					continue;
				if (qp - LUT_STRIDE >= a)
					continue; /* an earlier scanned word of the same run reports it */
				/* only the first seed run of the diagonal generates its HSPs */
				int first = 1;
				for (int32_t k = a - 1; k >= lo && first; k--) {
					if (!is_match(q, s, d, k))
						continue;
					int32_t e2 = k;
					while (k - 1 >= lo && is_match(q, s, d, k - 1))
						k--;
					if (o_blast_run_is_seed(qm, k, e2 + 1))
						first = 0;
				}
				if (!first)
					continue;
				emit_ctx c = { &tv[tid], qi, (int32_t)subj, st, qlen, slen, d, q, s };
				o_blast_diag_hsps(q, qm, qlen, s, slen, d, emit_hit, &c);
			}
		}
	}

	/* merge, E-value filter, order (S4, S5) */
	o_blast_stats stt;
	o_blast_stats_init(&stt, db->total, db->nseq, o_blast_gapped);
	size_t total = 0;
	for (int t = 0; t < threads; t++)
		total += tv[t].n;
	hit_key *hk = (hit_key *)malloc((total + 1) * sizeof(hit_key));
	size_t n = 0;
	for (int t = 0; t < threads; t++) {
		for (size_t k = 0; k < tv[t].n; k++) {
			const o_hit *h = &tv[t].h[k];
			int64_t qlen = queries->off[h->query + 1] - queries->off[h->query];
			if (o_blast_evalue(&stt, qlen, h->score) > 10.0)
				continue;
			hk[n].h = *h;
			hk[n].best = 0;
			n++;
		}
		free(tv[t].h);
	}
	free(tv);
	qsort(hk, n, sizeof(hit_key), cmp_group);
	for (size_t i = 0; i < n;) {
		size_t j = i;
		int32_t best = hk[i].h.score;
		while (j < n && hk[j].h.query == hk[i].h.query && hk[j].h.subject == hk[i].h.subject) {
			if (hk[j].h.score > best)
				best = hk[j].h.score;
			j++;
		}
		for (size_t k = i; k < j; k++)
			hk[k].best = best;
		i = j;
	}
	qsort(hk, n, sizeof(hit_key), cmp_final);
	/* S3c: hits of one (query, subject) that describe the same alignment.  A hit is dropped when a hit BEFORE it in
	 * the S5 order (dropped itself or not), on the same strand, starts at the same point, ends at the same point, or
	 * holds it (query range and subject range) -- the seeds of one alignment on either side of a gap all grow into it */
	if (o_blast_gapped) {
		size_t w = 0;
		uint8_t *drop = (uint8_t *)calloc(n + 1, 1);
		for (size_t i = 0; i < n;) {
			size_t j = i;
			while (j < n && hk[j].h.query == hk[i].h.query && hk[j].h.subject == hk[i].h.subject)
				j++;
			for (size_t a = i + 1; a < j; a++) {
				const o_hit *A = &hk[a].h;
				const int am = A->sstart > A->send;
				const int32_t as0 = am ? A->send : A->sstart, as1 = am ? A->sstart : A->send;
				for (size_t b = i; b < a && !drop[a]; b++) {
					const o_hit *B = &hk[b].h;
					const int bm = B->sstart > B->send;
					if (am != bm)
						continue;
					const int32_t bs0 = bm ? B->send : B->sstart, bs1 = bm ? B->sstart : B->send;
					if ((A->qstart == B->qstart && A->sstart == B->sstart) || (A->qend == B->qend && A->send == B->send) ||
					    (A->qstart >= B->qstart && A->qend <= B->qend && as0 >= bs0 && as1 <= bs1))
						drop[a] = 1;
				}
			}
			i = j;
		}
		for (size_t i = 0; i < n; i++)
			if (!drop[i])
				hk[w++] = hk[i];
		n = w;
		free(drop);
	}
	int32_t curq = -1, cursubj = -1, nsubj = 0;
	for (size_t i = 0; i < n; i++) {
		if (hk[i].h.query != curq) {
			curq = hk[i].h.query;
			cursubj = -1;
			nsubj = 0;
		}
		if (hk[i].h.subject != cursubj) {
			cursubj = hk[i].h.subject;
			nsubj++;
		}
		if (nsubj > O_BLAST_MAX_TARGETS)
			continue;
		hv_push(out, &hk[i].h);
	}
	free(hk);
	free(lut.head);
	free(lut.next);
	free(lut.qs);
	free(lut.qpos);
	free(rcb);
	free(maskf);
	free(maskr);
	return 0;
}

int o_blastn_files(const char *query_fa, const char *db_fa, const char *out_path, int threads)
{
	o_seqset q, d;
	if (o_seqset_read_fasta(&q, query_fa) < 0)
		return -1;
	if (o_seqset_read_fasta(&d, db_fa) < 0) {
		o_seqset_free(&q);
		return -1;
	}
	o_hitvec hv;
	o_blast_search(&q, &d, &hv, threads);
	o_blast_stats st;
	o_blast_stats_init(&st, d.total, d.nseq, o_blast_gapped);
	obuf out;
	obuf_init(&out);
	for (size_t i = 0; i < hv.n; i++)
		o_blast_format_hit(&hv.h[i], &q, &d, &st, &out);
	int rc = obuf_write_file(&out, out_path);
	obuf_free(&out);
	free(hv.h);
	o_seqset_free(&q);
	o_seqset_free(&d);
	return rc;
}
