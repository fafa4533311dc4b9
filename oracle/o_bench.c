/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * CPU baseline ("port") for bench.py: the oracle's classify -> taxcollector -> consensus chain on a
 * bounded sample of the synthetic workload, timed per stage.  The search is the megablast-style
 * restatement of o_blast.c (query lookup table, database scanned at stride 17) with OpenMP over
 * database chunks; the -outfmt 6 formatting and the taxcollector restatement run on contiguous blocks
 * of the table with one thread each (the stages are per line), the consensus restatement (a cursor walk
 * over two files) is single-threaded like the Perl.
 */
#define _GNU_SOURCE
#include "o_classify.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
	double gen_s, search_s, format_s, taxcollect_s, consensus_s;
	int64_t reads, hits, consensus_records;
	int32_t threads;
} o_bench_result;

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int o_bench_chain_files(const o_synth_cfg *cfg, int64_t first, int64_t n_reads, int threads, const char *taxdir,
			o_bench_result *res, const char *hits_out, const char *cons_out);

int o_bench_chain(const o_synth_cfg *cfg, int64_t first, int64_t n_reads, int threads, const char *taxdir,
		  o_bench_result *res)
{
	return o_bench_chain_files(cfg, first, n_reads, threads, taxdir, res, NULL, NULL);
}

/* same chain; optionally leaves the -outfmt 6 table and the consensus text in files (full-size parity tests) */
int o_bench_chain_files(const o_synth_cfg *cfg, int64_t first, int64_t n_reads, int threads, const char *taxdir,
			o_bench_result *res, const char *hits_out, const char *cons_out)
{
	memset(res, 0, sizeof *res);
	res->threads = threads;
	res->reads = n_reads;
	double t0 = now_s();
	o_seqset db, q;
	memset(&db, 0, sizeof db);
	memset(&q, 0, sizeof q);
	db.nseq = cfg->n_seq;
	db.total = cfg->n_seq * (int64_t)cfg->seq_len;
	db.off = (int64_t *)malloc((size_t)(db.nseq + 1) * sizeof(int64_t));
	db.header = (char **)calloc((size_t)db.nseq, sizeof(char *));
	db.base = (uint8_t *)malloc((size_t)db.total + 1);
	if (!db.off || !db.header || !db.base)
		return -1;
#pragma omp parallel for schedule(static) num_threads(threads)
	for (int64_t i = 0; i < db.nseq; i++) {
		db.off[i] = i * cfg->seq_len;
		o_synth_db_seq(cfg, i, db.base + i * cfg->seq_len);
		char h[64];
		snprintf(h, sizeof h, "gi|%ld|syn|S%ld|", (long)(1000 + i), (long)i);
		db.header[i] = strdup(h);
	}
	db.off[db.nseq] = db.total;
	q.nseq = n_reads;
	q.total = n_reads * (int64_t)cfg->read_len;
	q.off = (int64_t *)malloc((size_t)(n_reads + 1) * sizeof(int64_t));
	q.header = (char **)calloc((size_t)n_reads, sizeof(char *));
	q.base = (uint8_t *)malloc((size_t)q.total + 1);
	for (int64_t r = 0; r < n_reads; r++) {
		q.off[r] = r * cfg->read_len;
		o_synth_read(cfg, first + r, q.base + r * cfg->read_len, NULL, NULL, NULL);
		char h[32];
		snprintf(h, sizeof h, "r%ld", (long)(first + r));
		q.header[r] = strdup(h);
	}
	q.off[n_reads] = q.total;
	res->gen_s = now_s() - t0;

	t0 = now_s();
	o_hitvec hv;
	o_blast_search(&q, &db, &hv, threads);
	res->search_s = now_s() - t0;
	res->hits = (int64_t)hv.n;

	t0 = now_s();
	o_blast_stats st;
	o_blast_stats_init(&st, db.total, db.nseq, o_blast_gapped);
	obuf hits_txt;
	obuf_init(&hits_txt);
	int nt = threads > 0 ? threads : 1;
	if ((size_t)nt > hv.n / 1000 + 1)
		nt = (int)(hv.n / 1000 + 1);
	{
		obuf *part = (obuf *)calloc((size_t)nt, sizeof(obuf));
#pragma omp parallel for schedule(static, 1) num_threads(nt)
		for (int t = 0; t < nt; t++) {
			obuf_init(&part[t]);
			const size_t i0 = hv.n * (size_t)t / (size_t)nt, i1 = hv.n * (size_t)(t + 1) / (size_t)nt;
			for (size_t i = i0; i < i1; i++)
				o_blast_format_hit(&hv.h[i], &q, &db, &st, &part[t]);
		}
		for (int t = 0; t < nt; t++) {
			obuf_put(&hits_txt, part[t].p ? part[t].p : "", part[t].n);
			obuf_free(&part[t]);
		}
		free(part);
	}
	res->format_s = now_s() - t0;

	t0 = now_s();
	o_taxdb tax;
	o_tax_open(&tax, taxdir);
	obuf cls, rep;
	obuf_init(&cls);
	obuf_init(&rep);
	int rc = 0;
	{
		/* blocks of whole lines */
		const char *txt = hits_txt.p ? hits_txt.p : "";
		size_t *cut = (size_t *)calloc((size_t)nt + 1, sizeof(size_t));
		for (int t = 1; t < nt; t++) {
			size_t c = hits_txt.n * (size_t)t / (size_t)nt;
			while (c < hits_txt.n && c > 0 && txt[c - 1] != '\n')
				c++;
			cut[t] = c < cut[t - 1] ? cut[t - 1] : c;
		}
		cut[nt] = hits_txt.n;
		obuf *pc = (obuf *)calloc((size_t)nt, sizeof(obuf)), *pr = (obuf *)calloc((size_t)nt, sizeof(obuf));
		int *prc = (int *)calloc((size_t)nt, sizeof(int));
#pragma omp parallel for schedule(static, 1) num_threads(nt)
		for (int t = 0; t < nt; t++) {
			obuf_init(&pc[t]);
			obuf_init(&pr[t]);
			prc[t] = o_taxcollect_buf(&tax, txt + cut[t], cut[t + 1] - cut[t], &pc[t], &pr[t]);
		}
		for (int t = 0; t < nt; t++) {
			if (prc[t] < 0)
				rc = -1;
			obuf_put(&cls, pc[t].p ? pc[t].p : "", pc[t].n);
			obuf_put(&rep, pr[t].p ? pr[t].p : "", pr[t].n);
			obuf_free(&pc[t]);
			obuf_free(&pr[t]);
		}
		free(pc);
		free(pr);
		free(prc);
		free(cut);
	}
	res->taxcollect_s = now_s() - t0;

	t0 = now_s();
	char tmpl[] = "/tmp/pgx_rdp_XXXXXX";
	int fd = mkstemp(tmpl);
	if (fd >= 0)
		close(fd);
	o_synth_write_rdp(cfg, tmpl, first, n_reads);
	size_t rl = 0;
	char *rdp = o_read_file(tmpl, &rl);
	remove(tmpl);
	if (rdp) {
		/* The Perl never terminates on an RDP read without BLAST lines (about one synthetic read in
		 * 135 000 has no hit): such reads are left out of the RDP stream, which is what a user of the reference
		 * has to do by hand; the fused GPU path prints nothing for them either. */
		uint8_t *has = (uint8_t *)calloc((size_t)n_reads + 1, 1);
		for (size_t i = 0; i < hv.n; i++)
			has[hv.h[i].query] = 1;
		size_t w = 0, line = 0;
		for (size_t p0 = 0; p0 < rl;) {
			const char *nl = memchr(rdp + p0, '\n', rl - p0);
			const size_t e = nl ? (size_t)(nl - rdp) + 1 : rl;
			if (line >= (size_t)n_reads || has[line]) {
				memmove(rdp + w, rdp + p0, e - p0);
				w += e - p0;
			}
			p0 = e;
			line++;
		}
		rl = w;
		rdp[rl] = 0;
		free(has);
	}
	obuf cons, log;
	obuf_init(&cons);
	obuf_init(&log);
	if (rc == 0 && rdp)
		rc = o_consensus_buf(cls.p ? cls.p : "", cls.n, rdp, rl, &cons, &log);
	res->consensus_s = now_s() - t0;
	for (size_t i = 0; i + 1 < cons.n; i++)
		if (cons.p[i] == '\n' && cons.p[i + 1] == '#')
			res->consensus_records++;

	if (hits_out)
		obuf_write_file(&hits_txt, hits_out);
	if (cons_out)
		obuf_write_file(&cons, cons_out);
	free(rdp);
	obuf_free(&cons);
	obuf_free(&log);
	obuf_free(&cls);
	obuf_free(&rep);
	obuf_free(&hits_txt);
	o_tax_close(&tax);
	free(hv.h);
	o_seqset_free(&q);
	o_seqset_free(&db);
	return rc;
}
