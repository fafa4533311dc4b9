/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * CPU baseline ("port") for bench.py: the oracle's classify -> taxcollector -> consensus chain on a
 * bounded sample of the synthetic workload, timed per stage.  The search is the megablast-style
 * restatement of o_blast.c (query lookup table, database scanned at stride 17) with OpenMP over
 * database chunks; taxcollector and consensus are the single-threaded restatements of the Perl.
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
	o_blast_stats st = { 1.28, 0.46, 0.85, db.total, db.nseq };
	obuf hits_txt;
	obuf_init(&hits_txt);
	for (size_t i = 0; i < hv.n; i++)
		o_blast_format_hit(&hv.h[i], &q, &db, &st, &hits_txt);
	res->format_s = now_s() - t0;

	t0 = now_s();
	o_taxdb tax;
	o_tax_open(&tax, taxdir);
	obuf cls, rep;
	obuf_init(&cls);
	obuf_init(&rep);
	int rc = o_taxcollect_buf(&tax, hits_txt.p ? hits_txt.p : "", hits_txt.n, &cls, &rep);
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
