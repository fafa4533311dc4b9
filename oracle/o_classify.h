/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * Classify-stage restatement: sequence sets, the synthetic workload generator,
 * BLAST mode (spec "pgx-blastn v1", parity UNPINNED: BLAST+ 2.2.26 is not vendored by
 * the reference, Classify/Runblast/install_blast.sh:67; call sites README.md:96,
 * Scripts/run_multi_blastn.pl:56, Scripts/submit_MPI-blast.job:24) and SOAP mode
 * (pinned by goldens made with the reference's closed soap ELF, README.md:130-134).
 */
#ifndef PGX_ORACLE_CLASSIFY_H
#define PGX_ORACLE_CLASSIFY_H
#include "o_common.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---------- sequence sets ---------- */
#define O_AMB 4 /* any letter other than A C G T U: matches nothing, not even itself */

typedef struct {
	int64_t nseq;
	char **header;  /* header line without '>' */
	int64_t *off;   /* nseq+1 offsets into base[] */
	uint8_t *base;  /* 0..3 = A C G T, O_AMB */
	int64_t total;
} o_seqset;

int  o_seqset_read_fasta(o_seqset *s, const char *path);
int  o_seqset_from_text(o_seqset *s, const char *text, size_t len);
void o_seqset_free(o_seqset *s);
/* first whitespace-delimited word of the header */
size_t o_seq_id(const char *header, char *out, size_t cap);

/* ---------- synthetic workload (shared spec with the HIP generator) ---------- */
typedef struct {
	uint64_t seed;      /* 0x50414E47 */
	int64_t n_seq;      /* 666667 */
	int32_t seq_len;    /* 1500 */
	int64_t n_genus;    /* 20000 */
	uint64_t read_seed; /* 42 */
	int32_t read_len;   /* 150 */
} o_synth_cfg;

uint64_t o_synth_hash(uint64_t seed, uint64_t tag, uint64_t i, uint64_t j);
void o_synth_default(o_synth_cfg *c);
/* bases of DB sequence i into out[seq_len] (values 0..3) */
void o_synth_db_seq(const o_synth_cfg *c, int64_t i, uint8_t *out);
/* read r: bases into out[read_len]; optionally the truth (source sequence, offset, strand) */
void o_synth_read(const o_synth_cfg *c, int64_t r, uint8_t *out, int64_t *src_seq, int32_t *src_off, int *minus);
/* taxonomy shape derived from n_genus: counts per rank level 0..6 (domain..species) */
void o_synth_tax_counts(const o_synth_cfg *c, int64_t cnt[7]);
int64_t o_synth_taxid(const o_synth_cfg *c, int level, int64_t index);      /* dense ids, root = 1 */
int64_t o_synth_ancestor(const o_synth_cfg *c, int64_t seq, int level);     /* index at `level` of sequence's lineage */
void o_synth_name(const o_synth_cfg *c, int level, int64_t index, char *out); /* scientific name */
/* RDP stream for read r: which of ranks 0..5 (domain..genus) are present (10 % drop-outs) */
int  o_synth_rdp_mask(const o_synth_cfg *c, int64_t r);
/* write nodes.dmp / names.dmp / gi_taxid_nucl.dmp for the synthetic taxonomy into dir */
int  o_synth_write_taxdump(const o_synth_cfg *c, const char *dir);
int  o_synth_write_db_fasta(const o_synth_cfg *c, const char *path, int64_t first, int64_t count);
int  o_synth_write_reads_fasta(const o_synth_cfg *c, const char *path, int64_t first, int64_t count);
int  o_synth_write_rdp(const o_synth_cfg *c, const char *path, int64_t first, int64_t count);

/* ---------- BLAST mode, spec pgx-blastn v2 (v1 = its `-ungapped` form) ---------- */
#define O_BLAST_W 28
#define O_BLAST_REWARD 1
#define O_BLAST_PENALTY (-2)
#define O_BLAST_XDROP 10       /* ungapped: floor(20 bits * ln 2 / 1.28) */
#define O_BLAST_XDROP_GAP 54   /* gapped, final: floor(100 bits * ln 2 / 1.28) */
#define O_BLAST_MAX_TARGETS 500

typedef struct {
	int32_t query;    /* query ordinal */
	int32_t subject;  /* subject ordinal */
	int32_t qstart, qend;   /* 1-based, plus-strand query coordinates, qstart <= qend */
	int32_t sstart, send;   /* 1-based; sstart > send for minus-strand hits */
	int32_t score;          /* raw: floor(matches - 2*mismatches - 2.5*gap columns) */
	int32_t length, mismatch; /* alignment columns; mismatch columns */
	int32_t gapopen, gaps;    /* gap openings; gap columns */
} o_hit;

typedef struct {
	o_hit *h;
	size_t n, cap;
} o_hitvec;

typedef struct {
	double lambda, K, H;
	int64_t db_len, db_nseq;
	double alpha, beta; /* length adjustment: gapped 1/-2 linear: 1.5, -2; ungapped: lambda/H, 0 */
} o_blast_stats;
void o_blast_stats_init(o_blast_stats *st, int64_t db_len, int64_t db_nseq, int gapped);

/* one side of a gapped extension (o_gapped.c): extent in the query / subject, doubled score, differences */
typedef struct {
	int32_t i, j, s2, d;
	int32_t mism, gap_s, gap_q, gapopen; /* gap_s: query letters facing a gap in the subject row; gap_q: the reverse */
} o_gext;
void o_greedy_extend(const uint8_t *a, int32_t M, const uint8_t *b, int32_t N, int step, int prune, o_gext *out);
/* switches of the restatement (defaults 1, 1): gapped = 0 gives `blastn -ungapped` (spec v1); prune = 0 runs the greedy
 * extension without the result-neutral bound cut */
extern int o_blast_gapped, o_blast_prune;
/* S3d (o_dust.c): mask[i] = 1 for the bases of every perfect low-complexity interval; o_blast_dust = 0 is `-dust no` */
void o_dust_mask(const uint8_t *base, int32_t len, uint8_t *mask);
extern int o_blast_dust;

/* all HSPs of one diagonal (spec 4.x): q/s are base arrays, d = s_pos - q_pos */
int  o_blast_run_is_seed(const uint8_t *qmask, int32_t a, int32_t b);
void o_blast_diag_hsps(const uint8_t *q, const uint8_t *qmask, int32_t qlen, const uint8_t *s, int32_t slen, int64_t d,
		       void (*emit)(void *ctx, int32_t qlo, int32_t qhi, int32_t score, int32_t mism, int32_t seed), void *ctx);
/* search every query (both strands) against db; hits come back in the spec's output order */
int  o_blast_search(const o_seqset *queries, const o_seqset *db, o_hitvec *out, int threads);
int64_t o_blast_length_adjust(const o_blast_stats *st, int64_t qlen);
double o_blast_evalue(const o_blast_stats *st, int64_t qlen, int32_t score);
double o_blast_bitscore(const o_blast_stats *st, int32_t score);
void o_blast_format_evalue(double e, char out[32]);
void o_blast_format_bitscore(double b, char out[32]);
void o_blast_format_hit(const o_hit *h, const o_seqset *queries, const o_seqset *db, const o_blast_stats *st, obuf *out);
int  o_blastn_files(const char *query_fa, const char *db_fa, const char *out_path, int threads);

/* ---------- SOAP mode ---------- */
typedef struct {
	int match_mode; /* -M: 0,1,2 exact counts; 4 best (default) */
	int repeat;     /* -r: 0 none, 1 one, 2 all */
	int max_n;      /* -n: reads with more N are dropped (5) */
	int id_only;    /* -t */
} o_soap_opts;
int o_soap_files(const char *reads_fa, const char *ref_fa, const char *out_path, const char *unmapped_or_null,
		 const o_soap_opts *opt);
/* paired-end mode (soap.man:29-50): -a A -b B -o paired -2 unpaired [-u unmapped] -m MIN -x MAX; -3: a mate under 27 bases */
int o_soap_pe_files(const char *a_fa, const char *b_fa, const char *ref_fa, const char *out_path, const char *unpaired_path,
		    const char *unmapped_or_null, const o_soap_opts *opt, int min_ins, int max_ins);

int o_classify_main(int argc, char **argv);

/* ---------- Megaclust/megaclust2.pl, Megaclustable/megaclustable.pl (o_megaclust.c) ---------- */
typedef struct {
	const char *i, *o, *s, *e, *b, *c, *d; /* raw option texts, NULL = not given (getopts 'i:o:s:e:b:c:d:h') */
	int h;
} o_megaclust_opts;
double o_perl_num(const char *s, size_t n);
int o_megaclust2(const o_megaclust_opts *o, obuf *log);
int o_megaclust2_main(int argc, char **argv, obuf *log);    /* argv[0] is the program name */
int o_megaclustable_main(int argc, char **argv, obuf *log);

/* ---------- Trim/trim2.4.pl, FASTQ and QSEQ inputs (o_trim.c) ---------- */
typedef struct {
	const char *a, *b, *g, *t, *q; /* raw option texts, NULL = not given (getopts 'a:b:g:t:q:qc:lc:j') */
	int j;
} o_trim_opts;
int o_trim2(const o_trim_opts *o, obuf *out, obuf *fasta, int *fasta_made, int *qseq);
int o_trim2_main(int argc, char **argv, obuf *out); /* argv[0] is the program name; writes the script's files */
#ifdef __cplusplus
}
#endif
#endif
