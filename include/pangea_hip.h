/*
 * pangea_hip.h — C ABI of libpangea_hip.so, the MI355X (gfx950) classification-and-consensus
 * engine that replaces ONE hot path of PANGEA+ :
 *
 *     Classify/{Runblast,Runsoap}  ->  Tax_class/ncbitc.c (+ NCBI-taxcollector-0.01.pl)
 *                                  ->  Consensus/Consensus_BLAST_SOAP_RDP-1.1.pl
 *
 * The reference has no library / plugin / FFI API: its boundary is process + argv + files +
 * stdout.  Each entry point below is therefore the in-process equivalent of one
 * reference command line, cited as file:line relative to the reference tree; the thin CLIs
 * under pangea-plus_amd/cli/ map argv onto them and print the reference's bytes.
 *
 * Conventions: plain C types only; every function returns 0 or a negative pgx_status and
 * records a message retrievable with pgx_last_error() (thread local).  The caller owns every
 * path and buffer it passes; the library owns the opaque handles until the matching *_close.
 * All compute runs on the GPU selected with pgx_init(); there is NO CPU fallback: without a
 * usable HIP device every compute entry point fails with PGX_E_NODEVICE.
 */
#ifndef PANGEA_HIP_H
#define PANGEA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
	PGX_OK = 0,
	PGX_E_ARG = -1,       /* bad argument */
	PGX_E_IO = -2,        /* file cannot be opened / read / written */
	PGX_E_NODEVICE = -3,  /* no usable gfx950 device, or a HIP call failed */
	PGX_E_FORMAT = -4,    /* malformed input */
	PGX_E_NOMEM = -5,
	PGX_E_REFHANG = -6,   /* input on which the reference never terminates */
	PGX_E_LIMIT = -7      /* documented limit exceeded */
} pgx_status;

const char *pgx_last_error(void);
const char *pgx_version(void);
/* Select the HIP device for the calling process (one process per GPU): what `mpirun -np N` placing one mpiblastn rank
 * per node does in the reference's job file (Scripts/submit_MPI-blast.job:8,24).  Without a call the process uses
 * device $PGX_DEVICE, else device 0.  Every host thread that later enters the library is bound to that device. */
int pgx_init(int device);
int pgx_device_count(void);
int pgx_current_device(void); /* the device this process is bound to, -1 before the first compute call / pgx_init */

/* ------------------------------------------------------------------------------------------
 * Sequence database  —  `makeblastdb -in nt -out nt -dbtype nucl` (README.md:62) and
 * `2bwt-builder ref.fasta` (README.md:130).  One on-disk format serves both classify modes:
 * <prefix>.pgxdb = header, 2-bit packed bases, ambiguity runs, sequence offsets, ids.
 * The k-mer seed index is built on the device when the database is opened.
 * ------------------------------------------------------------------------------------------ */
typedef struct pgx_db pgx_db;
int pgx_db_build(const char *fasta_path, const char *prefix);
int pgx_db_open(const char *prefix, pgx_db **out);
int pgx_db_from_fasta(const char *fasta_path, pgx_db **out); /* build + open without the file */
void pgx_db_close(pgx_db *db);
int64_t pgx_db_num_seqs(const pgx_db *db);
int64_t pgx_db_num_bases(const pgx_db *db);
/* id (first word of the header) of subject i; pointer valid until pgx_db_close */
const char *pgx_db_seq_id(const pgx_db *db, int64_t i);

/* Device-resident arrays of an open database, for the one-off RCCL broadcast of the index
 * (SURVEY 5.8 / 8e).  `ptr` are device addresses owned by the handle. */
typedef struct {
	const char *name;
	void *ptr;
	size_t bytes;
} pgx_device_array;
int pgx_db_device_arrays(pgx_db *db, pgx_device_array *out, int cap); /* returns count */
/* A receiving rank allocates an empty database of the same shape, the caller broadcasts into
 * its arrays, then pgx_db_finish_import() makes it usable. */
typedef struct {
	int64_t n_seq, n_bases;
	int32_t has_amb, index_bits;
	int64_t n_postings;
	int32_t synthetic_ids; /* subject ids are the generator's "gi|<1000+i>|syn|S<i>|" */
} pgx_db_shape;
int pgx_db_get_shape(const pgx_db *db, pgx_db_shape *out);
int pgx_db_alloc_like(const pgx_db_shape *shape, pgx_db **out);
int pgx_db_finish_import(pgx_db *db);
/* Sums over the database as it stands in THIS GPU's memory, for ranks to compare after the one broadcast of a multi-GPU
 * run (the launcher this stands in for: reference Scripts/submit_MPI-blast.job:24; bench.py --dry-ranks): out[0] over the
 * packed base words, out[1] the sequence offsets, out[2] the seed index's bucket offsets, out[3] its postings -- each the
 * wrapping 64-bit sum of element i times (2 i + 1), computed on the device: elements in other places, or in another order,
 * give another sum (postings are position-sorted inside a bucket, so every rank's rebuild gives rank 0's order). */
int pgx_db_checksum(pgx_db *db, uint64_t out[4]);

/* ------------------------------------------------------------------------------------------
 * Classify, BLAST verb  —  `blastn -query F -db DB -outfmt 6 -out O` (README.md:96;
 * Scripts/run_multi_blastn.pl:56) and `mpiblastn in.fasta db out.txt N`
 * (Scripts/submit_MPI-blast.job:24).  Semantics: spec "pgx-blastn v2" (DESIGN.md): megablast seeds
 * (28), ungapped X-drop extension, then the greedy gapped extension and the duplicate-alignment
 * rule; `ungapped` (blastn's own -ungapped flag) stops after the ungapped stage (= spec v1).
 * BLAST+ itself is not vendored by the reference, so parity with NCBI's binary is unpinned.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
	const char *query_path; /* -query */
	const char *db_prefix;  /* -db    */
	const char *out_path;   /* -out   */
	int outfmt;             /* -outfmt, only 6 */
	int rank, world_size;   /* read sharding: this process handles block `rank` of `world_size` */
	int ungapped;           /* -ungapped */
	int no_dust;            /* -dust no (default: -dust "20 64 1", BLAST+'s default: low-complexity stretches of the query seed nothing) */
} pgx_blastn_opts;
int pgx_blastn_run(const pgx_blastn_opts *opts);
/* the same switch for searches through a database handle (pgx_blast_search, pgx_classify_consensus): per handle */
int pgx_db_set_ungapped(pgx_db *db, int ungapped);
int pgx_db_set_dust(pgx_db *db, int dust); /* 1 (default): DUST-masked bases of the reads seed nothing; 0: `-dust no` */
/* 1: every search through this handle computes the DUST window bits of its batch again, on the search's stream, as its
 * first stage -- query masking as `blastn` runs it, inside the search (reference README.md:96); 0 (default): the bits made
 * when the batch was imported are used (the same bits: they depend on the reads alone) */
int pgx_db_set_dust_each_search(pgx_db *db, int on);

/* ------------------------------------------------------------------------------------------
 * Classify, SOAP verb  —  `soap -a reads -D ref.index -o out -p 8 -M 4` (README.md:134;
 * flags soap.man:29-83).  Behaviour pinned by golden vectors from the reference's ELF.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
	const char *reads_path;    /* -a */
	const char *db_prefix;     /* -D (with or without the ".index" suffix) */
	const char *out_path;      /* -o */
	const char *unmapped_path; /* -u, may be NULL */
	int match_mode;            /* -M 4 (best hits, the reference's call) or 0 / 1 / 2 (exactly that many mismatches; reads <= 256 bases) */
	int repeat_mode;           /* -r 0|1|2, default 1 */
	int max_n;                 /* -n, default 5 */
	int report_id;             /* -t: the read's 0-based ordinal in the file instead of its name */
	/* paired-end run (soap.man:29-50; the reference's pipeline does not use it, README.md:134): read i of -a is the mate of
	 * read i of -b.  NULL reads_b_path = single-end.  Mates of 27 to 256 bases, -M 4, no -t (anything else: PGX_E_LIMIT /
	 * PGX_E_ARG, the ELF's behaviour there was not observed or is a crash) */
	const char *reads_b_path;  /* -b */
	const char *unpaired_path; /* -2: the placements of mates without a valid pair (required with -b) */
	int min_insert, max_insert; /* -m / -x, both 0 = soap's defaults 400 / 600 */
} pgx_soap_opts;
int pgx_soap_index(const char *fasta_path); /* writes <fasta>.index.pgxdb */
int pgx_soap_run(const pgx_soap_opts *opts);

/* ------------------------------------------------------------------------------------------
 * Taxonomy database  —  Tax_class/ncbitc.c.  Same file names in `dir` as the reference
 * keeps in its cwd (ncbitc.c:7-13) and byte-compatible .bin files (ncbitc.c:98-140).
 * ------------------------------------------------------------------------------------------ */
typedef struct pgx_taxdb pgx_taxdb;
typedef struct {
	int32_t tax_id, parent_tax_id;
	int8_t rank; /* enum order of ncbitc.c:39-69; -1 unknown */
	char embl_code[3];
	int16_t division_id;
	int8_t inherited_div_flag;
	int16_t genetic_code_id;
	int8_t inherited_GC_flag;
	int32_t mitochondrial_genetic_code_id;
	int8_t inherited_MGC_flag, GenBank_hidden_flag, hidden_subtree_root_flag;
} pgx_node; /* == struct nodes_dmp, 28 bytes */
typedef struct {
	int32_t tax_id;
	char name_txt[64], unique_name[64], name_class[64];
} pgx_name; /* == struct names_dmp, 196 bytes */

int pgx_tax_create(const char *dir);                    /* tax_class -c   (ncbitc.c:701-839, 995-998) */
int pgx_tax_open(const char *dir, pgx_taxdb **out);     /* loads the .bin files, uploads parent/rank arrays */
void pgx_tax_close(pgx_taxdb *db);
int pgx_tax_gi2taxid(const pgx_taxdb *db, int gi, int *taxid);          /* ncbitc.c:567-599 */
int pgx_tax_node(const pgx_taxdb *db, int taxid, pgx_node *out);        /* ncbitc.c:601-628 */
int pgx_tax_names(const pgx_taxdb *db, int taxid, pgx_name *buf, int cap); /* ncbitc.c:647-699; returns count */
int pgx_tax_format_node(const pgx_node *n, char *buf, size_t cap);      /* ncbitc.c:467-493 */
int pgx_tax_format_name(const pgx_name *n, char *buf, size_t cap);      /* ncbitc.c:559-565 */
/* the whole command line of ncbitc.c:860-1004; stdout/stderr text into malloc'd strings the
 * caller frees with pgx_free; returns the process exit status */
int pgx_tax_cli(int argc, char **argv, const char *dir, char **out_text, char **err_text);
void pgx_free(void *p);

/* Batched device lineage walk: for n GIs, gi -> taxid -> parent chain, keeping the nodes whose
 * rank is one of the 8 the driver prints (NCBI-taxcollector-0.01.pl:228-237), leaf first as
 * the Perl pushes them.  lineage[i*PGX_LINEAGE_SLOTS + k] = taxid of the k-th kept node, or
 * PGX_LIN_UNCLASSIFIED for the "[0]Unclassified;" element (taxcollector:289-290);
 * count[i] = number of elements (0 with status[i] != 0 when the gi has taxid 0 or the walk is
 * one the reference cannot finish).  Host pointers in, host pointers out. */
#define PGX_LINEAGE_SLOTS 16
#define PGX_LIN_UNCLASSIFIED (-2)
int pgx_tax_lineage_batch(pgx_taxdb *db, const int32_t *gi, int64_t n, int32_t *lineage, int32_t *count,
			  int32_t *status);

/* Tax annotate  —  `perl NCBI-taxcollector-0.01.pl -f in.tsv -o out.tsv > report.txt`
 * (README.md:109; NCBI-taxcollector-0.01.pl:20-164).  `report` receives the stdout text. */
int pgx_taxcollect_file(pgx_taxdb *db, const char *in_path, const char *out_path, char **report_text);

/* ------------------------------------------------------------------------------------------
 * Consensus  —  `perl Consensus_BLAST_SOAP_RDP-1.1.pl -b B -r R [-s S] -o O`
 * (README.md:152; Consensus_BLAST_SOAP_RDP-1.1.pl:8-244).  -s is opened and never read, as
 * in the reference (Consensus:40-46).  `log_text` receives the stdout text.
 * ------------------------------------------------------------------------------------------ */
int pgx_consensus_file(const char *b, const char *r, const char *s_or_null, const char *o, char **log_text);

/* ------------------------------------------------------------------------------------------
 * Fused device pipeline  (classify -> lineage -> consensus without intermediate files) and
 * the synthetic workload of BASELINE.md section 3, used by bench.py and the parity tests.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
	uint64_t seed;      /* 0x50414E47 */
	int64_t n_seq;      /* 666667 */
	int32_t seq_len;    /* 1500 */
	int64_t n_genus;    /* 20000 */
	uint64_t read_seed; /* 42 */
	int32_t read_len;   /* 150 */
} pgx_synth_cfg;
void pgx_synth_default(pgx_synth_cfg *cfg);
/* database generated directly in HBM (headers "gi|<1000+i>|syn|S<i>|"), index built on device */
int pgx_db_from_synth(const pgx_synth_cfg *cfg, pgx_db **out);
/* nodes.dmp / names.dmp / gi_taxid_nucl.dmp of the synthetic taxonomy, written to dir */
int pgx_synth_write_taxdump(const pgx_synth_cfg *cfg, const char *dir);

typedef struct pgx_reads pgx_reads; /* a batch of reads resident in HBM, both strands packed */
int pgx_reads_from_fasta(const char *path, int64_t first, int64_t count, pgx_reads **out);
/* the same from FASTA text in memory (copied), e.g. pgx_trim_file's fasta_text: Trim -> Classify without the
 * output_files/trim2/..._runblast.fasta file in between (README.md:34 -> :96) */
int pgx_reads_from_fasta_text(const char *text, size_t len, int64_t first, int64_t count, pgx_reads **out);
int pgx_reads_from_synth(const pgx_synth_cfg *cfg, int64_t first, int64_t count, pgx_reads **out);
/* the batch back as FASTA text (">name", one sequence line): the file trim2 hands to blastn (README.md:34 -> :96) */
int pgx_reads_write_fasta(const pgx_reads *r, const char *path);
/* Recompute the DUST window bits (spec S3d) of a resident batch; the import computed the same bits.  BLAST masks its
 * queries inside every search (README.md:96 calls blastn with its defaults); a caller who wants that cost inside the
 * search call rather than inside the import calls this first (bench.py: `dust_in_step`). */
int pgx_reads_redo_dust(pgx_reads *r);
void pgx_reads_close(pgx_reads *r);
int64_t pgx_reads_count(const pgx_reads *r);
/* packed bases of read i (2 bits per base, 32 per word, low bits first) for parity checks */
int pgx_reads_get(const pgx_reads *r, int64_t i, uint8_t *bases_out, int32_t cap, int32_t *len_out);

typedef struct {
	int32_t read, subject;
	int32_t qstart, qend; /* 1-based, plus-strand query coordinates */
	int32_t sstart, send; /* 1-based; sstart > send on the minus strand */
	int32_t score;        /* raw: floor(matches - 2 * mismatches - 2.5 * gap columns) */
	uint16_t mismatch;    /* mismatch columns */
	uint16_t gapopen;     /* gap openings */
} pgx_hit; /* 32 bytes.  The other columns follow from these (PGX_HIT_* below): with q = qend - qstart + 1 and
	    * s = |send - sstart| + 1, differences d = (q + s - 2 * score - ((q + s) & 1)) / 6, gap columns = d - mismatch,
	    * alignment length = (q + s + gap columns) / 2, identities = length - d */
#define PGX_HIT_QSPAN(h) ((h).qend - (h).qstart + 1)
#define PGX_HIT_SSPAN(h) (((h).send > (h).sstart ? (h).send - (h).sstart : (h).sstart - (h).send) + 1)
#define PGX_HIT_DIFFS(h) ((PGX_HIT_QSPAN(h) + PGX_HIT_SSPAN(h) - 2 * (h).score - ((PGX_HIT_QSPAN(h) + PGX_HIT_SSPAN(h)) & 1)) / 6)
#define PGX_HIT_GAPS(h) (PGX_HIT_DIFFS(h) - (int32_t)(h).mismatch)
#define PGX_HIT_LENGTH(h) ((PGX_HIT_QSPAN(h) + PGX_HIT_SSPAN(h) + PGX_HIT_GAPS(h)) / 2)

/* Device hit table of one batch.  Read r owns the slots [offset[r], offset[r + 1]); its rows, in -outfmt 6 order, are the
 * first count[r] of them.  The slots behind them hold what the spec removes AFTER the search: hits that describe the
 * same alignment as an earlier one (spec v2, S3c) and subjects beyond the 500th (S5); no text ever shows them. */
typedef struct pgx_hits pgx_hits;
int pgx_blast_search(pgx_db *db, pgx_reads *reads, pgx_hits **out);
void pgx_hits_close(pgx_hits *h);
int64_t pgx_hits_count(const pgx_hits *h); /* slots */
int pgx_hits_copy(const pgx_hits *h, pgx_hit *out, int64_t cap);           /* device -> host */
int pgx_hits_read_offsets(const pgx_hits *h, int64_t *out, int64_t cap);   /* n_reads+1 slot offsets */
int pgx_hits_read_counts(const pgx_hits *h, int64_t *out, int64_t cap);    /* n_reads row counts */
/* The rows of reads [first_read, first_read + n_reads) as a table of its own (read numbers and offsets from 0): what a
 * caller formats, downloads or hands to the next tool piece by piece when the whole table of a batch (9 GB for 10 M reads)
 * is too much at once -- the table of one `blastn` process is one file in the reference (README.md:96), a piece of it here.
 * Consensus records of the window are the batch's with `hit` minus the first offset of the window. */
int pgx_hits_slice(const pgx_hits *h, int64_t first_read, int64_t n_reads, pgx_hits **out);
/* -outfmt 6 text of the table, malloc'd (pgx_free) */
int pgx_hits_format(const pgx_hits *h, const pgx_db *db, const pgx_reads *reads, char **text, size_t *len);

/* subject -> lineage binding: resolves every subject id "gi|N|..." through the taxonomy on the
 * device (one walk per subject) and keeps per-subject lineage text + token ids in HBM */
int pgx_db_bind_taxonomy(pgx_db *db, pgx_taxdb *tax);
const char *pgx_db_subject_lineage(const pgx_db *db, int64_t subject);

/* RDP stream of a read batch: the five-tab text format (Consensus:126-132) or the synthetic one */
typedef struct pgx_rdp pgx_rdp;
int pgx_rdp_from_file(const char *path, const pgx_reads *reads, const pgx_db *db, pgx_rdp **out);
int pgx_rdp_from_synth(const pgx_synth_cfg *cfg, int64_t first, int64_t count, const pgx_db *db, pgx_rdp **out);
/* the assignments of a batch as the classifier's five-tab text (Consensus:126-132); pgx_rdp_from_file reads it back */
int pgx_rdp_write_file(const pgx_rdp *rdp, const pgx_reads *reads, const pgx_db *db, const char *path);
void pgx_rdp_close(pgx_rdp *r);

typedef struct {
	int32_t hit;     /* index into the hit table of the winning hit, -1 when the read has none */
	int32_t matches; /* "#Matches found: N" */
} pgx_consensus_rec;
/* per read: Consensus arg-max over the read's hits (Consensus:141-234) on the device */
int pgx_consensus_batch(const pgx_db *db, const pgx_hits *hits, const pgx_rdp *rdp, pgx_consensus_rec *out,
			int64_t cap);
/* the whole hot path for one resident batch: search + lineage + consensus, results left in HBM;
 * `out` may be NULL (bench) */
int pgx_classify_consensus(pgx_db *db, pgx_reads *reads, const pgx_rdp *rdp, pgx_hits **hits_out,
			   pgx_consensus_rec *out, int64_t cap);
/* the same with the third classifier stream of `Consensus_BLAST_SOAP_RDP-1.1.pl -s` (BASELINE config 5): the SOAP table
 * must be openable (PGX_E_IO with the script's message otherwise) and is otherwise ignored, as in the reference
 * (Consensus:40-46); NULL, "" and "0" mean "not given" (Perl truth) */
int pgx_classify_consensus_tri(pgx_db *db, pgx_reads *reads, const pgx_rdp *rdp, const char *soap_stream_path,
			       pgx_hits **hits_out, pgx_consensus_rec *out, int64_t cap);
/* ---- opt-in extension: a real three-way vote, spec "pgx-vote3 v1".  NOT reference behaviour: the
 * reference's Consensus opens the SOAP table and never reads it (Consensus_BLAST_SOAP_RDP-1.1.pl:40-46).  Per read with an
 * RDP line: B = lineage of its first BLAST row (best hit), S = lineage of its first row in the SOAP table `soap_path`
 * (soap.man: column 1 read, column 8 reference id), R = the RDP assignment; rank k is agreed when two of the three names
 * are equal and not empty; the record is the longest prefix of agreed ranks. */
typedef struct {
	int32_t depth;    /* ranks agreed from the domain down (0..7); -1: the read has no RDP line */
	uint32_t name[7]; /* token id of the agreed name per rank (text: pgx_vote3_format) */
	uint8_t votes[7]; /* 2 or 3 per agreed rank */
	uint8_t pad;
} pgx_vote_rec;
int pgx_vote3_batch(const pgx_db *db, const pgx_reads *reads, const pgx_hits *hits, const pgx_rdp *rdp, const char *soap_path,
		    pgx_vote_rec *out, int64_t cap);
/* "<read>\t[0]Name;[1]Name;...\t<depth>\t<votes as digits>\n" per read with an RDP line, malloc'd */
int pgx_vote3_format(const pgx_db *db, const pgx_reads *reads, const pgx_vote_rec *recs, int64_t n, char **text, size_t *len);

/* consensus text ("<hit line with lineage>\n#Matches found: N\n" per read), malloc'd */
int pgx_consensus_format(const pgx_db *db, const pgx_reads *reads, const pgx_hits *hits,
			 const pgx_consensus_rec *recs, int64_t n, char **text, size_t *len);
/* the same text written to `path` -- the `-o` file of Consensus_BLAST_SOAP_RDP-1.1.pl:52 -- piece by piece while the next
 * piece is rendered (no copy of the whole text on the host); *bytes = its size */
int pgx_consensus_format_file(const pgx_db *db, const pgx_reads *reads, const pgx_hits *hits,
			      const pgx_consensus_rec *recs, int64_t n, const char *path, size_t *bytes);

/* ---- after the consensus ------------------------------------------------------
 * Megaclust/megaclust2.pl:33-163 `-i consensus.txt -o table.csv [-s PCT] [-e EVALUE] [-b BITS] [-d DELIM] [-c X] [-h]`
 * (README.md:176).  The option texts are passed as the script's getopts('i:o:s:e:b:c:d:h') would hold them (NULL =
 * not given; Perl truth: "" and "0" count as not given, megaclust2.pl:48,61,66,71,139).  `log_text` receives the
 * bytes the script prints on stdout (run summary :161-163, or its usage/complaint :37-58), malloc'd.
 * The data lines of the table come in first-counted order; the script prints them in Perl hash order (`keys`, :155),
 * which is not defined. */
typedef struct {
	const char *in_path;  /* -i */
	const char *out_path; /* -o */
	const char *s, *e, *b, *c, *d;
	int help;             /* -h */
} pgx_megaclust_opts;
int pgx_megaclust_file(const pgx_megaclust_opts *o, char **log_text);
/* the same table straight from the consensus records of a batch in HBM (in_path / out_path are ignored):
 * `csv_text` is what megaclust2.pl would write for pgx_consensus_format()'s text of the same batch */
int pgx_megaclust_batch(const pgx_db *db, const pgx_reads *reads, const pgx_hits *hits, const pgx_consensus_rec *recs,
			int64_t n, const pgx_megaclust_opts *o, char **csv_text, size_t *csv_len, char **log_text);
/* Megaclustable/megaclustable.pl:17-128 `-m table.csv ... -t LEVEL -o out.txt` (README.md:185): argv is the
 * script's @ARGV (no program name).  Byte-identical output file; `log_text` = its stdout. */
int pgx_megaclustable(int argc, const char *const *argv, char **log_text);

/* Columns 11-12 (e-value, bit score) of a `blastn -outfmt 6` row (README.md:96) for a raw score, as the row formatter
 * prints them: lambda 1.28, K 0.46 (reward 1 / penalty -2), BLAST+ tabular number formats.  Host arithmetic only.
 * The reference's only record of its BLAST dependency's output, validation_dataset/Data-set_2_consensus.xlsx
 * (10 992 rows), is the known-answer set for it (tests/golden/blast_rows). */
int pgx_blast_score_columns(int32_t score, int64_t qlen, int64_t db_len, int64_t db_nseq, char evalue[32], char bits[32]);
/* the same with the statistics named: gapped != 0: spec S4 (the default search); 0: S4u, the columns of `blastn -ungapped` */
int pgx_blast_score_columns_v(int32_t score, int64_t qlen, int64_t db_len, int64_t db_nseq, int gapped, char evalue[32], char bits[32]);

/* --- the step before the path: Trim/trim2.4.pl == trim2.3.pl on FASTQ and QSEQ reads ---------
 * `perl trim2.3.pl -a reads_1 [-b reads_2] [-g GAP] [-t TRUNCATE]` (README.md:34; trim2.4.pl:46-167): the raw
 * getopts values (NULL = option not given; `-qc`, `-lc`, `-q`, `-j` have no effect on these formats, trim2.4.pl:51).
 * Line index, the running-sum quality rule (trim2.4.pl:529-578 FASTQ, :253-298 QSEQ), mate joining with GAP N's
 * (:228-245, :502-505) and the FASTA text are computed on the device.
 *   fasta_text  what the script writes to output_files/trim2/<basename of -a>_runblast.fasta (:115); NULL when the
 *               script exits before opening it (usage, unopenable file)
 *   log_text    the script's messages; its stdout is fasta_text followed by log_text in FASTQ mode (:487-515 print
 *               every record to both), log_text alone otherwise
 *   mode        PGX_TRIM_*; for PGX_TRIM_QSEQ the script also creates <dirname of -a>/singletons/<basename>_single.txt,
 *               empty (:176-178)
 * FASTA-format input (:117-143): with `-q QUAL` parse_fasta (:384-465) -- mode PGX_TRIM_FASTA_QUAL, everything it prints
 * is in log_text (the -q value, the records, the closing message) and fasta_text is the empty file it leaves; with `-j`
 * and `-b` join_fasta (:301-382, trim2.4.pl's text) -- mode PGX_TRIM_FASTA_JOIN, fasta_text is the joined FASTA, no
 * messages.  A negative -t returns PGX_E_ARG. */
typedef struct {
	const char *a, *b, *g, *t, *q;
	int j;
} pgx_trim_opts;
enum { PGX_TRIM_NONE = 0, PGX_TRIM_FASTQ = 1, PGX_TRIM_QSEQ = 2, PGX_TRIM_UNKNOWN = 3, PGX_TRIM_FASTA_QUAL = 4, PGX_TRIM_FASTA_JOIN = 5 };
int pgx_trim_file(const pgx_trim_opts *o, char **log_text, char **fasta_text, size_t *fasta_len, int *mode);

/* diagnostics (tools/probe_gather.py): 64-byte lines per second the device delivers to random 8-byte lane loads over a
 * table of `table_bytes` — the access shape of the seed stage's index and database fetches, i.e. the roof that stage
 * can be measured against instead of the streaming bandwidth */
int pgx_probe_gather(uint64_t table_bytes, int stream, double *lines_per_s, double *ms);

/* diagnostics (tools/probe_issue.py, bench.py's roofline.issue): what a SIMD issues for the gapped stage's instruction
 * mix, one instruction kind at a time (kinds 0..7: see csrc/probe.hip), with
 * `waves_per_simd` (1..8) resident wavefronts.  out[0] = vector wave-instructions per second per SIMD, out[1] = shader
 * cycles per vector instruction of one wavefront, out[2] = kernel ms, out[3] = shader clock (Hz).  The roof the gapped half of `blastn` (reference
 * README.md:96) is measured against next to the random-line roof (pgx_probe_gather): DESIGN.md section 7 found that stage
 * bound by single-use 64-byte lines first and by instruction issue second */
int pgx_probe_issue(int waves_per_simd, int kind, double *out);
const char *pgx_probe_issue_name(int kind); /* NULL past the last kind */

/* instrumentation for bench.py: HIP-event time (ms) of the kernels of the last pipeline call */
typedef struct {
	float seed_extend_ms, group_ms, sort_ms, consensus_ms, total_ms;
	int64_t probes, postings, candidates, hits;
	int64_t survivors; /* postings that pass the duplicate filter and get a diagonal mask built */
	float gapped_ms;     /* spec v2: the gapped stage between seed_extend and group */
	int64_t gapped_wide; /* HSPs the first tier of the gapped stage listed for the later ones */
	float dust_ms;       /* S3d recomputed inside the search (pgx_db_set_dust_each_search), else 0; part of total_ms */
	int32_t attempts;    /* 1, or more when a table or list of the step was too small and the step was repeated with larger ones */
} pgx_stage_times;
int pgx_last_stage_times(pgx_stage_times *out);

#ifdef __cplusplus
}
#endif
#endif /* PANGEA_HIP_H */
