/*
 * oracle/ — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the PANGEA+ classify -> tax_class -> consensus hot
 * path.  It is the checker for the HIP product under pangea-plus_amd/ and the
 * "port" CPU baseline of bench.py.  Nothing under pangea-plus_amd/ may include,
 * link or execute anything in this directory.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  Parity status:
 *   - taxdb / taxcollector / consensus : PINNED by golden vectors produced here by
 *     the reference's own C (Tax_class/ncbitc.c) and Perl, see oracle/gen_goldens.py
 *   - SOAP mode                         : PINNED by golden vectors produced by the
 *     reference's closed `soap` ELF (Classify/Runsoap/soap2.21release/soap)
 *   - BLAST mode                        : PARITY UNPINNED (NCBI BLAST+ 2.2.26 is
 *     not vendored by the reference: Classify/Runblast/install_blast.sh:67).  The
 *     restatement follows the public megablast description, spec "pgx-blastn v1".
 */
#ifndef PGX_ORACLE_COMMON_H
#define PGX_ORACLE_COMMON_H

#include <stdint.h>
#include <stdio.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- growable byte buffer (all text outputs are built in memory) ---------- */
typedef struct {
	char *p;
	size_t n, cap;
} obuf;

void obuf_init(obuf *b);
void obuf_free(obuf *b);
void obuf_put(obuf *b, const void *s, size_t n);
void obuf_puts(obuf *b, const char *s);
void obuf_printf(obuf *b, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int obuf_write_file(const obuf *b, const char *path);
char *o_read_file(const char *path, size_t *len); /* malloc'd, NUL-terminated */

/* ---------- taxonomy DB (Tax_class/ncbitc.c) ---------- */
#define O_NODE_SIZE 28   /* sizeof(struct nodes_dmp), ncbitc.c:98-114 */
#define O_NAME_SIZE 196  /* sizeof(struct names_dmp), ncbitc.c:128-133 */

typedef struct {
	int32_t tax_id, parent;
	int8_t rank;
	char embl[3];
	int16_t division;
	int8_t div_flag;
	int16_t gencode;
	int8_t gc_flag;
	int32_t mito;
	int8_t mgc_flag, hidden, subtree;
} o_node;

typedef struct {
	int32_t tax_id;
	char name_txt[64], unique_name[64], name_class[64];
} o_name;

typedef struct {
	int32_t *gi2tax;   size_t n_gi;     /* gi_taxid_nucl.dmp.bin  */
	uint8_t *nodes;    size_t n_nodes;  /* nodes.dmp.bin, 28-byte records */
	uint8_t *names;    int32_t n_names; /* names.dmp.bin, header count + 196-byte records */
	size_t names_file_records;          /* records physically present */
} o_taxdb;

int  o_tax_create(const char *dir);                 /* ncbitc.c:701-839 (-c) */
int  o_tax_open(o_taxdb *db, const char *dir);
void o_tax_close(o_taxdb *db);
int  o_tax_gi2taxid(const o_taxdb *db, int gi, int *taxid);       /* ncbitc.c:567-599 */
int  o_tax_node(const o_taxdb *db, int taxid, o_node *out);       /* ncbitc.c:601-628 */
void o_tax_format_node(const o_node *n, obuf *out);               /* ncbitc.c:467-493 */
int  o_tax_names(const o_taxdb *db, int taxid, obuf *out);        /* ncbitc.c:647-699 */
/* whole CLI: argv as the reference's main (ncbitc.c:860-1004); returns exit status */
int  o_tax_cli(int argc, char **argv, const char *dir, obuf *out, obuf *err);
const char *o_rank_name(int id);                                  /* ncbitc.c:273-398 */
int  o_rank_id(const char *s);                                    /* ncbitc.c:400-465 */

/* ---------- taxcollector (Tax_class/NCBI-taxcollector-0.01.pl) ---------- */
/* lineage for one gi, exactly the text the Perl prints between id and the numeric
 * columns.  Returns 0, or -1 for the inputs on which the reference never
 * terminates (hang #1). `report` receives the stdout progress text. */
int o_taxcollect_lineage(const o_taxdb *db, const char *gi_text, obuf *lineage, obuf *report);
int o_taxcollect_file(const o_taxdb *db, const char *in_path, const char *out_path, obuf *report);
int o_taxcollect_buf(const o_taxdb *db, const char *in, size_t in_len, obuf *out, obuf *report);

/* ---------- consensus (Consensus/Consensus_BLAST_SOAP_RDP-1.1.pl) ---------- */
int o_consensus_buf(const char *blast, size_t blast_len, const char *rdp, size_t rdp_len,
		    obuf *out, obuf *log);
int o_consensus_file(const char *b, const char *r, const char *s_or_null, const char *o, obuf *log);
/* opt-in extension "pgx-vote3 v1" (o_consensus.c): majority of BLAST top hit, SOAP best hit and RDP per rank */
int o_vote3_buf(const char *blast_class, size_t bl, const char *rdp, size_t rl, const char *soap_class, size_t sl, obuf *out);
int o_vote3_file(const char *blast_class, const char *rdp, const char *soap_class, const char *out_path);

#ifdef __cplusplus
}
#endif
#endif
