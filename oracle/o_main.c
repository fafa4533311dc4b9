/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 * Multi-call command line for the CPU restatement:
 *   pgx_oracle tax_class [-c|-s GI|-g GI|-t TAXID|-n TAXID|-v|-h]   (cwd = dump dir, as ncbitc.c:7-13)
 *   pgx_oracle taxcollector -f in.tsv -o out.tsv [-d taxdir]        (taxdir defaults to ./Tax_class)
 *   pgx_oracle consensus -b blast_class.tsv -r rdp.tsv [-s soap] -o out.txt
 *   pgx_oracle megaclust2 -i consensus.txt -o table.csv [-s -e -b -d -c -h]   (Megaclust/megaclust2.pl)
 *   pgx_oracle megaclustable -m a.csv b.csv ... -t LEVEL -o table.txt          (Megaclustable/megaclustable.pl)
 *   pgx_oracle trim2 -a reads [-b mates] [-g GAP] [-t TRUNCATE]                 (Trim/trim2.4.pl, FASTQ / QSEQ input)
 */
#include "o_common.h"
#include "o_classify.h"
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int run_taxcollector(int argc, char **argv)
{
	const char *f = NULL, *o = NULL, *d = "./Tax_class";
	int c;
	optind = 1;
	while ((c = getopt(argc, argv, "f:o:d:")) != -1) {
		if (c == 'f') f = optarg;
		else if (c == 'o') o = optarg;
		else if (c == 'd') d = optarg;
	}
	if (!f || !o) {
		/* NCBI-taxcollector-0.01.pl:22-27 */
		printf("Usage: perl taxcollector_ncbi-0.01.pl \n\t-f Classification results (tabular text file)\n\t-o Output file \n");
		return 0;
	}
	o_taxdb db;
	o_tax_open(&db, d);
	obuf rep;
	obuf_init(&rep);
	int rc = o_taxcollect_file(&db, f, o, &rep);
	fwrite(rep.p ? rep.p : "", 1, rep.n, stdout);
	obuf_free(&rep);
	o_tax_close(&db);
	return rc < 0 ? 3 : 0;
}

static int run_consensus(int argc, char **argv)
{
	const char *b = NULL, *r = NULL, *s = NULL, *o = NULL;
	int c;
	optind = 1;
	while ((c = getopt(argc, argv, "b:r:s:o:")) != -1) {
		if (c == 'b') b = optarg;
		else if (c == 'r') r = optarg;
		else if (c == 's') s = optarg;
		else if (c == 'o') o = optarg;
	}
	if (!b || !r || !o) {
		/* Consensus_BLAST_SOAP_RDP-1.1.pl:10-17 */
		printf("Usage: perl Consensus-1.0.pl \n\t-b Classification results (Blast)\n\t-r Classification results (RDP)\n\t-s Classification results (SOAP2)\n\t-o Output file (txt)\n");
		return 0;
	}
	obuf log;
	obuf_init(&log);
	int rc = o_consensus_file(b, r, s, o, &log);
	fwrite(log.p ? log.p : "", 1, log.n, stdout);
	obuf_free(&log);
	return rc < 0 ? 3 : 0;
}

int main(int argc, char **argv)
{
	if (argc < 2) {
		fprintf(stderr, "usage: pgx_oracle {tax_class|taxcollector|consensus|makedb|blastn|soap|synth} ...\n");
		return 2;
	}
	const char *verb = argv[1];
	if (strcmp(verb, "megaclust2") == 0 || strcmp(verb, "megaclustable") == 0) {
		obuf log;
		obuf_init(&log);
		int rc = verb[9] == '2' ? o_megaclust2_main(argc - 1, argv + 1, &log) : o_megaclustable_main(argc - 1, argv + 1, &log);
		fwrite(log.p ? log.p : "", 1, log.n, stdout);
		obuf_free(&log);
		return rc < 0 ? 2 : 0;
	}
	if (strcmp(verb, "trim2") == 0) {
		obuf out;
		obuf_init(&out);
		int rc = o_trim2_main(argc - 1, argv + 1, &out);
		fwrite(out.p ? out.p : "", 1, out.n, stdout);
		obuf_free(&out);
		if (rc == -1)
			fprintf(stderr, "pgx_oracle trim2: input or option not covered by the restatement (FASTA-format input, negative -t)\n");
		return rc < 0 ? 2 : 0;
	}
	if (strcmp(verb, "tax_class") == 0) {
		obuf out, err;
		obuf_init(&out);
		obuf_init(&err);
		int rc = o_tax_cli(argc - 1, argv + 1, ".", &out, &err);
		fwrite(out.p ? out.p : "", 1, out.n, stdout);
		fwrite(err.p ? err.p : "", 1, err.n, stderr);
		obuf_free(&out);
		obuf_free(&err);
		return rc;
	}
	if (strcmp(verb, "taxcollector") == 0)
		return run_taxcollector(argc - 1, argv + 1);
	if (strcmp(verb, "consensus") == 0)
		return run_consensus(argc - 1, argv + 1);
	if (strcmp(verb, "vote3") == 0) { /* vote3 BLAST_CLASS RDP SOAP_CLASS OUT: the opt-in three-way vote (pgx-vote3 v1) */
		if (argc != 6) {
			fprintf(stderr, "usage: pgx_oracle vote3 blast_class.tsv rdp.txt soap_class.tsv out.txt\n");
			return 2;
		}
		return o_vote3_file(argv[2], argv[3], argv[4], argv[5]) < 0 ? 3 : 0;
	}
	return o_classify_main(argc - 1, argv + 1);
}
