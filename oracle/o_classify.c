/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h). Command-line verbs of the classify stage:
 *   pgx_oracle blastn -query reads.fa -db db.fa -outfmt 6 -out hits.tsv [-num_threads N] [-ungapped] [-no_prune]   (README.md:96)
 *   pgx_oracle soap -a reads.fa -D ref.fa.index -o out.txt [-u unmapped] [-r 0|1|2] [-M 4] [-n 5] [-p N]  (README.md:134)
 *   pgx_oracle synth {db|reads|rdp|taxdump} --out PATH [--n-seq N --seq-len L --n-genus G --first A --count C]
 * The oracle reads the database FASTA directly (no makeblastdb / 2bwt-builder step).
 */
#include "o_classify.h"
#include <stdlib.h>
#include <string.h>

static const char *arg_of(int argc, char **argv, const char *flag, const char *dflt)
{
	for (int i = 1; i + 1 < argc; i++)
		if (strcmp(argv[i], flag) == 0)
			return argv[i + 1];
	return dflt;
}

static int run_blastn(int argc, char **argv)
{
	const char *q = arg_of(argc, argv, "-query", NULL), *d = arg_of(argc, argv, "-db", NULL);
	const char *o = arg_of(argc, argv, "-out", NULL), *f = arg_of(argc, argv, "-outfmt", "6");
	int nt = atoi(arg_of(argc, argv, "-num_threads", "1"));
	for (int i = 1; i < argc; i++) {
		if (strcmp(argv[i], "-ungapped") == 0) /* blastn's own flag: no gapped stage (spec v1) */
			o_blast_gapped = 0;
		if (strcmp(argv[i], "-no_prune") == 0) /* the greedy extension without its result-neutral bound cut */
			o_blast_prune = 0;
		if (strcmp(argv[i], "-dust") == 0 && i + 1 < argc && strcmp(argv[i + 1], "no") == 0) /* blastn -dust no */
			o_blast_dust = 0;
	}
	if (!q || !d || !o || strcmp(f, "6") != 0) {
		fprintf(stderr, "usage: blastn -query F -db DB.fa -outfmt 6 -out O\n");
		return 1;
	}
	return o_blastn_files(q, d, o, nt) ? 1 : 0;
}

static int run_soap(int argc, char **argv)
{
	const char *a = arg_of(argc, argv, "-a", NULL), *D = arg_of(argc, argv, "-D", NULL);
	const char *o = arg_of(argc, argv, "-o", NULL), *u = arg_of(argc, argv, "-u", NULL);
	o_soap_opts opt = { atoi(arg_of(argc, argv, "-M", "4")), atoi(arg_of(argc, argv, "-r", "1")),
			    atoi(arg_of(argc, argv, "-n", "5")), 0 };
	for (int i = 1; i < argc; i++)
		if (strcmp(argv[i], "-t") == 0)
			opt.id_only = 1;
	if (!a || !D || !o) {
		fprintf(stderr, "usage: soap -a reads.fa -D ref.fa.index -o out\n");
		return 1;
	}
	char ref[4096];
	snprintf(ref, sizeof ref, "%s", D);
	size_t n = strlen(ref);
	if (n > 6 && strcmp(ref + n - 6, ".index") == 0)
		ref[n - 6] = '\0';
	/* paired-end: -b B -2 unpaired -m MIN -x MAX (soap.man:29-50; defaults 400 / 600) */
	const char *b = arg_of(argc, argv, "-b", NULL);
	if (b) {
		const int rc = o_soap_pe_files(a, b, ref, o, arg_of(argc, argv, "-2", NULL), u, &opt, atoi(arg_of(argc, argv, "-m", "400")),
					       atoi(arg_of(argc, argv, "-x", "600")));
		if (rc == -3)
			fprintf(stderr, "soap: paired-end mates of fewer than 27 or more than 256 bases are outside what is restated\n");
		return rc ? 1 : 0;
	}
	return o_soap_files(a, ref, o, u, &opt) ? 1 : 0;
}

static int run_synth(int argc, char **argv)
{
	if (argc < 2)
		return 1;
	o_synth_cfg c;
	o_synth_default(&c);
	c.n_seq = atoll(arg_of(argc, argv, "--n-seq", "666667"));
	c.seq_len = atoi(arg_of(argc, argv, "--seq-len", "1500"));
	c.n_genus = atoll(arg_of(argc, argv, "--n-genus", "20000"));
	c.read_len = atoi(arg_of(argc, argv, "--read-len", "150"));
	int64_t first = atoll(arg_of(argc, argv, "--first", "0"));
	int64_t count = atoll(arg_of(argc, argv, "--count", "0"));
	const char *out = arg_of(argc, argv, "--out", NULL);
	if (!out)
		return 1;
	if (strcmp(argv[1], "db") == 0)
		return o_synth_write_db_fasta(&c, out, first, count ? count : c.n_seq) ? 1 : 0;
	if (strcmp(argv[1], "reads") == 0)
		return o_synth_write_reads_fasta(&c, out, first, count) ? 1 : 0;
	if (strcmp(argv[1], "rdp") == 0)
		return o_synth_write_rdp(&c, out, first, count) ? 1 : 0;
	if (strcmp(argv[1], "taxdump") == 0)
		return o_synth_write_taxdump(&c, out) ? 1 : 0;
	return 1;
}

int o_classify_main(int argc, char **argv)
{
	if (strcmp(argv[0], "blastn") == 0)
		return run_blastn(argc, argv);
	if (strcmp(argv[0], "soap") == 0)
		return run_soap(argc, argv);
	if (strcmp(argv[0], "synth") == 0)
		return run_synth(argc, argv);
	fprintf(stderr, "unknown verb %s\n", argv[0]);
	return 2;
}
