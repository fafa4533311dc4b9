// trim2 — drop-in for `perl Trim/trim2.3.pl -a reads_1 [-b reads_2] [-g GAP] [-t TRUNCATE]` (README.md:34; trim2.4.pl is
// the same program on FASTQ / QSEQ input): Getopt::Std's getopts('a:b:g:t:q:qc:lc:j') — letters a b g t q c take a
// value, l and j are flags (trim2.4.pl:51) — same stdout, same files: output_files/trim2/<basename>_runblast.fasta in
// the working directory (:112-115) and, for QSEQ input, an empty <dirname>/singletons/<basename>_single.txt (:176-178).
#include <libgen.h>
#include <sys/stat.h>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <string>
#include "pangea_hip.h"

static void mkdir_p(const std::string &path)
{
	for (size_t i = 1; i <= path.size(); i++)
		if (i == path.size() || path[i] == '/')
			mkdir(path.substr(0, i).c_str(), 0777);
}

int main(int argc, char **argv)
{
	pgx_trim_opts o;
	memset(&o, 0, sizeof o);
	const char *ignored = nullptr;
	int a = 1;
	while (a < argc && argv[a][0] == '-' && argv[a][1]) { // getopts stops at the first non-option word
		if (strcmp(argv[a], "--") == 0)
			break;
		const char *p = argv[a] + 1;
		a++;
		while (*p) {
			const char c = *p++;
			const char **dst = c == 'a' ? &o.a : c == 'b' ? &o.b : c == 'g' ? &o.g : c == 't' ? &o.t : c == 'q' ? &o.q : c == 'c' ? &ignored : nullptr;
			if (dst) {
				if (*p)
					*dst = p;
				else if (a < argc)
					*dst = argv[a++];
				break;
			} else if (c == 'j') {
				o.j = 1;
			} else if (c != 'l') {
				fprintf(stderr, "Unknown option: %c\n", c);
			}
		}
	}
	char *log = nullptr, *fasta = nullptr;
	size_t fasta_len = 0;
	int mode = PGX_TRIM_NONE;
	const int rc = pgx_trim_file(&o, &log, &fasta, &fasta_len, &mode);
	int status = 0;
	if (rc < 0) {
		fprintf(stderr, "trim2: %s\n", pgx_last_error());
		status = 3;
	} else {
		if (fasta) {
			std::string b1 = o.a, b2 = o.a;
			const std::string prefix = basename(&b1[0]), dir = dirname(&b2[0]);
			mkdir_p("output_files/trim2");
			const std::string path = "output_files/trim2/" + prefix + "_runblast.fasta";
			FILE *f = fopen(path.c_str(), "wb");
			if (!f || (fasta_len && fwrite(fasta, 1, fasta_len, f) != fasta_len) || fclose(f)) {
				fprintf(stderr, "trim2: cannot write %s: %s\n", path.c_str(), strerror(errno)); // `or die $!` (:115)
				status = 2;
			}
			if (mode == PGX_TRIM_FASTQ)
				fwrite(fasta, 1, fasta_len, stdout);
			if (mode == PGX_TRIM_QSEQ) {
				mkdir_p(dir + "/singletons");
				FILE *s = fopen((dir + "/singletons/" + prefix + "_single.txt").c_str(), "wb");
				if (s)
					fclose(s);
			}
		}
		if (log)
			fputs(log, stdout);
	}
	pgx_free(log);
	pgx_free(fasta);
	return status;
}
