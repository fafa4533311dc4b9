// megaclust2 — drop-in for `perl Megaclust/megaclust2.pl -i consensus.txt -o table.csv [-s -e -b -d -c -h]`
// (megaclust2.pl:33-76; README.md:176): Getopt::Std's getopts('i:o:s:e:b:c:d:h'), same stdout, exit status 0
// (2 where the script dies on an unopenable file).
#include <cstdio>
#include <cstring>
#include "pangea_hip.h"

int main(int argc, char **argv)
{
	pgx_megaclust_opts o;
	memset(&o, 0, sizeof o);
	int a = 1;
	while (a < argc && argv[a][0] == '-' && argv[a][1]) { // getopts stops at the first non-option word
		if (strcmp(argv[a], "--") == 0)
			break;
		const char *p = argv[a] + 1;
		a++;
		while (*p) {
			const char c = *p++;
			const char **dst = c == 'i' ? &o.in_path : c == 'o' ? &o.out_path : c == 's' ? &o.s : c == 'e' ? &o.e
					   : c == 'b' ? &o.b : c == 'c' ? &o.c : c == 'd' ? &o.d : nullptr;
			if (dst) {
				if (*p)
					*dst = p;
				else if (a < argc)
					*dst = argv[a++];
				break;
			} else if (c == 'h') {
				o.help = 1;
			} else {
				fprintf(stderr, "Unknown option: %c\n", c);
			}
		}
	}
	char *log = nullptr;
	const int rc = pgx_megaclust_file(&o, &log);
	if (log)
		fputs(log, stdout);
	pgx_free(log);
	if (rc < 0) {
		fprintf(stderr, "megaclust2: %s\n", pgx_last_error());
		return rc == PGX_E_IO ? 2 : 3;
	}
	return 0;
}
