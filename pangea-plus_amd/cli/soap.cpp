// soap — `soap -a reads -D ref.fa.index -o out [-u unmapped] [-M 4] [-r 0|1|2] [-n 5] [-p N]`
// (reference README.md:134, soap.man:29-83). -p is accepted and ignored: the GPU does the work.
// Paired-end: `-b mates -2 unpaired [-m 400] [-x 600]` (soap.man:29-50).
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include "pangea_hip.h"

int main(int argc, char **argv)
{
	pgx_soap_opts o = { nullptr, nullptr, nullptr, nullptr, 4, 1, 5, 0, nullptr, nullptr, 400, 600 };
	int c;
	while ((c = getopt(argc, argv, "a:D:o:u:M:r:n:p:tb:2:m:x:l:s:v:g:R")) != -1) {
		switch (c) {
		case 'a': o.reads_path = optarg; break;
		case 'D': o.db_prefix = optarg; break;
		case 'o': o.out_path = optarg; break;
		case 'u': o.unmapped_path = optarg; break;
		case 'M': o.match_mode = atoi(optarg); break;
		case 'r': o.repeat_mode = atoi(optarg); break;
		case 'n': o.max_n = atoi(optarg); break;
		case 't': o.report_id = 1; break;
		case 'p': break;
		case 'b': o.reads_b_path = optarg; break;
		case '2': o.unpaired_path = optarg; break;
		case 'm': o.min_insert = atoi(optarg); break;
		case 'x': o.max_insert = atoi(optarg); break;
		case 'R':
			fprintf(stderr, "soap: -R (long-insert pairs, RF orientation) is not implemented\n");
			return 1;
		default: break;
		}
	}
	if (!o.reads_path || !o.db_prefix || !o.out_path) {
		fprintf(stderr, "Usage: soap -a <query.file.a> -D <in.fasta.index> -o <alignment.output> [options]\n");
		return 1;
	}
	if (pgx_soap_run(&o) < 0) {
		fprintf(stderr, "soap: %s\n", pgx_last_error());
		return 2;
	}
	return 0;
}
