// 2bwt-builder — `2bwt-builder ref.fasta` (reference README.md:130): writes <ref.fasta>.index.pgxdb
#include <cstdio>
#include "pangea_hip.h"

int main(int argc, char **argv)
{
	if (argc < 2) {
		fprintf(stderr, "Usage: 2bwt-builder <reference.fasta>\n");
		return 1;
	}
	if (pgx_soap_index(argv[1]) < 0) {
		fprintf(stderr, "2bwt-builder: %s\n", pgx_last_error());
		return 2;
	}
	return 0;
}
