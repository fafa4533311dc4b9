// consensus — drop-in for `perl Consensus_BLAST_SOAP_RDP-1.1.pl -b B -r R [-s S] -o O`
// (Consensus_BLAST_SOAP_RDP-1.1.pl:8-55): same usage text, same stdout, exit status 0.
#include <cstdio>
#include <unistd.h>
#include "pangea_hip.h"

int main(int argc, char **argv)
{
	const char *b = nullptr, *r = nullptr, *s = nullptr, *o = nullptr;
	int c;
	while ((c = getopt(argc, argv, "b:r:s:o:")) != -1) {
		if (c == 'b') b = optarg;
		else if (c == 'r') r = optarg;
		else if (c == 's') s = optarg;
		else if (c == 'o') o = optarg;
	}
	if (!b || !r || !o) {
		printf("Usage: perl Consensus-1.0.pl \n\t-b Classification results (Blast)\n\t-r Classification results (RDP)\n\t-s Classification results (SOAP2)\n\t-o Output file (txt)\n");
		return 0;
	}
	char *log = nullptr;
	int rc = pgx_consensus_file(b, r, s, o, &log);
	if (log) fputs(log, stdout);
	pgx_free(log);
	if (rc < 0 && rc != PGX_E_IO) {
		fprintf(stderr, "consensus: %s\n", pgx_last_error());
		return 3;
	}
	return 0;
}
