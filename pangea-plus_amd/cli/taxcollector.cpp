// taxcollector — drop-in for `perl NCBI-taxcollector-0.01.pl -f in -o out > report.txt`
// (NCBI-taxcollector-0.01.pl:20-56). Like the Perl it looks for the taxonomy in ./Tax_class/
// (taxcollector:47); -d overrides that directory.
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include "pangea_hip.h"

int main(int argc, char **argv)
{
	const char *f = nullptr, *o = nullptr, *d = "./Tax_class";
	int c;
	while ((c = getopt(argc, argv, "f:o:d:")) != -1) {
		if (c == 'f') f = optarg;
		else if (c == 'o') o = optarg;
		else if (c == 'd') d = optarg;
	}
	if (!f || !o) {
		printf("Usage: perl taxcollector_ncbi-0.01.pl \n\t-f Classification results (tabular text file)\n\t-o Output file \n");
		return 0;
	}
	pgx_taxdb *db = nullptr;
	if (pgx_tax_open(d, &db) < 0) {
		fprintf(stderr, "taxcollector: %s\n", pgx_last_error());
		return 2;
	}
	char *report = nullptr;
	int rc = pgx_taxcollect_file(db, f, o, &report);
	if (report) fputs(report, stdout);
	pgx_free(report);
	pgx_tax_close(db);
	if (rc < 0 && rc != PGX_E_IO) {
		fprintf(stderr, "taxcollector: %s\n", pgx_last_error());
		return 3;
	}
	return 0; // the Perl exits 0 even when a file cannot be opened (taxcollector:31-34)
}
