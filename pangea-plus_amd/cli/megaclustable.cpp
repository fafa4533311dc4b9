// megaclustable — drop-in for `perl Megaclustable/megaclustable.pl -m table.csv ... -t LEVEL -o out.txt`
// (megaclustable.pl:17-52; README.md:185): the script walks @ARGV itself, so the words are passed through.
#include <cstdio>
#include "pangea_hip.h"

int main(int argc, char **argv)
{
	char *log = nullptr;
	const int rc = pgx_megaclustable(argc - 1, argv + 1, &log);
	if (log)
		fputs(log, stdout);
	pgx_free(log);
	if (rc < 0) {
		fprintf(stderr, "megaclustable: %s\n", pgx_last_error());
		return rc == PGX_E_IO ? 2 : 3;
	}
	return 0;
}
