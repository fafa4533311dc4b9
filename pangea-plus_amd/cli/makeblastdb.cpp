// makeblastdb — `makeblastdb -in nt -out nt -dbtype nucl` (reference README.md:62): writes <out>.pgxdb
#include <cstdio>
#include <cstring>
#include "pangea_hip.h"

int main(int argc, char **argv)
{
	const char *in = nullptr, *out = nullptr, *type = "nucl";
	for (int i = 1; i + 1 < argc; i++) {
		if (!strcmp(argv[i], "-in")) in = argv[++i];
		else if (!strcmp(argv[i], "-out")) out = argv[++i];
		else if (!strcmp(argv[i], "-dbtype")) type = argv[++i];
	}
	if (!in || strcmp(type, "nucl") != 0) {
		fprintf(stderr, "USAGE\n  makeblastdb -in <File_In> -out <database_name> -dbtype nucl\n");
		return 1;
	}
	if (!out) out = in;
	if (pgx_db_build(in, out) < 0) {
		fprintf(stderr, "makeblastdb: %s\n", pgx_last_error());
		return 2;
	}
	return 0;
}
