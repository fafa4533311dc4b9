// tax_class — drop-in for the reference's Tax_class/tax_class (ncbitc.c:860-1004): same flags, same
// stdout/stderr text, same exit status; files are looked up in the current directory (ncbitc.c:7-13).
#include <cstdio>
#include "pangea_hip.h"

int main(int argc, char **argv)
{
	char *out = nullptr, *err = nullptr;
	int rc = pgx_tax_cli(argc, argv, ".", &out, &err);
	if (out) fputs(out, stdout);
	if (err) fputs(err, stderr);
	pgx_free(out);
	pgx_free(err);
	return rc;
}
