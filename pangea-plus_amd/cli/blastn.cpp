// blastn — `blastn -query F -db DB -outfmt 6 -out O` (reference README.md:96,
// Scripts/run_multi_blastn.pl:56). -rank/-world_size shard the queries the way mpiblastn's ranks do
// (Scripts/submit_MPI-blast.job:24); concatenating the rank outputs in order gives the 1-process file.
// -gpu g binds the process to device g; without it rank r runs on device r mod (number of devices), so N ranks
// started by hand or by a job script spread over the GPUs of the node (`mpiblastn` is the launcher that does it).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "pangea_hip.h"

int main(int argc, char **argv)
{
	pgx_blastn_opts o = { nullptr, nullptr, nullptr, 6, 0, 1, 0, 0 };
	int gpu = -1;
	for (int i = 1; i < argc; i++)
		if (!strcmp(argv[i], "-ungapped")) // blastn's own flag: ungapped alignments only (spec v1)
			o.ungapped = 1;
	for (int i = 1; i + 1 < argc; i++) {
		if (!strcmp(argv[i], "-query")) o.query_path = argv[++i];
		else if (!strcmp(argv[i], "-db")) o.db_prefix = argv[++i];
		else if (!strcmp(argv[i], "-out")) o.out_path = argv[++i];
		else if (!strcmp(argv[i], "-outfmt")) o.outfmt = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-rank")) o.rank = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-world_size")) o.world_size = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-gpu")) gpu = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-dust")) o.no_dust = !strcmp(argv[++i], "no"); // "yes" and "20 64 1" are the default
		else if (!strcmp(argv[i], "-num_threads")) ++i; // accepted, the GPU does the work
	}
	if (!o.query_path || !o.db_prefix || !o.out_path) {
		fprintf(stderr, "USAGE\n  blastn -query <File_In> -db <database_name> -outfmt 6 -out <File_Out>\n");
		return 1;
	}
	if (gpu < 0) {
		const int n = pgx_device_count();
		gpu = n > 0 && o.rank > 0 ? o.rank % n : (getenv("PGX_DEVICE") ? atoi(getenv("PGX_DEVICE")) : 0);
	}
	if (pgx_init(gpu) < 0 || pgx_blastn_run(&o) < 0) {
		fprintf(stderr, "BLAST engine error: %s\n", pgx_last_error());
		return 2;
	}
	return 0;
}
