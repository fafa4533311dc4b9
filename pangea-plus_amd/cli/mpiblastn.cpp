// mpiblastn — `mpirun -np N mpiblastn <in.fasta> <db> <out.txt> <N>` (reference Scripts/submit_MPI-blast.job:24,
// Scripts/submit_multiple_MPI-blast.job:24): the query file is cut into N contiguous blocks of reads, block r is
// searched by rank r, the rank outputs concatenated in rank order are the one-process `blastn -outfmt 6` table.
//
// Here one process per GPU replaces one MPI rank per core: this program IS the launcher (no mpirun in front of it).
// The parent never touches a GPU.  It starts N children -- this same executable with a hidden `--rank r` -- child r
// binds to device r mod (number of devices) with pgx_init() and runs pgx_blastn_run(rank = r, world_size = N) into
// `<out>.rank<r>`; the parent appends the rank files to <out> in rank order, removes them and exits with the worst
// child status.  At most PGX_MPI_PER_GPU (default 4) children share a device at a time; further ranks wait for a slot
// (N = 8 on a one-GPU machine runs in two waves, on an 8-GPU node all at once).
// Extra flags after the four positional arguments are blastn's: -ungapped, -dust no.
#include <signal.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pangea_hip.h"

extern char **environ;

// a signal that ends the launcher ends its ranks too (mpirun does the same for the job of Scripts/submit_MPI-blast.job)
static volatile sig_atomic_t g_stop = 0;
static void on_signal(int sig) { g_stop = sig; }

static int usage()
{
	fprintf(stderr, "USAGE\n  mpiblastn <query.fasta> <database_name> <File_Out> <number of processes> [-ungapped] [-dust no]\n");
	return 1;
}

static std::string self_path()
{
	char buf[4096];
	const ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
	return n > 0 ? std::string(buf, (size_t)n) : std::string();
}

// `mpiblastn --devices`: the one place that asks the HIP runtime, in a process of its own
static int count_devices_in_child(const std::string &self)
{
	int fds[2];
	if (pipe(fds) != 0)
		return -1;
	posix_spawn_file_actions_t fa;
	posix_spawn_file_actions_init(&fa);
	posix_spawn_file_actions_adddup2(&fa, fds[1], 1);
	posix_spawn_file_actions_addclose(&fa, fds[0]);
	const char *av[] = { self.c_str(), "--devices", nullptr };
	pid_t pid;
	const int rc = posix_spawn(&pid, self.c_str(), &fa, nullptr, (char *const *)av, environ);
	posix_spawn_file_actions_destroy(&fa);
	close(fds[1]);
	if (rc != 0) {
		close(fds[0]);
		return -1;
	}
	char buf[64] = "";
	ssize_t got = 0, k;
	while (got < (ssize_t)sizeof buf - 1 && (k = read(fds[0], buf + got, sizeof buf - 1 - (size_t)got)) > 0)
		got += k;
	close(fds[0]);
	int st = 0;
	waitpid(pid, &st, 0);
	return atoi(buf);
}

static int run_rank(int argc, char **argv)
{
	// mpiblastn --rank r --gpu g <query> <db> <out> <N> [flags]
	const int rank = atoi(argv[2]), gpu = atoi(argv[4]);
	pgx_blastn_opts o = { argv[5], argv[6], argv[7], 6, rank, atoi(argv[8]), 0, 0 };
	for (int i = 9; i < argc; i++) {
		if (!strcmp(argv[i], "-ungapped"))
			o.ungapped = 1;
		else if (!strcmp(argv[i], "-dust") && i + 1 < argc)
			o.no_dust = !strcmp(argv[++i], "no");
	}
	if (pgx_init(gpu) < 0 || pgx_blastn_run(&o) < 0) {
		fprintf(stderr, "mpiblastn rank %d (device %d): %s\n", rank, gpu, pgx_last_error());
		return 2;
	}
	return 0;
}

int main(int argc, char **argv)
{
	if (argc == 2 && !strcmp(argv[1], "--devices")) {
		printf("%d\n", pgx_device_count());
		return 0;
	}
	if (argc >= 9 && !strcmp(argv[1], "--rank") && !strcmp(argv[3], "--gpu"))
		return run_rank(argc, argv);
	if (argc < 5)
		return usage();
	const char *query = argv[1], *dbname = argv[2], *out = argv[3];
	char *end = nullptr;
	const long N = strtol(argv[4], &end, 10);
	if (!end || *end || N < 1 || N > 4096) {
		fprintf(stderr, "mpiblastn: the number of processes must be 1..4096, not '%s'\n", argv[4]);
		return usage();
	}
	// the files are checked here so that one message comes out instead of N
	if (FILE *f = fopen(query, "rb"))
		fclose(f);
	else {
		fprintf(stderr, "mpiblastn: cannot open query file %s\n", query);
		return 2;
	}
	const std::string self = self_path();
	if (self.empty()) {
		fprintf(stderr, "mpiblastn: cannot find my own executable\n");
		return 2;
	}
	const int n_dev = count_devices_in_child(self);
	if (n_dev <= 0) {
		fprintf(stderr, "mpiblastn: no HIP device (libpangea_hip has no CPU path)\n");
		return 2;
	}
	FILE *fo = fopen(out, "wb");
	if (!fo) {
		fprintf(stderr, "mpiblastn: cannot open %s for writing\n", out);
		return 2;
	}
	int per_gpu = 4;
	if (const char *e = getenv("PGX_MPI_PER_GPU"))
		per_gpu = atoi(e) > 0 ? atoi(e) : per_gpu;

	struct sigaction sa;
	memset(&sa, 0, sizeof sa);
	sa.sa_handler = on_signal; // (no SA_RESTART: waitpid returns EINTR and the loop below sees g_stop)
	sigaction(SIGINT, &sa, nullptr);
	sigaction(SIGTERM, &sa, nullptr);
	sigaction(SIGHUP, &sa, nullptr);
	std::vector<pid_t> pid((size_t)N, (pid_t)-1);
	std::vector<int> status((size_t)N, -1);
	std::vector<int> busy((size_t)n_dev, 0);
	auto rank_file = [&](long r) { return std::string(out) + ".rank" + std::to_string(r); };
	long done = 0, running = 0;
	int worst = 0;
	bool stopping = false;
	while (done < N) {
		if (g_stop && !stopping) {
			stopping = true;
			fprintf(stderr, "mpiblastn: signal %d: ending the ranks\n", (int)g_stop);
			for (long r = 0; r < N; r++)
				if (pid[(size_t)r] > 0)
					kill(pid[(size_t)r], SIGTERM);
			worst = worst > 128 + (int)g_stop ? worst : 128 + (int)g_stop;
		}
		// start every waiting rank whose device has a free slot (rank r always runs on device r mod n_dev)
		for (long r = 0; r < N && !stopping; r++) {
			if (pid[(size_t)r] != -1)
				continue;
			const int g = (int)(r % n_dev);
			if (busy[(size_t)g] >= per_gpu)
				continue;
			const std::string rs = std::to_string(r), gs = std::to_string(g), ns = std::to_string(N), of = rank_file(r);
			std::vector<const char *> av = { self.c_str(), "--rank", rs.c_str(), "--gpu", gs.c_str(), query, dbname, of.c_str(), ns.c_str() };
			for (int i = 5; i < argc; i++)
				av.push_back(argv[i]);
			av.push_back(nullptr);
			pid_t p;
			const int e = posix_spawn(&p, self.c_str(), nullptr, nullptr, (char *const *)av.data(), environ);
			if (e != 0) {
				fprintf(stderr, "mpiblastn: cannot start rank %ld: %s\n", r, strerror(e));
				pid[(size_t)r] = -2;
				status[(size_t)r] = 2;
				worst = worst > 2 ? worst : 2;
				done++;
				continue;
			}
			pid[(size_t)r] = p;
			busy[(size_t)g]++;
			running++;
		}
		if (running == 0)
			break; // (every rank has ended or could not be started)
		int st = 0;
		const pid_t p = waitpid(-1, &st, 0);
		if (p < 0) {
			if (errno == EINTR)
				continue;
			fprintf(stderr, "mpiblastn: waitpid: %s\n", strerror(errno));
			worst = worst > 2 ? worst : 2;
			break;
		}
		for (long r = 0; r < N; r++)
			if (pid[(size_t)r] == p) {
				const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
				status[(size_t)r] = code;
				pid[(size_t)r] = -2; // ended
				busy[(size_t)(r % n_dev)]--;
				running--;
				done++;
				if (code > worst)
					worst = code;
				if (code)
					fprintf(stderr, "mpiblastn: rank %ld ended with status %d\n", r, code);
				break;
			}
	}
	// rank order = file order (all hits of a read are in its rank's file)
	int rc = worst;
	std::vector<char> buf(1 << 22);
	for (long r = 0; r < N; r++) {
		const std::string of = rank_file(r);
		if (rc == 0) {
			FILE *fi = fopen(of.c_str(), "rb");
			if (!fi) {
				fprintf(stderr, "mpiblastn: rank %ld left no output (%s)\n", r, of.c_str());
				rc = 2;
			} else {
				size_t k;
				while ((k = fread(buf.data(), 1, buf.size(), fi)) > 0)
					if (fwrite(buf.data(), 1, k, fo) != k) {
						fprintf(stderr, "mpiblastn: short write to %s\n", out);
						rc = 2;
						break;
					}
				fclose(fi);
			}
		}
		unlink(of.c_str());
	}
	if (fclose(fo) != 0 && rc == 0) {
		fprintf(stderr, "mpiblastn: cannot close %s\n", out);
		rc = 2;
	}
	return rc;
}
