// write-rate probe: 320 MB into a fresh file under the given directory, three ways
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
	const std::string dir = argc > 1 ? argv[1] : "/tmp";
	const size_t N = 320u << 20, piece = 40u << 20;
	char *src = (char *)malloc(N);
	memset(src, 'x', N);
	for (int rep = 0; rep < 2; rep++) {
		for (int T : { 1, 2, 4, 8 }) {
			for (int mode = 0; mode < 3; mode++) {
				if (mode == 0 && T > 1)
					continue;
				const std::string path = dir + "/wb_probe.bin";
				unlink(path.c_str());
				const double t0 = now();
				if (mode == 0) {
					FILE *f = fopen(path.c_str(), "wb");
					for (size_t o = 0; o < N; o += piece)
						fwrite(src + o, 1, piece, f);
					fclose(f);
				} else {
					int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
					for (size_t o = 0; o < N; o += piece) {
						char *m = nullptr;
						if (mode == 1) {
							if (posix_fallocate(fd, (off_t)o, (off_t)piece) != 0)
								perror("fallocate");
							m = (char *)mmap(nullptr, piece, PROT_WRITE, MAP_SHARED, fd, (off_t)o);
							if (m == MAP_FAILED) {
								perror("mmap");
								return 1;
							}
						}
						std::vector<std::thread> th;
						const size_t per = piece / T;
						for (int t = 0; t < T; t++)
							th.emplace_back([=] {
								if (mode == 1)
									memcpy(m + t * per, src + o + t * per, per);
								else
									for (size_t d = 0; d < per;) {
										ssize_t w = pwrite(fd, src + o + t * per + d, per - d, (off_t)(o + t * per + d));
										if (w <= 0)
											break;
										d += (size_t)w;
									}
							});
						for (auto &x : th)
							x.join();
						if (mode == 1)
							munmap(m, piece);
					}
					close(fd);
				}
				const double dt = now() - t0;
				printf("%s rep %d  %-22s T=%d  %.3f s  %.2f GB/s\n", dir.c_str(), rep, mode == 0 ? "fwrite" : (mode == 1 ? "fallocate+mmap+memcpy" : "pwrite"), T, dt,
				       N / dt / 1e9);
				unlink(path.c_str());
			}
		}
	}
	return 0;
}
