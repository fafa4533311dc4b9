// Status/error text, device selection, small host utilities, and the ring of pinned buffers that large copies between host
// memory and HBM go through (staged_upload / staged_download).
#include <chrono>
#include <cstring>
#include <vector>
#include <thread>
#include <mutex>
#include <atomic>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>

#include "common.hpp"
#include <sys/mman.h>

namespace pgx {

static thread_local char g_err[1024] = "";
static int g_device = -1;

void set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
}

const char *get_error() { return g_err; }

int fail(int status, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return status;
}

// The device of this process: what pgx_init() chose, else PGX_DEVICE, else device 0.  hipSetDevice() is a per-thread
// setting, so every host thread that enters the library is bound to the process's device on its first call (a thread of
// a process that chose device 3 used to land on device 0).
static thread_local int t_device = -1;

int require_device()
{
	if (g_device >= 0 && t_device == g_device)
		return 0;
	int want = g_device;
	if (want < 0) {
		int n = 0;
		hipError_t e = hipGetDeviceCount(&n);
		if (e != hipSuccess || n <= 0)
			return fail(PGX_E_NODEVICE,
				    "no HIP device: libpangea_hip has no CPU path (hipGetDeviceCount: %s, %d devices)",
				    hipGetErrorString(e), n);
		want = 0;
		if (const char *env = getenv("PGX_DEVICE")) {
			want = atoi(env);
			if (want < 0 || want >= n)
				return fail(PGX_E_ARG, "PGX_DEVICE=%s is outside 0..%d", env, n - 1);
		}
	}
	const hipError_t e = hipSetDevice(want);
	if (e != hipSuccess)
		return fail(PGX_E_NODEVICE, "hipSetDevice(%d) failed: %s", want, hipGetErrorString(e));
	g_device = want;
	t_device = want;
	return 0;
}

void Text::printf(const char *fmt, ...)
{
	char small[512];
	va_list ap, ap2;
	va_start(ap, fmt);
	va_copy(ap2, ap);
	int need = vsnprintf(small, sizeof small, fmt, ap);
	va_end(ap);
	if (need < 0) {
		va_end(ap2);
		return;
	}
	if ((size_t)need < sizeof small) {
		s.append(small, (size_t)need);
	} else {
		size_t old = s.size();
		s.resize(old + (size_t)need + 1);
		vsnprintf(&s[old], (size_t)need + 1, fmt, ap2);
		s.resize(old + (size_t)need);
	}
	va_end(ap2);
}

char *Text::release_malloc(size_t *len) const
{
	char *p = (char *)malloc(s.size() + 1);
	if (!p)
		return nullptr;
	memcpy(p, s.data(), s.size());
	p[s.size()] = '\0';
	if (len)
		*len = s.size();
	return p;
}

TextBlob::~TextBlob()
{
	if (map)
		munmap(map, n);
}

std::shared_ptr<const TextBlob> TextBlob::from_file(const char *path, bool *ok)
{
	const int fd = open(path, O_RDONLY);
	if (fd >= 0) {
		struct stat st;
		if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
			void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
			if (m != MAP_FAILED) {
				close(fd);
				auto b = std::make_shared<TextBlob>();
				b->map = m;
				b->p = (const char *)m;
				b->n = (size_t)st.st_size;
				*ok = true;
				return b;
			}
		}
		close(fd);
	}
	return std::make_shared<const TextBlob>(read_text_file(path, ok)); // (pipes, empty files, what cannot be mapped)
}

std::string read_text_file(const char *path, bool *ok)
{
	// one read() per gigabyte into a buffer sized from fstat (appending 64 KB pieces costs three copies of a
	// multi-hundred-megabyte read file); pipes and other size-less files fall back to the piecewise loop
	std::string out;
	const int fd = open(path, O_RDONLY);
	if (fd < 0) {
		*ok = false;
		return out;
	}
	struct stat st;
	if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
		out.resize((size_t)st.st_size);
		size_t got = 0;
		while (got < out.size()) {
			const ssize_t k = read(fd, &out[got], std::min<size_t>(out.size() - got, (size_t)1 << 30));
			if (k <= 0)
				break;
			got += (size_t)k;
		}
		out.resize(got);
	}
	char buf[1 << 16];
	ssize_t k;
	while ((k = read(fd, buf, sizeof buf)) > 0)
		out.append(buf, (size_t)k);
	close(fd);
	*ok = true;
	return out;
}

int write_text_file(const char *path, const std::string &s)
{
	FILE *f = fopen(path, "wb");
	if (!f)
		return fail(PGX_E_IO, "cannot open %s for writing", path);
	if (!s.empty() && fwrite(s.data(), 1, s.size(), f) != s.size()) {
		fclose(f);
		return fail(PGX_E_IO, "short write to %s", path);
	}
	if (fclose(f))
		return fail(PGX_E_IO, "close of %s failed", path);
	return 0;
}


// ------------------------------------------------------------------------------------------ staged copies
// kStageWorkers host threads, each with its own stream and two pinned buffers: a worker copies stripe k of the range into
// one buffer and starts its DMA while it fills the other.  The ring is made at the first large copy and lives as long as the
// process (pinning 64 MB costs milliseconds: not per call); one large copy at a time per process.
namespace {
constexpr int kStageWorkers = 4;
constexpr size_t kStageStripe = 8u << 20;
struct StageRing {
	std::mutex mu;
	int device = -1;
	char *pin[kStageWorkers][2] = {};
	hipStream_t stream[kStageWorkers] = {};
	hipEvent_t done[kStageWorkers][2] = {};
	bool ready = false;
	int init(int dev)
	{
		if (ready && device == dev)
			return 0;
		if (ready) // (the process moved to another device: streams and events belong to the old one)
			for (int w = 0; w < kStageWorkers; w++) {
				(void)hipStreamDestroy(stream[w]);
				for (int b = 0; b < 2; b++)
					(void)hipEventDestroy(done[w][b]);
			}
		for (int w = 0; w < kStageWorkers; w++) {
			for (int b = 0; b < 2; b++) {
				if (!pin[w][b])
					PGX_HIP(hipHostMalloc((void **)&pin[w][b], kStageStripe, hipHostMallocDefault));
				PGX_HIP(hipEventCreateWithFlags(&done[w][b], hipEventDisableTiming));
			}
			PGX_HIP(hipStreamCreateWithFlags(&stream[w], hipStreamNonBlocking));
		}
		device = dev;
		ready = true;
		return 0;
	}
};
StageRing g_ring;

int staged_copy(char *device_p, char *host_p, size_t bytes, bool to_device)
{
	PGX_TRY(require_device());
	// (what is in flight on the caller's streams and touches the device range must be over: hipMemcpy waited for it, too)
	PGX_HIP(hipDeviceSynchronize());
	std::lock_guard<std::mutex> lock(g_ring.mu);
	int dev = 0;
	PGX_HIP(hipGetDevice(&dev));
	PGX_TRY(g_ring.init(dev));
	const size_t n_stripes = (bytes + kStageStripe - 1) / kStageStripe;
	const int workers = (int)std::min<size_t>(kStageWorkers, n_stripes);
	std::atomic<int> failed{ 0 };
	auto work = [&](int w) {
		if (hipSetDevice(dev) != hipSuccess) {
			failed = 1;
			return;
		}
		int round = 0;
		for (size_t k = (size_t)w; k < n_stripes && !failed; k += (size_t)workers, round++) {
			const int b = round & 1;
			const size_t o = k * kStageStripe, len = std::min(kStageStripe, bytes - o);
			char *pin = g_ring.pin[w][b];
			if (round >= 2 && hipEventSynchronize(g_ring.done[w][b]) != hipSuccess) // the buffer's earlier DMA
				failed = 1;
			if (to_device) {
				memcpy(pin, host_p + o, len);
				if (hipMemcpyAsync(device_p + o, pin, len, hipMemcpyHostToDevice, g_ring.stream[w]) != hipSuccess)
					failed = 1;
				if (hipEventRecord(g_ring.done[w][b], g_ring.stream[w]) != hipSuccess)
					failed = 1;
			} else {
				// (the copy out of the pinned buffer needs its DMA finished: stripe k's DMA runs while stripe k - workers' bytes
				// are copied out)
				if (hipMemcpyAsync(pin, device_p + o, len, hipMemcpyDeviceToHost, g_ring.stream[w]) != hipSuccess)
					failed = 1;
				if (hipEventRecord(g_ring.done[w][b], g_ring.stream[w]) != hipSuccess)
					failed = 1;
				if (round >= 1) {
					const size_t ko = (k - (size_t)workers) * kStageStripe;
					if (hipEventSynchronize(g_ring.done[w][b ^ 1]) != hipSuccess)
						failed = 1;
					memcpy(host_p + ko, g_ring.pin[w][b ^ 1], std::min(kStageStripe, bytes - ko));
				}
			}
		}
		if (!to_device && round >= 1 && !failed) { // the worker's last stripe
			const size_t k = (size_t)w + (size_t)(round - 1) * (size_t)workers, o = k * kStageStripe;
			if (hipEventSynchronize(g_ring.done[w][(round - 1) & 1]) != hipSuccess)
				failed = 1;
			memcpy(host_p + o, g_ring.pin[w][(round - 1) & 1], std::min(kStageStripe, bytes - o));
		}
		if (hipStreamSynchronize(g_ring.stream[w]) != hipSuccess)
			failed = 1;
	};
	std::vector<std::thread> th;
	try {
		for (int w = 1; w < workers; w++)
			th.emplace_back(work, w);
	} catch (...) {
		failed = 1; // (no thread to be had)
	}
	if (!failed)
		work(0);
	for (auto &t : th)
		t.join();
	if (failed)
		return fail(PGX_E_NODEVICE, "staged copy of %zu bytes failed: %s", bytes, hipGetErrorString(hipGetLastError()));
	return 0;
}
} // namespace

int staged_upload(void *dst_device, const void *src_host, size_t bytes)
{
	return staged_copy((char *)dst_device, const_cast<char *>((const char *)src_host), bytes, true);
}
int staged_download(void *dst_host, const void *src_device, size_t bytes)
{
	return staged_copy(const_cast<char *>((const char *)src_device), (char *)dst_host, bytes, false);
}

// debugging aid (PGX_TRACE=1): synchronise and report after a stage, so a device fault names its kernel
void trace_point(const char *what)
{
	static const bool on = getenv("PGX_TRACE") != nullptr;
	if (!on)
		return;
	static const auto t0 = std::chrono::steady_clock::now();
	const hipError_t e = hipDeviceSynchronize();
	fprintf(stderr, "[pgx trace] %9.2f ms  %s: %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what,
		hipGetErrorString(e));
	fflush(stderr);
}

} // namespace pgx

extern "C" {

const char *pgx_last_error(void) { return pgx::get_error(); }
const char *pgx_version(void) { return "pangea_hip 0.4 (gfx950; pgx-blastn v2)"; }

int pgx_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

int pgx_init(int device)
{
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0)
		return pgx::fail(PGX_E_NODEVICE, "no HIP device: libpangea_hip has no CPU path (%s)",
				 hipGetErrorString(e));
	if (device < 0 || device >= n)
		return pgx::fail(PGX_E_ARG, "device %d out of range (0..%d)", device, n - 1);
	e = hipSetDevice(device);
	if (e != hipSuccess)
		return pgx::fail(PGX_E_NODEVICE, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
	pgx::g_device = device;
	pgx::t_device = device;
	return 0;
}

int pgx_current_device(void) { return pgx::g_device; }

void pgx_free(void *p) { free(p); }
}
