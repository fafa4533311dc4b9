// Status/error text, device selection, small host utilities.
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

// debugging aid (PGX_TRACE=1): synchronise and report after a stage, so a device fault names its kernel
void trace_point(const char *what)
{
	static const bool on = getenv("PGX_TRACE") != nullptr;
	if (!on)
		return;
	const hipError_t e = hipDeviceSynchronize();
	fprintf(stderr, "[pgx trace] %s: %s\n", what, hipGetErrorString(e));
	fflush(stderr);
}

} // namespace pgx

extern "C" {

const char *pgx_last_error(void) { return pgx::get_error(); }
const char *pgx_version(void) { return "pangea_hip 0.1 (gfx950; pgx-blastn v1)"; }

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
