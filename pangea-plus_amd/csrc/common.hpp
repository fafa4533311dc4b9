// Shared plumbing of libpangea_hip: status/error text, HIP call checking, device buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pangea_hip.h"

namespace pgx {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
const char *get_error();
int fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Every compute entry point goes through this first: the product has no CPU path.
int require_device();

#define PGX_HIP(call)                                                                              \
	do {                                                                                       \
		hipError_t e_ = (call);                                                            \
		if (e_ != hipSuccess)                                                              \
			return ::pgx::fail(PGX_E_NODEVICE, "%s failed: %s (%s:%d)", #call,         \
					   hipGetErrorString(e_), __FILE__, __LINE__);             \
	} while (0)

#define PGX_TRY(expr)                                                                              \
	do {                                                                                       \
		int rc_ = (expr);                                                                  \
		if (rc_ < 0)                                                                       \
			return rc_;                                                                \
	} while (0)

// No C++ exception crosses the C ABI: an entry point whose body sizes containers from its input runs inside guard()
template <typename F> int guard(const char *what, F &&body)
{
	try {
		return body();
	} catch (const std::bad_alloc &) {
		return fail(PGX_E_NOMEM, "%s: out of host memory", what);
	} catch (const std::length_error &) {
		return fail(PGX_E_FORMAT, "%s: a size in the input is not plausible", what);
	} catch (const std::exception &e) {
		return fail(PGX_E_FORMAT, "%s: %s", what, e.what());
	}
}

// Large copies between host memory and HBM go through the process's ring of pinned buffers (common.hip): handing the
// runtime a large PAGEABLE range (a file mapping, a vector, a numpy array) makes it pin those pages for the copy, and the
// driver then stops every queue of the process when the range is unmapped later -- measured as 20-30 ms stalls of the
// search that followed a file import.  `bytes` below kStagedCopyMin take the plain hipMemcpy.
constexpr size_t kStagedCopyMin = 1u << 20;
int staged_upload(void *dst_device, const void *src_host, size_t bytes);
int staged_download(void *dst_host, const void *src_device, size_t bytes);

// Owning device allocation. `front_pad` elements are kept in front of data() so that kernels
// may read a little before the first element (diagonals that start left of the database).
template <typename T> struct DevBuf {
	T *base = nullptr;
	size_t n = 0, pad = 0;
	DevBuf() = default;
	DevBuf(const DevBuf &) = delete;
	DevBuf &operator=(const DevBuf &) = delete;
	~DevBuf() { release(); }
	void release()
	{
		if (base)
			(void)hipFree(base);
		base = nullptr;
		n = pad = 0;
	}
	int alloc(size_t count, size_t front_pad = 0, size_t back_pad = 0, bool zero = false)
	{
		release();
		size_t total = count + front_pad + back_pad;
		if (total == 0)
			total = 1;
		hipError_t e = hipMalloc((void **)&base, total * sizeof(T));
		if (e != hipSuccess) {
			base = nullptr;
			return fail(PGX_E_NOMEM, "hipMalloc of %zu bytes failed: %s", total * sizeof(T),
				    hipGetErrorString(e));
		}
		n = count;
		pad = front_pad;
		if (zero || front_pad || back_pad) {
			e = hipMemset(base, 0, total * sizeof(T));
			if (e != hipSuccess)
				return fail(PGX_E_NODEVICE, "hipMemset failed: %s", hipGetErrorString(e));
		}
		return 0;
	}
	// grow-only variant for workspaces that live across calls (no hipMalloc in the steady state)
	int ensure(size_t count)
	{
		if (base && count <= n && pad == 0)
			return 0;
		return alloc(count + count / 8);
	}
	T *data() const { return base ? base + pad : nullptr; }
	size_t bytes() const { return n * sizeof(T); }
	int upload(const T *host, size_t count)
	{
		if (count > n)
			return fail(PGX_E_ARG, "upload larger than buffer");
		if (count == 0)
			return 0;
		if (count * sizeof(T) >= kStagedCopyMin)
			return staged_upload(data(), host, count * sizeof(T));
		hipError_t e = hipMemcpy(data(), host, count * sizeof(T), hipMemcpyHostToDevice);
		if (e != hipSuccess)
			return fail(PGX_E_NODEVICE, "hipMemcpy H2D failed: %s", hipGetErrorString(e));
		return 0;
	}
	int download(T *host, size_t count, size_t first = 0) const
	{
		if (count == 0)
			return 0;
		if (count * sizeof(T) >= kStagedCopyMin)
			return staged_download(host, data() + first, count * sizeof(T));
		hipError_t e = hipMemcpy(host, data() + first, count * sizeof(T), hipMemcpyDeviceToHost);
		if (e != hipSuccess)
			return fail(PGX_E_NODEVICE, "hipMemcpy D2H failed: %s", hipGetErrorString(e));
		return 0;
	}
};

// growable text buffer for the formatters
struct Text {
	std::string s;
	void printf(const char *fmt, ...) __attribute__((format(printf, 2, 3)));
	char *release_malloc(size_t *len) const;
};

std::string read_text_file(const char *path, bool *ok);
// The bytes of a text: an owned string, or a regular file mapped where the page cache holds it (no copy, no zero-filled
// buffer first: read_text_file spent 70 ms on a 330 MB read file, the populated mapping takes 20).
struct TextBlob {
	const char *p = nullptr;
	size_t n = 0;
	void *map = nullptr;
	std::string own;
	TextBlob() = default;
	explicit TextBlob(std::string &&s) : own(std::move(s)) { p = own.data(); n = own.size(); }
	TextBlob(const TextBlob &) = delete;
	TextBlob &operator=(const TextBlob &) = delete;
	~TextBlob();
	const char *data() const { return p; }
	size_t size() const { return n; }
	static std::shared_ptr<const TextBlob> from_file(const char *path, bool *ok);
};
int write_text_file(const char *path, const std::string &s);

void trace_point(const char *what);

// Perl semantics shared by the script drop-ins (megaclust.hip): numification of a string, truth of an option value
double perl_num(const char *s, size_t n);
bool perl_true(const char *v);

} // namespace pgx
