// Trim/trim2.4.pl (== trim2.3.pl on these inputs) for FASTQ and QSEQ reads — SURVEY 8(f) row 3, the step before the
// hot path: raw Illumina reads -> quality-trimmed, mate-joined FASTA that blastn / soap take.
//
// The whole read file is resident in HBM; everything that touches a base or a quality runs on the device:
//   k_nl_count / k_nl_write   line index of the text (one start offset per line), 16 KB tiles, two passes
//   k_fq_measure / k_qs_measure   one lane per output record: the script's running-sum quality rule
//                                 (trim2.4.pl:543-563, :272-287), the kept span of each mate, the record's output size
//   k_fq_emit / k_qs_emit     16 lanes per record write its FASTA text (header rewrite, white-space removal,
//                             '.' -> N, the $GAPSIZE N's between mates) at the offset an exclusive scan gave it
// The host keeps what the script does before it reads a record: getopts, the open checks, format detection on
// the first line, and the messages.  What the script's Perl actually computes (several statements have no
// effect) is written out in oracle/o_trim.c's header, which the parity tests pin on the reference's own output.
//
// Not covered: FASTA-format input (parse_fasta / join_fasta, trim2.4.pl:301-465) and a negative -t.
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

#include "engine.hpp"
#include "textops.hpp"

namespace pgx {

constexpr int kQualityCutoff = 20; // trim2.4.pl:34; -qc never reaches it (getopts spec :51)
constexpr uint32_t kLengthCutoff = 70; // trim2.4.pl:33; -lc never reaches it
constexpr uint32_t kZeroRead = 0xFFFFFFFFu; // the mate became the number 0 (:569, :293)

constexpr int kNlThreads = 256, kNlIter = 4;
constexpr uint64_t kNlTile = (uint64_t)kNlThreads * 16 * kNlIter; // 16 KB of text per block

// one flag bit (0x80) per byte of `w` equal to '\n', exact
__device__ __forceinline__ uint32_t nl_flags(uint32_t w)
{
	const uint32_t x = w ^ 0x0A0A0A0Au;
	const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
	return ~(t | x | 0x7F7F7F7Fu);
}

// 16 bytes at `off`: bit k of the result = byte k is a newline
__device__ __forceinline__ uint32_t nl_mask16(const uint8_t *__restrict__ text, uint64_t n, uint64_t off)
{
	uint32_t m = 0;
	if (off + 16 <= n) {
		const uint4 v = *reinterpret_cast<const uint4 *>(text + off);
		const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const uint32_t f = nl_flags(w[q]);
			m |= (((f >> 7) & 1u) | ((f >> 14) & 2u) | ((f >> 21) & 4u) | ((f >> 28) & 8u)) << (4 * q);
		}
	} else {
		for (int k = 0; k < 16 && off + k < n; k++)
			m |= (uint32_t)(text[off + k] == '\n') << k;
	}
	return m;
}

__global__ __launch_bounds__(kNlThreads) void k_nl_count(const uint8_t *__restrict__ text, uint64_t n, uint32_t *__restrict__ tile_cnt,
							  uint64_t n_tiles)
{
	__shared__ uint32_t s_w[kNlThreads / 64];
	const uint64_t tile = blockIdx.x;
	uint32_t c = 0;
	for (int it = 0; it < kNlIter; it++) {
		const uint64_t off = tile * kNlTile + ((uint64_t)it * kNlThreads + threadIdx.x) * 16;
		if (off < n)
			c += __popc(nl_mask16(text, n, off));
	}
	for (int d = 32; d; d >>= 1)
		c += __shfl_down(c, d, 64);
	if ((threadIdx.x & 63) == 0)
		s_w[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) {
		tile_cnt[tile] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
		if (tile == 0)
			tile_cnt[n_tiles] = 0;
	}
}

// start[1 + k] = offset of the byte after the k-th newline (start[0] = 0 is written by the host)
__global__ __launch_bounds__(kNlThreads) void k_nl_write(const uint8_t *__restrict__ text, uint64_t n, const uint64_t *__restrict__ tile_off,
							  uint64_t *__restrict__ start)
{
	__shared__ uint32_t s_w[kNlThreads / 64];
	const uint64_t tile = blockIdx.x;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint64_t carry = tile_off[tile];
	for (int it = 0; it < kNlIter; it++) {
		const uint64_t off = tile * kNlTile + ((uint64_t)it * kNlThreads + threadIdx.x) * 16;
		uint32_t m = off < n ? nl_mask16(text, n, off) : 0u;
		const uint32_t c = __popc(m);
		uint32_t incl = c;
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t o = __shfl_up(incl, d, 64);
			if (lane >= d)
				incl += o;
		}
		if (lane == 63)
			s_w[wave] = incl;
		__syncthreads();
		uint32_t before = 0, all = 0;
		for (int q = 0; q < kNlThreads / 64; q++) {
			if (q < wave)
				before += s_w[q];
			all += s_w[q];
		}
		uint64_t k = carry + before + (incl - c);
		while (m) {
			const int b = __ffs(m) - 1;
			m &= m - 1;
			start[1 + k++] = off + (uint64_t)b + 1;
		}
		carry += all;
		__syncthreads();
	}
}

struct TextView {
	const uint8_t *text;
	const uint64_t *start; // n_lines + 1 offsets
	uint64_t n_lines;
};

struct Span {
	uint64_t off;
	uint64_t n;
};

// line `i` with its newline, or an empty span past the end of the file (an undefined <FH> value)
__device__ __forceinline__ Span line_of(const TextView &t, uint64_t i)
{
	Span s = { 0, 0 };
	if (i < t.n_lines) {
		s.off = t.start[i];
		s.n = t.start[i + 1] - s.off;
	}
	return s;
}

__device__ __forceinline__ bool p_space_dev(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

// trim2.4.pl:535-563 / :264-287: first index at which the clamped running sum reaches its overall maximum
__device__ uint64_t quality_end(const uint8_t *__restrict__ text, uint64_t off, uint64_t n, int offset)
{
	long long max = 0, sum = 0;
	uint64_t end = 0;
	for_bytes(text, off, n, [&](uint8_t c, uint64_t a) {
		sum += (long long)c - offset - kQualityCutoff;
		if (sum > max) {
			max = sum;
			end = a;
		}
		if (sum < 0)
			sum = 0;
	});
	return end;
}

// ------------------------------------------------------------------------------------------------- FASTQ
struct FqRec {
	uint32_t keep1, keep2; // bytes of substr($sequence, 0, $end), or kZeroRead
};

// one lane per output record: 4 lines, or 8 with -b (mates interleaved in the -a file, trim2.4.pl:492-495)
__global__ __launch_bounds__(256) void k_fq_measure(TextView t, uint64_t n_rec, int paired, uint64_t gap, FqRec *__restrict__ rec,
						    uint64_t *__restrict__ out_len, uint32_t *__restrict__ too_long)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_rec)
		return;
	if (r == n_rec) {
		out_len[r] = 0;
		return;
	}
	const uint64_t l0 = r * (paired ? 8 : 4);
	Span hdr = line_of(t, l0);
	if (hdr.n && t.text[hdr.off + hdr.n - 1] == '\n')
		hdr.n--; // chomp($header1)
	uint64_t len = 1 + 4 + 1; // '>' ":AB\n" "\n"
	for_bytes(t.text, hdr.off, hdr.n, [&](uint8_t c, uint64_t) { len += c != '@'; }); // s/@//g
	FqRec f = { kZeroRead, kZeroRead };
	bool bad = false;
	for (int mate = 0; mate < (paired ? 2 : 1); mate++) {
		const Span seq = line_of(t, l0 + 4 * mate + 1), qual = line_of(t, l0 + 4 * mate + 3);
		bad |= seq.n >= 0x7FFFFFFFull;
		const uint64_t end = quality_end(t.text, qual.off, qual.n, 33);
		const uint64_t kept = end < seq.n ? end : seq.n;
		if (kept >= kLengthCutoff && !bad) {
			if (mate == 0) {
				f.keep1 = (uint32_t)kept;
				// $fastq1 =~ s/\s//g (also the tab of :571)
				for_bytes(t.text, seq.off, kept, [&](uint8_t c, uint64_t) { len += !p_space_dev(c); });
			} else {
				f.keep2 = (uint32_t)kept;
				len += kept + 1; // the second mate keeps its tab (:571, :508)
			}
		} else {
			len += 1; // "0"
		}
	}
	if (paired)
		len += gap;
	if (bad)
		atomicOr(too_long, 1u);
	rec[r] = f;
	out_len[r] = len;
}

// the FASTA text of record `r`, written by one lane group (trim2.4.pl:487-515)
__device__ void fq_emit_record(const TextView &t, uint64_t r, int paired, uint64_t gap, const FqRec f, uint64_t pos, char *__restrict__ out)
{
	const uint64_t l0 = r * (paired ? 8 : 4);
	Span hdr = line_of(t, l0);
	if (hdr.n && t.text[hdr.off + hdr.n - 1] == '\n')
		hdr.n--;
	auto same = [](uint8_t c) { return c; };
	w_lit(out, pos, ">", 1);
	w_copy(out, pos, t.text + hdr.off, hdr.n, [](uint8_t c) { return c != '@'; }, same);
	w_lit(out, pos, ":AB\n", 4);
	if (f.keep1 != kZeroRead)
		w_copy(out, pos, t.text + line_of(t, l0 + 1).off, f.keep1, [](uint8_t c) { return !p_space_dev(c); }, same);
	else
		w_lit(out, pos, "0", 1);
	if (paired) {
		w_fill(out, pos, 'N', gap);
		if (f.keep2 != kZeroRead) {
			w_copy(out, pos, t.text + line_of(t, l0 + 5).off, f.keep2, [](uint8_t) { return true; }, same);
			w_lit(out, pos, "\t", 1);
		} else {
			w_lit(out, pos, "0", 1);
		}
	}
	w_lit(out, pos, "\n", 1);
}

// a bounded grid of lane groups strides over the records
__global__ __launch_bounds__(256) void k_fq_emit(TextView t, uint64_t n_rec, int paired, uint64_t gap, const FqRec *__restrict__ rec,
						 const uint64_t *__restrict__ out_off, char *__restrict__ out)
{
	const uint64_t groups = (uint64_t)gridDim.x * (blockDim.x / kGroup);
	for (uint64_t r = (uint64_t)blockIdx.x * (blockDim.x / kGroup) + threadIdx.x / kGroup; r < n_rec; r += groups)
		fq_emit_record(t, r, paired, gap, rec[r], out_off[r], out);
}

// ------------------------------------------------------------------------------------------------- QSEQ
struct QsRec {
	uint64_t s1, s2;     // offsets of the kept bases of the two mates in their texts
	uint32_t n1, n2;     // kept bases
	uint32_t hdr_len;    // bytes of line A up to the end of field 7 (tabs become ':')
	uint32_t pad_dots;   // bits 0-3: ':' to append when the line has fewer than 8 fields; bit 4: '.' -> N applies
};

struct QsFields {
	Span f7, f8, f9;
	uint32_t hdr_len, pad;
};

// split(/\t/, chomp(line)): fields 7, 8, 9 and the join(':', @line[0..7]) prefix (trim2.4.pl:184-185, :218)
__device__ QsFields qs_split(const uint8_t *__restrict__ text, Span line)
{
	if (line.n && text[line.off + line.n - 1] == '\n')
		line.n--;
	QsFields q = { { 0, 0 }, { 0, 0 }, { 0, 0 }, 0, 0 };
	uint64_t st = 0;
	int k = 0;
	bool have_hdr = false;
	auto close_field = [&](uint64_t i) {
		const Span f = { line.off + st, i - st };
		if (k == 7) {
			q.f7 = f;
			q.hdr_len = (uint32_t)i;
			have_hdr = true;
		} else if (k == 8) {
			q.f8 = f;
		} else if (k == 9) {
			q.f9 = f;
		}
		k++;
		st = i + 1;
	};
	for_bytes(text, line.off, line.n, [&](uint8_t c, uint64_t i) {
		if (c == '\t' && k <= 9)
			close_field(i);
	});
	if (k <= 9)
		close_field(line.n); // the last field ends with the line
	if (!have_hdr) { // fewer than 8 fields: the whole line, then one ':' per missing field
		q.hdr_len = (uint32_t)line.n;
		q.pad = (uint32_t)(8 - k);
	}
	return q;
}

// trim_qseq (trim2.4.pl:253-298); t1 = int($TRUNCATE), t2 = int($TRUNCATE - 1)
__device__ bool qs_trim(const uint8_t *__restrict__ text, Span seq, Span qual, long long t1, long long t2, Span *kept)
{
	uint64_t cut = (uint64_t)t1 < seq.n ? (uint64_t)t1 : seq.n; // substr($seq, $TRUNCATE): past the end = empty
	seq.off += cut;
	seq.n -= cut;
	cut = (uint64_t)t1 < qual.n ? (uint64_t)t1 : qual.n;
	qual.off += cut;
	qual.n -= cut;
	uint64_t drop; // substr($seq, 0, $TRUNCATE-1) = '': a negative length leaves that many bases at the end
	if (t2 >= 0)
		drop = (uint64_t)t2 < seq.n ? (uint64_t)t2 : seq.n;
	else
		drop = (uint64_t)(-t2) < seq.n ? seq.n - (uint64_t)(-t2) : 0;
	seq.off += drop;
	seq.n -= drop;
	const uint64_t end = quality_end(text, qual.off, qual.n, 64);
	kept->off = seq.off;
	kept->n = end < seq.n ? end : seq.n;
	return kept->n >= kLengthCutoff;
}

__device__ __forceinline__ bool span_is(const uint8_t *__restrict__ text, Span s, char c) { return s.n == 1 && text[s.off] == (uint8_t)c; }

// one lane per line of file A and the line of the same number of file B (trim2.4.pl:180-246)
__global__ __launch_bounds__(256) void k_qs_measure(TextView a, TextView b, uint64_t n_rec, uint64_t gap, long long t1, long long t2,
						    QsRec *__restrict__ rec, uint64_t *__restrict__ out_len, uint32_t *__restrict__ too_long)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_rec)
		return;
	if (r == n_rec) {
		out_len[r] = 0;
		return;
	}
	const Span la = line_of(a, r), lb = line_of(b, r);
	const QsFields fa = qs_split(a.text, la), fb = qs_split(b.text, lb);
	Span k1 = fa.f8, k2 = fb.f8;
	bool zero1, zero2;
	uint32_t dots = 0;
	if (span_is(a.text, fa.f7, '1')) {
		zero1 = !qs_trim(a.text, fa.f8, fa.f9, t1, t2, &k1);
		zero2 = !qs_trim(b.text, fb.f8, fb.f9, t1, t2, &k2);
		dots = 16;
	} else {
		zero1 = span_is(a.text, k1, '0');
		zero2 = span_is(b.text, k2, '0');
	}
	QsRec q = { k1.off, k2.off, (uint32_t)k1.n, (uint32_t)k2.n, fa.hdr_len, fa.pad | dots };
	uint64_t len = 0;
	if (la.n >= 0x7FFFFFFFull || lb.n >= 0x7FFFFFFFull) {
		atomicOr(too_long, 1u);
	} else if (!zero1 && !zero2) {
		len = 1 + (uint64_t)fa.hdr_len + fa.pad + 4 + k1.n + gap + k2.n + 1;
	}
	rec[r] = q;
	out_len[r] = len;
}

// the FASTA text of pair `r`, written by one lane group (trim2.4.pl:218-245)
__device__ void qs_emit_record(const TextView &a, const TextView &b, uint64_t r, uint64_t gap, const QsRec q, uint64_t pos, char *__restrict__ out)
{
	const bool dots = (q.pad_dots & 16u) != 0;
	auto all = [](uint8_t) { return true; };
	auto base = [dots](uint8_t c) { return dots && c == '.' ? (uint8_t)'N' : c; };
	w_lit(out, pos, ">", 1);
	w_copy(out, pos, a.text + line_of(a, r).off, q.hdr_len, all, [](uint8_t c) { return c == '\t' ? (uint8_t)':' : c; });
	w_fill(out, pos, ':', q.pad_dots & 15u);
	w_lit(out, pos, ":AB\n", 4);
	w_copy(out, pos, a.text + q.s1, q.n1, all, base);
	w_fill(out, pos, 'N', gap);
	w_copy(out, pos, b.text + q.s2, q.n2, all, base);
	w_lit(out, pos, "\n", 1);
}

__global__ __launch_bounds__(256) void k_qs_emit(TextView a, TextView b, uint64_t n_rec, uint64_t gap, const QsRec *__restrict__ rec,
						 const uint64_t *__restrict__ out_off, char *__restrict__ out)
{
	const uint64_t groups = (uint64_t)gridDim.x * (blockDim.x / kGroup);
	for (uint64_t r = (uint64_t)blockIdx.x * (blockDim.x / kGroup) + threadIdx.x / kGroup; r < n_rec; r += groups)
		if (out_off[r + 1] != out_off[r]) // else a mate did not survive: nothing is written for the pair (:199-210)
			qs_emit_record(a, b, r, gap, rec[r], out_off[r], out);
}

// ------------------------------------------------------------------------------------------------- host side
struct DeviceText {
	DevBuf<uint8_t> text;
	DevBuf<uint64_t> start;
	uint64_t n = 0, n_lines = 0;
	TextView view() const { return TextView{ text.data(), start.data(), n_lines }; }
};

template <typename In, typename Out> static int exclusive_sum(const In *in, Out *out, size_t n)
{
	size_t bytes = 0;
	PGX_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, (Out)0, n, rocprim::plus<Out>()));
	DevBuf<uint8_t> tmp;
	PGX_TRY(tmp.alloc(bytes ? bytes : 1));
	PGX_HIP(rocprim::exclusive_scan(tmp.data(), bytes, in, out, (Out)0, n, rocprim::plus<Out>()));
	return 0;
}

// PGX_TRIM_TIMES=1: wall time of each stage to stderr (synchronising; tools/probe_trim.py reads it)
struct StageClock {
	bool on = getenv("PGX_TRIM_TIMES") != nullptr;
	std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
	void tick(const char *what)
	{
		if (!on)
			return;
		(void)hipDeviceSynchronize();
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[pgx trim] %-10s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
		last = now;
	}
};
static StageClock g_clock;

// the text into HBM and its line index
static int upload_lines(const std::string &s, DeviceText &d)
{
	d.n = s.size();
	PGX_TRY(d.text.alloc(d.n, 0, 16));
	PGX_TRY(d.text.upload((const uint8_t *)s.data(), d.n));
	g_clock.tick("upload");
	const uint64_t n_tiles = (d.n + kNlTile - 1) / kNlTile;
	if (n_tiles >= 0x7FFFFFFFull)
		return fail(PGX_E_LIMIT, "read file of %llu bytes is beyond the line indexer's range", (unsigned long long)d.n);
	uint64_t newlines = 0;
	DevBuf<uint32_t> cnt;
	DevBuf<uint64_t> off;
	PGX_TRY(cnt.alloc(n_tiles + 1, 0, 0, true));
	PGX_TRY(off.alloc(n_tiles + 1, 0, 0, true));
	if (n_tiles) {
		hipLaunchKernelGGL(k_nl_count, dim3((unsigned)n_tiles), dim3(kNlThreads), 0, 0, d.text.data(), d.n, cnt.data(), n_tiles);
		PGX_HIP(hipGetLastError());
		PGX_TRY((exclusive_sum<uint32_t, uint64_t>(cnt.data(), off.data(), (size_t)n_tiles + 1)));
		PGX_TRY(off.download(&newlines, 1, n_tiles));
	}
	const bool open_tail = d.n > 0 && s[d.n - 1] != '\n';
	d.n_lines = newlines + (open_tail ? 1 : 0);
	PGX_TRY(d.start.alloc(newlines + 2, 0, 0, true)); // start[0] = 0
	if (n_tiles) {
		hipLaunchKernelGGL(k_nl_write, dim3((unsigned)n_tiles), dim3(kNlThreads), 0, 0, d.text.data(), d.n, off.data(), d.start.data());
		PGX_HIP(hipGetLastError());
	}
	if (open_tail)
		PGX_HIP(hipMemcpy(d.start.data() + d.n_lines, &d.n, sizeof d.n, hipMemcpyHostToDevice));
	trace_point("trim line index");
	g_clock.tick("line index");
	return 0;
}

// blocks of four wavefronts for the record writers: enough to fill 256 CUs several times over
static unsigned emit_grid(uint64_t n_rec) { return (unsigned)std::min<uint64_t>((n_rec + 256 / kGroup - 1) / (256 / kGroup), 256u * 32u); }

static int take_output(const DevBuf<char> &out, uint64_t total, char **text, size_t *len)
{
	char *p = (char *)malloc(total + 1);
	if (!p)
		return fail(PGX_E_NOMEM, "malloc of %llu bytes for the trimmed FASTA failed", (unsigned long long)total);
	if (total)
		PGX_HIP(hipMemcpy(p, out.data(), total, hipMemcpyDeviceToHost));
	p[total] = 0;
	*text = p;
	*len = total;
	return 0;
}

static int trim_fastq_device(const std::string &a, bool paired, uint64_t gap, char **text, size_t *len)
{
	DeviceText da;
	PGX_TRY(upload_lines(a, da));
	const uint64_t per = paired ? 8 : 4, n_rec = (da.n_lines + per - 1) / per;
	if (n_rec >= 0x7FFFFFFFull)
		return fail(PGX_E_LIMIT, "%llu FASTQ records in one call", (unsigned long long)n_rec);
	DevBuf<FqRec> rec;
	DevBuf<uint64_t> out_len;
	DevBuf<uint32_t> flag;
	PGX_TRY(rec.alloc(n_rec));
	PGX_TRY(out_len.alloc(n_rec + 1));
	PGX_TRY(flag.alloc(1, 0, 0, true));
	hipLaunchKernelGGL(k_fq_measure, dim3((unsigned)((n_rec + 1 + 255) / 256)), dim3(256), 0, 0, da.view(), n_rec, paired ? 1 : 0, gap,
			   rec.data(), out_len.data(), flag.data());
	PGX_HIP(hipGetLastError());
	DevBuf<uint64_t> out_off;
	PGX_TRY(out_off.alloc(n_rec + 1));
	PGX_TRY((exclusive_sum<uint64_t, uint64_t>(out_len.data(), out_off.data(), (size_t)n_rec + 1)));
	uint64_t total = 0;
	uint32_t bad = 0;
	PGX_TRY(out_off.download(&total, 1, n_rec));
	PGX_TRY(flag.download(&bad, 1));
	g_clock.tick("measure");
	if (bad)
		return fail(PGX_E_LIMIT, "a FASTQ line of 2 GiB or more");
	DevBuf<char> out;
	PGX_TRY(out.alloc(total));
	if (n_rec) {
		hipLaunchKernelGGL(k_fq_emit, dim3(emit_grid(n_rec)), dim3(256), 0, 0, da.view(), n_rec, paired ? 1 : 0, gap, rec.data(),
				   out_off.data(), out.data());
		PGX_HIP(hipGetLastError());
	}
	trace_point("trim fastq");
	g_clock.tick("emit");
	const int rc = take_output(out, total, text, len);
	g_clock.tick("download");
	return rc;
}

static int trim_qseq_device(const std::string &a, const std::string &b, uint64_t gap, long long t1, long long t2, char **text, size_t *len)
{
	DeviceText da, db;
	PGX_TRY(upload_lines(a, da));
	PGX_TRY(upload_lines(b, db));
	const uint64_t n_rec = da.n_lines;
	if (n_rec >= 0x7FFFFFFFull)
		return fail(PGX_E_LIMIT, "%llu QSEQ lines in one call", (unsigned long long)n_rec);
	DevBuf<QsRec> rec;
	DevBuf<uint64_t> out_len;
	DevBuf<uint32_t> flag;
	PGX_TRY(rec.alloc(n_rec));
	PGX_TRY(out_len.alloc(n_rec + 1));
	PGX_TRY(flag.alloc(1, 0, 0, true));
	hipLaunchKernelGGL(k_qs_measure, dim3((unsigned)((n_rec + 1 + 255) / 256)), dim3(256), 0, 0, da.view(), db.view(), n_rec, gap, t1, t2,
			   rec.data(), out_len.data(), flag.data());
	PGX_HIP(hipGetLastError());
	DevBuf<uint64_t> out_off;
	PGX_TRY(out_off.alloc(n_rec + 1));
	PGX_TRY((exclusive_sum<uint64_t, uint64_t>(out_len.data(), out_off.data(), (size_t)n_rec + 1)));
	uint64_t total = 0;
	uint32_t bad = 0;
	PGX_TRY(out_off.download(&total, 1, n_rec));
	PGX_TRY(flag.download(&bad, 1));
	g_clock.tick("measure");
	if (bad)
		return fail(PGX_E_LIMIT, "a QSEQ line of 2 GiB or more");
	DevBuf<char> out;
	PGX_TRY(out.alloc(total));
	if (n_rec) {
		hipLaunchKernelGGL(k_qs_emit, dim3(emit_grid(n_rec)), dim3(256), 0, 0, da.view(), db.view(), n_rec, gap, rec.data(),
				   out_off.data(), out.data());
		PGX_HIP(hipGetLastError());
	}
	trace_point("trim qseq");
	g_clock.tick("emit");
	const int rc = take_output(out, total, text, len);
	g_clock.tick("download");
	return rc;
}

static const char *const kTrimUsage = // trim2.4.pl:54-63
	"Usage: perl trim2.pl \n"
	"\t-a raw illumina input file read 1\n"
	"\t-b raw illumina input file read 2 (if any) \n"
	"\t-g size of GAP between paired-ends (if any) \n"
	"\t-t truncate size (if any)\n"
	"\t-q quality file (in case of FASTA input)\n"
	"\t-qc quality cutoff value\n"
	"\t-j use this option for just joining a and b, without triming\n"
	"\t-lc minimum length \n"
	"Supported formats: FASTA, FASTQ and QSEQ.\n";

// field `k` of split("\t", line): trailing empty fields are dropped by Perl, which matters only for `eq` tests on
// an empty field — an empty field never equals "0"/"1"/"2", present or not
static std::string host_field(const std::string &line, int k)
{
	size_t st = 0;
	for (int i = 0;; i++) {
		const size_t tab = line.find('\t', st);
		const size_t en = tab == std::string::npos ? line.size() : tab;
		if (i == k)
			return line.substr(st, en - st);
		if (tab == std::string::npos)
			return std::string();
		st = en + 1;
	}
}

} // namespace pgx

using namespace pgx;

extern "C" int pgx_trim_file(const pgx_trim_opts *o, char **log_text, char **fasta_text, size_t *fasta_len, int *mode)
{
	if (!o || !log_text || !fasta_text || !fasta_len || !mode)
		return fail(PGX_E_ARG, "pgx_trim_file: null argument");
	*log_text = nullptr;
	*fasta_text = nullptr;
	*fasta_len = 0;
	*mode = PGX_TRIM_NONE;
	PGX_TRY(require_device());
	Text log;
	auto finish = [&](int rc) {
		*log_text = log.release_malloc(nullptr);
		return rc;
	};
	if (!perl_true(o->a)) { // :53
		log.s += kTrimUsage;
		return finish(0);
	}
	bool ok = false;
	g_clock.tick("start");
	const std::string a = read_text_file(o->a, &ok);
	if (!ok) {
		log.printf("Error: Unable to open %s.\n", o->a); // :68-71
		return finish(0);
	}
	const bool paired = perl_true(o->b);
	std::string b;
	if (paired) {
		b = read_text_file(o->b, &ok);
		if (!ok) {
			log.printf("Error: Unable to open %s.\n", o->b); // :78-81
			return finish(0);
		}
	}
	uint64_t gap = 189; // :36
	long long t1 = 11, t2 = 10; // :35
	if (perl_true(o->g)) { // :86-88; for ($r = 0; $r < $GAPSIZE; $r++)
		const double g = perl_num(o->g, strlen(o->g));
		gap = g > 0 ? (g > 2147483647.0 ? 2147483647ull : (uint64_t)std::ceil(g)) : 0;
	}
	if (perl_true(o->t)) { // :90-92
		const double t = perl_num(o->t, strlen(o->t));
		if (!(t > -1.0) || !(t < 2147483647.0))
			return finish(fail(PGX_E_ARG, "-t %s: a negative truncate size is not covered", o->t));
		t1 = (long long)t;
		t2 = (long long)(t - 1.0);
	}
	if (!a.empty() && a[0] == '>')
		return finish(fail(PGX_E_FORMAT, "%s is FASTA: only FASTQ and QSEQ input is covered (trim2.4.pl parse_fasta / join_fasta are not)", o->a));
	int rc = 0;
	g_clock.tick("read files");
	if (!a.empty() && a[0] == '@') { // :146-149
		*mode = PGX_TRIM_FASTQ;
		rc = trim_fastq_device(a, paired, gap, fasta_text, fasta_len);
	} else {
		// :152-156: the first line without its first byte
		std::string first;
		if (a.size() > 1) {
			const size_t nl = a.find('\n', 1);
			first = a.substr(1, (nl == std::string::npos ? a.size() : nl) - 1);
		}
		const std::string f7 = host_field(first, 7), f10 = host_field(first, 10);
		if ((f7 == "1" || f7 == "2") && (f10 == "0" || f10 == "1")) {
			log.s += "QSEQ file format found.\n";
			*mode = PGX_TRIM_QSEQ;
			rc = trim_qseq_device(a, b, gap, t1, t2, fasta_text, fasta_len);
		} else {
			log.s += "Error: file format not recognized.\n";
			*mode = PGX_TRIM_UNKNOWN;
			*fasta_text = (char *)calloc(1, 1); // RUNBLAST was opened and stays empty (:115)
		}
	}
	if (rc < 0)
		return finish(rc);
	log.s += "Trimming complete.\n"; // :167
	return finish(0);
}
