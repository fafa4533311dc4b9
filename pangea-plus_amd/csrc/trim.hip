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
// FASTA-format input (trim2.4.pl:117-143, :301-465; the last part of this file): `-q` = parse_fasta (sequence and quality
// files read line by line in step; the host numifies the quality fields, as it does the score columns of megaclust2; the
// running sum, the start/end carried from record to record and the printed text are the device's), `-j -b` = join_fasta.
// Not covered: a negative -t.
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

// line index of a text that lies in HBM: start[k] = offset of line k, start[n_lines] = behind the last line (the byte after
// its newline, or the end of a text that does not end in one).  Also the first stage of the RDP import (rdp_device.hip).
int device_line_index(const uint8_t *d_text, uint64_t n, bool open_tail, DevBuf<uint64_t> &start, uint64_t *n_lines_out)
{
	const uint64_t n_tiles = (n + kNlTile - 1) / kNlTile;
	if (n_tiles >= 0x7FFFFFFFull)
		return fail(PGX_E_LIMIT, "text of %llu bytes is beyond the line indexer's range", (unsigned long long)n);
	uint64_t newlines = 0;
	DevBuf<uint32_t> cnt;
	DevBuf<uint64_t> off;
	PGX_TRY(cnt.alloc(n_tiles + 1, 0, 0, true));
	PGX_TRY(off.alloc(n_tiles + 1, 0, 0, true));
	if (n_tiles) {
		hipLaunchKernelGGL(k_nl_count, dim3((unsigned)n_tiles), dim3(kNlThreads), 0, 0, d_text, n, cnt.data(), n_tiles);
		PGX_HIP(hipGetLastError());
		PGX_TRY((exclusive_sum<uint32_t, uint64_t>(cnt.data(), off.data(), (size_t)n_tiles + 1)));
		PGX_TRY(off.download(&newlines, 1, n_tiles));
	}
	const uint64_t n_lines = newlines + (open_tail ? 1 : 0);
	PGX_TRY(start.alloc(newlines + 2, 0, 0, true)); // start[0] = 0
	if (n_tiles) {
		hipLaunchKernelGGL(k_nl_write, dim3((unsigned)n_tiles), dim3(kNlThreads), 0, 0, d_text, n, off.data(), start.data());
		PGX_HIP(hipGetLastError());
	}
	if (open_tail)
		PGX_HIP(hipMemcpy(start.data() + n_lines, &n, sizeof n, hipMemcpyHostToDevice));
	*n_lines_out = n_lines;
	return 0;
}

// the text into HBM and its line index
static int upload_lines(const std::string &s, DeviceText &d)
{
	d.n = s.size();
	PGX_TRY(d.text.alloc(d.n, 0, 16));
	PGX_TRY(d.text.upload((const uint8_t *)s.data(), d.n));
	g_clock.tick("upload");
	PGX_TRY(device_line_index(d.text.data(), d.n, d.n > 0 && s[d.n - 1] != '\n', d.start, &d.n_lines));
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

// ------------------------------------------------------------------------------------------------- FASTA input
// What the two subs compute, statement by statement, is written out in oracle/o_trim.c's header (the cut-offs of parse_fasta
// are barewords that count as 0; start / end survive from record to record; the last record is never printed; join_fasta
// drops the line that makes eof() true inside a sequence loop).  Both are one lane per record: this is a format converter
// for a few thousand records, not a bandwidth kernel.

// flag[i] = line i holds a '>' (from line `first` on)
__global__ void k_fa_flags(TextView t, uint64_t first, uint32_t *__restrict__ flag)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= t.n_lines)
		return;
	uint32_t f = 0;
	if (i >= first) {
		const Span l = line_of(t, i);
		for (uint64_t k = 0; k < l.n && !f; k++)
			f = t.text[l.off + k] == '>';
	}
	flag[i] = f;
}

// lines[rank[i]] = i for the flagged lines (rank = exclusive scan of the flags)
__global__ void k_fa_scatter(const uint32_t *__restrict__ flag, const uint64_t *__restrict__ rank, uint64_t n_lines, uint64_t *__restrict__ lines)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_lines && flag[i])
		lines[rank[i]] = i;
}

struct FaRec {
	long long start, end; // the record's own `start` / `end` when it raised the sum (upd), later the ones in force when it is printed
	uint64_t n_trim;      // letters @FinalTrim holds
	uint32_t hdr_len;     // index($line, " ") + 1
	uint32_t upd;
};

// parse_fasta, one lane per record r = header line hdr[r] and the lines up to the next header
__global__ void k_pf_measure(TextView t, const uint64_t *__restrict__ hdr, uint64_t n_rec, const uint64_t *__restrict__ q_off, uint64_t n_qlines,
			     const double *__restrict__ q_val, FaRec *__restrict__ rec)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_rec)
		return;
	const uint64_t h = hdr[r], stop = r + 1 < n_rec ? hdr[r + 1] : t.n_lines;
	FaRec o;
	o.start = o.end = 0;
	o.n_trim = 0;
	o.upd = 0;
	const Span hl = line_of(t, h);
	o.hdr_len = 0;
	for (uint64_t k = 0; k < hl.n; k++)
		if (t.text[hl.off + k] == ' ') {
			o.hdr_len = (uint32_t)(k + 1);
			break;
		}
	double max = 0, sum = 0;
	long long first = 0, line_num = 0;
	for (uint64_t i = h + 1; i < stop; i++) {
		const Span l = line_of(t, i);
		o.n_trim += l.n > 1 ? l.n - 1 : 0; // every character of the line but the last
		if (i < n_qlines)
			for (uint64_t a = q_off[i]; a < q_off[i + 1]; a++) {
				sum += q_val[a]; // - "QUALITY_CUTOFF", a bareword that counts as 0 (:387)
				const long long pos = (long long)(a - q_off[i]) + 60 * line_num;
				if (sum > max) {
					max = sum;
					o.end = pos;
					o.start = first;
					o.upd = 1;
				}
				if (sum < 0) {
					sum = 0;
					first = pos;
				}
			}
		line_num++;
	}
	rec[r] = o;
}

// $start / $end are not reset at a header: a record that never raised the sum is printed with the previous record's;
// then the size of what each record prints (the last one prints nothing: its text would come with the next header)
__global__ void k_pf_carry(FaRec *__restrict__ rec, uint64_t n_rec, uint64_t *__restrict__ out_len)
{
	if (blockIdx.x || threadIdx.x)
		return;
	long long start = 0, end = 0;
	for (uint64_t r = 0; r < n_rec; r++) {
		if (rec[r].upd) {
			start = rec[r].start;
			end = rec[r].end;
		}
		rec[r].start = start;
		rec[r].end = end;
		uint64_t len = 0;
		// (end + 1 - start < 0 -- quality lines of more than 60 values can put the end before the start -- is the one case
		// in which the script's $Rejected branch fires, trim2.4.pl:404-408: the record prints nothing, not even its header)
		if (r + 1 < n_rec && end + 1 - start >= 0) {
			const long long last = end < (long long)rec[r].n_trim - 1 ? end : (long long)rec[r].n_trim - 1;
			const uint64_t letters = last >= start ? (uint64_t)(last - start + 1) : 0;
			len = rec[r].hdr_len + 1 + letters + (uint64_t)((end - start + 1) / 60) + 1;
		}
		out_len[r] = len;
	}
	out_len[n_rec] = 0;
}

__global__ void k_pf_emit(TextView t, const uint64_t *__restrict__ hdr, uint64_t n_rec, const FaRec *__restrict__ rec,
			  const uint64_t *__restrict__ out_off, char *__restrict__ out)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r + 1 >= n_rec)
		return;
	if (out_off[r + 1] == out_off[r])
		return; // (a rejected record: nothing is printed)
	const uint64_t h = hdr[r], stop = hdr[r + 1];
	const FaRec o = rec[r];
	char *w = out + out_off[r];
	const Span hl = line_of(t, h);
	for (uint32_t k = 0; k < o.hdr_len; k++)
		*w++ = (char)t.text[hl.off + k];
	*w++ = '\n';
	// letters start .. end of the record's stored letters, a line break after every 60th step (a step past the stored
	// letters prints nothing but still counts)
	uint64_t line = h + 1, base = 0; // `base` = index of the first stored letter of `line`
	Span l = line_of(t, line);
	int count_down = 60;
	for (long long a = o.start; a <= o.end; a++) {
		count_down--;
		if ((uint64_t)a < o.n_trim) {
			while (line < stop && (uint64_t)a >= base + (l.n > 1 ? l.n - 1 : 0)) {
				base += l.n > 1 ? l.n - 1 : 0;
				line++;
				l = line_of(t, line);
			}
			*w++ = (char)t.text[l.off + ((uint64_t)a - base)];
		}
		if (count_down == 0) {
			*w++ = '\n';
			count_down = 60;
		}
	}
	*w++ = '\n';
}

// join_fasta: the lines one iteration of the script's loop takes from one file.  `term` = the lines >= 1 that hold a '>'
// (n_term of them); iteration k scans from the line after terminator k - 1 (line 1 for k = 0) to terminator k; without
// one it takes the rest of the file, except the last line when that is not the first it takes (eof() ends the loop
// before that line is added) -- that line is then what the next header would be made of.
struct JfSeg {
	uint64_t first, stop;  // lines [first, stop) are joined
	long long header_line; // the line the NEXT iteration prints as this file's header, or -1 (undefined)
};
__device__ JfSeg jf_segment(const uint64_t *__restrict__ term, uint64_t n_term, uint64_t n_lines, uint64_t k)
{
	JfSeg g;
	const uint64_t s = k == 0 ? 1 : (k - 1 < n_term ? term[k - 1] + 1 : n_lines);
	g.first = g.stop = s < n_lines ? s : n_lines;
	g.header_line = -1;
	if (s >= n_lines)
		return g;
	if (k < n_term) {
		g.stop = term[k];
		g.header_line = (long long)term[k];
	} else if (n_lines - 1 > s) {
		g.stop = n_lines - 1;
		g.header_line = (long long)(n_lines - 1);
	} else {
		g.stop = n_lines;
	}
	return g;
}

__device__ __forceinline__ uint64_t chomped_len(const TextView &t, uint64_t i)
{
	const Span l = line_of(t, i);
	return l.n && t.text[l.off + l.n - 1] == '\n' ? l.n - 1 : l.n;
}

// EMIT = false: the size of iteration k's output; true: its text at out_off[k]
template <bool EMIT>
__global__ void k_jf(TextView a, TextView b, const uint64_t *__restrict__ term_a, uint64_t n_term_a, const uint64_t *__restrict__ term_b,
		     uint64_t n_term_b, uint64_t n_iter, long long gap, uint64_t *__restrict__ out_len, const uint64_t *__restrict__ out_off,
		     char *__restrict__ out)
{
	const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n_iter)
		return;
	uint64_t n = 0;
	char *w = EMIT ? out + out_off[k] : nullptr;
	auto put_line = [&](const TextView &t, long long i, bool chomp, bool drop_gt) {
		if (i < 0 || (uint64_t)i >= t.n_lines)
			return;
		const Span l = line_of(t, (uint64_t)i);
		const uint64_t len = chomp ? chomped_len(t, (uint64_t)i) : l.n;
		for (uint64_t c = 0; c < len; c++) {
			const char ch = (char)t.text[l.off + c];
			if (drop_gt && ch == '>')
				continue;
			if (EMIT)
				*w++ = ch;
			n++;
		}
	};
	auto put_char = [&](char ch) {
		if (EMIT)
			*w++ = ch;
		n++;
	};
	// the header line: first iteration = line 0 of both files, the second one's not chomped; later = the lines that ended
	// the previous iteration's sequence loops, chomped, and a line break
	if (k == 0) {
		put_line(a, 0, true, false);
		put_char('_');
		put_line(b, 0, false, true);
	} else {
		put_line(a, jf_segment(term_a, n_term_a, a.n_lines, k - 1).header_line, true, false);
		put_char('_');
		put_line(b, jf_segment(term_b, n_term_b, b.n_lines, k - 1).header_line, true, true);
		put_char('\n');
	}
	const JfSeg sa = jf_segment(term_a, n_term_a, a.n_lines, k), sb = jf_segment(term_b, n_term_b, b.n_lines, k);
	for (uint64_t i = sa.first; i < sa.stop; i++)
		put_line(a, (long long)i, true, false);
	for (long long g = 0; g < gap; g++)
		put_char('N');
	for (uint64_t i = sb.first; i < sb.stop; i++)
		put_line(b, (long long)i, true, false);
	put_char('\n');
	if (!EMIT)
		out_len[k] = n;
}

// the lines of `d` that hold a '>' from line `first` on, in order
static int flagged_lines(const DeviceText &d, uint64_t first, DevBuf<uint64_t> &lines, uint64_t *count)
{
	*count = 0;
	if (d.n_lines == 0)
		return 0;
	DevBuf<uint32_t> flag;
	DevBuf<uint64_t> rank;
	PGX_TRY(flag.alloc(d.n_lines + 1, 0, 0, true));
	PGX_TRY(rank.alloc(d.n_lines + 1));
	hipLaunchKernelGGL(k_fa_flags, dim3((unsigned)((d.n_lines + 255) / 256)), dim3(256), 0, 0, d.view(), first, flag.data());
	PGX_HIP(hipGetLastError());
	PGX_TRY((exclusive_sum<uint32_t, uint64_t>(flag.data(), rank.data(), (size_t)d.n_lines + 1)));
	PGX_TRY(rank.download(count, 1, d.n_lines));
	PGX_TRY(lines.alloc(*count + 1));
	hipLaunchKernelGGL(k_fa_scatter, dim3((unsigned)((d.n_lines + 255) / 256)), dim3(256), 0, 0, flag.data(), rank.data(), d.n_lines, lines.data());
	PGX_HIP(hipGetLastError());
	return 0;
}

// parse_fasta (:384-465): what it prints, appended to `log`
static int parse_fasta_device(const std::string &a, const std::string &q, Text &log)
{
	DeviceText da;
	PGX_TRY(upload_lines(a, da));
	DevBuf<uint64_t> hdr;
	uint64_t n_rec = 0;
	PGX_TRY(flagged_lines(da, 0, hdr, &n_rec));
	if (n_rec < 2)
		return 0; // a record is printed when the next header arrives
	// the quality file, line by line: chomp, split(/ /) (leading and inner empty fields stay, trailing ones go), numified
	std::vector<uint64_t> q_off(1, 0);
	std::vector<double> q_val;
	for (size_t s0 = 0; s0 < q.size();) {
		const char *nl = (const char *)memchr(q.data() + s0, '\n', q.size() - s0);
		const size_t e = nl ? (size_t)(nl - q.data()) : q.size();
		size_t n_keep = 0, nf = 0;
		for (size_t i = s0, f0 = s0; i <= e; i++)
			if (i == e || q[i] == ' ') {
				nf++;
				if (i > f0)
					n_keep = nf;
				f0 = i + 1;
			}
		nf = 0;
		for (size_t i = s0, f0 = s0; i <= e && nf < n_keep; i++)
			if (i == e || q[i] == ' ') {
				q_val.push_back(perl_num(q.data() + f0, i - f0));
				nf++;
				f0 = i + 1;
			}
		q_off.push_back(q_val.size());
		s0 = e + 1;
	}
	DevBuf<uint64_t> d_qoff;
	DevBuf<double> d_qval;
	PGX_TRY(d_qoff.alloc(q_off.size()));
	PGX_TRY(d_qoff.upload(q_off.data(), q_off.size()));
	PGX_TRY(d_qval.alloc(q_val.size() ? q_val.size() : 1));
	PGX_TRY(d_qval.upload(q_val.data(), q_val.size()));
	DevBuf<FaRec> rec;
	DevBuf<uint64_t> out_len, out_off;
	PGX_TRY(rec.alloc(n_rec));
	PGX_TRY(out_len.alloc(n_rec + 1));
	PGX_TRY(out_off.alloc(n_rec + 1));
	hipLaunchKernelGGL(k_pf_measure, dim3((unsigned)((n_rec + 127) / 128)), dim3(128), 0, 0, da.view(), hdr.data(), n_rec, d_qoff.data(),
			   (uint64_t)(q_off.size() - 1), d_qval.data(), rec.data());
	hipLaunchKernelGGL(k_pf_carry, dim3(1), dim3(1), 0, 0, rec.data(), n_rec, out_len.data());
	PGX_HIP(hipGetLastError());
	PGX_TRY((exclusive_sum<uint64_t, uint64_t>(out_len.data(), out_off.data(), (size_t)n_rec + 1)));
	uint64_t total = 0;
	PGX_TRY(out_off.download(&total, 1, n_rec));
	DevBuf<char> out;
	PGX_TRY(out.alloc(total ? total : 1));
	hipLaunchKernelGGL(k_pf_emit, dim3((unsigned)((n_rec + 127) / 128)), dim3(128), 0, 0, da.view(), hdr.data(), n_rec, rec.data(), out_off.data(),
			   out.data());
	PGX_HIP(hipGetLastError());
	const size_t at = log.s.size();
	log.s.resize(at + total);
	if (total)
		PGX_HIP(hipMemcpy(&log.s[at], out.data(), total, hipMemcpyDeviceToHost));
	trace_point("trim fasta + qual");
	return 0;
}

// join_fasta (:301-382): the text of the output file
static int join_fasta_device(const std::string &a, const std::string &b, long long gap, char **text, size_t *len)
{
	DeviceText da, db;
	PGX_TRY(upload_lines(a, da));
	PGX_TRY(upload_lines(b, db));
	DevBuf<uint64_t> ta, tb;
	uint64_t na = 0, nb = 0;
	PGX_TRY(flagged_lines(da, 1, ta, &na));
	PGX_TRY(flagged_lines(db, 1, tb, &nb));
	// the loop runs while the -a file still has a line after the last terminator
	uint64_t last_term = 0;
	if (na)
		PGX_TRY(ta.download(&last_term, 1, na - 1));
	const uint64_t n_iter = da.n_lines == 0 ? 0 : na + 1 - (na && last_term == da.n_lines - 1 ? 1 : 0);
	DevBuf<uint64_t> out_len, out_off;
	PGX_TRY(out_len.alloc(n_iter + 1, 0, 0, true));
	PGX_TRY(out_off.alloc(n_iter + 1));
	if (n_iter) {
		hipLaunchKernelGGL(k_jf<false>, dim3((unsigned)((n_iter + 127) / 128)), dim3(128), 0, 0, da.view(), db.view(), ta.data(), na, tb.data(), nb,
				   n_iter, gap, out_len.data(), (const uint64_t *)nullptr, (char *)nullptr);
		PGX_HIP(hipGetLastError());
	}
	PGX_TRY((exclusive_sum<uint64_t, uint64_t>(out_len.data(), out_off.data(), (size_t)n_iter + 1)));
	uint64_t total = 0;
	PGX_TRY(out_off.download(&total, 1, n_iter));
	DevBuf<char> out;
	PGX_TRY(out.alloc(total ? total : 1));
	if (n_iter) {
		hipLaunchKernelGGL(k_jf<true>, dim3((unsigned)((n_iter + 127) / 128)), dim3(128), 0, 0, da.view(), db.view(), ta.data(), na, tb.data(), nb,
				   n_iter, gap, out_len.data(), out_off.data(), out.data());
		PGX_HIP(hipGetLastError());
	}
	trace_point("trim join fasta");
	return take_output(out, total, text, len);
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
	int rc = 0;
	g_clock.tick("read files");
	if (!a.empty() && a[0] == '>') { // :117-143; RUNBLAST was opened before (:115) and stays empty unless -j writes to it
		if (o->j) {
			if (paired) {
				*mode = PGX_TRIM_FASTA_JOIN;
				long long jg = -1; // `if ($parameters{g})`: without -g no N's at all, $GAPSIZE is not consulted (:338)
				if (perl_true(o->g)) {
					const double g = perl_num(o->g, strlen(o->g));
					jg = g > 0 ? (g > 2147483647.0 ? 2147483647ll : (long long)std::ceil(g)) : 0;
				}
				return finish(join_fasta_device(a, b, jg, fasta_text, fasta_len)); // exit: no closing message (:124)
			}
			log.s += "Error. Input is -j for joining ends, but you did not provided both sequence a and b with -a and -b options.\n\n";
			*mode = PGX_TRIM_UNKNOWN;
			*fasta_text = (char *)calloc(1, 1);
			return finish(0);
		}
		*mode = PGX_TRIM_UNKNOWN;
		*fasta_text = (char *)calloc(1, 1);
		if (!perl_true(o->q)) {
			log.s += "Error: Please, specify the FASTA quality file with -q option.\n";
			return finish(0);
		}
		log.printf("%s\n", o->q); // :129
		const std::string q = read_text_file(o->q, &ok);
		if (!ok) {
			log.printf("Error: Unable to open %s required for FASTA file triming.\n", o->q);
			return finish(0);
		}
		*mode = PGX_TRIM_FASTA_QUAL;
		rc = parse_fasta_device(a, q, log);
		if (rc < 0)
			return finish(rc);
		log.s += "Trimming complete.\n";
		return finish(0);
	}
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
