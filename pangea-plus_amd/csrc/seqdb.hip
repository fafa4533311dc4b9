// Sequence database and read batches: FASTA -> 2-bit packing, the .pgxdb file, upload to HBM,
// the device-built 16-mer seed index, and the synthetic workload generator kernels.
//
// Replaces `makeblastdb -in nt -out nt -dbtype nucl` (reference README.md:62) and
// `2bwt-builder ref.fasta` (README.md:130); both reference tools are external/closed, the
// format here is the build's own.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <chrono>

#include <sys/stat.h>

#include "bitops.hpp"
#include "engine.hpp"
#include "textops.hpp"

namespace pgx {

// ------------------------------------------------------------------------------------------ host packing
// FASTA text split on the host (memchr/memcpy speed): headers, per-sequence base offsets and the sequence
// letters of all records back to back.  The 2-bit packing itself runs on the device (k_pack_*).
struct FastaLetters {
	std::vector<std::string> headers;
	std::vector<uint64_t> off; // n+1 offsets into letters
	std::string letters;
};

static void split_fasta_text(const std::string &text, FastaLetters &fl)
{
	fl.off.push_back(0);
	fl.letters.reserve(text.size());
	bool in_seq = false;
	const char *base = text.data();
	size_t i = 0, n = text.size();
	while (i < n) {
		const char *nl = (const char *)memchr(base + i, '\n', n - i);
		size_t e = nl ? (size_t)(nl - base) : n;
		size_t ll = e - i;
		if (ll && base[i + ll - 1] == '\r')
			ll--;
		if (ll && base[i] == '>') {
			if (in_seq)
				fl.off.push_back(fl.letters.size());
			fl.headers.emplace_back(base + i + 1, ll - 1);
			in_seq = true;
		} else if (in_seq && ll) {
			if (!memchr(base + i, ' ', ll) && !memchr(base + i, '\t', ll)) {
				fl.letters.append(base + i, ll);
			} else {
				for (size_t k = 0; k < ll; k++)
					if (base[i + k] != ' ' && base[i + k] != '\t')
						fl.letters.push_back(base[i + k]);
			}
		}
		i = e + 1;
	}
	if (in_seq)
		fl.off.push_back(fl.letters.size());
}

struct PackedSet {
	std::vector<std::string> headers;
	std::vector<uint64_t> off; // n+1 base offsets (back to back)
	std::vector<uint64_t> words, amb;
	bool any_amb = false;
};

__device__ __forceinline__ uint32_t letter_code(unsigned char c)
{
	// A C G T/U (any case) = 0..3, everything else 4 (ambiguous)
	const unsigned char u = c & 0xDF; // upper case
	return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : (u == 'T' || u == 'U') ? 3u : 4u;
}

// database layout: word w packs letters [32w, 32w+32) of the concatenation
__global__ void k_pack_db(const unsigned char *__restrict__ letters, uint64_t n, uint64_t *__restrict__ words,
			  uint64_t *__restrict__ amb, uint64_t n_words, unsigned int *__restrict__ any_amb)
{
	uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n_words)
		return;
	uint64_t bits = 0, flags = 0;
	for (int k = 0; k < 32; k++) {
		const uint64_t i = w * 32 + k;
		if (i >= n)
			break;
		const uint32_t c = letter_code(letters[i]);
		if (c < 4)
			bits |= (uint64_t)c << (2 * k);
		else
			flags |= 1ull << (2 * k);
	}
	words[w] = bits;
	if (amb)
		amb[w] = flags;
	if (flags)
		atomicOr(any_amb, 1u);
}

// read layout: every read starts on a word boundary; fold_to_g reads every ambiguous letter as G and counts them
__global__ void k_pack_reads(const unsigned char *__restrict__ letters, const uint64_t *__restrict__ off,
			     const uint32_t *__restrict__ woff, uint64_t n_reads, int fold_to_g, uint64_t *__restrict__ fwd,
			     uint64_t *__restrict__ amb, uint32_t *__restrict__ amb_count, unsigned int *__restrict__ any_amb,
			     const uint32_t *__restrict__ len = nullptr)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads)
		return;
	const uint64_t s = off[r], L = len ? (uint64_t)len[r] : off[r + 1] - s; // `len`: ranges that are not back to back (pieces)
	const uint32_t w0 = woff[r];
	uint32_t namb = 0;
	uint64_t bits = 0, flags = 0;
	// letters come as aligned 16-byte words (textops.hpp): a lane per read walking bytes would fetch every line 64 times
	for_bytes_at(letters + s, L, [&](uint8_t ch, uint64_t i) {
		const int k = (int)(i & 31);
		uint32_t c = letter_code(ch);
		if (c >= 4) {
			namb++;
			if (fold_to_g)
				c = 2;
		}
		if (c < 4)
			bits |= (uint64_t)c << (2 * k);
		else
			flags |= 1ull << (2 * k);
		if (k == 31 || i + 1 == L) {
			fwd[w0 + (i >> 5)] = bits;
			if (amb)
				amb[w0 + (i >> 5)] = flags;
			bits = flags = 0;
		}
	});
	if (amb_count)
		amb_count[r] = namb;
	if (namb && !fold_to_g)
		atomicOr(any_amb, 1u);
}

// FASTA text -> database packed on the device; result copied back into a PackedSet (the database keeps a
// host copy for the .pgxdb file and for SOAP row formatting)
static int pack_fasta_text(const std::string &text, PackedSet &ps)
{
	FastaLetters fl;
	split_fasta_text(text, fl);
	ps.headers.swap(fl.headers);
	ps.off.swap(fl.off);
	const uint64_t n = fl.letters.size(), nw = (n + 31) / 32;
	ps.words.assign(nw, 0);
	ps.amb.assign(nw, 0);
	ps.any_amb = false;
	if (n == 0)
		return 0;
	PGX_TRY(require_device());
	DevBuf<unsigned char> d_letters;
	DevBuf<uint64_t> d_w, d_a;
	DevBuf<unsigned int> d_flag;
	PGX_TRY(d_letters.alloc(n));
	PGX_TRY(d_letters.upload((const unsigned char *)fl.letters.data(), n));
	PGX_TRY(d_w.alloc(nw));
	PGX_TRY(d_a.alloc(nw));
	PGX_TRY(d_flag.alloc(1, 0, 0, true));
	hipLaunchKernelGGL(k_pack_db, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, 0, d_letters.data(), n, d_w.data(), d_a.data(),
			   nw, d_flag.data());
	PGX_HIP(hipGetLastError());
	unsigned int flag = 0;
	PGX_TRY(d_flag.download(&flag, 1));
	ps.any_amb = flag != 0;
	PGX_TRY(d_w.download(ps.words.data(), nw));
	PGX_TRY(d_a.download(ps.amb.data(), nw));
	return 0;
}

static std::string first_word(const std::string &h)
{
	size_t k = 0;
	while (k < h.size() && h[k] != ' ' && h[k] != '\t')
		k++;
	return h.substr(0, k);
}

int choose_index_bits(int64_t n_postings)
{
	// PGX_INDEX_BITS forces the bucket-table width (tests use 32 to exercise the direct-address paths on
	// small databases)
	if (const char *e = getenv("PGX_INDEX_BITS")) {
		int v = atoi(e);
		if (v >= 16 && v <= 32)
			return v;
	}
	int lg = 0;
	while (lg < 62 && (1ll << lg) < n_postings)
		lg++;
	int b = lg + 2;
	if (b < 16)
		b = 16;
	if (b > 32)
		b = 32;
	return b;
}

// ------------------------------------------------------------------------------------------ device: index build
__global__ void k_blk_subj(const uint32_t *__restrict__ seq_off, uint32_t n_seq, uint32_t *__restrict__ blk,
			   uint64_t n_blk)
{
	uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blk)
		return;
	uint64_t p = b << kBlkShift;
	uint32_t lo = 0, hi = n_seq; // last subject with seq_off <= p
	while (hi - lo > 1) {
		uint32_t mid = lo + (hi - lo) / 2;
		if (seq_off[mid] <= p)
			lo = mid;
		else
			hi = mid;
	}
	blk[b] = lo;
}

// posting = position of the 16-mer in the concatenated database (full 32 bits: databases up to 4.29 Gbp)
__global__ void k_seed_keys(const uint64_t *__restrict__ words, uint64_t n_pos, int bits, uint32_t *__restrict__ keys,
			    uint32_t *__restrict__ vals)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < n_pos; i += stride) {
		keys[i] = seed_bucket(kmer16(words, (int64_t)i), bits);
		vals[i] = (uint32_t)i;
	}
}

// post_ctx[i] = posting i with the database around it, in index order: x = the posting, y = the 16 bases from 13 left of the 16-mer,
// z = the 16 bases right of it.  k_seed_extend's duplicate and short-run filters read them from this
// contiguous stream instead of fetching a random database line per posting.
// The top bits of y belong to bases of the 16-mer itself, which the filters never look at; they carry two flags:
// bit 31: one of the 41 bases from 13 left of the 16-mer to 12 right of it is an ambiguity letter (the filters then
//         leave the posting alone);
// bit 30: the posting lies within 13 bases of its sequence's start: the probe 13 bases to the left lies (partly) in
//         the previous sequence, so the duplicate filter must not trust it.
__global__ void k_post_ctx(const uint64_t *__restrict__ words, const uint64_t *__restrict__ amb, const uint32_t *__restrict__ postings,
			   const uint32_t *__restrict__ seq_off, const uint32_t *__restrict__ blk_subj, uint64_t n,
			   uint3 *__restrict__ ctx)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) {
		const uint32_t raw = postings[i];
		const int64_t p = (int64_t)raw;
		uint32_t y = window16(words, p - kProbeStride) & 0x3FFFFFFFu;
		uint32_t sj = blk_subj[raw >> kBlkShift];
		while (seq_off[sj + 1] <= raw)
			sj++;
		if (raw - seq_off[sj] < (uint32_t)kProbeStride)
			y |= 0x40000000u;
		if (amb) {
			// spaced flags: 32 bases from p-13 (all of them count: up to p+18), then the 9 bases p+19 .. p+27
			const uint64_t a0 = window64(amb, p - kProbeStride), a1 = window64(amb, p + 19);
			if (a0 | (a1 & ((1ull << 18) - 1)))
				y |= 0x80000000u;
		}
		ctx[i] = make_uint3(raw, y, window16(words, p + kSeedK));
	}
}

// run heads of the sorted keys write the run length into counts[key]
__global__ void k_bucket_counts(const uint32_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ counts)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) {
		uint32_t k = keys[i];
		if (i > 0 && keys[i - 1] == k)
			continue;
		// end of the run: short linear probe (most buckets hold a few postings), then bisect
		uint64_t lo = i;
		int c = 0;
		while (lo + 1 < n && c < 32 && keys[lo + 1] == k) {
			lo++;
			c++;
		}
		if (lo + 1 < n && keys[lo + 1] == k) {
			uint64_t hi = n; // keys[lo] == k, first index with a larger key is in (lo, hi]
			while (hi - lo > 1) {
				uint64_t mid = lo + (hi - lo) / 2;
				if (keys[mid] == k)
					lo = mid;
				else
					hi = mid;
			}
		}
		counts[k] = (uint32_t)(lo + 1 - i);
	}
}

// ---- exclusive scan of a uint32 table of any length (tiles of 4096: 256 threads x 16 consecutive entries)
constexpr uint64_t kTile = 4096;

__device__ __forceinline__ uint32_t block_exclusive_256(uint32_t v, uint32_t *lds, uint32_t *total)
{
	// 256 threads; returns the exclusive prefix of v over the block, *total = block sum
	const int t = threadIdx.x, lane = t & 63, w = t >> 6;
	uint32_t incl = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t x = __shfl_up(incl, d);
		if (lane >= d)
			incl += x;
	}
	if (lane == 63)
		lds[w] = incl;
	__syncthreads();
	uint32_t base = 0;
	for (int k = 0; k < w; k++)
		base += lds[k];
	*total = lds[0] + lds[1] + lds[2] + lds[3];
	__syncthreads();
	return base + incl - v;
}

__global__ __launch_bounds__(256) void k_tile_sums(const uint32_t *__restrict__ a, uint64_t n, uint64_t n_tiles,
						    uint32_t *__restrict__ tile_sum)
{
	__shared__ uint32_t lds[4];
	for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint64_t i0 = tile * kTile + (uint64_t)threadIdx.x * 16;
		uint32_t v = 0;
#pragma unroll
		for (int k = 0; k < 16; k++)
			if (i0 + k < n)
				v += a[i0 + k];
		uint32_t tot;
		(void)block_exclusive_256(v, lds, &tot);
		if (threadIdx.x == 0)
			tile_sum[tile] = tot;
	}
}

// one block: exclusive scan of the tile sums in place
__global__ __launch_bounds__(1024) void k_tile_top(uint32_t *__restrict__ tile_sum, uint64_t n_tiles)
{
	__shared__ uint32_t wsum[16];
	__shared__ uint32_t carry_s;
	const int t = threadIdx.x, lane = t & 63, w = t >> 6;
	if (t == 0)
		carry_s = 0;
	__syncthreads();
	for (uint64_t base = 0; base < n_tiles; base += 1024) {
		const uint64_t i = base + t;
		const uint32_t v = i < n_tiles ? tile_sum[i] : 0u;
		uint32_t incl = v;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			uint32_t x = __shfl_up(incl, d);
			if (lane >= d)
				incl += x;
		}
		if (lane == 63)
			wsum[w] = incl;
		__syncthreads();
		uint32_t pre = carry_s;
		for (int k = 0; k < w; k++)
			pre += wsum[k];
		if (i < n_tiles)
			tile_sum[i] = pre + incl - v;
		__syncthreads();
		if (t == 1023)
			carry_s = pre + incl;
		__syncthreads();
	}
}

__global__ __launch_bounds__(256) void k_tile_scan(uint32_t *__restrict__ a, uint64_t n, uint64_t n_tiles,
						    const uint32_t *__restrict__ tile_sum)
{
	__shared__ uint32_t lds[4];
	for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint64_t i0 = tile * kTile + (uint64_t)threadIdx.x * 16;
		uint32_t x[16], v = 0;
#pragma unroll
		for (int k = 0; k < 16; k++) {
			x[k] = i0 + k < n ? a[i0 + k] : 0u;
			v += x[k];
		}
		uint32_t tot;
		uint32_t run = tile_sum[tile] + block_exclusive_256(v, lds, &tot);
#pragma unroll
		for (int k = 0; k < 16; k++) {
			if (i0 + k < n)
				a[i0 + k] = run;
			run += x[k];
		}
	}
}

// debugging aid (PGX_TRACE=1): structural check of the seed index
__global__ void k_index_check(const uint32_t *__restrict__ bucket_off, uint64_t nb, const uint32_t *__restrict__ postings,
			      uint64_t n_post, uint64_t n_bases, unsigned long long *__restrict__ bad)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t k = i; k < nb; k += stride)
		if (bucket_off[k] > bucket_off[k + 1] || bucket_off[k + 1] > n_post) {
			atomicAdd(&bad[0], 1ull);
			atomicMin(&bad[2], (unsigned long long)k);
			atomicMax(&bad[3], (unsigned long long)k);
		}
	for (uint64_t k = i; k < n_post; k += stride)
		if ((uint64_t)postings[k] + (uint64_t)kSeedK > n_bases)
			atomicAdd(&bad[1], 1ull);
}

void index_check(const pgx_db *db, const char *where)
{
	if (!getenv("PGX_TRACE") || !db || db->n_postings == 0)
		return;
	DevBuf<unsigned long long> bad;
	if (bad.alloc(4, 0, 0, true) < 0)
		return;
	const unsigned long long init[4] = { 0, 0, ~0ull, 0 };
	bad.upload(init, 4);
	hipLaunchKernelGGL(k_index_check, dim3(256 * 32), dim3(256), 0, 0, db->d_bucket_off.data(), 1ull << db->index_bits,
			   db->d_postings.data(), (uint64_t)db->n_postings, (uint64_t)db->n_bases, bad.data());
	unsigned long long h[4] = { 0, 0, 0, 0 };
	bad.download(h, 4);
	fprintf(stderr, "[pgx trace] index check at %s: %llu bad bucket offsets (first %llu last %llu), %llu bad postings\n", where,
		h[0], h[2], h[3], h[1]);
	if (h[0]) {
		uint32_t v[6];
		const uint64_t k0 = h[2] >= 2 ? h[2] - 2 : 0;
		db->d_bucket_off.download(v, 6, k0);
		fprintf(stderr, "[pgx trace]   bucket_off[%llu..] = %u %u %u %u %u %u\n", (unsigned long long)k0, v[0], v[1], v[2], v[3], v[4], v[5]);
	}
	fflush(stderr);
}

int db_build_index(pgx_db *db)
{
	db->n_postings = db->n_bases >= kSeedK ? db->n_bases - kSeedK + 1 : 0;
	db->index_bits = choose_index_bits(db->n_postings);
	const uint64_t n = (uint64_t)db->n_postings;
	const uint64_t nb = 1ull << db->index_bits;
	PGX_TRY(db->d_bucket_off.alloc(nb + 1, 0, 0, true));
	PGX_TRY(db->d_postings.alloc(n ? n : 1));
	if (n == 0)
		return 0;
	DevBuf<uint32_t> keys_in, keys_out, vals_in;
	PGX_TRY(keys_in.alloc(n));
	PGX_TRY(keys_out.alloc(n));
	PGX_TRY(vals_in.alloc(n));
	int grid = (int)std::min<uint64_t>((n + 255) / 256, 256 * 32);
	hipLaunchKernelGGL(k_seed_keys, dim3(grid), dim3(256), 0, 0, db->d_words.data(), n, db->index_bits, keys_in.data(), vals_in.data());
	PGX_HIP(hipGetLastError());
	size_t tmp_bytes = 0;
	PGX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in.data(), keys_out.data(), vals_in.data(),
					  db->d_postings.data(), n, 0, db->index_bits));
	DevBuf<uint8_t> tmp;
	PGX_TRY(tmp.alloc(tmp_bytes));
	PGX_HIP(rocprim::radix_sort_pairs(tmp.data(), tmp_bytes, keys_in.data(), keys_out.data(), vals_in.data(),
					  db->d_postings.data(), n, 0, db->index_bits));
	hipLaunchKernelGGL(k_bucket_counts, dim3(grid), dim3(256), 0, 0, keys_out.data(), n, db->d_bucket_off.data());
	PGX_HIP(hipGetLastError());
	keys_in.release();
	vals_in.release();
	PGX_TRY(db->d_post_ctx.alloc(n));
	hipLaunchKernelGGL(k_post_ctx, dim3(grid), dim3(256), 0, 0, db->d_words.data(), db->has_amb ? db->d_amb.data() : (const uint64_t *)nullptr,
			   db->d_postings.data(), db->d_seq_off.data(), db->d_blk_subj.data(), n, db->d_post_ctx.data());
	PGX_HIP(hipGetLastError());
	// exclusive scan of the counts in place -> bucket offsets; the extra last element becomes n.
	// (Own three-pass scan with 64-bit indexing: the table has 2^32 + 1 entries at full size, and
	// rocprim::exclusive_scan returned a doubled prefix for the last 4 097 of them.)
	{
		const uint64_t total = nb + 1, n_tiles = (total + kTile - 1) / kTile;
		DevBuf<uint32_t> tile_sum;
		PGX_TRY(tile_sum.alloc(n_tiles));
		const int tg = (int)std::min<uint64_t>(n_tiles, 256 * 64);
		hipLaunchKernelGGL(k_tile_sums, dim3(tg), dim3(256), 0, 0, db->d_bucket_off.data(), total, n_tiles, tile_sum.data());
		hipLaunchKernelGGL(k_tile_top, dim3(1), dim3(1024), 0, 0, tile_sum.data(), n_tiles);
		hipLaunchKernelGGL(k_tile_scan, dim3(tg), dim3(256), 0, 0, db->d_bucket_off.data(), total, n_tiles, tile_sum.data());
		PGX_HIP(hipGetLastError());
		PGX_HIP(hipDeviceSynchronize());
	}
	PGX_HIP(hipDeviceSynchronize());
	index_check(db, "db_build_index");
	return 0;
}

// blk_info[b] = { s, seq_off[s], seq_off[s+1], seq_off[s+2] } for s = blk_subj[b]: the subject of a position and
// its bounds from ONE 16-byte load (one dependent memory hop less in k_seed_extend than blk_subj -> seq_off)
__global__ void k_blk_info(const uint32_t *__restrict__ seq_off, uint32_t n_seq, const uint32_t *__restrict__ blk,
			   uint4 *__restrict__ info, uint64_t n_blk)
{
	uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blk)
		return;
	const uint32_t s = blk[b];
	const uint32_t s1 = s + 1 < n_seq ? s + 1 : n_seq, s2 = s + 2 < n_seq ? s + 2 : n_seq;
	info[b] = make_uint4(s, seq_off[s], seq_off[s1], seq_off[s2]);
}

// bit b of amb_blk: block b (512 bases = 16 flag words) holds an ambiguity letter.  Diagonals that touch no such
// block build their flags without reading the ambiguity words at all.
__global__ void k_amb_blocks(const uint64_t *__restrict__ amb, uint64_t n_words, uint64_t n_blk, uint32_t *__restrict__ bits)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // one 32-block word of the bitmap per thread
	if (w * 32 >= n_blk)
		return;
	uint32_t v = 0;
	for (int k = 0; k < 32; k++) {
		const uint64_t b = w * 32 + k;
		uint64_t any = 0;
		for (int j = 0; j < 16; j++) {
			const uint64_t i = b * 16 + j;
			if (i < n_words)
				any |= amb[i];
		}
		if (any)
			v |= 1u << k;
	}
	bits[w] = v;
}

int db_build_blk_info(pgx_db *db)
{
	const uint64_t n_blk = ((uint64_t)db->n_bases >> kBlkShift) + 2;
	if (db->has_amb) {
		const uint64_t n_bm = (n_blk + 31) / 32 + 1;
		PGX_TRY(db->d_amb_blk.alloc(n_bm, 0, 0, true));
		hipLaunchKernelGGL(k_amb_blocks, dim3((unsigned)((n_bm + 255) / 256)), dim3(256), 0, 0, db->d_amb.data(),
				   ((uint64_t)db->n_bases + 31) / 32, n_blk, db->d_amb_blk.data());
		PGX_HIP(hipGetLastError());
	}
	PGX_TRY(db->d_blk_info.alloc(n_blk));
	if (db->n_seq > 0) {
		hipLaunchKernelGGL(k_blk_info, dim3((unsigned)((n_blk + 255) / 256)), dim3(256), 0, 0, db->d_seq_off.data(),
				   (uint32_t)db->n_seq, db->d_blk_subj.data(), db->d_blk_info.data(), n_blk);
		PGX_HIP(hipGetLastError());
	}
	return 0;
}

// block tables derived from d_seq_off (subject of every 512-base block, then the 16-byte block records)
static int db_build_blocks(pgx_db *db)
{
	uint64_t n_blk = ((uint64_t)db->n_bases >> kBlkShift) + 2;
	PGX_TRY(db->d_blk_subj.alloc(n_blk));
	if (db->n_seq > 0) {
		hipLaunchKernelGGL(k_blk_subj, dim3((unsigned)((n_blk + 255) / 256)), dim3(256), 0, 0,
				   db->d_seq_off.data(), (uint32_t)db->n_seq, db->d_blk_subj.data(), n_blk);
		PGX_HIP(hipGetLastError());
	}
	return db_build_blk_info(db);
}

static int db_upload_offsets(pgx_db *db)
{
	PGX_TRY(db->d_seq_off.alloc((size_t)db->n_seq + 1));
	PGX_TRY(db->d_seq_off.upload(db->h_seq_off.data(), (size_t)db->n_seq + 1));
	return db_build_blocks(db);
}

int db_upload_and_index(pgx_db *db)
{
	PGX_TRY(require_device());
	if (db->n_bases >= (1ll << 32) - 64)
		return fail(PGX_E_LIMIT, "database of %lld bases exceeds the 32-bit position limit of this build",
			    (long long)db->n_bases);
	size_t nw = ((size_t)db->n_bases + 31) / 32;
	PGX_TRY(db->d_words.alloc(nw, 24, 24, true));
	PGX_TRY(db->d_words.upload(db->h_words.data(), std::min(nw, db->h_words.size())));
	if (db->has_amb) {
		PGX_TRY(db->d_amb.alloc(nw, 24, 24, true));
		PGX_TRY(db->d_amb.upload(db->h_amb.data(), std::min(nw, db->h_amb.size())));
	}
	PGX_TRY(db_upload_offsets(db));
	return db_build_index(db);
}

static void synth_ids(pgx_db *db)
{
	db->ids.clear();
	db->ids.reserve((size_t)db->n_seq);
	char buf[64];
	for (int64_t i = 0; i < db->n_seq; i++) {
		snprintf(buf, sizeof buf, "gi|%lld|syn|S%lld|", (long long)(1000 + i), (long long)i);
		db->ids.emplace_back(buf);
	}
}

static int db_from_packed(PackedSet &ps, pgx_db **out)
{
	pgx_db *db = new pgx_db();
	db->n_seq = (int64_t)ps.headers.size();
	db->n_bases = (int64_t)ps.off.back();
	db->has_amb = ps.any_amb;
	db->h_seq_off.resize(ps.off.size());
	for (size_t i = 0; i < ps.off.size(); i++)
		db->h_seq_off[i] = (uint32_t)ps.off[i];
	for (auto &h : ps.headers)
		db->ids.push_back(first_word(h));
	db->h_words.swap(ps.words);
	db->h_amb.swap(ps.amb);
	*out = db;
	return 0;
}

// ------------------------------------------------------------------------------------------ .pgxdb file
static const char kMagic[8] = { 'P', 'G', 'X', 'D', 'B', '1', 0, 0 };

static int db_write_file(const pgx_db *db, const char *prefix)
{
	std::string path = std::string(prefix) + ".pgxdb";
	FILE *f = fopen(path.c_str(), "wb");
	if (!f)
		return fail(PGX_E_IO, "cannot open %s for writing", path.c_str());
	int64_t hdr[4] = { db->n_seq, db->n_bases, db->has_amb ? 1 : 0, (int64_t)db->h_words.size() };
	bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(hdr, sizeof hdr, 1, f) == 1;
	ok = ok && fwrite(db->h_seq_off.data(), 4, db->h_seq_off.size(), f) == db->h_seq_off.size();
	ok = ok && (db->h_words.empty() || fwrite(db->h_words.data(), 8, db->h_words.size(), f) == db->h_words.size());
	if (db->has_amb)
		ok = ok && fwrite(db->h_amb.data(), 8, db->h_amb.size(), f) == db->h_amb.size();
	for (auto &id : db->ids) {
		uint32_t l = (uint32_t)id.size();
		ok = ok && fwrite(&l, 4, 1, f) == 1 && (l == 0 || fwrite(id.data(), 1, l, f) == l);
	}
	if (fclose(f) || !ok)
		return fail(PGX_E_IO, "short write to %s", path.c_str());
	return 0;
}

static int db_read_file(const char *prefix, pgx_db **out)
{
	std::string path = std::string(prefix) + ".pgxdb";
	FILE *f = fopen(path.c_str(), "rb");
	if (!f)
		return fail(PGX_E_IO, "cannot open database %s", path.c_str());
	char magic[8];
	int64_t hdr[4];
	if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kMagic, 8) != 0 || fread(hdr, sizeof hdr, 1, f) != 1) {
		fclose(f);
		return fail(PGX_E_FORMAT, "%s is not a pgxdb file", path.c_str());
	}
	// the header is checked against the file before anything is sized by it (a damaged file used to end in bad_alloc)
	struct stat sb;
	const bool have_size = fstat(fileno(f), &sb) == 0;
	const int64_t n_seq = hdr[0], n_bases = hdr[1], n_words = hdr[3];
	const bool sane = n_seq >= 0 && n_bases >= 0 && n_bases < (1ll << 32) - 64 && n_seq <= n_bases + 1 && n_words == (n_bases + 31) / 32 &&
			  (!have_size || (int64_t)sb.st_size >= 40 + 4 * (n_seq + 1) + 8 * n_words * (hdr[2] ? 2 : 1) + 4 * n_seq);
	if (!sane) {
		fclose(f);
		return fail(PGX_E_FORMAT, "%s: header does not fit the file (%lld sequences, %lld bases, %lld words)", path.c_str(),
			    (long long)n_seq, (long long)n_bases, (long long)n_words);
	}
	pgx_db *db = new pgx_db();
	db->n_seq = hdr[0];
	db->n_bases = hdr[1];
	db->has_amb = hdr[2] != 0;
	db->h_seq_off.resize((size_t)db->n_seq + 1);
	db->h_words.resize((size_t)hdr[3]);
	bool ok = fread(db->h_seq_off.data(), 4, db->h_seq_off.size(), f) == db->h_seq_off.size();
	ok = ok && (db->h_words.empty() || fread(db->h_words.data(), 8, db->h_words.size(), f) == db->h_words.size());
	if (db->has_amb) {
		db->h_amb.resize((size_t)hdr[3]);
		ok = ok && fread(db->h_amb.data(), 8, db->h_amb.size(), f) == db->h_amb.size();
	}
	for (int64_t i = 0; ok && i < db->n_seq; i++) {
		uint32_t l;
		ok = fread(&l, 4, 1, f) == 1 && l < (1u << 20);
		std::string id(ok ? l : 0, '\0');
		ok = ok && (l == 0 || fread(&id[0], 1, l, f) == l);
		db->ids.push_back(id);
	}
	fclose(f);
	if (!ok) {
		delete db;
		return fail(PGX_E_FORMAT, "%s is truncated", path.c_str());
	}
	*out = db;
	return 0;
}

// ------------------------------------------------------------------------------------------ synthetic DB on device
__device__ __forceinline__ uint32_t synth_db_base(uint64_t seed, uint64_t i, uint64_t g, uint32_t j)
{
	uint64_t a = synth_hash(seed, 1, g, j >> 5);
	uint32_t b = (uint32_t)(a >> (2 * (j & 31))) & 3;
	uint64_t m = synth_hash(seed, 2, i, j >> 2);
	uint32_t f = (uint32_t)(m >> (16 * (j & 3))) & 0xFFFF;
	if (f < 1966)
		b = (b + 1 + f % 3) & 3;
	return b;
}

// one thread per output word (32 bases); sequences are back to back, so a word may straddle two
__global__ void k_synth_db(uint64_t seed, uint64_t n_seq, uint32_t seq_len, uint64_t n_genus,
			   uint64_t *__restrict__ words, uint64_t n_words, uint64_t n_bases)
{
	uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n_words)
		return;
	uint64_t out = 0;
	uint64_t p0 = w * 32;
	for (int k = 0; k < 32; k++) {
		uint64_t p = p0 + k;
		if (p >= n_bases)
			break;
		uint64_t i = p / seq_len;
		uint32_t j = (uint32_t)(p - i * seq_len);
		uint64_t g = (uint64_t)(((unsigned __int128)i * n_genus) / n_seq);
		out |= (uint64_t)synth_db_base(seed, i, g, j) << (2 * k);
	}
	words[w] = out;
}

// ------------------------------------------------------------------------------------------ reads
// reverse complement of every read, one thread per output word
__global__ void k_revcomp(const uint64_t *__restrict__ fwd, const uint32_t *__restrict__ woff,
			  const uint32_t *__restrict__ len, uint64_t n_reads, uint64_t *__restrict__ rc, int is_mask)
{
	uint64_t r = (uint64_t)blockIdx.x;
	for (; r < n_reads; r += gridDim.x) {
		uint32_t L = len[r], w0 = woff[r], nw = (L + 31) / 32;
		for (uint32_t w = threadIdx.x; w < nw; w += blockDim.x) {
			// rc bases [32w, 32w+32) = forward bases [L-32w-32, L-32w) reversed
			int64_t s = (int64_t)L - 32 * (int64_t)w - 32;
			uint64_t x;
			if (s >= 0) {
				int64_t wi = s >> 5;
				int sh = (int)(s & 31) * 2;
				uint64_t lo = fwd[w0 + wi];
				uint64_t hi = (sh && wi + 1 < nw) ? fwd[w0 + wi + 1] : 0;
				x = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
			} else {
				// fewer than 32 forward bases remain: they are bases [0, 32+s)
				x = fwd[w0] << (2 * (int)(-s));
			}
			uint64_t y = reverse_groups(x);
			int valid = (int)((int64_t)L - 32 * (int64_t)w);
			if (valid > 32)
				valid = 32;
			uint64_t keep = valid >= 32 ? ~0ull : ((1ull << (2 * valid)) - 1);
			if (!is_mask)
				y = ~y;
			rc[w0 + w] = y & keep;
		}
	}
}

// Where a read's packed words go.  A strand of up to 256 bases (8 words) never straddles a 64-byte line, longer ones start
// on a line: the seed stage fetches both strands of every read and the gapped stage one strand per HSP, each as whole
// lines -- 150-base reads (5 words) at a stride of 5 words straddled a line in three cases of five (1.6 lines per fetch;
// MI355X delivers ~50 G random lines a second whatever their use).  Costs 3 words in 8 for 150-base reads (0.5 GB per 10 M).
static inline uint64_t place_read_words(uint64_t cursor, uint64_t nw)
{
	if (nw > 8 || (cursor & 7) + nw > 8)
		cursor = (cursor + 7) & ~7ull;
	return cursor;
}

__global__ void k_synth_reads(uint64_t seed, uint64_t n_seq, uint32_t seq_len, uint64_t n_genus, uint64_t read_seed,
			      uint32_t read_len, uint64_t first, uint64_t count, uint32_t words_per_read, uint32_t stride,
			      uint64_t *__restrict__ fwd)
{
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= count * words_per_read)
		return;
	uint64_t ri = t / words_per_read;
	uint32_t w = (uint32_t)(t - ri * words_per_read);
	uint64_t r = first + ri;
	uint64_t u = synth_hash(read_seed, 3, r, 0);
	uint64_t i = (u & 0xFFFFFFFFull) % n_seq;
	uint32_t off = (uint32_t)((u >> 32) % (uint64_t)(seq_len - read_len + 1));
	int minus = (int)(synth_hash(read_seed, 3, r, 1) & 1);
	uint64_t g = (uint64_t)(((unsigned __int128)i * n_genus) / n_seq);
	uint64_t out = 0;
	for (int k = 0; k < 32; k++) {
		uint32_t pos = w * 32 + k; // position in the emitted read
		if (pos >= read_len)
			break;
		uint32_t j = minus ? read_len - 1 - pos : pos; // position in the sampled window
		uint32_t b = synth_db_base(seed, i, g, off + j);
		uint64_t e = synth_hash(read_seed, 4, r, j >> 2);
		uint32_t f = (uint32_t)(e >> (16 * (j & 3))) & 0xFFFF;
		if (f < 655)
			b = (b + 1 + f % 3) & 3;
		if (minus)
			b = 3 - b;
		out |= (uint64_t)b << (2 * k);
	}
	fwd[ri * stride + w] = out;
}

static int reads_build_classes(pgx_reads *rd)
{
	auto words_for = [](uint32_t L) { return L <= 192u ? 3 : (L <= 320u ? 5 : (L <= 512u ? 8 : 0)); };
	const size_t n = (size_t)rd->n;
	rd->classes.clear();
	const bool per_read_amb = rd->has_amb && rd->h_read_amb.size() == n;
	const bool per_read_dust = rd->has_dust && rd->h_read_dust.size() == n;
	bool uniform = true;
	const int w0 = words_for(n ? rd->h_len[0] : (uint32_t)rd->max_len);
	if (!rd->synthetic) // synthetic batches have one length and no ambiguity letters
		for (size_t r = 0; r < n && uniform; r++)
			uniform = words_for(rd->h_len[r]) == w0 && !(per_read_amb && rd->h_read_amb[r] != rd->h_read_amb[0]);
	if (per_read_dust) // (a few per cent of random reads hold a homopolymer of seven: they get a class of their own)
		for (size_t r = 0; r < n && uniform; r++)
			uniform = rd->h_read_dust[r] == rd->h_read_dust[0];
	if (uniform) {
		const int a = rd->has_amb && (!per_read_amb || (n && rd->h_read_amb[0]));
		rd->classes.push_back({ a, w0, 0u, (uint32_t)n, false, per_read_dust && n && rd->h_read_dust[0] ? 1 : 0 });
		return 0;
	}
	std::vector<uint32_t> ids[16];
	for (size_t r = 0; r < n; r++) {
		const int a = rd->has_amb && (!per_read_amb || rd->h_read_amb[r]);
		const int d = per_read_dust && rd->h_read_dust[r];
		const int w = words_for(rd->h_len[r]);
		ids[(d * 2 + a) * 4 + (w == 3 ? 0 : w == 5 ? 1 : w == 8 ? 2 : 3)].push_back((uint32_t)r);
	}
	std::vector<uint32_t> all;
	all.reserve(n);
	static const int kWords[4] = { 3, 5, 8, 0 };
	for (int c = 0; c < 16; c++)
		if (!ids[c].empty()) {
			rd->classes.push_back({ (c / 4) & 1, kWords[c % 4], (uint32_t)all.size(), (uint32_t)ids[c].size(), true, c / 8 });
			all.insert(all.end(), ids[c].begin(), ids[c].end());
		}
	PGX_TRY(rd->d_class_list.alloc(all.size()));
	return rd->d_class_list.upload(all.data(), all.size());
}

static int reads_finish(pgx_reads *rd)
{
	// d_fwd (and d_fwd_amb) are filled; build offsets/lengths on device and the rc strand
	PGX_TRY(rd->d_len.alloc((size_t)rd->n));
	PGX_TRY(rd->d_len.upload(rd->h_len.data(), (size_t)rd->n));
	if (!rd->d_woff.base) {
		PGX_TRY(rd->d_woff.alloc((size_t)rd->n + 1));
		PGX_TRY(rd->d_woff.upload(rd->h_woff.data(), (size_t)rd->n + 1));
	}
	PGX_TRY(rd->d_rc.alloc((size_t)rd->n_words + 24, 0, 0, true));
	if (rd->n == 0)
		return reads_build_classes(rd);
	int grid = (int)std::min<int64_t>(rd->n, 65535 * 16);
	int block = rd->max_len > 2048 ? 256 : 64;
	hipLaunchKernelGGL(k_revcomp, dim3(grid), dim3(block), 0, 0, rd->d_fwd.data(), rd->d_woff.data(),
			   rd->d_len.data(), (uint64_t)rd->n, rd->d_rc.data(), 0);
	PGX_HIP(hipGetLastError());
	if (rd->has_amb) {
		PGX_TRY(rd->d_rc_amb.alloc((size_t)rd->n_words + 24, 0, 0, true));
		hipLaunchKernelGGL(k_revcomp, dim3(grid), dim3(block), 0, 0, rd->d_fwd_amb.data(), rd->d_woff.data(),
				   rd->d_len.data(), (uint64_t)rd->n, rd->d_rc_amb.data(), 1);
		PGX_HIP(hipGetLastError());
	}
	PGX_HIP(hipDeviceSynchronize());
	if (!getenv("PGX_NO_DUST"))
		PGX_TRY(reads_dust(rd));
	return reads_build_classes(rd); // (after the DUST pass: reads with masked bases are a search class of their own)
}


// FASTA -> read batch. fold_to_g: every non-ACGT letter is read as G (what the reference's soap does,
// observed) and `amb_count[i]` receives how many letters of read i were folded.
// ------------------------------------------------------------------------------------------ FASTA text on the device
// The read file is uploaded as it is; lines, records, letters and names are found by kernels (the host never
// walks the text).  Same rules as split_fasta_text: a record starts at a line whose first byte is '>', lines
// before the first record are skipped, a trailing CR is dropped, blanks and tabs inside sequence lines are
// dropped, every other byte of a sequence line is a letter.  Files of 4 GiB and more take the host splitter.
__global__ void k_fa_line_flags(const unsigned char *__restrict__ text, uint32_t n, uint8_t *__restrict__ flag)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		flag[i] = i == 0 || text[i - 1] == '\n';
}

__global__ void k_fa_line_starts(const uint8_t *__restrict__ flag, const uint32_t *__restrict__ idx, uint32_t n,
				 uint32_t *__restrict__ line_start)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && flag[i])
		line_start[idx[i]] = i;
}

// per line: is it a header, how many letters it holds, and (headers) where the name's first word ends
__global__ void k_fa_line_info(const unsigned char *__restrict__ text, const uint32_t *__restrict__ line_start, uint32_t n_lines,
			       uint32_t *__restrict__ hdr, uint32_t *__restrict__ nlet, uint32_t *__restrict__ name_len)
{
	const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
	if (l >= n_lines)
		return;
	const uint32_t s = line_start[l];
	uint32_t e = line_start[l + 1];
	if (e > s && text[e - 1] == '\n')
		e--;
	if (e > s && text[e - 1] == '\r')
		e--;
	const bool h = e > s && text[s] == '>';
	uint32_t c = 0, nl = 0;
	if (h) {
		nl = e - (s + 1);
		bool found = false;
		for_bytes(text, (uint64_t)s + 1, (uint64_t)(e - (s + 1)), [&](uint8_t b, uint64_t i) {
			if (!found && (b == ' ' || b == '\t')) {
				nl = (uint32_t)i;
				found = true;
			}
		});
	} else {
		for_bytes(text, (uint64_t)s, (uint64_t)(e - s), [&](uint8_t b, uint64_t) { c += b != ' ' && b != '\t'; });
	}
	hdr[l] = h ? 1u : 0u;
	nlet[l] = c;
	name_len[l] = nl;
}

// letters of lines in front of the first record do not count
__global__ void k_fa_drop_preamble(const uint32_t *__restrict__ rec_incl, uint32_t n_lines, uint32_t *__restrict__ nlet)
{
	const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
	if (l < n_lines && rec_incl[l] == 0)
		nlet[l] = 0;
}

__global__ void k_fa_copy_letters(const unsigned char *__restrict__ text, const uint32_t *__restrict__ line_start,
				  const uint32_t *__restrict__ hdr, const uint32_t *__restrict__ nlet,
				  const uint32_t *__restrict__ let_off, const uint32_t *__restrict__ rec_incl,
				  const uint32_t *__restrict__ name_len, uint32_t n_lines, unsigned char *__restrict__ letters,
				  uint32_t *__restrict__ rec_let_off, uint32_t *__restrict__ rec_name_off, uint32_t *__restrict__ rec_name_len)
{
	// one group of kGroup lanes per line (textops.hpp): coalesced reads and writes, blanks squeezed out by ballot
	const uint32_t l = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup);
	if (l >= n_lines)
		return;
	const uint32_t s = line_start[l];
	if (hdr[l]) {
		if ((threadIdx.x & (kGroup - 1)) == 0) {
			const uint32_t r = rec_incl[l] - 1;
			rec_let_off[r] = let_off[l];
			rec_name_off[r] = s + 1;
			rec_name_len[r] = name_len[l];
		}
		return;
	}
	if (nlet[l] == 0)
		return; // an empty line, or a line in front of the first record
	uint32_t e = line_start[l + 1];
	if (e > s && text[e - 1] == '\n')
		e--;
	if (e > s && text[e - 1] == '\r')
		e--;
	uint64_t pos = let_off[l];
	w_copy(reinterpret_cast<char *>(letters), pos, text + s, (uint64_t)(e - s), [](uint8_t c) { return c != ' ' && c != '\t'; },
	       [](uint8_t c) { return c; });
}

struct DeviceFasta {
	DevBuf<unsigned char> d_letters;
	std::vector<uint32_t> let_off; // n_rec + 1 letter offsets
	// the names (first word of each header) behind one another, gathered on the device while the text is there: n_rec + 1
	// offsets and the bytes, on the device and on the host.  The batch keeps THESE, not the file: round 3 kept the file
	// mapped for the batch's life and read names from it on demand (a file rewritten in place changed them)
	DevBuf<unsigned char> d_names;
	DevBuf<uint32_t> d_name_at;
	std::vector<uint32_t> name_at;
	std::string names;
};

// names of the records into one blob: one lane per record (a name is a handful of bytes)
__global__ void k_fa_gather_names(const unsigned char *__restrict__ text, const uint32_t *__restrict__ rec_name_off,
				  const uint32_t *__restrict__ name_at, uint32_t n_rec, unsigned char *__restrict__ names)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_rec)
		return;
	const uint32_t a = name_at[r], n = name_at[r + 1] - a, s = rec_name_off[r];
	for (uint32_t k = 0; k < n; k++)
		names[a + k] = text[s + k];
}

static int u32_scan(const uint32_t *in, uint32_t *out, size_t n, bool inclusive)
{
	size_t bytes = 0;
	if (inclusive)
		PGX_HIP(rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::plus<uint32_t>()));
	else
		PGX_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, n, rocprim::plus<uint32_t>()));
	DevBuf<uint8_t> tmp;
	PGX_TRY(tmp.alloc(bytes ? bytes : 1));
	if (inclusive)
		PGX_HIP(rocprim::inclusive_scan(tmp.data(), bytes, in, out, n, rocprim::plus<uint32_t>()));
	else
		PGX_HIP(rocprim::exclusive_scan(tmp.data(), bytes, in, out, 0u, n, rocprim::plus<uint32_t>()));
	return 0;
}

static int fasta_split_device(const char *text, size_t n_bytes, DeviceFasta &out)
{
	const uint32_t n = (uint32_t)n_bytes;
	out.let_off.assign(1, 0);
	if (n == 0)
		return out.d_letters.alloc(1, 0, 16);
	DevBuf<unsigned char> d_text;
	DevBuf<uint8_t> d_flag;
	DevBuf<uint32_t> d_idx;
	PGX_TRY(d_text.alloc(n, 0, 16)); // + padding: the kernels read aligned 16-byte words
	PGX_TRY(d_text.upload((const unsigned char *)text, n));
	PGX_TRY(d_flag.alloc(n));
	PGX_TRY(d_idx.alloc(n));
	const unsigned gb = (n + 255) / 256;
	hipLaunchKernelGGL(k_fa_line_flags, dim3(gb), dim3(256), 0, 0, d_text.data(), n, d_flag.data());
	{
		size_t bytes = 0;
		PGX_HIP(rocprim::exclusive_scan(nullptr, bytes, d_flag.data(), d_idx.data(), 0u, (size_t)n, rocprim::plus<uint32_t>()));
		DevBuf<uint8_t> tmp;
		PGX_TRY(tmp.alloc(bytes ? bytes : 1));
		PGX_HIP(rocprim::exclusive_scan(tmp.data(), bytes, d_flag.data(), d_idx.data(), 0u, (size_t)n, rocprim::plus<uint32_t>()));
	}
	uint32_t last_idx = 0;
	uint8_t last_flag = 0;
	PGX_TRY(d_idx.download(&last_idx, 1, n - 1));
	PGX_TRY(d_flag.download(&last_flag, 1, n - 1));
	const uint32_t n_lines = last_idx + last_flag;
	DevBuf<uint32_t> d_line_start, d_hdr, d_nlet, d_name_len, d_rec_incl, d_let_off;
	PGX_TRY(d_line_start.alloc((size_t)n_lines + 1));
	PGX_TRY(d_hdr.alloc((size_t)n_lines + 1, 0, 0, true));
	PGX_TRY(d_nlet.alloc((size_t)n_lines + 1, 0, 0, true));
	PGX_TRY(d_name_len.alloc((size_t)n_lines + 1));
	PGX_TRY(d_rec_incl.alloc((size_t)n_lines + 1));
	PGX_TRY(d_let_off.alloc((size_t)n_lines + 1));
	hipLaunchKernelGGL(k_fa_line_starts, dim3(gb), dim3(256), 0, 0, d_flag.data(), d_idx.data(), n, d_line_start.data());
	PGX_HIP(hipMemcpy(d_line_start.data() + n_lines, &n, sizeof n, hipMemcpyHostToDevice)); // sentinel: end of the last line
	d_flag.release();
	d_idx.release();
	const unsigned gl = (n_lines + 255) / 256;
	hipLaunchKernelGGL(k_fa_line_info, dim3(gl), dim3(256), 0, 0, d_text.data(), d_line_start.data(), n_lines, d_hdr.data(),
			   d_nlet.data(), d_name_len.data());
	PGX_HIP(hipGetLastError());
	PGX_TRY(u32_scan(d_hdr.data(), d_rec_incl.data(), (size_t)n_lines + 1, true));
	hipLaunchKernelGGL(k_fa_drop_preamble, dim3(gl), dim3(256), 0, 0, d_rec_incl.data(), n_lines, d_nlet.data());
	PGX_TRY(u32_scan(d_nlet.data(), d_let_off.data(), (size_t)n_lines + 1, false));
	uint32_t n_rec = 0, n_let = 0;
	PGX_TRY(d_rec_incl.download(&n_rec, 1, n_lines));
	PGX_TRY(d_let_off.download(&n_let, 1, n_lines));
	DevBuf<uint32_t> d_rlo, d_rno, d_rnl;
	PGX_TRY(d_rlo.alloc((size_t)n_rec + 1));
	PGX_TRY(d_rno.alloc((size_t)n_rec + 1));
	PGX_TRY(d_rnl.alloc((size_t)n_rec + 1));
	PGX_TRY(out.d_letters.alloc(n_let ? n_let : 1, 0, 16));
	hipLaunchKernelGGL(k_fa_copy_letters, dim3((unsigned)(((uint64_t)n_lines * kGroup + 255) / 256)), dim3(256), 0, 0, d_text.data(), d_line_start.data(), d_hdr.data(), d_nlet.data(),
			   d_let_off.data(), d_rec_incl.data(), d_name_len.data(), n_lines, out.d_letters.data(), d_rlo.data(), d_rno.data(),
			   d_rnl.data());
	PGX_HIP(hipGetLastError());
	out.let_off.resize((size_t)n_rec + 1);
	PGX_TRY(d_rlo.download(out.let_off.data(), n_rec));
	out.let_off[n_rec] = n_let;
	// the names, compact
	PGX_HIP(hipMemsetAsync(d_rnl.data() + n_rec, 0, sizeof(uint32_t), 0));
	PGX_TRY(out.d_name_at.alloc((size_t)n_rec + 1));
	PGX_TRY(u32_scan(d_rnl.data(), out.d_name_at.data(), (size_t)n_rec + 1, false));
	out.name_at.resize((size_t)n_rec + 1);
	PGX_TRY(out.d_name_at.download(out.name_at.data(), (size_t)n_rec + 1));
	const uint32_t n_name_bytes = out.name_at[n_rec];
	PGX_TRY(out.d_names.alloc(n_name_bytes ? n_name_bytes : 1, 0, 16));
	if (n_rec)
		hipLaunchKernelGGL(k_fa_gather_names, dim3((n_rec + 255) / 256), dim3(256), 0, 0, d_text.data(), d_rno.data(), out.d_name_at.data(), n_rec,
				   out.d_names.data());
	PGX_HIP(hipGetLastError());
	out.names.resize(n_name_bytes);
	if (n_name_bytes)
		PGX_TRY(out.d_names.download((unsigned char *)&out.names[0], n_name_bytes));
	return 0;
}

// ------------------------------------------------------------------------------------------ pieces of reads with long N runs
// Spec S3: a letter that is not A/C/G/T matches nothing, so every alignment pays at least 2 for each such letter of
// the read it spans; the gapped extension (X = 54, compared with the best score 19 differences earlier) abandons every
// path inside a run of kSplitRun = 28 or more of them (2 * 28 > 54, 28 >= 19), the ungapped one (X = 10) long before,
// and no seed overlaps one: no hit holds a letter of such a run.  The stretches between such runs can therefore be searched as reads of their own (most
// of them free of ambiguity letters: the plain kernels, the small flag classes) and their hits put back per read
// with the query coordinates shifted (search_pipeline).  This is the shape trim2 -g hands over: two mates joined
// by 100 or 189 N's (Trim/trim2.4.pl:228-245, :502-505).  Stretches shorter than a seed word hold no hit and are dropped.
constexpr uint32_t kSplitRun = 28, kMinPiece = 28;

// f(start, length) for every stretch of read letters [s, s + L) kept by the rule above, left to right
template <typename F> __device__ __forceinline__ void for_pieces(const unsigned char *__restrict__ letters, uint64_t s, uint64_t L, F f)
{
	uint64_t st = 0, run = 0; // st: start of the current stretch; run: unknown letters seen just before position i
	auto known_at = [&](uint64_t i) { // a known letter, or the end of the read, at position i
		if (run >= kSplitRun) { // the stretch ended where the run began
			const uint64_t en = i - run;
			if (en - st >= kMinPiece)
				f(st, en - st);
			st = i;
		}
		run = 0;
	};
	for_bytes_at(letters + s, L, [&](uint8_t c, uint64_t i) {
		if (letter_code(c) >= 4)
			run++;
		else
			known_at(i);
	});
	known_at(L);
	if (L - st >= kMinPiece && st < L)
		f(st, L - st);
}

// pieces per read; reads with fewer than kSplitRun unknown letters are one piece without a look at their letters
__global__ void k_piece_count(const unsigned char *__restrict__ letters, const uint64_t *__restrict__ off, const uint32_t *__restrict__ namb,
			      uint64_t n_reads, uint32_t *__restrict__ n_pieces, unsigned int *__restrict__ any_split)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_reads)
		return;
	if (r == n_reads) {
		n_pieces[r] = 0;
		return;
	}
	uint32_t c = 1;
	if (namb[r] >= kSplitRun) {
		const uint64_t s = off[r], L = off[r + 1] - s;
		c = 0;
		bool whole = false;
		for_pieces(letters, s, L, [&](uint64_t st, uint64_t n) {
			c++;
			whole = st == 0 && n == L;
		});
		if (!(c == 1 && whole))
			atomicOr(any_split, 1u);
	}
	n_pieces[r] = c;
}

__global__ void k_piece_fill(const unsigned char *__restrict__ letters, const uint64_t *__restrict__ off, const uint32_t *__restrict__ namb,
			     uint64_t n_reads, const uint32_t *__restrict__ first, uint64_t *__restrict__ p_start, uint32_t *__restrict__ p_len,
			     uint32_t *__restrict__ p_parent, uint32_t *__restrict__ p_qoff)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads)
		return;
	const uint64_t s = off[r], L = off[r + 1] - s;
	uint32_t k = first[r];
	auto put = [&](uint64_t st, uint64_t n) {
		p_start[k] = s + st;
		p_len[k] = (uint32_t)n;
		p_parent[k] = (uint32_t)r;
		p_qoff[k] = (uint32_t)st;
		k++;
	};
	if (namb[r] >= kSplitRun)
		for_pieces(letters, s, L, put);
	else
		put(0, L);
}

static int reads_finish(pgx_reads *rd);

// builds rd->pieces (a batch of its own: packed strands, classes) when a read of the batch holds a long N run
static int reads_build_pieces(pgx_reads *rd, const unsigned char *d_letters, const uint64_t *d_loff, const uint32_t *d_namb)
{
	const uint64_t n = (uint64_t)rd->n;
	if (n == 0 || n >= 0x7FFFFFFFull || rd->max_len >= (1 << 30))
		return 0;
	DevBuf<uint32_t> d_pc, d_first;
	DevBuf<unsigned int> d_flag;
	PGX_TRY(d_pc.alloc(n + 1));
	PGX_TRY(d_flag.alloc(1, 0, 0, true));
	hipLaunchKernelGGL(k_piece_count, dim3((unsigned)((n + 1 + 127) / 128)), dim3(128), 0, 0, d_letters, d_loff, d_namb, n, d_pc.data(),
			   d_flag.data());
	PGX_HIP(hipGetLastError());
	unsigned int any = 0;
	PGX_TRY(d_flag.download(&any, 1));
	if (!any)
		return 0;
	PGX_TRY(d_first.alloc(n + 1));
	PGX_TRY(u32_scan(d_pc.data(), d_first.data(), (size_t)n + 1, false));
	uint32_t np = 0;
	PGX_TRY(d_first.download(&np, 1, n));
	std::unique_ptr<pgx_reads> pc(new pgx_reads());
	pc->n = np;
	DevBuf<uint64_t> d_pstart;
	DevBuf<uint32_t> d_plen;
	PGX_TRY(d_pstart.alloc((size_t)np + 1));
	PGX_TRY(d_plen.alloc((size_t)np + 1));
	PGX_TRY(rd->d_piece_parent.alloc((size_t)np + 1));
	PGX_TRY(rd->d_piece_qoff.alloc((size_t)np + 1));
	hipLaunchKernelGGL(k_piece_fill, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, 0, d_letters, d_loff, d_namb, n, d_first.data(),
			   d_pstart.data(), d_plen.data(), rd->d_piece_parent.data(), rd->d_piece_qoff.data());
	PGX_HIP(hipGetLastError());
	pc->h_len.resize(np);
	PGX_TRY(d_plen.download(pc->h_len.data(), np));
	pc->h_woff.resize((size_t)np + 1);
	uint64_t nw = 0;
	for (uint32_t k = 0; k < np; k++) {
		nw = place_read_words(nw, (pc->h_len[k] + 31) / 32);
		pc->h_woff[k] = (uint32_t)nw;
		nw += (pc->h_len[k] + 31) / 32;
		pc->max_len = std::max<int32_t>(pc->max_len, (int32_t)pc->h_len[k]);
	}
	if (nw >= 0xFFFFFFFFull)
		return 0; // keep the batch unsplit rather than overflow the 32-bit word offsets
	pc->h_woff[np] = (uint32_t)nw;
	pc->n_words = (int64_t)nw;
	DevBuf<uint32_t> d_pnamb;
	DevBuf<unsigned int> d_pflag;
	PGX_TRY(pc->d_woff.alloc((size_t)np + 1));
	PGX_TRY(pc->d_woff.upload(pc->h_woff.data(), (size_t)np + 1));
	PGX_TRY(pc->d_fwd.alloc((size_t)nw + 24, 0, 0, true));
	PGX_TRY(pc->d_fwd_amb.alloc((size_t)nw + 24, 0, 0, true));
	PGX_TRY(d_pnamb.alloc(np ? np : 1));
	PGX_TRY(d_pflag.alloc(1, 0, 0, true));
	if (np) {
		hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)((np + 127) / 128)), dim3(128), 0, 0, d_letters, d_pstart.data(), pc->d_woff.data(),
				   (uint64_t)np, 0, pc->d_fwd.data(), pc->d_fwd_amb.data(), d_pnamb.data(), d_pflag.data(), d_plen.data());
		PGX_HIP(hipGetLastError());
	}
	unsigned int pflag = 0;
	PGX_TRY(d_pflag.download(&pflag, 1));
	pc->has_amb = pflag != 0;
	if (!pc->has_amb) {
		pc->d_fwd_amb.release();
	} else {
		std::vector<uint32_t> na(np);
		PGX_TRY(d_pnamb.download(na.data(), np));
		pc->h_read_amb.resize(np);
		for (uint32_t k = 0; k < np; k++)
			pc->h_read_amb[k] = na[k] != 0;
	}
	PGX_TRY(reads_finish(pc.get()));
	rd->d_piece_first.release();
	PGX_TRY(rd->d_piece_first.alloc(n + 1));
	PGX_HIP(hipMemcpy(rd->d_piece_first.data(), d_first.data(), (n + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice));
	rd->pieces = std::move(pc);
	return 0;
}

int reads_from_fasta_ex(const char *path, int64_t first, int64_t count, bool fold_to_g, std::vector<uint32_t> *amb_count,
			pgx_reads **out)
{
	if (!path || !out)
		return fail(PGX_E_ARG, "pgx_reads_from_fasta: null argument");
	PGX_TRY(require_device());
	const bool trace = getenv("PGX_TRACE") != nullptr;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
		return std::chrono::duration<double, std::milli>(b - a).count();
	};
	const auto t_begin = now();
	bool ok;
	std::shared_ptr<const TextBlob> text = TextBlob::from_file(path, &ok);
	if (!ok)
		return fail(PGX_E_IO, "cannot open query file %s", path);
	if (trace)
		fprintf(stderr, "[pgx trace] reads_from_fasta: file %.1f ms\n", ms(t_begin, now()));
	return reads_from_fasta_text(std::move(text), first, count, fold_to_g, amb_count, out);
}

// the same for FASTA text already in memory (pgx_blastn_run streams large query files through this in pieces)
int reads_from_fasta_text(std::shared_ptr<const TextBlob> text_ptr, int64_t first, int64_t count, bool fold_to_g,
			  std::vector<uint32_t> *amb_count, pgx_reads **out)
{
	const TextBlob &text = *text_ptr;
	PGX_TRY(require_device());
	const bool trace = getenv("PGX_TRACE") != nullptr;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
		return std::chrono::duration<double, std::milli>(b - a).count();
	};
	const auto t_read = now();
	// records, letters and names: found on the device for files under 4 GiB, by the host splitter otherwise
	DeviceFasta df;
	std::vector<uint64_t> rec_off; // n_rec + 1 letter offsets
	pgx_reads *rd = new pgx_reads();
	FastaLetters fl;
	const bool on_device = text.size() < (1ull << 32) - 2;
	if (on_device) {
		int rc0 = fasta_split_device(text.data(), text.size(), df);
		if (rc0 < 0) {
			delete rd;
			return rc0;
		}
		rec_off.assign(df.let_off.begin(), df.let_off.end());
	} else {
		split_fasta_text(std::string(text.data(), text.size()), fl); // (4 GiB and more: the host splitter works on a string)
		rec_off = fl.off;
	}
	const auto t_split = now();
	int64_t total = (int64_t)rec_off.size() - 1;
	if (first < 0)
		first = 0;
	if (first > total)
		first = total;
	if (count < 0 || first + count > total)
		count = total - first;
	rd->n = count;
	rd->first = first;
	rd->h_len.resize((size_t)count);
	rd->h_woff.resize((size_t)count + 1);
	uint64_t nw = 0;
	rd->name_off.resize((size_t)count);
	rd->name_len.resize((size_t)count);
	std::string own_names; // host splitter: names are copied out of the header strings
	for (int64_t i = 0; i < count; i++) {
		uint64_t L = rec_off[(size_t)(first + i) + 1] - rec_off[(size_t)(first + i)];
		rd->h_len[(size_t)i] = (uint32_t)L;
		nw = place_read_words(nw, (L + 31) / 32);
		rd->h_woff[(size_t)i] = (uint32_t)nw;
		nw += (L + 31) / 32;
		if ((int32_t)L > rd->max_len)
			rd->max_len = (int32_t)L;
		if (on_device) {
			rd->name_off[(size_t)i] = df.name_at[(size_t)(first + i)] - df.name_at[(size_t)first];
			rd->name_len[(size_t)i] = df.name_at[(size_t)(first + i) + 1] - df.name_at[(size_t)(first + i)];
		} else {
			const std::string nm = first_word(fl.headers[(size_t)(first + i)]);
			rd->name_off[(size_t)i] = own_names.size();
			rd->name_len[(size_t)i] = (uint32_t)nm.size();
			own_names += nm;
		}
	}
	rd->h_woff[(size_t)count] = (uint32_t)nw;
	rd->n_words = (int64_t)nw;
	// letters of the selected block are packed by a kernel
	const uint64_t l0 = rec_off[(size_t)first], l1 = rec_off[(size_t)(first + count)];
	std::vector<uint64_t> loff((size_t)count + 1);
	for (int64_t i = 0; i <= count; i++)
		loff[(size_t)i] = rec_off[(size_t)(first + i)] - l0;
	DevBuf<unsigned char> d_letters_host;
	int rc = 0;
	const unsigned char *d_letters_ptr = nullptr;
	if (on_device) {
		d_letters_ptr = df.d_letters.data() + l0;
		// the batch's own copy of its names (host and device); the file's text is let go when this call returns
		const uint32_t nb0 = df.name_at[(size_t)first], nb1 = df.name_at[(size_t)(first + count)];
		rd->h_text = std::make_shared<const TextBlob>(df.names.substr(nb0, nb1 - nb0));
		rc = rd->d_names.alloc(nb1 - nb0 ? nb1 - nb0 : 1, 0, 16);
		if (rc == 0 && nb1 > nb0 && hipMemcpy(rd->d_names.data(), df.d_names.data() + nb0, nb1 - nb0, hipMemcpyDeviceToDevice) != hipSuccess)
			rc = fail(PGX_E_NODEVICE, "copy of the read names failed");
	} else {
		rd->h_text = std::make_shared<const TextBlob>(std::move(own_names));
		rc = rd->d_names.alloc(rd->h_text->size() ? rd->h_text->size() : 1, 0, 16);
		if (rc == 0) rc = rd->d_names.upload((const unsigned char *)rd->h_text->data(), rd->h_text->size());
		rc = d_letters_host.alloc(l1 - l0 ? l1 - l0 : 1, 0, 16);
		if (rc == 0) rc = d_letters_host.upload((const unsigned char *)fl.letters.data() + l0, l1 - l0);
		d_letters_ptr = d_letters_host.data();
	}
	if (rc == 0) {
		std::vector<uint32_t> at((size_t)count + 1, 0);
		for (int64_t i = 0; i < count; i++)
			at[(size_t)i + 1] = at[(size_t)i] + rd->name_len[(size_t)i];
		rc = rd->d_name_at.alloc((size_t)count + 1);
		if (rc == 0) rc = rd->d_name_at.upload(at.data(), at.size());
	}
	DevBuf<uint64_t> d_loff;
	DevBuf<uint32_t> d_namb;
	DevBuf<unsigned int> d_flag;
	if (rc == 0) rc = d_loff.alloc((size_t)count + 1);
	if (rc == 0) rc = d_loff.upload(loff.data(), loff.size());
	if (rc == 0) rc = d_namb.alloc(count ? (size_t)count : 1);
	if (rc == 0) rc = d_flag.alloc(1, 0, 0, true);
	if (rc == 0) rc = rd->d_woff.alloc((size_t)count + 1);
	if (rc == 0) rc = rd->d_woff.upload(rd->h_woff.data(), (size_t)count + 1);
	if (rc == 0) rc = rd->d_fwd.alloc((size_t)nw + 24, 0, 0, true);
	if (rc == 0 && !fold_to_g) rc = rd->d_fwd_amb.alloc((size_t)nw + 24, 0, 0, true);
	if (rc == 0 && count > 0) {
		hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)((count + 127) / 128)), dim3(128), 0, 0, d_letters_ptr, d_loff.data(),
				   rd->d_woff.data(), (uint64_t)count, fold_to_g ? 1 : 0, rd->d_fwd.data(),
				   fold_to_g ? (uint64_t *)nullptr : rd->d_fwd_amb.data(), d_namb.data(), d_flag.data());
		if (hipGetLastError() != hipSuccess)
			rc = fail(PGX_E_NODEVICE, "k_pack_reads launch failed");
	}
	unsigned int flag = 0;
	if (rc == 0) rc = d_flag.download(&flag, 1);
	rd->has_amb = flag != 0 && !fold_to_g;
	if (rc == 0 && !rd->has_amb)
		rd->d_fwd_amb.release();
	if (rc == 0 && rd->has_amb) {
		// which reads carry an ambiguity letter: the search sends only those through the ambiguity-aware kernels
		std::vector<uint32_t> na((size_t)count);
		rc = d_namb.download(na.data(), (size_t)count);
		rd->h_read_amb.resize((size_t)count);
		for (int64_t i = 0; i < count; i++)
			rd->h_read_amb[(size_t)i] = na[(size_t)i] != 0;
	}
	if (rc == 0 && amb_count) {
		amb_count->assign((size_t)count, 0);
		rc = d_namb.download(amb_count->data(), (size_t)count);
	}
	if (rc == 0 && fold_to_g) {
		// SOAP row formatting echoes the read as aligned: keep a host copy of the packed forward strand
		rd->h_fwd.assign(nw + 2, 0);
		rc = rd->d_fwd.download(rd->h_fwd.data(), (size_t)nw);
	}
	if (rc == 0 && rd->has_amb && getenv("PGX_NO_PIECES") == nullptr)
		rc = reads_build_pieces(rd, d_letters_ptr, d_loff.data(), d_namb.data());
	const auto t_pack = now();
	if (rc == 0)
		rc = reads_finish(rd);
	if (rc < 0) {
		delete rd;
		return rc;
	}
	if (trace) {
		(void)hipDeviceSynchronize();
		fprintf(stderr, "[pgx trace] reads_from_fasta: split %.1f ms, tables+pack %.1f ms, strands %.1f ms\n", ms(t_read, t_split),
			ms(t_split, t_pack), ms(t_pack, now()));
	}
	*out = rd;
	return 0;
}

// a database whose ambiguity codes read as G (SOAP mode), built from a file database
int db_fold_amb_to_g(const pgx_db *src, pgx_db **out)
{
	pgx_db *db = new pgx_db();
	db->n_seq = src->n_seq;
	db->n_bases = src->n_bases;
	db->has_amb = false;
	db->h_seq_off = src->h_seq_off;
	db->ids = src->ids;
	db->h_words = src->h_words;
	if (src->has_amb)
		for (size_t w = 0; w < db->h_words.size() && w < src->h_amb.size(); w++)
			db->h_words[w] |= src->h_amb[w] << 1; // flagged bases hold code 0: setting the high bit makes them G
	int rc = db_upload_and_index(db);
	if (rc < 0) {
		delete db;
		return rc;
	}
	*out = db;
	return 0;
}

int db_read_host(const char *prefix, pgx_db **out) { return db_read_file(prefix, out); }

// number of FASTA records of a file (lines starting with '>'), without packing anything
// records (lines that start with '>') in a piece of FASTA text; `at_line_start` carries over between pieces
int64_t fasta_count_records_text(const char *base, size_t len, bool *at_line_start)
{
	int64_t n = 0;
	size_t i = 0;
	bool bol = *at_line_start;
	while (i < len) {
		if (bol && base[i] == '>')
			n++;
		const char *nl = (const char *)memchr(base + i, '\n', len - i);
		if (!nl) {
			bol = false;
			i = len;
			break;
		}
		i = (size_t)(nl - base) + 1;
		bol = true;
	}
	*at_line_start = bol;
	return n;
}

int64_t fasta_count_records(const char *path)
{
	FILE *f = fopen(path, "rb");
	if (!f)
		return -1;
	std::vector<char> buf(8u << 20);
	int64_t n = 0;
	bool bol = true;
	size_t k;
	while ((k = fread(buf.data(), 1, buf.size(), f)) > 0)
		n += fasta_count_records_text(buf.data(), k, &bol);
	fclose(f);
	return n;
}

} // namespace pgx

std::string pgx_reads::name_of(int64_t i) const
{
	if (synthetic)
		return "r" + std::to_string(first + i);
	return std::string(h_text->data() + name_off[(size_t)i], name_len[(size_t)i]);
}

namespace pgx {
template <typename T> __global__ __launch_bounds__(256) void k_wrapping_sum(const T *__restrict__ a, uint64_t n, unsigned long long *__restrict__ out)
{
	unsigned long long acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
		acc += (unsigned long long)a[i] * (2ull * i + 1ull); // (position-weighted: a permuted array sums differently)
	for (int sh = 32; sh >= 1; sh >>= 1)
		acc += __shfl_xor(acc, sh);
	if ((threadIdx.x & 63) == 0)
		atomicAdd(out, acc);
}
} // namespace pgx

using namespace pgx;

extern "C" {

int pgx_db_build(const char *fasta_path, const char *prefix)
{
	if (!fasta_path || !prefix)
		return fail(PGX_E_ARG, "pgx_db_build: null argument");
	bool ok;
	std::string text = read_text_file(fasta_path, &ok);
	if (!ok)
		return fail(PGX_E_IO, "cannot open FASTA file %s", fasta_path);
	PackedSet ps;
	pack_fasta_text(text, ps);
	pgx_db *db = nullptr;
	db_from_packed(ps, &db);
	int rc = db_write_file(db, prefix);
	delete db;
	return rc;
}

int pgx_db_open(const char *prefix, pgx_db **out)
{
	if (!prefix || !out)
		return fail(PGX_E_ARG, "pgx_db_open: null argument");
	PGX_TRY(require_device());
	return pgx::guard("pgx_db_open", [&]() -> int {
		pgx_db *db = nullptr;
		PGX_TRY(db_read_file(prefix, &db));
		int rc = db_upload_and_index(db);
		if (rc < 0) {
			delete db;
			return rc;
		}
		*out = db;
		return 0;
	});
}

int pgx_db_from_fasta(const char *fasta_path, pgx_db **out)
{
	if (!fasta_path || !out)
		return fail(PGX_E_ARG, "pgx_db_from_fasta: null argument");
	PGX_TRY(require_device());
	bool ok;
	std::string text = read_text_file(fasta_path, &ok);
	if (!ok)
		return fail(PGX_E_IO, "cannot open FASTA file %s", fasta_path);
	PackedSet ps;
	pack_fasta_text(text, ps);
	pgx_db *db = nullptr;
	db_from_packed(ps, &db);
	int rc = db_upload_and_index(db);
	if (rc < 0) {
		delete db;
		return rc;
	}
	*out = db;
	return 0;
}

void pgx_db_close(pgx_db *db) { delete db; }
int64_t pgx_db_num_seqs(const pgx_db *db) { return db ? db->n_seq : 0; }
int64_t pgx_db_num_bases(const pgx_db *db) { return db ? db->n_bases : 0; }
const char *pgx_db_seq_id(const pgx_db *db, int64_t i)
{
	if (!db || i < 0 || i >= db->n_seq)
		return nullptr;
	return db->ids[(size_t)i].c_str();
}

void pgx_synth_default(pgx_synth_cfg *c)
{
	c->seed = 0x50414E47ull;
	c->n_seq = 666667;
	c->seq_len = 1500;
	c->n_genus = 20000;
	c->read_seed = 42;
	c->read_len = 150;
}

int pgx_db_from_synth(const pgx_synth_cfg *cfg, pgx_db **out)
{
	if (!cfg || !out || cfg->n_seq <= 0 || cfg->seq_len <= 0 || cfg->n_genus <= 0)
		return fail(PGX_E_ARG, "pgx_db_from_synth: bad configuration");
	PGX_TRY(require_device());
	pgx_db *db = new pgx_db();
	db->n_seq = cfg->n_seq;
	db->n_bases = cfg->n_seq * (int64_t)cfg->seq_len;
	if (db->n_bases >= (1ll << 32) - 64) {
		delete db;
		return fail(PGX_E_LIMIT, "synthetic database exceeds the 32-bit position limit");
	}
	db->h_seq_off.resize((size_t)db->n_seq + 1);
	for (int64_t i = 0; i <= db->n_seq; i++)
		db->h_seq_off[(size_t)i] = (uint32_t)(i * cfg->seq_len);
	db->synthetic_ids = true;
	synth_ids(db);
	size_t nw = ((size_t)db->n_bases + 31) / 32;
	int rc = db->d_words.alloc(nw, 24, 24, true);
	if (rc == 0) {
		hipLaunchKernelGGL(k_synth_db, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, 0, cfg->seed,
				   (uint64_t)cfg->n_seq, (uint32_t)cfg->seq_len, (uint64_t)cfg->n_genus,
				   db->d_words.data(), (uint64_t)nw, (uint64_t)db->n_bases);
		if (hipGetLastError() != hipSuccess)
			rc = fail(PGX_E_NODEVICE, "k_synth_db launch failed");
	}
	if (rc == 0)
		rc = db_upload_offsets(db);
	if (rc == 0)
		rc = db_build_index(db);
	if (rc < 0) {
		delete db;
		return rc;
	}
	*out = db;
	return 0;
}

int pgx_reads_from_synth(const pgx_synth_cfg *cfg, int64_t first, int64_t count, pgx_reads **out)
{
	if (!cfg || !out || count < 0 || cfg->read_len <= 0 || cfg->read_len > cfg->seq_len)
		return fail(PGX_E_ARG, "pgx_reads_from_synth: bad configuration");
	PGX_TRY(require_device());
	pgx_reads *rd = new pgx_reads();
	rd->n = count;
	rd->first = first;
	rd->synthetic = true;
	uint32_t wpr = ((uint32_t)cfg->read_len + 31) / 32;
	// (equal reads: a fixed stride with the property of place_read_words)
	const uint32_t stride = wpr <= 1 ? 1 : wpr <= 2 ? 2 : wpr <= 4 ? 4 : (wpr + 7u) & ~7u;
	if ((uint64_t)count * stride >= 0xFFFFFFFFull)
		return fail(PGX_E_LIMIT, "pgx_reads_from_synth: more than 2^32 words in one batch");
	rd->n_words = count * (int64_t)stride;
	rd->max_len = cfg->read_len;
	rd->h_len.assign((size_t)count, (uint32_t)cfg->read_len);
	rd->h_woff.resize((size_t)count + 1);
	for (int64_t i = 0; i <= count; i++)
		rd->h_woff[(size_t)i] = (uint32_t)(i * stride);
	int rc = rd->d_fwd.alloc((size_t)rd->n_words + 24, 0, 0, true);
	if (rc == 0 && count > 0) {
		uint64_t nt = (uint64_t)count * wpr;
		hipLaunchKernelGGL(k_synth_reads, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, 0, cfg->seed,
				   (uint64_t)cfg->n_seq, (uint32_t)cfg->seq_len, (uint64_t)cfg->n_genus, cfg->read_seed,
				   (uint32_t)cfg->read_len, (uint64_t)first, (uint64_t)count, wpr, stride, rd->d_fwd.data());
		if (hipGetLastError() != hipSuccess)
			rc = fail(PGX_E_NODEVICE, "k_synth_reads launch failed");
	}
	if (rc == 0)
		rc = reads_finish(rd);
	if (rc < 0) {
		delete rd;
		return rc;
	}
	trace_point("reads_from_synth");
	*out = rd;
	return 0;
}

int pgx_reads_from_fasta(const char *path, int64_t first, int64_t count, pgx_reads **out)
{
	return pgx::reads_from_fasta_ex(path, first, count, false, nullptr, out);
}

int pgx_reads_from_fasta_text(const char *text, size_t len, int64_t first, int64_t count, pgx_reads **out)
{
	if ((!text && len) || !out)
		return fail(PGX_E_ARG, "pgx_reads_from_fasta_text: null argument");
	return pgx::reads_from_fasta_text(std::make_shared<const pgx::TextBlob>(std::string(text ? text : "", len)), first, count, false, nullptr, out);
}

// A batch back as FASTA text (">name" + one sequence line per read): the hand-over file between Trim and Classify
// (README.md:34 -> :96) for batches that were made on the device, and the input of bench.py's file-to-file line.
int pgx_reads_write_fasta(const pgx_reads *r, const char *path)
{
	if (!r || !path)
		return fail(PGX_E_ARG, "pgx_reads_write_fasta: null argument");
	return pgx::guard("pgx_reads_write_fasta", [&]() -> int {
		std::vector<uint64_t> w((size_t)r->n_words + 1, 0), a;
		PGX_TRY(r->d_fwd.download(w.data(), (size_t)r->n_words));
		if (r->has_amb) {
			a.assign((size_t)r->n_words + 1, 0);
			PGX_TRY(r->d_fwd_amb.download(a.data(), (size_t)r->n_words));
		}
		FILE *f = fopen(path, "wb");
		if (!f)
			return fail(PGX_E_IO, "cannot open %s for writing", path);
		std::string out;
		out.reserve(64u << 20);
		bool ok = true;
		for (int64_t i = 0; i < r->n && ok; i++) {
			out += '>';
			out += r->name_of(i);
			out += '\n';
			const uint32_t L = r->h_len[(size_t)i], w0 = r->h_woff[(size_t)i];
			const size_t at = out.size();
			out.resize(at + L + 1);
			for (uint32_t k = 0; k < L; k++) {
				const uint64_t word = w[w0 + (k >> 5)];
				char c = "ACGT"[(word >> (2 * (k & 31))) & 3];
				if (!a.empty() && ((a[w0 + (k >> 5)] >> (2 * (k & 31))) & 1))
					c = 'N';
				out[at + k] = c;
			}
			out[at + L] = '\n';
			if (out.size() > (60u << 20)) {
				ok = fwrite(out.data(), 1, out.size(), f) == out.size();
				out.clear();
			}
		}
		ok = ok && fwrite(out.data(), 1, out.size(), f) == out.size();
		if (fclose(f) != 0 || !ok)
			return fail(PGX_E_IO, "short write to %s", path);
		return 0;
	});
}

void pgx_reads_close(pgx_reads *r) { delete r; }
int64_t pgx_reads_count(const pgx_reads *r) { return r ? r->n : 0; }

int pgx_reads_get(const pgx_reads *r, int64_t i, uint8_t *bases_out, int32_t cap, int32_t *len_out)
{
	if (!r || i < 0 || i >= r->n || !bases_out)
		return fail(PGX_E_ARG, "pgx_reads_get: bad argument");
	uint32_t L = r->h_len[(size_t)i], w0 = r->h_woff[(size_t)i], nw = (L + 31) / 32;
	std::vector<uint64_t> w(nw), a(nw, 0);
	PGX_TRY(r->d_fwd.download(w.data(), nw, w0));
	if (r->has_amb)
		PGX_TRY(r->d_fwd_amb.download(a.data(), nw, w0));
	for (uint32_t k = 0; k < L && (int32_t)k < cap; k++) {
		uint8_t b = (uint8_t)((w[k >> 5] >> (2 * (k & 31))) & 3);
		if ((a[k >> 5] >> (2 * (k & 31))) & 1)
			b = 4;
		bases_out[k] = b;
	}
	if (len_out)
		*len_out = (int32_t)L;
	return 0;
}

int pgx_db_get_shape(const pgx_db *db, pgx_db_shape *out)
{
	if (!db || !out)
		return fail(PGX_E_ARG, "pgx_db_get_shape: null argument");
	out->n_seq = db->n_seq;
	out->n_bases = db->n_bases;
	out->has_amb = db->has_amb;
	out->index_bits = db->index_bits;
	out->n_postings = db->n_postings;
	out->synthetic_ids = db->synthetic_ids;
	return 0;
}

// What a receiving rank gets over RCCL: by default only the primary data (packed bases, ambiguity flags, sequence
// offsets: 0.25 GB at 1 Gbp); block tables and the seed index (33 GB) are rebuilt locally by pgx_db_finish_import in
// the time a broadcast of them would take one xGMI hop.  PGX_BCAST_INDEX=1 ships the built index instead (measurement).
static bool bcast_whole_index()
{
	const char *e = getenv("PGX_BCAST_INDEX");
	return e && atoi(e) != 0;
}

int pgx_db_device_arrays(pgx_db *db, pgx_device_array *out, int cap)
{
	if (!db || !out)
		return fail(PGX_E_ARG, "pgx_db_device_arrays: null argument");
	int n = 0;
	auto add = [&](const char *name, void *p, size_t bytes) {
		if (n < cap) {
			out[n].name = name;
			out[n].ptr = p;
			out[n].bytes = bytes;
		}
		n++;
	};
	add("words", db->d_words.data(), db->d_words.bytes());
	if (db->has_amb)
		add("amb", db->d_amb.data(), db->d_amb.bytes());
	add("seq_off", db->d_seq_off.data(), db->d_seq_off.bytes());
	if (bcast_whole_index()) {
		add("blk_subj", db->d_blk_subj.data(), db->d_blk_subj.bytes());
		add("bucket_off", db->d_bucket_off.data(), db->d_bucket_off.bytes());
		add("postings", db->d_postings.data(), db->d_postings.bytes());
		add("post_ctx", db->d_post_ctx.data(), db->d_post_ctx.bytes());
	}
	return n;
}

int pgx_db_checksum(pgx_db *db, uint64_t out[4])
{
	if (!db || !out)
		return fail(PGX_E_ARG, "pgx_db_checksum: null argument");
	PGX_TRY(require_device());
	return guard("pgx_db_checksum", [&]() -> int {
		DevBuf<unsigned long long> d;
		PGX_TRY(d.alloc(4, 0, 0, true));
		const unsigned grid = 256 * 8;
		if (db->d_words.n)
			hipLaunchKernelGGL(k_wrapping_sum<uint64_t>, dim3(grid), dim3(256), 0, 0, db->d_words.data(), (uint64_t)db->d_words.n, d.data() + 0);
		if (db->d_seq_off.n)
			hipLaunchKernelGGL(k_wrapping_sum<uint32_t>, dim3(grid), dim3(256), 0, 0, db->d_seq_off.data(), (uint64_t)db->n_seq + 1, d.data() + 1);
		if (db->d_bucket_off.base && db->index_bits > 0)
			hipLaunchKernelGGL(k_wrapping_sum<uint32_t>, dim3(grid), dim3(256), 0, 0, db->d_bucket_off.data(), ((uint64_t)1 << db->index_bits) + 1,
					   d.data() + 2);
		if (db->d_postings.base && db->n_postings > 0)
			hipLaunchKernelGGL(k_wrapping_sum<uint32_t>, dim3(grid), dim3(256), 0, 0, db->d_postings.data(), (uint64_t)db->n_postings, d.data() + 3);
		PGX_HIP(hipGetLastError());
		unsigned long long h[4];
		PGX_TRY(d.download(h, 4));
		for (int i = 0; i < 4; i++)
			out[i] = h[i];
		return 0;
	});
}

int pgx_db_alloc_like(const pgx_db_shape *s, pgx_db **out)
{
	if (!s || !out)
		return fail(PGX_E_ARG, "pgx_db_alloc_like: null argument");
	PGX_TRY(require_device());
	pgx_db *db = new pgx_db();
	db->n_seq = s->n_seq;
	db->n_bases = s->n_bases;
	db->has_amb = s->has_amb != 0;
	db->index_bits = s->index_bits;
	db->n_postings = s->n_postings;
	db->synthetic_ids = s->synthetic_ids != 0;
	size_t nw = ((size_t)db->n_bases + 31) / 32;
	int rc = db->d_words.alloc(nw, 24, 24, true);
	if (rc == 0 && db->has_amb)
		rc = db->d_amb.alloc(nw, 24, 24, true);
	if (rc == 0)
		rc = db->d_seq_off.alloc((size_t)db->n_seq + 1);
	if (bcast_whole_index()) {
		if (rc == 0)
			rc = db->d_blk_subj.alloc(((size_t)db->n_bases >> kBlkShift) + 2);
		if (rc == 0)
			rc = db->d_bucket_off.alloc((1ull << db->index_bits) + 1);
		if (rc == 0)
			rc = db->d_postings.alloc(db->n_postings ? (size_t)db->n_postings : 1);
		if (rc == 0)
			rc = db->d_post_ctx.alloc(db->n_postings ? (size_t)db->n_postings : 1);
	}
	if (rc < 0) {
		delete db;
		return rc;
	}
	*out = db;
	return 0;
}

int pgx_db_finish_import(pgx_db *db)
{
	if (!db)
		return fail(PGX_E_ARG, "pgx_db_finish_import: null argument");
	db->h_seq_off.resize((size_t)db->n_seq + 1);
	PGX_TRY(db->d_seq_off.download(db->h_seq_off.data(), (size_t)db->n_seq + 1));
	if (db->h_seq_off[(size_t)db->n_seq] != (uint32_t)db->n_bases)
		return fail(PGX_E_FORMAT, "imported sequence offsets do not end at the database length");
	if (db->d_bucket_off.base) {
		PGX_TRY(db_build_blk_info(db)); // the index came with the broadcast: only the derived block records are made here
	} else {
		PGX_TRY(db_build_blocks(db));
		PGX_TRY(db_build_index(db));
	}
	if (db->synthetic_ids)
		pgx::synth_ids(db);
	else if (db->ids.empty())
		return fail(PGX_E_ARG, "imported database has no subject ids");
	return 0;
}
}
