// Reads with more hits than one wavefront orders in LDS (> 64), of any size: a conserved 16S region against a 16S
// collection gives a read tens of thousands of hits, so this path must not be quadratic.
//
// Spec S5 order = (best score of the subject desc, subject asc, score desc, qstart, qend, sstart, send, mismatch,
// gapopen).  It is produced by five stable segmented radix sorts (one segment per read), least significant key first:
//     0  send, mismatch, gap openings (gapped hits: send no longer follows from the other columns)
//     1  sstart, then strand (minus before plus: with equal qstart/qend/sstart that is `send` ascending)
//     2  qstart, qend
//     3  subject asc, score desc      -> the head of every subject run holds the subject's best score
//     4  best score desc
// After sort 3 the hits of a subject are in the S5 order of their HSPs: spec v2's duplicate-alignment rule S3c is
// applied there (a dropped hit gets the last place of the read in sort 4); then the 500-subject cut (subjects are
// contiguous in the order) and the write-back into the hit table.
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <mutex>

#include "engine.hpp"

namespace pgx {

constexpr uint32_t kFragmentedRead = kFragmented;

__global__ void k_big_lens(const uint32_t *__restrict__ off, const uint32_t *__restrict__ big_list, uint32_t n_big,
			   uint32_t *__restrict__ len)
{
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k < n_big) {
		const uint32_t r = big_list[k];
		len[k] = off[r + 1] - off[r];
	}
	if (k == n_big)
		len[k] = 0;
}

// one block per big read: its hits, from wherever the seed kernel left them, into the compact work table
__global__ __launch_bounds__(256) void k_big_gather(const pgx_hit *__restrict__ hits, const pgx_hit *__restrict__ scratch,
						     const uint32_t *__restrict__ read_start, const uint32_t *__restrict__ off,
						     const uint32_t *__restrict__ big_list, uint32_t n_big,
						     const uint32_t *__restrict__ seg_off, pgx_hit *__restrict__ work,
						     uint32_t *__restrict__ vals)
{
	for (uint32_t k = blockIdx.x; k < n_big; k += gridDim.x) {
		const uint32_t r = big_list[k], o = off[r], n = off[r + 1] - o, g0 = seg_off[k];
		const uint32_t st0 = read_start[r];
		const pgx_hit *src = st0 == kFragmentedRead ? hits + o : scratch + st0;
		for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
			work[g0 + j] = src[j];
			vals[g0 + j] = g0 + j;
		}
	}
}

template <int PASS>
__global__ void k_big_keys(const pgx_hit *__restrict__ work, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ best,
			   uint32_t total, unsigned long long *__restrict__ keys)
{
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= total)
		return;
	const uint32_t v = vals[g];
	const pgx_hit h = work[v];
	unsigned long long k;
	if (PASS == 0)
		k = ((unsigned long long)(uint32_t)h.send << 32) | ((unsigned long long)h.mismatch << 16) | (unsigned long long)h.gapopen;
	else if (PASS == 1)
		k = ((unsigned long long)(uint32_t)h.sstart << 1) | (h.send > h.sstart ? 1ull : 0ull);
	else if (PASS == 2)
		k = ((unsigned long long)(uint32_t)h.qstart << 32) | (unsigned long long)(uint32_t)h.qend;
	else if (PASS == 3)
		k = ((unsigned long long)(uint32_t)h.subject << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)h.score);
	else
		k = (unsigned long long)(0xFFFFFFFFu - best[v]);
	keys[g] = k;
}

// after pass 3 (subject asc, score desc inside a read): the head of a subject run carries the best score.  The head of
// every hit's run by a running maximum of "position if a run starts here" over the segment (a block scan per 256 hits with
// a carry): linear in the segment however long a run is -- walking back to the head per hit was quadratic in the run
__global__ __launch_bounds__(256) void k_big_best(const pgx_hit *__restrict__ work, const uint32_t *__restrict__ vals,
						   const uint32_t *__restrict__ seg_off, uint32_t n_big, uint32_t *__restrict__ best)
{
	__shared__ uint32_t s_head[256];
	__shared__ uint32_t s_carry;
	for (uint32_t k = blockIdx.x; k < n_big; k += gridDim.x) {
		const uint32_t g0 = seg_off[k], g1 = seg_off[k + 1];
		if (threadIdx.x == 0)
			s_carry = g0;
		__syncthreads();
		for (uint32_t c0 = g0; c0 < g1; c0 += 256) {
			const uint32_t g = c0 + threadIdx.x;
			uint32_t v = 0, head = 0;
			if (g < g1) {
				v = vals[g];
				const bool starts = g == g0 || work[vals[g - 1]].subject != work[v].subject;
				head = starts ? g + 1 : 0u; // (0: no run starts here)
			}
			s_head[threadIdx.x] = head;
			__syncthreads();
			for (int d = 1; d < 256; d <<= 1) {
				const uint32_t o = (int)threadIdx.x >= d ? s_head[threadIdx.x - d] : 0u;
				__syncthreads();
				if (o > s_head[threadIdx.x])
					s_head[threadIdx.x] = o;
				__syncthreads();
			}
			const uint32_t carry = s_carry;
			const uint32_t h = s_head[threadIdx.x] ? s_head[threadIdx.x] - 1 : carry;
			if (g < g1)
				best[v] = (uint32_t)work[vals[h]].score;
			__syncthreads();
			if (threadIdx.x == 255)
				s_carry = h;
			__syncthreads();
		}
	}
}

// Spec v2, S3c inside one subject run (hits in S5 order): a hit is dropped when an EARLIER hit of the run, on the same
// strand, starts at the same point, ends at the same point, or holds it.  Dropped hits sort behind every kept one.
// (The rule is pairwise by definition: the walk is over the earlier hits of the SAME (read, subject) run only and stops at
// the first drop -- quadratic in the hits one read has on one subject, not in the read's hits.)
__global__ __launch_bounds__(256) void k_big_dedup(const pgx_hit *__restrict__ work, const uint32_t *__restrict__ vals,
						    const uint32_t *__restrict__ seg_off, uint32_t n_big, uint32_t *__restrict__ best,
						    uint32_t *__restrict__ seg_drop)
{
	for (uint32_t k = blockIdx.x; k < n_big; k += gridDim.x) {
		const uint32_t g0 = seg_off[k], g1 = seg_off[k + 1];
		for (uint32_t g = g0 + threadIdx.x; g < g1; g += blockDim.x) {
			const uint32_t v = vals[g];
			const pgx_hit a = work[v];
			const bool am = a.sstart > a.send;
			const int as0 = am ? a.send : a.sstart, as1 = am ? a.sstart : a.send;
			bool drop = false;
			for (uint32_t h = g; h > g0 && !drop;) {
				h--;
				const pgx_hit b = work[vals[h]];
				if (b.subject != a.subject)
					break;
				const bool bm = b.sstart > b.send;
				if (bm != am)
					continue;
				const int bs0 = bm ? b.send : b.sstart, bs1 = bm ? b.sstart : b.send;
				drop = (a.qstart == b.qstart && a.sstart == b.sstart) || (a.qend == b.qend && a.send == b.send) ||
				       (a.qstart >= b.qstart && a.qend <= b.qend && as0 >= bs0 && as1 <= bs1);
			}
			if (drop) {
				best[v] = 0u; // sort 4 places it behind the kept hits of the read
				atomicAdd(&seg_drop[k], 1u);
			}
		}
	}
}

// final order back into the hit table; read_cnt = hits kept by the 500-subject limit (spec S5)
__global__ __launch_bounds__(256) void k_big_write(const pgx_hit *__restrict__ work, const uint32_t *__restrict__ vals,
						    const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ off,
						    const uint32_t *__restrict__ big_list, uint32_t n_big, pgx_hit *__restrict__ hits,
						    uint32_t *__restrict__ read_cnt, const uint32_t *__restrict__ seg_drop)
{
	__shared__ uint32_t s_warp[4];
	__shared__ uint32_t s_carry, s_keep;
	for (uint32_t k = blockIdx.x; k < n_big; k += gridDim.x) {
		const uint32_t r = big_list[k], o = off[r], g0 = seg_off[k], n = seg_off[k + 1] - g0;
		const uint32_t n_kept = n - seg_drop[k]; // the dropped hits (S3c) are the last ones of the segment
		if (threadIdx.x == 0) {
			s_carry = 0;
			s_keep = n_kept;
		}
		__syncthreads();
		for (uint32_t base = 0; base < n; base += blockDim.x) {
			const uint32_t j = base + threadIdx.x;
			int flag = 0;
			pgx_hit h;
			if (j < n) {
				h = work[vals[g0 + j]];
				hits[o + j] = h;
				flag = j < n_kept && (j == 0 || work[vals[g0 + j - 1]].subject != h.subject);
			}
			// running count of distinct subjects up to and including position j
			const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
			const unsigned long long m = __ballot(flag);
			const uint32_t in_wave = (uint32_t)__popcll(m & (lane == 63 ? ~0ull : ((2ull << lane) - 1)));
			if (lane == 63)
				s_warp[w] = (uint32_t)__popcll(m);
			__syncthreads();
			uint32_t before = s_carry;
			for (int q = 0; q < w; q++)
				before += s_warp[q];
			const uint32_t subjects = before + in_wave;
			if (j < n && flag && subjects == 501u)
				atomicMin(&s_keep, j);
			__syncthreads();
			if (threadIdx.x == 0)
				s_carry += s_warp[0] + s_warp[1] + s_warp[2] + s_warp[3];
			__syncthreads();
		}
		if (threadIdx.x == 0)
			read_cnt[r] = s_keep;
		__syncthreads();
	}
}

// hits: the ordered hit table (big reads' slots are rewritten); scratch/read_start: the seed kernel's table
int sort_big_reads(pgx_hit *hits, const pgx_hit *scratch, const uint32_t *read_start, const uint32_t *off, uint32_t *read_cnt,
		   const uint32_t *big_list, uint32_t n_big, bool gapped)
{
	if (n_big == 0)
		return 0;
	// the work tables below are shared by every handle of the process: one caller at a time (a rare path)
	static std::mutex mu;
	std::lock_guard<std::mutex> lock(mu);
	// work tables live across calls (grow-only): a batch with a handful of big reads would otherwise pay a dozen
	// hipMalloc / hipFree pairs (≈ 1.3 ms) for a few microseconds of sorting
	// (never freed: a static destructor would call hipFree after the runtime may be gone; released and re-made when the
	// process moves to another device)
	struct BigWork {
		DevBuf<uint32_t> seg, va, vb, best, seg_drop;
		DevBuf<pgx_hit> work;
		DevBuf<unsigned long long> ka, kb;
		DevBuf<uint8_t> scan_tmp, sort_tmp;
		int device = -1;
	};
	static BigWork *bw = nullptr;
	int dev = 0;
	PGX_HIP(hipGetDevice(&dev));
	if (bw && bw->device != dev) {
		delete bw; // tables of another device
		bw = nullptr;
	}
	if (!bw) {
		bw = new BigWork();
		bw->device = dev;
	}
	DevBuf<uint32_t> &seg = bw->seg, &va = bw->va, &vb = bw->vb, &best = bw->best, &seg_drop = bw->seg_drop;
	DevBuf<pgx_hit> &work = bw->work;
	DevBuf<unsigned long long> &ka = bw->ka, &kb = bw->kb;
	DevBuf<uint8_t> &scan_tmp = bw->scan_tmp, &sort_tmp = bw->sort_tmp;
	PGX_TRY(seg.ensure((size_t)n_big + 1));
	PGX_TRY(seg_drop.ensure((size_t)n_big + 1));
	PGX_HIP(hipMemsetAsync(seg_drop.data(), 0, ((size_t)n_big + 1) * sizeof(uint32_t), 0));
	hipLaunchKernelGGL(k_big_lens, dim3((n_big + 1 + 255) / 256), dim3(256), 0, 0, off, big_list, n_big, seg.data());
	PGX_HIP(hipGetLastError());
	{
		size_t bytes = 0;
		PGX_HIP(rocprim::exclusive_scan(nullptr, bytes, seg.data(), seg.data(), 0u, (size_t)n_big + 1, rocprim::plus<uint32_t>()));
		PGX_TRY(scan_tmp.ensure(bytes ? bytes : 1));
		PGX_HIP(rocprim::exclusive_scan(scan_tmp.data(), bytes, seg.data(), seg.data(), 0u, (size_t)n_big + 1, rocprim::plus<uint32_t>()));
	}
	uint32_t total = 0;
	PGX_TRY(seg.download(&total, 1, n_big));
	if (total == 0)
		return 0;
	PGX_TRY(work.ensure(total));
	PGX_TRY(va.ensure(total));
	PGX_TRY(vb.ensure(total));
	PGX_TRY(best.ensure(total));
	PGX_TRY(ka.ensure(total));
	PGX_TRY(kb.ensure(total));
	const unsigned seg_grid = std::min<uint32_t>(n_big, 256u * 16u);
	hipLaunchKernelGGL(k_big_gather, dim3(seg_grid), dim3(256), 0, 0, hits, scratch, read_start, off, big_list, n_big, seg.data(),
			   work.data(), va.data());
	PGX_HIP(hipGetLastError());
	size_t sort_bytes = 0;
	PGX_HIP(rocprim::segmented_radix_sort_pairs(nullptr, sort_bytes, ka.data(), kb.data(), va.data(), vb.data(), total, n_big,
						    seg.data(), seg.data() + 1, 0, 64));
	PGX_TRY(sort_tmp.ensure(sort_bytes ? sort_bytes : 1));
	const unsigned el_grid = (total + 255) / 256;
	uint32_t *cur = va.data(), *nxt = vb.data();
	auto sort_pass = [&](int end_bit) -> int {
		PGX_HIP(hipGetLastError());
		PGX_HIP(rocprim::segmented_radix_sort_pairs(sort_tmp.data(), sort_bytes, ka.data(), kb.data(), cur, nxt, total, n_big, seg.data(),
							    seg.data() + 1, 0, end_bit));
		std::swap(cur, nxt);
		return 0;
	};
	hipLaunchKernelGGL(k_big_keys<0>, dim3(el_grid), dim3(256), 0, 0, work.data(), cur, best.data(), total, ka.data());
	PGX_TRY(sort_pass(64));
	hipLaunchKernelGGL(k_big_keys<1>, dim3(el_grid), dim3(256), 0, 0, work.data(), cur, best.data(), total, ka.data());
	PGX_TRY(sort_pass(33));
	hipLaunchKernelGGL(k_big_keys<2>, dim3(el_grid), dim3(256), 0, 0, work.data(), cur, best.data(), total, ka.data());
	PGX_TRY(sort_pass(64));
	hipLaunchKernelGGL(k_big_keys<3>, dim3(el_grid), dim3(256), 0, 0, work.data(), cur, best.data(), total, ka.data());
	PGX_TRY(sort_pass(64));
	hipLaunchKernelGGL(k_big_best, dim3(seg_grid), dim3(256), 0, 0, work.data(), cur, seg.data(), n_big, best.data());
	if (gapped)
		hipLaunchKernelGGL(k_big_dedup, dim3(seg_grid), dim3(256), 0, 0, work.data(), cur, seg.data(), n_big, best.data(), seg_drop.data());
	hipLaunchKernelGGL(k_big_keys<4>, dim3(el_grid), dim3(256), 0, 0, work.data(), cur, best.data(), total, ka.data());
	PGX_TRY(sort_pass(32));
	hipLaunchKernelGGL(k_big_write, dim3(seg_grid), dim3(256), 0, 0, work.data(), cur, seg.data(), off, big_list, n_big, hits, read_cnt, seg_drop.data());
	PGX_HIP(hipGetLastError());
	return 0;
}

} // namespace pgx
