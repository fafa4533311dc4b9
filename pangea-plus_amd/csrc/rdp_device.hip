// The RDP stream of a batch, parsed on the device.
//
// Input format: the classifier's five-tab text that Consensus_BLAST_SOAP_RDP-1.1.pl reads (reference :126-132):
//     <read id> TAB TAB TAB TAB TAB <name> TAB <rank> TAB <confidence> TAB <name> TAB <rank> TAB <confidence> ...
// and the script's cursor rule: a line belongs to the first read at or after the cursor that carries its name (:141, :211,
// :216-220).  pgx_rdp_from_file (annotate.hip) did all of it on the host cores -- 0.10-0.12 s for 2 M lines, the largest
// part of the file-to-file leg; this is the same import with the text in HBM:
//   line index          k_nl_count / k_nl_write (trim.hip: the kernels that index read files)
//   k_rdp_lines         one lane per line, bytes in aligned 16-byte words: the first run of five tabs (end of the id), the
//                       id's FNV-1a hash, the field region behind it (cut at a second five-tab group, trailing tabs
//                       dropped), the number of name fields
//   k_rdp_name_table    one lane per read of the batch: its name's hash into an open-addressed table (64-bit compare-and-
//                       swap); two reads with one hash -> the host form takes over (names that repeat need its chains)
//   k_rdp_match         one lane per line: the read of that name (hash probe, then the bytes compared)
//   cursor rule         with names that do not repeat, "first read at or after the cursor" accepts exactly the lines
//                       whose read lies beyond every read named before: an exclusive prefix maximum (rocPRIM scan)
//   k_rdp_accept        accepted lines -> present[read], line of the read, triplets of the read; scan -> offsets
//   k_rdp_fields        one lane per read with a line: the fields, name and rank texts by their FNV-1a hash into two small
//                       device hash sets that remember where the first copy of each text lies in the file
//   host                the few thousand DISTINCT name / rank texts: the script's cleaning (:159-160) and the token table
//                       (the one step that is host work by nature: the token ids are the database's)
//   k_rdp_codes         slots -> token ids, rank indices and the packed codes the consensus kernels read
// Field texts are identified by their 64-bit hash (a collision among the few thousand distinct names of a file has
// probability ~1e-12); read names are compared byte for byte.
#include <chrono>
#include <cstring>
#include <memory>

#include <rocprim/rocprim.hpp>

#include "engine.hpp"

namespace pgx {

namespace {

constexpr uint64_t kFnvBasis = 1469598103934665603ull, kFnvPrime = 1099511628211ull;
constexpr uint32_t kNone = 0xFFFFFFFFu;

struct LineInfo {
	uint32_t id_len;   // bytes in front of the first five-tab run (the whole line without one)
	uint32_t rest_len; // field region: starts id_len + 5 bytes into the line; 0 = no fields (or no five-tab run)
	uint32_t n_names;  // name fields in it
	uint32_t has_five;
};

// bytes [s, e) of the text, in order, through aligned 16-byte loads (lanes are a line apart: a byte load per lane and
// byte asks the cache for the same line sixteen times over)
template <typename F> __device__ __forceinline__ void for_bytes(const uint8_t *__restrict__ text, uint64_t s, uint64_t e, F f)
{
	for (uint64_t off = s & ~15ull; off < e; off += 16) {
		const uint4 v = *reinterpret_cast<const uint4 *>(text + off); // (the text is padded to whole 16-byte words)
		const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int k = 0; k < 16; k++) {
			const uint64_t pos = off + (uint64_t)k;
			if (pos >= s && pos < e)
				f(pos, (uint8_t)(w[k >> 2] >> (8 * (k & 3))));
		}
	}
}

__device__ __forceinline__ void line_span(const uint8_t *__restrict__ text, const uint64_t *__restrict__ start, uint32_t i, uint64_t &s, uint64_t &e)
{
	s = start[i];
	e = start[i + 1];
	if (e > s && text[e - 1] == '\n')
		e--;
}

__global__ __launch_bounds__(256) void k_rdp_lines(const uint8_t *__restrict__ text, const uint64_t *__restrict__ start, uint32_t n_lines,
						    LineInfo *__restrict__ info, uint64_t *__restrict__ id_hash)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_lines)
		return;
	uint64_t s, e;
	line_span(text, start, i, s, e);
	uint64_t h = kFnvBasis, h_at_run = kFnvBasis, id_h = kFnvBasis; // hash so far; as it stood when the current tab run began; the id's
	uint32_t run = 0;                                                // tabs in a row up to here
	bool five = false, cut = false, any_text = false;
	uint64_t five_at = 0, rest0 = 0, last_text = 0; // first byte of the first five-tab run; first byte behind it; last byte of the fields that is no tab
	uint32_t tabs = 0, tabs_at_text = 0;            // behind the run: tabs so far; tabs in front of the last text byte
	for_bytes(text, s, e, [&](uint64_t pos, uint8_t c) {
		if (cut)
			return;
		if (!five) {
			if (c == '\t') {
				if (run == 0)
					h_at_run = h;
				if (++run == 5) {
					five = true;
					five_at = pos - 4;
					rest0 = pos + 1;
					id_h = h_at_run;
					run = 0;
					return;
				}
			} else {
				run = 0;
			}
			h = (h ^ c) * kFnvPrime;
			return;
		}
		if (c == '\t') {
			tabs++;
			if (++run == 5)
				cut = true; // a second five-tab group ends the fields (its first tab lies behind the last text byte)
		} else {
			run = 0;
			any_text = true;
			last_text = pos;
			tabs_at_text = tabs;
		}
	});
	LineInfo li;
	li.has_five = five ? 1u : 0u;
	li.id_len = five ? (uint32_t)(five_at - s) : (uint32_t)(e - s);
	li.rest_len = five && any_text ? (uint32_t)(last_text + 1 - rest0) : 0u;
	li.n_names = five && any_text ? (tabs_at_text + 3u) / 3u : 0u; // fields = tabs + 1; every third one, from the first, is a name
	info[i] = li;
	id_hash[i] = five ? id_h : h;
}

// open-addressed table over 64-bit hashes (bit 0 forced: 0 = empty slot)
__device__ __forceinline__ uint32_t table_insert(unsigned long long *__restrict__ keys, uint32_t mask, uint64_t h, bool *fresh)
{
	const unsigned long long key = h | 1ull;
	for (uint32_t k = (uint32_t)(h >> 8) & mask, tries = 0; tries <= mask; k = (k + 1) & mask, tries++) {
		// (a plain look first: nearly every field text of a file is already in the set -- a few thousand taxa for millions
		// of lines -- and a compare-and-swap per field on the slot of "Bacteria" is two million atomics on one address,
		// which takes ~90 M a second: 21 ms of a 500 000-line file's 50)
		const unsigned long long seen = __atomic_load_n(&keys[k], __ATOMIC_RELAXED);
		if (seen == key) {
			*fresh = false;
			return k;
		}
		if (seen != 0ull)
			continue;
		const unsigned long long old = atomicCAS(&keys[k], 0ull, key);
		if (old == 0ull) {
			*fresh = true;
			return k;
		}
		if (old == key) {
			*fresh = false;
			return k;
		}
	}
	*fresh = false;
	return kNone;
}

__device__ __forceinline__ uint32_t table_find(const unsigned long long *__restrict__ keys, uint32_t mask, uint64_t h)
{
	const unsigned long long key = h | 1ull;
	for (uint32_t k = (uint32_t)(h >> 8) & mask, tries = 0; tries <= mask; k = (k + 1) & mask, tries++) {
		const unsigned long long cur = keys[k];
		if (cur == key)
			return k;
		if (cur == 0ull)
			return kNone;
	}
	return kNone;
}

__global__ __launch_bounds__(256) void k_rdp_name_table(const unsigned char *__restrict__ names, const uint32_t *__restrict__ name_at, uint32_t n_reads,
							 unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t mask, uint32_t *__restrict__ flags)
{
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;
	if (r >= n_reads)
		return;
	uint64_t h = kFnvBasis;
	for (uint32_t k = name_at[r]; k < name_at[r + 1]; k++)
		h = (h ^ names[k]) * kFnvPrime;
	bool fresh;
	const uint32_t slot = table_insert(keys, mask, h, &fresh);
	if (slot == kNone || !fresh)
		atomicOr(&flags[0], 1u); // two reads with one hash (a name that repeats): the host form
	else
		vals[slot] = r;
}

__global__ __launch_bounds__(256) void k_rdp_match(const uint8_t *__restrict__ text, const uint64_t *__restrict__ start, uint32_t n_lines,
						    const LineInfo *__restrict__ info, const uint64_t *__restrict__ id_hash,
						    const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t mask,
						    const unsigned char *__restrict__ names, const uint32_t *__restrict__ name_at, uint32_t *__restrict__ cand1)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_lines)
		return;
	uint32_t v = 0; // read + 1, 0 = no read of that name
	const uint32_t slot = table_find(keys, mask, id_hash[i]);
	if (slot != kNone) {
		const uint32_t r = vals[slot], a = name_at[r], n = name_at[r + 1] - a;
		if (n == info[i].id_len) {
			const uint64_t s = start[i];
			bool same = true;
			for (uint32_t k = 0; k < n && same; k++)
				same = names[a + k] == text[s + k];
			if (same)
				v = r + 1;
		}
	}
	cand1[i] = v;
}

// a line is taken when its read lies beyond every read named by a line before it (header)
__global__ __launch_bounds__(256) void k_rdp_accept(const uint32_t *__restrict__ cand1, const uint32_t *__restrict__ before_max, uint32_t n_lines,
						     const LineInfo *__restrict__ info, uint8_t *__restrict__ present, uint32_t *__restrict__ line_of,
						     uint32_t *__restrict__ trips)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_lines)
		return;
	const uint32_t v = cand1[i];
	if (v == 0 || v <= before_max[i])
		return;
	const uint32_t r = v - 1;
	present[r] = 1;
	line_of[r] = i;
	trips[r] = info[i].n_names;
}

struct FieldSets {
	unsigned long long *name_keys, *rank_keys;
	uint2 *name_where, *rank_where; // file offset (low 32 bits in .x, bits 32.. in the top byte of .y) and length of a text's first copy
	uint32_t name_mask, rank_mask;
	uint32_t *name_list, *rank_list; // slots in the order they were filled
	uint32_t *counts;                 // [0] names listed, [1] ranks listed, [2] a table was full
};

__device__ __forceinline__ uint2 where_of(uint64_t off, uint32_t len) { return make_uint2((uint32_t)off, (uint32_t)(off >> 32) << 24 | (len & 0xFFFFFFu)); }

__global__ __launch_bounds__(256) void k_rdp_fields(const uint8_t *__restrict__ text, const uint64_t *__restrict__ start, uint32_t n_reads,
						     const LineInfo *__restrict__ info, const uint32_t *__restrict__ line_of, const uint32_t *__restrict__ off,
						     FieldSets fs, uint32_t *__restrict__ name_slot, uint32_t *__restrict__ rank_slot)
{
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;
	if (r >= n_reads)
		return;
	const uint32_t i = line_of[r];
	if (i == kNone)
		return;
	const LineInfo li = info[i];
	if (li.rest_len == 0)
		return;
	const uint64_t s = start[i] + li.id_len + 5, e = s + li.rest_len;
	uint32_t t = off[r] - 1; // triplet of the current field (the first name field moves it to off[r])
	uint32_t k = 0;          // field number
	uint64_t h = kFnvBasis, a = s;
	auto close = [&](uint64_t x) { // field k = [a, x)
		if (k % 3 == 0) {
			t++;
			bool fresh;
			const uint32_t slot = table_insert(fs.name_keys, fs.name_mask, h, &fresh);
			if (slot == kNone) {
				atomicOr(&fs.counts[2], 1u);
			} else if (fresh) {
				fs.name_where[slot] = where_of(a, (uint32_t)(x - a));
				fs.name_list[atomicAdd(&fs.counts[0], 1u)] = slot;
			}
			name_slot[t] = slot;
			rank_slot[t] = kNone; // (a name without a rank field behind it)
		} else if (k % 3 == 1) {
			bool fresh;
			const uint32_t slot = table_insert(fs.rank_keys, fs.rank_mask, h, &fresh);
			if (slot == kNone) {
				atomicOr(&fs.counts[2], 1u);
			} else if (fresh) {
				fs.rank_where[slot] = where_of(a, (uint32_t)(x - a));
				fs.rank_list[atomicAdd(&fs.counts[1], 1u)] = slot;
			}
			rank_slot[t] = slot;
		}
		k++;
		a = x + 1;
		h = kFnvBasis;
	};
	for_bytes(text, s, e, [&](uint64_t pos, uint8_t c) {
		if (c == '\t')
			close(pos);
		else
			h = (h ^ c) * kFnvPrime;
	});
	close(e);
}

__global__ __launch_bounds__(256) void k_rdp_pick(const uint32_t *__restrict__ list, const uint2 *__restrict__ where, uint32_t n, uint2 *__restrict__ out)
{
	const uint32_t j = blockIdx.x * 256u + threadIdx.x;
	if (j < n)
		out[j] = where[list[j]];
}

__global__ __launch_bounds__(256) void k_rdp_spread(const uint32_t *__restrict__ list, const uint32_t *__restrict__ v, uint32_t n, uint32_t *__restrict__ table)
{
	const uint32_t j = blockIdx.x * 256u + threadIdx.x;
	if (j < n)
		table[list[j]] = v[j];
}

__global__ __launch_bounds__(256) void k_rdp_codes(const uint32_t *__restrict__ name_slot, const uint32_t *__restrict__ rank_slot, uint32_t n_trip,
						    const uint32_t *__restrict__ name_tok, const uint32_t *__restrict__ rank_idx, uint32_t *__restrict__ name,
						    int8_t *__restrict__ rank, uint32_t *__restrict__ code)
{
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t >= n_trip)
		return;
	const uint32_t tk = name_tok[name_slot[t]];
	const int8_t rk = rank_slot[t] == kNone ? (int8_t)-1 : (int8_t)(uint8_t)rank_idx[rank_slot[t]];
	name[t] = tk;
	rank[t] = rk;
	code[t] = (tk << 3) | (uint32_t)(rk + 1);
}

struct MaxOp {
	__device__ __host__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

template <typename T, typename Op> int scan_exclusive(const T *in, T *out, T init, size_t n, Op op)
{
	size_t bytes = 0;
	PGX_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, init, n, op));
	DevBuf<uint8_t> tmp;
	PGX_TRY(tmp.alloc(bytes ? bytes : 1));
	PGX_HIP(rocprim::exclusive_scan(tmp.data(), bytes, in, out, init, n, op));
	return 0;
}

} // namespace

int rdp_from_text_device(const char *text, size_t n_bytes, const pgx_reads *reads, pgx_db *db, pgx_rdp **out)
{
	const size_t n = (size_t)reads->n;
	if (reads->synthetic || !reads->d_name_at.base || n == 0 || n >= 0x7FFFFFFFull || n_bytes == 0 || n_bytes >= (1ull << 40))
		return 1; // (names not resident, or sizes this form does not hold)
	const bool trace = getenv("PGX_TRACE") != nullptr;
	auto t_prev = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (!trace)
			return;
		(void)hipDeviceSynchronize();
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[pgx trace] rdp on the device, %s: %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
		t_prev = now;
	};
	DevBuf<uint8_t> d_text;
	PGX_TRY(d_text.alloc(n_bytes, 0, 16));
	PGX_TRY(d_text.upload((const uint8_t *)text, n_bytes));
	lap("upload");
	DevBuf<uint64_t> d_start;
	uint64_t n_lines64 = 0;
	PGX_TRY(device_line_index(d_text.data(), n_bytes, text[n_bytes - 1] != '\n', d_start, &n_lines64));
	if (n_lines64 == 0 || n_lines64 >= 0x7FFFFFFFull)
		return 1;
	const uint32_t n_lines = (uint32_t)n_lines64;
	const unsigned gl = (n_lines + 255) / 256, gr = (unsigned)((n + 255) / 256);
	DevBuf<LineInfo> d_info;
	DevBuf<uint64_t> d_idh;
	PGX_TRY(d_info.alloc(n_lines));
	PGX_TRY(d_idh.alloc(n_lines));
	hipLaunchKernelGGL(k_rdp_lines, dim3(gl), dim3(256), 0, 0, d_text.data(), d_start.data(), n_lines, d_info.data(), d_idh.data());
	// the reads by name
	uint32_t cap = 1024;
	while (cap < 2 * n + 1)
		cap <<= 1;
	DevBuf<unsigned long long> d_keys;
	DevBuf<uint32_t> d_vals, d_flags;
	PGX_TRY(d_keys.alloc(cap, 0, 0, true));
	PGX_TRY(d_vals.alloc(cap));
	PGX_TRY(d_flags.alloc(4, 0, 0, true));
	hipLaunchKernelGGL(k_rdp_name_table, dim3(gr), dim3(256), 0, 0, reads->d_names.data(), reads->d_name_at.data(), (uint32_t)n, d_keys.data(), d_vals.data(),
			   cap - 1, d_flags.data());
	DevBuf<uint32_t> d_cand, d_before;
	PGX_TRY(d_cand.alloc(n_lines));
	PGX_TRY(d_before.alloc(n_lines));
	hipLaunchKernelGGL(k_rdp_match, dim3(gl), dim3(256), 0, 0, d_text.data(), d_start.data(), n_lines, d_info.data(), d_idh.data(), d_keys.data(), d_vals.data(),
			   cap - 1, reads->d_names.data(), reads->d_name_at.data(), d_cand.data());
	PGX_HIP(hipGetLastError());
	uint32_t flag = 0;
	PGX_TRY(d_flags.download(&flag, 1));
	if (flag)
		return 1; // names that repeat inside the batch: the host form walks its chains
	PGX_TRY(scan_exclusive<uint32_t>(d_cand.data(), d_before.data(), 0u, n_lines, MaxOp()));
	pgx_rdp *rd = new pgx_rdp();
	std::unique_ptr<pgx_rdp> hold(rd);
	rd->n = (int64_t)n;
	DevBuf<uint32_t> d_line_of, d_trips;
	PGX_TRY(rd->d_present.alloc(n, 0, 0, true));
	PGX_TRY(d_line_of.alloc(n));
	PGX_TRY(d_trips.alloc(n + 1, 0, 0, true));
	PGX_HIP(hipMemset(d_line_of.data(), 0xFF, n * sizeof(uint32_t)));
	hipLaunchKernelGGL(k_rdp_accept, dim3(gl), dim3(256), 0, 0, d_cand.data(), d_before.data(), n_lines, d_info.data(), rd->d_present.data(), d_line_of.data(),
			   d_trips.data());
	PGX_TRY(rd->d_off.alloc(n + 1));
	PGX_TRY(scan_exclusive<uint32_t>(d_trips.data(), rd->d_off.data(), 0u, n + 1, rocprim::plus<uint32_t>()));
	uint32_t n_trip = 0, most = 0;
	PGX_TRY(rd->d_off.download(&n_trip, 1, n));
	{
		// the most triplets any read carries (bounds the consensus kernels' compare grid)
		DevBuf<uint32_t> d_most;
		PGX_TRY(d_most.alloc(1));
		size_t bytes = 0;
		PGX_HIP(rocprim::reduce(nullptr, bytes, d_trips.data(), d_most.data(), 0u, n, MaxOp()));
		DevBuf<uint8_t> tmp;
		PGX_TRY(tmp.alloc(bytes ? bytes : 1));
		PGX_HIP(rocprim::reduce(tmp.data(), bytes, d_trips.data(), d_most.data(), 0u, n, MaxOp()));
		PGX_TRY(d_most.download(&most, 1));
	}
	rd->max_trip = (int)std::min<uint32_t>(most, 8);
	lap("lines, reads, cursor rule");
	// the fields
	DevBuf<uint32_t> d_name_slot, d_rank_slot;
	PGX_TRY(d_name_slot.alloc(n_trip ? n_trip : 1));
	PGX_TRY(d_rank_slot.alloc(n_trip ? n_trip : 1));
	const uint32_t name_cap = 1u << 21, rank_cap = 1u << 12; // (distinct texts: at most half of these)
	DevBuf<unsigned long long> d_nkeys, d_rkeys;
	DevBuf<uint2> d_nwhere, d_rwhere;
	DevBuf<uint32_t> d_nlist, d_rlist, d_counts;
	PGX_TRY(d_nkeys.alloc(name_cap, 0, 0, true));
	PGX_TRY(d_rkeys.alloc(rank_cap, 0, 0, true));
	PGX_TRY(d_nwhere.alloc(name_cap));
	PGX_TRY(d_rwhere.alloc(rank_cap));
	PGX_TRY(d_nlist.alloc(name_cap));
	PGX_TRY(d_rlist.alloc(rank_cap));
	PGX_TRY(d_counts.alloc(4, 0, 0, true));
	FieldSets fs{ d_nkeys.data(), d_rkeys.data(), d_nwhere.data(), d_rwhere.data(), name_cap - 1, rank_cap - 1, d_nlist.data(), d_rlist.data(), d_counts.data() };
	hipLaunchKernelGGL(k_rdp_fields, dim3(gr), dim3(256), 0, 0, d_text.data(), d_start.data(), (uint32_t)n, d_info.data(), d_line_of.data(), rd->d_off.data(), fs,
			   d_name_slot.data(), d_rank_slot.data());
	PGX_HIP(hipGetLastError());
	uint32_t counts[4] = { 0, 0, 0, 0 };
	PGX_TRY(d_counts.download(counts, 4));
	if (counts[2] || counts[0] > name_cap / 2 || counts[1] > rank_cap / 2)
		return 1; // more distinct texts than the device sets hold: the host form
	lap("fields");
	// the distinct texts: cleaned and numbered on the host (the token ids are the database's own table)
	// (only the listed slots travel: the sets are sized for the worst file, a real one fills a few thousand slots)
	std::vector<uint32_t> nlist(counts[0]), rlist(counts[1]);
	std::vector<uint2> nwhere(counts[0]), rwhere(counts[1]);
	DevBuf<uint2> d_pick;
	PGX_TRY(d_pick.alloc(std::max<uint32_t>(std::max(counts[0], counts[1]), 1u)));
	auto pick = [&](const DevBuf<uint32_t> &list, const DevBuf<uint2> &where, uint32_t cnt, std::vector<uint32_t> &hl, std::vector<uint2> &hw) -> int {
		if (!cnt)
			return 0;
		hipLaunchKernelGGL(k_rdp_pick, dim3((cnt + 255) / 256), dim3(256), 0, 0, list.data(), where.data(), cnt, d_pick.data());
		PGX_HIP(hipGetLastError());
		PGX_TRY(list.download(hl.data(), cnt));
		return d_pick.download(hw.data(), cnt);
	};
	PGX_TRY(pick(d_nlist, d_nwhere, counts[0], nlist, nwhere));
	PGX_TRY(pick(d_rlist, d_rwhere, counts[1], rlist, rwhere));
	auto span = [&](const uint2 &w, std::string &t) -> bool {
		const uint64_t at = (uint64_t)w.x | ((uint64_t)(w.y >> 24) << 32);
		const uint32_t len = w.y & 0xFFFFFFu;
		if (at + len > n_bytes)
			return false;
		t.assign(text + at, len);
		return true;
	};
	std::string field;
	std::vector<uint32_t> ntok(counts[0] ? counts[0] : 1), ridx(counts[1] ? counts[1] : 1);
	for (uint32_t j = 0; j < counts[0]; j++) {
		if (!span(nwhere[j], field))
			return fail(PGX_E_FORMAT, "RDP import: a field lies outside the file");
		ntok[j] = db->intern(clean_rdp_name(field));
	}
	for (uint32_t j = 0; j < counts[1]; j++) {
		if (!span(rwhere[j], field))
			return fail(PGX_E_FORMAT, "RDP import: a field lies outside the file");
		ridx[j] = (uint32_t)(uint8_t)rdp_rank_index(field);
	}
	DevBuf<uint32_t> d_name_tok, d_rank_idx, d_vals2;
	PGX_TRY(d_name_tok.alloc(name_cap));
	PGX_TRY(d_rank_idx.alloc(rank_cap));
	PGX_TRY(d_vals2.alloc(std::max<uint32_t>(std::max(counts[0], counts[1]), 1u)));
	auto spread = [&](const DevBuf<uint32_t> &list, const std::vector<uint32_t> &v, uint32_t cnt, DevBuf<uint32_t> &table) -> int {
		if (!cnt)
			return 0;
		PGX_TRY(d_vals2.upload(v.data(), cnt));
		hipLaunchKernelGGL(k_rdp_spread, dim3((cnt + 255) / 256), dim3(256), 0, 0, list.data(), d_vals2.data(), cnt, table.data());
		PGX_HIP(hipGetLastError());
		return 0;
	};
	PGX_TRY(spread(d_nlist, ntok, counts[0], d_name_tok));
	PGX_TRY(spread(d_rlist, ridx, counts[1], d_rank_idx));
	PGX_TRY(rd->d_name.alloc(n_trip ? n_trip : 1));
	PGX_TRY(rd->d_rank.alloc(n_trip ? n_trip : 1));
	PGX_TRY(rd->d_code.alloc(n_trip ? n_trip : 1));
	if (n_trip)
		hipLaunchKernelGGL(k_rdp_codes, dim3((n_trip + 255) / 256), dim3(256), 0, 0, d_name_slot.data(), d_rank_slot.data(), n_trip, d_name_tok.data(),
				   d_rank_idx.data(), rd->d_name.data(), rd->d_rank.data(), rd->d_code.data());
	PGX_HIP(hipGetLastError());
	PGX_HIP(hipDeviceSynchronize());
	lap("distinct texts, codes");
	*out = hold.release();
	return 0;
}

} // namespace pgx
