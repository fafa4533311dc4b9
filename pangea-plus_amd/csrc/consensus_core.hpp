// Device core of the consensus arg-max (Consensus/Consensus_BLAST_SOAP_RDP-1.1.pl:141-234), shared
// by the fused pipeline (classify.hip) and the file verb (annotate.hip).  Text has been interned
// to integers on the host: token ids for names, small ints for rank indices, and string-ORDER
// ranks for the similarity column, so the Perl's `eq`/`gt`/`lt` become integer operations.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pgx {

// Perl `gt` on the decimal texts of two non-negative integers (Consensus:191,199):
// "10" gt "9" is false, "14" gt "6" is false, "6" gt "14" is true
__device__ __forceinline__ int dec_digits(uint32_t v)
{
	return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6
	       : v < 10000000u ? 7 : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}

__device__ __forceinline__ bool dec_str_gt(uint32_t a, uint32_t b)
{
	if (a == b)
		return false;
	const int da = dec_digits(a), db = dec_digits(b);
	if (da == db)
		return a > b; // equal length: text order is numeric order
	// pad the shorter text with zeros to the longer length: a differing digit decides exactly as in the
	// text comparison; if none differs the shorter text is a prefix of the longer one and is the smaller
	const uint64_t p10[10] = { 1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull };
	if (da < db) {
		const uint64_t as = (uint64_t)a * p10[db - da];
		return as > (uint64_t)b; // equal means prefix: a is shorter, so a is not greater
	}
	const uint64_t bs = (uint64_t)b * p10[da - db];
	return (uint64_t)a >= bs; // equal means b is a prefix of a: the longer a is greater
}

// Order-preserving integer for the decimal TEXT of v: digits left-aligned to 10 places, ties (one text a
// prefix of the other, e.g. "1" / "10") broken by length.  str_key(a) > str_key(b)  <=>  "a" gt "b".
__device__ __forceinline__ uint64_t dec_str_key(uint32_t v)
{
	const uint64_t p10[10] = { 1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull };
	const int d = dec_digits(v);
	return ((uint64_t)v * p10[10 - d]) * 16ull + (uint64_t)d;
}

// the same order in 32 bits, for v < 10^8
__device__ __forceinline__ uint32_t dec_str_key32(uint32_t v)
{
	// (selects, not a table of powers: the table is memory, and its load sat in the middle of the ordering kernel's chain)
	const bool d1 = v < 10u, d2 = v < 100u, d3 = v < 1000u, d4 = v < 10000u, d5 = v < 100000u, d6 = v < 1000000u, d7 = v < 10000000u;
	const uint32_t mul = d1 ? 10000000u : d2 ? 1000000u : d3 ? 100000u : d4 ? 10000u : d5 ? 1000u : d6 ? 100u : d7 ? 10u : 1u;
	const uint32_t d = d1 ? 1u : d2 ? 2u : d3 ? 3u : d4 ? 4u : d5 ? 5u : d6 ? 6u : d7 ? 7u : 8u;
	return (v * mul) * 16u + d;
}

// number of (a,b) with blasttax[a+1] eq clean(rdptax[b]) and index1 eq index2 (Consensus:154-184).
// tok: the hit's lineage tokens (rank,name,rank,name,...); token id 0 is the empty string, which is
// also what an undefined name compares as; rank index -1 is undef on either side.
__device__ __forceinline__ uint32_t rank_matches(const uint32_t *tok, uint32_t ntok, const int8_t *tok_rank,
						  const uint32_t *rdp_name, const int8_t *rdp_rank, uint32_t r0, uint32_t r1)
{
	uint32_t rm = 0;
	for (uint32_t a = 0; a < ntok; a += 2) {
		const int i1 = tok_rank[tok[a]];
		const uint32_t name = a + 1 < ntok ? tok[a + 1] : 0u;
		for (uint32_t b = r0; b < r1; b++)
			rm += (name == rdp_name[b]) && (i1 == (int)rdp_rank[b]);
	}
	return rm;
}

// the order-dependent selection of Consensus:186-204
struct ArgmaxState {
	uint32_t maxrm = 0, maxcnt = 0, cursim = 0;
	int32_t win = -1;
	uint64_t kmaxrm = 16ull * 0 + 1, kmaxcnt = 16ull * 0 + 1; // dec_str_key(0)
	// same selection with the text keys precomputed per hit (k_sort_consensus computes them in parallel)
	__device__ __forceinline__ void step_keys(int32_t index, uint32_t rm, uint64_t krm, uint64_t kcnt, uint32_t sim)
	{
		if (krm > kmaxrm) {
			kmaxrm = krm;
			maxrm = rm;
			win = index;
			cursim = sim;
		}
		if ((kcnt > kmaxcnt || cursim < sim) && rm == maxrm) {
			kmaxcnt = kcnt;
			win = index;
			cursim = sim;
		}
	}
	__device__ __forceinline__ void step(int32_t index, uint32_t rm, uint32_t cnt, uint32_t sim)
	{
		if (dec_str_gt(rm, maxrm)) {
			maxrm = rm;
			win = index;
			cursim = sim;
		}
		if ((dec_str_gt(cnt, maxcnt) || cursim < sim) && rm == maxrm) {
			maxcnt = cnt;
			win = index;
			cursim = sim;
		}
	}
};

} // namespace pgx
