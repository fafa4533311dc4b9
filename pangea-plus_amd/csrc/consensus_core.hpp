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
__device__ __forceinline__ bool dec_str_gt(uint32_t a, uint32_t b)
{
	if (a == b)
		return false;
	uint32_t pa = 1, pb = 1; // 10^(digits-1)
	while (a / pa >= 10)
		pa *= 10;
	while (b / pb >= 10)
		pb *= 10;
	while (pa && pb) {
		uint32_t da = (a / pa) % 10, db = (b / pb) % 10;
		if (da != db)
			return da > db;
		pa /= 10;
		pb /= 10;
	}
	return pa != 0; // the longer text wins when the shorter one is its prefix
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
