/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * BLAST mode, spec "pgx-blastn v2", stage S3b: the gapped extension of an initial (ungapped) HSP.
 * PARITY UNPINNED against NCBI BLAST+ 2.2.26 (not vendored: Classify/Runblast/install_blast.sh:67; the reference only
 * calls it, README.md:96).  What the reference does hold is its output: validation_dataset/Data-set_2_consensus.xlsx,
 * 10 933 of whose 10 992 rows have gapopen > 0 (row 3: 81.87 % identity over 1 186 columns with 64 gap openings).
 * Rows like that need (1) a gapped stage and (2) a drop-off far above the ungapped one.  This file restates the
 * PUBLISHED algorithm megablast uses for both:
 *
 *   Zhang, Schwartz, Wagner, Miller: "A greedy algorithm for aligning DNA sequences", J Comput Biol 7 (2000), fig. 4
 *   match +1, mismatch -2, gap column -(2 + 1/2) = -2.5 (the linear cost the greedy formulation needs; it is the
 *   per-gap-column cost the spreadsheet rows obey: tests/test_oracle_blast_rows.py), X = 54 = floor(100 bits * ln 2 /
 *   1.28), blastn's final gapped drop-off.
 *
 * In that formulation an alignment from the anchor to (i, j) with d differences (mismatches + gap columns) scores
 * S = (i + j) / 2 - 3 d; everything here is kept doubled (S2 = i + j - 6 d) so that it stays integral.  R(d, k) is the
 * furthest i reached on diagonal k = i - j with d differences.  Stated rules where the paper leaves a choice:
 *   - a cell whose move would use a letter past either sequence end is dead; there is no other end rule;
 *   - a cell is dead when its score BEFORE sliding is below T[d - 19] - X (19 = floor((X + 1/2) / 3) + 1, T[d'] = best
 *     score seen with at most d' differences, 0 for d' < 0) -- the paper's X-drop test;
 *   - R(d, k) = max of the three parents; among equal values the parent on the same diagonal (mismatch) wins, then
 *     k - 1 (a gap in the subject row), then k + 1 (a gap in the query row);
 *   - the extension ends at the first cell (d ascending, then k ascending) that reaches the best score;
 *   - at most 1000 differences on one side of the anchor.
 * `prune` adds one cut that cannot change the result (tests run both ways): a cell whose score cannot pass the best
 * one even if every remaining letter matched -- min(M - k/2, N + k/2) - 3 d in S units -- is dead; so is its whole
 * subtree (the bound of a child is below its parent's), and no surviving cell ever has such a parent.
 */
#include "o_classify.h"
#include <stdlib.h>
#include <string.h>

#define G_X2 (2 * O_BLAST_XDROP_GAP)
#define G_LAG ((2 * O_BLAST_XDROP_GAP + 1) / 6 + 1) /* floor((X + 1/2) / 3) + 1 = 19 */
#define G_DMAX 1000
#define G_NONE (-0x3FFFFFFF)

/* exploration statistics, when o_greedy_stats is set (single-threaded runs only): sides by last difference count
 * explored, cells evaluated */
int o_greedy_stats = 0;
long long o_greedy_depth_hist[64], o_greedy_cells;

static inline int g_match(uint8_t a, uint8_t b)
{
	return a < 4 && a == b;
}

/* One direction.  a[x * step] for x in [0, M), b[y * step] for y in [0, N); step = +1 (right of the anchor) or -1. */
void o_greedy_extend(const uint8_t *a, int32_t M, const uint8_t *b, int32_t N, int step, int prune, o_gext *out)
{
	memset(out, 0, sizeof *out);
	int32_t i = 0;
	while (i < M && i < N && g_match(a[(int64_t)i * step], b[(int64_t)i * step]))
		i++;
	out->i = out->j = i;
	out->s2 = 2 * i;
	if (i == M || i == N) {
		if (o_greedy_stats)
			o_greedy_depth_hist[0]++;
		return; /* a sequence end: every further cell only loses */
	}
	const int32_t dcap = G_DMAX;
	/* per d: R and the move taken, diagonals -d .. d at index k + d */
	int32_t *Rprev = (int32_t *)malloc((size_t)(2 * dcap + 3) * sizeof(int32_t));
	int32_t *Rcur = (int32_t *)malloc((size_t)(2 * dcap + 3) * sizeof(int32_t));
	int32_t *T = (int32_t *)malloc((size_t)(dcap + 1) * sizeof(int32_t));
	uint8_t **mv = (uint8_t **)calloc((size_t)dcap + 1, sizeof(uint8_t *)); /* bits 0-1: parent (0 same k, 1 k-1, 2 k+1); bit 2: slid */
	/* Rprev is indexed k + dcap + 1 so that k - 1 / k + 1 never leave the array */
#define RP(k) Rprev[(k) + dcap + 1]
#define RC(k) Rcur[(k) + dcap + 1]
	int32_t L = 0, U = 0;
	RP(0) = i;
	T[0] = 2 * i;
	int32_t best = 2 * i, best_d = 0, best_k = 0, best_i = i;
	int32_t d;
	for (d = 1; d <= dcap; d++) {
		const int32_t tcmp = d - G_LAG >= 0 ? T[d - G_LAG] : 0;
		mv[d] = (uint8_t *)malloc((size_t)(2 * d + 1));
		int32_t nl = 1 << 30, nu = -(1 << 30);
		for (int32_t k = L - 1; k <= U + 1; k++) {
			int32_t v = G_NONE, par = 0;
			if (k >= L && k <= U && RP(k) != G_NONE) {
				v = RP(k) + 1;
				par = 0;
			}
			if (k - 1 >= L && RP(k - 1) != G_NONE && RP(k - 1) + 1 > v) {
				v = RP(k - 1) + 1;
				par = 1;
			}
			if (k + 1 <= U && RP(k + 1) != G_NONE && RP(k + 1) > v) {
				v = RP(k + 1);
				par = 2;
			}
			int32_t ii = v, jj = v - k;
			int dead = v == G_NONE || ii > M || jj > N || jj < 0;
			if (!dead && ii + jj - 6 * d < tcmp - G_X2)
				dead = 1;
			if (!dead && prune) {
				/* 2 * (min(M - k/2, N + k/2) - 3 d): the score if every remaining letter matched */
				const int32_t ub = (2 * M - k < 2 * N + k ? 2 * M - k : 2 * N + k) - 6 * d;
				if (ub <= best)
					dead = 1;
			}
			if (o_greedy_stats)
				o_greedy_cells += !dead;
			if (dead) {
				RC(k) = G_NONE;
				mv[d][k + d] = 0;
				continue;
			}
			const int32_t i0 = ii;
			while (ii < M && jj < N && g_match(a[(int64_t)ii * step], b[(int64_t)jj * step])) {
				ii++;
				jj++;
			}
			RC(k) = ii;
			mv[d][k + d] = (uint8_t)(par | (ii > i0 ? 4 : 0));
			const int32_t s2 = ii + jj - 6 * d;
			if (s2 > best) {
				best = s2;
				best_d = d;
				best_k = k;
				best_i = ii;
			}
			if (k < nl)
				nl = k;
			if (k > nu)
				nu = k;
		}
		T[d] = best;
		if (nl > nu)
			break;
		for (int32_t k = L - 1; k <= U + 1; k++)
			RP(k) = RC(k);
		L = nl;
		U = nu;
	}
	if (o_greedy_stats)
		o_greedy_depth_hist[d < 63 ? d : 63]++;
	/* traceback from the best cell: the moves of the path (bits 0-1 kind, bit 2: the cell slid over matches after it) */
	out->i = best_i;
	out->j = best_i - best_k;
	out->s2 = best;
	{
		uint8_t *seq = (uint8_t *)malloc((size_t)best_d + 1);
		int32_t k = best_k;
		for (int32_t dd = best_d; dd >= 1; dd--) {
			const uint8_t m = mv[dd][k + dd];
			seq[dd] = m;
			const int par = m & 3;
			if (par == 0)
				out->mism++;
			else if (par == 1)
				out->gap_s++; /* a query letter faces a gap in the subject row */
			else
				out->gap_q++; /* a subject letter faces a gap in the query row */
			k = par == 1 ? k - 1 : (par == 2 ? k + 1 : k);
		}
		/* a gap column opens a gap unless the column before it is a gap column of the same row */
		int last_kind = 0, last_slid = 1;
		for (int32_t dd = 1; dd <= best_d; dd++) {
			const int par = seq[dd] & 3;
			if (par != 0 && !(par == last_kind && !last_slid))
				out->gapopen++;
			last_kind = par;
			last_slid = (seq[dd] & 4) != 0;
		}
		free(seq);
	}
	out->d = best_d;
	for (int32_t dd = 1; dd <= dcap && mv[dd]; dd++)
		free(mv[dd]);
	free(mv);
	free(T);
	free(Rprev);
	free(Rcur);
#undef RP
#undef RC
}
