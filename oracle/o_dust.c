/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * BLAST mode, spec "pgx-blastn v2", S3d: low-complexity masking of the QUERY for seeding (`blastn -dust "20 64 1"`, BLAST+'s
 * default; reference call site README.md:96 uses the defaults).  PARITY UNPINNED (BLAST+ is not vendored).  This restates
 * the DEFINITION of the published algorithm (Morgulis, Gertz, Schaffer, Agarwala: "A fast and symmetric DUST
 * implementation to mask low-complexity DNA sequences", J Comput Biol 13 (2006)), not its incremental bookkeeping:
 *   - the sequence is read as overlapping triplets; an interval of l >= 2 consecutive triplets scores
 *     S = sum over triplet values t of c_t (c_t - 1) / 2, divided by (l - 1), c_t = occurrences of t in the interval;
 *   - an interval of at most W - 2 = 62 triplets (window W = 64 bases) is PERFECT when S > T = 2.0 (level 20) and no
 *     sub-interval of it scores higher;
 *   - every base of every perfect interval is masked (a triplet interval [a, b] covers bases a .. b + 2); the linker of 1
 *     joins touching intervals, which the union already does.
 * A triplet that holds a letter other than A C G T belongs to no interval.  Scores are compared as exact fractions.
 * The mask only removes SEEDS (a 28-base window that touches a masked base seeds nothing); extensions run through
 * masked letters, as in BLAST (soft masking).
 */
#include "o_classify.h"
#include <stdlib.h>
#include <string.h>

#define DUST_W 64
#define DUST_LEVEL 20
#define DUST_MAXT (DUST_W - 2)

typedef struct {
	int32_t r, q; /* score r / q (q = triplets - 1); q = 0: no score */
} frac;

static inline int frac_gt(frac a, frac b) /* a > b; a missing score is below every score */
{
	if (a.q == 0)
		return 0;
	if (b.q == 0)
		return 1;
	return (int64_t)a.r * b.q > (int64_t)b.r * a.q;
}

void o_dust_mask(const uint8_t *base, int32_t len, uint8_t *mask)
{
	memset(mask, 0, (size_t)(len > 0 ? len : 0));
	const int32_t nt = len - 2;
	if (nt < 2)
		return;
	int16_t *trip = (int16_t *)malloc((size_t)nt * sizeof(int16_t));
	for (int32_t i = 0; i < nt; i++)
		trip[i] = (base[i] < 4 && base[i + 1] < 4 && base[i + 2] < 4) ? (int16_t)(base[i] * 16 + base[i + 1] * 4 + base[i + 2]) : (int16_t)-1;
	/* best[b - a]: the highest score of any sub-interval of [a, b] (the interval itself included); row a + 1 is kept */
	frac *below = (frac *)calloc((size_t)DUST_MAXT + 1, sizeof(frac)), *row = (frac *)calloc((size_t)DUST_MAXT + 1, sizeof(frac));
	for (int32_t a = nt - 1; a >= 0; a--) {
		int32_t cnt[64];
		memset(cnt, 0, sizeof cnt);
		int32_t r = 0;
		frac left = { 0, 0 }; /* best of [a, b - 1] */
		int32_t b;
		for (b = a; b < nt && b - a < DUST_MAXT && trip[b] >= 0; b++) {
			r += cnt[trip[b]]++;
			frac s = { r, b - a };
			/* sub-intervals: everything inside [a + 1, b] (row below, same b: index b - a - 1) and inside [a, b - 1] */
			frac sub = left;
			if (b > a && frac_gt(below[b - a - 1], sub))
				sub = below[b - a - 1];
			if (s.q > 0 && (int64_t)s.r * 10 > (int64_t)DUST_LEVEL * s.q && !frac_gt(sub, s))
				for (int32_t k = a; k <= b + 2; k++)
					mask[k] = 1;
			frac best = frac_gt(s, sub) ? s : sub;
			row[b - a] = best;
			left = best;
		}
		/* row a becomes the row below; entries past its end carry no score (an interval through a bad triplet) */
		for (int32_t k = b - a; k <= DUST_MAXT; k++)
			row[k].q = 0, row[k].r = 0;
		frac *t = below;
		below = row;
		row = t;
	}
	free(below);
	free(row);
	free(trip);
}
