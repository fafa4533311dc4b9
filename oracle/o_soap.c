/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * SOAP mode: CPU restatement of what the reference's closed `soap` 2.21 binary does for
 * `soap -a reads.fa -D ref.fa.index -o out -M 4 [-r 0|1|2] [-n 5] [-u unmapped]`
 * (README.md:130-134, soap.man:29-83).  There is no source; every rule below was observed
 * by running the ELF in this container (oracle/gen_goldens_soap.py) and is pinned by
 * tests/golden/soap/:
 *   - reference and read letters other than A C G T (any case) are read as G, except that a
 *     run of >= 10 such letters in the reference is cut out of the index (2bwt-builder's
 *     .ann segments): no hit may overlap it;
 *   - reads shorter than 27 or with more than -n (5) non-ACGT letters are not aligned;
 *   - full-length ungapped hits with <= 2 mismatches on either strand; -M 4 keeps the hits
 *     with the fewest mismatches; a hit must end before the last base of its segment
 *     (sequence end or the base in front of a cut run);
 *   - row: name, sequence as aligned (reverse complement for '-'), 'h' x len, number of
 *     equal-best hits, 'a', len, strand, reference id, 1-based leftmost position, number
 *     of mismatches, one "<ref>-><offset & 255><read><qual-64>" per mismatch, "<len>M", MD;
 *   - mismatch entries ascend by offset unless one lies in the last 13 bases (then descend);
 *     qual-64 prints as 40 except -64 for a lone mismatch at offset 0 of a '-' hit;
 *   - MD omits zero counts between adjacent mismatches; -r 1 also omits a trailing zero.
 * Not pinned (documented deviation): the ORDER of equal-best rows under -r 2 (suffix-array
 * order in the ELF; here subject, position, strand) and WHICH hit -r 1 prints when there are
 * several (pseudo-random in the ELF; here the first in that order).
 */
#include "o_classify.h"
#include <stdlib.h>
#include <string.h>

#define SEED_K 9

typedef struct {
	int32_t subject, pos, strand, nmis;
	int32_t mis[2];
} shit;

static const char k_letters[5] = "ACGT";

static int cmp_shit(const void *a, const void *b)
{
	const shit *x = (const shit *)a, *y = (const shit *)b;
	if (x->subject != y->subject)
		return x->subject < y->subject ? -1 : 1;
	if (x->pos != y->pos)
		return x->pos < y->pos ? -1 : 1;
	return x->strand - y->strand;
}

static void format_row(obuf *out, const char *name, const uint8_t *rd, int32_t L, const shit *h, int nbest,
		       const o_seqset *ref, int repeat)
{
	char id[512];
	o_seq_id(ref->header[h->subject], id, sizeof id);
	obuf_puts(out, name);
	obuf_puts(out, "\t");
	for (int32_t k = 0; k < L; k++)
		obuf_put(out, &k_letters[rd[k]], 1);
	obuf_puts(out, "\t");
	for (int32_t k = 0; k < L; k++)
		obuf_puts(out, "h");
	obuf_printf(out, "\t%d\ta\t%d\t%c\t%s\t%d\t%d", nbest, L, h->strand ? '-' : '+', id, h->pos + 1, h->nmis);
	const uint8_t *rs = ref->base + ref->off[h->subject] + h->pos;
	int32_t m[2] = { h->mis[0], h->mis[1] };
	if (h->nmis == 2 && m[1] >= L - 13) {
		int32_t t = m[0];
		m[0] = m[1];
		m[1] = t;
	}
	for (int k = 0; k < h->nmis; k++) {
		int q = (h->nmis == 1 && h->strand && m[k] == 0) ? -64 : 40;
		/* the ELF keeps the offset in 8 bits and reads the read base through it */
		obuf_printf(out, "\t%c->%d%c%d", k_letters[rs[m[k]]], m[k] & 255, k_letters[rd[m[k] & 255]], q);
	}
	obuf_printf(out, "\t%dM\t", L);
	int32_t run = 0, first = 1;
	for (int32_t k = 0; k < L; k++) {
		if (rd[k] == rs[k]) {
			run++;
			continue;
		}
		if (first || run > 0)
			obuf_printf(out, "%d", run);
		obuf_put(out, &k_letters[rs[k]], 1);
		run = 0;
		first = 0;
	}
	if (run > 0 || repeat != 1 || first)
		obuf_printf(out, "%d", run);
	obuf_puts(out, "\n");
}

int o_soap_files(const char *reads_fa, const char *ref_fa, const char *out_path, const char *unmapped_or_null,
		 const o_soap_opts *opt)
{
	if (opt->match_mode != 4 && (opt->match_mode < 0 || opt->match_mode > 2))
		return -2; /* -M 4 (the reference's documented call, README.md:134) and -M 0 / 1 / 2 (soap.man:73-82) */
	o_seqset ref, reads;
	if (o_seqset_read_fasta(&ref, ref_fa) < 0)
		return -1;
	if (o_seqset_read_fasta(&reads, reads_fa) < 0) {
		o_seqset_free(&ref);
		return -1;
	}
	/* segments: maximal pieces of each sequence between runs of >= 10 non-ACGT letters */
	size_t nseg = 0, segcap = (size_t)ref.nseq + 16;
	int64_t *seg_lo = (int64_t *)malloc(segcap * sizeof(int64_t));
	int64_t *seg_hi = (int64_t *)malloc(segcap * sizeof(int64_t));
	for (int64_t si = 0; si < ref.nseq; si++) {
		int64_t a = ref.off[si], e = ref.off[si + 1], start = a;
		for (int64_t k = a; k <= e;) {
			int64_t r = k;
			while (r < e && ref.base[r] >= 4)
				r++;
			if (k == e || r - k >= 10) {
				if (k > start) {
					if (nseg == segcap) {
						segcap *= 2;
						seg_lo = (int64_t *)realloc(seg_lo, segcap * sizeof(int64_t));
						seg_hi = (int64_t *)realloc(seg_hi, segcap * sizeof(int64_t));
					}
					seg_lo[nseg] = start;
					seg_hi[nseg] = k;
					nseg++;
				}
				start = r;
			}
			k = r > k ? r : k + 1;
		}
	}
	for (int64_t k = 0; k < ref.total; k++)
		if (ref.base[k] >= 4)
			ref.base[k] = 2;

	/* 9-mer index of the reference */
	const uint32_t nb = 1u << (2 * SEED_K);
	int32_t *cnt = (int32_t *)calloc(nb + 1, sizeof(int32_t));
	for (int64_t p = 0; p + SEED_K <= ref.total; p++) {
		uint32_t w = 0;
		for (int k = 0; k < SEED_K; k++)
			w = (w << 2) | ref.base[p + k];
		cnt[w + 1]++;
	}
	for (uint32_t w = 0; w < nb; w++)
		cnt[w + 1] += cnt[w];
	int32_t *post = (int32_t *)malloc(((size_t)ref.total + 1) * sizeof(int32_t));
	int32_t *fill = (int32_t *)malloc((size_t)nb * sizeof(int32_t));
	memcpy(fill, cnt, (size_t)nb * sizeof(int32_t));
	for (int64_t p = 0; p + SEED_K <= ref.total; p++) {
		uint32_t w = 0;
		for (int k = 0; k < SEED_K; k++)
			w = (w << 2) | ref.base[p + k];
		post[fill[w]++] = (int32_t)p;
	}
	free(fill);

	obuf out, unm;
	obuf_init(&out);
	obuf_init(&unm);
	size_t hcap = 1024;
	shit *hits = (shit *)malloc(hcap * sizeof(shit));
	for (int64_t ri = 0; ri < reads.nseq; ri++) {
		int64_t o = reads.off[ri];
		int32_t L = (int32_t)(reads.off[ri + 1] - o);
		char name[512];
		o_seq_id(reads.header[ri], name, sizeof name);
		uint8_t *fw = (uint8_t *)malloc((size_t)L + 1), *rv = (uint8_t *)malloc((size_t)L + 1);
		int nn = 0;
		for (int32_t k = 0; k < L; k++) {
			uint8_t b = reads.base[o + k];
			if (b >= 4) {
				nn++;
				b = 2;
			}
			fw[k] = b;
		}
		for (int32_t k = 0; k < L; k++)
			rv[k] = (uint8_t)(3 - fw[L - 1 - k]);
		size_t nh = 0;
		if (L >= 27 && nn <= opt->max_n) {
			for (int st = 0; st < 2; st++) {
				const uint8_t *rd = st ? rv : fw;
				/* <= 2 mismatches leave one of three disjoint 9-mers intact */
				int32_t so[3] = { 0, L / 3, 2 * (L / 3) };
				for (int sk = 0; sk < 3; sk++) {
					uint32_t w = 0;
					for (int k = 0; k < SEED_K; k++)
						w = (w << 2) | rd[so[sk] + k];
					for (int32_t e = cnt[w]; e < cnt[w + 1]; e++) {
						int64_t gp = (int64_t)post[e] - so[sk];
						if (gp < 0)
							continue;
						/* subject of gp */
						int64_t lo = 0, hi = ref.nseq;
						while (hi - lo > 1) {
							int64_t mid = (lo + hi) / 2;
							if (ref.off[mid] <= gp)
								lo = mid;
							else
								hi = mid;
						}
						int32_t pos = (int32_t)(gp - ref.off[lo]);
						int32_t slen = (int32_t)(ref.off[lo + 1] - ref.off[lo]);
						(void)slen;
						/* segment containing gp; the hit must end before its last base */
						size_t sl = 0, sh = nseg;
						while (sh - sl > 1) {
							size_t mid = (sl + sh) / 2;
							if (seg_lo[mid] <= gp)
								sl = mid;
							else
								sh = mid;
						}
						if (nseg == 0 || gp < seg_lo[sl] || gp + L >= seg_hi[sl])
							continue;
						const uint8_t *rs = ref.base + gp;
						shit h = { (int32_t)lo, pos, st, 0, { 0, 0 } };
						int bad = 0;
						for (int32_t k = 0; k < L; k++)
							if (rd[k] != rs[k]) {
								if (h.nmis == 2) {
									bad = 1;
									break;
								}
								h.mis[h.nmis++] = k;
							}
						if (bad)
							continue;
						/* the same placement can be reached through several seeds */
						int dup = 0;
						for (size_t x = 0; x < nh; x++)
							if (hits[x].subject == h.subject && hits[x].pos == h.pos &&
							    hits[x].strand == h.strand)
								dup = 1;
						if (dup)
							continue;
						if (nh == hcap) {
							hcap *= 2;
							hits = (shit *)realloc(hits, hcap * sizeof(shit));
						}
						hits[nh++] = h;
					}
				}
			}
		}
		/* -M 4: the placements with the fewest mismatches; -M 0 / 1 / 2 (observed on the ELF): the placements with
		 * EXACTLY that many, whether or not a better one exists */
		int best = 3;
		if (opt->match_mode == 4) {
			for (size_t x = 0; x < nh; x++)
				if (hits[x].nmis < best)
					best = hits[x].nmis;
		} else {
			best = opt->match_mode;
		}
		size_t nbest = 0;
		for (size_t x = 0; x < nh; x++)
			if (hits[x].nmis == best)
				hits[nbest++] = hits[x];
		qsort(hits, nbest, sizeof(shit), cmp_shit);
		int printed = 0;
		if (nbest > 0 && !(opt->repeat == 0 && nbest > 1)) {
			size_t lim = opt->repeat == 2 ? nbest : 1;
			for (size_t x = 0; x < lim; x++)
			{
				/* -t (soap.man:48): the read's 0-based ordinal in the file instead of its name (observed on the ELF) */
				char idn[32];
				snprintf(idn, sizeof idn, "%lld", (long long)ri);
				format_row(&out, opt->id_only ? idn : name, hits[x].strand ? rv : fw, L, &hits[x], (int)nbest, &ref,
					   opt->repeat);
			}
			printed = 1;
		}
		if (!printed && !(nbest > 1)) {
			obuf_printf(&unm, ">%s\n", name);
			for (int32_t k = 0; k < L; k++)
				obuf_put(&unm, &k_letters[fw[k]], 1);
			obuf_puts(&unm, "\n");
		}
		free(fw);
		free(rv);
	}
	int rc = obuf_write_file(&out, out_path);
	if (unmapped_or_null && *unmapped_or_null)
		obuf_write_file(&unm, unmapped_or_null);
	obuf_free(&out);
	obuf_free(&unm);
	free(hits);
	free(seg_lo);
	free(seg_hi);
	free(cnt);
	free(post);
	o_seqset_free(&ref);
	o_seqset_free(&reads);
	return rc;
}
