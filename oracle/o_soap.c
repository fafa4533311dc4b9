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

/* paired-end runs print the two entries of a two-mismatch row in another order than single-end runs do (both observed on the
 * ELF): single-end: descending when one lies in the last 13 bases; paired-end: descending when one lies at or behind offset
 * 2 s, s = 7 for reads under 32 bases, 10 under 39, else a third of the length (rounded down); tests/golden/soap/pe_sweep_* holds
 * the sweep of lengths 27-120 that pins the three ranges */
static int pe_desc_from(int32_t L) { return 2 * (L < 32 ? 7 : (L < 39 ? 10 : L / 3)); }

static void format_row(obuf *out, const char *name, const uint8_t *rd, int32_t L, const shit *h, int nbest,
		       const o_seqset *ref, int repeat, char mate, int pe)
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
	obuf_printf(out, "\t%d\t%c\t%d\t%c\t%s\t%d\t%d", nbest, mate, L, h->strand ? '-' : '+', id, h->pos + 1, h->nmis);
	const uint8_t *rs = ref->base + ref->off[h->subject] + h->pos;
	int32_t m[2] = { h->mis[0], h->mis[1] };
	if (h->nmis == 2 && m[1] >= (pe ? pe_desc_from(L) : L - 13)) {
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

/* the reference as the ELF's index sees it: non-ACGT letters read as G, the segments between runs of >= 10 of them, a 9-mer
 * index over the folded letters */
typedef struct {
	o_seqset ref;
	size_t nseg;
	int64_t *seg_lo, *seg_hi;
	int32_t *cnt, *post;
} soap_index;

static void soap_index_free(soap_index *ix)
{
	free(ix->seg_lo);
	free(ix->seg_hi);
	free(ix->cnt);
	free(ix->post);
	o_seqset_free(&ix->ref);
}

static int soap_index_build(soap_index *ix, const char *ref_fa)
{
	memset(ix, 0, sizeof *ix);
	if (o_seqset_read_fasta(&ix->ref, ref_fa) < 0)
		return -1;
	o_seqset *ref = &ix->ref;
	/* segments: maximal pieces of each sequence between runs of >= 10 non-ACGT letters */
	size_t nseg = 0, segcap = (size_t)ref->nseq + 16;
	int64_t *seg_lo = (int64_t *)malloc(segcap * sizeof(int64_t));
	int64_t *seg_hi = (int64_t *)malloc(segcap * sizeof(int64_t));
	for (int64_t si = 0; si < ref->nseq; si++) {
		int64_t a = ref->off[si], e = ref->off[si + 1], start = a;
		for (int64_t k = a; k <= e;) {
			int64_t r = k;
			while (r < e && ref->base[r] >= 4)
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
	for (int64_t k = 0; k < ref->total; k++)
		if (ref->base[k] >= 4)
			ref->base[k] = 2;
	/* 9-mer index of the reference */
	const uint32_t nb = 1u << (2 * SEED_K);
	int32_t *cnt = (int32_t *)calloc(nb + 1, sizeof(int32_t));
	for (int64_t p = 0; p + SEED_K <= ref->total; p++) {
		uint32_t w = 0;
		for (int k = 0; k < SEED_K; k++)
			w = (w << 2) | ref->base[p + k];
		cnt[w + 1]++;
	}
	for (uint32_t w = 0; w < nb; w++)
		cnt[w + 1] += cnt[w];
	int32_t *post = (int32_t *)malloc(((size_t)ref->total + 1) * sizeof(int32_t));
	int32_t *fill = (int32_t *)malloc((size_t)nb * sizeof(int32_t));
	memcpy(fill, cnt, (size_t)nb * sizeof(int32_t));
	for (int64_t p = 0; p + SEED_K <= ref->total; p++) {
		uint32_t w = 0;
		for (int k = 0; k < SEED_K; k++)
			w = (w << 2) | ref->base[p + k];
		post[fill[w]++] = (int32_t)p;
	}
	free(fill);
	ix->nseg = nseg;
	ix->seg_lo = seg_lo;
	ix->seg_hi = seg_hi;
	ix->cnt = cnt;
	ix->post = post;
	return 0;
}

/* a read as the ELF sees it: non-ACGT letters read as G (nn = how many there were), both strands */
static void soap_read_strands(const o_seqset *reads, int64_t ri, uint8_t **fw_out, uint8_t **rv_out, int32_t *L_out, int *nn_out)
{
	const int64_t o = reads->off[ri];
	const int32_t L = (int32_t)(reads->off[ri + 1] - o);
	uint8_t *fw = (uint8_t *)malloc((size_t)L + 1), *rv = (uint8_t *)malloc((size_t)L + 1);
	int nn = 0;
	for (int32_t k = 0; k < L; k++) {
		uint8_t b = reads->base[o + k];
		if (b >= 4) {
			nn++;
			b = 2;
		}
		fw[k] = b;
	}
	for (int32_t k = 0; k < L; k++)
		rv[k] = (uint8_t)(3 - fw[L - 1 - k]);
	*fw_out = fw;
	*rv_out = rv;
	*L_out = L;
	*nn_out = nn;
}

/* every placement of the read (either strand) with at most two mismatches: full length, ungapped, inside one segment and
 * ending before the segment's last base; appended to *hits (grown as needed), returns their number */
static size_t soap_enum(const soap_index *ix, const uint8_t *fw, const uint8_t *rv, int32_t L, shit **hits_io, size_t *hcap_io, int pe)
{
	const o_seqset *ref = &ix->ref;
	shit *hits = *hits_io;
	size_t hcap = *hcap_io, nh = 0;
	for (int st = 0; st < 2; st++) {
		const uint8_t *rd = st ? rv : fw;
		/* <= 2 mismatches leave one of three disjoint 9-mers intact */
		int32_t so[3] = { 0, L / 3, 2 * (L / 3) };
		for (int sk = 0; sk < 3; sk++) {
			uint32_t w = 0;
			for (int k = 0; k < SEED_K; k++)
				w = (w << 2) | rd[so[sk] + k];
			for (int32_t e = ix->cnt[w]; e < ix->cnt[w + 1]; e++) {
				int64_t gp = (int64_t)ix->post[e] - so[sk];
				if (gp < 0)
					continue;
				/* subject of gp */
				int64_t lo = 0, hi = ref->nseq;
				while (hi - lo > 1) {
					int64_t mid = (lo + hi) / 2;
					if (ref->off[mid] <= gp)
						lo = mid;
					else
						hi = mid;
				}
				int32_t pos = (int32_t)(gp - ref->off[lo]);
				/* segment containing gp; the hit must end before its last base */
				size_t sl = 0, sh = ix->nseg;
				while (sh - sl > 1) {
					size_t mid = (sl + sh) / 2;
					if (ix->seg_lo[mid] <= gp)
						sl = mid;
					else
						sh = mid;
				}
				if (ix->nseg == 0 || gp < ix->seg_lo[sl] || gp + L >= ix->seg_hi[sl])
					continue;
				const uint8_t *rs = ref->base + gp;
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
				/* observed on the ELF, paired-end runs only: a read of exactly 32 bases is not placed where it has two
				 * mismatches that both lie in its first 20 bases (33 to 38 bases and 27 to 31: placed) */
				if (pe && L == 32 && h.nmis == 2 && h.mis[1] < 20)
					continue;
				/* the same placement can be reached through several seeds */
				int dup = 0;
				for (size_t x = 0; x < nh; x++)
					if (hits[x].subject == h.subject && hits[x].pos == h.pos && hits[x].strand == h.strand)
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
	*hits_io = hits;
	*hcap_io = hcap;
	return nh;
}

static void soap_unmapped(obuf *unm, const char *name, const uint8_t *fw, int32_t L)
{
	obuf_printf(unm, ">%s\n", name);
	for (int32_t k = 0; k < L; k++)
		obuf_put(unm, &k_letters[fw[k]], 1);
	obuf_puts(unm, "\n");
}

int o_soap_files(const char *reads_fa, const char *ref_fa, const char *out_path, const char *unmapped_or_null,
		 const o_soap_opts *opt)
{
	if (opt->match_mode != 4 && (opt->match_mode < 0 || opt->match_mode > 2))
		return -2; /* -M 4 (the reference's documented call, README.md:134) and -M 0 / 1 / 2 (soap.man:73-82) */
	soap_index ix;
	o_seqset reads;
	if (soap_index_build(&ix, ref_fa) < 0)
		return -1;
	if (o_seqset_read_fasta(&reads, reads_fa) < 0) {
		soap_index_free(&ix);
		return -1;
	}
	obuf out, unm;
	obuf_init(&out);
	obuf_init(&unm);
	size_t hcap = 1024;
	shit *hits = (shit *)malloc(hcap * sizeof(shit));
	for (int64_t ri = 0; ri < reads.nseq; ri++) {
		char name[512];
		o_seq_id(reads.header[ri], name, sizeof name);
		uint8_t *fw, *rv;
		int32_t L;
		int nn;
		soap_read_strands(&reads, ri, &fw, &rv, &L, &nn);
		size_t nh = 0;
		if (L >= 27 && nn <= opt->max_n)
			nh = soap_enum(&ix, fw, rv, L, &hits, &hcap, 0);
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
				format_row(&out, opt->id_only ? idn : name, hits[x].strand ? rv : fw, L, &hits[x], (int)nbest, &ix.ref,
					   opt->repeat, 'a', 0);
			}
			printed = 1;
		}
		if (!printed && !(nbest > 1))
			soap_unmapped(&unm, name, fw, L);
		free(fw);
		free(rv);
	}
	int rc = obuf_write_file(&out, out_path);
	if (unmapped_or_null && *unmapped_or_null)
		obuf_write_file(&unm, unmapped_or_null);
	obuf_free(&out);
	obuf_free(&unm);
	free(hits);
	soap_index_free(&ix);
	o_seqset_free(&reads);
	return rc;
}

/* a valid pair of placements: one subject, opposite strands, both with at most `level` mismatches, and (position of the '-'
 * mate - position of the '+' mate) + the A mate's length inside [min_ins, max_ins] */
static int soap_pair_ok(const shit *a, const shit *b, int32_t La, int level, int min_ins, int max_ins)
{
	if (a->subject != b->subject || a->strand == b->strand || a->nmis > level || b->nmis > level)
		return 0;
	const shit *plus = a->strand ? b : a, *minus = a->strand ? a : b;
	const int ins = minus->pos - plus->pos + La;
	return ins >= min_ins && ins <= max_ins;
}

/* Paired-end mode, `soap -a A -b B -D ref.index -o paired -2 unpaired [-u unmapped] -m MIN -x MAX` (soap.man:29-50).  The
 * reference's pipeline never calls it (README.md:134 is single-ended); every rule below was observed by running the ELF
 * (oracle/gen_goldens_soap.py, paired-end part) and is pinned by tests/golden/soap/pe_*:
 *   - pair i = read i of A with read i of B; a mate shorter than 27 bases is outside what is restated (the ELF places
 *     some of them beside their partner and crashes on others), and so is one above 256 bases (-l; not observed): such input
 *     is refused (-3);
 *   - pairs are looked for among the placements with at most k mismatches of either mate, k = 0, then 1, then 2: the first
 *     k that gives a valid pair decides, and all valid pairs at that k are the pair's result (with a mate whose best
 *     placement has two mismatches that is every placement of both mates);
 *   - a pair of placements is valid when they lie on one subject, one mate on '+' and the other on '-', and
 *     (position of the '-' mate - position of the '+' mate) + LENGTH OF THE A MATE lies in [MIN, MAX] (both included) --
 *     the A mate's length whichever strand it is on: with mates of 40 and 100 bases true inserts of 460..660 are taken
 *     when A is the short mate and 400..600 when A is the long one;
 *   - -r 2: all valid pairs, the A rows first, then the B rows in the same order, column 4 = the number of pairs, column 5
 *     = a / b; -r 1: one pair (the ELF's pick is not reproducible: here the first in (subject, position) order of the A
 *     placement, then of the B placement); -r 0: the pair only if it is the only one;
 *   - no valid pair: every placement of either mate goes to the unpaired file (-2) as single-end rows (column 4 = the
 *     mate's number of placements; -r 1: one of them; -r 0: only a unique one); a mate without a placement -- or, under
 *     -r 0, with several, or a pair with several valid pairs -- goes to the unmapped file;
 *   - the order of rows inside a read pair is the ELF's index order and is not restated: rows are compared as sets. */
int o_soap_pe_files(const char *a_fa, const char *b_fa, const char *ref_fa, const char *out_path, const char *unpaired_path,
		    const char *unmapped_or_null, const o_soap_opts *opt, int min_ins, int max_ins)
{
	soap_index ix;
	o_seqset ra, rb;
	if (soap_index_build(&ix, ref_fa) < 0)
		return -1;
	if (o_seqset_read_fasta(&ra, a_fa) < 0) {
		soap_index_free(&ix);
		return -1;
	}
	if (o_seqset_read_fasta(&rb, b_fa) < 0) {
		soap_index_free(&ix);
		o_seqset_free(&ra);
		return -1;
	}
	obuf out, un2, unm;
	obuf_init(&out);
	obuf_init(&un2);
	obuf_init(&unm);
	size_t capa = 1024, capb = 1024;
	shit *ha = (shit *)malloc(capa * sizeof(shit)), *hb = (shit *)malloc(capb * sizeof(shit));
	int rc = 0;
	const int64_t n = ra.nseq < rb.nseq ? ra.nseq : rb.nseq;
	for (int64_t ri = 0; ri < n && rc == 0; ri++) {
		char na[512], nb[512];
		o_seq_id(ra.header[ri], na, sizeof na);
		o_seq_id(rb.header[ri], nb, sizeof nb);
		uint8_t *fa, *va, *fb, *vb;
		int32_t La, Lb;
		int nna, nnb;
		soap_read_strands(&ra, ri, &fa, &va, &La, &nna);
		soap_read_strands(&rb, ri, &fb, &vb, &Lb, &nnb);
		if (La < 27 || Lb < 27 || La > 256 || Lb > 256) {
			rc = -3;
		} else {
			size_t nha = nna <= opt->max_n ? soap_enum(&ix, fa, va, La, &ha, &capa, 1) : 0;
			size_t nhb = nnb <= opt->max_n ? soap_enum(&ix, fb, vb, Lb, &hb, &capb, 1) : 0;
			qsort(ha, nha, sizeof(shit), cmp_shit);
			qsort(hb, nhb, sizeof(shit), cmp_shit);
			/* the pairs: among the placements with at most k mismatches (each mate), for the smallest k = 0, 1, 2 that
			 * gives a valid pair at all */
			size_t npairs = 0;
			int level = 0;
			for (; level <= 2 && npairs == 0; level++)
				for (size_t x = 0; x < nha; x++)
					for (size_t y = 0; y < nhb; y++)
						npairs += soap_pair_ok(&ha[x], &hb[y], La, level, min_ins, max_ins);
			level--;
			for (int mate = 0; mate < 2 && npairs > 0 && !(opt->repeat == 0 && npairs > 1); mate++) {
				size_t seen = 0; /* A rows first, then the B rows of the same pairs in the same order */
				for (size_t x = 0; x < nha; x++)
					for (size_t y = 0; y < nhb; y++) {
						if (!soap_pair_ok(&ha[x], &hb[y], La, level, min_ins, max_ins))
							continue;
						if (opt->repeat != 2 && seen++ > 0)
							continue;
						if (mate == 0)
							format_row(&out, na, ha[x].strand ? va : fa, La, &ha[x], (int)npairs, &ix.ref, opt->repeat, 'a', 1);
						else
							format_row(&out, nb, hb[y].strand ? vb : fb, Lb, &hb[y], (int)npairs, &ix.ref, opt->repeat, 'b', 1);
					}
			}
			if (npairs == 0 || (opt->repeat == 0 && npairs > 1)) {
				/* unpaired: the mates on their own */
				for (int mate = 0; mate < 2; mate++) {
					const shit *h = mate ? hb : ha;
					const size_t nh = mate ? nhb : nha;
					const char *nm = mate ? nb : na;
					const uint8_t *fw = mate ? fb : fa, *rv = mate ? vb : va;
					const int32_t L = mate ? Lb : La;
					const int alone = npairs == 0; /* (a pair dropped by -r 0 for having several pairings: both mates unmapped) */
					if (alone && nh > 0 && !(opt->repeat == 0 && nh > 1)) {
						const size_t lim = opt->repeat == 2 ? nh : 1;
						for (size_t x = 0; x < lim; x++)
							format_row(&un2, nm, h[x].strand ? rv : fw, L, &h[x], (int)nh, &ix.ref, opt->repeat, mate ? 'b' : 'a', 1);
					} else {
						soap_unmapped(&unm, nm, fw, L);
					}
				}
			}
		}
		free(fa);
		free(va);
		free(fb);
		free(vb);
	}
	if (rc == 0)
		rc = obuf_write_file(&out, out_path);
	if (rc == 0 && unpaired_path && *unpaired_path)
		rc = obuf_write_file(&un2, unpaired_path);
	if (rc == 0 && unmapped_or_null && *unmapped_or_null)
		obuf_write_file(&unm, unmapped_or_null);
	obuf_free(&out);
	obuf_free(&un2);
	obuf_free(&unm);
	free(ha);
	free(hb);
	soap_index_free(&ix);
	o_seqset_free(&ra);
	o_seqset_free(&rb);
	return rc;
}
