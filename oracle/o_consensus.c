/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * CPU restatement of Consensus/Consensus_BLAST_SOAP_RDP-1.1.pl: per RDP read, the
 * BLAST(+lineage) hit with the most (rank,name) pairs equal to the RDP assignment,
 * with the script's order-dependent STRING comparisons (gt/lt/eq) as tie-break.
 * Pinned by tests/golden/consensus/.
 */
#include "o_common.h"
#include <stdlib.h>
#include <string.h>

typedef struct {
	const char *p;
	size_t n;
} sv; /* string view; p == NULL means Perl undef (stringifies to "") */

static int sv_eq(sv a, sv b)
{
	return a.n == b.n && (a.n == 0 || memcmp(a.p, b.p, a.n) == 0);
}

/* Perl `cmp` on byte strings */
static int sv_cmp(sv a, sv b)
{
	size_t m = a.n < b.n ? a.n : b.n;
	int c = m ? memcmp(a.p, b.p, m) : 0;
	if (c)
		return c;
	return a.n < b.n ? -1 : (a.n > b.n ? 1 : 0);
}

static sv sv_int(char buf[24], long v)
{
	int n = snprintf(buf, 24, "%ld", v);
	sv s = { buf, (size_t)n };
	return s;
}

/* split(/\t\t|\t/, line) with Perl's trailing-empty-field removal (Consensus:110) */
static int split_tabs(const char *line, size_t len, sv *f, int maxf)
{
	int n = 0;
	size_t s = 0, i = 0;
	if (len == 0)
		return 0;
	while (i < len) {
		if (line[i] == '\t') {
			if (n < maxf) {
				f[n].p = line + s;
				f[n].n = i - s;
				n++;
			}
			i += (i + 1 < len && line[i + 1] == '\t') ? 2 : 1;
			s = i;
		} else {
			i++;
		}
	}
	if (n < maxf) {
		f[n].p = line + s;
		f[n].n = len - s;
		n++;
	}
	while (n > 0 && f[n - 1].n == 0)
		n--;
	return n;
}

/* split(/\t/, s), trailing empties removed (Consensus:132) */
static int split_tab1(const char *s, size_t len, sv *f, int maxf)
{
	int n = 0;
	size_t st = 0;
	if (len == 0)
		return 0;
	for (size_t i = 0; i <= len; i++) {
		if (i == len || s[i] == '\t') {
			if (n < maxf) {
				f[n].p = s + st;
				f[n].n = i - st;
				n++;
			}
			st = i + 1;
		}
	}
	while (n > 0 && f[n - 1].n == 0)
		n--;
	return n;
}

static int is_ws(char c)
{
	return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v';
}

/* Consensus:116-122: split on '[', ']', ';', join with ' ', split on whitespace */
static int lineage_tokens(sv tax, sv *tok, int maxt)
{
	int n = 0;
	size_t i = 0;
	while (i < tax.n) {
		while (i < tax.n && (is_ws(tax.p[i]) || tax.p[i] == '[' || tax.p[i] == ']' ||
				     tax.p[i] == ';'))
			i++;
		size_t s = i;
		while (i < tax.n && !(is_ws(tax.p[i]) || tax.p[i] == '[' || tax.p[i] == ']' ||
				      tax.p[i] == ';'))
			i++;
		if (i > s && n < maxt) {
			tok[n].p = tax.p + s;
			tok[n].n = i - s;
			n++;
		}
	}
	return n;
}

/* Consensus:159-160: strip '"' and '\', then every [\W\d_]  => ASCII letters stay */
static size_t clean_rdp_name(sv in, char *out)
{
	size_t k = 0;
	for (size_t i = 0; i < in.n; i++) {
		char c = in.p[i];
		if ((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))
			out[k++] = c;
	}
	return k;
}

static int blast_rank_index(sv t)
{
	/* Consensus:74-82,164-166: keys "0".."6"; anything else is undef (-1) */
	if (t.p && t.n == 1 && t.p[0] >= '0' && t.p[0] <= '6')
		return t.p[0] - '0';
	return -1;
}

static int rdp_rank_index(sv t)
{
	/* Consensus:64-72,168-170 */
	static const char *const r[7] = { "domain", "phylum", "class",  "order",
					  "family", "genus",  "species" };
	if (!t.p)
		return -1;
	for (int i = 0; i < 7; i++)
		if (strlen(r[i]) == t.n && memcmp(r[i], t.p, t.n) == 0)
			return i;
	return -1;
}

#define MAXTOK 512

int o_consensus_buf(const char *blast, size_t blast_len, const char *rdp, size_t rdp_len,
		    obuf *out, obuf *log)
{
	/* @read1 = <READ1> (Consensus:88) */
	size_t nlines = 0, cap = 1024;
	sv *lines = (sv *)malloc(cap * sizeof(sv));
	for (size_t s = 0; s < blast_len;) {
		const char *eol = memchr(blast + s, '\n', blast_len - s);
		size_t e = eol ? (size_t)(eol - blast) : blast_len;
		if (nlines == cap) {
			cap *= 2;
			lines = (sv *)realloc(lines, cap * sizeof(sv));
		}
		lines[nlines].p = blast + s;
		lines[nlines].n = e - s; /* chomped */
		nlines++;
		s = eol ? e + 1 : blast_len;
	}

	size_t i = 0;
	int found = -1; /* undef */
	long maxblastcount = 0, maxrankmatches = 0;
	sv tempresult = { NULL, 0 };
	sv blastsim = { NULL, 0 };
	char simbuf[24], b1[24], b2[24];
	char *clean = (char *)malloc(rdp_len + 1);
	int rc = 0;

	for (size_t rs = 0; rs < rdp_len && rc == 0;) {
		const char *eol = memchr(rdp + rs, '\n', rdp_len - rs);
		size_t re = eol ? (size_t)(eol - rdp) : rdp_len;
		sv rdpline = { rdp + rs, re - rs };
		rs = eol ? re + 1 : rdp_len;

		/* split(/\t\t\t\t\t/, rdpline) (Consensus:126) */
		sv rid = rdpline, rrest = { NULL, 0 };
		for (size_t k = 0; k + 5 <= rdpline.n; k++)
			if (memcmp(rdpline.p + k, "\t\t\t\t\t", 5) == 0) {
				rid.n = k;
				size_t st = k + 5, en = rdpline.n;
				/* a further 5-tab separator ends element [1] */
				for (size_t q = st; q + 5 <= rdpline.n; q++)
					if (memcmp(rdpline.p + q, "\t\t\t\t\t", 5) == 0) {
						en = q;
						break;
					}
				rrest.p = rdpline.p + st;
				rrest.n = en - st;
				break;
			}
		sv rt[MAXTOK];
		int nr = rrest.p ? split_tab1(rrest.p, rrest.n, rt, MAXTOK) : 0;
		/* clean every name slot once (idempotent in the reference) */
		sv rname[MAXTOK];
		size_t coff = 0;
		for (int b = 0; b < nr; b += 3) {
			size_t k = clean_rdp_name(rt[b], clean + coff);
			rname[b].p = clean + coff;
			rname[b].n = k;
			coff += k;
		}

		for (;;) { /* GETBLAST (Consensus:103) */
			sv bline = { NULL, 0 };
			if (i < nlines)
				bline = lines[i];
			sv bf[64];
			int nbf = bline.p ? split_tabs(bline.p, bline.n, bf, 64) : 0;
			sv bid = { NULL, 0 }, btax = { NULL, 0 }, bsim = { NULL, 0 };
			if (nbf > 0)
				bid = bf[0];
			if (nbf > 1)
				btax = bf[1];
			if (nbf > 2)
				bsim = bf[2];

			if (sv_eq(bid, rid)) {
				if (i >= nlines) {
					rc = -1; /* empty RDP id after the table ends: never terminates */
					break;
				}
				found = 1;
				sv bt[MAXTOK];
				int nb = lineage_tokens(btax, bt, MAXTOK);
				long rankmatches = 0;
				for (int a = 0; a < nb; a += 2) {
					sv bname = { NULL, 0 };
					if (a + 1 < nb)
						bname = bt[a + 1];
					int i1 = blast_rank_index(bt[a]);
					for (int b = 0; b < nr; b += 3) {
						sv rrank = { NULL, 0 };
						if (b + 1 < nr)
							rrank = rt[b + 1];
						int i2 = rdp_rank_index(rrank);
						if (sv_eq(bname, rname[b]) && i1 == i2)
							rankmatches++;
					}
				}
				long blastcount = nb;
				sv s_rm = sv_int(b1, rankmatches);
				sv s_max = sv_int(b2, maxrankmatches);
				if (sv_cmp(s_rm, s_max) > 0) { /* gt (Consensus:191) */
					maxrankmatches = rankmatches;
					tempresult = bline;
					blastsim = bsim;
				}
				char c1[24], c2[24];
				sv s_bc = sv_int(c1, blastcount), s_mbc = sv_int(c2, maxblastcount);
				s_max = sv_int(b2, maxrankmatches);
				if ((sv_cmp(s_bc, s_mbc) > 0 || sv_cmp(blastsim, bsim) < 0) &&
				    sv_eq(s_rm, s_max)) { /* Consensus:199 */
					maxblastcount = blastcount;
					tempresult = bline;
					blastsim = bsim;
				}
				i++;
				continue;
			}
			if (found == 0) {
				/* Consensus:216-220 */
				obuf_puts(log, "not found: ");
				obuf_put(log, bid.p ? bid.p : "", bid.n);
				obuf_puts(log, "\t ");
				obuf_put(log, rid.p, rid.n);
				obuf_puts(log, "\n");
				if (i >= nlines) {
					rc = -1; /* hang #2: cursor runs past the table forever */
					break;
				}
				i++;
				continue;
			}
			if (found == 1) {
				/* Consensus:223-234 */
				obuf_put(out, tempresult.p ? tempresult.p : "", tempresult.n);
				obuf_puts(out, "\n");
				obuf_printf(out, "#Matches found: %ld\n", maxrankmatches);
				found = 0;
				maxblastcount = 0;
				maxrankmatches = 0;
				simbuf[0] = '0';
				blastsim.p = simbuf;
				blastsim.n = 1;
			}
			break; /* next RDP line */
		}
	}
	free(clean);
	free(lines);
	return rc;
}

int o_consensus_file(const char *b, const char *r, const char *s_or_null, const char *o, obuf *log)
{
	/* Consensus:19-55: banner, open checks in order b, r, s, then echo of -o */
	obuf_puts(log, "\nLoading input files...\n");
	size_t bl, rl;
	char *bb = o_read_file(b, &bl);
	if (!bb) {
		obuf_printf(log, "Error: Unable to open %s file.\n", b);
		return 0;
	}
	char *rb = o_read_file(r, &rl);
	if (!rb) {
		obuf_printf(log, "Error: Unable to open %s file.\n", r);
		free(bb);
		return 0;
	}
	if (s_or_null && *s_or_null) {
		FILE *f = fopen(s_or_null, "r"); /* opened, never read (Consensus:40-46) */
		if (!f) {
			obuf_printf(log, "Error: Unable to open %s file.\n", s_or_null);
			free(bb);
			free(rb);
			return 0;
		}
		fclose(f);
	}
	obuf_printf(log, "%s\n", o);
	obuf out;
	obuf_init(&out);
	int rc = o_consensus_buf(bb, bl, rb, rl, &out, log);
	if (obuf_write_file(&out, o) < 0) {
		obuf_printf(log, "Error: Unable to open output file %s.\n", o);
		rc = 0;
	} else if (rc == 0) {
		obuf_puts(log, "\nDone!\n");
	}
	obuf_free(&out);
	free(bb);
	free(rb);
	return rc;
}


/* ------------------------------------------------------------------------------------------------------------------
 * Three-way vote -- an OPT-IN EXTENSION, not reference behaviour: the reference's Consensus opens the
 * SOAP stream and never reads it (Consensus_BLAST_SOAP_RDP-1.1.pl:40-46).  Spec "pgx-vote3 v1":
 *   per read of the RDP stream, three lineages as (rank index 0..6 -> name): B = the lineage of the read's FIRST row in the
 *   BLAST table after taxcollector (the best hit, S5 order); S = the same for the read's first row in the SOAP table
 *   after taxcollector; R = the RDP assignment with the script's name cleaning (Consensus:159-160) and rank names
 *   (Consensus:64-72).  Lineages are tokenised as the script does (Consensus:116-122); a rank's name is the first pair
 *   that carries its tag.  Rank k is AGREED when two of the three names are equal and not empty (B = S, else B = R,
 *   else S = R); the consensus is the longest prefix of agreed ranks.
 *   One line per RDP read: name, "[k]Name;" for the agreed prefix, its length, the votes per rank as digits.
 */
typedef struct {
	sv name[7];
} lin7;

static void lin_from_lineage(sv tax, lin7 *l)
{
	sv tok[MAXTOK];
	int n = lineage_tokens(tax, tok, MAXTOK);
	memset(l, 0, sizeof *l);
	for (int a = 0; a + 1 < n; a += 2) {
		int k = blast_rank_index(tok[a]);
		if (k >= 0 && l->name[k].p == NULL)
			l->name[k] = tok[a + 1];
	}
}

/* first row of every read in a taxcollector table: read -> lineage (column 2); rows of a read are consecutive */
typedef struct {
	sv read, lineage;
} first_row;

static size_t first_rows(const char *t, size_t len, first_row **out)
{
	size_t n = 0, cap = 1024, s = 0;
	first_row *v = (first_row *)malloc(cap * sizeof *v);
	while (s < len) {
		const char *nl = (const char *)memchr(t + s, '\n', len - s);
		size_t e = nl ? (size_t)(nl - t) : len;
		sv f[4];
		int nf = split_tabs(t + s, e - s, f, 4);
		if (nf >= 2 && (n == 0 || !sv_eq(v[n - 1].read, f[0]))) {
			if (n == cap) {
				cap *= 2;
				v = (first_row *)realloc(v, cap * sizeof *v);
			}
			v[n].read = f[0];
			v[n].lineage = f[1];
			n++;
		}
		s = e + 1;
	}
	*out = v;
	return n;
}

static const first_row *find_row(const first_row *v, size_t n, sv read, size_t *cursor)
{
	/* both tables follow the read order of the RDP stream: walk forward from the cursor, else search everything */
	for (size_t i = *cursor; i < n; i++)
		if (sv_eq(v[i].read, read)) {
			*cursor = i + 1;
			return &v[i];
		}
	for (size_t i = 0; i < *cursor && i < n; i++)
		if (sv_eq(v[i].read, read))
			return &v[i];
	return NULL;
}

int o_vote3_buf(const char *blast_class, size_t bl, const char *rdp, size_t rl, const char *soap_class, size_t sl, obuf *out)
{
	first_row *bv = NULL, *svv = NULL;
	size_t bn = first_rows(blast_class, bl, &bv), sn = first_rows(soap_class, sl, &svv), bc = 0, sc = 0, s = 0;
	static const char kFive[] = "\t\t\t\t\t";
	while (s < rl) {
		const char *nl = (const char *)memchr(rdp + s, '\n', rl - s);
		size_t e = nl ? (size_t)(nl - rdp) : rl;
		const char *line = rdp + s;
		size_t len = e - s;
		s = e + 1;
		const char *five = NULL;
		for (size_t i = 0; i + 5 <= len; i++)
			if (memcmp(line + i, kFive, 5) == 0) {
				five = line + i;
				break;
			}
		sv read = { line, five ? (size_t)(five - line) : len };
		lin7 B, S, R;
		char clean[7][256];
		memset(&B, 0, sizeof B);
		memset(&S, 0, sizeof S);
		memset(&R, 0, sizeof R);
		const first_row *fb = find_row(bv, bn, read, &bc), *fs = find_row(svv, sn, read, &sc);
		if (fb)
			lin_from_lineage(fb->lineage, &B);
		if (fs)
			lin_from_lineage(fs->lineage, &S);
		if (five) {
			sv f[64];
			int nf = split_tab1(five + 5, len - read.n - 5, f, 64);
			for (int k = 0; k + 1 < nf; k += 3) {
				int ri = rdp_rank_index(f[k + 1]);
				if (ri >= 0 && R.name[ri].p == NULL) {
					size_t cn = clean_rdp_name(f[k].n < 255 ? f[k] : (sv){ f[k].p, 255 }, clean[ri]);
					R.name[ri].p = clean[ri];
					R.name[ri].n = cn;
				}
			}
		}
		obuf_put(out, read.p, read.n);
		obuf_puts(out, "\t");
		char votes[8];
		int depth = 0;
		for (int k = 0; k < 7; k++) {
			sv b = B.name[k], ss = S.name[k], r = R.name[k], w = { NULL, 0 };
			int v = 0;
			if (b.n && ss.n && sv_eq(b, ss)) {
				w = b;
				v = 2 + (r.n && sv_eq(r, b));
			} else if (b.n && r.n && sv_eq(b, r)) {
				w = b;
				v = 2;
			} else if (ss.n && r.n && sv_eq(ss, r)) {
				w = ss;
				v = 2;
			}
			if (!v)
				break;
			obuf_printf(out, "[%d]", k);
			obuf_put(out, w.p, w.n);
			obuf_puts(out, ";");
			votes[depth++] = (char)('0' + v);
		}
		votes[depth] = 0;
		obuf_printf(out, "\t%d\t%s\n", depth, votes);
	}
	free(bv);
	free(svv);
	return 0;
}

int o_vote3_file(const char *blast_class, const char *rdp, const char *soap_class, const char *out_path)
{
	size_t bl = 0, rl = 0, sl = 0;
	char *b = o_read_file(blast_class, &bl), *r = o_read_file(rdp, &rl), *s = o_read_file(soap_class, &sl);
	int rc = -1;
	if (b && r && s) {
		obuf out;
		obuf_init(&out);
		o_vote3_buf(b, bl, r, rl, s, sl, &out);
		rc = obuf_write_file(&out, out_path);
		obuf_free(&out);
	}
	free(b);
	free(r);
	free(s);
	return rc;
}
