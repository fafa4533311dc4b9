/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * CPU restatement of Tax_class/NCBI-taxcollector-0.01.pl: per hit line, gi ->
 * lineage string "[0]..;[1]..;...[6]..;" and re-emission of the numeric columns.
 * The Perl drives ./tax_class through backticks; here the same lookups are made
 * in memory, but every text-level rule (digit tests on whole elements, '_'
 * substitution, 6->5 duplication, 7->9) is applied to the same strings in the
 * same order.  Pinned by tests/golden/taxcollect/.
 */
#include "o_common.h"
#include <stdlib.h>
#include <string.h>

/* NCBI-taxcollector-0.01.pl:228-237 */
static const char *const k_ranklist[8] = { "superkingdom", "phylum", "class",  "order",
					   "family",       "genus",  "species", "kingdom" };

static void strip_spaces(const char *in, char *out)
{
	/* taxcollector:250-252 removes tabs and blanks from the node line */
	while (*in) {
		if (*in != ' ' && *in != '\t')
			*out++ = *in;
		in++;
	}
	*out = '\0';
}

/* get_name (taxcollector:188-224): first `-n` line whose class field contains
 * "scientific name"; the name is trimmed of leading/trailing whitespace.
 * Returns 1 and appends "<name>;|" when found. */
static int push_name(const o_taxdb *db, int taxid, obuf *cls)
{
	obuf lines;
	obuf_init(&lines);
	if (o_tax_names(db, taxid, &lines) < 0) {
		obuf_free(&lines);
		return 0;
	}
	int found = 0;
	char *p = lines.p;
	while (p && *p) {
		char *eol = strchr(p, '\n');
		if (eol)
			*eol = '\0';
		/* s/\t//g then split on '|' */
		char *f[8];
		int nf = 0;
		char *q = p;
		f[nf++] = q;
		for (; *q && nf < 8; q++)
			if (*q == '|') {
				*q = '\0';
				f[nf++] = q + 1;
			}
		if (nf > 3 && strstr(f[3], "scientific name")) {
			char *s = f[1];
			char tmp[256];
			size_t k = 0;
			for (; *s && k < sizeof tmp - 1; s++)
				if (*s != '\t')
					tmp[k++] = *s;
			tmp[k] = '\0';
			char *b = tmp;
			while (*b == ' ' || *b == '\n' || *b == '\r' || *b == '\f' || *b == '\v')
				b++;
			size_t e = strlen(b);
			while (e > 0 && (b[e - 1] == ' ' || b[e - 1] == '\n' || b[e - 1] == '\r' ||
					 b[e - 1] == '\f' || b[e - 1] == '\v'))
				e--;
			b[e] = '\0';
			obuf_puts(cls, b);
			obuf_puts(cls, ";|");
			found = 1;
			break;
		}
		p = eol ? eol + 1 : NULL;
	}
	obuf_free(&lines);
	return found;
}

/* get_uptaxa (taxcollector:226-300). Returns 0 or -1 where the reference does not
 * terminate (a failed `-t` lookup feeds garbage back into the recursion). */
static int uptaxa(const o_taxdb *db, int taxid, obuf *cls, int depth)
{
	if (depth > 4096)
		return -1; /* parent cycle: the Perl recurses until it dies */
	o_node nd;
	if (o_tax_node(db, taxid, &nd) < 0)
		return -1; /* "Error." on stdout -> undefined recursion in the Perl */
	char rank[40];
	strip_spaces(o_rank_name(nd.rank), rank);
	int idx = -1;
	for (int i = 0; i < 8; i++)
		if (strcmp(rank, k_ranklist[i]) == 0)
			idx = i;
	if (idx >= 0) {
		obuf_printf(cls, "[%d]", idx);
		push_name(db, taxid, cls);
		if (idx == 0)
			return 0; /* superkingdom reached (taxcollector:275-279) */
		return uptaxa(db, nd.parent, cls, depth + 1);
	}
	if (strcmp(rank, "norank") == 0) {
		if (nd.parent == 1) {
			obuf_puts(cls, "[0]Unclassified;|"); /* taxcollector:289-290 */
			return 0;
		}
		return uptaxa(db, nd.parent, cls, depth + 1);
	}
	return uptaxa(db, nd.parent, cls, depth + 1);
}

static int has_char(const char *s, size_t n, char c)
{
	return memchr(s, c, n) != NULL;
}

static int is_perl_space(char c)
{
	return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v';
}

int o_taxcollect_lineage(const o_taxdb *db, const char *gi_text, obuf *lineage, obuf *report)
{
	if (!gi_text || !*gi_text)
		return -1; /* `./tax_class -s` without argument: hang #1 */
	int gi = atoi(gi_text);
	int taxid = 0;
	obuf cls;
	obuf_init(&cls);

	/* parse_taxonomy (taxcollector:166-186) on the text `tax_class -s` prints */
	int rc = o_tax_gi2taxid(db, gi, &taxid);
	if (rc < 0) {
		obuf_free(&cls);
		return -1; /* "Error." */
	}
	if (taxid == 0) {
		obuf_puts(report, "Searching upper node for TAXID 0\n.\n");
		obuf_printf(report, "\n\nTAXID zero GI = %s.\n\n", gi_text);
		obuf_printf(&cls, "Unidentified(GI:%s);|", gi_text);
	} else {
		/* first printed node = the leaf, unless the leaf or its parent is the
		 * root, in which case -s prints nothing (ncbitc.c:943-953) */
		o_node leaf;
		if (taxid == 1 || o_tax_node(db, taxid, &leaf) < 0 || leaf.parent == 1) {
			obuf_free(&cls);
			return -1;
		}
		/* line[0] is the tax_id field of the record found at index taxid-1 */
		obuf_printf(report, "Searching upper node for TAXID %d.\n", leaf.tax_id);
		if (uptaxa(db, leaf.tax_id, &cls, 0) < 0) {
			obuf_free(&cls);
			return -1;
		}
		obuf_printf(report, "Done for TAXID %d.\n", leaf.tax_id);
	}

	/* taxcollector:96-144: split the concatenation on '|' (trailing empty fields
	 * dropped), walk it from the last element (root side) to the first */
	const char *el[256];
	size_t eln[256];
	int n = 0;
	if (cls.n) {
		const char *s = cls.p, *end = cls.p + cls.n;
		while (s <= end && n < 256) {
			const char *bar = memchr(s, '|', (size_t)(end - s));
			if (!bar)
				bar = end;
			el[n] = s;
			eln[n] = (size_t)(bar - s);
			n++;
			if (bar == end)
				break;
			s = bar + 1;
		}
		while (n > 0 && eln[n - 1] == 0)
			n--;
	}
	int any5 = 0;
	for (int i = 0; i < n; i++)
		if (has_char(el[i], eln[i], '5'))
			any5 = 1;
	for (int i = n - 1; i >= 0; i--) {
		char tmp[512];
		size_t len = eln[i] < sizeof tmp - 1 ? eln[i] : sizeof tmp - 1;
		memcpy(tmp, el[i], len);
		tmp[len] = '\0';
		if (has_char(tmp, len, '6')) {
			for (size_t k = 0; k < len; k++)
				if (is_perl_space(tmp[k]))
					tmp[k] = '_';
			if (!any5) {
				char *six = strchr(tmp, '6');
				*six = '5';
				obuf_put(lineage, tmp, len);
				*six = '6';
				obuf_put(lineage, tmp, len);
			} else {
				obuf_put(lineage, tmp, len);
			}
		} else {
			char *sev = memchr(tmp, '7', len);
			if (sev)
				*sev = '9';
			obuf_put(lineage, tmp, len);
		}
	}
	obuf_free(&cls);
	return 0;
}

/* split(/\ |\t\t|\t/, line): alternation tried in order at each position; leading
 * empty fields kept, trailing empty fields dropped (Perl split). Returns count. */
static int split_id(char *line, char **f, int maxf)
{
	int n = 0;
	char *s = line;
	f[n++] = s;
	while (*s) {
		int sep = 0;
		if (*s == ' ')
			sep = 1;
		else if (s[0] == '\t' && s[1] == '\t')
			sep = 2;
		else if (*s == '\t')
			sep = 1;
		if (sep) {
			*s = '\0';
			s += sep;
			if (n < maxf)
				f[n++] = s;
			else
				break;
		} else {
			s++;
		}
	}
	while (n > 0 && f[n - 1][0] == '\0')
		n--;
	return n;
}

int o_taxcollect_buf(const o_taxdb *db, const char *in, size_t in_len, obuf *out, obuf *report)
{
	const char *p = in, *end = in + in_len;
	while (p < end) {
		const char *eol = memchr(p, '\n', (size_t)(end - p));
		size_t len = eol ? (size_t)(eol - p) : (size_t)(end - p);
		char *line = (char *)malloc(len + 1);
		memcpy(line, p, len);
		line[len] = '\0';
		p = eol ? eol + 1 : end;

		/* @gi = split(/\|/, line); an empty list ends the program (taxcollector:83-87) */
		int only_bars = 1;
		for (size_t k = 0; k < len; k++)
			if (line[k] != '|')
				only_bars = 0;
		if (only_bars) {
			free(line);
			return 0;
		}
		char *gi_text = NULL;
		char *bar1 = strchr(line, '|');
		char *gi_copy = NULL;
		if (bar1) {
			char *bar2 = strchr(bar1 + 1, '|');
			size_t gl = bar2 ? (size_t)(bar2 - bar1 - 1) : strlen(bar1 + 1);
			gi_copy = (char *)malloc(gl + 1);
			memcpy(gi_copy, bar1 + 1, gl);
			gi_copy[gl] = '\0';
			gi_text = gi_copy;
		}
		obuf lin;
		obuf_init(&lin);
		int rc = o_taxcollect_lineage(db, gi_text, &lin, report);
		free(gi_copy);
		if (rc < 0) {
			obuf_free(&lin);
			free(line);
			return -1;
		}
		char *f[64];
		int nf = split_id(line, f, 64);
		obuf_puts(out, nf > 0 ? f[0] : "");
		obuf_puts(out, "\t");
		obuf_put(out, lin.p ? lin.p : "", lin.n);
		for (int i = 2; i <= 12; i++)
			if (i < nf && f[i][0] != '\0') {
				obuf_puts(out, "\t");
				obuf_puts(out, f[i]);
			}
		obuf_puts(out, "\n");
		obuf_free(&lin);
		free(line);
	}
	return 0;
}

int o_taxcollect_file(const o_taxdb *db, const char *in_path, const char *out_path, obuf *report)
{
	size_t len;
	char *in = o_read_file(in_path, &len);
	if (!in) {
		/* taxcollector:31-34 */
		obuf_printf(report, "Error: Unable to open classification results file %s.\n", in_path);
		return 1;
	}
	obuf out;
	obuf_init(&out);
	int rc = o_taxcollect_buf(db, in, len, &out, report);
	int wrc = obuf_write_file(&out, out_path);
	obuf_free(&out);
	free(in);
	if (wrc < 0) {
		obuf_printf(report, "Error: Unable to open output file %s.\n", out_path);
		return 1;
	}
	return rc;
}
