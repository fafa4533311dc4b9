/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * CPU restatement of the two reference scripts that consume the Consensus output:
 *   Megaclust/megaclust2.pl          threshold filter + per-lineage counting  -> "OTU,times_hit" CSV
 *   Megaclustable/megaclustable.pl   rank-level pivot of several such CSVs    -> tab-separated table
 * PINNED by golden vectors produced by running the reference's own Perl (oracle/gen_goldens_megaclust.py).
 * megaclust2.pl prints its table with `keys %h` (Perl hash order, randomised per process), so the order of its
 * data lines is not defined by the reference; this restatement uses first-counted order and the tests compare
 * the data lines as a multiset.  megaclustable.pl is order-deterministic given its input files: byte parity.
 */
#include "o_common.h"
#include "o_classify.h"
#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------- Perl numification (what `<`, `>`, `+=` do to a string; perlnumber / grok_number) ---------- */
static int perl_isspace(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

double o_perl_num(const char *s, size_t n)
{
	size_t i = 0;
	while (i < n && perl_isspace((unsigned char)s[i]))
		i++;
	size_t st = i;
	if (i < n && (s[i] == '+' || s[i] == '-'))
		i++;
	/* Inf / Infinity / NaN, any case */
	if (n - i >= 3 && tolower((unsigned char)s[i]) == 'i' && tolower((unsigned char)s[i + 1]) == 'n' &&
	    tolower((unsigned char)s[i + 2]) == 'f')
		return s[st] == '-' ? -INFINITY : INFINITY;
	if (n - i >= 3 && tolower((unsigned char)s[i]) == 'n' && tolower((unsigned char)s[i + 1]) == 'a' &&
	    tolower((unsigned char)s[i + 2]) == 'n')
		return NAN;
	size_t d0 = i;
	while (i < n && isdigit((unsigned char)s[i]))
		i++;
	size_t nd = i - d0;
	if (i < n && s[i] == '.') {
		i++;
		size_t f0 = i;
		while (i < n && isdigit((unsigned char)s[i]))
			i++;
		nd += i - f0;
	}
	if (nd == 0)
		return 0.0; /* no numeric prefix ("abc", "", ".") */
	if (i < n && (s[i] == 'e' || s[i] == 'E')) {
		size_t j = i + 1;
		if (j < n && (s[j] == '+' || s[j] == '-'))
			j++;
		if (j < n && isdigit((unsigned char)s[j])) {
			while (j < n && isdigit((unsigned char)s[j]))
				j++;
			i = j;
		}
	}
	char buf[128];
	size_t len = i - st;
	if (len >= sizeof buf)
		len = sizeof buf - 1;
	memcpy(buf, s + st, len);
	buf[len] = 0;
	return strtod(buf, NULL); /* the prefix holds decimal digits only: no hex, no "infinity" */
}

/* ---------- split /\t\t|\t\s|\s\t|\t/ (megaclust2.pl:96), first `max` fields ---------- */
typedef struct {
	const char *p;
	size_t n;
	int defined;
} o_field;

static int split_fields(const char *s, size_t n, o_field *f, int max)
{
	int nf = 0;
	size_t start = 0, i = 0;
	/* Perl's split drops trailing empty fields: collect all, trim afterwards */
	o_field all[64];
	int na = 0;
	while (i < n) {
		size_t sep = 0;
		if (s[i] == '\t') {
			if (i + 1 < n && perl_isspace((unsigned char)s[i + 1]))
				sep = 2; /* \t\t or \t\s */
			else
				sep = 1;
		} else if (perl_isspace((unsigned char)s[i]) && i + 1 < n && s[i + 1] == '\t') {
			sep = 2; /* \s\t */
		}
		if (sep) {
			if (na < 64) {
				all[na].p = s + start;
				all[na].n = i - start;
				all[na].defined = 1;
				na++;
			}
			i += sep;
			start = i;
		} else {
			i++;
		}
	}
	if (na < 64) {
		all[na].p = s + start;
		all[na].n = n - start;
		all[na].defined = 1;
		na++;
	}
	while (na > 0 && all[na - 1].n == 0)
		na--;
	for (int k = 0; k < max; k++) {
		if (k < na) {
			f[k] = all[k];
			nf++;
		} else {
			f[k].p = "";
			f[k].n = 0;
			f[k].defined = 0;
		}
	}
	return nf;
}

/* ---------- tiny string-keyed table kept in insertion order ---------- */
typedef struct {
	char **key;
	size_t *klen;
	long long *val;
	size_t n, cap;
	size_t *slot; /* open addressing: index + 1, 0 = empty */
	size_t nslot;
} strtab;

static uint64_t fnv(const char *p, size_t n)
{
	uint64_t h = 1469598103934665603ull;
	for (size_t i = 0; i < n; i++)
		h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
	return h;
}
static void st_init(strtab *t)
{
	memset(t, 0, sizeof *t);
	t->nslot = 1024;
	t->slot = calloc(t->nslot, sizeof *t->slot);
}
static void st_free(strtab *t)
{
	for (size_t i = 0; i < t->n; i++)
		free(t->key[i]);
	free(t->key);
	free(t->klen);
	free(t->val);
	free(t->slot);
}
static void st_rehash(strtab *t)
{
	free(t->slot);
	t->nslot *= 2;
	t->slot = calloc(t->nslot, sizeof *t->slot);
	for (size_t i = 0; i < t->n; i++) {
		size_t h = fnv(t->key[i], t->klen[i]) & (t->nslot - 1);
		while (t->slot[h])
			h = (h + 1) & (t->nslot - 1);
		t->slot[h] = i + 1;
	}
}
/* index of key, inserting it (value 0) if absent; *isnew tells */
static size_t st_get(strtab *t, const char *k, size_t n, int *isnew)
{
	size_t h = fnv(k, n) & (t->nslot - 1);
	while (t->slot[h]) {
		size_t i = t->slot[h] - 1;
		if (t->klen[i] == n && memcmp(t->key[i], k, n) == 0) {
			*isnew = 0;
			return i;
		}
		h = (h + 1) & (t->nslot - 1);
	}
	if (t->n == t->cap) {
		t->cap = t->cap ? t->cap * 2 : 256;
		t->key = realloc(t->key, t->cap * sizeof *t->key);
		t->klen = realloc(t->klen, t->cap * sizeof *t->klen);
		t->val = realloc(t->val, t->cap * sizeof *t->val);
	}
	size_t i = t->n++;
	t->key[i] = malloc(n + 1);
	memcpy(t->key[i], k, n);
	t->key[i][n] = 0;
	t->klen[i] = n;
	t->val[i] = 0;
	t->slot[h] = i + 1;
	*isnew = 1;
	if (t->n * 2 > t->nslot)
		st_rehash(t);
	return i;
}

/* ---------- megaclust2.pl ---------- */
static const char *const kUsage = /* megaclust2.pl:166-185 */
	"Usage:\n"
	"\t\t   cluster-blast-output.pl -i infile -o outfile [options]\n"
	"\t\t   \n"
	"\t\t   Required options:\n"
	"\t\t   -i input BLAST tabular results file (megablast or blastall -m 8)\n"
	"\t\t   -o output file name\n"
	"\n"
	"\t\t   Optional parameters:\n"
	"\t\t   -s similarity lower threshold (percent, between 0-100) (default 95)\n"
	"\t\t   -e e-value upper threshold (default 1e-20)\n"
	"\t\t   -b bitscore lower threshold (default 200)\n"
	"\t\t   -d delimiter (default to comma)\n"
	"\t\t   \n"
	"\t\t   Optional switches:\n"
	"\t\t   -c count every query hit (if -c not given, then only count\n"
	"\t\t\t\t\t     any query-genome pair as one genome hit)\n"
	"\t\t   -h print usage summary\n"
	"\t\t   ";

/* Perl truth of an option value: undef, "" and "0" are false */
static int perl_true(const char *v) { return v && v[0] && !(v[0] == '0' && v[1] == 0); }

/* opts: raw option texts as getopts('i:o:s:e:b:c:d:h') would hold them (NULL = not given) */
int o_megaclust2(const o_megaclust_opts *o, obuf *log)
{
	if (o->h) { /* :37-40 */
		obuf_puts(log, kUsage);
		obuf_puts(log, "\n");
		return 0;
	}
	if (!(perl_true(o->i) && perl_true(o->o))) { /* :42-46 */
		obuf_puts(log, "Must specify both an input and output filename\n");
		obuf_puts(log, kUsage);
		obuf_puts(log, "\n");
		return 0;
	}
	double sim = 95; /* :48-58 */
	if (perl_true(o->s)) {
		double v = o_perl_num(o->s, strlen(o->s));
		if (v < 0 || v > 100) {
			obuf_puts(log, "similarity threshold must be between 0 and 100\n");
			obuf_puts(log, kUsage);
			obuf_puts(log, "\n");
			return 0;
		}
		sim = v;
	}
	double ev = 1e-20; /* :60-63 */
	if (perl_true(o->e))
		ev = o_perl_num(o->e, strlen(o->e));
	double bits = 200; /* :65-68 */
	if (perl_true(o->b))
		bits = o_perl_num(o->b, strlen(o->b));
	const char *delim = perl_true(o->d) ? o->d : ","; /* :70-73 */
	const int count_all = perl_true(o->c);           /* :139 */

	size_t len;
	char *text = o_read_file(o->i, &len);
	if (!text)
		return -2; /* die "couldn't open infile" (:75) */
	FILE *fo = fopen(o->o, "w");
	if (!fo) {
		free(text);
		return -2; /* die "couldn't open outfile" (:76) */
	}
	long long processed = 0, beyond = 0;
	strtab subj, pair;
	st_init(&subj);
	st_init(&pair);
	obuf key;
	obuf_init(&key);
	for (size_t s = 0; s < len;) {
		const char *nl = memchr(text + s, '\n', len - s);
		size_t e = nl ? (size_t)(nl - text) : len; /* chomp: the newline only */
		const char *line = text + s;
		size_t n = e - s;
		s = nl ? e + 1 : len;
		if (n > 0 && line[0] == '#') /* :81 */
			continue;
		processed++;
		o_field f[13];
		split_fields(line, n, f, 13);
		const double pid = f[2].defined ? o_perl_num(f[2].p, f[2].n) : 0.0;
		const double e_value = f[10].defined ? o_perl_num(f[10].p, f[10].n) : 0.0;
		const double bitscore = f[11].defined ? o_perl_num(f[11].p, f[11].n) : 0.0;
		if (pid < sim || e_value > ev || bitscore < bits) { /* :132-137 */
			beyond++;
			continue;
		}
		int isnew;
		size_t si = st_get(&subj, f[1].p, f[1].n, &isnew);
		if (count_all) {
			subj.val[si]++;
		} else {
			/* :143-147: a (subject, query) pair counts once */
			key.n = 0;
			obuf_put(&key, f[1].p, f[1].n);
			obuf_put(&key, "\0", 1);
			obuf_put(&key, f[0].p, f[0].n);
			int pnew;
			st_get(&pair, key.p, key.n, &pnew);
			if (pnew)
				subj.val[si]++;
		}
	}
	fprintf(fo, "OTU%stimes_hit\n", delim); /* :152-153 */
	for (size_t i = 0; i < subj.n; i++) {
		fwrite(subj.key[i], 1, subj.klen[i], fo);
		fprintf(fo, "%s%lld\n", delim, subj.val[i]);
	}
	fclose(fo);
	obuf_printf(log, "Run complete:\n%lld hits examined\n%lld hits beyond thresholds and therefore not counted.\n", processed,
		    beyond); /* :161-163 */
	obuf_free(&key);
	st_free(&subj);
	st_free(&pair);
	free(text);
	return 0;
}

/* Getopt::Std::getopts('i:o:s:e:b:c:d:h') over argv[1..] */
int o_megaclust2_main(int argc, char **argv, obuf *log)
{
	o_megaclust_opts o;
	memset(&o, 0, sizeof o);
	int a = 1;
	while (a < argc && argv[a][0] == '-' && argv[a][1]) {
		if (strcmp(argv[a], "--") == 0) {
			a++;
			break;
		}
		const char *p = argv[a] + 1;
		a++;
		while (*p) {
			const char c = *p++;
			const char **dst = c == 'i' ? &o.i : c == 'o' ? &o.o : c == 's' ? &o.s : c == 'e' ? &o.e : c == 'b' ? &o.b
					   : c == 'c' ? &o.c : c == 'd' ? &o.d : NULL;
			if (dst) {
				if (*p) {
					*dst = p;
				} else if (a < argc) {
					*dst = argv[a++];
				}
				break;
			} else if (c == 'h') {
				o.h = 1;
			} else {
				fprintf(stderr, "Unknown option: %c\n", c);
			}
		}
	}
	return o_megaclust2(&o, log);
}

/* ---------- megaclustable.pl ---------- */
/* Perl index(str, sub, pos) */
static long p_index(const char *s, size_t n, const char *sub, long pos)
{
	size_t m = strlen(sub);
	if (pos < 0)
		pos = 0;
	if ((size_t)pos > n)
		pos = (long)n;
	if (m == 0)
		return pos;
	for (size_t i = (size_t)pos; i + m <= n; i++)
		if (memcmp(s + i, sub, m) == 0)
			return (long)i;
	return -1;
}
/* Perl substr(str, off, len) for off >= 0: negative len leaves that many characters off the end */
static void p_substr(const char *s, size_t n, long off, long len, const char **p, size_t *pn)
{
	*p = "";
	*pn = 0;
	if (off < 0 || (size_t)off > n)
		return;
	long end = len >= 0 ? off + len : (long)n + len;
	if (end > (long)n)
		end = (long)n;
	if (end <= off)
		return;
	*p = s + off;
	*pn = (size_t)(end - off);
}

typedef struct {
	int is_num; /* 0: raw text as read from the file, 1: number (after +=), -1: undef */
	char *txt;
	double num;
} cell;

static void cell_print(const cell *c, FILE *fo)
{
	if (c->is_num < 0 || (c->is_num == 0 && c->txt[0] == 0)) { /* `eq ""` -> 0 (:120-123) */
		fputs("0", fo);
	} else if (c->is_num == 0) {
		fputs(c->txt, fo);
	} else {
		fprintf(fo, "%.15g", c->num);
	}
}

int o_megaclustable_main(int argc, char **argv, obuf *log)
{
	/* argv[1..] is @ARGV */
	const int nargs = argc - 1;
	if (nargs - 1 < 5) { /* :17-21 */
		obuf_puts(log, "Please enter the correct parameters.\n");
		return 0;
	}
	const char *output = NULL;
	char level[64] = "";
	char **files = calloc((size_t)nargs + 1, sizeof *files);
	int nfiles = 0, m_in = 0;
	for (int a = 1; a <= nargs; a++) { /* :25-52; `$mIN` (:38) is a typo: -t never leaves the file list mode */
		if (strcmp(argv[a], "-m") == 0) {
			m_in = 1;
		} else if (strcmp(argv[a], "-o") == 0) {
			m_in = 0;
			a++;
			output = a <= nargs ? argv[a] : NULL;
		} else if (strcmp(argv[a], "-t") == 0) {
			a++;
			const char *t = a <= nargs ? argv[a] : "";
			const double v = o_perl_num(t, strlen(t));
			if (v > 6 || v < 0) {
				obuf_puts(log, "You must enter a number between 0 and 6 for taxonomy level where 0 = domain and 6 = species.\n");
				free(files);
				return 0;
			}
			snprintf(level, sizeof level, "[%s]", t);
		} else if (m_in) {
			files[nfiles++] = argv[a];
		}
	}
	char **taxa = NULL;
	size_t size = 0, tcap = 0;
	cell **table = calloc((size_t)nfiles + 1, sizeof *table);
	size_t *tlen = calloc((size_t)nfiles + 1, sizeof *tlen);
	for (int b = 0; b < nfiles; b++) {
		size_t len;
		char *text = o_read_file(files[b], &len);
		if (!text) { /* :63-67 */
			obuf_printf(log, "Unable to open %s\nMake sure you entered the extension when entering the file name.\n", files[b]);
			free(files);
			return 0;
		}
		size_t fcap = size + 16, fn = size;
		cell *file = calloc(fcap, sizeof *file);
		for (size_t a = 0; a < size; a++) { /* :69-72 */
			file[a].is_num = 0;
			file[a].txt = strdup("0");
		}
		for (size_t s = 0; s < len;) {
			const char *nl = memchr(text + s, '\n', len - s);
			size_t e = nl ? (size_t)(nl - text) : len;
			const char *line = text + s;
			size_t n = e - s;
			s = nl ? e + 1 : len;
			long loc = p_index(line, n, level, 0); /* :76 */
			if (loc < 0)
				continue;
			loc += 3;
			long end = p_index(line, n, ";", loc);
			if (end == -1)
				end = p_index(line, n, ",", loc);
			const char *np;
			size_t nn;
			p_substr(line, n, loc, end - loc, &np, &nn);
			long num_start = p_index(line, n, ",", end) + 1;
			const char *vp;
			size_t vn;
			p_substr(line, n, num_start, (long)n - num_start, &vp, &vn);
			int found = 0;
			for (size_t a = 0; a < size; a++) { /* :89-96 */
				if (strlen(taxa[a]) == nn && memcmp(taxa[a], np, nn) == 0) {
					found = 1;
					const double cur = file[a].is_num == 1 ? file[a].num : o_perl_num(file[a].txt, strlen(file[a].txt));
					file[a].num = cur + o_perl_num(vp, vn);
					file[a].is_num = 1;
				}
			}
			if (!found) { /* :98-103 */
				if (size == tcap) {
					tcap = tcap ? tcap * 2 : 64;
					taxa = realloc(taxa, tcap * sizeof *taxa);
				}
				taxa[size] = strndup(np, nn);
				if (fn == fcap) {
					fcap *= 2;
					file = realloc(file, fcap * sizeof *file);
				}
				file[fn].is_num = 0;
				file[fn].txt = strndup(vp, vn);
				file[fn].num = 0;
				fn++;
				size++;
			}
		}
		free(text);
		table[b + 1] = file;
		tlen[b + 1] = fn;
	}
	FILE *fo = output ? fopen(output, "w") : NULL;
	if (!fo) {
		free(files);
		return -2; /* die $! (:111) */
	}
	for (int a = 1; a <= nfiles; a++)
		fprintf(fo, "\t%d", a);
	for (size_t a = 0; a < size; a++) {
		fputs("\n", fo);
		cell name = { 0, taxa[a], 0 };
		cell_print(&name, fo);
		fputs("\t", fo);
		for (int b = 1; b <= nfiles; b++) {
			cell undef = { -1, NULL, 0 };
			cell_print(a < tlen[b] ? &table[b][a] : &undef, fo);
			fputs("\t", fo);
		}
	}
	fclose(fo);
	free(files);
	return 0;
}
