/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * CPU restatement of the reference's read pre-processing script for its FASTQ and QSEQ inputs:
 *   Trim/trim2.4.pl (and trim2.3.pl, which README.md:34 calls; the two differ only inside join_fasta)
 * PINNED by golden vectors produced by running the reference's own Perl on seeded inputs
 * (oracle/gen_goldens_trim.py -> tests/golden/trim).
 *
 * What the script computes, as its Perl actually behaves (several of its statements have no effect):
 *   - getopts('a:b:g:t:q:qc:lc:j') (trim2.4.pl:51) declares the letters a b g t q c (with a value) and l j (flags);
 *     `-qc`/`-lc` therefore never reach $QUALITY_CUTOFF / $LENGTH_CUTOFF, which stay 20 and 70 (:33-34, :94-100);
 *   - the format is the first byte of the -a file (:104-105, :117, :146): '@' FASTQ, '>' FASTA (below),
 *     otherwise QSEQ if field 7 of the first line is 1|2 and field 10 is 0|1 (:152-156);
 *   - the quality rule (:543-563, :272-287) is a running sum of (quality - cutoff) clamped at 0 from below;
 *     `end` = the first index where the sum reaches its overall maximum; the read keeps bases [0, end) — the base
 *     at `end` itself is dropped by substr (:565, :289) — and `$seq[$a] = "N"` (:285, :561) writes to an unrelated
 *     array, so no base is ever masked;
 *   - a kept part shorter than 70 gives the number 0 (:569, :293), which FASTQ mode PRINTS as the sequence "0";
 *   - FASTQ: phred+33 (:547); with -b the mates are read interleaved from the -a file (:492-495; the -b file is
 *     only opened), joined by $GAPSIZE N's (:502-505), and the second mate keeps a trailing tab (:571, :508);
 *   - QSEQ: phred+64 (:273); the first $TRUNCATE bases and qualities are cut, then $TRUNCATE-1 more bases but not
 *     qualities (:259-262); '.' becomes N (:189-190); a pair is written only if both mates survive (:199-245);
 *   - FASTA input (:117-143), trim2.4.pl's text (trim2.3.pl's join_fasta differs: it prints to STDOUT and tests eof
 *     inside the loop conditions).  Without -j: needs -q; prints the -q value, then parse_fasta (:384-465) reads the
 *     sequence file and the quality file LINE BY LINE in step.  Its cut-offs are the barewords LENGTH_CUTOFF and
 *     QUALITY_CUTOFF (:386-387) -- strings that count as 0 -- so no record is ever rejected and the running sum is the sum
 *     of the quality numbers themselves; `end` = last position (field index + 60 x line number) where the sum reached a
 *     new maximum, `start` = the position of the last reset; both SURVIVE from record to record when a record never
 *     raises the sum.  A record is printed -- to STDOUT, not to the output file -- when the NEXT header line arrives
 *     (so the last record never is): the header up to and including its first blank ("" without one), then the stored
 *     letters start..end, a line break after every 60th, and one more line break.  A line stores all its characters
 *     but the last (the line break; a base if the file's last line has none).  With -j and -b: join_fasta (:301-382)
 *     writes, per pair of records of the two files, "<header 1>_<header 2 without '>'>" and the two sequences (lines
 *     joined) with -g N's between -- none without -g -- to the output file; the line that makes eof() true inside a
 *     sequence loop (the last line of a file, when it is not the record's first) is never added; no closing message.
 */
#include "o_common.h"
#include "o_classify.h"
#include <errno.h>
#include <libgen.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#define QUALITY_CUTOFF 20 /* trim2.4.pl:34 */
#define LENGTH_CUTOFF 70  /* trim2.4.pl:33 */

static int p_true(const char *v) { return v && v[0] && !(v[0] == '0' && v[1] == 0); }
static int p_space(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

typedef struct {
	const char *p;
	size_t n;
} span;

/* <FH>: the next line including its "\n"; {NULL,0} at end of file */
typedef struct {
	const char *p;
	size_t n, pos;
} reader;

static span next_line(reader *r)
{
	span s = { NULL, 0 };
	if (!r->p || r->pos >= r->n)
		return s;
	const char *st = r->p + r->pos;
	const char *nl = memchr(st, '\n', r->n - r->pos);
	s.p = st;
	s.n = nl ? (size_t)(nl - st) + 1 : r->n - r->pos;
	r->pos += s.n;
	return s;
}

/* `while ($line = <FH>)` is `while (defined($line = <FH>))` in Perl: only end of file ends the loop */
static int line_is_false(span s) { return s.p == NULL; }

static span chomp(span s)
{
	if (s.n && s.p[s.n - 1] == '\n')
		s.n--;
	return s;
}

/* :535-563 / :264-287 */
static size_t quality_end(const unsigned char *q, size_t n, int offset)
{
	long max = 0, sum = 0;
	size_t end = 0;
	for (size_t a = 0; a < n; a++) {
		sum += (long)q[a] - offset - QUALITY_CUTOFF;
		if (sum > max) {
			max = sum;
			end = a;
		}
		if (sum < 0)
			sum = 0;
	}
	return end;
}

/* trim_fastq (:529-578): returns 0 ("0") or 1 with *seq = substr($sequence, 0, $end) */
static int trim_fastq(span sequence, span quality, span *seq)
{
	const size_t end = quality_end((const unsigned char *)quality.p, quality.n, 33);
	seq->p = sequence.p;
	seq->n = end < sequence.n ? end : sequence.n;
	return seq->n >= LENGTH_CUTOFF;
}

static void put_gap(obuf *a, obuf *b, long gap)
{
	for (long r = 0; r < gap; r++) {
		if (a)
			obuf_put(a, "N", 1);
		if (b)
			obuf_put(b, "N", 1);
	}
}

/* parse_fastq (:467-524) */
static void parse_fastq(reader *r1, int paired, long gap, obuf *out, obuf *fasta)
{
	for (;;) {
		span header1 = next_line(r1);
		if (line_is_false(header1))
			break;
		span sequence = next_line(r1);
		(void)next_line(r1);
		span quality = next_line(r1);
		span s1;
		const int keep1 = trim_fastq(sequence, quality, &s1);
		header1 = chomp(header1);
		obuf rec;
		obuf_init(&rec);
		obuf_put(&rec, ">", 1);
		for (size_t i = 0; i < header1.n; i++) /* s/@//g */
			if (header1.p[i] != '@')
				obuf_put(&rec, header1.p + i, 1);
		obuf_puts(&rec, ":AB\n");
		if (keep1) {
			for (size_t i = 0; i < s1.n; i++) /* $seq . "\t" . ""  then s/\s//g */
				if (!p_space((unsigned char)s1.p[i]))
					obuf_put(&rec, s1.p + i, 1);
		} else {
			obuf_put(&rec, "0", 1);
		}
		if (paired) {
			(void)next_line(r1);
			span sequence_2 = next_line(r1);
			(void)next_line(r1);
			span quality_2 = next_line(r1);
			span s2;
			const int keep2 = trim_fastq(sequence_2, quality_2, &s2);
			put_gap(&rec, NULL, gap);
			if (keep2) {
				obuf_put(&rec, s2.p, s2.n); /* chomp has nothing to remove: the value ends with the tab */
				obuf_put(&rec, "\t", 1);
			} else {
				obuf_put(&rec, "0", 1);
			}
		}
		obuf_put(&rec, "\n", 1);
		obuf_put(out, rec.p, rec.n);
		obuf_put(fasta, rec.p, rec.n);
		obuf_free(&rec);
	}
}

/* split(/\t/, $line): field k of a chomped line ({NULL,0} when the line has fewer fields) */
static span field(span line, int k)
{
	span f = { NULL, 0 };
	size_t st = 0;
	if (!line.p)
		return f;
	for (int i = 0;; i++) {
		const char *tab = st <= line.n ? memchr(line.p + st, '\t', line.n - st) : NULL;
		const size_t en = tab ? (size_t)(tab - line.p) : line.n;
		if (i == k) {
			f.p = line.p + st;
			f.n = en - st;
			return f;
		}
		if (!tab)
			return f;
		st = en + 1;
	}
}

static int span_eq(span s, const char *lit) { return s.p != NULL && s.n == strlen(lit) && memcmp(s.p, lit, s.n) == 0; }

/* trim_qseq (:253-298) on field 8 / field 9; t1 = int($TRUNCATE), t2 = int($TRUNCATE - 1) */
static int trim_qseq(span seq, span qual, long t1, long t2, span *kept)
{
	/* $seq = substr($seq, $TRUNCATE); $qual = substr($qual, $TRUNCATE): beyond the end gives undef (empty) */
	size_t cut = (size_t)t1 < seq.n ? (size_t)t1 : seq.n;
	seq.p += cut;
	seq.n -= cut;
	cut = (size_t)t1 < qual.n ? (size_t)t1 : qual.n;
	qual.p += cut;
	qual.n -= cut;
	/* substr($seq, 0, $TRUNCATE-1) = '': a negative length leaves that many characters at the end */
	size_t drop;
	if (t2 >= 0)
		drop = (size_t)t2 < seq.n ? (size_t)t2 : seq.n;
	else
		drop = (size_t)(-t2) < seq.n ? seq.n - (size_t)(-t2) : 0;
	seq.p += drop;
	seq.n -= drop;
	const size_t end = quality_end((const unsigned char *)qual.p, qual.n, 64);
	kept->p = seq.p;
	kept->n = end < seq.n ? end : seq.n;
	return kept->n >= LENGTH_CUTOFF;
}

static void put_header(obuf *o, span line)
{
	/* join(':', @line[0..7]) (:218): missing fields are empty strings */
	for (int k = 0; k < 8; k++) {
		span f = field(line, k);
		if (k)
			obuf_put(o, ":", 1);
		if (f.p)
			obuf_put(o, f.p, f.n);
	}
}

/* parse_qseq (:169-250) */
static void parse_qseq(reader *r1, reader *r2, long gap, long t1, long t2, obuf *fasta)
{
	for (;;) {
		span l1 = next_line(r1);
		if (line_is_false(l1))
			break;
		l1 = chomp(l1);
		span l2 = chomp(next_line(r2));
		span s1 = field(l1, 8), s2 = field(l2, 8);
		int zero1, zero2, dots = 0;
		if (span_eq(field(l1, 7), "1")) {
			span k1, k2;
			zero1 = !trim_qseq(s1, field(l1, 9), t1, t2, &k1);
			zero2 = !trim_qseq(s2, field(l2, 9), t1, t2, &k2);
			s1 = k1;
			s2 = k2;
			dots = 1; /* s/\./N/g (:189-190) */
		} else {
			zero1 = span_eq(s1, "0");
			zero2 = span_eq(s2, "0");
		}
		if (zero1 || zero2)
			continue; /* :199-210: singletons are not written anywhere */
		obuf_put(fasta, ">", 1);
		put_header(fasta, l1);
		obuf_puts(fasta, ":AB\n");
		for (size_t i = 0; i < s1.n; i++)
			obuf_put(fasta, dots && s1.p[i] == '.' ? "N" : s1.p + i, 1);
		put_gap(fasta, NULL, gap);
		for (size_t i = 0; i < s2.n; i++)
			obuf_put(fasta, dots && s2.p[i] == '.' ? "N" : s2.p + i, 1);
		obuf_put(fasta, "\n", 1);
	}
}

/* ---- FASTA input ---- */
static int has_gt(span s) { return s.p != NULL && memchr(s.p, '>', s.n) != NULL; }
static int at_eof(const reader *r) { return !r->p || r->pos >= r->n; }

/* trim2.4.pl:384-465 */
static void parse_fasta(reader *r1, reader *rq, obuf *out)
{
	int rejected = -1;
	char *trim = NULL; /* @FinalTrim */
	size_t n_trim = 0, cap = 0;
	long end = 0, start = 0, first = 0, line_num = 0;
	double max = 0, sum = 0;
	span header = { NULL, 0 };
	for (;;) {
		span ls = next_line(r1);
		if (line_is_false(ls))
			break;
		span lq = next_line(rq);
		if (has_gt(ls)) {
			/* $length = $end + 1 - $start < "LENGTH_CUTOFF" (= 0): only a NEGATIVE length rejects -- quality lines of more
			 * than 60 values can put $end before $first (positions are a + 60 x line number) -- and the record is then not
			 * printed at all (:404-408) */
			if (end + 1 - start < 0)
				rejected = 1;
			if (rejected == 0) {
				if (header.p)
					obuf_put(out, header.p, header.n);
				obuf_puts(out, "\n");
				int count_down = 60;
				for (long a = start; a <= end; a++) {
					count_down--;
					if (a >= 0 && (size_t)a < n_trim)
						obuf_put(out, trim + a, 1);
					if (count_down == 0) {
						obuf_puts(out, "\n");
						count_down = 60;
					}
				}
				obuf_puts(out, "\n");
			}
			rejected = 0;
			const char *sp = (const char *)memchr(ls.p, ' ', ls.n);
			header.p = ls.p;
			header.n = sp ? (size_t)(sp - ls.p) + 1 : 0;
			n_trim = 0;
			max = 0;
			sum = 0;
			first = 0;
			line_num = 0;
		} else if (rejected == 0) {
			/* every character but the last */
			if (ls.n > 1) {
				if (n_trim + ls.n > cap) {
					cap = 2 * (n_trim + ls.n) + 64;
					trim = (char *)realloc(trim, cap);
				}
				memcpy(trim + n_trim, ls.p, ls.n - 1);
				n_trim += ls.n - 1;
			}
			/* chomp; split(/ /): leading and inner empty fields stay, trailing ones go */
			span q = lq.p ? chomp(lq) : lq;
			size_t nf = 0, last_nonempty = 0;
			for (size_t i = 0, f0 = 0; q.p && i <= q.n; i++)
				if (i == q.n || q.p[i] == ' ') {
					nf++;
					if (i > f0)
						last_nonempty = nf;
					f0 = i + 1;
				}
			size_t a = 0;
			for (size_t i = 0, f0 = 0; q.p && i <= q.n && a < last_nonempty; i++)
				if (i == q.n || q.p[i] == ' ') {
					sum += o_perl_num(q.p + f0, i - f0); /* - "QUALITY_CUTOFF" (= 0) */
					if (sum > max) {
						max = sum;
						end = (long)a + 60 * line_num;
						start = first;
					}
					if (sum < 0) {
						sum = 0;
						first = (long)a + 60 * line_num;
					}
					a++;
					f0 = i + 1;
				}
			line_num++;
		}
	}
	free(trim);
}

/* trim2.4.pl:301-382; `gap` < 0: -g not given (or false) */
static void join_fasta(reader *r1, reader *r2, long gap, obuf *fasta)
{
	int first = 1;
	span header1 = { NULL, 0 }, header2 = { NULL, 0 };
	obuf seq;
	obuf_init(&seq);
	for (;;) {
		span l1 = next_line(r1);
		if (line_is_false(l1))
			break;
		l1 = chomp(l1);
		if (!first) {
			span h1 = header1.p ? chomp(header1) : header1, h2 = header2.p ? chomp(header2) : header2;
			if (h1.p)
				obuf_put(fasta, h1.p, h1.n);
			obuf_puts(fasta, "_");
			for (size_t i = 0; h2.p && i < h2.n; i++)
				if (h2.p[i] != '>')
					obuf_put(fasta, h2.p + i, 1);
			obuf_puts(fasta, "\n");
		}
		span l2;
		if (first) {
			l2 = next_line(r2);
			first = 0;
			obuf_put(fasta, l1.p, l1.n);
			obuf_puts(fasta, "_");
			for (size_t i = 0; l2.p && i < l2.n; i++) /* (not chomped: its line break ends the header line) */
				if (l2.p[i] != '>')
					obuf_put(fasta, l2.p + i, 1);
			l1 = next_line(r1);
		}
		while (!has_gt(l1)) { /* grep !/>/ on an undefined line is true as well */
			if (l1.p) {
				span c = chomp(l1);
				obuf_put(&seq, c.p, c.n);
			}
			l1 = next_line(r1);
			if (at_eof(r1))
				break;
		}
		header1 = l1;
		for (long i = 0; i < gap; i++)
			obuf_puts(&seq, "N");
		l2 = next_line(r2);
		while (!has_gt(l2)) {
			if (l2.p) {
				span c = chomp(l2);
				obuf_put(&seq, c.p, c.n);
			}
			l2 = next_line(r2);
			if (at_eof(r2))
				break;
		}
		header2 = l2;
		if (seq.n)
			obuf_put(fasta, seq.p, seq.n);
		obuf_puts(fasta, "\n");
		seq.n = 0;
	}
	obuf_free(&seq);
}

static const char *const kUsage = /* :54-63 */
	"Usage: perl trim2.pl \n"
	"\t-a raw illumina input file read 1\n"
	"\t-b raw illumina input file read 2 (if any) \n"
	"\t-g size of GAP between paired-ends (if any) \n"
	"\t-t truncate size (if any)\n"
	"\t-q quality file (in case of FASTA input)\n"
	"\t-qc quality cutoff value\n"
	"\t-j use this option for just joining a and b, without triming\n"
	"\t-lc minimum length \n"
	"Supported formats: FASTA, FASTQ and QSEQ.\n";

/* Returns 0, or -1 for what this restatement does not cover (FASTA-format input, a negative -t).
 * *fasta_made = RUNBLAST was opened (:115); *qseq = the singletons directory and file were created (:176-178). */
int o_trim2(const o_trim_opts *o, obuf *out, obuf *fasta, int *fasta_made, int *qseq)
{
	*fasta_made = 0;
	*qseq = 0;
	if (!p_true(o->a)) {
		obuf_puts(out, kUsage);
		return 0;
	}
	size_t an = 0, bn = 0;
	char *a = o_read_file(o->a, &an), *b = NULL;
	if (!a) {
		obuf_printf(out, "Error: Unable to open %s.\n", o->a);
		return 0;
	}
	const int paired = p_true(o->b);
	if (paired) {
		b = o_read_file(o->b, &bn);
		if (!b) {
			obuf_printf(out, "Error: Unable to open %s.\n", o->b);
			free(a);
			return 0;
		}
	}
	long gap = 189, t1 = 11, t2 = 10; /* :35-36 */
	if (p_true(o->g)) {
		const double g = o_perl_num(o->g, strlen(o->g));
		gap = g > 0 ? (g > 2147483647.0 ? 2147483647L : (long)ceil(g)) : 0; /* for ($r = 0; $r < $GAPSIZE; $r++) */
	}
	int rc = 0;
	if (p_true(o->t)) {
		const double t = o_perl_num(o->t, strlen(o->t));
		if (!(t > -1.0) || !(t < 2147483647.0)) {
			rc = -1; /* substr with a negative offset: not restated */
		} else {
			t1 = (long)t;
			t2 = (long)(t - 1.0);
		}
	}
	*fasta_made = rc == 0;
	reader r1 = { a, an, 0 }, r2 = { b, bn, 0 };
	if (rc == 0 && an > 0 && a[0] == '>') {
		/* :117-143 */
		if (o->j) {
			if (paired) {
				long jg = -1;
				if (p_true(o->g)) {
					const double g = o_perl_num(o->g, strlen(o->g));
					jg = g > 0 ? (g > 2147483647.0 ? 2147483647L : (long)ceil(g)) : 0;
				}
				join_fasta(&r1, &r2, jg, fasta);
			} else {
				obuf_puts(out, "Error. Input is -j for joining ends, but you did not provided both sequence a and b with -a and -b options.\n\n");
			}
			free(a);
			free(b);
			return 0; /* exit: no closing message */
		}
		if (p_true(o->q)) {
			obuf_printf(out, "%s\n", o->q);
			size_t qn = 0;
			char *q = o_read_file(o->q, &qn);
			if (!q) {
				obuf_printf(out, "Error: Unable to open %s required for FASTA file triming.\n", o->q);
				free(a);
				free(b);
				return 0;
			}
			reader rq = { q, qn, 0 };
			parse_fasta(&r1, &rq, out);
			free(q);
		} else {
			obuf_puts(out, "Error: Please, specify the FASTA quality file with -q option.\n");
			free(a);
			free(b);
			return 0;
		}
	} else if (rc == 0 && an > 0 && a[0] == '@') {
		parse_fastq(&r1, paired, gap, out, fasta);
	} else if (rc == 0) {
		/* :152-156: the first line without its first byte */
		span first = { NULL, 0 };
		if (an > 1) {
			reader t = { a + 1, an - 1, 0 };
			first = chomp(next_line(&t));
		}
		span f7 = field(first, 7), f10 = field(first, 10);
		/* split drops trailing empty fields, and no field after 10 may rescue an empty field 10 */
		if ((span_eq(f7, "1") || span_eq(f7, "2")) && (span_eq(f10, "0") || span_eq(f10, "1"))) {
			obuf_puts(out, "QSEQ file format found.\n");
			*qseq = 1;
			parse_qseq(&r1, &r2, gap, t1, t2, fasta);
		} else {
			obuf_puts(out, "Error: file format not recognized.\n");
		}
	}
	if (rc == 0)
		obuf_puts(out, "Trimming complete.\n");
	free(a);
	free(b);
	return rc;
}

static int mkdir_p(const char *path)
{
	char tmp[4096];
	snprintf(tmp, sizeof tmp, "%s", path);
	for (char *p = tmp + 1; *p; p++)
		if (*p == '/') {
			*p = 0;
			if (mkdir(tmp, 0777) && errno != EEXIST)
				return -1;
			*p = '/';
		}
	return mkdir(tmp, 0777) && errno != EEXIST ? -1 : 0;
}

/* argv[0] is the program name; same side effects as the script: output_files/trim2/<basename>_runblast.fasta in the
 * working directory, <dirname>/singletons/<basename>_single.txt (empty) for QSEQ input */
int o_trim2_main(int argc, char **argv, obuf *out)
{
	o_trim_opts o;
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
			const char *ignored = NULL;
			const char **dst = c == 'a' ? &o.a : c == 'b' ? &o.b : c == 'g' ? &o.g : c == 't' ? &o.t : c == 'q' ? &o.q : c == 'c' ? &ignored : NULL;
			if (dst) {
				if (*p)
					*dst = p;
				else if (a < argc)
					*dst = argv[a++];
				break;
			} else if (c == 'j') {
				o.j = 1;
			} else if (c == 'l') {
				/* a flag: 'l' is first seen in the spec followed by 'c', not ':' */
			} else {
				fprintf(stderr, "Unknown option: %c\n", c);
			}
		}
	}
	obuf fasta;
	obuf_init(&fasta);
	int made = 0, qseq = 0;
	int rc = o_trim2(&o, out, &fasta, &made, &qseq);
	if (made) {
		char path[4096], b1[4096], b2[4096];
		snprintf(b1, sizeof b1, "%s", o.a);
		snprintf(b2, sizeof b2, "%s", o.a);
		const char *prefix = basename(b1), *dir = dirname(b2);
		if (mkdir_p("output_files/trim2"))
			rc = -2;
		snprintf(path, sizeof path, "output_files/trim2/%s_runblast.fasta", prefix);
		if (rc != -2 && obuf_write_file(&fasta, path))
			rc = -2;
		if (qseq) {
			snprintf(path, sizeof path, "%s/singletons", dir);
			mkdir_p(path);
			snprintf(path, sizeof path, "%s/singletons/%s_single.txt", dir, prefix);
			FILE *f = fopen(path, "wb");
			if (f)
				fclose(f);
		}
	}
	obuf_free(&fasta);
	return rc;
}
