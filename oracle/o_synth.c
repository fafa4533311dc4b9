/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * Synthetic workload of BASELINE.md section 3 / SURVEY 8(d): a 16S-like database of
 * `n_genus` random ancestors, each sequence = its genus ancestor with ~3 % substitutions,
 * a 7-rank synthetic taxonomy, 150-bp reads with 1 % substitutions on either strand and a
 * synthetic RDP stream.  Counter-based (splitmix64) so that the HIP generator in
 * pangea-plus_amd/csrc/synth.hip can produce the same bytes independently; tests compare
 * the two.
 */
#include "o_classify.h"
#include <stdlib.h>
#include <string.h>

static inline uint64_t sm64(uint64_t z)
{
	z += 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

uint64_t o_synth_hash(uint64_t seed, uint64_t tag, uint64_t i, uint64_t j)
{
	return sm64(sm64(sm64(seed + tag) + i) + j);
}

void o_synth_default(o_synth_cfg *c)
{
	c->seed = 0x50414E47ULL;
	c->n_seq = 666667;
	c->seq_len = 1500;
	c->n_genus = 20000;
	c->read_seed = 42;
	c->read_len = 150;
}

static inline int64_t genus_of(const o_synth_cfg *c, int64_t seq)
{
	return (int64_t)(((__int128)seq * c->n_genus) / c->n_seq);
}

/* base j of DB sequence i */
static inline uint8_t db_base(const o_synth_cfg *c, int64_t i, int64_t g, int32_t j)
{
	uint64_t a = o_synth_hash(c->seed, 1, (uint64_t)g, (uint64_t)(j >> 5));
	uint8_t b = (uint8_t)((a >> (2 * (j & 31))) & 3);
	uint64_t m = o_synth_hash(c->seed, 2, (uint64_t)i, (uint64_t)(j >> 2));
	uint32_t f = (uint32_t)((m >> (16 * (j & 3))) & 0xFFFF);
	if (f < 1966) /* 3 % of 65536 */
		b = (uint8_t)((b + 1 + f % 3) & 3);
	return b;
}

void o_synth_db_seq(const o_synth_cfg *c, int64_t i, uint8_t *out)
{
	int64_t g = genus_of(c, i);
	for (int32_t j = 0; j < c->seq_len; j++)
		out[j] = db_base(c, i, g, j);
}

void o_synth_read(const o_synth_cfg *c, int64_t r, uint8_t *out, int64_t *src_seq, int32_t *src_off, int *minus)
{
	uint64_t u = o_synth_hash(c->read_seed, 3, (uint64_t)r, 0);
	int64_t i = (int64_t)((u & 0xFFFFFFFFULL) % (uint64_t)c->n_seq);
	int32_t off = (int32_t)((u >> 32) % (uint64_t)(c->seq_len - c->read_len + 1));
	int mn = (int)(o_synth_hash(c->read_seed, 3, (uint64_t)r, 1) & 1);
	int64_t g = genus_of(c, i);
	int32_t L = c->read_len;
	for (int32_t j = 0; j < L; j++) {
		uint8_t b = db_base(c, i, g, off + j);
		uint64_t e = o_synth_hash(c->read_seed, 4, (uint64_t)r, (uint64_t)(j >> 2));
		uint32_t f = (uint32_t)((e >> (16 * (j & 3))) & 0xFFFF);
		if (f < 655) /* 1 % */
			b = (uint8_t)((b + 1 + f % 3) & 3);
		if (mn)
			out[L - 1 - j] = (uint8_t)(3 - b);
		else
			out[j] = b;
	}
	if (src_seq)
		*src_seq = i;
	if (src_off)
		*src_off = off;
	if (minus)
		*minus = mn;
}

static int64_t max1(int64_t v)
{
	return v < 1 ? 1 : v;
}

void o_synth_tax_counts(const o_synth_cfg *c, int64_t cnt[7])
{
	cnt[6] = c->n_seq;
	cnt[5] = c->n_genus;
	cnt[4] = max1(cnt[5] / 5);
	cnt[3] = max1(cnt[4] / 4);
	cnt[2] = max1(cnt[3] / 5);
	cnt[1] = max1(cnt[2] / 5);
	cnt[0] = max1(cnt[1] / 20);
}

int64_t o_synth_taxid(const o_synth_cfg *c, int level, int64_t index)
{
	int64_t cnt[7], base = 2;
	o_synth_tax_counts(c, cnt);
	for (int l = 0; l < level; l++)
		base += cnt[l];
	return base + index;
}

int64_t o_synth_ancestor(const o_synth_cfg *c, int64_t seq, int level)
{
	int64_t cnt[7];
	o_synth_tax_counts(c, cnt);
	int64_t idx = seq; /* level 6 */
	for (int l = 6; l > level; l--)
		idx = (int64_t)(((__int128)idx * cnt[l - 1]) / cnt[l]);
	return idx;
}

static void alpha5(int64_t k, char *out)
{
	for (int p = 4; p >= 0; p--) {
		out[p] = (char)('a' + k % 26);
		k /= 26;
	}
	out[5] = '\0';
}

void o_synth_name(const o_synth_cfg *c, int level, int64_t index, char *out)
{
	static const char *const pre[6] = { "Dom", "Phy", "Cls", "Ord", "Fam", "Gen" };
	char a[8];
	if (level < 6) {
		alpha5(index, a);
		sprintf(out, "%s%s", pre[level], a);
	} else {
		char g[8];
		alpha5(o_synth_ancestor(c, index, 5), g);
		alpha5(index, a);
		sprintf(out, "Gen%s sp%s", g, a);
	}
}

int o_synth_rdp_mask(const o_synth_cfg *c, int64_t r)
{
	int m = 0;
	for (int k = 0; k < 6; k++)
		if (o_synth_hash(c->read_seed, 5, (uint64_t)r, (uint64_t)k) % 10 != 0)
			m |= 1 << k;
	return m;
}

static const char *const k_rank_names[7] = { "superkingdom", "phylum", "class", "order",
					     "family",       "genus",  "species" };

int o_synth_write_taxdump(const o_synth_cfg *c, const char *dir)
{
	char path[4096], nm[64];
	int64_t cnt[7];
	o_synth_tax_counts(c, cnt);
	snprintf(path, sizeof path, "%s/nodes.dmp", dir);
	FILE *fn = fopen(path, "w");
	snprintf(path, sizeof path, "%s/names.dmp", dir);
	FILE *fm = fopen(path, "w");
	snprintf(path, sizeof path, "%s/gi_taxid_nucl.dmp", dir);
	FILE *fg = fopen(path, "w");
	if (!fn || !fm || !fg)
		return -1;
	fprintf(fn, "1\t|\t1\t|\tno rank\t|\t\t|\t8\t|\t0\t|\t1\t|\t0\t|\t0\t|\t0\t|\t0\t|\t0\t|\t\t|\n");
	fprintf(fm, "1\t|\troot\t|\t\t|\tscientific name\t|\n");
	for (int l = 0; l < 7; l++) {
		for (int64_t k = 0; k < cnt[l]; k++) {
			int64_t id = o_synth_taxid(c, l, k);
			int64_t par = 1;
			if (l > 0)
				par = o_synth_taxid(c, l - 1, (int64_t)(((__int128)k * cnt[l - 1]) / cnt[l]));
			fprintf(fn, "%ld\t|\t%ld\t|\t%s\t|\t\t|\t0\t|\t1\t|\t11\t|\t1\t|\t0\t|\t1\t|\t0\t|\t0\t|\t\t|\n",
				(long)id, (long)par, k_rank_names[l]);
			o_synth_name(c, l, k, nm);
			fprintf(fm, "%ld\t|\t%s\t|\t\t|\tscientific name\t|\n", (long)id, nm);
		}
	}
	/* the reference's -n lookup never finds the last record of names.dmp (ncbitc.c:665): pad it */
	fprintf(fm, "%ld\t|\tzz sentinel\t|\t\t|\tsynonym\t|\n", (long)o_synth_taxid(c, 6, cnt[6] - 1));
	for (int64_t i = 0; i < c->n_seq; i++)
		fprintf(fg, "%ld\t%ld\n", (long)(1000 + i), (long)o_synth_taxid(c, 6, i));
	fclose(fn);
	fclose(fm);
	fclose(fg);
	return 0;
}

static const char k_letters[5] = "ACGT";

int o_synth_write_db_fasta(const o_synth_cfg *c, const char *path, int64_t first, int64_t count)
{
	FILE *f = fopen(path, "w");
	if (!f)
		return -1;
	uint8_t *b = (uint8_t *)malloc((size_t)c->seq_len);
	char *line = (char *)malloc((size_t)c->seq_len + 2);
	for (int64_t i = first; i < first + count; i++) {
		o_synth_db_seq(c, i, b);
		for (int32_t j = 0; j < c->seq_len; j++)
			line[j] = k_letters[b[j]];
		line[c->seq_len] = '\n';
		fprintf(f, ">gi|%ld|syn|S%ld|\n", (long)(1000 + i), (long)i);
		fwrite(line, 1, (size_t)c->seq_len + 1, f);
	}
	free(b);
	free(line);
	fclose(f);
	return 0;
}

int o_synth_write_reads_fasta(const o_synth_cfg *c, const char *path, int64_t first, int64_t count)
{
	FILE *f = fopen(path, "w");
	if (!f)
		return -1;
	uint8_t *b = (uint8_t *)malloc((size_t)c->read_len);
	char *line = (char *)malloc((size_t)c->read_len + 2);
	for (int64_t r = first; r < first + count; r++) {
		o_synth_read(c, r, b, NULL, NULL, NULL);
		for (int32_t j = 0; j < c->read_len; j++)
			line[j] = k_letters[b[j]];
		line[c->read_len] = '\n';
		fprintf(f, ">r%ld\n", (long)r);
		fwrite(line, 1, (size_t)c->read_len + 1, f);
	}
	free(b);
	free(line);
	fclose(f);
	return 0;
}

int o_synth_write_rdp(const o_synth_cfg *c, const char *path, int64_t first, int64_t count)
{
	/* five-tab RDP "allrank" lines (Consensus_BLAST_SOAP_RDP-1.1.pl:126-132) */
	static const char *const rr[6] = { "domain", "phylum", "class", "order", "family", "genus" };
	FILE *f = fopen(path, "w");
	if (!f)
		return -1;
	uint8_t *b = (uint8_t *)malloc((size_t)c->read_len);
	char nm[64];
	for (int64_t r = first; r < first + count; r++) {
		int64_t src;
		o_synth_read(c, r, b, &src, NULL, NULL);
		int mask = o_synth_rdp_mask(c, r);
		fprintf(f, "r%ld\t\t\t\t", (long)r);
		for (int k = 0; k < 6; k++) {
			if (!(mask & (1 << k)))
				continue;
			o_synth_name(c, k, o_synth_ancestor(c, src, k), nm);
			fprintf(f, "\t%s\t%s\t0.9", nm, rr[k]);
		}
		fputc('\n', f);
	}
	free(b);
	fclose(f);
	return 0;
}
