/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h). FASTA -> base arrays. */
#include "o_classify.h"
#include <stdlib.h>
#include <string.h>

static uint8_t code_of(char c)
{
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': case 'U': case 'u': return 3;
	default: return O_AMB;
	}
}

int o_seqset_from_text(o_seqset *s, const char *text, size_t len)
{
	memset(s, 0, sizeof *s);
	size_t cap_seq = 1024, cap_base = len + 1;
	s->header = (char **)malloc(cap_seq * sizeof(char *));
	s->off = (int64_t *)malloc((cap_seq + 1) * sizeof(int64_t));
	s->base = (uint8_t *)malloc(cap_base);
	size_t i = 0;
	int in_seq = 0;
	while (i < len) {
		size_t e = i;
		while (e < len && text[e] != '\n')
			e++;
		size_t ll = e - i;
		if (ll && text[i + ll - 1] == '\r')
			ll--;
		if (ll && text[i] == '>') {
			if ((size_t)s->nseq == cap_seq) {
				cap_seq *= 2;
				s->header = (char **)realloc(s->header, cap_seq * sizeof(char *));
				s->off = (int64_t *)realloc(s->off, (cap_seq + 1) * sizeof(int64_t));
			}
			char *h = (char *)malloc(ll);
			memcpy(h, text + i + 1, ll - 1);
			h[ll - 1] = '\0';
			s->header[s->nseq] = h;
			s->off[s->nseq] = s->total;
			s->nseq++;
			in_seq = 1;
		} else if (in_seq) {
			for (size_t k = 0; k < ll; k++) {
				char c = text[i + k];
				if (c == ' ' || c == '\t')
					continue;
				s->base[s->total++] = code_of(c);
			}
		}
		i = e + 1;
	}
	s->off[s->nseq] = s->total;
	return 0;
}

int o_seqset_read_fasta(o_seqset *s, const char *path)
{
	size_t len;
	char *t = o_read_file(path, &len);
	if (!t) {
		memset(s, 0, sizeof *s);
		return -1;
	}
	int rc = o_seqset_from_text(s, t, len);
	free(t);
	return rc;
}

void o_seqset_free(o_seqset *s)
{
	for (int64_t i = 0; i < s->nseq; i++)
		free(s->header[i]);
	free(s->header);
	free(s->off);
	free(s->base);
	memset(s, 0, sizeof *s);
}

size_t o_seq_id(const char *header, char *out, size_t cap)
{
	size_t n = 0;
	while (header[n] && header[n] != ' ' && header[n] != '\t' && n + 1 < cap) {
		out[n] = header[n];
		n++;
	}
	out[n] = '\0';
	return n;
}
