/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h). Small utilities. */
#include "o_common.h"
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

void obuf_init(obuf *b)
{
	b->p = NULL;
	b->n = b->cap = 0;
}

void obuf_free(obuf *b)
{
	free(b->p);
	obuf_init(b);
}

static void obuf_reserve(obuf *b, size_t extra)
{
	if (b->n + extra + 1 <= b->cap)
		return;
	size_t nc = b->cap ? b->cap * 2 : 256;
	while (nc < b->n + extra + 1)
		nc *= 2;
	b->p = (char *)realloc(b->p, nc);
	if (!b->p)
		abort();
	b->cap = nc;
}

void obuf_put(obuf *b, const void *s, size_t n)
{
	obuf_reserve(b, n);
	memcpy(b->p + b->n, s, n);
	b->n += n;
	b->p[b->n] = '\0';
}

void obuf_puts(obuf *b, const char *s)
{
	obuf_put(b, s, strlen(s));
}

void obuf_printf(obuf *b, const char *fmt, ...)
{
	va_list ap, ap2;
	va_start(ap, fmt);
	va_copy(ap2, ap);
	int need = vsnprintf(NULL, 0, fmt, ap);
	va_end(ap);
	if (need < 0) {
		va_end(ap2);
		return;
	}
	obuf_reserve(b, (size_t)need);
	vsnprintf(b->p + b->n, (size_t)need + 1, fmt, ap2);
	va_end(ap2);
	b->n += (size_t)need;
}

int obuf_write_file(const obuf *b, const char *path)
{
	FILE *f = fopen(path, "wb");
	if (!f)
		return -1;
	if (b->n && fwrite(b->p, 1, b->n, f) != b->n) {
		fclose(f);
		return -1;
	}
	return fclose(f) ? -1 : 0;
}

char *o_read_file(const char *path, size_t *len)
{
	FILE *f = fopen(path, "rb");
	if (!f)
		return NULL;
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	if (sz < 0) {
		fclose(f);
		return NULL;
	}
	char *p = (char *)malloc((size_t)sz + 1);
	if (!p) {
		fclose(f);
		return NULL;
	}
	size_t got = fread(p, 1, (size_t)sz, f);
	fclose(f);
	p[got] = '\0';
	if (len)
		*len = got;
	return p;
}
