/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h).
 *
 * Checks the shortcut the device takes in spec S3d (csrc/dust.hip: k_dust_trigger) against the DEFINITION restated in
 * o_dust.c: the published algorithm (Morgulis et al. 2006) keeps, per position, the pair count r_w of the window of the last
 * 62 triplets and the length L of that window's longest suffix in which no triplet occurs more than 4 times, and only looks
 * for perfect intervals ending at the position when 10 r_w > 20 L; it then walks the suffixes longer than that suffix.
 * The device does the same walk and lists a read when one of those suffixes scores above the level (an interval above the
 * level exists exactly when a perfect one does: its best sub-interval), then runs the definition on the listed reads alone,
 * between the first and last such position.  This program generates reads (uniform, biased, with noisy repeats of unit
 * 1-6, with an N) and fails if a read with a masked base (definition) is not listed; it also prints how many are listed.
 * usage: fuzz_dust [reads]     (tests/test_oracle_classify.py runs it with 60 000)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
void o_dust_mask(const uint8_t *base, int32_t len, uint8_t *mask);
static int dust_trigger(const uint8_t *base, int len)
{
	int w[64], head = 0, size = 0, cw[64] = { 0 }, cv[64] = { 0 }, rw = 0, rv = 0, L = 0, l = 0, t = 0, trig = 0;
	for (int i = 0; i < len; i++) {
		if (base[i] >= 4) {
			l = 0; t = 0; head = size = 0; memset(cw, 0, sizeof cw); memset(cv, 0, sizeof cv); rw = rv = L = 0;
			continue;
		}
		l++;
		t = ((t << 2) | base[i]) & 63;
		if (l < 3) continue;
		if (size >= 62) {
			int s = w[head]; head = (head + 1) & 63; size--;
			rw -= --cw[s];
			if (L > size) { L--; rv -= --cv[s]; }
		}
		w[(head + size) & 63] = t; size++;
		L++;
		rw += cw[t]++;
		rv += cv[t]++;
		if (cv[t] * 10 > 20 * 2) {
			int s;
			do { s = w[(head + size - L) & 63]; rv -= --cv[s]; L--; } while (s != t);
		}
		if (rw * 10 > L * 20) {
			trig |= 1;
			int c[64], r = rv;
			memcpy(c, cv, sizeof c);
			for (int k = size - L - 1; k >= 0; k--) {
				int tt = w[(head + k) & 63];
				r += c[tt]++;
				if (r * 10 > 20 * (size - k - 1)) { trig |= 2; break; }
			}
		}
	}
	return trig;
}
static uint64_t sd = 88172645463325252ull;
static uint32_t rnd(void) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (uint32_t)(sd >> 11); }
int main(int argc, char **argv)
{
	long n = argc > 1 ? atol(argv[1]) : 100000;
	long bad = 0, nm = 0, nt = 0, nm_u = 0, nt_u = 0, nu = 0;
	uint8_t b[512], m[512];
	for (long it = 0; it < n; it++) {
		int kind = rnd() % 6;
		int len = kind == 0 ? 150 : 20 + rnd() % 300;
		int bias = rnd() % 4;
		for (int i = 0; i < len; i++) {
			uint32_t x = rnd();
			b[i] = (kind == 5 && bias) ? ((x % 10) < 8 ? (x >> 8) % 2 * 3 : (x >> 8) & 3) : (x & 3); /* AT-rich */
		}
		if (kind >= 2 && kind <= 4) { /* repeats with noise */
			int nrep = 1 + rnd() % 3;
			for (int r = 0; r < nrep; r++) {
				int unit = 1 + rnd() % 6, ul[6], s0 = rnd() % len, rl = 4 + rnd() % 60, noise = rnd() % 12;
				for (int u = 0; u < unit; u++) ul[u] = rnd() & 3;
				for (int i = s0; i < s0 + rl && i < len; i++) {
					b[i] = ul[(i - s0) % unit];
					if (noise && rnd() % 12 < (uint32_t)noise / 3) b[i] = rnd() & 3;
				}
			}
		}
		if (kind == 4 && rnd() % 3 == 0) b[rnd() % len] = 4; /* an N */
		o_dust_mask(b, len, m);
		int any = 0;
		for (int i = 0; i < len; i++) any |= m[i];
		int tr = dust_trigger(b, len);
		nm += any; nt += tr & 1;
		static long ns, ns_u;
		ns += (tr >> 1) & 1;
		if (kind == 0) { nu++; nm_u += any; nt_u += tr & 1; ns_u += (tr >> 1) & 1; }
		if (it == n - 1) printf("strong: all %ld, uniform %ld\n", ns, ns_u);
		if (any && !(tr & 2)) {
			if (bad < 5) { printf("COUNTEREXAMPLE len %d: ", len); for (int i = 0; i < len; i++) putchar("ACGTN"[b[i]]); putchar('\n'); }
			bad++;
		}
	}
	printf("reads %ld masked %ld triggered %ld counterexamples %ld | uniform 150: %ld masked %ld triggered %ld\n", n, nm, nt, bad, nu, nm_u, nt_u);
	return bad != 0;
}
