"""Oracle classify stage: SOAP mode against goldens made by the reference's closed soap ELF,
BLAST mode (spec pgx-blastn v2 and its `-ungapped` form v1, parity unpinned) against an independent
brute-force restatement of the same spec in Python, and the gapped extension's score against a plain
dynamic-programming alignment."""
import os
import subprocess

import pytest

from conftest import run_cmd

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def soap_rows(path):
    rows = {}
    for l in open(path):
        f = l.rstrip("\n").split("\t")
        rows.setdefault(f[0], set()).add(tuple(f[1:]))
    return rows


def test_soap_mode_matches_reference_binary(gold, oracle_bin, tmp_path):
    g = os.path.join(gold, "soap")
    out2, unm, out1 = tmp_path / "r2.txt", tmp_path / "unm.txt", tmp_path / "r1.txt"
    rc, _, _ = run_cmd([oracle_bin, "soap", "-a", os.path.join(g, "reads.fa"), "-D", os.path.join(g, "ref.fa.index"),
                        "-o", str(out2), "-u", str(unm), "-r", "2", "-M", "4"])
    assert rc == 0
    # -r 2: identical rows per read as a SET (row order among equal-best hits is suffix-array
    # order inside the ELF and is a documented deviation)
    want, got = soap_rows(os.path.join(g, "out_r2.txt")), soap_rows(str(out2))
    assert got == want
    assert sum(len(v) for v in got.values()) == 609
    assert unm.read_bytes() == open(os.path.join(g, "unmapped_r2.txt"), "rb").read()
    # -r 1: identical bytes for reads with a unique best hit; same hit count column otherwise
    rc, _, _ = run_cmd([oracle_bin, "soap", "-a", os.path.join(g, "reads.fa"), "-D", os.path.join(g, "ref.fa.index"),
                        "-o", str(out1), "-r", "1", "-M", "4"])
    assert rc == 0
    w1 = [l for l in open(os.path.join(g, "out_r1.txt"))]
    g1 = [l for l in open(out1)]
    assert len(w1) == len(g1)
    for a, b in zip(w1, g1):
        fa, fb = a.split("\t"), b.split("\t")
        assert fa[0] == fb[0] and fa[3] == fb[3]
        if fa[3] == "1":
            assert a == b


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_soap_match_modes_match_reference_binary(gold, oracle_bin, tmp_path, mode):
    import gzip
    g = os.path.join(gold, "soap")
    out, unm, want = tmp_path / "m.txt", tmp_path / "u.txt", tmp_path / "want.txt"
    want.write_bytes(gzip.open(os.path.join(g, "out_M%d.txt.gz" % mode), "rb").read())
    rc, _, _ = run_cmd([oracle_bin, "soap", "-a", os.path.join(g, "reads_short.fa"), "-D", os.path.join(g, "ref.fa.index"), "-o", str(out),
                        "-u", str(unm), "-r", "2", "-M", str(mode)])
    assert rc == 0
    assert soap_rows(str(out)) == soap_rows(str(want))
    assert unm.read_bytes() == open(os.path.join(g, "unmapped_M%d.txt" % mode), "rb").read()


PE_SETS = [("r2", "pe_a.fa", "pe_b.fa", 400, 600, 2, "pe_%s_r2.txt.gz", 458, 1324),
           ("r0", "pe_a.fa", "pe_b.fa", 400, 600, 0, "pe_%s_r0.txt.gz", 178, 62),
           ("sweep", "pe_sweep_a.fa", "pe_sweep_b.fa", 300, 700, 2, "pe_sweep_%s.txt.gz", 2332, 171)]


@pytest.mark.parametrize("case", PE_SETS, ids=[c[0] for c in PE_SETS])
def test_soap_paired_end_matches_reference_binary(gold, oracle_bin, tmp_path, case):
    """`soap -a A -b B -2 unpaired -m MIN -x MAX` (soap.man:29-50) against the rows the closed ELF printed for the pairs of
    oracle/gen_goldens_soap.py (generate_pe): inserts either side of both limits, mates of unequal length, swapped mates,
    outward and same-strand pairs, 0-3 planted mismatches, a mate that maps nowhere, N-rich mates, and a sweep of mate
    lengths 27-120 with two mismatches at swept offsets (the order of the two entries of a row).  Rows as sets per read
    (the order among a read's rows is the ELF's suffix-array order, a documented deviation); the unmapped list byte for byte."""
    import gzip
    tag, a, b, lo, hi, r, names, n_paired, n_unpaired = case
    g = os.path.join(gold, "soap")
    o, u2, un = tmp_path / "o.txt", tmp_path / "u2.txt", tmp_path / "un.txt"
    rc, _, _ = run_cmd([oracle_bin, "soap", "-a", os.path.join(g, a), "-b", os.path.join(g, b), "-D", os.path.join(g, "ref.fa.index"),
                        "-o", str(o), "-2", str(u2), "-u", str(un), "-m", str(lo), "-x", str(hi), "-r", str(r), "-M", "4"])
    assert rc == 0
    for kind, path, n in (("paired", o, n_paired), ("unpaired", u2, n_unpaired)):
        want = tmp_path / ("want_" + kind)
        want.write_bytes(gzip.open(os.path.join(g, names % kind), "rb").read())
        assert soap_rows(str(path)) == soap_rows(str(want)), kind
        assert sum(1 for _ in open(path)) == n, kind
    assert un.read_bytes() == gzip.open(os.path.join(g, names % "unmapped"), "rb").read()


def test_soap_paired_end_refuses_what_is_not_restated(oracle_bin, gold, tmp_path):
    """Mates of fewer than 27 bases (the ELF itself crashes on some of them in paired-end runs) are refused, not guessed."""
    g = os.path.join(gold, "soap")
    a, b = tmp_path / "a.fa", tmp_path / "b.fa"
    a.write_text(">p/1\nACGTACGTACGTACGTACGTACG\n")
    b.write_text(">p/2\nACGTACGTACGTACGTACGTACGTACGTACGT\n")
    rc, _, err = run_cmd([oracle_bin, "soap", "-a", str(a), "-b", str(b), "-D", os.path.join(g, "ref.fa.index"), "-o", str(tmp_path / "o"),
                          "-2", str(tmp_path / "u")])
    assert rc != 0 and b"27" in err


# ---------------------------------------------------------------- BLAST mode brute force
def read_fa(p):
    out = []
    for l in open(p):
        l = l.strip()
        if l.startswith(">"):
            out.append([l[1:].split()[0], ""])
        else:
            out[-1][1] += l.upper()
    return out


def dust_mask(seq, W=64, level=20):
    """S3d: the definition of symmetric DUST (Morgulis et al. 2006): triplet intervals of at most W - 2 triplets whose score
    (sum of c(c-1)/2 over triplet values, divided by triplets - 1) exceeds level / 10 and is not beaten by a sub-interval."""
    from fractions import Fraction
    n, nt = len(seq), len(seq) - 2
    mask = [False] * n
    if nt < 2:
        return mask
    trip = [seq[i:i + 3] if all(c in "ACGT" for c in seq[i:i + 3]) else None for i in range(nt)]
    best = {}   # (a, b) -> highest score of any sub-interval of [a, b] with at least two triplets, or None
    for a in range(nt - 1, -1, -1):
        cnt, r = {}, 0
        for b in range(a, min(nt, a + W - 2)):
            if trip[b] is None:
                break
            r += cnt.get(trip[b], 0)
            cnt[trip[b]] = cnt.get(trip[b], 0) + 1
            s = Fraction(r, b - a) if b > a else None
            subs = [x for x in (best.get((a + 1, b)), best.get((a, b - 1))) if x is not None]
            sub = max(subs) if subs else None
            if s is not None and s * 10 > level and (sub is None or sub <= s):
                for k in range(a, b + 3):
                    mask[k] = True
            cands = [x for x in (s, sub) if x is not None]
            best[(a, b)] = max(cands) if cands else None
    return mask


def run_is_seed(qmask, a, b, W=28):
    if b - a < W:
        return False
    if qmask is None:
        return True
    clean = 0
    for k in range(a, b):
        clean = 0 if qmask[k] else clean + 1
        if clean >= W:
            return True
    return False


def diag_hsps(q, s, d, W=28, X=10, qmask=None):
    lo, hi = max(0, -d), min(len(q), len(s) - d)
    res = []
    if hi - lo < W:
        return res
    m = [q[k] in "ACGT" and q[k] == s[k + d] for k in range(lo, hi)]
    mm = lambda k: m[k - lo]
    covered, i = lo, lo
    while i < hi:
        if not mm(i):
            i += 1
            continue
        j = i
        while j < hi and mm(j):
            j += 1
        if run_is_seed(qmask, i, j, W) and i >= covered:
            best = cur = 0
            bl = i
            for k in range(i - 1, lo - 1, -1):
                cur += 1 if mm(k) else -2
                if cur > best:
                    best, bl = cur, k
                elif best - cur > X:
                    break
            bestr = cur = 0
            br = j - 1
            for k in range(j, hi):
                cur += 1 if mm(k) else -2
                if cur > bestr:
                    bestr, br = cur, k
                elif bestr - cur > X:
                    break
            mism = sum(1 for k in range(bl, br + 1) if not mm(k))
            # S3b anchor: first base of the run of matches that holds the last matching position at or before the middle
            anchor = bl + (br - bl) // 2
            while not mm(anchor):
                anchor -= 1
            while anchor > bl and mm(anchor - 1):
                anchor -= 1
            res.append((bl, br, (j - i) + best + bestr, mism, anchor))
            covered = br + 1
        i = j
    return res


def brute_force(queries, db):
    hits = set()
    for qn, q in queries:
        L = len(q)
        rcq = "".join(COMP.get(c, "N") for c in reversed(q))
        for sn, s in db:
            for strand, qq in ((0, q), (1, rcq)):
                for d in range(-(L - 28), len(s) - 28 + 1):
                    for bl, br, score, mism, _seed in diag_hsps(qq, s, d):
                        if strand == 0:
                            t = (bl + 1, br + 1, bl + d + 1, br + d + 1)
                        else:
                            t = (L - br, L - bl, br + d + 1, bl + d + 1)
                        hits.add((qn, sn, br - bl + 1, mism) + t)
    return hits


def test_blast_mode_equals_brute_force_of_the_spec(oracle_bin, tmp_path):
    db, rd, out = tmp_path / "db.fa", tmp_path / "reads.fa", tmp_path / "hits.tsv"
    shape = ["--n-seq", "24", "--seq-len", "330", "--n-genus", "3", "--read-len", "150"]
    assert run_cmd([oracle_bin, "synth", "db", "--out", str(db)] + shape)[0] == 0
    assert run_cmd([oracle_bin, "synth", "reads", "--out", str(rd), "--count", "14"] + shape)[0] == 0
    # sprinkle ambiguity codes and a lower-case stretch
    txt = db.read_text().split("\n")
    txt[1] = txt[1][:100] + "NNNN" + txt[1][104:200].lower() + txt[1][200:]
    txt[3] = txt[3][:50] + "R" + txt[3][51:]
    db.write_text("\n".join(txt))
    r = rd.read_text().split("\n")
    r[1] = r[1][:70] + "N" + r[1][71:]
    rd.write_text("\n".join(r))
    want = brute_force(read_fa(str(rd)), read_fa(str(db)))
    for nt in ("1", "3"):
        rc, so, se = run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out",
                              str(out), "-num_threads", nt, "-ungapped"])
        assert rc == 0, se
        got = set()
        lines = out.read_text().splitlines()
        for l in lines:
            f = l.split("\t")
            assert len(f) == 12 and f[5] == "0"
            got.add((f[0], f[1], int(f[3]), int(f[4]), int(f[6]), int(f[7]), int(f[8]), int(f[9])))
            assert f[2] == "%.2f" % (100.0 * (int(f[3]) - int(f[4])) / int(f[3]))
        assert len(got) == len(lines)
        assert got == want
        assert len(got) > 40
        if nt == "1":
            first = lines
        else:
            assert lines == first  # thread count never changes the bytes


# ---------------------------------------------------------------- gapped stage (spec v2, S3b / S3c)
X_GAP, LAG, NONE = 54, 19, -10 ** 9


def greedy(a, b, prune=False):
    """Zhang, Schwartz, Wagner, Miller (2000), fig. 4, with the rules oracle/o_gapped.c states; a, b are the letters
    on one side of the anchor, nearest first.  Returns (i, j, doubled score, mismatches, gap openings, gap columns)."""
    M, N = len(a), len(b)
    ok = lambda x, y: x in "ACGT" and x == y
    i = 0
    while i < M and i < N and ok(a[i], b[i]):
        i += 1
    if i == M or i == N:
        return i, i, 2 * i, 0, 0, 0
    prev, L, U = {0: i}, 0, 0
    T, best, moves = [2 * i], (2 * i, 0, 0, i), {}
    d = 0
    while d < 1000:
        d += 1
        tcmp = T[d - LAG] if d - LAG >= 0 else 0
        cur = {}
        for k in range(L - 1, U + 2):
            v, par = NONE, 0
            if L <= k <= U and k in prev:
                v, par = prev[k] + 1, 0
            if k - 1 >= L and (k - 1) in prev and prev[k - 1] + 1 > v:
                v, par = prev[k - 1] + 1, 1
            if k + 1 <= U and (k + 1) in prev and prev[k + 1] > v:
                v, par = prev[k + 1], 2
            ii, jj = v, v - k
            if v == NONE or ii > M or jj > N or jj < 0 or ii + jj - 6 * d < tcmp - 2 * X_GAP:
                continue
            if prune and min(2 * M - k, 2 * N + k) - 6 * d <= best[0]:
                continue
            i0 = ii
            while ii < M and jj < N and ok(a[ii], b[jj]):
                ii, jj = ii + 1, jj + 1
            cur[k] = ii
            moves[(d, k)] = (par, ii > i0)
            if ii + jj - 6 * d > best[0]:
                best = (ii + jj - 6 * d, d, k, ii)
        T.append(best[0])
        if not cur:
            break
        prev, L, U = cur, min(cur), max(cur)
    s2, bd, k, bi = best
    path = []
    for dd in range(bd, 0, -1):
        par, slid = moves[(dd, k)]
        path.append((par, slid))
        k = k - 1 if par == 1 else (k + 1 if par == 2 else k)
    path.reverse()
    mism = sum(1 for p, _ in path if p == 0)
    gaps = len(path) - mism
    opens, last = 0, (0, True)
    for par, slid in path:
        if par != 0 and not (par == last[0] and not last[1]):
            opens += 1
        last = (par, slid)
    return bi, bi - best[2], s2, mism, opens, gaps


def dp_best(a, b):
    """Best score of an alignment of a prefix of a with a prefix of b (match 1, mismatch -2, gap column -2.5), doubled."""
    import numpy as np
    M, N = len(a), len(b)
    H = np.full((M + 1, N + 1), -10 ** 9, dtype=np.int64)
    H[0, :] = -5 * np.arange(N + 1)
    H[:, 0] = -5 * np.arange(M + 1)
    bv = np.frombuffer(b.encode(), dtype=np.uint8)
    for i in range(1, M + 1):
        sub = np.where((bv == ord(a[i - 1])) & (a[i - 1] in "ACGT"), 2, -4)
        diag = H[i - 1, :-1] + sub
        up = H[i - 1, 1:] - 5
        row = np.maximum(diag, up)
        # gaps along the row: prefix scan of max(row[j], row[j-1] - 5)
        cur = H[i, 0]
        for j in range(1, N + 1):
            cur = max(row[j - 1], cur - 5)
            H[i, j] = cur
    return int(H.max())


def mutate(rng, s, subs, indels):
    s = list(s)
    for _ in range(subs):
        p = rng.randrange(len(s))
        s[p] = rng.choice([c for c in "ACGT" if c != s[p]])
    for _ in range(indels):
        p = rng.randrange(1, len(s) - 1)
        if rng.random() < 0.5:
            del s[p:p + rng.choice([1, 1, 2, 3])]
        else:
            s[p:p] = [rng.choice("ACGT") for _ in range(rng.choice([1, 1, 2, 3]))]
    return "".join(s)


def test_greedy_extension_is_optimal_and_bound_cut_is_neutral(oracle_bin):
    """The oracle's C greedy extension equals the Python restatement field by field, with and without the bound cut,
    and its score is the optimum of a plain dynamic-programming alignment (sequences this similar never drop by X)."""
    import ctypes
    import random
    lib = ctypes.CDLL(os.path.join(os.path.dirname(oracle_bin), "..", "liboracle.so"))

    class Ext(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int32) for n in ("i", "j", "s2", "d", "mism", "gap_s", "gap_q", "gapopen")]
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    rng = random.Random(11)
    n_gapped = 0
    for case in range(160):
        L = rng.choice([40, 90, 150, 150, 220])
        a = "".join(rng.choice("ACGT") for _ in range(L))
        b = mutate(rng, a, rng.randrange(0, 1 + L // 14), rng.randrange(0, 4)) + "".join(rng.choice("ACGT") for _ in range(rng.choice([0, 30])))
        if case % 9 == 0:
            a = a[:L // 2] + "N" + a[L // 2 + 1:]
        if case % 10 == 0:
            b = b[:len(b) * 2 // 3]   # the subject ends first
        want = greedy(a, b)
        assert greedy(a, b, prune=True) == want
        ab = bytes(code.get(c, 4) for c in a)
        bb = bytes(code.get(c, 4) for c in b)
        for prune in (0, 1):
            e = Ext()
            lib.o_greedy_extend(ab, len(a), bb, len(b), 1, prune, ctypes.byref(e))
            assert (e.i, e.j, e.s2, e.mism, e.gapopen, e.gap_s + e.gap_q) == want, (case, prune)
        # backwards over the reversed arrays: the same extension
        e = Ext()
        ra, rb = ab[::-1], bb[::-1]
        pa = (ctypes.c_char * len(ra)).from_buffer_copy(ra)
        pb = (ctypes.c_char * len(rb)).from_buffer_copy(rb)
        lib.o_greedy_extend(ctypes.byref(pa, len(ra) - 1), len(a), ctypes.byref(pb, len(rb) - 1), len(b), -1, 1, ctypes.byref(e))
        assert (e.i, e.j, e.s2, e.mism, e.gapopen, e.gap_s + e.gap_q) == want, case
        i, j, s2, mism, opens, gaps = want
        assert s2 == i + j - 6 * (mism + gaps) and opens <= gaps
        assert s2 == dp_best(a, b), case
        n_gapped += gaps > 0
    assert n_gapped > 60


def brute_force_v2(queries, db, dust=True):
    """Spec v2: the initial HSPs of v1 (seeds: exact runs with a 28-base window free of DUST-masked query bases, S3d), each
    extended with gaps from its anchor (S3b), then the hits of one (query, subject) that describe one alignment reduced to
    the first in the S5 order (S3c)."""
    rows = []
    for qn, q in queries:
        Lq = len(q)
        rcq = "".join(COMP.get(c, "N") for c in reversed(q))
        mf = dust_mask(q) if dust else None
        masks = (mf, mf[::-1] if mf is not None else None)
        for sn, s in db:
            group = []
            for strand, qq in ((0, q), (1, rcq)):
                for d in range(-(Lq - 28), len(s) - 28 + 1):
                    for _bl, _br, _score, _mism, seed in diag_hsps(qq, s, d, qmask=masks[strand]):
                        qa, sa = seed, seed + d
                        li, lj, ls2, lm, lo, lg = greedy(qq[:qa][::-1], s[:sa][::-1])
                        ri, rj, rs2, rm, ro, rg = greedy(qq[qa:], s[sa:])
                        bl, br, sl, sr = qa - li, qa + ri - 1, sa - lj, sa + rj - 1
                        gaps = lg + rg
                        length = ((br - bl + 1) + (sr - sl + 1) + gaps) // 2
                        if strand == 0:
                            t = (bl + 1, br + 1, sl + 1, sr + 1)
                        else:
                            t = (Lq - br, Lq - bl, sr + 1, sl + 1)
                        group.append(dict(score=(ls2 + rs2) >> 1, q0=t[0], q1=t[1], s0=t[2], s1=t[3], mm=lm + rm,
                                          go=lo + ro, length=length, gaps=gaps))
            group.sort(key=lambda h: (-h["score"], h["q0"], h["q1"], h["s0"], h["s1"], h["mm"], h["go"]))
            for a in range(len(group)):
                A = group[a]
                am = A["s0"] > A["s1"]
                drop = False
                for B in group[:a]:
                    if (B["s0"] > B["s1"]) != am:
                        continue
                    alo, ahi = min(A["s0"], A["s1"]), max(A["s0"], A["s1"])
                    blo, bhi = min(B["s0"], B["s1"]), max(B["s0"], B["s1"])
                    if ((A["q0"], A["s0"]) == (B["q0"], B["s0"]) or (A["q1"], A["s1"]) == (B["q1"], B["s1"])
                            or (A["q0"] >= B["q0"] and A["q1"] <= B["q1"] and alo >= blo and ahi <= bhi)):
                        drop = True
                        break
                if not drop:
                    rows.append((qn, sn, "%.2f" % (100.0 * (A["length"] - A["mm"] - A["gaps"]) / A["length"]), A["length"], A["mm"],
                                 A["go"], A["q0"], A["q1"], A["s0"], A["s1"]))
    return set(rows), len(rows)


def test_blast_mode_v2_equals_brute_force_of_the_spec(oracle_bin, tmp_path):
    import random
    db, rd, out = tmp_path / "db.fa", tmp_path / "reads.fa", tmp_path / "hits.tsv"
    shape = ["--n-seq", "24", "--seq-len", "330", "--n-genus", "3", "--read-len", "150"]
    assert run_cmd([oracle_bin, "synth", "db", "--out", str(db)] + shape)[0] == 0
    assert run_cmd([oracle_bin, "synth", "reads", "--out", str(rd), "--count", "16"] + shape)[0] == 0
    txt = db.read_text().split("\n")
    txt[1] = txt[1][:100] + "NNNN" + txt[1][104:200].lower() + txt[1][200:]
    txt[3] = txt[3][:50] + "R" + txt[3][51:]
    db.write_text("\n".join(txt))
    # 454-style reads: every other read gets insertions and deletions
    rng = random.Random(3)
    r = rd.read_text().split("\n")
    for k in range(1, len(r), 2):
        if r[k] and (k // 2) % 2 == 1:
            r[k] = mutate(rng, r[k], 1, rng.randrange(1, 4))
    r[1] = r[1][:70] + "N" + r[1][71:]
    rd.write_text("\n".join(r))
    want, n_want = brute_force_v2(read_fa(str(rd)), read_fa(str(db)))
    assert n_want == len(want)
    outs = []
    for flags in (["-num_threads", "1"], ["-num_threads", "3"], ["-num_threads", "2", "-no_prune"]):
        rc, so, se = run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(out)] + flags)
        assert rc == 0, se
        lines = out.read_text().splitlines()
        got = set()
        for l in lines:
            f = l.split("\t")
            assert len(f) == 12
            got.add((f[0], f[1], f[2], int(f[3]), int(f[4]), int(f[5]), int(f[6]), int(f[7]), int(f[8]), int(f[9])))
        assert len(got) == len(lines)
        assert got == want
        outs.append(lines)
    assert outs[0] == outs[1] == outs[2]   # neither threads nor the bound cut change a byte
    assert sum(1 for l in outs[0] if l.split("\t")[5] != "0") > 20   # gapopen > 0 rows exist
    assert len(want) > 40


def test_dust_mask_and_masked_seeds(oracle_bin, tmp_path):
    """S3d: the oracle's DUST mask equals the Python restatement of the definition on random and low-complexity sequences
    (homopolymers, di- and tri-nucleotide repeats, N's), and the tables with and without `-dust no` equal the brute force."""
    import ctypes
    import random
    lib = ctypes.CDLL(os.path.join(os.path.dirname(oracle_bin), "..", "liboracle.so"))
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    rng = random.Random(4)

    def lowc(n):
        kind = rng.randrange(5)
        if kind == 0:
            return rng.choice("ACGT") * n
        if kind == 1:
            u = rng.choice(["AC", "GT", "AT", "CG", "AG"])
            return (u * n)[:n]
        if kind == 2:
            u = "".join(rng.choice("ACGT") for _ in range(3))
            return (u * n)[:n]
        if kind == 3:
            return "".join(rng.choice("AAAAAAC") for _ in range(n))
        return "".join(rng.choice("ACGT") for _ in range(n))
    n_masked = 0
    seqs = []
    for case in range(120):
        L = rng.choice([5, 30, 64, 150, 150, 300])
        s = "".join(rng.choice("ACGT") for _ in range(L))
        for _ in range(rng.choice([0, 1, 1, 2])):
            n = rng.choice([6, 7, 8, 12, 20, 40, 70])
            if n < L:
                p_ = rng.randrange(0, L - n)
                s = s[:p_] + lowc(n) + s[p_ + n:]
        if case % 7 == 0 and L > 10:
            s = s[:L // 3] + "N" + s[L // 3 + 1:]
        want = dust_mask(s)
        buf = (ctypes.c_uint8 * max(1, len(s)))()
        lib.o_dust_mask(bytes(code.get(c, 4) for c in s), len(s), buf)
        assert [bool(x) for x in buf[:len(s)]] == want, (case, s)
        n_masked += any(want)
        seqs.append(s)
    assert n_masked > 25
    assert not any(dust_mask("".join(rng.choice("ACGT") for _ in range(150)))[:0])
    assert all(dust_mask("A" * 7)) and not any(dust_mask("A" * 6))   # seven of one letter: five triplets, 10 / 4 > 2
    # tables: a small database that holds the low-complexity stretches too, reads cut from it
    dbs = [s for s in seqs if len(s) >= 150][:14]
    db, rd, out = tmp_path / "db.fa", tmp_path / "reads.fa", tmp_path / "hits.tsv"
    db.write_text("".join(">gi|%d|x|d%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(dbs)))
    reads = []
    for i in range(30):
        s = rng.choice(dbs)
        o = rng.randrange(0, len(s) - 100 + 1)
        w = list(s[o:o + rng.choice([100, 150])])
        for p_ in rng.sample(range(len(w)), rng.choice([0, 1, 2])):
            w[p_] = rng.choice("ACGT")
        w = "".join(w)
        if i % 2:
            w = "".join(COMP.get(c, "N") for c in reversed(w))
        reads.append(">m%d\n%s\n" % (i, w))
    rd.write_text("".join(reads))
    got = {}
    for flags, dust in ((["-dust", "no"], False), ([], True)):
        assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(out)] + flags)[0] == 0
        rows = set()
        for l in out.read_text().splitlines():
            f = l.split("\t")
            rows.add((f[0], f[1], f[2], int(f[3]), int(f[4]), int(f[5]), int(f[6]), int(f[7]), int(f[8]), int(f[9])))
        want, _n = brute_force_v2(read_fa(str(rd)), read_fa(str(db)), dust=dust)
        assert rows == want, dust
        got[dust] = rows
    assert got[True] != got[False] and len(got[True]) > 10   # the mask removes seeds, not everything


def test_blast_statistics_and_formatting(oracle_bin, tmp_path):
    # one exact 150-mer against a one-sequence database: S = 150
    import ctypes
    import math
    lib = ctypes.CDLL(os.path.join(os.path.dirname(oracle_bin), "..", "liboracle.so"))

    class St(ctypes.Structure):
        _fields_ = [("lam", ctypes.c_double), ("K", ctypes.c_double), ("H", ctypes.c_double),
                    ("db_len", ctypes.c_int64), ("db_nseq", ctypes.c_int64), ("alpha", ctypes.c_double), ("beta", ctypes.c_double)]
    lib.o_blast_evalue.restype = ctypes.c_double
    lib.o_blast_bitscore.restype = ctypes.c_double
    lib.o_blast_length_adjust.restype = ctypes.c_int64
    st = St()
    lib.o_blast_stats_init(ctypes.byref(st), ctypes.c_int64(1000000500), ctypes.c_int64(666667), 0)
    assert (st.lam, st.K, st.H, st.beta) == (1.28, 0.46, 0.85, 0.0)
    adj = lib.o_blast_length_adjust(ctypes.byref(st), ctypes.c_int64(150))
    # ungapped: fixed point of ell = (ln K + ln((m-ell)(n-N ell))) / H, floor
    ell = 0.0
    for _ in range(100):
        ell = (math.log(0.46) + math.log((150 - ell) * (1000000500 - 666667 * ell))) / 0.85
    assert adj == int(ell)
    # gapped (spec v2): ell = alpha / lambda * (ln K + ln(...)) + beta with alpha 1.5, beta -2
    lib.o_blast_stats_init(ctypes.byref(st), ctypes.c_int64(1000000500), ctypes.c_int64(666667), 1)
    assert (st.alpha, st.beta) == (1.5, -2.0)
    adj = lib.o_blast_length_adjust(ctypes.byref(st), ctypes.c_int64(150))
    ell = 0.0
    for _ in range(100):
        ell = 1.5 / 1.28 * (math.log(0.46) + math.log((150 - ell) * (1000000500 - 666667 * ell))) - 2.0
    assert adj == int(ell)
    bits = lib.o_blast_bitscore(ctypes.byref(st), ctypes.c_int32(150))
    assert abs(bits - (1.28 * 150 - math.log(0.46)) / math.log(2)) < 1e-9
    buf = ctypes.create_string_buffer(32)
    for val, want in ((0.0, "0.0"), (1e-200, "0.0"), (1.3e-100, "1e-100"), (2.4e-15, "2e-15"), (0.00091, "0.001"),
                      (0.05, "0.050"), (0.5, "0.50"), (5.04, "5.0"), (12.0, "   12")):
        lib.o_blast_format_evalue(ctypes.c_double(val), buf)
        assert buf.value.decode() == want, val
    for val, want in ((937.7, " 937"), (1402.2, "1402"), (87.94, "87.9"), (52.8, "52.8"), (99.95, "  99"), (5.51, " 5.5"),
                      (12345.6, "1.235e+04")):
        lib.o_blast_format_bitscore(ctypes.c_double(val), buf)
        assert buf.value.decode() == want, val


def test_synthetic_workload_shape(oracle_bin, tmp_path):
    shape = ["--n-seq", "400", "--seq-len", "300", "--n-genus", "20"]
    d = tmp_path / "Tax_class"
    d.mkdir()
    assert run_cmd([oracle_bin, "synth", "taxdump", "--out", str(d)] + shape)[0] == 0
    assert run_cmd([oracle_bin, "tax_class", "-c"], cwd=d)[0] == 0
    rc, so, _ = run_cmd([oracle_bin, "tax_class", "-s", "1000"], cwd=d)
    ranks = [l.split(" | ")[2] for l in so.decode().splitlines()]
    assert ranks == ["species", "genus", "family", "order", "class", "phylum"]  # domain's parent is the root
    hits = tmp_path / "h.tsv"
    hits.write_text("r0\tgi|1000|syn|S0|\t100.00\t150\t0\t0\t1\t150\t1\t150\t1e-70\t 278\n"
                    "r1\tgi|1399|syn|S399|\t100.00\t150\t0\t0\t1\t150\t1\t150\t1e-70\t 278\n")
    out = tmp_path / "hc.tsv"
    assert run_cmd([oracle_bin, "taxcollector", "-f", str(hits), "-o", str(out), "-d", str(d)])[0] == 0
    l0, l1 = out.read_text().splitlines()
    assert l0.split("\t")[1] == "[0]Domaaaaa;[1]Phyaaaaa;[2]Clsaaaaa;[3]Ordaaaaa;[4]Famaaaaa;[5]Genaaaaa;[6]Genaaaaa_spaaaaa;"
    assert l1.split("\t")[1].endswith("[5]Genaaaat;[6]Genaaaat_spaaapj;")
    rdp = tmp_path / "rdp.tsv"
    assert run_cmd([oracle_bin, "synth", "rdp", "--out", str(rdp), "--count", "50"] + shape)[0] == 0
    for i, l in enumerate(rdp.read_text().splitlines()):
        rid, rest = l.split("\t\t\t\t\t")
        assert rid == "r%d" % i
        f = rest.split("\t")
        assert len(f) % 3 == 0 and set(f[1::3]) <= {"domain", "phylum", "class", "order", "family", "genus"}


def test_dust_trigger_is_necessary_for_a_masked_base():
    """The device runs the DUST definition only on reads in which the published algorithm's own trigger (10 r_w > 20 L) fires
    somewhere (csrc/dust.hip).  oracle/fuzz_dust.c: no read with a masked base (definition, o_dust.c) lacks a trigger."""
    import subprocess
    from conftest import ORACLE_DIR
    subprocess.check_call(["make", "-C", ORACLE_DIR, "bin/fuzz_dust"], stdout=subprocess.DEVNULL)
    p = subprocess.run([os.path.join(ORACLE_DIR, "bin", "fuzz_dust"), "60000"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "counterexamples 0" in p.stdout
