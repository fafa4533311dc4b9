"""Oracle classify stage: SOAP mode against goldens made by the reference's closed soap ELF,
BLAST mode (spec pgx-blastn v1, parity unpinned) against an independent brute-force
restatement of the same spec in Python."""
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


def diag_hsps(q, s, d, W=28, X=10):
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
        if j - i >= W and i >= covered:
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
            res.append((bl, br, (j - i) + best + bestr, mism))
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
                    for bl, br, score, mism in diag_hsps(qq, s, d):
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
                              str(out), "-num_threads", nt])
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


def test_blast_statistics_and_formatting(oracle_bin, tmp_path):
    # one exact 150-mer against a one-sequence database: S = 150
    import ctypes
    import math
    lib = ctypes.CDLL(os.path.join(os.path.dirname(oracle_bin), "..", "liboracle.so"))

    class St(ctypes.Structure):
        _fields_ = [("lam", ctypes.c_double), ("K", ctypes.c_double), ("H", ctypes.c_double),
                    ("db_len", ctypes.c_int64), ("db_nseq", ctypes.c_int64)]
    lib.o_blast_evalue.restype = ctypes.c_double
    lib.o_blast_bitscore.restype = ctypes.c_double
    lib.o_blast_length_adjust.restype = ctypes.c_int64
    st = St(1.28, 0.46, 0.85, 1000000500, 666667)
    adj = lib.o_blast_length_adjust(ctypes.byref(st), ctypes.c_int64(150))
    # fixed point of ell = (ln K + ln((m-ell)(n-N ell))) / H, floor
    ell = 0.0
    for _ in range(100):
        ell = (math.log(0.46) + math.log((150 - ell) * (1000000500 - 666667 * ell))) / 0.85
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
