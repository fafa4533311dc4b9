"""Seeded read files for the trim2 parity tests (tests/test_gpu_trim.py) and for the oracle-vs-reference sweep
(oracle/sweep_trim_vs_reference.py): FASTQ and QSEQ text with the quality shapes the running-sum rule distinguishes,
optionally damaged the way real files are."""
import random


def quality_string(rng, n, base):
    """One of the shapes the running-sum rule distinguishes: clean, tail drop, noise, dips, all low, near the cutoff."""
    kind = rng.randrange(6)
    if kind == 0:
        q = [rng.randint(30, 40) for _ in range(n)]
    elif kind == 1:
        cut = rng.randint(n // 2, n)
        q = [rng.randint(30, 40) if j < cut else rng.randint(2, 10) for j in range(n)]
    elif kind == 2:
        q = [rng.randint(2, 40) for _ in range(n)]
    elif kind == 3:
        step = rng.randint(7, 40)
        q = [2 if j % step == 0 else 38 for j in range(n)]
    elif kind == 4:
        q = [rng.randint(2, 12) for _ in range(n)]
    else:
        q = [20 + rng.choice((-2, -1, 0, 1, 2)) for _ in range(n)]
    return "".join(chr(base + v) for v in q)


def fastq_text(seed, n, lmin, lmax):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        L = rng.randint(lmin, lmax)
        seq = "".join(rng.choice("ACGTN" if i % 17 == 0 else "ACGT") for _ in range(L))
        out.append("@M%d:%d@%d extra\n%s\n+\n%s\n" % (seed, i, i % 7, seq, quality_string(rng, L, 33)))
    return "".join(out).encode("latin-1")


def qseq_text(seed, n, lmin, lmax):
    rng = random.Random(seed)
    a, b = [], []
    for i in range(n):
        xy = [str(rng.randint(1, 8)), str(rng.randint(1101, 2316)), str(rng.randint(1000, 20000)), str(rng.randint(1000, 20000))]
        for mate, dst in ((1, a), (2, b)):
            L = rng.randint(lmin, lmax)
            seq = "".join(rng.choice("ACGT.") if rng.random() < 0.02 else rng.choice("ACGT") for _ in range(L))
            dst.append("\t".join(["HWI-X", "12"] + xy + ["TTAGGC", str(mate), seq, quality_string(rng, L, 64), rng.choice("01")]) + "\n")
    return "".join(a).encode("latin-1"), "".join(b).encode("latin-1")


def damaged(rng, text):
    """Damage a read file the way real files are damaged: CRLF ends, a cut tail, blank or short lines in the middle."""
    k = rng.randrange(6)
    if k == 0:
        return text.replace(b"\n", b"\r\n")
    if k == 1 and len(text) > 1:
        return text[:rng.randrange(len(text) // 2, len(text))]
    if k == 2:
        lines = text.split(b"\n")
        for _ in range(5):
            lines.insert(rng.randrange(len(lines)), rng.choice([b"", b"x", b"\t\t", b"@", b"+"]))
        return b"\n".join(lines)
    return text


def random_case(seed, max_fastq, max_qseq):
    """(file a, file b or None, -g or None, -t or None) of one random case."""
    rng = random.Random(seed)
    g = rng.choice([None, "0", "1", "7", "100", "2.5", "abc"])
    t = rng.choice([None, None, "1", "5", "11", "30", "00", "2.7"])
    if rng.random() < 0.5:
        a = damaged(rng, fastq_text(seed, rng.randint(1, max_fastq), rng.choice([20, 60, 70]), rng.choice([71, 150, 400])))
        if not a.startswith(b"@"):
            a = b"@" + a
        b = b"" if rng.random() < 0.5 else None
    else:
        a, b = qseq_text(seed, rng.randint(1, max_qseq), rng.choice([10, 80, 100]), rng.choice([100, 152, 300]))
        first = a.split(b"\n", 1)[0] + b"\n"      # the first line decides the format: keep it intact
        a, b = first + damaged(rng, a[len(first):]), damaged(rng, b)
        if rng.random() < 0.15:
            b = None
    return a, b, g, t
