#!/usr/bin/env python3
"""Throughput of the trim2 drop-in on a synthetic FASTQ / QSEQ file (PGX_TRIM_TIMES=1 prints the stage times), with
the oracle (C restatement, one core) timed on the same file.  Usage: python3 tools/probe_trim.py [records] [fastq|qseq]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PGX_TRIM_TIMES"] = "1"


def qualities(rng, n, L, base):
    q = rng.integers(25, 41, size=(n, L), dtype=np.int16)
    cut = rng.integers(L // 2, L + 1, size=n)
    q[np.arange(L)[None, :] >= cut[:, None]] = 4  # tail drop
    noisy = rng.random(n) < 0.2
    q[noisy] = rng.integers(2, 41, size=(int(noisy.sum()), L), dtype=np.int16)
    return (q + base).astype(np.uint8)


def fastq_file(path, n, L=150, seed=1):
    rng = np.random.default_rng(seed)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
    qual = qualities(rng, n, L, 33)
    hdr = np.char.add(np.char.add("@HWI-ST:7:1101:", np.arange(n).astype(str)), ":2000 1:N:0")
    w = max(len(h) for h in hdr[-3:])
    hb = np.frombuffer(np.char.ljust(hdr, w).astype("S%d" % w).tobytes(), dtype=np.uint8).reshape(n, w)
    nl = np.full((n, 1), 10, np.uint8)
    plus = np.full((n, 1), ord("+"), np.uint8)
    rec = np.concatenate([hb, nl, seq, nl, plus, nl, qual, nl], axis=1)
    rec.tofile(path)
    return rec.size


def qseq_files(pa, pb, n, L=150, seed=1):
    rng = np.random.default_rng(seed)
    total = 0
    for mate, path in ((1, pa), (2, pb)):
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
        qual = qualities(rng, n, L, 64)
        head = np.char.add(np.char.add("HWI-ST\t7\t3\t1101\t", np.char.zfill(np.arange(n).astype(str), 9)), "\t2000\tACGTAC\t%d\t" % mate)
        w = len(head[0])
        hb = np.frombuffer(head.astype("S%d" % w).tobytes(), dtype=np.uint8).reshape(n, w)
        tab = np.full((n, 1), 9, np.uint8)
        tail = np.tile(np.frombuffer(b"\t1\n", dtype=np.uint8), (n, 1))
        rec = np.concatenate([hb, seq, tab, qual, tail], axis=1)
        rec.tofile(path)
        total += rec.size
    return total


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    kind = sys.argv[2] if len(sys.argv) > 2 else "fastq"
    import pangea_plus_amd as pg
    pg.init(0)
    d = tempfile.mkdtemp(prefix="pgx_trim_")
    a, b = os.path.join(d, "a.txt"), os.path.join(d, "b.txt")
    if kind == "fastq":
        size = fastq_file(a, n)
        open(b, "wb").close()
        args = dict(b=None)
        argv = ["-a", "a.txt"]
    else:
        size = qseq_files(a, b, n)
        args = dict(b=b, g="100")
        argv = ["-a", "a.txt", "-b", "b.txt", "-g", "100"]
    print("%s: %d records, %.1f MB of text" % (kind, n, size / 1e6))
    for rep in range(3):
        t0 = time.time()
        out, fasta, mode = pg.trim2(a, **args)
        dt = time.time() - t0
        print("device call %d: %.3f s  -> %.2f M records/s, %.2f GB/s of input; FASTA %d bytes" % (rep, dt, n / dt / 1e6, size / dt / 1e9, len(fasta)))
    oracle = os.path.join(ROOT, "oracle", "bin", "pgx_oracle")
    t0 = time.time()
    subprocess.run([oracle, "trim2"] + argv, cwd=d, stdout=subprocess.DEVNULL, check=True)
    dt = time.time() - t0
    want = open(os.path.join(d, "output_files", "trim2", "a.txt_runblast.fasta"), "rb").read()
    print("oracle (1 core): %.3f s -> %.2f M records/s; identical: %s" % (dt, n / dt / 1e6, want == fasta))


if __name__ == "__main__":
    main()
