#!/usr/bin/env python3
"""End-to-end timing of the PANGEA+ chain on one GPU, raw reads to abundance table (README.md:34-185):
QSEQ mate files -> trim2 -g 100 -> classify vs the 1 Gbp database -> lineage -> consensus with an RDP stream ->
consensus text -> megaclust2 table.  Mates are consecutive reads of the synthetic stream (so both mates of a joined
read hit their own subjects), the RDP lines are the synthetic ones of the first mate.
Usage: python3 tools/probe_pipeline.py [pairs]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pangea_plus_amd as pg  # noqa: E402
from pangea_plus_amd import _capi  # noqa: E402

ORACLE = os.path.join(ROOT, "oracle", "bin", "pgx_oracle")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    pg.init(0)
    cfg = pg.SynthCfg.default()
    d = tempfile.mkdtemp(prefix="pgx_pipe_")
    t0 = time.time()
    _capi._check(pg.lib().pgx_synth_write_taxdump(_capi.C.byref(cfg), d.encode()))
    pg.TaxDb.create(d)
    db = pg.Db.from_synth(cfg)
    db.bind_taxonomy(pg.TaxDb.open(d))
    print("set-up (taxonomy .bin, 1 Gbp database + index): %.1f s" % (time.time() - t0), flush=True)

    # raw input files (untimed): QSEQ lines of the two mates, and the RDP stream of the first mates under the joined names
    fa, rdp = os.path.join(d, "reads.fa"), os.path.join(d, "rdp_all.tsv")
    subprocess.check_call([ORACLE, "synth", "reads", "--out", fa, "--count", str(2 * n)])
    subprocess.check_call([ORACLE, "synth", "rdp", "--out", rdp, "--count", str(2 * n)])
    seqs = np.frombuffer(open(fa, "rb").read(), dtype=np.uint8)
    lines = open(fa, "rb").read().split(b"\n")[1::2]
    L = len(lines[0])
    seq = np.frombuffer(b"".join(lines[:2 * n]), dtype=np.uint8).reshape(2 * n, L)
    del seqs, lines
    rng = np.random.default_rng(3)
    qual = rng.integers(28, 41, size=(2 * n, L), dtype=np.int16)
    cut = rng.integers(int(L * 0.8), L + 1, size=2 * n)
    qual[np.arange(L)[None, :] >= cut[:, None]] = 4
    qual = (qual + 64).astype(np.uint8)
    idx = np.char.zfill(np.arange(n).astype(str), 9)
    for mate, path in ((1, "a.txt"), (2, "b.txt")):
        head = np.char.add(np.char.add("p\t", idx), "\t1\t1\t1\t1\t0\t%d\t" % mate)
        w = len(head[0])
        hb = np.frombuffer(head.astype("S%d" % w).tobytes(), dtype=np.uint8).reshape(n, w)
        tab = np.full((n, 1), 9, np.uint8)
        tail = np.tile(np.frombuffer(b"\t1\n", dtype=np.uint8), (n, 1))
        np.concatenate([hb, seq[mate - 1::2], tab, qual[mate - 1::2], tail], axis=1).tofile(os.path.join(d, path))
    with open(rdp, "rb") as f, open(os.path.join(d, "rdp.tsv"), "wb") as g:
        for i, line in enumerate(f):
            if i % 2 == 0 and i < 2 * n:
                g.write(b"p:%09d:1:1:1:1:0:1:AB" % (i // 2) + line[line.index(b"\t"):])
    size = os.path.getsize(os.path.join(d, "a.txt")) + os.path.getsize(os.path.join(d, "b.txt"))
    print("input: %d pairs, %.0f MB of QSEQ text" % (n, size / 1e6), flush=True)

    for rep in range(2):
        t = [time.time()]
        messages, fasta, mode = pg.trim2(os.path.join(d, "a.txt"), b=os.path.join(d, "b.txt"), g=100)
        t.append(time.time())
        reads = pg.Reads.from_fasta_text(fasta)
        t.append(time.time())
        rdps = pg.Rdp.from_file(os.path.join(d, "rdp.tsv"), reads, db)
        t.append(time.time())
        hits, recs = _capi.classify_consensus(db, reads, rdps)
        st = _capi.stage_times()
        t.append(time.time())
        text = _capi.consensus_format(db, reads, hits, recs)
        t.append(time.time())
        csv, log = pg.megaclust_batch(db, reads, hits, recs, s="80", b="100", e="1e-20")
        t.append(time.time())
        names = ("trim2 (files -> FASTA text)", "reads batch from the text", "RDP stream from its file", "classify + consensus",
                 "consensus text", "megaclust2 table")
        print("run %d: %d joined reads kept, %d hits (kernels: seed %.1f ms, group %.1f ms, order+consensus %.1f ms), %d MB consensus text, %d table lines"
              % (rep, len(reads), len(hits), st.seed_extend_ms, st.group_ms, st.sort_ms, len(text) >> 20, csv.count(b"\n")))
        for k, name in enumerate(names):
            print("    %-30s %7.3f s" % (name, t[k + 1] - t[k]))
        print("    %-30s %7.3f s  = %.2f M pairs/s raw reads to table" % ("total", t[-1] - t[0], n / (t[-1] - t[0]) / 1e6), flush=True)
        del hits, reads, rdps


if __name__ == "__main__":
    main()
