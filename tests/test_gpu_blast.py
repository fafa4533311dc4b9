"""GPU parity, BLAST verb: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, byte for byte on the -outfmt 6 text."""
import os

import numpy as np
import pytest

from conftest import run_cmd

pytestmark = pytest.mark.gpu

SHAPE = dict(n_seq=2000, seq_len=600, n_genus=50, read_len=150)
SHAPE_ARGS = ["--n-seq", "2000", "--seq-len", "600", "--n-genus", "50", "--read-len", "150"]
N_READS = 3000


@pytest.fixture(scope="module")
def pg():
    import pangea_plus_amd as pg
    pg.init(0)
    return pg


@pytest.fixture(scope="module")
def workload(tmp_path_factory, oracle_bin):
    d = tmp_path_factory.mktemp("blast")
    db, rd, out = d / "db.fa", d / "reads.fa", d / "oracle.tsv"
    assert run_cmd([oracle_bin, "synth", "db", "--out", str(db)] + SHAPE_ARGS)[0] == 0
    assert run_cmd([oracle_bin, "synth", "reads", "--out", str(rd), "--count", str(N_READS)] + SHAPE_ARGS)[0] == 0
    rc, _, se = run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(out),
                         "-num_threads", "8"], timeout=600)
    assert rc == 0, se
    return d


def test_synthetic_generators_agree(pg, workload):
    from pangea_plus_amd import _capi
    cfg = pg.SynthCfg.default(**SHAPE)
    reads = pg.Reads.from_synth(cfg, 0, N_READS)
    want = [l.strip() for l in open(workload / "reads.fa") if not l.startswith(">")]
    for i in (0, 1, 2, 17, 1234, N_READS - 1):
        got = "".join("ACGTN"[b] for b in reads.get(i))
        assert got == want[i], i
    # a batch that starts in the middle of the stream
    tail = pg.Reads.from_synth(cfg, 1000, 10)
    assert "".join("ACGTN"[b] for b in tail.get(3)) == want[1003]


def test_blast_synth_device_path_matches_oracle_bytes(pg, workload):
    from pangea_plus_amd import _capi
    cfg = pg.SynthCfg.default(**SHAPE)
    db = pg.Db.from_synth(cfg)
    reads = pg.Reads.from_synth(cfg, 0, N_READS)
    hits = _capi.blast_search(db, reads)
    text = hits.format(db, reads)
    want = open(workload / "oracle.tsv", "rb").read()
    assert len(want) > 100000
    assert text == want
    t = _capi.stage_times()
    assert t.hits == len(hits) and t.probes == N_READS * 2 * 11


def test_blast_fasta_path_matches_oracle_bytes(pg, workload, tmp_path):
    out = tmp_path / "hits.tsv"
    pg.makeblastdb(str(workload / "db.fa"), str(tmp_path / "db"))
    pg.blastn(str(workload / "reads.fa"), str(tmp_path / "db"), str(out))
    assert out.read_bytes() == open(workload / "oracle.tsv", "rb").read()
    # read sharding: concatenating the blocks of 3 ranks gives the same file
    parts = b""
    for rk in range(3):
        p = tmp_path / ("part%d.tsv" % rk)
        pg.blastn(str(workload / "reads.fa"), str(tmp_path / "db"), str(p), rank=rk, world_size=3)
        parts += p.read_bytes()
    assert parts == out.read_bytes()
