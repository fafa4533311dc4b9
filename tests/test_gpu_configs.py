"""GPU parity on the BASELINE.json configurations that fit one GPU (configs 1, 2 and 5; config 3 is test_gpu_fullsize.py,
config 4 is config 3 sharded: test_gpu_multirank.py + the driver's 8-GPU run)."""
import gzip
import hashlib
import json
import os
import shutil
import time

import numpy as np
import pytest

from conftest import GOLD, run_cmd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    import pangea_plus_amd as pg
    pg.init(0)
    return pg


# ------------------------------------------------------------------------------------------------ config 1
@pytest.fixture(scope="module")
def config1(tmp_path_factory):
    """tests/golden/config1 (oracle/gen_goldens_config1.py): the reference's 373 RDP 16S sequences against themselves; the
    tables after oracle blastn were made by the REFERENCE's tax_class, taxcollector Perl and Consensus Perl."""
    g = os.path.join(GOLD, "config1")
    d = tmp_path_factory.mktemp("config1")
    with gzip.open(os.path.join(g, "rdp373_gi.fa.gz"), "rb") as a, open(d / "rdp373_gi.fa", "wb") as b:
        shutil.copyfileobj(a, b)
    (d / "Tax_class").mkdir()
    for n in ("nodes.dmp", "names.dmp", "gi_taxid_nucl.dmp"):
        shutil.copy(os.path.join(g, n), d / "Tax_class" / n)
    return g, d, json.load(open(os.path.join(g, "meta.json")))


def test_config1_file_verbs_reproduce_the_reference_chain(pg, config1):
    """`makeblastdb`, `blastn`, `tax_class -c`, `taxcollector`, `consensus` as the README runs them (README.md:62-152), on the
    real lower-case / IUPAC 16S FASTA: every intermediate table has the recorded hash, the consensus text the recorded bytes."""
    g, d, meta = config1
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()  # noqa: E731
    pg.makeblastdb(str(d / "rdp373_gi.fa"), str(d / "db"))
    pg.blastn(os.path.join(g, "queries.fa"), str(d / "db"), str(d / "blast.tsv"))
    rows = open(d / "blast.tsv").readlines()
    assert "".join(rows[:200]) == open(os.path.join(g, "blast.head.tsv")).read()
    assert len(rows) == meta["blast_rows"] and sha(d / "blast.tsv") == meta["blast_sha256"]
    assert sum(1 for l in rows if l.split("\t")[5] != "0") == meta["gapped_rows"] > 3000   # real 16S: nearly every row is gapped
    rc, out, _ = pg.tax_class(["-c"], cwd=str(d / "Tax_class"))
    assert rc == 0
    report = pg.taxcollector(str(d / "blast.tsv"), str(d / "blast_class.txt"), taxdir=str(d / "Tax_class"))
    assert "".join(open(d / "blast_class.txt").readlines()[:200]) == open(os.path.join(g, "blast_class.head.txt")).read()
    assert sha(d / "blast_class.txt") == meta["blast_class_sha256"]
    assert hashlib.sha256(report).hexdigest() == meta["taxcollector_stdout_sha256"]
    log = pg.consensus(str(d / "blast_class.txt"), os.path.join(g, "rdp.txt"), str(d / "consensus.txt"))
    assert (d / "consensus.txt").read_bytes() == open(os.path.join(g, "consensus.txt"), "rb").read()
    assert log.replace(str(d / "consensus.txt").encode(), b"<OUT>") == open(os.path.join(g, "consensus.stdout"), "rb").read()


def test_config1_fused_path_gives_the_reference_consensus(pg, config1):
    from pangea_plus_amd import _capi
    g, d, meta = config1
    pg.TaxDb.create(str(d / "Tax_class"))
    tax = pg.TaxDb.open(str(d / "Tax_class"))
    db = pg.Db.from_fasta(str(d / "rdp373_gi.fa"))
    db.bind_taxonomy(tax)
    reads = pg.Reads.from_fasta(os.path.join(g, "queries.fa"))
    rdp = pg.Rdp.from_file(os.path.join(g, "rdp.txt"), reads, db)
    hits, recs = _capi.classify_consensus(db, reads, rdp)
    table = hits.format(db, reads)
    assert hashlib.sha256(table).hexdigest() == meta["blast_sha256"]
    assert _capi.consensus_format(db, reads, hits, recs) == open(os.path.join(g, "consensus.txt"), "rb").read()
    # 1 400-base queries: every initial HSP goes through the wide gapped kernels (all but a handful are listed for them)
    # (and none is counted twice: the figure is the first tier's own list entries)
    slots = hits.read_offsets(len(reads))[-1]
    assert 0.999 * slots <= _capi.stage_times().gapped_wide <= slots


# ------------------------------------------------------------------------------------------------ config 2
def test_config2_full_length_queries_against_50_megabases(pg, oracle_bin, tmp_path):
    """BASELINE config 2: 9 178 x 1 400-bp queries against the first 50 Mbp of the synthetic 16S-like database (33 334 x
    1 500 bp), seed + extend only: the -outfmt 6 table against the oracle, byte for byte; the rate goes to DESIGN.md."""
    from pangea_plus_amd import _capi
    shape = ["--n-seq", "33334", "--seq-len", "1500", "--n-genus", "1000", "--read-len", "1400"]
    db_fa, rd_fa, want = tmp_path / "db.fa", tmp_path / "q.fa", tmp_path / "want.tsv"
    assert run_cmd([oracle_bin, "synth", "db", "--out", str(db_fa)] + shape, timeout=600)[0] == 0
    assert run_cmd([oracle_bin, "synth", "reads", "--out", str(rd_fa), "--count", "9178"] + shape, timeout=600)[0] == 0
    t0 = time.time()
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd_fa), "-db", str(db_fa), "-outfmt", "6", "-out", str(want), "-num_threads",
                    str(min(os.cpu_count() or 1, 16))], timeout=3000)[0] == 0
    t_oracle = time.time() - t0
    cfg = pg.SynthCfg.default(n_seq=33334, seq_len=1500, n_genus=1000, read_len=1400)
    db = pg.Db.from_synth(cfg)
    reads = pg.Reads.from_synth(cfg, 0, 9178)
    hits = _capi.blast_search(db, reads)   # (first call: sizes the tables)
    t0 = time.time()
    hits = _capi.blast_search(db, reads)
    t_gpu = time.time() - t0
    st = _capi.stage_times()
    text = hits.format(db, reads)
    wantb = want.read_bytes()
    rows = wantb.count(b"\n")
    print("config 2: %d rows; oracle %.1f s; device search %.3f s (seed %.1f ms, gapped %.1f ms, order %.1f ms) = %.0f queries/s"
          % (rows, t_oracle, t_gpu, st.seed_extend_ms, st.gapped_ms, st.sort_ms, 9178 / t_gpu))
    assert rows > 200000
    assert text == wantb
    out = os.environ.get("PGX_CONFIG2_REPORT")
    if out:
        json.dump({"rows": rows, "oracle_s": t_oracle, "device_s": t_gpu, "seed_ms": st.seed_extend_ms, "gapped_ms": st.gapped_ms,
                   "sort_ms": st.sort_ms, "queries_per_s": 9178 / t_gpu}, open(out, "w"))


# ------------------------------------------------------------------------------------------------ config 5
def test_config5_third_stream_is_opened_and_ignored(pg, oracle_bin, tmp_path):
    """BASELINE config 5 on one GPU: BLAST + SOAP + RDP.  The SOAP classification of the same reads (this build's soap verb) is
    handed to the fused path as the third stream: it must be openable, and -- as in the reference, Consensus:40-46 -- the
    records do not depend on it."""
    from pangea_plus_amd import _capi
    shape = dict(n_seq=3000, seq_len=500, n_genus=60, read_len=150)
    args = ["--n-seq", "3000", "--seq-len", "500", "--n-genus", "60", "--read-len", "150"]
    n = 3000
    (tmp_path / "Tax_class").mkdir()
    for what, path, extra in (("db", "db.fa", []), ("reads", "reads.fa", ["--count", str(n)]), ("rdp", "rdp.tsv", ["--count", str(n)]),
                              ("taxdump", "Tax_class", [])):
        assert run_cmd([oracle_bin, "synth", what, "--out", str(tmp_path / path)] + extra + args)[0] == 0
    pg.TaxDb.create(str(tmp_path / "Tax_class"))
    pg.soap_index(str(tmp_path / "db.fa"))
    pg.soap(str(tmp_path / "reads.fa"), str(tmp_path / "db.fa.index"), str(tmp_path / "soap.txt"), M=4, r=2)
    assert os.path.getsize(tmp_path / "soap.txt") > 100000
    cfg = pg.SynthCfg.default(**shape)
    db = pg.Db.from_synth(cfg)
    db.bind_taxonomy(pg.TaxDb.open(str(tmp_path / "Tax_class")))
    reads = pg.Reads.from_synth(cfg, 0, n)
    rdp = pg.Rdp.from_synth(cfg, 0, n, db)
    h2, r2 = _capi.classify_consensus(db, reads, rdp)
    h3, r3 = _capi.classify_consensus(db, reads, rdp, soap=str(tmp_path / "soap.txt"))
    assert (r2 == r3).all() and (r2["hit"] >= 0).sum() > n * 0.9
    assert _capi.consensus_format(db, reads, h3, r3) == _capi.consensus_format(db, reads, h2, r2)
    for falsy in ("", "0"):   # Perl truth: the option counts as not given
        _, rf = _capi.classify_consensus(db, reads, rdp, soap=falsy)
        assert (rf == r2).all()
    with pytest.raises(_capi.PangeaError) as e:
        _capi.classify_consensus(db, reads, rdp, soap=str(tmp_path / "missing.txt"))
    assert e.value.status == -2 and "Unable to open" in str(e.value)
    # and the file verb with -s: same bytes with and without
    pg.makeblastdb(str(tmp_path / "db.fa"), str(tmp_path / "db"))
    pg.blastn(str(tmp_path / "reads.fa"), str(tmp_path / "db"), str(tmp_path / "hits.tsv"))
    pg.taxcollector(str(tmp_path / "hits.tsv"), str(tmp_path / "hits_class.tsv"), taxdir=str(tmp_path / "Tax_class"))
    pg.consensus(str(tmp_path / "hits_class.tsv"), str(tmp_path / "rdp.tsv"), str(tmp_path / "c2.txt"))
    pg.consensus(str(tmp_path / "hits_class.tsv"), str(tmp_path / "rdp.tsv"), str(tmp_path / "c3.txt"), s=str(tmp_path / "soap.txt"))
    assert (tmp_path / "c2.txt").read_bytes() == (tmp_path / "c3.txt").read_bytes() == _capi.consensus_format(db, reads, h2, r2)
    # and the checker's chain, with the third stream given to its consensus verb as well (not product against product):
    # oracle blastn -> oracle taxcollector -> oracle consensus -b -r -s, the restatement the reference Perl pins (goldens)
    assert run_cmd([oracle_bin, "blastn", "-query", str(tmp_path / "reads.fa"), "-db", str(tmp_path / "db.fa"), "-outfmt", "6", "-out",
                    str(tmp_path / "o_hits.tsv"), "-num_threads", "8"], timeout=600)[0] == 0
    assert run_cmd([oracle_bin, "taxcollector", "-f", str(tmp_path / "o_hits.tsv"), "-o", str(tmp_path / "o_class.tsv"), "-d",
                    str(tmp_path / "Tax_class")], timeout=600)[0] == 0
    assert run_cmd([oracle_bin, "consensus", "-b", str(tmp_path / "o_class.tsv"), "-r", str(tmp_path / "rdp.tsv"), "-s", str(tmp_path / "soap.txt"),
                    "-o", str(tmp_path / "o_c3.txt")], timeout=600)[0] == 0
    assert (tmp_path / "o_c3.txt").read_bytes() == (tmp_path / "c3.txt").read_bytes()


def test_config5_opt_in_three_way_vote(pg, oracle_bin, tmp_path):
    """The opt-in extension of SURVEY 8(f) row 4 (spec pgx-vote3 v1; NOT reference behaviour): the SOAP table takes part.
    Checker: `pgx_oracle vote3` over the taxcollector'd BLAST table, the RDP stream and the taxcollector'd SOAP table
    (the SOAP rows rewritten as a 12-column table so that taxcollector gives them lineages)."""
    from pangea_plus_amd import _capi
    shape = dict(n_seq=3000, seq_len=500, n_genus=60, read_len=150)
    args = ["--n-seq", "3000", "--seq-len", "500", "--n-genus", "60", "--read-len", "150"]
    n = 3000
    (tmp_path / "Tax_class").mkdir()
    for what, path, extra in (("db", "db.fa", []), ("reads", "reads.fa", ["--count", str(n)]), ("rdp", "rdp.tsv", ["--count", str(n)]),
                              ("taxdump", "Tax_class", [])):
        assert run_cmd([oracle_bin, "synth", what, "--out", str(tmp_path / path)] + extra + args)[0] == 0
    pg.TaxDb.create(str(tmp_path / "Tax_class"))
    pg.soap_index(str(tmp_path / "db.fa"))
    pg.soap(str(tmp_path / "reads.fa"), str(tmp_path / "db.fa.index"), str(tmp_path / "soap.txt"), M=4, r=2)
    # drop the SOAP rows of every 7th read: those reads vote with two streams only
    rows = [l for l in open(tmp_path / "soap.txt") if int(hashlib.md5(l.split("\t")[0].encode()).hexdigest(), 16) % 7]
    open(tmp_path / "soap.txt", "w").writelines(rows)
    with open(tmp_path / "soap12.tsv", "w") as f:
        for l in rows:
            c = l.rstrip("\n").split("\t")
            f.write("\t".join([c[0], c[7], "100.00", c[5], "0", "0", "1", c[5], c[8], c[8], "0.0", "100"]) + "\n")
    cfg = pg.SynthCfg.default(**shape)
    db = pg.Db.from_synth(cfg)
    db.bind_taxonomy(pg.TaxDb.open(str(tmp_path / "Tax_class")))
    reads = pg.Reads.from_synth(cfg, 0, n)
    rdp = pg.Rdp.from_synth(cfg, 0, n, db)
    hits, _ = _capi.classify_consensus(db, reads, rdp)
    recs, text = _capi.vote3(db, reads, hits, rdp, str(tmp_path / "soap.txt"))
    # the checker's chain, all on the CPU
    assert run_cmd([oracle_bin, "blastn", "-query", str(tmp_path / "reads.fa"), "-db", str(tmp_path / "db.fa"), "-outfmt", "6", "-out",
                    str(tmp_path / "o_hits.tsv"), "-num_threads", "8"], timeout=600)[0] == 0
    for a, b in (("o_hits.tsv", "o_hits_class.tsv"), ("soap12.tsv", "o_soap_class.tsv")):
        assert run_cmd([oracle_bin, "taxcollector", "-f", str(tmp_path / a), "-o", str(tmp_path / b), "-d", str(tmp_path / "Tax_class")])[0] == 0
    assert run_cmd([oracle_bin, "vote3", str(tmp_path / "o_hits_class.tsv"), str(tmp_path / "rdp.tsv"), str(tmp_path / "o_soap_class.tsv"),
                    str(tmp_path / "o_vote.txt")])[0] == 0
    want = (tmp_path / "o_vote.txt").read_bytes()
    assert text == want
    d = recs["depth"]
    assert (d >= 0).all() and (d >= 5).mean() > 0.5 and (recs["votes"] == 3).any() and (recs["votes"] == 2).any()
    with pytest.raises(_capi.PangeaError) as e:
        _capi.vote3(db, reads, hits, rdp, str(tmp_path / "missing.txt"))
    assert e.value.status == -2
