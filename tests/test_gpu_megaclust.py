"""GPU parity of the two post-consensus steps: the product's `megaclust2` and
`megaclustable` command lines against the goldens printed by the reference's own Perl, and the fused
pgx_megaclust_batch against the oracle run on the consensus text of the same batch."""
import os

import pytest

from conftest import ROOT, run_cmd
from test_oracle_megaclust import (megaclust_cases, megaclustable_cases, run_megaclust_case, run_megaclustable_case,
                                   table_key)

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "pangea-plus_amd", "bin")
SHAPE = dict(n_seq=3000, seq_len=500, n_genus=60, read_len=150)
ARGS = ["--n-seq", "3000", "--seq-len", "500", "--n-genus", "60", "--read-len", "150"]
N = 4000


@pytest.mark.parametrize("name,info", megaclust_cases())
def test_megaclust2_cli_matches_reference(name, info, tmp_path):
    run_megaclust_case([os.path.join(BIN, "megaclust2")], name, info, tmp_path)


@pytest.mark.parametrize("name,info", megaclustable_cases())
def test_megaclustable_cli_matches_reference(name, info, tmp_path):
    run_megaclustable_case([os.path.join(BIN, "megaclustable")], name, info, tmp_path)


@pytest.fixture(scope="module")
def batch(tmp_path_factory, oracle_bin):
    import pangea_plus_amd as pg
    from pangea_plus_amd import _capi
    pg.init(0)
    d = tmp_path_factory.mktemp("mc")
    (d / "Tax_class").mkdir()
    assert run_cmd([oracle_bin, "synth", "taxdump", "--out", str(d / "Tax_class")] + ARGS)[0] == 0
    assert run_cmd([oracle_bin, "tax_class", "-c"], cwd=d / "Tax_class")[0] == 0
    cfg = pg.SynthCfg.default(**SHAPE)
    db = pg.Db.from_synth(cfg)
    db.bind_taxonomy(pg.TaxDb.open(str(d / "Tax_class")))
    reads = pg.Reads.from_synth(cfg, 0, N)
    rdp = pg.Rdp.from_synth(cfg, 0, N, db)
    hits, recs = _capi.classify_consensus(db, reads, rdp)
    (d / "consensus.txt").write_bytes(_capi.consensus_format(db, reads, hits, recs))
    return pg, _capi, d, db, reads, hits, recs


@pytest.mark.parametrize("opts", [
    {}, {"s": "80", "b": "100", "e": "1e-20"}, {"s": "99.33"}, {"s": "99.34"}, {"b": "250"}, {"b": "277"}, {"b": "278"},
    {"e": "1e-60"}, {"e": "3e-73"}, {"e": "2e-73"}, {"c": "1"}, {"d": ";", "s": "97.5"}, {"s": "0.0", "b": "0.0"},
])
def test_fused_megaclust_equals_oracle_on_the_consensus_text(batch, opts, oracle_bin, tmp_path):
    pg, _capi, d, db, reads, hits, recs = batch
    argv = []
    for k, v in opts.items():
        argv += ["-" + k, v]
    rc, out, err = run_cmd([oracle_bin, "megaclust2", "-i", str(d / "consensus.txt"), "-o", str(tmp_path / "want.csv")] + argv)
    assert rc == 0
    csv, log = pg.megaclust_batch(db, reads, hits, recs, **opts)
    assert log == out
    want = (tmp_path / "want.csv").read_bytes()
    assert table_key(csv) == table_key(want)
    if not opts:
        assert len(want.split(b"\n")) > 500     # a real table, not an empty one
    # and the file verb on the same text
    assert pg.megaclust2(str(d / "consensus.txt"), str(tmp_path / "got.csv"), **opts) == out
    assert table_key((tmp_path / "got.csv").read_bytes()) == table_key(want)


def test_tables_pivot_like_the_reference_script(batch, oracle_bin, tmp_path):
    """reads -> abundance table: two threshold levels of the same batch pivoted at every rank, product vs oracle
    (the oracle's megaclustable is pinned byte for byte by the reference goldens)."""
    pg, _capi, d, db, reads, hits, recs = batch
    for name, o in (("a.csv", {"s": "80", "b": "100"}), ("b.csv", {"s": "99", "b": "250"})):
        csv, _ = pg.megaclust_batch(db, reads, hits, recs, **o)
        (tmp_path / name).write_bytes(csv)
    for level in range(7):
        argv = ["-m", "a.csv", "b.csv", "-t", str(level), "-o", "want.txt"]
        assert run_cmd([oracle_bin, "megaclustable"] + argv, cwd=tmp_path)[0] == 0
        got_argv = ["-m", str(tmp_path / "a.csv"), str(tmp_path / "b.csv"), "-t", str(level), "-o", str(tmp_path / "got.txt")]
        assert pg.megaclustable(got_argv) == b""
        assert (tmp_path / "got.txt").read_bytes() == (tmp_path / "want.txt").read_bytes()
    assert len((tmp_path / "want.txt").read_bytes()) > 1000
