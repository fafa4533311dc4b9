"""GPU parity, tax-annotate and consensus verbs: HIP path (C ABI / CLIs) against the golden vectors made
by the reference's own C and Perl, and against the oracle on larger seeded inputs."""
import glob
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

from conftest import run_cmd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "pangea-plus_amd", "bin")


@pytest.fixture(scope="module")
def pg():
    import pangea_plus_amd as pg
    pg.init(0)
    return pg


@pytest.fixture(scope="module")
def taxdir(tmp_path_factory, pg, gold):
    d = tmp_path_factory.mktemp("gtax") / "Tax_class"
    d.mkdir()
    for n in ("nodes.dmp", "names.dmp", "gi_taxid_nucl.dmp"):
        shutil.copy(os.path.join(gold, "tax", n), d / n)
    pg.TaxDb.create(str(d))
    return d


def test_taxcollector_matches_reference_goldens(pg, taxdir, gold, tmp_path):
    cases = sorted(glob.glob(os.path.join(gold, "taxcollect", "*.in.tsv")))
    assert len(cases) >= 5
    for inp in cases:
        name = os.path.basename(inp)[:-len(".in.tsv")]
        out = tmp_path / (name + ".out.tsv")
        report = pg.taxcollector(inp, str(out), taxdir=str(taxdir))
        assert out.read_bytes() == open(os.path.join(gold, "taxcollect", name + ".out.tsv"), "rb").read(), name
        assert report == open(os.path.join(gold, "taxcollect", name + ".report.txt"), "rb").read(), name
    # the executable, run the way the reference is (cwd = parent of Tax_class/)
    inp = os.path.join(gold, "taxcollect", "basic.in.tsv")
    p = subprocess.run([os.path.join(BIN, "taxcollector"), "-f", inp, "-o", str(tmp_path / "cli.tsv")],
                       cwd=taxdir.parent, stdout=subprocess.PIPE)
    assert p.returncode == 0
    assert (tmp_path / "cli.tsv").read_bytes() == open(os.path.join(gold, "taxcollect", "basic.out.tsv"), "rb").read()
    assert p.stdout == open(os.path.join(gold, "taxcollect", "basic.report.txt"), "rb").read()


def test_taxcollector_defines_the_reference_hang_inputs(pg, taxdir, tmp_path):
    for line in ("q\tgi|40|gb|A|\t1\n", "q\tgi|41|gb|A|\t1\n", "q\tS000860299\t1\n"):
        inp = tmp_path / "h.tsv"
        inp.write_text("q0\tgi|5|gb|A|\t99.0\t1\n" + line)
        with pytest.raises(pg.PangeaError) as e:
            pg.taxcollector(str(inp), str(tmp_path / "o.tsv"), taxdir=str(taxdir))
        assert e.value.status == -6
        # the lines before the offending one are still written, as the oracle does
        assert (tmp_path / "o.tsv").read_text().startswith("q0\t[0]Bacteria;")


def test_lineage_batch_walk(pg, taxdir):
    with pg.TaxDb.open(str(taxdir)) as db:
        lin, cnt, st = db.lineage_batch([5, 7, 9, 20, 40, 41, 21, 0, 12])
        assert list(st) == [0, 0, 0, 1, 2, 2, 0, 2, 0]
        assert list(lin[0][:cnt[0]]) == [25, 24, 23, 22, 21, 20, 2]          # species .. superkingdom
        assert list(lin[1][:cnt[1]]) == [35, 34, 33, 32, 30]                  # kingdom kept, no-rank skipped
        assert list(lin[2][:cnt[2]]) == [61, 2]
        assert list(lin[6][:cnt[6]]) == [52, -2]                              # "[0]Unclassified;"
        assert db.gi2taxid(7) == 35


def test_taxcollector_random_walks_match_oracle(pg, oracle_bin, tmp_path):
    shape = ["--n-seq", "3000", "--seq-len", "200", "--n-genus", "150"]
    d = tmp_path / "Tax_class"
    d.mkdir()
    cfg = pg.SynthCfg.default(n_seq=3000, seq_len=200, n_genus=150)
    from pangea_plus_amd import _capi
    _capi._check(pg.lib().pgx_synth_write_taxdump(__import__("ctypes").byref(cfg), str(d).encode()))
    # the product's dump writer and the oracle's agree byte for byte
    d2 = tmp_path / "o" / "Tax_class"
    d2.mkdir(parents=True)
    assert run_cmd([oracle_bin, "synth", "taxdump", "--out", str(d2)] + shape)[0] == 0
    for n in ("nodes.dmp", "names.dmp", "gi_taxid_nucl.dmp"):
        assert (d / n).read_bytes() == (d2 / n).read_bytes(), n
    pg.TaxDb.create(str(d))
    rng = random.Random(5)
    lines = []
    for i in range(4000):
        gi = rng.choice([rng.randrange(1000, 4000), rng.randrange(1, 999), rng.randrange(4000, 5000)])
        lines.append("q%d\tgi|%d|syn|S|\t%.2f\t150\t1\t0\t1\t150\t1\t150\t1e-50\t %d\n" % (i, gi, rng.uniform(80, 100), rng.randrange(100, 300)))
    inp = tmp_path / "in.tsv"
    inp.write_text("".join(lines))
    rep = pg.taxcollector(str(inp), str(tmp_path / "p.tsv"), taxdir=str(d))
    rc, so, _ = run_cmd([oracle_bin, "taxcollector", "-f", str(inp), "-o", str(tmp_path / "o.tsv"), "-d", str(d)])
    assert rc == 0
    assert (tmp_path / "p.tsv").read_bytes() == (tmp_path / "o.tsv").read_bytes()
    assert rep == so


def test_consensus_matches_reference_goldens(pg, gold, tmp_path):
    cases = sorted(glob.glob(os.path.join(gold, "consensus", "*.blast.tsv")))
    assert len(cases) >= 13
    for b in cases:
        name = os.path.basename(b)[:-len(".blast.tsv")]
        out = tmp_path / (name + ".out.txt")
        log = pg.consensus(b, os.path.join(gold, "consensus", name + ".rdp.tsv"), str(out))
        assert out.read_bytes() == open(os.path.join(gold, "consensus", name + ".out.txt"), "rb").read(), name
        assert log.replace(str(out).encode(), b"@OUT@") == open(os.path.join(gold, "consensus", name + ".log.txt"), "rb").read(), name
    # -s is opened and ignored; the executable prints the same bytes
    b = os.path.join(gold, "consensus", "basic.blast.tsv")
    r = os.path.join(gold, "consensus", "basic.rdp.tsv")
    s = tmp_path / "soap.txt"
    s.write_text("whatever\n")
    p = subprocess.run([os.path.join(BIN, "consensus"), "-b", b, "-r", r, "-s", str(s), "-o", str(tmp_path / "c.txt")],
                       stdout=subprocess.PIPE)
    assert p.returncode == 0
    assert (tmp_path / "c.txt").read_bytes() == open(os.path.join(gold, "consensus", "basic.out.txt"), "rb").read()


def test_consensus_hang_input_is_defined(pg, tmp_path):
    b, r = tmp_path / "b.tsv", tmp_path / "r.tsv"
    b.write_text("q1\t[0]Bacteria;\t99.0\t1\n")
    r.write_text("q1\t\t\t\t\tBacteria\tdomain\t1.0\nq2\t\t\t\t\tBacteria\tdomain\t1.0\n")
    with pytest.raises(pg.PangeaError) as e:
        pg.consensus(str(b), str(r), str(tmp_path / "o"))
    assert e.value.status == -6


def test_consensus_random_tables_match_oracle(pg, oracle_bin, tmp_path):
    rng = random.Random(11)
    names = ["Bacteria", "Firmicutes", "Bacilli", "Bacillales", "Bacillaceae", "Bacillus", "Bacillus_subtilis", "Bar9 division",
             "Eukaryota", "Metazoa", "x", "Candidatus_Foo6", "Unclassified", ""]
    ranks = ["0", "1", "2", "3", "4", "5", "6", "9", "7", "x", ""]
    rdpranks = ["domain", "phylum", "class", "order", "family", "genus", "species", "kingdom", "", "rootrank"]
    bl, rd = [], []
    for q in range(1500):
        nh = rng.choice([1, 1, 2, 3, 5, 9, 14])
        for _ in range(nh):
            lin = "".join("[%s]%s;" % (rng.choice(ranks), rng.choice(names)) for _ in range(rng.randrange(0, 9)))
            if rng.random() < 0.05:
                lin = "Unidentified(GI:%d);" % rng.randrange(1, 99)
            sim = rng.choice(["%.2f" % rng.uniform(5, 100), "100.00", "9.5", "", "99"])
            sep = rng.choice(["\t", "\t", "\t\t"])
            bl.append("q%d\t%s%s%s\t150\t1\t0\t1\t150\t1\t150\t1e-9\t99\n" % (q, lin, sep, sim))
        if rng.random() < 0.03:
            continue  # a BLAST-only read: skipped with a stdout note
        trip = []
        for _ in range(rng.randrange(0, 8)):
            n = rng.choice(names)
            if rng.random() < 0.3:
                n = '"%s %d"' % (n, rng.randrange(9))
            trip += [n, rng.choice(rdpranks), "0.%d" % rng.randrange(10)]
        if rng.random() < 0.1 and trip:
            trip = trip[:-1]
        rd.append("q%d\t\t\t\t\t%s\n" % (q, "\t".join(trip)))
    (tmp_path / "b.tsv").write_text("".join(bl))
    (tmp_path / "r.tsv").write_text("".join(rd))
    log = pg.consensus(str(tmp_path / "b.tsv"), str(tmp_path / "r.tsv"), str(tmp_path / "p.txt"))
    rc, so, _ = run_cmd([oracle_bin, "consensus", "-b", str(tmp_path / "b.tsv"), "-r", str(tmp_path / "r.tsv"), "-o",
                         str(tmp_path / "o.txt")])
    assert rc == 0
    assert (tmp_path / "p.txt").read_bytes() == (tmp_path / "o.txt").read_bytes()
    assert log.replace(str(tmp_path / "p.txt").encode(), b"@") == so.replace(str(tmp_path / "o.txt").encode(), b"@")


@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("PGX_TAX_SEEDS", "1,2,3,4,5,6").split(",")])
def test_taxcollector_random_taxonomies_match_oracle(pg, oracle_bin, tmp_path, seed):
    """Random irregular trees (any mix of ranks, several names per node, gi numbers without a node): the same generator
    the oracle is cross-checked with against the reference's C + Perl (oracle/sweep_taxcollect_vs_reference.py).  Where
    the reference never terminates both report it and have written the same lines before."""
    from tax_inputs import hits, taxonomy, write_dumps
    rng = random.Random(seed)
    nodes, names, gis = taxonomy(rng)
    d = tmp_path / "Tax_class"
    d.mkdir()
    write_dumps(str(d), nodes, names, gis)
    pg.TaxDb.create(str(d))
    (tmp_path / "in.tsv").write_text(hits(rng, gis))
    rc, so, _ = run_cmd([oracle_bin, "taxcollector", "-f", str(tmp_path / "in.tsv"), "-o", str(tmp_path / "o.tsv"), "-d", str(d)])
    if rc == 0:
        rep = pg.taxcollector(str(tmp_path / "in.tsv"), str(tmp_path / "p.tsv"), taxdir=str(d))
        assert rep == so
    else:
        with pytest.raises(pg.PangeaError) as e:
            pg.taxcollector(str(tmp_path / "in.tsv"), str(tmp_path / "p.tsv"), taxdir=str(d))
        assert e.value.status == -6
    assert (tmp_path / "p.tsv").read_bytes() == (tmp_path / "o.tsv").read_bytes()


@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("PGX_TAX_SEEDS", "1,2,3,4,5,6").split(",")])
def test_tax_class_cli_on_random_taxonomies_matches_oracle(oracle_bin, tmp_path, seed):
    """`tax_class -c` and every `-s/-g/-t/-n` lookup on a random taxonomy: the oracle is cross-checked against the
    reference's own C on the same generator (oracle/sweep_tax_class_vs_reference.py), incl. the name searches that
    probe position 0 of names.dmp.bin (the reference closes its file there and finds nothing afterwards)."""
    from tax_inputs import taxonomy, write_dumps
    rng = random.Random(seed)
    nodes, names, gis = taxonomy(rng)
    dirs = []
    for tool in ("product", "oracle"):
        d = tmp_path / tool
        d.mkdir()
        write_dumps(str(d), nodes, names, gis)
        dirs.append(d)
    cmds = ([os.path.join(BIN, "tax_class")], [oracle_bin, "tax_class"])
    assert run_cmd(cmds[0] + ["-c"], cwd=dirs[0])[0] == 0 and run_cmd(cmds[1] + ["-c"], cwd=dirs[1])[0] == 0
    for n in ("gi_taxid_nucl.dmp.bin", "nodes.dmp.bin", "names.dmp.bin"):
        assert (dirs[0] / n).read_bytes() == (dirs[1] / n).read_bytes(), n
    queries = [["-s", str(g)] for g, _ in gis] + [["-g", str(g)] for g, _ in gis[:8]] + [["-s", "0"], ["-g", "0"], ["-s", str(gis[-1][0] + 5)]]
    queries += [["-t", str(t)] for t, _, _, _ in nodes] + [["-n", str(t)] for t, _, _, _ in nodes] + [["-t", "9999"], ["-n", "9999"], ["-n", "5"]]
    for q in queries:
        a, b = run_cmd(cmds[0] + q, cwd=dirs[0]), run_cmd(cmds[1] + q, cwd=dirs[1])
        assert (a[0], a[1], bool(a[2])) == (b[0], b[1], bool(b[2])), q
