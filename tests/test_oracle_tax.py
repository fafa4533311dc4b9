"""Oracle (CPU restatement of Tax_class/ncbitc.c and NCBI-taxcollector-0.01.pl) against the
golden vectors that oracle/gen_goldens.py produced with the reference's own C and Perl."""
import glob
import hashlib
import json
import os
import shutil

import pytest

from conftest import run_cmd


@pytest.fixture(scope="module")
def taxdir(tmp_path_factory, oracle_bin, gold):
    d = tmp_path_factory.mktemp("otax") / "Tax_class"
    d.mkdir()
    for n in ("nodes.dmp", "names.dmp", "gi_taxid_nucl.dmp"):
        shutil.copy(os.path.join(gold, "tax", n), d / n)
    rc, _, _ = run_cmd([oracle_bin, "tax_class", "-c"], cwd=d)
    assert rc == 0
    return d


@pytest.fixture(scope="module")
def golden(gold):
    return json.load(open(os.path.join(gold, "tax", "tax_class_golden.json")))


def test_create_binaries_byte_exact(taxdir, golden):
    for n, h in golden["sha256"].items():
        assert hashlib.sha256((taxdir / n).read_bytes()).hexdigest() == h, n
    for n, s in golden["sizes"].items():
        assert (taxdir / n).stat().st_size == s, n


def test_names_records_fieldwise(taxdir, golden):
    nb = (taxdir / "names.dmp.bin").read_bytes()
    cnt = int.from_bytes(nb[:4], "little")
    assert cnt == len(golden["names_records"])
    for i, want in enumerate(golden["names_records"]):
        r = nb[4 + 196 * i: 4 + 196 * (i + 1)]
        got = [int.from_bytes(r[:4], "little", signed=True)] + \
              [r[4 + 64 * k: 68 + 64 * k].split(b"\0")[0].decode("latin-1") for k in range(3)]
        assert got == want, i


def test_cli_stdout_and_status(taxdir, golden, oracle_bin):
    for case in golden["cli"]:
        if case["args"] == ["-c"]:
            continue
        rc, so, se = run_cmd([oracle_bin, "tax_class"] + case["args"], cwd=taxdir)
        assert so.decode("latin-1") == case["stdout"], case["args"]
        assert rc == case["rc"], case["args"]


def test_taxcollector_goldens(taxdir, gold, oracle_bin, tmp_path):
    cases = sorted(glob.glob(os.path.join(gold, "taxcollect", "*.in.tsv")))
    assert len(cases) >= 5
    for inp in cases:
        name = os.path.basename(inp)[:-len(".in.tsv")]
        out = tmp_path / (name + ".out.tsv")
        rc, so, se = run_cmd([oracle_bin, "taxcollector", "-f", inp, "-o", str(out), "-d", str(taxdir)])
        assert rc == 0, name
        assert out.read_bytes() == open(os.path.join(gold, "taxcollect", name + ".out.tsv"), "rb").read(), name
        assert so == open(os.path.join(gold, "taxcollect", name + ".report.txt"), "rb").read(), name


def test_taxcollector_reference_hang_inputs_are_flagged(taxdir, oracle_bin, tmp_path):
    # gi 40 -> leaf is a child of the root; gi 41 -> taxid in a nodes.dmp gap; no '|' at all.
    # The reference never terminates on these; the restatement reports them.
    for line in ("q\tgi|40|gb|A|\t1\n", "q\tgi|41|gb|A|\t1\n", "q\tS000860299\t1\n"):
        inp = tmp_path / "h.tsv"
        inp.write_text(line)
        rc, _, _ = run_cmd([oracle_bin, "taxcollector", "-f", str(inp), "-o", str(tmp_path / "o"), "-d", str(taxdir)])
        assert rc == 3
