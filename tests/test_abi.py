"""No-GPU checks of the product: the C-ABI library loads, exports every symbol the header declares,
refuses to compute without a device (no CPU fallback), and the host-only verbs (tax_class single
lookups, -c) reproduce the reference's bytes."""
import hashlib
import json
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pg():
    import pangea_plus_amd as pg
    if not os.path.exists(pg.lib_path):
        import importlib.util
        spec = importlib.util.spec_from_file_location("pgx_build", os.path.join(ROOT, "pangea-plus_amd", "build.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        b.build_all()
    return pg


def header_symbols():
    text = open(os.path.join(ROOT, "include", "pangea_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pg):
    from pangea_plus_amd import _capi
    declared = header_symbols()
    assert len(declared) >= 50
    lib = pg.lib()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert missing == []
    assert sorted(_capi.SYMBOLS) == declared


def test_product_never_links_the_oracle(pg):
    out = subprocess.run(["ldd", pg.lib_path], stdout=subprocess.PIPE, text=True).stdout
    assert "liboracle" not in out and "amdhip64" in out
    for root, _, files in os.walk(os.path.join(ROOT, "pangea-plus_amd")):
        for f in files:
            if f.endswith((".hip", ".hpp", ".cpp", ".py")):
                src = open(os.path.join(root, f), errors="replace").read()
                assert "o_common.h" not in src and "liboracle" not in src and "pgx_oracle" not in src, f


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is visible")
def test_compute_fails_loudly_without_a_device(pg, tmp_path):
    with pytest.raises(pg.PangeaError) as e:
        pg.init(0)
    assert e.value.status == -3
    fa = tmp_path / "x.fa"
    fa.write_text(">gi|1|x|y|\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    with pytest.raises(pg.PangeaError) as e:
        pg.Db.from_fasta(str(fa))
    assert e.value.status == -3 and "no CPU path" in str(e.value)
    with pytest.raises(pg.PangeaError):
        pg.Reads.from_fasta(str(fa))
    with pytest.raises(pg.PangeaError):
        pg.consensus("a", "b", "c")


@pytest.fixture(scope="module")
def taxdir(tmp_path_factory, pg, gold):
    d = tmp_path_factory.mktemp("ptax") / "Tax_class"
    d.mkdir()
    for n in ("nodes.dmp", "names.dmp", "gi_taxid_nucl.dmp"):
        shutil.copy(os.path.join(gold, "tax", n), d / n)
    rc, out, err = pg.tax_class(["-c"], cwd=str(d))
    assert rc == 0
    return d


def test_tax_create_is_byte_compatible_with_the_reference(taxdir, gold):
    golden = json.load(open(os.path.join(gold, "tax", "tax_class_golden.json")))
    for n, h in golden["sha256"].items():
        assert hashlib.sha256((taxdir / n).read_bytes()).hexdigest() == h, n
    nb = (taxdir / "names.dmp.bin").read_bytes()
    assert int.from_bytes(nb[:4], "little") == len(golden["names_records"])
    for i, want in enumerate(golden["names_records"]):
        r = nb[4 + 196 * i: 4 + 196 * (i + 1)]
        got = [int.from_bytes(r[:4], "little", signed=True)] + \
              [r[4 + 64 * k: 68 + 64 * k].split(b"\0")[0].decode("latin-1") for k in range(3)]
        assert got == want, i


def test_tax_class_cli_bytes_and_status(pg, taxdir, gold):
    golden = json.load(open(os.path.join(gold, "tax", "tax_class_golden.json")))
    exe = os.path.join(ROOT, "pangea-plus_amd", "bin", "tax_class")
    for case in golden["cli"]:
        rc, out, err = pg.tax_class(case["args"], cwd=str(taxdir))
        assert out.decode("latin-1") == case["stdout"], case["args"]
        assert rc == case["rc"], case["args"]
    # and through the real executable, as the Perl driver would call it
    for case in golden["cli"][::5]:
        p = subprocess.run([exe] + case["args"], cwd=taxdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.stdout.decode("latin-1") == case["stdout"] and p.returncode == case["rc"], case["args"]
