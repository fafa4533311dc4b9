"""Oracle restatement of Consensus_BLAST_SOAP_RDP-1.1.pl against goldens made by the Perl."""
import glob
import os

from conftest import run_cmd


def test_consensus_goldens(gold, oracle_bin, tmp_path):
    cases = sorted(glob.glob(os.path.join(gold, "consensus", "*.blast.tsv")))
    assert len(cases) >= 13
    for b in cases:
        name = os.path.basename(b)[:-len(".blast.tsv")]
        r = os.path.join(gold, "consensus", name + ".rdp.tsv")
        out = tmp_path / (name + ".out.txt")
        rc, so, se = run_cmd([oracle_bin, "consensus", "-b", b, "-r", r, "-o", str(out)])
        assert rc == 0, name
        assert out.read_bytes() == open(os.path.join(gold, "consensus", name + ".out.txt"), "rb").read(), name
        want_log = open(os.path.join(gold, "consensus", name + ".log.txt"), "rb").read()
        assert so.replace(str(out).encode(), b"@OUT@") == want_log, name


def test_soap_stream_is_opened_and_ignored(gold, oracle_bin, tmp_path):
    b = os.path.join(gold, "consensus", "basic.blast.tsv")
    r = os.path.join(gold, "consensus", "basic.rdp.tsv")
    s = tmp_path / "soap.txt"
    s.write_text("r1\tACGT\thhhh\t1\ta\t4\t+\tgi|5|gb|A|\t1\t0\t4M\t4\n")
    out = tmp_path / "o.txt"
    rc, so, _ = run_cmd([oracle_bin, "consensus", "-b", b, "-r", r, "-s", str(s), "-o", str(out)])
    assert rc == 0
    assert out.read_bytes() == open(os.path.join(gold, "consensus", "basic.out.txt"), "rb").read()
    rc, so, _ = run_cmd([oracle_bin, "consensus", "-b", b, "-r", r, "-s", str(tmp_path / "missing"), "-o", str(out)])
    assert b"Error: Unable to open" in so


def test_reference_hang_input_is_flagged(gold, oracle_bin, tmp_path):
    # an RDP read without BLAST lines at or after the cursor: the Perl loops forever
    b = tmp_path / "b.tsv"
    r = tmp_path / "r.tsv"
    b.write_text("q1\t[0]Bacteria;\t99.0\t1\n")
    r.write_text("q1\t\t\t\t\tBacteria\tdomain\t1.0\nq2\t\t\t\t\tBacteria\tdomain\t1.0\n")
    rc, _, _ = run_cmd([oracle_bin, "consensus", "-b", str(b), "-r", str(r), "-o", str(tmp_path / "o")])
    assert rc == 3
