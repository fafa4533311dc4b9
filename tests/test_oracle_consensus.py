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


def test_three_way_vote_rules(oracle_bin, tmp_path):
    """pgx-vote3 v1 (opt-in extension, oracle/o_consensus.c): two of three equal non-empty names agree a rank; the result is
    the longest agreed prefix; a read missing from a table votes with the other two."""
    five = "\t\t\t\t\t"
    lin = lambda *names: "".join("[%d]%s;" % (k, n) for k, n in enumerate(names))
    blast = ["a\t" + lin("Bacteria", "Firmicutes", "Bacilli", "X") + "\t99.0\t100",
             "a\t" + lin("Archaea") + "\t90.0\t100",                      # not the first row of the read: no vote
             "b\t" + lin("Bacteria", "Firmicutes") + "\t99.0\t100",
             "d\t" + lin("Bacteria", "Proteobacteria") + "\t99.0\t100"]
    soap = ["a\t" + lin("Bacteria", "Firmicutes", "Clostridia") + "\t100.00\t50",
            "c\t" + lin("Bacteria", "Firmicutes") + "\t100.00\t50",
            "d\t" + lin("Bacteria", "Firmicutes") + "\t100.00\t50"]
    rdp = ["a" + five + "Bacteria\tdomain\t1.0\tFirmicutes\tphylum\t0.9\t\"Bacilli\"\tclass\t0.8",
           "b" + five + "Bacteria\tdomain\t1.0\tActinobacteria\tphylum\t0.9",
           "c" + five + "Bacteria\tdomain\t1.0\tFirmicutes\tphylum\t0.9",
           "d" + five + "Archaea\tdomain\t1.0\tFirmicutes\tphylum\t0.9",
           "e" + five + "Bacteria\tdomain\t1.0"]
    for name, rows in (("b.tsv", blast), ("s.tsv", soap), ("r.txt", rdp)):
        (tmp_path / name).write_text("\n".join(rows) + "\n")
    assert run_cmd([oracle_bin, "vote3", str(tmp_path / "b.tsv"), str(tmp_path / "r.txt"), str(tmp_path / "s.tsv"),
                    str(tmp_path / "o.txt")])[0] == 0
    got = (tmp_path / "o.txt").read_text().splitlines()
    assert got == ["a\t[0]Bacteria;[1]Firmicutes;[2]Bacilli;\t3\t332",   # B = S = R twice, then B = R against S
                   "b\t[0]Bacteria;\t1\t2",                              # no SOAP row: B = R on the domain only
                   "c\t[0]Bacteria;[1]Firmicutes;\t2\t22",               # no BLAST row: S = R
                   "d\t[0]Bacteria;[1]Firmicutes;\t2\t22",               # B = S on the domain, S = R on the phylum
                   "e\t\t0\t"]
