"""GPU parity of the step before the hot path: the product's `trim2` command line against the goldens
printed by the reference's own Perl (Trim/trim2.4.pl == trim2.3.pl on FASTQ / QSEQ input), and pgx_trim_file against the
oracle on seeded read files large enough to cross the line indexer's tiles and the scans' blocks."""
import os
import random

import pytest

from conftest import ROOT, run_cmd
from test_oracle_trim import run_trim_case, trim_cases

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "pangea-plus_amd", "bin")


@pytest.mark.parametrize("name,info", trim_cases())
def test_trim2_cli_matches_reference(name, info, tmp_path):
    run_trim_case([os.path.join(BIN, "trim2")], name, info, tmp_path)


def quality_string(rng, n, base):
    """One of the shapes the running-sum rule distinguishes: clean, tail drop, noise, dips, all low, near the cutoff."""
    kind = rng.randrange(6)
    if kind == 0:
        q = [rng.randint(30, 40) for _ in range(n)]
    elif kind == 1:
        cut = rng.randint(n // 2, n)
        q = [rng.randint(30, 40) if j < cut else rng.randint(2, 10) for j in range(n)]
    elif kind == 2:
        q = [rng.randint(2, 40) for _ in range(n)]
    elif kind == 3:
        step = rng.randint(7, 40)
        q = [2 if j % step == 0 else 38 for j in range(n)]
    elif kind == 4:
        q = [rng.randint(2, 12) for _ in range(n)]
    else:
        q = [20 + rng.choice((-2, -1, 0, 1, 2)) for _ in range(n)]
    return "".join(chr(base + v) for v in q)


def fastq_text(seed, n, lmin, lmax):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        L = rng.randint(lmin, lmax)
        seq = "".join(rng.choice("ACGTN" if i % 17 == 0 else "ACGT") for _ in range(L))
        out.append("@M%d:%d@%d extra\n%s\n+\n%s\n" % (seed, i, i % 7, seq, quality_string(rng, L, 33)))
    return "".join(out).encode("latin-1")


def qseq_text(seed, n, lmin, lmax):
    rng = random.Random(seed)
    a, b = [], []
    for i in range(n):
        xy = [str(rng.randint(1, 8)), str(rng.randint(1101, 2316)), str(rng.randint(1000, 20000)), str(rng.randint(1000, 20000))]
        for mate, dst in ((1, a), (2, b)):
            L = rng.randint(lmin, lmax)
            seq = "".join(rng.choice("ACGT.") if rng.random() < 0.02 else rng.choice("ACGT") for _ in range(L))
            dst.append("\t".join(["HWI-X", "12"] + xy + ["TTAGGC", str(mate), seq, quality_string(rng, L, 64), rng.choice("01")]) + "\n")
    return "".join(a).encode("latin-1"), "".join(b).encode("latin-1")


def oracle_trim(oracle_bin, work, argv):
    rc, out, err = run_cmd([oracle_bin, "trim2"] + argv, cwd=work, timeout=600)
    assert rc == 0, err
    return out, (work / "output_files" / "trim2" / "a.txt_runblast.fasta").read_bytes()


@pytest.mark.parametrize("paired", [False, True])
def test_fastq_file_equals_oracle(paired, oracle_bin, tmp_path):
    import pangea_plus_amd as pg
    pg.init(0)
    (tmp_path / "a.txt").write_bytes(fastq_text(5 + paired, 60001, 40, 260))
    (tmp_path / "b.txt").write_bytes(b"")
    argv = ["-a", "a.txt"] + (["-b", "b.txt", "-g", "37"] if paired else [])
    want_out, want_fasta = oracle_trim(oracle_bin, tmp_path, argv)
    out, fasta, mode = pg.trim2(str(tmp_path / "a.txt"), b=str(tmp_path / "b.txt") if paired else None, g="37" if paired else None)
    assert mode == pg._capi.TRIM_FASTQ
    assert fasta == want_fasta
    assert fasta + out == want_out  # FASTQ mode prints every record on stdout as well (trim2.4.pl:487-515)
    assert fasta.count(b">") == (60001 + paired) // (2 if paired else 1)  # every started record is printed, kept or "0"


@pytest.mark.parametrize("t", [None, "5", "30"])
def test_qseq_files_equal_oracle(t, oracle_bin, tmp_path):
    import pangea_plus_amd as pg
    pg.init(0)
    a, b = qseq_text(9, 50000, 90, 200)
    (tmp_path / "a.txt").write_bytes(a)
    (tmp_path / "b.txt").write_bytes(b)
    argv = ["-a", "a.txt", "-b", "b.txt", "-g", "100"] + (["-t", t] if t else [])
    want_out, want_fasta = oracle_trim(oracle_bin, tmp_path, argv)
    out, fasta, mode = pg.trim2(str(tmp_path / "a.txt"), b=str(tmp_path / "b.txt"), g="100", t=t)
    assert mode == pg._capi.TRIM_QSEQ
    assert fasta == want_fasta
    assert out == want_out == b"QSEQ file format found.\nTrimming complete.\n"
    # size-independent properties: a written pair has both mates >= 70 bases around exactly 100 N's, and no '.' survives
    lines = fasta.split(b"\n")[:-1]
    assert len(lines) % 2 == 0 and len(lines) > 1000
    for hdr, seq in zip(lines[0::2], lines[1::2]):
        assert hdr.startswith(b">HWI-X:12:") and hdr.endswith(b":1:AB")
        assert b"." not in seq and len(seq) >= 240


def test_trimmed_fasta_feeds_the_read_importer(tmp_path):
    """The runblast FASTA is what blastn takes next (README.md:96): it imports as one read per FASTQ record."""
    import pangea_plus_amd as pg
    pg.init(0)
    (tmp_path / "a.txt").write_bytes(fastq_text(3, 5000, 80, 160))
    out, fasta, mode = pg.trim2(str(tmp_path / "a.txt"))
    (tmp_path / "reads.fa").write_bytes(fasta)
    reads = pg.Reads.from_fasta(str(tmp_path / "reads.fa"))
    assert len(reads) == 5000


def test_fasta_input_and_negative_truncate_are_declined(tmp_path):
    import pangea_plus_amd as pg
    pg.init(0)
    (tmp_path / "a.fa").write_bytes(b">r1\nACGT\n")
    with pytest.raises(pg.PangeaError, match="FASTA"):
        pg.trim2(str(tmp_path / "a.fa"))
    (tmp_path / "a.fq").write_bytes(b"@r\nACGT\n+\nIIII\n")
    with pytest.raises(pg.PangeaError, match="negative"):
        pg.trim2(str(tmp_path / "a.fq"), t="-3")
