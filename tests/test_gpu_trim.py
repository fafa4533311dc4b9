"""GPU parity of the step before the hot path: the product's `trim2` command line against the goldens
printed by the reference's own Perl (Trim/trim2.4.pl == trim2.3.pl on FASTQ / QSEQ input), and pgx_trim_file against the
oracle on seeded read files large enough to cross the line indexer's tiles and the scans' blocks."""
import os
import random

import pytest

from conftest import ROOT, run_cmd
from test_oracle_trim import run_trim_case, run_trim_fasta_case, trim_cases, trim_fasta_cases
from trim_inputs import damaged as _messy, fastq_text, qseq_text, random_case

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "pangea-plus_amd", "bin")


@pytest.mark.parametrize("name,info", trim_cases())
def test_trim2_cli_matches_reference(name, info, tmp_path):
    run_trim_case([os.path.join(BIN, "trim2")], name, info, tmp_path)


@pytest.mark.parametrize("name,info", trim_fasta_cases())
def test_trim2_cli_fasta_modes_match_reference(name, info, tmp_path):
    """FASTA-format input: `-q QUAL` (parse_fasta) and `-j -b` (join_fasta) against what trim2.4.pl printed and wrote."""
    run_trim_fasta_case([os.path.join(BIN, "trim2")], name, info, tmp_path)


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
    # and without the file in between: same batch from the text in memory
    direct = pg.Reads.from_fasta_text(fasta)
    assert len(direct) == 5000
    for i in (0, 1, 77, 4999):
        assert (direct.get(i) == reads.get(i)).all()


def test_fasta_files_at_size_equal_oracle(oracle_bin, tmp_path):
    """The two FASTA modes on files large enough to cross the line indexer's tiles: 30 000 records with a quality file,
    and two files of 30 000 multi-line records joined with 25 N's."""
    import pangea_plus_amd as pg
    pg.init(0)
    rng = random.Random(31)
    seq, qual, b = [], [], []
    for i in range(30000):
        n = rng.randint(20, 320)
        s = "".join(rng.choice("ACGT") for _ in range(n))
        qv = [rng.choice((0, 0, 3, 17, 25, 40, -4)) if i % 11 == 0 else rng.randint(2, 40) for _ in range(n)]
        if i % 97 == 0:
            qv = [0] * n   # never raises the sum: printed with the previous record's range
        seq.append(">s%d len=%d\n" % (i, n) + "".join(s[k:k + 60] + "\n" for k in range(0, n, 60)))
        qual.append(">s%d len=%d\n" % (i, n) + "".join(" ".join(str(v) for v in qv[k:k + 60]) + "\n" for k in range(0, n, 60)))
        m = rng.randint(20, 320)
        t = "".join(rng.choice("ACGT") for _ in range(m))
        b.append(">s%d/2\n" % i + "".join(t[k:k + 70] + "\n" for k in range(0, m, 70)))
    (tmp_path / "a.txt").write_text("".join(seq))
    (tmp_path / "q.txt").write_text("".join(qual))
    (tmp_path / "b.txt").write_text("".join(b))
    rc, want, err = run_cmd([oracle_bin, "trim2", "-a", "a.txt", "-q", "q.txt"], cwd=tmp_path, timeout=600)
    assert rc == 0, err
    out, fasta, mode = pg.trim2(str(tmp_path / "a.txt"), q=str(tmp_path / "q.txt"))
    assert mode == pg._capi.TRIM_FASTA_QUAL and fasta == b""
    assert out.replace(str(tmp_path / "q.txt").encode(), b"q.txt", 1) == want and out.count(b"\n>s") == 29999  # all but the last record
    rc, _, err = run_cmd([oracle_bin, "trim2", "-a", "a.txt", "-b", "b.txt", "-j", "-g", "25"], cwd=tmp_path, timeout=600)
    assert rc == 0, err
    want_fasta = (tmp_path / "output_files" / "trim2" / "a.txt_runblast.fasta").read_bytes()
    out, fasta, mode = pg.trim2(str(tmp_path / "a.txt"), b=str(tmp_path / "b.txt"), g="25", j=True)
    assert mode == pg._capi.TRIM_FASTA_JOIN and out == b""
    assert fasta == want_fasta and fasta.count(b"N" * 25) >= 30000


def test_negative_truncate_is_declined(tmp_path):
    import pangea_plus_amd as pg
    pg.init(0)
    (tmp_path / "a.fq").write_bytes(b"@r\nACGT\n+\nIIII\n")
    with pytest.raises(pg.PangeaError, match="negative"):
        pg.trim2(str(tmp_path / "a.fq"), t="-3")


def test_trimmed_pairs_through_blastn_equal_the_oracle_chain(oracle_bin, tmp_path):
    """README.md:34 -> :96 as one chain: QSEQ mates cut from database sequences, trim2 -g 100 (mates joined by 100 N's:
    the read shape Trim hands to Classify), then blastn.  The product's chain must give the oracle chain's table, and
    the joined reads must find both mates on their source sequence."""
    import pangea_plus_amd as pg
    pg.init(0)
    rng = random.Random(77)
    n_seq, n_pairs, L = 400, 1500, 151
    seqs = ["".join(rng.choice("ACGT") for _ in range(1400)) for _ in range(n_seq)]
    (tmp_path / "db.fa").write_text("".join(">gi|%d|x|s%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(seqs)))
    a, b = [], []
    for i in range(n_pairs):
        src = rng.randrange(n_seq)
        o1 = rng.randrange(0, 500)
        o2 = o1 + rng.randrange(300, 700)
        for mate, off, dst in ((1, o1, a), (2, o2, b)):
            seq = list(seqs[src][off:off + L])
            for p_ in rng.sample(range(L), rng.choice((0, 0, 1, 3))):
                seq[p_] = rng.choice("ACGT.")
            qual = [40] * L if i % 5 else [40] * rng.randint(60, L) + [2] * L
            dst.append("\t".join(["M", "1", "1", str(src), str(i), "0", "ACGT", str(mate), "".join(seq),
                                  "".join(chr(64 + q) for q in qual[:L]), "1"]) + "\n")
    (tmp_path / "a.txt").write_text("".join(a))
    (tmp_path / "b.txt").write_text("".join(b))
    argv = ["-a", "a.txt", "-b", "b.txt", "-g", "100"]
    want_out, want_fasta = oracle_trim(oracle_bin, tmp_path, argv)
    messages, fasta, mode = pg.trim2(str(tmp_path / "a.txt"), b=str(tmp_path / "b.txt"), g=100)
    assert fasta == want_fasta and messages == want_out
    (tmp_path / "reads.fa").write_bytes(fasta)
    n_reads = fasta.count(b">")
    assert 1000 < n_reads < n_pairs                      # the pairs with a short high-quality prefix were dropped
    rc, _, se = run_cmd([oracle_bin, "blastn", "-query", str(tmp_path / "reads.fa"), "-db", str(tmp_path / "db.fa"), "-outfmt", "6",
                         "-out", str(tmp_path / "want.tsv"), "-num_threads", "8"], timeout=900)
    assert rc == 0, se
    pg.makeblastdb(str(tmp_path / "db.fa"), str(tmp_path / "db"))
    pg.blastn(str(tmp_path / "reads.fa"), str(tmp_path / "db"), str(tmp_path / "got.tsv"))
    got = (tmp_path / "got.tsv").read_bytes()
    assert got == (tmp_path / "want.tsv").read_bytes()
    # both mates of a joined read land on the sequence they were cut from (header field 3 carries its number)
    rows = {}
    for line in got.decode().splitlines():
        f = line.split("\t")
        rows.setdefault(f[0], []).append(f)
    both = 0
    for name, hits in rows.items():
        src = "gi|%d|x|s%s|" % (int(name.split(":")[3]) + 1, name.split(":")[3])
        own = [h for h in hits if h[1] == src]
        left = any(int(h[7]) <= 160 for h in own)
        right = any(int(h[6]) >= 200 for h in own)
        both += left and right
    assert both > 0.9 * n_reads


@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("PGX_TRIM_SEEDS", "31,32,33").split(",")])
def test_trim_random_files_and_options_equal_oracle(seed, oracle_bin, tmp_path):
    import pangea_plus_amd as pg
    pg.init(0)
    a, b, g, t = random_case(seed, 3000, 2500)
    (tmp_path / "a.txt").write_bytes(a)
    argv = ["-a", "a.txt"]
    if b is not None:
        (tmp_path / "b.txt").write_bytes(b)
        argv += ["-b", "b.txt"]
    if g is not None:
        argv += ["-g", g]
    if t is not None:
        argv += ["-t", t]
    want_out, want_fasta = oracle_trim(oracle_bin, tmp_path, argv)
    messages, fasta, mode = pg.trim2(str(tmp_path / "a.txt"), b=str(tmp_path / "b.txt") if b is not None else None, g=g, t=t)
    assert fasta == want_fasta
    assert (fasta + messages if mode == pg._capi.TRIM_FASTQ else messages) == want_out
