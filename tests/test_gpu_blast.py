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
    # odd batch sizes and offsets: two short reads share a wavefront, the last read of an odd batch has no partner
    lines = want.decode().splitlines(True)
    for first, count in ((0, 1), (7, 1), (5, 333), (1001, 1999), (2998, 2)):
        sub = pg.Reads.from_synth(cfg, first, count)
        got = _capi.blast_search(db, sub).format(db, sub).decode()
        names = {"r%d" % i for i in range(first, first + count)}
        assert got == "".join(l for l in lines if l.split("\t", 1)[0] in names), (first, count)


def test_blast_fasta_path_matches_oracle_bytes(pg, workload, tmp_path, monkeypatch):
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
    # large query files are streamed in pieces that end at record boundaries: force tiny pieces (a few hundred reads,
    # and one smaller than a record) and shard on top of it
    for piece in ("50000", "7001", "100"):
        monkeypatch.setenv("PGX_BLASTN_PIECE", piece)
        s = tmp_path / ("stream%s.tsv" % piece)
        pg.blastn(str(workload / "reads.fa"), str(tmp_path / "db"), str(s))
        assert s.read_bytes() == out.read_bytes(), piece
    # a batch whose hit table would not fit is halved until it does (here: a pretend limit of 5 000 hits)
    monkeypatch.setenv("PGX_HIT_LIMIT", "5000")
    s = tmp_path / "halved.tsv"
    pg.blastn(str(workload / "reads.fa"), str(tmp_path / "db"), str(s))
    assert s.read_bytes() == out.read_bytes()
    monkeypatch.delenv("PGX_HIT_LIMIT")
    parts = b""
    for rk in range(3):
        p = tmp_path / ("spart%d.tsv" % rk)
        pg.blastn(str(workload / "reads.fa"), str(tmp_path / "db"), str(p), rank=rk, world_size=3)
        parts += p.read_bytes()
    assert parts == out.read_bytes()


def _blast_text(pg, db_fa, reads_fa, tmp_path, tag):
    out = tmp_path / (tag + ".tsv")
    pg.makeblastdb(str(db_fa), str(tmp_path / (tag + "_db")))
    pg.blastn(str(reads_fa), str(tmp_path / (tag + "_db")), str(out))
    return out.read_bytes()


def test_direct_address_index_paths_match_oracle(pg, workload, oracle_bin, tmp_path, monkeypatch):
    """bits = 32 turns on the index-side duplicate filter (and its proxy for seeds that straddle two
    subjects); it is chosen automatically only above 0.5 Gbp, so force it here."""
    monkeypatch.setenv("PGX_INDEX_BITS", "32")
    assert _blast_text(pg, workload / "db.fa", workload / "reads.fa", tmp_path, "b32") == \
        open(workload / "oracle.tsv", "rb").read()
    # adjacent database sequences that continue each other: seeds straddle the boundaries
    import random
    rng = random.Random(9)
    g = "".join(rng.choice("ACGT") for _ in range(3000))
    cuts = [0, 200, 413, 655, 1000, 1013, 1500, 2100, 3000]
    db = tmp_path / "adj.fa"
    db.write_text("".join(">gi|%d|x|p%d|\n%s\n" % (i + 1, i, g[a:b]) for i, (a, b) in enumerate(zip(cuts, cuts[1:]))))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads = []
    for i in range(400):
        o = rng.randrange(0, 3000 - 150)
        w = list(g[o:o + 150])
        for p_ in rng.sample(range(150), rng.choice([0, 0, 1, 2, 4])):
            w[p_] = rng.choice([b for b in "ACGT" if b != w[p_]])
        w = "".join(w)
        if i % 2:
            w = "".join(comp[c] for c in reversed(w))
        reads.append(">a%d\n%s\n" % (i, w))
    rd = tmp_path / "adj_reads.fa"
    rd.write_text("".join(reads))
    want = tmp_path / "adj_oracle.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want)])[0] == 0
    assert len(want.read_bytes()) > 20000
    assert _blast_text(pg, db, rd, tmp_path, "adj32") == want.read_bytes()
    monkeypatch.delenv("PGX_INDEX_BITS")
    assert _blast_text(pg, db, rd, tmp_path, "adjauto") == want.read_bytes()


def test_direct_address_index_top_buckets(pg, oracle_bin, tmp_path, monkeypatch):
    """16-mers ending in a run of T are the numerically largest keys: the last entries of the 2^32 + 1 bucket
    table (a library scan once returned a doubled prefix for exactly the last 4 097 of them).  Poly-T / poly-A
    stretches of 30-45 bases put seeds there on both strands."""
    import random
    rng = random.Random(33)
    seqs = []
    for i in range(12):
        left = "".join(rng.choice("ACGT") for _ in range(260))
        right = "".join(rng.choice("ACGT") for _ in range(260))
        run = ("T" if i % 2 == 0 else "A") * rng.randrange(30, 46)
        seqs.append(left + "G" + run + "C" + right)
    db = tmp_path / "polyt.fa"
    db.write_text("".join(">gi|%d|x|t%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(seqs)))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads = []
    for i in range(120):
        s = seqs[i % len(seqs)]
        o = rng.randrange(150, 330)
        w = list(s[o:o + 150])
        for p_ in rng.sample(range(150), rng.choice([0, 1, 2])):
            w[p_] = rng.choice([b for b in "ACGT" if b != w[p_]])
        w = "".join(w)
        if i % 3 == 0:
            w = "".join(comp[c] for c in reversed(w))
        reads.append(">t%d\n%s\n" % (i, w))
    rd = tmp_path / "polyt_reads.fa"
    rd.write_text("".join(reads))
    want = tmp_path / "polyt_oracle.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want)])[0] == 0
    assert len(want.read_bytes()) > 5000
    monkeypatch.setenv("PGX_INDEX_BITS", "32")
    assert _blast_text(pg, db, rd, tmp_path, "polyt32") == want.read_bytes()
    monkeypatch.delenv("PGX_INDEX_BITS")
    assert _blast_text(pg, db, rd, tmp_path, "polytauto") == want.read_bytes()


def test_long_queries_ambiguity_codes_and_heavy_reads(pg, oracle_bin, tmp_path, monkeypatch):
    """1 400-bp queries (config 2 shape: lazy masks, >64 probes per strand), IUPAC codes on both sides, and a
    repeated database in which every read collects hundreds of hits (fragmented output, big-read ordering,
    500-subject limit)."""
    import random
    rng = random.Random(21)
    base = "".join(rng.choice("ACGT") for _ in range(1600))
    seqs = []
    for i in range(700):
        s = list(base)
        for p_ in rng.sample(range(1600), 16):
            s[p_] = rng.choice("ACGT")
        if i % 7 == 0:
            for p_ in rng.sample(range(1600), 5):
                s[p_] = rng.choice("NRYKM")
        seqs.append("".join(s))
    db = tmp_path / "rep.fa"
    db.write_text("".join(">gi|%d|x|r%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(seqs)))
    reads = []
    for i in range(60):
        L = rng.choice([150, 150, 400, 1400])
        o = rng.randrange(0, 1600 - L)
        w = list(rng.choice(seqs)[o:o + L])
        for p_ in rng.sample(range(L), max(1, L // 60)):
            w[p_] = rng.choice("ACGTN")
        reads.append(">h%d\n%s\n" % (i, "".join(w)))
    rd = tmp_path / "rep_reads.fa"
    rd.write_text("".join(reads))
    want = tmp_path / "rep_oracle.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want),
                    "-num_threads", "8"], timeout=900)[0] == 0
    lines = want.read_text().splitlines()
    per_read = {}
    for l in lines:
        per_read.setdefault(l.split("\t")[0], set()).add(l.split("\t")[1])
    assert max(len(v) for v in per_read.values()) == 500      # the 500-subject limit is exercised
    assert _blast_text(pg, db, rd, tmp_path, "rep") == want.read_bytes()
    monkeypatch.setenv("PGX_INDEX_BITS", "32")
    assert _blast_text(pg, db, rd, tmp_path, "rep32") == want.read_bytes()


def test_very_long_query_and_thousands_of_hits(pg, oracle_bin, tmp_path):
    """(a) one query longer than 65 535 bases sends the whole batch, short reads included, through the
    segmented-sort ordering (the packed LDS keys hold 16-bit query coordinates); (b) reads that hit 3 000
    subjects each: ordering and the 500-subject cut must not depend on a per-read size limit."""
    import random
    rng = random.Random(77)
    big = "".join(rng.choice("ACGT") for _ in range(70000))
    others = ["".join(rng.choice("ACGT") for _ in range(900)) for _ in range(30)]
    db = tmp_path / "long.fa"
    db.write_text(">gi|1|x|big|\n%s\n" % big + "".join(">gi|%d|x|s%d|\n%s\n" % (i + 2, i, s) for i, s in enumerate(others)))
    q = list(big[1500:1500 + 66000])
    for p_ in rng.sample(range(len(q)), 400):
        q[p_] = rng.choice([b for b in "ACGT" if b != q[p_]])
    reads = [">longq\n%s\n" % "".join(q)]
    for i in range(40):
        s = others[i % len(others)]
        o = rng.randrange(0, 900 - 150)
        reads.append(">s%d\n%s\n" % (i, s[o:o + 150]))
    rd = tmp_path / "long_reads.fa"
    rd.write_text("".join(reads))
    want = tmp_path / "long_oracle.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want)], timeout=600)[0] == 0
    assert len(want.read_bytes()) > 2000 and b"longq\t" in want.read_bytes()
    assert _blast_text(pg, db, rd, tmp_path, "long") == want.read_bytes()

    base = "".join(rng.choice("ACGT") for _ in range(300))
    seqs = []
    for i in range(3000):
        s = list(base)
        for p_ in rng.sample(range(300), 3):
            s[p_] = rng.choice("ACGT")
        seqs.append("".join(s))
    db2 = tmp_path / "many.fa"
    db2.write_text("".join(">gi|%d|x|m%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(seqs)))
    reads = []
    for i in range(12):
        o = rng.randrange(0, 300 - 150)
        w = list(rng.choice(seqs)[o:o + 150])
        for p_ in rng.sample(range(150), 2):
            w[p_] = rng.choice("ACGT")
        reads.append(">m%d\n%s\n" % (i, "".join(w)))
    rd2 = tmp_path / "many_reads.fa"
    rd2.write_text("".join(reads))
    want2 = tmp_path / "many_oracle.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd2), "-db", str(db2), "-outfmt", "6", "-out", str(want2)], timeout=600)[0] == 0
    lines = want2.read_bytes().decode().splitlines()
    assert len({l.split("\t")[1] for l in lines if l.startswith("m0\t")}) == 500
    assert _blast_text(pg, db2, rd2, tmp_path, "many") == want2.read_bytes()


@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("PGX_FUZZ_SEEDS", "101,202,303").split(",")])
def test_random_mixtures_match_oracle(pg, oracle_bin, tmp_path, seed):
    """Fuzz: databases of 40-3 000-base sequences (repeats, N runs, IUPAC letters, near-identical copies), reads of
    20-320 bases from either strand with substitutions and Ns: short reads (< 28: no hit possible), dense (<= 192) and
    long (> 192) reads in one batch, hits at sequence ends, reads spanning two adjacent sequences."""
    import random
    rng = random.Random(seed)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "R": "Y", "Y": "R"}
    seqs = []
    for i in range(150):
        L = rng.choice([40, 60, 150, 300, 700, 1500, 3000])
        if seqs and rng.random() < 0.35:
            s = list(rng.choice(seqs))[:L]
            for p_ in rng.sample(range(len(s)), max(1, len(s) // 40)):
                s[p_] = rng.choice("ACGT")
            s = "".join(s)
        else:
            s = "".join(rng.choice("ACGT") for _ in range(L))
        if rng.random() < 0.2 and L > 100:
            a = rng.randrange(0, L - 30)
            s = s[:a] + "N" * rng.choice([1, 3, 12]) + s[a:]
        if rng.random() < 0.1:
            a = rng.randrange(0, len(s))
            s = s[:a] + rng.choice("RYKM") + s[a + 1:]
        if rng.random() < 0.1 and L > 200:
            a = rng.randrange(0, L - 60)
            s = s[:a] + s[a:a + 40] * 2 + s[a:]     # tandem repeat
        seqs.append(s)
    db = tmp_path / "fz.fa"
    db.write_text("".join(">gi|%d|x|f%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(seqs)))
    cat = "".join(seqs)
    reads = []
    for i in range(1500):
        L = rng.choice([20, 27, 28, 29, 40, 75, 100, 150, 150, 150, 192, 193, 250, 320])
        if rng.random() < 0.15:
            o = rng.randrange(0, max(1, len(cat) - L))
            w = list(cat[o:o + L])                  # may span two adjacent database sequences
        else:
            s = rng.choice(seqs)
            if len(s) < L:
                w = list(s)
            else:
                o = rng.choice([0, len(s) - L, rng.randrange(0, len(s) - L + 1)])
                w = list(s[o:o + L])
        for p_ in rng.sample(range(len(w)), rng.choice([0, 0, 1, 2, 3, len(w) // 20])):
            w[p_] = rng.choice("ACGTN")
        if rng.random() < 0.35 and len(w) > 40:       # 454-style: insertions and deletions (the gapped stage, spec v2)
            for _ in range(rng.choice([1, 1, 2, 3])):
                p_ = rng.randrange(1, len(w) - 1)
                if rng.random() < 0.5:
                    del w[p_:p_ + rng.choice([1, 1, 2, 4])]
                else:
                    w[p_:p_] = [rng.choice("ACGT") for _ in range(rng.choice([1, 1, 2, 4]))]
        w = "".join(w)
        if rng.random() < 0.5:
            w = "".join(comp.get(c, "N") for c in reversed(w))
        reads.append(">z%d\n%s\n" % (i, w))
    rd = tmp_path / "fz_reads.fa"
    rd.write_text("".join(reads))
    want = tmp_path / "fz_oracle.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want)], timeout=600)[0] == 0
    assert len(want.read_bytes()) > 50000
    assert _blast_text(pg, db, rd, tmp_path, "fz") == want.read_bytes()


def test_ungapped_flag_gives_spec_v1(pg, workload, oracle_bin, tmp_path):
    """`blastn -ungapped` (BLAST's own flag) stops after the ungapped stage: the round-1 tables, gapopen 0 everywhere;
    through the file verb and through a database handle."""
    from pangea_plus_amd import _capi
    want = tmp_path / "v1.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(workload / "reads.fa"), "-db", str(workload / "db.fa"), "-outfmt", "6",
                    "-out", str(want), "-num_threads", "8", "-ungapped"], timeout=600)[0] == 0
    v1 = want.read_bytes()
    assert v1 != open(workload / "oracle.tsv", "rb").read() and all(l.split(b"\t")[5] == b"0" for l in v1.splitlines())
    pg.makeblastdb(str(workload / "db.fa"), str(tmp_path / "db"))
    out = tmp_path / "hits.tsv"
    pg.blastn(str(workload / "reads.fa"), str(tmp_path / "db"), str(out), ungapped=True)
    assert out.read_bytes() == v1
    cfg = pg.SynthCfg.default(**SHAPE)
    db = pg.Db.from_synth(cfg)
    db.set_ungapped(True)
    reads = pg.Reads.from_synth(cfg, 0, N_READS)
    assert _capi.blast_search(db, reads).format(db, reads) == v1
    db.set_ungapped(False)
    assert _capi.blast_search(db, reads).format(db, reads) == open(workload / "oracle.tsv", "rb").read()
    # the e-value / bit-score columns follow the TABLE: a table of a gapped search formatted after the handle was switched
    # to -ungapped (and the other way round) keeps its own statistics
    v2_hits = _capi.blast_search(db, reads)
    db.set_ungapped(True)
    assert v2_hits.format(db, reads) == open(workload / "oracle.tsv", "rb").read()
    v1_hits = _capi.blast_search(db, reads)
    db.set_ungapped(False)
    assert v1_hits.format(db, reads) == v1
    ev, bs = _capi.blast_score_columns(28, 150, 10**9, 666667)
    evu, bsu = _capi.blast_score_columns(28, 150, 10**9, 666667, gapped=False)
    assert bs == bsu and ev != evu     # (the length adjustment differs, the bit score does not)


def test_dust_inside_every_search_gives_the_same_table(pg, workload):
    """pgx_db_set_dust_each_search: S3d computed again by the search itself, on its stream, without a host wait -- the same bits,
    the same table, and the stage shows up in the stage times.  Low-complexity reads among the ordinary ones."""
    from pangea_plus_amd import _capi
    import random
    rng = random.Random(5)
    text = open(workload / "reads.fa").read().split(">")[1:201]
    recs = []
    for i, rec in enumerate(text):
        name, seq = rec.split("\n", 1)
        seq = seq.replace("\n", "")
        if i % 5 == 0:
            p_ = rng.randrange(10, 100)
            seq = seq[:p_] + rng.choice(["A" * 20, "AC" * 12, "GAT" * 9]) + seq[p_ + 24:]
        recs.append(">%s\n%s\n" % (name, seq))
    fa = workload / "dusty.fa"
    fa.write_text("".join(recs))
    db = pg.Db.from_fasta(str(workload / "db.fa"))
    reads = pg.Reads.from_fasta(str(fa))
    want = _capi.blast_search(db, reads).format(db, reads)
    assert _capi.stage_times().dust_ms == 0
    db.set_dust_each_search(True)
    for _ in range(2):
        assert _capi.blast_search(db, reads).format(db, reads) == want
        assert _capi.stage_times().dust_ms > 0
    db.set_dust(False)
    nodust = _capi.blast_search(db, reads).format(db, reads)
    assert _capi.stage_times().dust_ms == 0 and nodust != want      # (-dust no: more seeds, and no masking stage)


def test_issue_probe_reports_a_rate(pg):
    """pgx_probe_issue (the measured instruction roof bench.py quotes beside the gapped stage): every kind runs and reports a
    plausible rate; two-operand adds issue faster than three-operand maxima."""
    import ctypes as C
    L = pg.lib()
    L.pgx_probe_issue_name.restype = C.c_char_p
    names = []
    k = 0
    while L.pgx_probe_issue_name(k):
        names.append(L.pgx_probe_issue_name(k).decode())
        k += 1
    assert names[0] == "v_add_u32" and names[1] == "v_max3_i32" and len(names) > 20
    rate = {}
    for kind in (0, 1):
        out = (C.c_double * 4)()
        assert L.pgx_probe_issue(4, kind, out) == 0
        rate[kind] = out[0]
        assert 1e8 < out[0] < 3e9
    assert rate[0] > 1.3 * rate[1]
    out = (C.c_double * 4)()
    assert L.pgx_probe_issue(1, 0, out) == 0 and 1e9 < out[3] < 3e9     # the shader clock, from a lone wavefront per SIMD


@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("PGX_INDEL_SEEDS", "5,6").split(",")])
def test_reads_with_insertions_and_deletions(pg, oracle_bin, tmp_path, seed):
    """Spec v2: 454 / Ion-style reads (homopolymer length errors, random indels) and 1 400-base queries with indels against
    a 16S-like family: gapopen > 0 rows, seeds on either side of a gap merged into one row (S3c), the one-wavefront-per-HSP
    kernel for the long queries and for sides with more than 15 differences."""
    import random
    rng = random.Random(seed)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    anc = "".join(rng.choice("ACGT") for _ in range(1500))
    # homopolymer runs, where length errors live
    for _ in range(25):
        a = rng.randrange(0, 1450)
        anc = anc[:a] + rng.choice("ACGT") * rng.randrange(3, 9) + anc[a + 6:]
    anc = anc[:1500]
    seqs = []
    for i in range(60):
        s = list(anc)
        for p_ in rng.sample(range(len(s)), 30 + (i % 5) * 25):
            s[p_] = rng.choice("ACGT")
        for _ in range(i % 4):
            p_ = rng.randrange(10, len(s) - 10)
            if rng.random() < 0.5:
                del s[p_:p_ + rng.randrange(1, 6)]
            else:
                s[p_:p_] = [rng.choice("ACGT") for _ in range(rng.randrange(1, 6))]
        seqs.append("".join(s))
    db = tmp_path / "fam.fa"
    db.write_text("".join(">gi|%d|x|s%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(seqs)))
    reads = []
    for i in range(500):
        L = rng.choice([100, 150, 150, 250, 400, 400, 1400])
        s = rng.choice(seqs)
        o = rng.randrange(0, max(1, len(s) - L))
        w = list(s[o:o + L])
        for p_ in rng.sample(range(len(w)), rng.randrange(0, 1 + len(w) // 40)):
            w[p_] = rng.choice("ACGT")
        for _ in range(rng.choice([0, 1, 1, 2, 3, 6]) * (1 + L // 500)):
            p_ = rng.randrange(1, len(w) - 1)
            if rng.random() < 0.5:           # homopolymer length error
                run = 1
                while p_ + run < len(w) and w[p_ + run] == w[p_]:
                    run += 1
                if rng.random() < 0.5 and run > 1:
                    del w[p_]
                else:
                    w.insert(p_, w[p_])
            elif rng.random() < 0.5:
                del w[p_:p_ + rng.choice([1, 2, 3, 7])]
            else:
                w[p_:p_] = [rng.choice("ACGT") for _ in range(rng.choice([1, 2, 3, 7]))]
        w = "".join(w)
        if i % 2:
            w = "".join(comp[c] for c in reversed(w))
        reads.append(">i%d\n%s\n" % (i, w))
    rd = tmp_path / "indel_reads.fa"
    rd.write_text("".join(reads))
    want = tmp_path / "indel_oracle.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want), "-num_threads", "8"],
                   timeout=900)[0] == 0
    rows = want.read_bytes().splitlines()
    gapped = [r for r in rows if r.split(b"\t")[5] != b"0"]
    assert len(rows) > 10000 and len(gapped) > len(rows) // 4
    assert max(int(r.split(b"\t")[5]) for r in gapped) >= 8
    assert _blast_text(pg, db, rd, tmp_path, "indel") == want.read_bytes()


@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("PGX_DUST_SEEDS", "8,9").split(",")])
def test_low_complexity_reads_are_masked_for_seeding(pg, oracle_bin, tmp_path, seed):
    """Spec v2, S3d (`-dust "20 64 1"`, BLAST+'s default): homopolymers, di- and tri-nucleotide repeats, AT-rich stretches
    in reads and database; seeds inside masked stretches vanish, extensions run through them; `-dust no` switches it off.
    Reads of 100-600 bases (dense flags in registers and the any-length path) and 1 400 bases."""
    import random
    rng = random.Random(seed)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}

    def lowc(n):
        kind = rng.randrange(5)
        if kind == 0:
            return rng.choice("ACGT") * n
        if kind == 1:
            return (rng.choice(["AC", "GT", "AT", "CG", "AG"]) * n)[:n]
        if kind == 2:
            return ("".join(rng.choice("ACGT") for _ in range(3)) * n)[:n]
        if kind == 3:
            return "".join(rng.choice("AAAAAAT") for _ in range(n))
        return "".join(rng.choice("ACGT") for _ in range(n))
    seqs = []
    for i in range(80):
        L = rng.choice([300, 800, 1600])
        s = "".join(rng.choice("ACGT") for _ in range(L))
        for _ in range(rng.choice([1, 2, 4])):
            n = rng.choice([7, 9, 15, 30, 45, 70, 120])
            p_ = rng.randrange(0, L - n)
            s = s[:p_] + lowc(n) + s[p_ + n:]
        if seqs and i % 3 == 0:
            base = list(rng.choice(seqs))
            for p_ in rng.sample(range(len(base)), len(base) // 40):
                base[p_] = rng.choice("ACGT")
            s = "".join(base)
        seqs.append(s)
    db = tmp_path / "lc.fa"
    db.write_text("".join(">gi|%d|x|l%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(seqs)))
    reads = []
    for i in range(700):
        s = rng.choice(seqs)
        L = min(len(s), rng.choice([100, 150, 150, 250, 400, 600, 1400]))
        o = rng.randrange(0, len(s) - L + 1)
        w = list(s[o:o + L])
        for p_ in rng.sample(range(L), rng.choice([0, 1, 2, L // 50])):
            w[p_] = rng.choice("ACGTN")
        if rng.random() < 0.3:
            p_ = rng.randrange(1, L - 1)
            if rng.random() < 0.5:
                del w[p_:p_ + rng.choice([1, 2, 5])]
            else:
                w[p_:p_] = list(lowc(rng.choice([1, 3, 8])))
        w = "".join(w)
        if i % 2:
            w = "".join(comp[c] for c in reversed(w))
        reads.append(">c%d\n%s\n" % (i, w))
    rd = tmp_path / "lc_reads.fa"
    rd.write_text("".join(reads))
    want, want_off = tmp_path / "lc_oracle.tsv", tmp_path / "lc_oracle_nodust.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want), "-num_threads", "8"], timeout=900)[0] == 0
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want_off), "-num_threads", "8",
                    "-dust", "no"], timeout=900)[0] == 0
    assert want.read_bytes() != want_off.read_bytes() and len(want.read_bytes()) > 20000
    pg.makeblastdb(str(db), str(tmp_path / "lcdb"))
    out = tmp_path / "lc.tsv"
    pg.blastn(str(rd), str(tmp_path / "lcdb"), str(out))
    assert out.read_bytes() == want.read_bytes()
    pg.blastn(str(rd), str(tmp_path / "lcdb"), str(out), dust=False)
    assert out.read_bytes() == want_off.read_bytes()


def test_command_lines_with_the_reference_flags(workload, tmp_path):
    """`makeblastdb -in F -out P -dbtype nucl` (README.md:62) and `blastn -query F -db P -outfmt 6 -out O`
    (README.md:96) as executables, incl. the rank/world_size extension that replaces mpiblastn's partition."""
    import subprocess
    from conftest import ROOT
    bin_dir = os.path.join(ROOT, "pangea-plus_amd", "bin")
    p = subprocess.run([os.path.join(bin_dir, "makeblastdb"), "-in", str(workload / "db.fa"), "-out", str(tmp_path / "nt"), "-dbtype", "nucl"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    p = subprocess.run([os.path.join(bin_dir, "blastn"), "-query", str(workload / "reads.fa"), "-db", str(tmp_path / "nt"), "-outfmt", "6",
                        "-out", str(tmp_path / "hits.tsv")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    assert (tmp_path / "hits.tsv").read_bytes() == open(workload / "oracle.tsv", "rb").read()
    parts = b""
    for rk in range(2):
        o = tmp_path / ("r%d.tsv" % rk)
        p = subprocess.run([os.path.join(bin_dir, "blastn"), "-query", str(workload / "reads.fa"), "-db", str(tmp_path / "nt"), "-outfmt", "6",
                            "-out", str(o), "-rank", str(rk), "-world_size", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr
        parts += o.read_bytes()
    assert parts == open(workload / "oracle.tsv", "rb").read()
    # a missing query file is an error with a message, not a crash
    p = subprocess.run([os.path.join(bin_dir, "blastn"), "-query", str(tmp_path / "nope.fa"), "-db", str(tmp_path / "nt"), "-outfmt", "6",
                        "-out", str(tmp_path / "x.tsv")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and p.stderr


@pytest.mark.parametrize("read_len", [193, 250, 320, 321, 450, 512, 513])
def test_reads_of_200_to_512_bases_keep_their_flags_in_registers(pg, oracle_bin, tmp_path, read_len):
    """MiSeq / 454-length reads: diagonals of up to 320 (512) bases are held as 5 (8) x 64 flags in registers (two reads per
    wavefront, one candidate per diagonal); 513 falls to the any-length path.  Clean ACGT, so the unambiguous kernels run."""
    from pangea_plus_amd import _capi
    args = ["--n-seq", "1500", "--seq-len", "700", "--n-genus", "40", "--read-len", str(read_len)]
    rd, out = tmp_path / "reads.fa", tmp_path / "oracle.tsv"
    db_fa = tmp_path / "db.fa"
    assert run_cmd([oracle_bin, "synth", "db", "--out", str(db_fa)] + args)[0] == 0
    assert run_cmd([oracle_bin, "synth", "reads", "--out", str(rd), "--count", "1201"] + args)[0] == 0
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db_fa), "-outfmt", "6", "-out", str(out), "-num_threads", "8"],
                   timeout=600)[0] == 0
    cfg = pg.SynthCfg.default(n_seq=1500, seq_len=700, n_genus=40, read_len=read_len)
    db = pg.Db.from_synth(cfg)
    reads = pg.Reads.from_synth(cfg, 0, 1201)
    hits = _capi.blast_search(db, reads)
    want = out.read_bytes()
    assert len(want) > 100000
    assert hits.format(db, reads) == want


def test_degenerate_inputs(pg, oracle_bin, tmp_path):
    """Empty read files, reads shorter than a seed, a one-sequence database shorter than a word, a database with an
    empty record: no crash, the same (mostly empty) table as the oracle."""
    from pangea_plus_amd import _capi
    db = tmp_path / "tiny.fa"
    db.write_text(">gi|1|x|a|\nACGTACGTACGTACGTACGT\n>gi|2|x|empty|\n>gi|3|x|b|\n" + "ACGTTGCAAGGCTTAACCGGATATCGCGAATTCCGGTTAACC" * 3 + "\n")
    cases = {
        "none": "",
        "short": ">s1\nACGT\n>s2\nACGTACGTACGTAC\n>s3\n\n",
        "mixed": ">m1\nACGTTGCAAGGCTTAACCGGATATCGCGAATTCCGGTTAACC\n>m2\nAC\n>m3\n" + "ACGTTGCAAGGCTTAACCGGATATCGCGAATTCCGGTTAACC" * 2 + "\n",
    }
    for tag, text in cases.items():
        rd = tmp_path / (tag + ".fa")
        rd.write_text(text)
        want = tmp_path / (tag + ".oracle.tsv")
        assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want)])[0] == 0
        assert _blast_text(pg, db, rd, tmp_path, "deg_" + tag) == want.read_bytes(), tag
    assert (tmp_path / "mixed.oracle.tsv").read_bytes()     # the mixed case does find hits


@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("PGX_PIECE_SEEDS", "5,6").split(",")])
def test_reads_with_long_unknown_runs_are_searched_in_pieces(pg, oracle_bin, tmp_path, monkeypatch, seed):
    """Reads holding a run of 6 or more non-ACGT letters (mates joined by N's, masked middles) are searched as the
    stretches between the runs and put back together (engine.hpp: pgx_reads::pieces); the oracle searches them whole.
    Runs of every length around the threshold, at the ends, several per read, both sides on one diagonal or not."""
    import random
    rng = random.Random(seed)
    seqs = ["".join(rng.choice("ACGT") for _ in range(900)) for _ in range(250)]
    for i in range(0, 250, 9):  # near copies: several subjects per read, mismatches next to the runs
        s = list(seqs[i - 1])
        for p_ in rng.sample(range(900), 25):
            s[p_] = rng.choice("ACGT")
        seqs[i] = "".join(s)
    db = tmp_path / "db.fa"
    db.write_text("".join(">gi|%d|x|s%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(seqs)))

    def unknown(n):
        return "".join(rng.choice("NNNNNRYKMSWBDHV") for _ in range(n))

    reads = []
    for i in range(700):
        kind = i % 7
        src = rng.choice(seqs)
        if kind == 0:      # masked middle: both sides stay on one diagonal
            L = rng.randint(120, 420)
            o = rng.randrange(0, 900 - L)
            w = list(src[o:o + L])
            a = rng.randrange(0, L - 20)
            k = rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 30, 100])
            w[a:a + k] = unknown(min(k, L - a))
            r = "".join(w)
        elif kind == 1:    # mates of one subject joined by a gap of another length than their distance
            o1 = rng.randrange(0, 300)
            o2 = o1 + rng.randint(150, 400)
            r = src[o1:o1 + rng.randint(60, 150)] + "N" * rng.choice([5, 6, 7, 100, 189]) + src[o2:o2 + rng.randint(60, 150)]
        elif kind == 2:    # mates of two subjects, reverse-complemented second mate
            other = rng.choice(seqs)
            m2 = other[100:100 + 130][::-1].translate(str.maketrans("ACGT", "TGCA"))
            r = src[50:180] + "N" * rng.choice([6, 11, 100]) + m2
        elif kind == 3:    # runs at the ends, short stretches between runs
            core = src[200:200 + rng.randint(28, 200)]
            r = unknown(rng.choice([0, 3, 6, 20])) + core + unknown(rng.choice([0, 5, 6, 40])) + src[500:500 + rng.choice([10, 27, 28, 60])] + unknown(rng.choice([0, 6]))
        elif kind == 4:    # several runs, pieces with short runs inside
            parts = []
            for _ in range(rng.randint(2, 5)):
                o = rng.randrange(0, 750)
                p_ = list(src[o:o + rng.randint(20, 140)])
                if rng.random() < 0.5 and len(p_) > 40:
                    p_[rng.randrange(len(p_))] = "N"
                parts.append("".join(p_))
                parts.append(unknown(rng.choice([2, 5, 6, 7, 9, 15])))
            r = "".join(parts[:-1])
        elif kind == 5:    # nothing to find: only unknown letters, or stretches below a word
            r = unknown(rng.choice([1, 5, 6, 40, 300])) if rng.random() < 0.5 else (src[0:20] + unknown(8) + src[40:67] + unknown(6) + src[100:110])
        else:              # a plain read, and a long one (lazy class) with a run
            if rng.random() < 0.5:
                r = src[100:250]
            else:
                r = src[0:400] + unknown(50) + src[450:900]
        reads.append(">q%d\n%s\n" % (i, r))
    rd = tmp_path / "reads.fa"
    rd.write_text("".join(reads))
    want = tmp_path / "want.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want),
                    "-num_threads", "8"], timeout=900)[0] == 0
    assert len(want.read_bytes()) > 50000
    assert _blast_text(pg, db, rd, tmp_path, "pieces") == want.read_bytes()
    monkeypatch.setenv("PGX_NO_PIECES", "1")
    assert _blast_text(pg, db, rd, tmp_path, "whole") == want.read_bytes()


@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("PGX_TIER2_SEEDS", "71,72").split(",")])
def test_reads_of_350_to_500_bases_with_19_to_40_differences_a_side(pg, oracle_bin, tmp_path, seed):
    """The second tier of the lane-per-HSP kernel (rows for 40 differences a side, the X-drop history from 19 on): reads of
    350-500 bases, 6-10 % away from their sources, with indels: most sides need more than 18 differences."""
    import random
    from pangea_plus_amd import _capi
    rng = random.Random(seed)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    anc = "".join(rng.choice("ACGT") for _ in range(1600))
    seqs = []
    for i in range(40):
        s = list(anc)
        for p_ in rng.sample(range(len(s)), 20 + (i % 4) * 20):
            s[p_] = rng.choice("ACGT")
        seqs.append("".join(s))
    db = tmp_path / "fam.fa"
    db.write_text("".join(">gi|%d|x|s%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(seqs)))
    reads = []
    for i in range(300):
        L = rng.choice([350, 450, 500, 512])
        s = rng.choice(seqs)
        o = rng.randrange(0, len(s) - L)
        w = list(s[o:o + L])
        for p_ in rng.sample(range(len(w)), int(len(w) * rng.choice([0.05, 0.07, 0.09]))):
            w[p_] = rng.choice("ACGT")
        for _ in range(rng.choice([0, 1, 2, 4])):
            p_ = rng.randrange(40, len(w) - 40)
            if rng.random() < 0.5:
                del w[p_:p_ + rng.choice([1, 2, 3])]
            else:
                w[p_:p_] = [rng.choice("ACGT") for _ in range(rng.choice([1, 2, 3]))]
        w = "".join(w)[:512]
        if i % 2:
            w = "".join(comp[c] for c in reversed(w))
        reads.append(">t%d\n%s\n" % (i, w))
    rd = tmp_path / "tier2_reads.fa"
    rd.write_text("".join(reads))
    want = tmp_path / "tier2_oracle.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want), "-num_threads", "8"],
                   timeout=900)[0] == 0
    rows = want.read_bytes().splitlines()
    assert len(rows) > 5000 and max(int(r.split(b"\t")[4]) for r in rows) >= 38   # mismatches of a row: both sides deep
    assert _blast_text(pg, db, rd, tmp_path, "tier2") == want.read_bytes()
    assert 1000 < _capi.stage_times().gapped_wide <= len(rows) + 1000   # the first tier listed them, each counted once


def test_long_queries_that_overhang_the_first_and_the_last_subject(pg, oracle_bin, tmp_path):
    """1 500-base queries whose only match is a short subject at the very start / the very end of the database: the window of
    letters the wide gapped kernel stages (read length + 2 x 62 + 64 letters around the anchor) would begin ~1 400 letters
    before the packed words, or end that far behind them -- more than their 768 letters of padding (gapped.hip
    staged such windows; those HSPs now read their letters in memory).  Both strands, both ends."""
    import random
    rng = random.Random(77)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))  # noqa: E731
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda s: "".join(comp[c] for c in reversed(s))  # noqa: E731
    first, last = rnd(90), rnd(90)
    middle = [rnd(rng.choice([300, 700, 1600])) for _ in range(40)]
    seqs = [first] + middle + [last]
    db = tmp_path / "ends.fa"
    db.write_text("".join(">gi|%d|x|e%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(seqs)))
    def with_errors(s):
        w = list(s)
        w[30] = comp[w[30]]              # a substitution
        del w[55]                        # and a deletion: both sides of the anchor do gapped work
        return "".join(w)
    reads = [
        rnd(1410) + with_errors(first),        # the match is the query's tail, on the first subject: window starts far in front
        with_errors(last) + rnd(1410),         # the match is the query's head, on the last subject: window ends far behind
        rc(rnd(1410) + with_errors(last)),     # the same on the other strand
        rc(with_errors(first) + rnd(1410)),
        middle[3][100:250],                    # and ordinary reads beside them
        rc(middle[7][0:150]),
    ]
    rd = tmp_path / "ends_reads.fa"
    rd.write_text("".join(">o%d\n%s\n" % (i, s) for i, s in enumerate(reads)))
    want = tmp_path / "ends_oracle.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want), "-num_threads", "4"],
                   timeout=600)[0] == 0
    rows = want.read_text().splitlines()
    assert {r.split("\t")[0] for r in rows} >= {"o0", "o1", "o2", "o3", "o4", "o5"}
    assert any(r.split("\t")[5] != "0" for r in rows if r.startswith(("o0", "o1", "o2", "o3")))  # gapped rows among them
    assert _blast_text(pg, db, rd, tmp_path, "ends") == want.read_bytes()


def test_long_queries_whose_gaps_drift_off_the_diagonal_lanes(pg, oracle_bin, tmp_path):
    """The one-lane-per-diagonal kernel (k_gapped_diag) holds diagonals -32 .. 31 of the anchor's and six-bit counts of gap
    columns / openings.  Queries of 900-1 600 bases whose gaps all lean one way -- five to nine deletions (or insertions) of
    6-9 bases, 120+ matching bases apart, so that each is crossed -- drift 40-70 diagonals and hold up to ~80 gap columns:
    those HSPs must come out of the LDS-row kernels behind it, and the table must still be the oracle's."""
    import random
    rng = random.Random(4242)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))  # noqa: E731
    rc = lambda s: "".join(comp[c] for c in reversed(s))  # noqa: E731
    subjects = [rnd(rng.choice([1700, 2000])) for _ in range(24)]
    db = tmp_path / "drift.fa"
    db.write_text("".join(">gi|%d|x|d%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(subjects)))
    reads = []
    for i in range(96):
        s = subjects[i % len(subjects)]
        L = rng.choice([900, 1200, 1600])
        o = rng.randrange(0, len(s) - L)
        w = list(s[o:o + L])
        n_gaps = rng.choice([5, 6, 7, 8, 9])
        kind = i % 3                      # 0: deletions only, 1: insertions only, 2: alternating (no drift: stays with the lanes)
        step = (L - 100) // (n_gaps + 1)
        for g in range(n_gaps, 0, -1):    # from the far end, so earlier positions stay valid
            p_ = 50 + g * step + rng.randrange(-10, 10)
            n = rng.choice([6, 7, 8, 9])
            if kind == 0 or (kind == 2 and g % 2):
                del w[p_:p_ + n]
            else:
                w[p_:p_] = list(rnd(n))
        for p_ in rng.sample(range(len(w)), len(w) // 60):
            w[p_] = rng.choice("ACGT")
        w = "".join(w)
        reads.append(">q%d\n%s\n" % (i, rc(w) if i % 2 else w))
    rd = tmp_path / "drift_reads.fa"
    rd.write_text("".join(reads))
    want = tmp_path / "drift_oracle.tsv"
    assert run_cmd([oracle_bin, "blastn", "-query", str(rd), "-db", str(db), "-outfmt", "6", "-out", str(want), "-num_threads", "8"],
                   timeout=900)[0] == 0
    rows = [r.split(b"\t") for r in want.read_bytes().splitlines()]
    # rows that cross many gaps one way: alignment lengths on query and subject differ by the drift
    drift = [abs((int(r[7]) - int(r[6])) - abs(int(r[9]) - int(r[8]))) for r in rows]
    assert max(drift) >= 40 and sum(d >= 33 for d in drift) >= 10
    assert max(int(r[5]) for r in rows) >= 5
    assert _blast_text(pg, db, rd, tmp_path, "drift") == want.read_bytes()
