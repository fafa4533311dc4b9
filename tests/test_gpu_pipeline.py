"""GPU parity, the whole hot path: classify -> lineage -> consensus fused on the device, against the
oracle chain (blastn -> taxcollector -> consensus through files) on the same seeded workload."""
import ctypes
import os

import numpy as np
import pytest

from conftest import run_cmd

pytestmark = pytest.mark.gpu

SHAPE = dict(n_seq=3000, seq_len=500, n_genus=60, read_len=150)
ARGS = ["--n-seq", "3000", "--seq-len", "500", "--n-genus", "60", "--read-len", "150"]
N = 4000


@pytest.fixture(scope="module")
def pg():
    import pangea_plus_amd as pg
    pg.init(0)
    return pg


@pytest.fixture(scope="module")
def chain(tmp_path_factory, oracle_bin):
    d = tmp_path_factory.mktemp("chain")
    (d / "Tax_class").mkdir()
    assert run_cmd([oracle_bin, "synth", "db", "--out", str(d / "db.fa")] + ARGS)[0] == 0
    assert run_cmd([oracle_bin, "synth", "reads", "--out", str(d / "reads.fa"), "--count", str(N)] + ARGS)[0] == 0
    assert run_cmd([oracle_bin, "synth", "rdp", "--out", str(d / "rdp.tsv"), "--count", str(N)] + ARGS)[0] == 0
    assert run_cmd([oracle_bin, "synth", "taxdump", "--out", str(d / "Tax_class")] + ARGS)[0] == 0
    assert run_cmd([oracle_bin, "tax_class", "-c"], cwd=d / "Tax_class")[0] == 0
    assert run_cmd([oracle_bin, "blastn", "-query", str(d / "reads.fa"), "-db", str(d / "db.fa"), "-outfmt", "6", "-out",
                    str(d / "hits.tsv"), "-num_threads", "8"], timeout=600)[0] == 0
    assert run_cmd([oracle_bin, "taxcollector", "-f", str(d / "hits.tsv"), "-o", str(d / "hits_class.tsv"), "-d",
                    str(d / "Tax_class")], timeout=600)[0] == 0
    assert run_cmd([oracle_bin, "consensus", "-b", str(d / "hits_class.tsv"), "-r", str(d / "rdp.tsv"), "-o",
                    str(d / "consensus.txt")], timeout=600)[0] == 0
    return d


def test_fused_pipeline_equals_oracle_chain(pg, chain):
    from pangea_plus_amd import _capi
    cfg = pg.SynthCfg.default(**SHAPE)
    db = pg.Db.from_synth(cfg)
    tax = pg.TaxDb.open(str(chain / "Tax_class"))
    db.bind_taxonomy(tax)
    assert db.subject_lineage(0) == "[0]Domaaaaa;[1]Phyaaaaa;[2]Clsaaaaa;[3]Ordaaaaa;[4]Famaaaaa;[5]Genaaaaa;[6]Genaaaaa_spaaaaa;"
    reads = pg.Reads.from_synth(cfg, 0, N)
    rdp = pg.Rdp.from_synth(cfg, 0, N, db)
    hits, recs = _capi.classify_consensus(db, reads, rdp)
    assert hits.format(db, reads) == (chain / "hits.tsv").read_bytes()
    text = _capi.consensus_format(db, reads, hits, recs)
    # the same text straight into a file (pgx_consensus_format_file: rendered and written piece by piece)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        assert _capi.consensus_format_file(db, reads, hits, recs, td + "/c.txt") == len(text)
        assert open(td + "/c.txt", "rb").read() == text
    want = (chain / "consensus.txt").read_bytes()
    assert len(want) > 100000
    assert text == want
    # the same through the file-format RDP stream, and consensus over an existing hit table
    rdp2 = pg.Rdp.from_file(str(chain / "rdp.tsv"), reads, db)
    recs2 = np.zeros(N, dtype=_capi.REC_DTYPE)
    _capi._check(pg.lib().pgx_consensus_batch(db.ptr, hits.ptr, rdp2.ptr, recs2.ctypes.data, N))
    assert (recs2 == recs).all()
    # reads handed over as a FASTA file: names come from the file's text, tables are rendered on the device
    freads = pg.Reads.from_fasta(str(chain / "reads.fa"))
    frdp = pg.Rdp.from_file(str(chain / "rdp.tsv"), freads, db)
    fhits, frecs = _capi.classify_consensus(db, freads, frdp)
    assert fhits.format(db, freads) == (chain / "hits.tsv").read_bytes()
    assert _capi.consensus_format(db, freads, fhits, frecs) == want


def test_file_verbs_chain_equals_oracle_chain(pg, chain, tmp_path):
    # the three reference command lines, run one after the other on files
    pg.makeblastdb(str(chain / "db.fa"), str(tmp_path / "db"))
    pg.blastn(str(chain / "reads.fa"), str(tmp_path / "db"), str(tmp_path / "hits.tsv"))
    assert (tmp_path / "hits.tsv").read_bytes() == (chain / "hits.tsv").read_bytes()
    pg.taxcollector(str(tmp_path / "hits.tsv"), str(tmp_path / "hits_class.tsv"), taxdir=str(chain / "Tax_class"))
    assert (tmp_path / "hits_class.tsv").read_bytes() == (chain / "hits_class.tsv").read_bytes()
    pg.consensus(str(tmp_path / "hits_class.tsv"), str(chain / "rdp.tsv"), str(tmp_path / "consensus.txt"))
    assert (tmp_path / "consensus.txt").read_bytes() == (chain / "consensus.txt").read_bytes()


def test_pident_rounding_equals_printf(pg):
    # the device orders hits by the TEXT of pident; its hundredths must be what printf("%.2f") prints
    lib = ctypes.CDLL(None)
    bad = 0
    for L in list(range(28, 400)) + [1000, 1400, 1999]:
        for m in range(max(1, L - 60), L + 1):
            want = int(round(float("%.2f" % (100.0 * m / L)) * 100))
            q, r = divmod(10000 * m, L)
            got = q + 1 if 2 * r > L else q
            if 2 * r == L:
                d = 100.0 * m / L
                import math
                s = math.fma(d, 200.0, -(2 * q + 1)) if hasattr(math, "fma") else (d * 200.0 - (2 * q + 1))
                got = q + 1 if s > 0 else (q if s < 0 else q + (q & 1))
            bad += got != want
    assert bad == 0


@pytest.mark.parametrize("variant", [int(x) for x in os.environ.get("PGX_RAGGED_VARIANTS", "0,1,2,3").split(",")])
def test_ragged_lineages_and_repeated_rdp_triplets(pg, chain, tmp_path, oracle_bin, variant):
    """Lineages of different depth inside one read's hits and agreement counts of two digits: the
    order-dependent selection of Consensus:186-204 has no closed form there (k_sort_consensus's literal walk)
    and Perl's text comparison of the counts ("10" lt "9") decides."""
    from pangea_plus_amd import _capi
    tdir = tmp_path / "Tax_class"
    tdir.mkdir()
    for f in ("names.dmp", "gi_taxid_nucl.dmp"):
        (tdir / f).write_bytes((chain / "Tax_class" / f).read_bytes())
    out = []
    for line in (chain / "Tax_class" / "nodes.dmp").read_text().splitlines(True):
        cols = line.split("\t|\t")
        t = int(cols[0])
        if variant == 0:
            if cols[2] == "species" and t % 3 == 0:
                cols[2] = "no rank"          # every third species loses its rank: shorter lineage text
            if cols[2] == "genus" and t % 5 == 0:
                cols[2] = "subgenus"
        elif variant == 1:
            # depth varies at every level, so that a read's hits (one genus family) mix three or four depths
            if cols[2] == "species" and t % 2 == 0:
                cols[2] = "no rank"
            if cols[2] == "family" and t % 3 == 0:
                cols[2] = "no rank"
            if cols[2] == "phylum" and t % 2 == 1:
                cols[2] = "superphylum"
        elif variant == 2:
            # most species unranked, a few genera too: the top hit is often the shorter lineage
            if cols[2] == "species" and t % 7 != 0:
                cols[2] = "no rank"
            if cols[2] == "genus" and t % 4 == 0:
                cols[2] = "no rank"
        else:
            # seeded random: any ranked node below the domain loses or changes its rank with its own probability
            import random
            rr = random.Random(variant * 1000003 + t)
            if cols[2] in ("phylum", "class", "order", "family", "genus", "species") and rr.random() < 0.1 * (1 + variant % 5):
                cols[2] = rr.choice(["no rank", "no rank", "subgenus", "tribe", "species group"])
        out.append("\t|\t".join(cols))
    (tdir / "nodes.dmp").write_text("".join(out))
    assert run_cmd([oracle_bin, "tax_class", "-c"], cwd=tdir)[0] == 0
    rows = (chain / "rdp.tsv").read_text().splitlines()
    for i in range(variant % 4, len(rows), 4 - variant % 3):     # some reads name their triplets three times over
        head, trip = rows[i].split("\t", 5)[:5], rows[i].split("\t", 5)[5]
        rows[i] = "\t".join(head + [trip, trip, trip])
    for i in range(1, len(rows), 5):     # and some agree with nothing: every hit of the read counts 0 matches
        head, trip = rows[i].split("\t", 5)[:5], rows[i].split("\t", 5)[5].split("\t")
        for k in range(0, len(trip), 3):
            trip[k] = "Zzz" + trip[k][::-1]
        rows[i] = "\t".join(head + trip)
    (tmp_path / "rdp.tsv").write_text("\n".join(rows) + "\n")
    assert run_cmd([oracle_bin, "taxcollector", "-f", str(chain / "hits.tsv"), "-o", str(tmp_path / "hits_class.tsv"),
                    "-d", str(tdir)], timeout=600)[0] == 0
    assert run_cmd([oracle_bin, "consensus", "-b", str(tmp_path / "hits_class.tsv"), "-r", str(tmp_path / "rdp.tsv"),
                    "-o", str(tmp_path / "consensus.txt")], timeout=600)[0] == 0
    want = (tmp_path / "consensus.txt").read_bytes()
    assert want != (chain / "consensus.txt").read_bytes()

    cfg = pg.SynthCfg.default(**SHAPE)
    db = pg.Db.from_synth(cfg)
    db.bind_taxonomy(pg.TaxDb.open(str(tdir)))
    reads = pg.Reads.from_synth(cfg, 0, N)
    rdp = pg.Rdp.from_file(str(tmp_path / "rdp.tsv"), reads, db)
    hits, recs = _capi.classify_consensus(db, reads, rdp)
    assert _capi.consensus_format(db, reads, hits, recs) == want
    if variant < 3:   # the hand-made variants are built to reach two-digit counts; the random ones need not
        assert int((recs["matches"] >= 10).sum()) > 0
    assert int(((recs["matches"] == 0) & (recs["hit"] >= 0)).sum()) > 100
    # and the file verb on the same tables
    pg.consensus(str(tmp_path / "hits_class.tsv"), str(tmp_path / "rdp.tsv"), str(tmp_path / "c2.txt"))
    assert (tmp_path / "c2.txt").read_bytes() == want


def test_mate_joined_reads_through_the_fused_pipeline(pg, chain, tmp_path, oracle_bin):
    """The shape Trim hands over (two mates joined by 100 N's): searched in pieces, ordered and voted on per joined
    read; table and consensus text equal to the oracle chain run on the same FASTA."""
    from pangea_plus_amd import _capi
    lines = (chain / "reads.fa").read_text().split("\n")
    names, seqs = lines[0::2], lines[1::2]
    joined, keep = [], set()
    for i in range(0, 1200, 2):
        joined.append("%s\n%s%s%s\n" % (names[i], seqs[i][:130], "N" * 100, seqs[i + 1][:130]))
        keep.add(names[i][1:].split()[0])
    (tmp_path / "joined.fa").write_text("".join(joined))
    rdp_lines = [l for l in (chain / "rdp.tsv").read_text().splitlines(True) if l.split("\t")[0] in keep]
    assert len(rdp_lines) == len(keep)
    (tmp_path / "rdp.tsv").write_text("".join(rdp_lines))
    assert run_cmd([oracle_bin, "blastn", "-query", str(tmp_path / "joined.fa"), "-db", str(chain / "db.fa"), "-outfmt", "6", "-out",
                    str(tmp_path / "hits.tsv"), "-num_threads", "8"], timeout=600)[0] == 0
    assert run_cmd([oracle_bin, "taxcollector", "-f", str(tmp_path / "hits.tsv"), "-o", str(tmp_path / "hits_class.tsv"), "-d",
                    str(chain / "Tax_class")], timeout=600)[0] == 0
    assert run_cmd([oracle_bin, "consensus", "-b", str(tmp_path / "hits_class.tsv"), "-r", str(tmp_path / "rdp.tsv"), "-o",
                    str(tmp_path / "consensus.txt")], timeout=600)[0] == 0
    cfg = pg.SynthCfg.default(**SHAPE)
    db = pg.Db.from_synth(cfg)
    db.bind_taxonomy(pg.TaxDb.open(str(chain / "Tax_class")))
    reads = pg.Reads.from_fasta(str(tmp_path / "joined.fa"))
    rdp = pg.Rdp.from_file(str(tmp_path / "rdp.tsv"), reads, db)
    hits, recs = _capi.classify_consensus(db, reads, rdp)
    want_hits = (tmp_path / "hits.tsv").read_bytes()
    assert len(want_hits) > 100000
    assert hits.format(db, reads) == want_hits
    assert _capi.consensus_format(db, reads, hits, recs) == (tmp_path / "consensus.txt").read_bytes()


def test_rdp_lines_of_other_reads_and_repeated_names(pg, chain, tmp_path):
    """An RDP file may hold lines of reads that are not in the batch (another shard of the run): they are skipped,
    one probe each.  A name carried by two reads takes the lines of that name in order."""
    from pangea_plus_amd import _capi
    lines = (chain / "reads.fa").read_text().split("\n")
    names, seqs = lines[0::2], lines[1::2]
    sub = "".join("%s\n%s\n" % (names[i], seqs[i]) for i in range(0, 2000, 2))
    sub += "%s\n%s\n" % (names[10], seqs[11])                 # the name of read 10 once more, other bases
    (tmp_path / "sub.fa").write_text(sub)
    rdp_all = (chain / "rdp.tsv").read_text().splitlines(True)
    by_name = {l.split("\t")[0]: l for l in rdp_all}
    own = [by_name[names[i][1:]] for i in range(0, 2000, 2)] + [by_name[names[11][1:]].replace(names[11][1:], names[10][1:], 1)]
    (tmp_path / "own.tsv").write_text("".join(own))
    # the whole stream (4000 lines, reads of other shards between the batch's own) plus the second line of the repeated name
    (tmp_path / "all.tsv").write_text("".join(rdp_all) + own[-1])
    cfg = pg.SynthCfg.default(**SHAPE)
    db = pg.Db.from_synth(cfg)
    db.bind_taxonomy(pg.TaxDb.open(str(chain / "Tax_class")))
    reads = pg.Reads.from_fasta(str(tmp_path / "sub.fa"))
    assert len(reads) == 1001
    got = []
    for f in ("own.tsv", "all.tsv"):
        rdp = pg.Rdp.from_file(str(tmp_path / f), reads, db)
        hits, recs = _capi.classify_consensus(db, reads, rdp)
        got.append((hits.format(db, reads), _capi.consensus_format(db, reads, hits, recs), recs.copy()))
    assert got[0][0] == got[1][0] and got[0][1] == got[1][1] and (got[0][2] == got[1][2]).all()
    assert len(got[0][1]) > 20000


def test_batch_written_as_files_reads_back_the_same(pg, chain, tmp_path):
    """pgx_reads_write_fasta / pgx_rdp_write_file: a batch made on the device, written as the FASTA and five-tab RDP text
    the reference's tools exchange, read back through the file entry points (RDP lines parsed on all host cores), gives
    the same consensus text."""
    from pangea_plus_amd import _capi
    cfg = pg.SynthCfg.default(**SHAPE)
    db = pg.Db.from_synth(cfg)
    db.bind_taxonomy(pg.TaxDb.open(str(chain / "Tax_class")))
    reads = pg.Reads.from_synth(cfg, 0, N)
    rdp = pg.Rdp.from_synth(cfg, 0, N, db)
    reads.write_fasta(str(tmp_path / "r.fa"))
    assert (tmp_path / "r.fa").read_bytes() == (chain / "reads.fa").read_bytes()
    rdp.write_file(str(tmp_path / "rdp.txt"), reads, db)
    hits, recs = _capi.classify_consensus(db, reads, rdp)
    want = _capi.consensus_format(db, reads, hits, recs)
    assert want == (chain / "consensus.txt").read_bytes()
    r2 = pg.Reads.from_fasta(str(tmp_path / "r.fa"))
    p2 = pg.Rdp.from_file(str(tmp_path / "rdp.txt"), r2, db)
    h2, c2 = _capi.classify_consensus(db, r2, p2)
    assert _capi.consensus_format(db, r2, h2, c2) == want
    # lines of reads that are not in the batch, a read without a line, repeated names: the cursor rule (Consensus:141-220)
    lines = (tmp_path / "rdp.txt").read_text().splitlines(True)
    mixed = ["zz_not_here\t\t\t\t\tBacteria\tdomain\t0.9\n"] + lines[:100] + lines[101:] + [lines[5]]
    (tmp_path / "rdp_mixed.txt").write_text("".join(mixed))
    p3 = pg.Rdp.from_file(str(tmp_path / "rdp_mixed.txt"), r2, db)
    h3, c3 = _capi.classify_consensus(db, r2, p3)
    assert (c3["hit"][100] == -2) and (np.delete(c3, 100) == np.delete(c2, 100)).all()


def test_rdp_text_in_odd_shapes_parses_as_the_oracle_reads_it(pg, chain, tmp_path, oracle_bin):
    """The RDP import (mapped file, lines and fields on all host cores) against the checker's sequential reading of the same
    text (oracle `consensus`, itself pinned by the reference Perl's goldens): no newline at the end of the file, lines
    without the five-tab group, seven tabs in a row, a second five-tab group, trailing tabs, a single-field tail.  Lines
    that name no read of the batch (the reference's walk never gets past one: "not found" to the end of the table) are
    this library's own rule -- skipped, one probe each -- and must leave the result as if they were not there."""
    from pangea_plus_amd import _capi
    rows = (chain / "rdp.tsv").read_text().splitlines()
    assert len(rows) >= 2000

    def build(foreign):
        out = []
        for i, r in enumerate(rows):
            rid, rest = r.split("\t\t\t\t\t", 1)
            f = rest.split("\t")
            k = i % 16
            if k == 1:
                out.append(rid)                                            # the id alone
            elif k == 2:
                out.append(rid + "\t\t\t\t\t\t\t" + rest)                  # seven tabs: two empty fields in front
            elif k == 3:
                out.append(rid + "\t\t\t\t\t" + rest + "\t\t\t\t\t" + rest)  # a second five-tab group ends the fields
            elif k == 4:
                out.append(rid + "\t\t\t\t\t" + rest + "\t\t\t")           # trailing tabs
            elif k == 6:
                out.append(rid + "\t\t\t\t\t" + "\t".join(f[:4]))          # the second triplet cut after its name
            elif k == 5 and foreign:
                out += ["", r]                                             # an empty line
            elif k == 7 and foreign:
                out += ["nobody_%d\t\t\t\t\t%s" % (i, rest), r]             # a read that does not exist
            elif k == 8 and foreign:
                out += [rid + "\tx\t\t\t\t\t" + rest, r]                   # a tab inside what stands before the five tabs
            elif k == 9 and foreign:
                out += [rid + "x\t\t\t\t" + rest, r]                       # four tabs only: the whole line is the id
            elif k == 10 and foreign and i > 40:
                out += [rows[i - 37].replace("genus", "class"), r]          # a read that lies behind the cursor: its line is skipped
            else:
                out.append(r)
        return "\n".join(out)                                             # and no newline at the very end

    (tmp_path / "odd.tsv").write_text(build(False))
    (tmp_path / "odd_foreign.tsv").write_text(build(True))
    assert run_cmd([oracle_bin, "consensus", "-b", str(chain / "hits_class.tsv"), "-r", str(tmp_path / "odd.tsv"),
                    "-o", str(tmp_path / "consensus.txt")], timeout=600)[0] == 0
    want = (tmp_path / "consensus.txt").read_bytes()
    assert want != (chain / "consensus.txt").read_bytes() and len(want) > 10000
    cfg = pg.SynthCfg.default(**SHAPE)
    db = pg.Db.from_synth(cfg)
    db.bind_taxonomy(pg.TaxDb.open(str(chain / "Tax_class")))
    reads = pg.Reads.from_fasta(str(chain / "reads.fa"))
    # both forms of the import: on the device (rdp_device.hip, the default for a batch made from a file) and on the host
    # cores (PGX_RDP_HOST=1, with 1, 3 and all threads); the table each makes is also written back as text and compared
    back = {}
    try:
        for threads in ("device", "1", "3", None):
            os.environ.pop("PGX_RDP_THREADS", None)
            os.environ.pop("PGX_RDP_HOST", None)
            if threads != "device":
                os.environ["PGX_RDP_HOST"] = "1"
                if threads:
                    os.environ["PGX_RDP_THREADS"] = threads
            for name in ("odd.tsv", "odd_foreign.tsv"):
                rdp = pg.Rdp.from_file(str(tmp_path / name), reads, db)
                hits, recs = _capi.classify_consensus(db, reads, rdp)
                assert _capi.consensus_format(db, reads, hits, recs) == want, (threads, name)
                rdp.write_file(str(tmp_path / "back.tsv"), reads, db)
                back.setdefault(name, (tmp_path / "back.tsv").read_bytes())
                assert back[name] == (tmp_path / "back.tsv").read_bytes() and len(back[name]) > 10000, (threads, name)
    finally:
        os.environ.pop("PGX_RDP_THREADS", None)
        os.environ.pop("PGX_RDP_HOST", None)
    # an empty file and a file of newlines: no read has a line
    for name, body in (("empty.tsv", ""), ("newlines.tsv", "\n\n\n")):
        (tmp_path / name).write_text(body)
        rdp = pg.Rdp.from_file(str(tmp_path / name), reads, db)
        hits, recs = _capi.classify_consensus(db, reads, rdp)
        assert (recs["hit"] == -2).all()


def test_rdp_import_on_the_device_and_on_the_host_agree_on_awkward_files(pg, chain, tmp_path):
    """The two forms of the RDP import (rdp_device.hip; the host cores, PGX_RDP_HOST=1) on files neither was written for:
    CRLF line ends, lines in reverse order (only the first survives the cursor rule), every line twice, a 20 KB id, lines that
    are five tabs and nothing else, names with quotes, digits and high bytes, a name field that is empty, fields behind a
    second five-tab group, and a batch in which two reads share a name (the device form hands such a batch to the host
    form).  The tables are compared as the text `pgx_rdp_write_file` makes of them and through the consensus records."""
    from pangea_plus_amd import _capi
    rows = (chain / "rdp.tsv").read_text().splitlines()[:1200]
    cfg = pg.SynthCfg.default(**SHAPE)
    db = pg.Db.from_synth(cfg)
    db.bind_taxonomy(pg.TaxDb.open(str(chain / "Tax_class")))
    reads = pg.Reads.from_fasta(str(chain / "reads.fa"))
    files = {}
    files["crlf"] = "\r\n".join(rows) + "\r\n"
    files["reverse"] = "\n".join(reversed(rows)) + "\n"
    files["twice"] = "\n".join(r for row in rows for r in (row, row)) + "\n"
    files["long_id"] = "x" * 20000 + "\t\t\t\t\tBacteria\tdomain\t1.0\n" + "\n".join(rows[:50]) + "\n" + "y" * 70000
    files["bare_tabs"] = "\t\t\t\t\t\n".join(rows[:300]) + "\n\t\t\t\t\t"
    odd = []
    for i, row in enumerate(rows[:600]):
        rid, rest = row.split("\t\t\t\t\t", 1)
        f = rest.split("\t")
        if i % 5 == 0:
            f[0] = '"' + f[0] + '" 16S_7'
        elif i % 5 == 1:
            f[0] = ""
        elif i % 5 == 2:
            f[3] = f[3] + "\xe9\xff"
        elif i % 5 == 3:
            f = f[:5]
        odd.append(rid + "\t\t\t\t\t" + "\t".join(f) + ("\t\t\t\t\tignored\tgenus\t0.1" if i % 7 == 0 else ""))
    files["odd_names"] = "\n".join(odd)
    for name, body in files.items():
        (tmp_path / (name + ".tsv")).write_bytes(body.encode("latin-1"))
    hits = _capi.blast_search(db, reads)

    def both(name, rd):
        out = []
        for form in ("device", "host"):
            os.environ.pop("PGX_RDP_HOST", None)
            if form == "host":
                os.environ["PGX_RDP_HOST"] = "1"
            try:
                rdp = pg.Rdp.from_file(str(tmp_path / (name + ".tsv")), rd, db)
            finally:
                os.environ.pop("PGX_RDP_HOST", None)
            rdp.write_file(str(tmp_path / "back.tsv"), rd, db)
            h2, recs = _capi.classify_consensus(db, rd, rdp)
            out.append(((tmp_path / "back.tsv").read_bytes(), recs.tobytes()))
        assert out[0] == out[1], name
        return out[0][0]
    texts = {name: both(name, reads) for name in files}
    assert texts["twice"] == texts["crlf"].replace(b"\r", b"") or len(texts["twice"]) > 1000
    assert texts["reverse"].count(b"\n") <= 2          # the cursor passes every read but the last one named first
    assert texts["bare_tabs"].count(b"\n") >= 300
    # two reads with one name: the batch goes to the host form, whichever form was asked for
    fa = (chain / "reads.fa").read_text().split(">")[1:1201]
    fa[7] = fa[3].split("\n", 1)[0] + "\n" + fa[7].split("\n", 1)[1]
    (tmp_path / "dup.fa").write_text("".join(">" + x for x in fa))
    dup = pg.Reads.from_fasta(str(tmp_path / "dup.fa"))
    (tmp_path / "dup.tsv").write_text("\n".join(rows) + "\n")
    assert both("dup", dup).count(b"\n") > 1000
