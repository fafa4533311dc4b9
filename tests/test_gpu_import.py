"""GPU: the path a non-root rank takes in bench.py — allocate an empty database of the same shape, receive
the device arrays (here a device-to-device copy through torch tensors that alias library-owned HBM, plus
a 1-rank RCCL broadcast), finish the import — gives a database that answers identically."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_imported_database_answers_identically():
    import torch
    import torch.distributed as dist
    import pangea_plus_amd as pg
    from pangea_plus_amd import _capi
    assert torch.cuda.is_available() and torch.cuda.device_count() >= 1
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda:0")
    pg.init(0)
    cfg = pg.SynthCfg.default(n_seq=1500, seq_len=400, n_genus=40)
    src = pg.Db.from_synth(cfg)
    dst = pg.Db.alloc_like(src.shape())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for (na, va), (nb, vb) in zip(src.device_arrays(), dst.device_arrays()):
            assert na == nb
            ta = torch.as_tensor(va, device="cuda:0")
            tb = torch.as_tensor(vb, device="cuda:0")
            assert ta.data_ptr() == va.__cuda_array_interface__["data"][0]  # zero-copy alias
            dist.broadcast(ta, src=0)                                        # RCCL call on library-owned memory
            tb.copy_(ta)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    dst.finish_import()
    reads = pg.Reads.from_synth(cfg, 0, 800)
    a = _capi.blast_search(src, reads)
    b = _capi.blast_search(dst, reads)
    assert len(a) > 1000
    assert a.format(src, reads) == b.format(dst, reads)


def test_fasta_is_split_on_the_device_like_the_host_rules(tmp_path):
    """Read files are uploaded as text and split by kernels (lines, records, letters, names): CRLF, blank lines,
    wrapped sequences, blanks inside sequence lines, text in front of the first record, empty records, no final
    newline, header words after the name, windows (first, count)."""
    import random
    import pangea_plus_amd as pg
    pg.init(0)
    rng = random.Random(5)
    recs, text = [], ["junk before any record\n", "ACGT\n"]
    for i in range(300):
        L = rng.choice([0, 1, 31, 32, 33, 64, 150, 150, 150, 400])
        seq = "".join(rng.choice("ACGTNacgtRY") for _ in range(L))
        name = "read%d" % i
        eol = "\r\n" if i % 3 == 0 else "\n"
        text.append(">" + name + (" extra words\there" if i % 4 == 0 else "") + eol)
        body = seq
        if i % 5 == 0 and L > 40:      # wrapped at 60 with blanks and tabs sprinkled in
            parts = [body[k:k + 60] for k in range(0, L, 60)]
            body = eol.join(p[:10] + " " + p[10:20] + "\t" + p[20:] for p in parts)
        text.append(body + eol)
        if i % 7 == 0:
            text.append(eol)
        recs.append((name, seq))
    blob = "".join(text)
    blob = blob[:-1] if blob.endswith("\n") else blob   # no final newline
    fa = tmp_path / "messy.fa"
    fa.write_bytes(blob.encode())
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    reads = pg.Reads.from_fasta(str(fa))
    assert len(reads) == len(recs)
    lib = pg.lib()
    for i, (name, seq) in enumerate(recs):
        got = reads.get(i)
        want = [code.get(c.upper(), 4) for c in seq]
        assert list(got) == want, (i, name)
    window = pg.Reads.from_fasta(str(fa), 17, 40)
    assert len(window) == 40
    assert list(window.get(0)) == [code.get(c.upper(), 4) for c in recs[17][1]]
    # names travel with the batch: they appear in the formatted hit table of a search against the reads themselves
    db_fa = tmp_path / "db.fa"
    db_fa.write_text("".join(">gi|%d|x|%s|\n%s\n" % (i + 1, n_, s) for i, (n_, s) in enumerate(recs) if len(s) >= 150 and "N" not in s.upper()
                             and "R" not in s.upper() and "Y" not in s.upper()) or ">gi|1|x|none|\nACGT\n")
    db = pg.Db.from_fasta(str(db_fa))
    from pangea_plus_amd import _capi
    hits = _capi.blast_search(db, reads)
    rows = hits.format(db, reads).decode().splitlines()
    names_hit = {r.split("\t")[0] for r in rows}
    assert names_hit <= {n_ for n_, _ in recs}


def test_damaged_database_file_and_bad_rank_are_errors_not_crashes(tmp_path):
    """A .pgxdb whose header does not fit the file returns PGX_E_FORMAT (it used to end in std::bad_alloc across the C ABI),
    and `blastn -rank` outside [0, world_size) is refused instead of writing an empty table."""
    import struct
    import pangea_plus_amd as pg
    from pangea_plus_amd import _capi
    pg.init(0)
    fa = tmp_path / "d.fa"
    fa.write_text(">gi|1|x|a|\n" + "ACGT" * 100 + "\n>gi|2|x|b|\n" + "GATTACA" * 40 + "\n")
    pg.makeblastdb(str(fa), str(tmp_path / "d"))
    good = (tmp_path / "d.pgxdb").read_bytes()
    pg.Db.open(str(tmp_path / "d")).close()
    for field, value in ((0, -5), (0, 1 << 40), (1, -1), (1, 1 << 50), (3, 1 << 45), (3, 0)):
        hdr = list(struct.unpack("<4q", good[8:40]))
        hdr[field] = value
        (tmp_path / "bad.pgxdb").write_bytes(good[:8] + struct.pack("<4q", *hdr) + good[40:])
        with pytest.raises(_capi.PangeaError) as e:
            pg.Db.open(str(tmp_path / "bad"))
        assert e.value.status == -4, (field, value)
    (tmp_path / "bad.pgxdb").write_bytes(good[:60])
    with pytest.raises(_capi.PangeaError):
        pg.Db.open(str(tmp_path / "bad"))
    q = tmp_path / "q.fa"
    q.write_text(">r\n" + "ACGT" * 40 + "\n")
    for rk, ws in ((2, 2), (-1, 2), (5, 3)):
        with pytest.raises(_capi.PangeaError) as e:
            pg.blastn(str(q), str(tmp_path / "d"), str(tmp_path / "o.tsv"), rank=rk, world_size=ws)
        assert e.value.status == -1


@pytest.mark.parametrize("target_bytes", [(1 << 20) - 4096, (1 << 20) + 4096, (8 << 20) + 17, 24 << 20, (70 << 20) + 12345])
def test_fasta_text_sizes_around_the_copy_ring_stripes(target_bytes, tmp_path):
    """Host copies of 1 MB and more go through the process's ring of pinned 8 MB buffers (common.hip: staged_upload /
    staged_download): texts just below and above the threshold, one byte into a second stripe, whole stripes, and more
    stripes than the ring has buffers (every worker reuses both of its buffers).  The batch must hold exactly the file's reads:
    count, every length, and the letters of reads at the start, at every stripe boundary and at the end; the FASTA written
    back from HBM must be the text that went in."""
    import numpy as np
    import pangea_plus_amd as pg
    pg.init(0)
    rng = np.random.default_rng(target_bytes)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs, size, i = [], 0, 0
    while size < target_bytes:
        L = int(rng.integers(60, 180))
        rec = b">r%d\n" % i + letters[rng.integers(0, 4, L)].tobytes() + b"\n"
        recs.append(rec)
        size += len(rec)
        i += 1
    text = b"".join(recs)
    reads = pg.Reads.from_fasta_text(text)
    assert len(reads) == len(recs)
    offs = np.cumsum([0] + [len(r) for r in recs])
    picks = {0, 1, len(recs) - 1, len(recs) - 2}
    for edge in range(8 << 20, len(text), 8 << 20):      # the reads that straddle a stripe boundary, and their neighbours
        k = int(np.searchsorted(offs, edge)) - 1
        picks |= {max(k - 1, 0), k, min(k + 1, len(recs) - 1)}
    for k in sorted(picks):
        want = recs[k].split(b"\n")[1]
        got = bytes(b"ACGTN"[c] for c in reads.get(k))
        assert got == want, k
    out = tmp_path / "back.fa"
    reads.write_fasta(str(out))
    assert out.read_bytes() == text
