"""Timing aid: -outfmt 6 rendering of a batch's hit table (device formatter) and the blastn command line end to end."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
cfg = pg.SynthCfg.default()
db = pg.Db.from_synth(cfg)
n = int(os.environ.get("N", "2000000"))
reads = pg.Reads.from_synth(cfg, 0, n)
hits = _capi.blast_search(db, reads)
for it in range(2):
    t0 = time.time(); text = hits.format(db, reads); t1 = time.time()
    print("format %d rows: %.2f s, %.0f MB -> %.1f M rows/s" % (text.count(b"\n"), t1 - t0, len(text) / 1e6, text.count(b"\n") / (t1 - t0) / 1e6), flush=True)
import ctypes as C, tempfile
d = tempfile.mkdtemp()
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), d.encode()))
pg.TaxDb.create(d); db.bind_taxonomy(pg.TaxDb.open(d))
rdp = pg.Rdp.from_synth(cfg, 0, n, db)
h2, recs = _capi.classify_consensus(db, reads, rdp)
for it in range(2):
    t0 = time.time(); text = _capi.consensus_format(db, reads, h2, recs); t1 = time.time()
    print("consensus text for %d reads: %.2f s, %.0f MB -> %.1f M reads/s" % (n, t1 - t0, len(text) / 1e6, n / (t1 - t0) / 1e6), flush=True)
