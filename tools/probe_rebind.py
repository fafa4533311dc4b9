"""Debugging aid: does re-binding a taxonomy or re-creating batches change the stage times?"""
import os, sys, ctypes as C, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
cfg = pg.SynthCfg.default()
n = 2_000_000
db = pg.Db.from_synth(cfg)
d = tempfile.mkdtemp()
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), d.encode()))
pg.TaxDb.create(d)
tax = pg.TaxDb.open(d)
def run(tag, reads, rdp, k=3):
    for it in range(k):
        _capi.classify_consensus(db, reads, rdp, want_records=False, want_hits=False); st = _capi.stage_times()
        print("%s: seed %.1f group %.2f sort %.1f total %.1f ms; hits %d postings %d survivors %d candidates %d" % (tag, st.seed_extend_ms, st.group_ms, st.sort_ms, st.total_ms, st.hits, st.postings, st.survivors, st.candidates), flush=True)
db.bind_taxonomy(tax)
reads = pg.Reads.from_synth(cfg, 0, n); rdp = pg.Rdp.from_synth(cfg, 0, n, db)
run("first bind", reads, rdp)
reads2 = pg.Reads.from_synth(cfg, n, n); rdp2 = pg.Rdp.from_synth(cfg, n, n, db)
run("new batch, same bind", reads2, rdp2)
db.bind_taxonomy(tax)
run("re-bind (same taxonomy object), old batch", reads, rdp)
rdp3 = pg.Rdp.from_synth(cfg, 0, n, db)
run("re-bind, new rdp", reads, rdp3)
import time
time.sleep(3.0)
run("after 3 s of idle, nothing re-bound", reads, rdp3, k=6)
