"""Timing aid: the whole chain on the 16S-like worst case (every read hits thousands of subjects; 500-subject cut,
one-lane-per-read consensus for the reads with more than 64 hits)."""
import os, sys, time, ctypes as C, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
cfg = pg.SynthCfg.default(n_seq=20000, seq_len=1500, n_genus=1, read_len=150)
d = tempfile.mkdtemp()
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), d.encode()))
pg.TaxDb.create(d)
db = pg.Db.from_synth(cfg); db.bind_taxonomy(pg.TaxDb.open(d))
for n in (10000, 100000):
    reads = pg.Reads.from_synth(cfg, 0, n); rdp = pg.Rdp.from_synth(cfg, 0, n, db)
    for it in range(2):
        t0 = time.perf_counter()
        hits, recs = _capi.classify_consensus(db, reads, rdp)
        dt = time.perf_counter() - t0
        st = _capi.stage_times()
        kept = int((recs["hit"] >= 0).sum())
        print("reads=%d slots=%d wall %.3f s: seed %.1f group %.1f sort+consensus %.1f ms; %d reads with a winner" % (
            n, len(hits), dt, st.seed_extend_ms, st.group_ms, st.sort_ms + st.consensus_ms, kept), flush=True)
        del hits
