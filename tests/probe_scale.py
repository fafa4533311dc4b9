"""Scratch probe (not a test): full-size synthetic DB build + a search batch, prints timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
cfg = pg.SynthCfg.default()
t0 = time.time(); db = pg.Db.from_synth(cfg); t1 = time.time()
print("db build+index: %.2fs shape=%s" % (t1 - t0, db.shape()), flush=True)
reads = pg.Reads.from_synth(cfg, 0, n_reads); t2 = time.time()
print("reads: %.2fs" % (t2 - t1), flush=True)
for it in range(3):
    t = time.time(); hits = _capi.blast_search(db, reads); dt = time.time() - t
    st = _capi.stage_times()
    print("search %d: wall %.3fs hits=%d seed_extend=%.1fms group=%.1fms sort=%.1fms total=%.1fms probes=%d postings=%d -> %.2f Mreads/s" % (
        it, dt, len(hits), st.seed_extend_ms, st.group_ms, st.sort_ms, st.total_ms, st.probes, st.postings, n_reads / st.total_ms / 1e3), flush=True)
    del hits
