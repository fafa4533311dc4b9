"""Timing aid: a 16S-like worst case: every read hits (nearly) every subject of a 20 000-sequence family."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
cfg = pg.SynthCfg.default(n_seq=20000, seq_len=1500, n_genus=1, read_len=150)
db = pg.Db.from_synth(cfg)
for n in (1000, 10000):
    reads = pg.Reads.from_synth(cfg, 0, n)
    for it in range(2):
        t0 = time.perf_counter()
        h = _capi.blast_search(db, reads)
        dt = time.perf_counter() - t0
        st = _capi.stage_times()
        cnt = h.read_offsets(n)
        print("reads=%d hits=%d (%.0f per read) wall %.3f s: seed %.1f ms group %.1f ms sort %.1f ms" % (
            n, len(h), len(h) / n, dt, st.seed_extend_ms, st.group_ms, st.sort_ms), flush=True)
        del h
