"""Timing aid: the search stage for other read lengths (the two-reads-per-wavefront fast path holds reads of <= 192 bases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
n = 2_000_000
base = pg.SynthCfg.default()
db = pg.Db.from_synth(base)
for L in [int(x) for x in os.environ.get('LENS', '100,150,192,200,250,300,400,500,600').split(',')]:
    cfg = pg.SynthCfg.default(read_len=L)
    reads = pg.Reads.from_synth(cfg, 0, n)
    for it in range(2):
        h = _capi.blast_search(db, reads); st = _capi.stage_times(); k = len(h); del h
    print("read_len=%d: seed %.1f ms sort %.1f ms total %.1f ms -> %.1f M reads/s, %.1f hits/read" % (
        L, st.seed_extend_ms, st.sort_ms, st.total_ms, n / st.total_ms / 1e3, k / n), flush=True)
