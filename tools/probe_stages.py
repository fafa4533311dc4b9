"""Profiling aid: stage-truncated timing of k_seed_extend (PGX_SEED_STOP) at full scale."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
cfg = pg.SynthCfg.default()
db = pg.Db.from_synth(cfg)
reads = pg.Reads.from_synth(cfg, 0, 10_000_000)
for stop in [int(x) for x in os.environ.get('STOPS', '1,2,3,4,5,6,7,0').split(',')]:
    os.environ["PGX_SEED_STOP"] = str(stop)
    for it in range(2):
        h = _capi.blast_search(db, reads); st = _capi.stage_times(); del h
    print("stop=%d seed_extend=%.1f ms sort=%.1f ms hits=%d" % (stop, st.seed_extend_ms, st.sort_ms, st.hits), flush=True)
