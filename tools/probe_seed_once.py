"""Profiling aid: one 2 M-read search (honours PGX_SEED_STOP) for PMC attribution by stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
cfg = pg.SynthCfg.default()
db = pg.Db.from_synth(cfg)
reads = pg.Reads.from_synth(cfg, 0, 2_000_000)
h = _capi.blast_search(db, reads)
print("seed_extend_ms", _capi.stage_times().seed_extend_ms)
