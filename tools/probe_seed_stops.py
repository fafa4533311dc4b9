"""Measurement aid (PGX_STAGE_PROBES build): the seed kernel truncated after each of its stages (PGX_SEED_STOP), 10 M reads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
cfg = pg.SynthCfg.default()
db = pg.Db.from_synth(cfg)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reads = pg.Reads.from_synth(cfg, 0, n)
for stop in ("1", "8", "9", "2", "3", "4", "5", "6", "7", "0"):
    os.environ["PGX_SEED_STOP"] = stop
    t = []
    for _ in range(3):
        try:
            _capi.blast_search(db, reads)
        except Exception as e:  # noqa: BLE001  (a truncated kernel leaves tables the later stages may refuse)
            pass
        t.append(_capi.stage_times().seed_extend_ms)
    print("PGX_SEED_STOP=%s seed_extend_ms %.2f" % (stop, min(t)), flush=True)
