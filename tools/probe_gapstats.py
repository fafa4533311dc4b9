"""Lane utilisation of the gapped stage's lane-per-HSP kernel (measurement build: PGX_STAGE_PROBES=1 python pangea-plus_amd/build.py).
Counters of csrc/gapped.hip (g_gap_stats) over one search of N synthetic reads against the bench database."""
import ctypes as C
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
pg.init(0)
cfg = pg.SynthCfg.default(read_len=int(os.environ.get("READ_LEN", "150")))  # READ_LEN=500: the deep rows
tmp = tempfile.mkdtemp()
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), tmp.encode()))
pg.TaxDb.create(tmp)
db = pg.Db.from_synth(cfg)
db.bind_taxonomy(pg.TaxDb.open(tmp))
reads = pg.Reads.from_synth(cfg, 0, n)
rdp = pg.Rdp.from_synth(cfg, 0, n, db)
f = getattr(pg.lib(), "pgx_gap_stats", None)
if f is None:
    sys.exit("this library was not built with PGX_STAGE_PROBES=1")
_capi.classify_consensus(db, reads, rdp, want_records=False, want_hits=False)
z = (C.c_ulonglong * 8)()
f(z, 1)
_capi.classify_consensus(db, reads, rdp, want_records=False, want_hits=False)
st = _capi.stage_times()
f(z, 1)
steps, cells, slides, levels, rounds, lanes_on, walk, live_levels = [int(x) for x in z]
print(json.dumps({"reads": n, "gapped_ms": st.gapped_ms, "rounds": rounds, "lanes_on_per_round": lanes_on / max(rounds, 1),
                  "levels_per_round": levels / max(rounds, 1), "cell_steps_per_round": steps / max(rounds, 1),
                  "lane_cells_per_round": cells / max(rounds, 1), "utilisation_of_cell_steps": cells / max(steps * 64, 1),
                  "slide_rounds_per_cell_step": slides / max(steps, 1), "b0_walk_rounds_per_round": walk / max(rounds, 1),
                  "live_lanes_per_level": live_levels / max(levels, 1), "cells_per_side": cells / max(lanes_on, 1)}, indent=1))
