"""Profiling aid: stage-truncated timing of k_sort_consensus (PGX_SORT_STOP) at full scale."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, tempfile
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
cfg = pg.SynthCfg.default()
tmp = tempfile.mkdtemp()
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), tmp.encode()))
pg.TaxDb.create(tmp)
tax = pg.TaxDb.open(tmp)
db = pg.Db.from_synth(cfg)
db.bind_taxonomy(tax)
n = 10_000_000
reads = pg.Reads.from_synth(cfg, 0, n)
rdp = pg.Rdp.from_synth(cfg, 0, n, db)
for stop in (1, 2, 3, 5, 6, 4, 0, 7, 0):   # 7: every subject record read from an L2-resident part of the table (wrong answers, right cost)
    os.environ["PGX_SORT_STOP"] = str(stop)
    for it in range(2):
        _capi.classify_consensus(db, reads, rdp, want_records=False, want_hits=False)
        st = _capi.stage_times()
    print("stop=%d seed_extend=%.1f ms sort+consensus=%.1f ms" % (stop, st.seed_extend_ms, st.sort_ms + st.consensus_ms), flush=True)
