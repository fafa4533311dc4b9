"""Quick stage timing of the fused path on the bench workload (no CPU baseline, no inclusive leg): median of a few steps.
usage: python tools/quick_bench.py [reads] [steps]"""
import os, sys, tempfile, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = pg.SynthCfg.default()
tmp = tempfile.mkdtemp(prefix="pgx_qb_")
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), tmp.encode()))
pg.TaxDb.create(tmp)
tax = pg.TaxDb.open(tmp)
db = pg.Db.from_synth(cfg)
db.bind_taxonomy(tax)
if os.environ.get("QB_DUST", "1") != "0":
    db.set_dust_each_search(True)
reads = pg.Reads.from_synth(cfg, 0, n)
rdp = pg.Rdp.from_synth(cfg, 0, n, db)
rows = []
for i in range(steps + 2):
    _capi.classify_consensus(db, reads, rdp, want_records=False, want_hits=False)
    st = _capi.stage_times()
    if i >= 2:
        rows.append((st.seed_extend_ms, st.gapped_ms, st.sort_ms, st.total_ms, st.hits, st.gapped_wide, st.dust_ms))
rows.sort(key=lambda r: r[3])
m = rows[len(rows) // 2]
print("reads=%d dust=%.2f seed=%.2f gapped=%.2f sort=%.2f total=%.2f ms hits=%d wide=%d -> %.1f M reads/s" % (n, m[6], m[0], m[1], m[2], m[3], m[4], m[5], n / m[3] / 1e3), flush=True)
