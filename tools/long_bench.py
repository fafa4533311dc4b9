"""Timing aid for long reads: python tools/long_bench.py <read_len> <reads> [n_seq seq_len n_genus] -- search only, stage times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
L = int(sys.argv[1]); n = int(sys.argv[2])
kw = {}
if len(sys.argv) > 5:
    kw = dict(n_seq=int(sys.argv[3]), seq_len=int(sys.argv[4]), n_genus=int(sys.argv[5]))
cfg = pg.SynthCfg.default(read_len=L, **kw)
db = pg.Db.from_synth(cfg)
reads = pg.Reads.from_synth(cfg, 0, n)
for it in range(int(os.environ.get("ITERS", "3"))):
    h = _capi.blast_search(db, reads); st = _capi.stage_times(); k = len(h); del h
    print("read_len=%d n=%d: seed %.2f gapped %.2f sort %.2f total %.2f ms -> %.3f M reads/s, %.1f hits/read wide=%d" % (
        L, n, st.seed_extend_ms, st.gapped_ms, st.sort_ms, st.total_ms, n / st.total_ms / 1e3, k / n, st.gapped_wide), flush=True)
