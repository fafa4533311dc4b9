"""Timing aid: database shapes other than the bench's (short subjects: several per 512-base block; small database:
hashed bucket table)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
n = 2_000_000
for tag, kw in (("1 Gbp of 1500-bp subjects", dict()),
                ("1 Gbp of 250-bp subjects", dict(n_seq=4_000_000, seq_len=250, n_genus=120_000, read_len=150)),
                ("0.4 Gbp of 100-bp subjects", dict(n_seq=4_000_000, seq_len=100, n_genus=120_000, read_len=100)),
                ("50 Mbp of 1500-bp subjects", dict(n_seq=33_334, seq_len=1500, n_genus=1000, read_len=150))):
    cfg = pg.SynthCfg.default(**kw)
    db = pg.Db.from_synth(cfg)
    reads = pg.Reads.from_synth(cfg, 0, n)
    for it in range(2):
        h = _capi.blast_search(db, reads); st = _capi.stage_times(); k = len(h); del h
    print("%s (index bits %d): seed %.1f ms sort %.1f ms -> %.1f M reads/s, %.1f hits/read, %.0f postings/read" % (
        tag, db.shape()[3], st.seed_extend_ms, st.sort_ms, n / st.total_ms / 1e3, k / n, st.postings / n), flush=True)
    del db, reads
