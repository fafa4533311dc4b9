"""Timing aid: the read shape Trim hands to Classify (README.md:34 -> :96): two kept mates of ~130 bases joined by
GAP N's (trim2 -g 100 / the default 189).  1 Gbp synthetic database, mates = consecutive reads of the synthetic stream."""
import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
cfg = pg.SynthCfg.default()
db = pg.Db.from_synth(cfg)
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "reads.fa")
subprocess.check_call([os.path.join(ROOT, "oracle/bin/pgx_oracle"), "synth", "reads", "--out", fa, "--count", str(2 * n)])
seqs = open(fa).read().split("\n")[1::2]
for gap, keep in ((0, 150), (100, 130), (189, 130)):
    f2 = os.path.join(tmp, "joined%d.fa" % gap)
    with open(f2, "w") as f:
        for i in range(n):
            f.write(">p%d\n%s%s%s\n" % (i, seqs[2 * i][:keep], "N" * gap, seqs[2 * i + 1][:keep]))
    reads = pg.Reads.from_fasta(f2)
    for it in range(2):
        h = _capi.blast_search(db, reads); st = _capi.stage_times(); k = len(h); del h
    print("gap %3d (%d bases per read): seed %.1f ms sort %.1f ms -> %.1f M joined reads/s, %.1f hits/read"
          % (gap, 2 * keep + gap, st.seed_extend_ms, st.sort_ms, n / st.total_ms / 1e3, k / n), flush=True)
