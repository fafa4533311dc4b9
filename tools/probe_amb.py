"""Timing aid: cost of ambiguity letters in the reads (one N in 1 % of the reads switches the batch to the AMB kernels)."""
import os, sys, time, subprocess, tempfile, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
n = 1_000_000
cfg = pg.SynthCfg.default()
db = pg.Db.from_synth(cfg)
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "reads.fa")
subprocess.check_call([os.path.join(ROOT, "oracle/bin/pgx_oracle"), "synth", "reads", "--out", fa, "--count", str(n)])
lines = open(fa).read().split("\n")
rng = random.Random(1)
for frac, tag in ((0.0, "clean"), (0.01, "1pct"), (1.0, "all")):
    out = list(lines)
    for i in range(1, len(out), 2):
        if out[i] and rng.random() < frac:
            p = rng.randrange(len(out[i])); out[i] = out[i][:p] + "N" + out[i][p + 1:]
    f2 = os.path.join(tmp, tag + ".fa"); open(f2, "w").write("\n".join(out))
    reads = pg.Reads.from_fasta(f2)
    for it in range(2):
        h = _capi.blast_search(db, reads); st = _capi.stage_times(); k = len(h); del h
    print("%s: seed %.1f ms sort %.1f ms -> %.1f M reads/s, %.1f hits/read" % (tag, st.seed_extend_ms, st.sort_ms, n / st.total_ms / 1e3, k / n), flush=True)
