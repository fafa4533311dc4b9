"""Timing aid: a 1 Gbp database with IUPAC letters in 1 % of its sequences against the same database without them."""
import os, subprocess, sys, tempfile, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
tmp = tempfile.mkdtemp()
O = os.path.join(ROOT, "oracle/bin/pgx_oracle")
fa = os.path.join(tmp, "nt.fa")
subprocess.check_call([O, "synth", "db", "--out", fa])
cfg = pg.SynthCfg.default()
n = 2_000_000
reads = pg.Reads.from_synth(cfg, 0, n)
rng = random.Random(7)
for tag in ("clean", "iupac"):
    if tag == "iupac":
        out = []
        with open(fa) as f:
            for line in f:
                if not line.startswith(">") and rng.random() < 0.01:
                    p = rng.randrange(len(line) - 1); line = line[:p] + rng.choice("NRYK") + line[p + 1:]
                out.append(line)
        open(fa, "w").write("".join(out)); del out
    pg.makeblastdb(fa, os.path.join(tmp, tag))
    db = pg.Db.open(os.path.join(tmp, tag))
    for sw in ("", "1"):
        if sw:
            os.environ["PGX_NO_AMB_BLK"] = sw
        else:
            os.environ.pop("PGX_NO_AMB_BLK", None)
        for it in range(3):
            h = _capi.blast_search(db, reads); st = _capi.stage_times(); k = len(h); del h
        print("%s database (has_amb=%d, block bitmap %s): seed %.1f ms sort %.1f ms -> %.1f M reads/s, %d hits" % (
            tag, db.shape()[2], "off" if sw else "on", st.seed_extend_ms, st.sort_ms, n / st.total_ms / 1e3, k), flush=True)
    os.environ.pop("PGX_NO_AMB_BLK", None)
    del db
