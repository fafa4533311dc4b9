"""Timing aid: makeblastdb / blastn verbs on a 1 Gbp FASTA database (file -> .pgxdb -> open + index)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pangea_plus_amd as pg
pg.init(0)
tmp = tempfile.mkdtemp()
O = os.path.join(ROOT, "oracle/bin/pgx_oracle")
fa = os.path.join(tmp, "nt.fa")
t0 = time.time(); subprocess.check_call([O, "synth", "db", "--out", fa]); print("oracle wrote %.0f MB of FASTA in %.1f s" % (os.path.getsize(fa) / 1e6, time.time() - t0), flush=True)
t0 = time.time(); pg.makeblastdb(fa, os.path.join(tmp, "nt")); t1 = time.time()
print("makeblastdb: %.2f s" % (t1 - t0), flush=True)
t0 = time.time(); db = pg.Db.open(os.path.join(tmp, "nt")); t1 = time.time()
print("open + index: %.2f s (%d sequences)" % (t1 - t0, db.shape()[0]), flush=True)
