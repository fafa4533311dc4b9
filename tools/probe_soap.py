"""Timing aid, SOAP mode (SURVEY 8(d) CPU-baseline plan 3): 1 M synthetic reads vs a 50 Mbp slice, the same
inputs the reference's closed `soap -p 8 -M 4 -r 2` was timed on in the build container (23.7 s, 812 865 rows)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pangea_plus_amd as pg
pg.init(0)
tmp = tempfile.mkdtemp()
O = os.path.join(ROOT, "oracle/bin/pgx_oracle")
A = ["--n-seq", "33334", "--seq-len", "1500", "--n-genus", "1000", "--read-len", "150"]
ref, reads = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fa")
subprocess.check_call([O, "synth", "db", "--out", ref] + A)
subprocess.check_call([O, "synth", "reads", "--out", reads, "--count", "1000000"] + A)
t0 = time.time(); pg.soap_index(ref); t1 = time.time()
print("index build (2bwt-builder verb): %.2f s" % (t1 - t0), flush=True)
for it in range(2):
    t0 = time.time()
    pg.soap(reads, ref + ".index", os.path.join(tmp, "out.txt"), r=2)
    t1 = time.time()
    rows = sum(1 for _ in open(os.path.join(tmp, "out.txt"), "rb"))
    print("soap -M 4 -r 2: %.2f s for 1000000 reads -> %.2f M reads/s end to end (file in, text out); %d rows" % (
        t1 - t0, 1.0 / (t1 - t0), rows), flush=True)
