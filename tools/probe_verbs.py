"""Timing aid: the reference's command lines as file verbs, end to end on files (200 k reads vs the 1 Gbp synthetic
database): makeblastdb-equivalent index from memory, blastn, taxcollector, consensus, megaclust2, megaclustable."""
import ctypes as C, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
n = int(os.environ.get("N", "200000"))
cfg = pg.SynthCfg.default()
d = tempfile.mkdtemp()
os.mkdir(os.path.join(d, "Tax_class"))
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), os.path.join(d, "Tax_class").encode()))
t0 = time.time(); pg.TaxDb.create(os.path.join(d, "Tax_class")); print("tax_class -c: %.2f s" % (time.time() - t0), flush=True)
O = os.path.join(ROOT, "oracle/bin/pgx_oracle")
subprocess.check_call([O, "synth", "reads", "--out", os.path.join(d, "reads.fa"), "--count", str(n)])
subprocess.check_call([O, "synth", "rdp", "--out", os.path.join(d, "rdp.tsv"), "--count", str(n)])
db = pg.Db.from_synth(cfg)
reads = pg.Reads.from_fasta(os.path.join(d, "reads.fa"))
t0 = time.time(); hits = _capi.blast_search(db, reads); open(os.path.join(d, "hits.tsv"), "wb").write(hits.format(db, reads)); t1 = time.time()
rows = sum(1 for _ in open(os.path.join(d, "hits.tsv"), "rb"))
print("search + -outfmt 6 file: %.2f s, %d rows, %.0f MB" % (t1 - t0, rows, os.path.getsize(os.path.join(d, "hits.tsv")) / 1e6), flush=True)
t0 = time.time(); pg.taxcollector(os.path.join(d, "hits.tsv"), os.path.join(d, "hits_class.tsv"), taxdir=os.path.join(d, "Tax_class")); t1 = time.time()
print("taxcollector verb: %.2f s -> %.2f M hits/s" % (t1 - t0, rows / (t1 - t0) / 1e6), flush=True)
t0 = time.time()
try:
    pg.consensus(os.path.join(d, "hits_class.tsv"), os.path.join(d, "rdp.tsv"), os.path.join(d, "cons.txt"))
except pg.PangeaError as e:  # a read without hits stops the reference's cursor walk: lines before it are written
    print("consensus verb stopped where the reference hangs:", str(e)[:90], flush=True)
t1 = time.time()
done = sum(1 for l in open(os.path.join(d, "cons.txt"), "rb") if l.startswith(b"#"))
print("consensus verb: %.2f s for %d reads -> %.2f M reads/s" % (t1 - t0, done, done / (t1 - t0) / 1e6), flush=True)
t0 = time.time(); pg.megaclust2(os.path.join(d, "cons.txt"), os.path.join(d, "m80.csv"), s=80, b=100); t1 = time.time()
print("megaclust2 verb: %.2f s" % (t1 - t0), flush=True)
