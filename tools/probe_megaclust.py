"""Timing aid: the post-consensus steps at bench scale (10 M reads vs 1 Gbp): fused table from HBM records,
the file verb on the formatted text of 1 M reads, the pivot at every rank."""
import os, sys, time, tempfile, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
cfg = pg.SynthCfg.default()
d = tempfile.mkdtemp()
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), d.encode()))
pg.TaxDb.create(d); tax = pg.TaxDb.open(d)
db = pg.Db.from_synth(cfg); db.bind_taxonomy(tax)
n = int(os.environ.get("N", "10000000"))
reads = pg.Reads.from_synth(cfg, 0, n); rdp = pg.Rdp.from_synth(cfg, 0, n, db)
hits, recs = _capi.classify_consensus(db, reads, rdp)
for it in range(3):
    t0 = time.perf_counter()
    csv, log = pg.megaclust_batch(db, reads, hits, recs, s="80", b="100")
    t1 = time.perf_counter()
    print("fused megaclust: %.1f ms for %d reads -> %.1f M reads/s; table %d lines; %s" % (
        1e3 * (t1 - t0), n, n / (t1 - t0) / 1e6, csv.count(b"\n"), log.split(b"\n")[1:3]), flush=True)
m = min(n, 1_000_000)
sub_reads = pg.Reads.from_synth(cfg, 0, m); sub_rdp = pg.Rdp.from_synth(cfg, 0, m, db)
h2, r2 = _capi.classify_consensus(db, sub_reads, sub_rdp)
text = _capi.consensus_format(db, sub_reads, h2, r2)
open(os.path.join(d, "cons.txt"), "wb").write(text)
for it in range(2):
    t0 = time.perf_counter()
    pg.megaclust2(os.path.join(d, "cons.txt"), os.path.join(d, "t80.csv"), s="80", b="100")
    t1 = time.perf_counter()
    print("file verb: %.1f ms for %d reads (%.0f MB text) -> %.2f M reads/s" % (1e3 * (t1 - t0), m, len(text) / 1e6, m / (t1 - t0) / 1e6), flush=True)
pg.megaclust2(os.path.join(d, "cons.txt"), os.path.join(d, "t99.csv"), s="99", b="250")
for level in (0, 3, 5):
    t0 = time.perf_counter()
    pg.megaclustable(["-m", os.path.join(d, "t80.csv"), os.path.join(d, "t99.csv"), "-t", str(level), "-o", os.path.join(d, "piv.txt")])
    t1 = time.perf_counter()
    print("pivot level %d: %.1f ms, %d rows" % (level, 1e3 * (t1 - t0), open(os.path.join(d, "piv.txt"), "rb").read().count(b"\n")), flush=True)
