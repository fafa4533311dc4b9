import ctypes as C, os, sys, tempfile, time
sys.path.insert(0, "/root/repo")
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
n = 2_000_000
cfg = pg.SynthCfg.default()
tmp = tempfile.mkdtemp(prefix="pgx_p2_")
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), tmp.encode()))
pg.TaxDb.create(tmp)
db = pg.Db.from_synth(cfg)
db.bind_taxonomy(pg.TaxDb.open(tmp))
fa, rf = os.path.join(tmp, "r.fa"), os.path.join(tmp, "r.rdp")
reads = pg.Reads.from_synth(cfg, 0, n); rdp = pg.Rdp.from_synth(cfg, 0, n, db)
reads.write_fasta(fa); rdp.write_file(rf, reads, db)
del reads, rdp
for it in range(3):
    t0 = time.perf_counter()
    r = pg.Reads.from_fasta(fa)
    t1 = time.perf_counter()
    p = pg.Rdp.from_file(rf, r, db)
    t2 = time.perf_counter()
    ts = []
    for k in range(3):
        a = time.perf_counter()
        hits, recs = _capi.classify_consensus(db, r, p)
        b = time.perf_counter()
        st = _capi.stage_times()
        ts.append((b - a) * 1e3)
        ts.append(-st.total_ms)
        del hits
    print("iter %d: fasta %.1f rdp %.1f classify calls %s kernels %.1f" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ["%.1f" % x for x in ts], st.total_ms), flush=True)
    # synthetic batch of the same reads, same handle
    rs = pg.Reads.from_synth(cfg, 0, n); ps = pg.Rdp.from_synth(cfg, 0, n, db)
    ts = []
    for k in range(2):
        a = time.perf_counter()
        hits, recs = _capi.classify_consensus(db, rs, ps)
        ts.append((time.perf_counter() - a) * 1e3)
        del hits
    print("   synthetic batch: %s" % ["%.1f" % x for x in ts], flush=True)
    del r, p, rs, ps
