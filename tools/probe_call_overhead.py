"""Wall time of one `classify_consensus` call beside the stage times of its kernels, on a batch the size of the file-to-file
sample: what the call costs on top of its kernels (table allocation, record download).  usage: python tools/probe_call_overhead.py [reads]"""
import ctypes as C, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
cfg = pg.SynthCfg.default()
tmp = tempfile.mkdtemp(prefix="pgx_call_")
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), tmp.encode()))
pg.TaxDb.create(tmp)
db = pg.Db.from_synth(cfg)
db.bind_taxonomy(pg.TaxDb.open(tmp))
reads = pg.Reads.from_synth(cfg, 0, n)
rdp = pg.Rdp.from_synth(cfg, 0, n, db)
for want_hits, want_records in ((True, True), (True, True), (True, True), (False, True), (False, False), (True, False), (True, True)):
    t0 = time.perf_counter()
    res = _capi.classify_consensus(db, reads, rdp, want_records=want_records, want_hits=want_hits)
    t1 = time.perf_counter()
    st = _capi.stage_times()
    del res
    t2 = time.perf_counter()
    print("hits=%d records=%d: call %.1f ms, kernels %.1f ms (attempts %d), release %.1f ms" % (want_hits, want_records, (t1 - t0) * 1e3, st.total_ms,
                                                                                            st.attempts, (t2 - t1) * 1e3), flush=True)
