"""One-off deep check: N reads (default 200 000) of the bench stream at offset FIRST through the oracle chain on the host
cores and through the fused GPU path; the -outfmt 6 tables and the consensus texts must be identical.
env: N, FIRST, READ_LEN (default 150)."""
import ctypes as C, hashlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
from test_gpu_fullsize import OCfg, ORes
pg.init(0)
n = int(os.environ.get("N", "200000")); first = int(os.environ.get("FIRST", "1000000"))
cfg = pg.SynthCfg.default(read_len=int(os.environ.get("READ_LEN", "150")))
d = tempfile.mkdtemp()
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), d.encode()))
pg.TaxDb.create(d)
db = pg.Db.from_synth(cfg); db.bind_taxonomy(pg.TaxDb.open(d))
lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
oc = OCfg(cfg.seed, cfg.n_seq, cfg.seq_len, cfg.n_genus, cfg.read_seed, cfg.read_len)
res = ORes()
lib.o_bench_chain_files.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_char_p, C.c_void_p, C.c_char_p, C.c_char_p]
hp, cp = os.path.join(d, "oh.tsv"), os.path.join(d, "oc.txt")
assert lib.o_bench_chain_files(C.byref(oc), first, n, min(os.cpu_count() or 1, 16), d.encode(), C.byref(res), hp.encode(), cp.encode()) == 0
reads = pg.Reads.from_synth(cfg, first, n); rdp = pg.Rdp.from_synth(cfg, first, n, db)
hits, recs = _capi.classify_consensus(db, reads, rdp)
a, b = hits.format(db, reads), open(hp, "rb").read()
print("hit tables: %d rows, %s" % (a.count(b"\n"), "IDENTICAL" if a == b else "DIFFERENT"), flush=True)
a2, b2 = _capi.consensus_format(db, reads, hits, recs), open(cp, "rb").read()
print("consensus texts: %d bytes, %s" % (len(a2), "IDENTICAL" if a2 == b2 else "DIFFERENT"), flush=True)
sys.exit(0 if (a == b and a2 == b2) else 1)
