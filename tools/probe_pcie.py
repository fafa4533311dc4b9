"""Measurement aid: PCIe-inclusive rate — reads handed over as a host FASTA file (parse + 2-bit pack on the host,
upload, reverse complements, then the same fused pipeline)."""
import ctypes as C, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
cfg = pg.SynthCfg.default()
tmp = tempfile.mkdtemp()
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), tmp.encode()))
pg.TaxDb.create(tmp); tax = pg.TaxDb.open(tmp)
db = pg.Db.from_synth(cfg); db.bind_taxonomy(tax)
fa = os.path.join(tmp, "reads.fa")
subprocess.check_call([os.path.join(ROOT, "oracle/bin/pgx_oracle"), "synth", "reads", "--out", fa, "--count", str(n)])
rdp = pg.Rdp.from_synth(cfg, 0, n, db)
for it in range(2):
    t0 = time.time(); reads = pg.Reads.from_fasta(fa); t1 = time.time()
    _capi.classify_consensus(db, reads, rdp, want_records=False, want_hits=False); t2 = time.time()
    print("n=%d  host FASTA->HBM %.2fs  pipeline %.3fs  => %.2f M reads/s inclusive (%.1f M/s resident)" % (
        n, t1 - t0, t2 - t1, n / (t2 - t0) / 1e6, n / (t2 - t1) / 1e6), flush=True)
