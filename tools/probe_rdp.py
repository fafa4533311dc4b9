import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
import tempfile
tmp = tempfile.mkdtemp()
cfg = pg.SynthCfg.default(n_seq=20000, n_genus=600)
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), tmp.encode()))
pg.TaxDb.create(tmp); tax = pg.TaxDb.open(tmp)
db = pg.Db.from_synth(cfg); db.bind_taxonomy(tax)
n = 2000000
reads = pg.Reads.from_synth(cfg, 0, n); rdp = pg.Rdp.from_synth(cfg, 0, n, db)
reads.write_fasta(tmp + "/r.fa"); rdp.write_file(tmp + "/rdp.txt", reads, db)
r = pg.Reads.from_fasta(tmp + "/r.fa")
for k in range(2):
    t0 = time.time(); p = pg.Rdp.from_file(tmp + "/rdp.txt", r, db); print("rdp_from_file %.3f s" % (time.time() - t0), flush=True)
