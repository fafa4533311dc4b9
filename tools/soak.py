"""Stability aid: many pipeline calls in one process (fresh batches, hits kept and dropped, records on and off);
free HBM must not shrink from call to call and results must repeat."""
import os, sys, ctypes as C, tempfile, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
cfg = pg.SynthCfg.default()
d = tempfile.mkdtemp()
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), d.encode()))
pg.TaxDb.create(d); tax = pg.TaxDb.open(d)
db = pg.Db.from_synth(cfg); db.bind_taxonomy(tax)
n = 2_000_000
first_digest = None
for it in range(24):
    first = (it % 3) * n
    reads = pg.Reads.from_synth(cfg, first, n); rdp = pg.Rdp.from_synth(cfg, first, n, db)
    if it % 2 == 0:
        hits, recs = _capi.classify_consensus(db, reads, rdp)
        dg = hashlib.md5(recs.tobytes()).hexdigest() + ":%d" % len(hits)
        del hits
    else:
        _capi.classify_consensus(db, reads, rdp, want_records=False, want_hits=False)
        dg = "-"
    if it % 6 == 0:
        if first_digest is None:
            first_digest = dg
        assert dg == first_digest, (it, dg, first_digest)
    free, total = torch.cuda.mem_get_info()
    print("call %2d first=%8d  %s  free HBM %.2f GB" % (it, first, dg[:20], free / 1e9), flush=True)
    del reads, rdp
print("soak ok")
