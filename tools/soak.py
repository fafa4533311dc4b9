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

# ---- the file-to-file leg (mapped FASTA text, RDP import with its work arrays freed on a thread of their own): results
# repeat, free HBM and the process's resident memory do not creep
import resource
def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * resource.getpagesize() / 1e6
m = 500_000
seen = {}
rss0 = None
for it in range(12):
    first = (it % 3) * m
    fa, rf, out = os.path.join(d, "soak.fa"), os.path.join(d, "soak_rdp.txt"), os.path.join(d, "soak_out.txt")
    reads = pg.Reads.from_synth(cfg, first, m); rdp = pg.Rdp.from_synth(cfg, first, m, db)
    reads.write_fasta(fa); rdp.write_file(rf, reads, db)
    del reads, rdp
    r = pg.Reads.from_fasta(fa); p = pg.Rdp.from_file(rf, r, db)
    hits, recs = _capi.classify_consensus(db, r, p)
    _capi.consensus_format_file(db, r, hits, recs, out)
    dg = hashlib.md5(open(out, "rb").read()).hexdigest()
    assert seen.setdefault(first, dg) == dg, (it, first)
    del r, p, hits, recs
    free, total = torch.cuda.mem_get_info()
    if it == 5:
        rss0 = rss_mb()
    print("file call %2d first=%8d  %s  free HBM %.2f GB  rss %.0f MB" % (it, first, dg[:12], free / 1e9, rss_mb()), flush=True)
assert rss_mb() < rss0 + 300, (rss0, rss_mb())
print("file soak ok")
