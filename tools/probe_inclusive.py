"""The file-to-file leg of bench.py alone (FASTA + RDP text in, consensus text out), a few times, with the RDP import in both
forms.  usage: python tools/probe_inclusive.py [reads]   (PGX_TRACE=1 for the stages inside the imports)"""
import ctypes as C, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
import bench
pg.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
cfg = pg.SynthCfg.default()
tmp = tempfile.mkdtemp(prefix="pgx_incl_")
_capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), tmp.encode()))
pg.TaxDb.create(tmp)
db = pg.Db.from_synth(cfg)
db.bind_taxonomy(pg.TaxDb.open(tmp))
for form in ("device", "device", "host", "device"):
    if form == "host":
        os.environ["PGX_RDP_HOST"] = "1"
    else:
        os.environ.pop("PGX_RDP_HOST", None)
    r = bench.inclusive(pg, _capi, cfg, db, tmp, 0, n)
    print(form, json.dumps({"reads_per_s": round(r["value"]), **{k: round(v, 4) for k, v in r["stages_s"].items()},
                            "kernels_ms": round(r["classify_consensus_kernels_ms"], 1), "attempts": r["classify_consensus_attempts"]}), flush=True)
