"""Timing aid: consensus stage when the lineages of a read's hits differ in depth (the order-dependent walk instead
of the closed form)."""
import os, sys, ctypes as C, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
from pangea_plus_amd import _capi
pg.init(0)
cfg = pg.SynthCfg.default()
n = 2_000_000
db = pg.Db.from_synth(cfg)
for tag in os.environ.get("TAGS", "uniform,ragged").split(","):
    d = tempfile.mkdtemp()
    _capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), d.encode()))
    if tag == "ragged":
        out = []
        for line in open(os.path.join(d, "nodes.dmp")):
            cols = line.split("\t|\t")
            if cols[2] == "species" and int(cols[0]) % 3 == 0:
                cols[2] = "no rank"
            out.append("\t|\t".join(cols))
        open(os.path.join(d, "nodes.dmp"), "w").write("".join(out))
    pg.TaxDb.create(d)
    db.bind_taxonomy(pg.TaxDb.open(d))
    reads = pg.Reads.from_synth(cfg, 0, n); rdp = pg.Rdp.from_synth(cfg, 0, n, db)
    for it in range(4):
        _capi.classify_consensus(db, reads, rdp, want_records=False, want_hits=False); st = _capi.stage_times()
        print("%s: seed %.1f ms group %.2f sort %.1f consensus %.1f total %.1f ms" % (tag, st.seed_extend_ms, st.group_ms, st.sort_ms, st.consensus_ms, st.total_ms), flush=True)
