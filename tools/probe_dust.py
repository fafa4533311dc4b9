"""Batch import cost of DUST (spec S3d): 10 M synthetic 150-base reads, with and without low-complexity stretches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
import torch

pg.init(0)
cfg = pg.SynthCfg.default()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = pg.Reads.from_synth(cfg, rep * n, n)
    torch.cuda.synchronize()
    print("from_synth %d reads (generate + strands + DUST + classes): %.1f ms" % (n, 1e3 * (time.perf_counter() - t0)))
    del r
