"""The roof of the seed stage's access shape: 64-byte lines per second for random 8-byte lane loads (pgx_probe_gather),
for tables that fit the Infinity Cache and tables far beyond it, plain and non-temporal loads."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
pg.init(0)
for gib in (0.125, 1, 16, 32):
    for stream in (0, 1):
        lps, ms = C.c_double(), C.c_double()
        pg._capi._check(pg.lib().pgx_probe_gather(int(gib * (1 << 30)), stream, C.byref(lps), C.byref(ms)))
        print("table %6.3f GiB, %s loads: %6.1f G lines/s = %5.2f TB/s of 64-byte lines (%.2f ms)"
              % (gib, "non-temporal" if stream else "plain       ", lps.value / 1e9, lps.value * 64 / 1e12, ms.value), flush=True)
