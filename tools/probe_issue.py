"""The issue roof of the gapped stage's instruction mix (pgx_probe_issue): vector wave-instructions per second per SIMD at
1, 2, 4 and 8 resident wavefronts, one instruction kind at a time; the shader clock is measured in the 1- and 2-wave runs."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangea_plus_amd as pg
pg.init(0)
L = pg.lib()
L.pgx_probe_issue_name.restype = C.c_char_p
k = 0
print("%-28s %8s %8s %8s %8s   %s" % ("instruction", "1 wave", "2", "4", "8", "G instr/s/SIMD; [cycles per instruction of a lone wavefront]; cycles/instr/SIMD at saturation"))
while L.pgx_probe_issue_name(k):
    name = L.pgx_probe_issue_name(k).decode()
    row, lone, clk = [], 0, 2.4e9
    for w in (1, 2, 4, 8):
        out = (C.c_double * 4)()
        pg._capi._check(L.pgx_probe_issue(w, k, out))
        row.append(out[0] / 1e9)
        if w == 1:
            lone, clk = out[1], out[3]
    print("%-28s %8.3f %8.3f %8.3f %8.3f   [%.2f]  %.2f" % (name, row[0], row[1], row[2], row[3], lone, clk / 1e9 / max(row)), flush=True)
    k += 1
