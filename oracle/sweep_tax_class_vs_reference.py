#!/usr/bin/env python3
"""Cross-check of the restatement of Tax_class/ncbitc.c (oracle/o_taxdb.c) against the reference's own C, compiled in
place (oracle/_ref/tax_class), on RANDOM taxonomies: `-c` on the same dumps (gi and node tables byte for byte), then
`-s`, `-g`, `-t`, `-n` for every gi / taxid of the dumps and a few that are not there (stdout + exit status).
Runs only where oracle/_ref exists; writes nothing into the repository.
Usage: python3 oracle/sweep_tax_class_vs_reference.py [first_seed] [count]"""
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
from tax_inputs import taxonomy, write_dumps  # noqa: E402

ORACLE = os.path.join(HERE, "bin", "pgx_oracle")
REF_TAX = os.path.join(HERE, "_ref", "tax_class")


def run(cmd, cwd):
    p = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=30)
    return p.returncode, p.stdout, bool(p.stderr)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    bad = calls = crashed = 0
    for seed in range(first, first + count):
        rng = random.Random(seed)
        nodes, names, gis = taxonomy(rng)
        dirs = []
        for tool in ("ref", "oracle"):
            d = tempfile.mkdtemp(prefix="pgx_xsweep_")
            write_dumps(d, nodes, names, gis)
            if tool == "ref":
                shutil.copy(REF_TAX, os.path.join(d, "tax_class"))
            dirs.append(d)
        cmd = ([os.path.join(dirs[0], "tax_class")], [ORACLE, "tax_class"])
        assert run(cmd[0] + ["-c"], dirs[0])[0] == run(cmd[1] + ["-c"], dirs[1])[0] == 0
        same = all(open(os.path.join(dirs[0], n), "rb").read() == open(os.path.join(dirs[1], n), "rb").read()
                   for n in ("gi_taxid_nucl.dmp.bin", "nodes.dmp.bin"))
        queries = [["-s", str(g)] for g, _ in gis] + [["-g", str(g)] for g, _ in gis[:8]] + [["-s", "0"], ["-s", str(gis[-1][0] + 5)], ["-g", "0"]]
        queries += [["-t", str(t)] for t, _, _, _ in nodes] + [["-n", str(t)] for t, _, _, _ in nodes] + [["-t", "9999"], ["-n", "9999"], ["-n", "5"]]
        for q in queries:
            calls += 1
            ref = run(cmd[0] + q, dirs[0])
            if ref[0] < 0:
                # the reference died on its own double fclose (a name search that probed position 0, ncbitc.c:637-640):
                # undefined there; the restatement defines it as "0" on stdout with text on stderr
                crashed += 1
                got = run(cmd[1] + q, dirs[1])
                if got != (0, b"0\n", True):
                    same = False
                    print("seed %d: %s: reference crashed, oracle gave %r" % (seed, " ".join(q), got))
                    break
                continue
            if ref != run(cmd[1] + q, dirs[1]):
                same = False
                print("seed %d: %s differs" % (seed, " ".join(q)))
                break
        bad += not same
        for d in dirs:
            shutil.rmtree(d, ignore_errors=True)
    print("%d taxonomies, %d command lines (%d crashed in the reference), %d taxonomies differ" % (count, calls, crashed, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
