#!/usr/bin/env python3
"""Cross-check of the restatement (oracle/o_trim.c) against the reference's own Perl on random damaged read files
and option mixes, beyond the committed goldens.  Runs only where the reference tree is present (this container);
nothing is written into the repository.  Usage: python3 oracle/sweep_trim_vs_reference.py [first_seed] [count]"""
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.environ.get("PGX_REFERENCE", "/root/reference")
ORACLE = os.path.join(HERE, "bin", "pgx_oracle")


def run(cmd, cwd):
    p = subprocess.run(["timeout", "120"] + cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    path = os.path.join(cwd, "output_files", "trim2", "a.txt_runblast.fasta")
    return p.stdout, open(path, "rb").read() if os.path.exists(path) else None


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    from trim_inputs import random_case
    bad = 0
    for seed in range(first, first + count):
        a, b, g, t = random_case(seed, 400, 300)
        argv = ["-a", "a.txt"] + (["-b", "b.txt"] if b is not None else []) + (["-g", g] if g is not None else []) + (["-t", t] if t is not None else [])
        outs = []
        for cmd in (["perl", os.path.join(REF, "Trim", "trim2.4.pl")], [ORACLE, "trim2"]):
            d = tempfile.mkdtemp(prefix="pgx_sweep_")
            open(os.path.join(d, "a.txt"), "wb").write(a)
            if b is not None:
                open(os.path.join(d, "b.txt"), "wb").write(b)
            outs.append(run(cmd + argv, d))
            shutil.rmtree(d, ignore_errors=True)
        if outs[0] != outs[1]:
            bad += 1
            print("seed %d differs: argv %s" % (seed, argv))
    print("%d cases, %d differ" % (count, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
