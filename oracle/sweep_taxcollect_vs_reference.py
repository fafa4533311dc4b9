#!/usr/bin/env python3
"""Cross-check of the restatement (oracle/o_taxdb.c + o_taxcollect.c) against the reference's own code — its C
tax_class (compiled in place into oracle/_ref) driven by its Perl NCBI-taxcollector-0.01.pl — on RANDOM taxonomies
and hit tables, beyond the committed goldens: random trees with any mix of ranks (missing, repeated, unlisted),
names with digits / blanks / several name classes, gi numbers with and without a taxid.  Inputs on which the Perl
never terminates are cut off by a time-out; there the restatement must report PGX/oracle "REFHANG" with
the same lines written before the stop.  Runs only where the reference tree is present; writes nothing into the
repository.  Usage: python3 oracle/sweep_taxcollect_vs_reference.py [first_seed] [count]"""
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_goldens as G  # noqa: E402  (write_dumps, the paths of the reference scripts)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
from tax_inputs import hits, taxonomy  # noqa: E402

ORACLE = os.path.join(HERE, "bin", "pgx_oracle")
def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    bad = hung = 0
    for seed in range(first, first + count):
        rng = random.Random(seed)
        G.NODES, G.NAMES, G.GIS = taxonomy(rng)
        d = tempfile.mkdtemp(prefix="pgx_tsweep_")
        G.make_taxdir(d)
        open(os.path.join(d, "in.tsv"), "w").write(hits(rng, G.GIS))
        try:
            p = subprocess.run(["perl", G.PERL_TAXCOL, "-f", "in.tsv", "-o", "ref.tsv"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=25)
            ref = (p.stdout, open(os.path.join(d, "ref.tsv"), "rb").read())
        except subprocess.TimeoutExpired:
            ref = None
        q = subprocess.run([ORACLE, "taxcollector", "-f", "in.tsv", "-o", "got.tsv", "-d", "Tax_class"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        got = (q.stdout, open(os.path.join(d, "got.tsv"), "rb").read() if os.path.exists(os.path.join(d, "got.tsv")) else None)
        if ref is None:
            hung += 1
            ok = q.returncode != 0  # the restatement stops with its REFHANG status where the Perl loops for ever
        else:
            ok = ref == got and q.returncode == 0
        if not ok:
            bad += 1
            keep = "/tmp/pgx_tsweep_fail_%d" % seed
            shutil.copytree(d, keep, dirs_exist_ok=True)
            print("seed %d differs (kept in %s; reference %s, oracle rc %d)" % (seed, keep, "timed out" if ref is None else "finished", q.returncode))
        shutil.rmtree(d, ignore_errors=True)
    print("%d cases, %d differ, %d reference time-outs" % (count, bad, hung))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
