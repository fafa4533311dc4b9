#!/usr/bin/env python3
"""Cross-check of the restatement (oracle/o_consensus.c) against the reference's own Perl
(Consensus/Consensus_BLAST_SOAP_RDP-1.1.pl) on random classified-hit tables and RDP streams, beyond the committed
goldens: ragged and repeated lineages, unknown rank tags, quoted / digit-bearing RDP names, pident texts whose string
order differs from their numeric order, reads without an RDP line.  Every RDP read has BLAST lines (the reference
never terminates otherwise).  Runs only where the reference tree is present; writes nothing into the
repository.  Usage: python3 oracle/sweep_consensus_vs_reference.py [first_seed] [count]"""
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PGX_REFERENCE", "/root/reference")
PERL = os.path.join(REF, "Consensus", "Consensus_BLAST_SOAP_RDP-1.1.pl")
ORACLE = os.path.join(HERE, "bin", "pgx_oracle")
RANKS = ["domain", "phylum", "class", "order", "family", "genus", "species"]
NUM = "150\t1\t0\t1\t150\t11\t160\t2e-70\t270"


def names(rng):
    pools = []
    for k in range(7):
        pools.append([("N%d%s" % (k, "".join(rng.choice("abcxyz_") for _ in range(rng.randint(2, 7))))).strip("_") or "Nx" for _ in range(4)])
    return pools


def lineage(rng, truth, pools):
    toks = []
    for k in range(7):
        if rng.random() < 0.2:
            continue
        name = truth[k] if rng.random() < 0.75 else rng.choice(pools[k])
        tag = k if rng.random() < 0.93 else rng.choice([9, 7, k])
        toks.append("[%d]%s;" % (tag, name))
    if rng.random() < 0.08:
        toks = toks * 2
    if rng.random() < 0.04:
        return "Unidentified(GI:%d);" % rng.randint(1, 99)
    return "".join(toks) or "[0]%s;" % truth[0]


def pident(rng):
    return rng.choice(["100.00", "99.99", "99.33", "9.50", "10.00", "95.00", "88.00", "%.2f" % rng.uniform(70, 100), "%.2f" % rng.uniform(70, 100)])


def rdp_line(rng, rid, truth, pools):
    trip = []
    for k in range(7):
        if rng.random() < 0.25:
            continue
        name = truth[k] if rng.random() < 0.8 else rng.choice(pools[k])
        style = rng.randrange(5)
        if style == 0:
            name = '"%s"' % name
        elif style == 1:
            name = name + " 1"
        rank = RANKS[k] if rng.random() < 0.95 else rng.choice(["kingdom", "subclass", ""])
        trip += [name, rank, "%.2f" % rng.random()]
    if rng.random() < 0.05:
        trip = trip[:-1]  # odd token count
    return rid + "\t\t\t\t\t" + "\t".join(trip) + "\n"


def case(seed):
    rng = random.Random(seed)
    pools = names(rng)
    blast, rdp = [], []
    for i in range(rng.randint(1, 40)):
        rid = "q%d" % i
        truth = [rng.choice(p) for p in pools]
        sep = "\t\t" if rng.random() < 0.1 else "\t"
        for _ in range(rng.randint(1, 8)):
            blast.append("%s\t%s%s%s\t%s\n" % (rid, lineage(rng, truth, pools), sep, pident(rng), NUM))
        if rng.random() < 0.9:
            rdp.append(rdp_line(rng, rid, truth, pools))
    if not rdp:
        rdp.append(rdp_line(rng, "q0", [p[0] for p in pools], pools))
    return "".join(blast), "".join(rdp)


def run(cmd, d, out):
    try:
        p = subprocess.run(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20)
    except subprocess.TimeoutExpired:
        return None
    res = open(out, "rb").read() if os.path.exists(out) else None
    return p.stdout.replace(out.encode(), b"@OUT@"), res


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    bad = hung = 0
    for seed in range(first, first + count):
        b, r = case(seed)
        d = tempfile.mkdtemp(prefix="pgx_csweep_")
        open(os.path.join(d, "b.tsv"), "w").write(b)
        open(os.path.join(d, "r.tsv"), "w").write(r)
        ref = run(["perl", PERL, "-b", "b.tsv", "-r", "r.tsv", "-o", os.path.join(d, "ref.txt")], d, os.path.join(d, "ref.txt"))
        got = run([ORACLE, "consensus", "-b", "b.tsv", "-r", "r.tsv", "-o", os.path.join(d, "got.txt")], d, os.path.join(d, "got.txt"))
        if ref is None:
            hung += 1
        elif ref != got:
            bad += 1
            keep = "/tmp/pgx_csweep_fail_%d" % seed
            shutil.copytree(d, keep, dirs_exist_ok=True)
            print("seed %d differs (kept in %s)" % (seed, keep))
        shutil.rmtree(d, ignore_errors=True)
    print("%d cases, %d differ, %d reference time-outs" % (count, bad, hung))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
