"""debug aid: the inputs of tests/test_gpu_blast.py::test_reads_with_insertions_and_deletions[seed] through the checker and the
device, rows that differ per read (usage: python tools/dbg_dust_indel.py [seed])"""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pangea_plus_amd as pg
pg.init(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = random.Random(seed)
comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
anc = "".join(rng.choice("ACGT") for _ in range(1500))
for _ in range(25):
    a = rng.randrange(0, 1450)
    anc = anc[:a] + rng.choice("ACGT") * rng.randrange(3, 9) + anc[a + 6:]
anc = anc[:1500]
seqs = []
for i in range(60):
    s = list(anc)
    for p_ in rng.sample(range(len(s)), 30 + (i % 5) * 25):
        s[p_] = rng.choice("ACGT")
    for _ in range(i % 4):
        p_ = rng.randrange(10, len(s) - 10)
        if rng.random() < 0.5:
            del s[p_:p_ + rng.randrange(1, 6)]
        else:
            s[p_:p_] = [rng.choice("ACGT") for _ in range(rng.randrange(1, 6))]
    seqs.append("".join(s))
d = tempfile.mkdtemp()
db = os.path.join(d, "fam.fa")
open(db, "w").write("".join(">gi|%d|x|s%d|\n%s\n" % (i + 1, i, s) for i, s in enumerate(seqs)))
reads = []
rtext = {}
for i in range(500):
    L = rng.choice([100, 150, 150, 250, 400, 400, 1400])
    s = rng.choice(seqs)
    o = rng.randrange(0, max(1, len(s) - L))
    w = list(s[o:o + L])
    for p_ in rng.sample(range(len(w)), rng.randrange(0, 1 + len(w) // 40)):
        w[p_] = rng.choice("ACGT")
    for _ in range(rng.choice([0, 1, 1, 2, 3, 6]) * (1 + L // 500)):
        p_ = rng.randrange(1, len(w) - 1)
        if rng.random() < 0.5:
            run = 1
            while p_ + run < len(w) and w[p_ + run] == w[p_]:
                run += 1
            if rng.random() < 0.5 and run > 1:
                del w[p_]
            else:
                w.insert(p_, w[p_])
        elif rng.random() < 0.5:
            del w[p_:p_ + rng.choice([1, 2, 3, 7])]
        else:
            w[p_:p_] = [rng.choice("ACGT") for _ in range(rng.choice([1, 2, 3, 7]))]
    w = "".join(w)
    if i % 2:
        w = "".join(comp[c] for c in reversed(w))
    reads.append(">i%d\n%s\n" % (i, w))
    rtext["i%d" % i] = w
rd = os.path.join(d, "r.fa")
open(rd, "w").write("".join(reads))
want = os.path.join(d, "want.tsv")
subprocess.check_call([os.path.join(ROOT, "oracle", "bin", "pgx_oracle"), "blastn", "-query", rd, "-db", db, "-outfmt", "6", "-out", want, "-num_threads", "8"])
pg.makeblastdb(db, os.path.join(d, "db"))
got = os.path.join(d, "got.tsv")
pg.blastn(rd, os.path.join(d, "db"), got)
A, B = open(want).read().splitlines(), open(got).read().splitlines()
print("rows", len(A), len(B), "equal" if A == B else "DIFFERENT")
from collections import defaultdict
ga, gb = defaultdict(list), defaultdict(list)
for l in A: ga[l.split("\t")[0]].append(l)
for l in B: gb[l.split("\t")[0]].append(l)
nbad = 0
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_oracle_classify import dust_mask
for k in sorted(set(ga) | set(gb), key=lambda x: int(x[1:])):
    if ga[k] != gb[k]:
        nbad += 1
        if nbad <= 4:
            print("read", k, "len", len(rtext[k]), "rows want/got", len(ga[k]), len(gb[k]))
            print(rtext[k])
            print("".join("x" if m else "." for m in dust_mask(rtext[k])))
            sa, sb = set(ga[k]), set(gb[k])
            for l in sorted(sa - sb)[:3]: print("  only want:", l)
            for l in sorted(sb - sa)[:3]: print("  only got :", l)
print("reads that differ:", nbad)
