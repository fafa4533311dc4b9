"""Seeded random taxonomies and hit tables for the taxcollector parity checks: the oracle-vs-reference sweep
(oracle/sweep_taxcollect_vs_reference.py) and the product-vs-oracle test (tests/test_gpu_annotate.py)."""
import os

RANKS = ["no rank", "superkingdom", "kingdom", "phylum", "class", "order", "family", "genus", "species", "subspecies",
         "species group", "tribe", "varietas", "subclass", "superfamily"]
WORDS = ["Bacillus", "subtilis", "sp.", "X1", "Foo6", "bar", "Candidatus", "group", "7A", "uncultured", "bacterium",
         "division", "str.", "K-12", "alpha", "Beta5", "gamma", "delta"]


def taxonomy(rng):
    """A random tree whose lineages pass through a superkingdom below the root (what the walk of the Perl stops at,
    taxcollector:226-300); a few gi numbers point at the root's own children or at no node at all (time-out inputs)."""
    n = rng.randint(8, 45)
    ids = sorted(rng.sample(range(10, 400), n))
    nodes = [(1, 1, "no rank", ""), (2, 1 if rng.random() < 0.5 else 3, "superkingdom", ""), (3, 1, "no rank", ""),
             (4, 3, "superkingdom", "")]
    below = [2, 4]
    for t in ids:
        p = rng.choice(below[-6:]) if rng.random() < 0.8 else rng.choice(below)
        nodes.append((t, p, rng.choice(RANKS[2:] + ["no rank"]), ""))
        below.append(t)
    names = {1: [("root", "", "scientific name")], 2: [("Bacteria", "", "scientific name")],
             3: [("cellular organisms", "", "scientific name")], 4: [("Eukaryota", "", "scientific name")]}
    for t in ids:
        k = rng.randint(1, 3)
        lst = []
        sci = rng.randrange(k)
        for j in range(k):
            nm = " ".join(rng.choice(WORDS) for _ in range(rng.randint(1, 3)))
            lst.append((nm, "", "scientific name" if j == sci else rng.choice(["synonym", "common name", "genbank common name"])))
        names[t] = lst
    gis = []
    g = 0
    for _ in range(rng.randint(5, 40)):
        g += rng.randint(1, 4)
        r = rng.random()
        gis.append((g, rng.choice(ids) if r < 0.995 else (rng.choice([3, 1, 450]))))
    return nodes, names, gis


def hits(rng, gis):
    out = []
    top = gis[-1][0] + 3
    for i in range(rng.randint(1, 25)):
        gi = rng.choice(gis)[0] if rng.random() < 0.9 else rng.randint(1, top)
        out.append("q%d\tgi|%d|gb|ACC%d.1|\t%.2f\t150\t1\t0\t1\t150\t11\t160\t2e-70\t 270\n" % (i // 2, gi, i, rng.uniform(80, 100)))
    return "".join(out)


def write_dumps(d, nodes, names, gis):
    """nodes.dmp / names.dmp / gi_taxid_nucl.dmp in the NCBI dump format (ncbitc.c:511-557 reads them back)."""
    with open(os.path.join(d, "nodes.dmp"), "w") as f:
        for t, p, r, e in sorted(nodes):
            f.write(f"{t}\t|\t{p}\t|\t{r}\t|\t{e}\t|\t0\t|\t1\t|\t11\t|\t1\t|\t0\t|\t1\t|\t0\t|\t0\t|\t\t|\n")
    with open(os.path.join(d, "names.dmp"), "w") as f:
        for t in sorted(names):
            for n, u, c in names[t]:
                f.write(f"{t}\t|\t{n}\t|\t{u}\t|\t{c}\t|\n")
    with open(os.path.join(d, "gi_taxid_nucl.dmp"), "w") as f:
        for g, t in gis:
            f.write(f"{g}\t{t}\n")
