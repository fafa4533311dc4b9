#!/usr/bin/env python3
"""BASELINE config 1 ("validation_dataset/rdp_download_373seqs.fa vs itself, blastn -outfmt 6 -> tax_class -> Consensus"):
the plumbing fixture under tests/golden/config1/, made by running the chain here.

  data       the reference's 373 RDP 16S sequences (lower case, IUPAC letters, 1 200-1 500 bases), relabelled
             `gi|<1000+i>|rdp|<RDP id>|` because NCBI-taxcollector-0.01.pl never terminates on an id without `gi|N|`
             (SURVEY 3.4); a synthetic 7-rank taxonomy in NCBI dump format whose genus / species names are the ones in the
             FASTA headers; a synthetic RDP stream (five-tab format, Consensus:126-132) for the queries
  step 1     oracle `blastn -query first N sequences -db all 373` (spec pgx-blastn v2: BLAST+ itself is not vendored)
  step 2     the REFERENCE's tax_class (oracle/_ref/tax_class -c) and the REFERENCE's NCBI-taxcollector-0.01.pl
  step 3     the REFERENCE's Consensus_BLAST_SOAP_RDP-1.1.pl
Stored: the inputs (gzip), the consensus text and the two stdout logs, sha256 + line counts of the two intermediate
tables (they are megabytes).  Only data is written to the repo.

Usage: python3 oracle/gen_goldens_config1.py [N_QUERIES]
"""
import gzip
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("PGX_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "config1")
ORACLE = os.path.join(HERE, "bin", "pgx_oracle")
REF_TAX = os.path.join(HERE, "_ref", "tax_class")


def read_fasta(path):
    recs = []
    for line in open(path):
        line = line.rstrip("\r\n")
        if line.startswith(">"):
            recs.append([line[1:], []])
        elif recs:
            recs[-1][1].append(line)
    return [(h, "".join(s)) for h, s in recs]


def main():
    n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    recs = read_fasta(os.path.join(REF, "validation_dataset", "rdp_download_373seqs.fa"))
    assert len(recs) == 373
    os.makedirs(OUT, exist_ok=True)
    # ---- taxonomy: genus = first word of the organism name, species = first two words; the ranks above are synthetic
    # groups of genera (sorted), so that neighbours in the alphabet share family / order / ...
    def organism(h):
        name = h.split(" ", 1)[1].split(";")[0]
        name = re.sub(r"\(T\)", "", name).strip()
        words = [w for w in re.split(r"\s+", name) if w]
        genus = re.sub(r"[^A-Za-z]", "", words[0]) or "Unnamed"
        species = genus + " " + (re.sub(r"[^A-Za-z0-9.-]", "", words[1]) if len(words) > 1 else "sp.")
        return genus, species
    orgs = [organism(h) for h, _ in recs]
    genera = sorted({g for g, _ in orgs})
    species = sorted({s for _, s in orgs})
    nodes, names = [(1, 1, "no rank")], {1: "root"}
    nxt = [2]

    def add(parent, rank, name):
        t = nxt[0]
        nxt[0] += 1
        nodes.append((t, parent, rank))
        names[t] = name
        return t
    dom = add(1, "superkingdom", "Bacteria")
    gen_tax, cur = {}, {}
    for gi_, g in enumerate(genera):
        for lvl, (rank, per, stem) in enumerate((("phylum", 64, "Phy"), ("class", 32, "Cls"), ("order", 16, "Ord"), ("family", 4, "Fam"))):
            key = (lvl, gi_ // per)
            if key not in cur:
                parent = dom if lvl == 0 else cur[(lvl - 1, gi_ // (64, 32, 16, 4)[lvl - 1])]
                cur[key] = add(parent, rank, "%s%s" % (stem, "abcdefghijklmnopqrstuvwxyz"[(gi_ // per) % 26] * (1 + (gi_ // per) // 26)))
        gen_tax[g] = add(cur[(3, gi_ // 4)], "genus", g)
    sp_tax = {s: add(gen_tax[s.split(" ")[0]], "species", s) for s in species}
    tdir = tempfile.mkdtemp(prefix="pgx_cfg1_")
    tax = os.path.join(tdir, "Tax_class")
    os.makedirs(tax)
    nodes.sort()
    with open(os.path.join(tax, "nodes.dmp"), "w") as f:
        for t, p, r in nodes:
            f.write("%d\t|\t%d\t|\t%s\t|\t\t|\t0\t|\t0\t|\t11\t|\t0\t|\t0\t|\t0\t|\t0\t|\t0\t|\t\t|\n" % (t, p, r))
    with open(os.path.join(tax, "names.dmp"), "w") as f:
        for t in sorted(names):
            f.write("%d\t|\t%s\t|\t\t|\tscientific name\t|\n" % (t, names[t]))
    with open(os.path.join(tax, "gi_taxid_nucl.dmp"), "w") as f:
        for i, (_, s) in enumerate(orgs):
            f.write("%d\t%d\n" % (1000 + i, sp_tax[s]))
    # ---- sequences, gi-relabelled; the first n_q are the queries
    fa = os.path.join(tdir, "rdp373_gi.fa")
    with open(fa, "w") as f:
        for i, (h, s) in enumerate(recs):
            f.write(">gi|%d|rdp|%s| %s\n" % (1000 + i, h.split(" ", 1)[0], h.split(" ", 1)[1]))
            for k in range(0, len(s), 80):
                f.write(s[k:k + 80] + "\n")
    qfa = os.path.join(tdir, "queries.fa")
    with open(qfa, "w") as f:
        for i, (h, s) in enumerate(recs[:n_q]):
            f.write(">q%d_%s\n%s\n" % (i, h.split(" ", 1)[0], s))
    # ---- RDP stream of the queries: the truth lineage with rank drop-outs and a few wrong genera
    parent = {t: p for t, p, _ in nodes}
    rank = {t: r for t, _, r in nodes}
    rdp = os.path.join(tdir, "rdp.txt")
    with open(rdp, "w") as f:
        for i, (h, s) in enumerate(recs[:n_q]):
            t = sp_tax[orgs[i][1]]
            chain = []
            while t != 1:
                chain.append(t)
                t = parent[t]
            chain.reverse()
            cols = []
            for t in chain[:-1]:  # domain .. genus
                r = {"superkingdom": "domain"}.get(rank[t], rank[t])
                if (i * 7 + t) % 10 == 0:
                    continue      # a rank the classifier did not report
                nm = names[t]
                if r == "genus" and i % 6 == 5:
                    nm = genera[(genera.index(nm) + 1) % len(genera)]  # a wrong genus
                cols += [('"%s"' % nm) if i % 4 == 0 else nm, r, "%.2f" % (0.5 + ((i + t) % 50) / 100.0)]
            f.write("q%d_%s\t\t\t\t\t%s\n" % (i, h.split(" ", 1)[0], "\t".join(cols)))
    # ---- the chain
    blast = os.path.join(tdir, "blast.tsv")
    subprocess.check_call([ORACLE, "blastn", "-query", qfa, "-db", fa, "-outfmt", "6", "-out", blast, "-num_threads", "8"])
    shutil.copy(REF_TAX, os.path.join(tax, "tax_class"))  # the Perl runs ./tax_class inside Tax_class/ (taxcollector:47,170)
    subprocess.check_call(["./tax_class", "-c"], cwd=tax, stdout=subprocess.DEVNULL)
    bclass = os.path.join(tdir, "blast_class.txt")
    p = subprocess.run(["perl", os.path.join(REF, "Tax_class", "NCBI-taxcollector-0.01.pl"), "-f", blast, "-o", bclass], cwd=tdir,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=7200)
    tc_log = p.stdout
    cons = os.path.join(tdir, "consensus.txt")
    p = subprocess.run(["perl", os.path.join(REF, "Consensus", "Consensus_BLAST_SOAP_RDP-1.1.pl"), "-b", bclass, "-r", rdp, "-o", cons],
                       cwd=tdir, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=3600)
    cons_log = p.stdout.replace(cons.encode(), b"<OUT>")
    # ---- store
    def gz(src, name):
        with open(src, "rb") as a, gzip.GzipFile(os.path.join(OUT, name), "wb", mtime=0) as b:
            shutil.copyfileobj(a, b)
    gz(fa, "rdp373_gi.fa.gz")
    for n in ("nodes.dmp", "names.dmp", "gi_taxid_nucl.dmp"):
        shutil.copy(os.path.join(tax, n), os.path.join(OUT, n))
    shutil.copy(qfa, os.path.join(OUT, "queries.fa"))
    shutil.copy(rdp, os.path.join(OUT, "rdp.txt"))
    shutil.copy(cons, os.path.join(OUT, "consensus.txt"))
    open(os.path.join(OUT, "consensus.stdout"), "wb").write(cons_log)
    sha = lambda pth: hashlib.sha256(open(pth, "rb").read()).hexdigest()  # noqa: E731
    meta = {"n_queries": n_q, "blast_rows": sum(1 for _ in open(blast)), "blast_sha256": sha(blast),
            "blast_class_rows": sum(1 for _ in open(bclass)), "blast_class_sha256": sha(bclass),
            "taxcollector_stdout_sha256": hashlib.sha256(tc_log).hexdigest(),
            "gapped_rows": sum(1 for l in open(blast) if l.split("\t")[5] != "0")}
    json.dump(meta, open(os.path.join(OUT, "meta.json"), "w"), indent=1)
    # a head of the tables for a human reader (and to debug a hash mismatch)
    open(os.path.join(OUT, "blast.head.tsv"), "w").write("".join(open(blast).readlines()[:200]))
    open(os.path.join(OUT, "blast_class.head.txt"), "w").write("".join(open(bclass).readlines()[:200]))
    print(json.dumps(meta))
    shutil.rmtree(tdir)


if __name__ == "__main__":
    main()
