"""SOAP-mode golden vectors, produced by the reference's closed `soap` / `2bwt-builder`
ELFs (Classify/Runsoap/soap2.21release).  Called from oracle/gen_goldens.py.

Reference data used: a 48-sequence subset of validation_dataset/rdp_download_373seqs.fa
(the validation set the reference ships; it carries IUPAC codes and near-duplicate 16S
sequences, which is what makes multi-hit and ambiguity behaviour observable).  Reads are
synthetic (seeded), planted with 0..3 substitutions on either strand, plus edge cases:
lengths 20..400, N runs, sequence ends, lower case, IUPAC read characters.

Outputs (tests/golden/soap/): ref.fa, reads.fa, out_r2.txt (-M 4 -r 2), out_r1.txt (-M 4 -r 1),
unmapped_r2.txt (-u).  The ~660 MB index files stay in the scratch directory.
"""
import os
import random
import shutil
import subprocess

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def rc(s):
    return "".join(COMP.get(c, "N") for c in reversed(s))


def read_fasta(p):
    seqs, name, buf = [], None, []
    for l in open(p):
        l = l.rstrip("\n")
        if l.startswith(">"):
            if name is not None:
                seqs.append((name, "".join(buf)))
            name, buf = l[1:], []
        else:
            buf.append(l)
    seqs.append((name, "".join(buf)))
    return seqs


def other(b, rng):
    return rng.choice([x for x in "ACGT" if x != b])


def mutate(s, positions, rng):
    x = list(s)
    for p in positions:
        x[p] = other(x[p], rng)
    return "".join(x)


def generate(scratch, gold, soap_dir, ref_root="/root/reference"):
    out = os.path.join(gold, "soap")
    os.makedirs(out, exist_ok=True)
    rng = random.Random(20261003)
    allseq = read_fasta(os.path.join(ref_root, "validation_dataset", "rdp_download_373seqs.fa"))
    byid = {n.split()[0]: i for i, (n, _) in enumerate(allseq)}
    want = ["S000469148", "S000414323", "S000469450", "S000385166", "S000414322", "S000384778", "S000433096",
            "S000436258", "S000388840", "S000870881"]
    idx = [byid[w] for w in want]
    for i in range(len(allseq)):
        if len(idx) >= 48:
            break
        if i not in idx:
            idx.append(i)
    idx.sort()
    ref = [allseq[i] for i in idx]
    work = os.path.join(scratch, "soap")
    os.makedirs(work, exist_ok=True)
    with open(os.path.join(work, "ref.fa"), "w") as f:
        for n, s in ref:
            f.write(">" + n + "\n")
            for k in range(0, len(s), 80):
                f.write(s[k:k + 80] + "\n")
    # what the aligner sees: upper case, every non-ACGT letter read as G (observed)
    clean = [(n.split()[0], "".join(c if c in "ACGT" else "G" for c in s.upper())) for n, s in ref]

    reads = []
    for r in range(320):
        si = rng.randrange(len(clean))
        name, s = clean[si]
        L = 150
        off = rng.randrange(0, len(s) - L)
        w = s[off:off + L]
        k = rng.choice([0, 0, 1, 1, 2, 2, 2, 3])
        w = mutate(w, rng.sample(range(L), k), rng)
        strand = rng.choice("+-")
        reads.append((f"r{r}_{name}_{off + 1}_{strand}_{k}", w if strand == "+" else rc(w)))
    name, s = clean[9]
    for L in (20, 26, 27, 30, 40, 47, 48, 60, 100, 149, 151, 200, 255, 256, 257, 300, 400):
        w = s[200:200 + L]
        reads.append((f"len{L}", w))
        reads.append((f"len{L}_mm2", mutate(w, [3, L - 4], rng)))
        reads.append((f"len{L}_mm2_rc", rc(mutate(w, [3, L - 4], rng))))
    w = s[300:450]
    gpos = [i for i, c in enumerate(w) if c == "G"]
    for k in (1, 2, 4, 5, 6, 7):
        x = list(w)
        for p in gpos[:k]:
            x[p] = "N"
        reads.append((f"N_on_G_{k}", "".join(x)))
    reads.append(("N_mm_50", w[:50] + "N" + w[51:]))
    reads.append(("N_mm_50_100", w[:50] + "N" + w[51:100] + "N" + w[101:]))
    reads.append(("N_mm_rc", rc(w[:50] + "N" + w[51:])))
    reads.append(("iupac_read", w[:60] + "R" + w[61:90] + "y" + w[91:]))
    reads.append(("lower_case", w.lower()))
    reads.append(("name with blanks", w))
    for o in (0, 1, 136, 137, 148, 149):
        m = mutate(w, [o], rng)
        reads.append((f"plus_o{o}", m))
        reads.append((f"minus_o{o}", rc(m)))
    for a, b in ((0, 1), (0, 140), (5, 60), (80, 145), (100, 136), (100, 137), (140, 145), (74, 75)):
        m = mutate(w, [a, b], rng)
        reads.append((f"plus_{a}_{b}", m))
        reads.append((f"minus_{a}_{b}", rc(m)))
    for k in (0, 1, 2):
        reads.append((f"end_minus{k}", s[len(s) - 150 - k:len(s) - k]))
        reads.append((f"end_minus{k}_rc", rc(s[len(s) - 150 - k:len(s) - k])))
    reads.append(("seq_start", s[:150]))
    reads.append(("seq_start_rc", rc(s[:150])))
    reads.append(("span_two_seqs", clean[9][1][-75:] + clean[10][1][:75]))
    reads.append(("random_read", "".join(rng.choice("ACGT") for _ in range(150))))
    with open(os.path.join(work, "reads.fa"), "w") as f:
        for n, s_ in reads:
            f.write(f">{n}\n{s_}\n")

    builder = os.path.join(soap_dir, "2bwt-builder")
    soap = os.path.join(soap_dir, "soap")
    subprocess.run([builder, "ref.fa"], cwd=work, check=True, timeout=600, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    for r, tag in (("2", "r2"), ("1", "r1")):
        subprocess.run([soap, "-a", "reads.fa", "-D", "ref.fa.index", "-o", f"out_{tag}.txt", "-u",
                        f"unmapped_{tag}.txt", "-p", "1", "-M", "4", "-r", r], cwd=work, check=True, timeout=600,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # -M 0 / 1 / 2 (soap.man:73-82) on the reads of at most 256 bases: for longer reads the mode applies to the first
    # 256 bases only (-l, soap.man:60-63), which is not restated
    short = [(n, s_) for n, s_ in reads if len(s_) <= 256]
    with open(os.path.join(work, "reads_short.fa"), "w") as f:
        for n, s_ in short:
            f.write(f">{n}\n{s_}\n")
    for m in ("0", "1", "2"):
        subprocess.run([soap, "-a", "reads_short.fa", "-D", "ref.fa.index", "-o", f"out_M{m}.txt", "-u", f"unmapped_M{m}.txt", "-p", "1",
                        "-M", m, "-r", "2"], cwd=work, check=True, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([soap, "-a", "reads_short.fa", "-D", "ref.fa.index", "-o", "out_t.txt", "-p", "1", "-M", "4", "-r", "2", "-t"], cwd=work,
                   check=True, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)  # -t: ordinals instead of names
    import gzip
    for n in ("ref.fa", "reads.fa", "out_r2.txt", "out_r1.txt", "unmapped_r2.txt", "reads_short.fa", "unmapped_M0.txt", "unmapped_M1.txt",
              "unmapped_M2.txt"):
        shutil.copy(os.path.join(work, n), os.path.join(out, n))
    for tag in ("M0", "M1", "M2", "t"):
        with open(os.path.join(work, f"out_{tag}.txt"), "rb") as a, gzip.GzipFile(os.path.join(out, f"out_{tag}.txt.gz"), "wb", mtime=0) as b:
            shutil.copyfileobj(a, b)
    rows = sum(1 for _ in open(os.path.join(out, "out_r2.txt")))
    print(f"soap: {len(reads)} reads, {rows} rows (-r 2)")


def generate_pe(scratch, gold, soap_dir):
    """Paired-end goldens (soap.man:29-50: -b, -2, -m, -x), produced by the reference's closed ELF against the reference of
    the single-end goldens (tests/golden/soap/ref.fa).  Pairs are cut from its sequences: inserts around the [-m, -x]
    boundaries, mates of unequal length (the ELF measures the insert with the A mate's length), 0-3 planted mismatches
    per mate, A on either strand, same-strand and outward pairs (never paired), a mate that maps nowhere, a mate with more
    than -n N's, mates of 27-120 bases with two mismatches at swept offsets (the order of the two entries of a row; the
    ELF's blind spot at 32 bases).  Mates under 27 bases are left out: the ELF crashes on some of them.
    Outputs (tests/golden/soap/): pe_a.fa, pe_b.fa, pe_{paired,unpaired,unmapped}_r{2,0}.txt[.gz] (-m 400 -x 600) and, for the
    length sweep, pe_sweep_{a,b}.fa + pe_sweep_{paired,unpaired,unmapped}.txt.gz (-m 300 -x 700 -r 2).
    usage: python3 oracle/gen_goldens_soap.py pe"""
    import gzip
    out = os.path.join(gold, "soap")
    work = os.path.join(scratch, "soap_pe")
    os.makedirs(work, exist_ok=True)
    shutil.copy(os.path.join(out, "ref.fa"), os.path.join(work, "ref.fa"))
    rng = random.Random(20261005)
    seqs = [(n.split()[0], "".join(c if c in "ACGT" else "G" for c in s.upper())) for n, s in read_fasta(os.path.join(work, "ref.fa"))]

    def frag(ins):
        while True:
            n, s = rng.choice(seqs)
            if len(s) < ins + 20:
                continue
            o = rng.randrange(0, len(s) - ins)
            return n, o, s[o:o + ins]

    def planted(s, k):
        return mutate(s, rng.sample(range(2, len(s) - 2), k), rng)
    pairs = []
    for rep in range(2):
        for ins in (399, 400, 401, 500, 599, 600, 601, 650):
            for ma, mb in ((0, 0), (1, 0), (0, 2), (2, 2), (1, 1)):
                n, o, f = frag(ins)
                pairs.append((f"FR_i{ins}_a{ma}_b{mb}_{n}_{o}", planted(f[:60], ma), planted(rc(f[-60:]), mb)))
        for ll, lr in ((40, 100), (100, 40), (30, 90), (90, 30)):
            for ins in (380, 400, 440, 460, 560, 600, 620, 660):
                for swap in (0, 1):
                    n, o, f = frag(ins)
                    left, right = f[:ll], rc(f[-lr:])
                    pairs.append((f"len{ll}_{lr}_i{ins}_sw{swap}_{n}_{o}", right if swap else left, left if swap else right))
        for ins in (450, 550):
            n, o, f = frag(ins)
            pairs.append((f"outward_i{ins}_{n}_{o}", rc(f[:60]), f[-60:]))
            n, o, f = frag(ins)
            pairs.append((f"same_strand_i{ins}_{n}_{o}", f[:60], f[-60:]))
            n, o, f = frag(ins)
            pairs.append((f"a_three_mismatches_i{ins}_{n}_{o}", planted(f[:60], 3), rc(f[-60:])))
            n, o, f = frag(ins)
            pairs.append((f"b_three_mismatches_i{ins}_{n}_{o}", f[:60], planted(rc(f[-60:]), 3)))
            n, o, f = frag(ins)
            pairs.append((f"b_random_i{ins}_{n}_{o}", f[:60], "".join(rng.choice("ACGT") for _ in range(60))))
            n, o, f = frag(ins)
            m2 = rc(f[-60:])
            pairs.append((f"b_seven_N_i{ins}_{n}_{o}", f[:60], m2[:10] + "NNNNNNN" + m2[17:]))
            pairs.append((f"b_two_N_i{ins}_{n}_{o}", f[:60], m2[:10] + "N" + m2[11:30] + "n" + m2[31:]))
    with open(os.path.join(work, "pe_a.fa"), "w") as fa, open(os.path.join(work, "pe_b.fa"), "w") as fb:
        for k, (tag, m1, m2) in enumerate(pairs):
            fa.write(f">p{k}_{tag}/1\n{m1}\n")
            fb.write(f">p{k}_{tag}/2\n{m2}\n")
    sweep = []
    for L in (27, 28, 30, 31, 32, 33, 36, 38, 39, 40, 41, 44, 48, 50, 60, 64, 75, 90, 100, 120):
        for mx in sorted(set(range(max(6, L // 3), L - 1, 1 if L <= 44 else 3)) | {2 * (L // 3) - 1, 2 * (L // 3)}):
            n, o, f = frag(500)
            m2 = list(f[-L:])
            lo = rng.randrange(1, mx - 1)
            for p in (lo, mx):
                m2[p] = other(m2[p], rng)
            sweep.append((f"L{L}_lo{lo}_mx{mx}", f[:L], rc("".join(m2))))
    with open(os.path.join(work, "pe_sweep_a.fa"), "w") as fa, open(os.path.join(work, "pe_sweep_b.fa"), "w") as fb:
        for k, (tag, m1, m2) in enumerate(sweep):
            fa.write(f">w{k}_{tag}/1\n{m1}\n")
            fb.write(f">w{k}_{tag}/2\n{m2}\n")
    builder, soap = os.path.join(soap_dir, "2bwt-builder"), os.path.join(soap_dir, "soap")
    if not os.path.exists(os.path.join(work, "ref.fa.index.bwt")):
        subprocess.run([builder, "ref.fa"], cwd=work, check=True, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for r in ("2", "0"):
        subprocess.run([soap, "-a", "pe_a.fa", "-b", "pe_b.fa", "-D", "ref.fa.index", "-o", f"pe_paired_r{r}.txt", "-2", f"pe_unpaired_r{r}.txt",
                        "-u", f"pe_unmapped_r{r}.txt", "-m", "400", "-x", "600", "-p", "1", "-M", "4", "-r", r], cwd=work, check=True,
                       timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([soap, "-a", "pe_sweep_a.fa", "-b", "pe_sweep_b.fa", "-D", "ref.fa.index", "-o", "pe_sweep_paired.txt", "-2",
                    "pe_sweep_unpaired.txt", "-u", "pe_sweep_unmapped.txt", "-m", "300", "-x", "700", "-p", "1", "-M", "4", "-r", "2"], cwd=work,
                   check=True, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for n in ("pe_a.fa", "pe_b.fa", "pe_sweep_a.fa", "pe_sweep_b.fa"):
        shutil.copy(os.path.join(work, n), os.path.join(out, n))
    for n in ["pe_%s_r%s.txt" % (k, r) for k in ("paired", "unpaired", "unmapped") for r in ("2", "0")] + \
             ["pe_sweep_%s.txt" % k for k in ("paired", "unpaired", "unmapped")]:
        with open(os.path.join(work, n), "rb") as a, gzip.GzipFile(os.path.join(out, n + ".gz"), "wb", mtime=0) as b:
            shutil.copyfileobj(a, b)
    print("soap paired-end: %d + %d pairs; rows: %s" % (len(pairs), len(sweep), {n: sum(1 for _ in open(os.path.join(work, n)))
                                                                                     for n in ("pe_paired_r2.txt", "pe_unpaired_r2.txt", "pe_sweep_paired.txt")}))


if __name__ == "__main__":
    import sys
    import tempfile
    HERE = os.path.dirname(os.path.abspath(__file__))
    if sys.argv[1:] == ["pe"]:
        with tempfile.TemporaryDirectory(prefix="pgx_soap_pe_", dir="/tmp") as scratch:
            generate_pe(scratch, os.path.join(HERE, "..", "tests", "golden"), "/root/reference/Classify/Runsoap/soap2.21release")
    else:
        sys.exit("usage: python3 oracle/gen_goldens_soap.py pe   (the single-end goldens: oracle/gen_goldens.py soap)")
