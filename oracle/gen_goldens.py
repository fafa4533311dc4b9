#!/usr/bin/env python3
"""Generate tests/golden/ by running the REFERENCE's own code in this container.

oracle/ is test infrastructure.  This script is the committed recipe behind every
fixture under tests/golden/{tax,taxcollect,consensus,soap}: it feeds small synthetic
inputs to

  * oracle/_ref/tax_class        (gcc build of /root/reference/Tax_class/ncbitc.c,
                                  made by oracle/Makefile, Tax_class/Makefile:2)
  * /root/reference/Tax_class/NCBI-taxcollector-0.01.pl          (perl 5.34)
  * /root/reference/Consensus/Consensus_BLAST_SOAP_RDP-1.1.pl    (perl 5.34)
  * /root/reference/Classify/Runsoap/soap2.21release/{2bwt-builder,soap}  (closed ELF)

and stores inputs + the bytes they print.  Only data is written to the repo; no
reference source or binary is copied.  All runs happen in a scratch directory and
under `timeout` (two reference inputs never terminate).

Usage:  python3 oracle/gen_goldens.py [tax] [taxcollect] [consensus] [soap]
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("PGX_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
REF_TAX = os.path.join(HERE, "_ref", "tax_class")
PERL_TAXCOL = os.path.join(REF, "Tax_class", "NCBI-taxcollector-0.01.pl")
PERL_CONS = os.path.join(REF, "Consensus", "Consensus_BLAST_SOAP_RDP-1.1.pl")
SOAP_DIR = os.path.join(REF, "Classify", "Runsoap", "soap2.21release")

# ----------------------------------------------------------------------------- mini taxonomy
# (taxid, parent, rank, embl)
NODES = [
    (1, 1, "no rank", ""), (2, 13, "superkingdom", ""), (13, 1, "no rank", ""),
    (20, 2, "phylum", ""), (21, 20, "class", ""), (22, 21, "order", ""), (23, 22, "family", ""),
    (24, 23, "genus", ""), (25, 24, "species", "BS"), (26, 25, "no rank", ""),
    (30, 13, "superkingdom", ""), (31, 30, "no rank", ""), (32, 31, "kingdom", ""),
    (33, 32, "phylum", ""), (34, 33, "genus", ""), (35, 34, "species", "BT"),
    (40, 1, "superkingdom", ""), (41, 40, "family", ""), (42, 41, "genus", ""), (43, 42, "species", "HC"),
    (50, 1, "no rank", ""), (51, 50, "no rank", ""), (52, 51, "species", ""),
    (60, 2, "no rank", ""), (61, 60, "species", ""), (62, 2, "species", ""),
    (70, 2, "phylum", ""), (71, 70, "genus", ""), (72, 71, "species", ""), (73, 72, "subspecies", ""),
    (80, 20, "class", ""), (81, 80, "species", ""),
    (90, 1, "no rank", ""),
    (100, 24, "species", "X"),
    (110, 24, "species group", ""), (111, 110, "species", ""),
    (120, 21, "order", ""), (121, 120, "family", ""), (122, 121, "genus", ""), (123, 122, "species", ""),
    (124, 120, "species", ""),
    (130, 23, "tribe", ""), (131, 130, "genus", ""), (132, 131, "varietas", ""),
    (140, 2, "made up rank", ""),
    (3000, 2, "species", ""),
]
NAMES = {
    1: [("all", "", "synonym"), ("root", "", "scientific name")],
    2: [("Bacteria", "Bacteria <prokaryote>", "scientific name"), ("eubacteria", "", "genbank common name")],
    13: [("cellular organisms", "", "scientific name")],
    20: [("Firmicutes", "", "scientific name")], 21: [("Bacilli", "", "scientific name")],
    22: [("Bacillales", "", "scientific name")], 23: [("Bacillaceae", "", "scientific name")],
    24: [("Bacillus", "Bacillus <bacterium>", "scientific name")],
    25: [("Vibrio subtilis", "", "synonym"), ("Bacillus subtilis", "", "scientific name")],
    26: [("Bacillus subtilis str. X1", "", "scientific name")],
    30: [("Eukaryota", "", "scientific name")], 31: [("Opisthokonta", "", "scientific name")],
    32: [("Metazoa", "", "scientific name")], 33: [("Chordata", "", "scientific name")],
    34: [("Bos", "", "scientific name")],
    35: [("bovine", "", "common name"), ("Bos taurus", "", "scientific name"), ("cattle", "", "genbank common name")],
    40: [("Viruses", "", "scientific name")], 41: [("Flaviviridae", "", "scientific name")],
    42: [("Hepacivirus", "", "scientific name")], 43: [("Hepatitis C virus", "", "scientific name")],
    50: [("unclassified sequences", "", "scientific name")], 51: [("metagenomes", "", "scientific name")],
    52: [("soil metagenome", "", "scientific name")],
    60: [("environmental samples", "environmental samples <bacteria>", "scientific name")],
    61: [("uncultured bacterium", "", "scientific name")],
    62: [("Bacterium sp. direct", "", "scientific name")],
    70: [("Bar7 division", "", "scientific name")], 71: [("Candidatus Foo6", "", "scientific name")],
    72: [("Candidatus Foo6 bar", "", "scientific name")],
    73: [("Candidatus Foo6 bar subsp. baz", "", "scientific name")],
    80: [("Class5X", "", "scientific name")], 81: [("Lonely species", "", "scientific name")],
    90: [("other sequences", "", "scientific name")],
    100: [("Bacillus sp. with an extraordinarily long strain designation ABCDEFGHIJ-0123456789-XYZ", "", "scientific name")],
    110: [("Bacillus cereus group", "", "scientific name")], 111: [("Bacillus cereus", "", "scientific name")],
    120: [("Order6ales", "", "scientific name")], 121: [("Famaceae", "", "scientific name")],
    122: [("Gen", "", "scientific name")], 123: [("Gen sp", "", "scientific name")],
    124: [("Direct sp", "", "scientific name")],
    130: [("Tribeae", "", "scientific name")], 131: [("Tribogenus", "", "scientific name")],
    132: [("Tribogenus var. x", "", "scientific name")],
    140: [("Oddity", "", "scientific name")],
    3000: [("zzz last record", "", "scientific name")],
}
GIS = [(1, 25), (2, 25), (5, 25), (7, 35), (9, 61), (10, 62), (11, 26), (12, 73), (15, 24), (16, 72),
       (21, 52), (22, 43), (23, 81), (24, 100), (28, 111), (29, 123), (31, 124), (33, 3000),
       (34, 132), (40, 90), (41, 999), (42, 140)]


def write_dumps(d):
    with open(os.path.join(d, "nodes.dmp"), "w") as f:
        for t, p, r, e in sorted(NODES):
            f.write(f"{t}\t|\t{p}\t|\t{r}\t|\t{e}\t|\t0\t|\t1\t|\t11\t|\t1\t|\t0\t|\t1\t|\t0\t|\t0\t|\t\t|\n")
    with open(os.path.join(d, "names.dmp"), "w") as f:
        for t in sorted(NAMES):
            for n, u, c in NAMES[t]:
                f.write(f"{t}\t|\t{n}\t|\t{u}\t|\t{c}\t|\n")
    with open(os.path.join(d, "gi_taxid_nucl.dmp"), "w") as f:
        for g, t in GIS:
            f.write(f"{g}\t{t}\n")


def run(cmd, cwd, timeout=60, stdin=None):
    p = subprocess.run(cmd, cwd=cwd, timeout=timeout, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout, p.stderr


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def make_taxdir(scratch):
    """scratch/Tax_class with dumps, reference binaries and the reference tax_class."""
    td = os.path.join(scratch, "Tax_class")
    os.makedirs(td, exist_ok=True)
    write_dumps(td)
    shutil.copy(REF_TAX, os.path.join(td, "tax_class"))
    rc, _, _ = run(["./tax_class", "-c"], td)
    assert rc == 0
    return td


# ----------------------------------------------------------------------------- tax goldens
TAX_CLI_CASES = (
    [["-s", str(g)] for g in (1, 2, 5, 7, 9, 10, 11, 12, 15, 16, 21, 22, 23, 24, 28, 29, 31, 33, 34, 40, 41, 42,
                               3, 20, 1000, 0, -5)]
    + [["-g", str(g)] for g in (1, 7, 20, 40, 41, 0)]
    + [["-t", str(t)] for t in (1, 2, 25, 35, 100, 140, 999, 3000, 5000, 0)]
    + [["-n", str(t)] for t in (1, 2, 25, 35, 100, 3000, 999, 4, 13, 124, 140)]
    + [[], ["-h"], ["-x"], ["-s"], ["-v", "-s", "5"], ["-v", "-t", "35"], ["-v", "-g", "7"], ["--search", "7"],
       ["--search-node", "25"], ["--search-name", "35"], ["--search-gi", "9"], ["-s", "7", "-t", "2"], ["-s", "12abc"]]
)


def gen_tax(scratch):
    out = os.path.join(GOLD, "tax")
    os.makedirs(out, exist_ok=True)
    td = make_taxdir(scratch)
    for n in ("nodes.dmp", "names.dmp", "gi_taxid_nucl.dmp"):
        shutil.copy(os.path.join(td, n), os.path.join(out, n))
    cases = []
    for args in TAX_CLI_CASES:
        rc, so, se = run(["./tax_class"] + args, td)
        # getopt's own diagnostics carry argv[0]; keep stderr only as a presence flag
        cases.append({"args": args, "rc": rc, "stdout": so.decode("latin-1"), "stderr_nonempty": bool(se)})
    # names.dmp.bin carries uninitialised stack bytes after each NUL (ncbitc.c:803), so it is
    # pinned field-wise: every record as the three C strings.
    nb = open(os.path.join(td, "names.dmp.bin"), "rb").read()
    cnt = int.from_bytes(nb[:4], "little")
    recs = []
    for i in range(cnt):
        r = nb[4 + 196 * i: 4 + 196 * (i + 1)]
        recs.append([int.from_bytes(r[:4], "little", signed=True)] +
                    [r[4 + 64 * k: 68 + 64 * k].split(b"\0")[0].decode("latin-1") for k in range(3)])
    meta = {
        "sha256": {n: sha(os.path.join(td, n)) for n in ("gi_taxid_nucl.dmp.bin", "nodes.dmp.bin")},
        "sizes": {n: os.path.getsize(os.path.join(td, n)) for n in
                  ("gi_taxid_nucl.dmp.bin", "nodes.dmp.bin", "names.dmp.bin")},
        "names_records": recs,
        "cli": cases,
    }
    json.dump(meta, open(os.path.join(out, "tax_class_golden.json"), "w"), indent=1)
    print("tax:", len(cases), "CLI cases")


# ----------------------------------------------------------------------------- taxcollector goldens
def hit(q, gi, db="gb", acc="ACC1.1", rest="99.33\t150\t1\t0\t1\t150\t11\t160\t2e-70\t 270"):
    return f"{q}\tgi|{gi}|{db}|{acc}|\t{rest}\n"


TAXCOL_INPUTS = {
    "basic": "".join([
        hit("q1", 5), hit("q1", 7, "ref", "NM_1.1", "95.00\t100\t5\t0\t1\t100\t200\t101\t1e-40\t 170"),
        hit("q2", 9, rest="90.00\t150\t15\t0\t1\t150\t1\t150\t3e-50\t 195"),
        hit("q3", 20, rest="95.00\t100\t5\t0\t1\t100\t1\t100\t1e-40\t 170"),
        hit("q4", 10, rest="88.00\t50\t6\t0\t1\t50\t1\t50\t1e-10\t87.9"),
        hit("q5", 11), hit("q6 desc", 12), hit("q7", 15), hit("q8", 21), hit("q9", 22), hit("q10", 12),
        hit("q11", 16), hit("q12", 23), hit("q13", 24), hit("q14", 28), hit("q15", 29), hit("q16", 31),
        hit("q17", 33), hit("q18", 34), hit("q19", 26), hit("q20", 27), hit("q21", 56), hit("q22", 1000),
        hit("q23", 42), hit("q24", 1), hit("q25", 2),
    ]),
    # SOAP-style row (13 columns, gi-labelled reference name in column 8)
    "soap_row": "r1\tACGTACGT\thhhhhhhh\t1\ta\t150\t-\tgi|5|gb|AAA.1|\t99\t1\tG->18T40\t150M\t18G131\n",
    # double tab between id and subject (older BLAST wrappers), extra spaces in bitscore column
    "tabs_spaces": "q1\t\tgi|7|gb|A|\t81.87\t1186\t148\t64\t226\t1375\t128\t1282\t0.0\t  937\n"
                   "q2\tgi|9|gb|B|\t78.63\t1535\t218\t99\t1\t1474\t1\t1486\t0.0\t 917\n",
    # an empty line ends the program (taxcollector:83-87)
    "blank_line_stops": hit("q1", 5) + "\n" + hit("q2", 7),
    "no_trailing_newline": hit("q1", 5) + hit("q2", 7).rstrip("\n"),
}


def gen_taxcollect(scratch):
    out = os.path.join(GOLD, "taxcollect")
    os.makedirs(out, exist_ok=True)
    make_taxdir(scratch)
    for name, text in TAXCOL_INPUTS.items():
        inp = os.path.join(scratch, name + ".in.tsv")
        open(inp, "w").write(text)
        outp = os.path.join(scratch, name + ".out.tsv")
        rc, so, se = run(["perl", PERL_TAXCOL, "-f", inp, "-o", outp], scratch, timeout=120)
        assert rc == 0, (name, rc, se[:300])
        shutil.copy(inp, os.path.join(out, name + ".in.tsv"))
        shutil.copy(outp, os.path.join(out, name + ".out.tsv"))
        open(os.path.join(out, name + ".report.txt"), "wb").write(so)
    print("taxcollect:", len(TAXCOL_INPUTS), "cases")


# ----------------------------------------------------------------------------- consensus goldens
def rdp(rid, *trip):
    return rid + "\t\t\t\t\t" + "\t".join(trip) + "\n"


FULL = "[0]Bacteria;[1]Firmicutes;[2]Bacilli;[3]Bacillales;[4]Bacillaceae;[5]Bacillus;[6]Bacillus_subtilis;"
BOS = "[0]Eukaryota;[9]Metazoa;[1]Chordata;[5]Bos;[6]Bos_taurus;"
UNC = "[0]Bacteria;[5]uncultured_bacterium;[6]uncultured_bacterium;"
NUM = "150\t1\t0\t1\t150\t11\t160\t2e-70\t270"
RDP_BAC = ("Bacteria", "domain", "1.0", '"Firmicutes"', "phylum", "1.0", '"Bacilli"', "class", "1.0",
           "Bacillales", "order", "1.0", '"Bacillaceae 1"', "family", "0.95", "Bacillus", "genus", "0.9")


def bl(q, lin, sim, num=NUM, sep="\t"):
    return f"{q}\t{lin}{sep}{sim}\t{num}\n"


CONS_CASES = {
    "basic": (
        bl("q1", FULL, "99.33") + bl("q1", BOS, "95.00") + bl("q2", UNC, "90.00") + bl("q2", FULL, "100.00")
        + bl("q3", "Unidentified(GI:20);", "95.00") + bl("q4", FULL, "88.00"),
        rdp("q1", *RDP_BAC) + rdp("q2", "Bacteria", "domain", "1.0") + rdp("q3", *RDP_BAC) + rdp("q4", *RDP_BAC[:9]),
    ),
    # string comparisons: "14" gt "6" is false, "90.00" lt "100.00" is false -> first hit kept
    "string_compare_first_kept": (
        bl("q1", UNC, "90.00") + bl("q1", FULL, "100.00"),
        rdp("q1", "Bacteria", "domain", "1.0"),
    ),
    # token counts 4 -> 6 -> 10 with sims 95 -> 96 -> 94: the 6-token hit wins
    "string_compare_counts": (
        bl("q1", "[0]Bacteria;[1]Firmicutes;", "95.00") + bl("q1", "[0]Bacteria;[1]Firmicutes;[2]Bacilli;", "96.00")
        + bl("q1", "[0]Bacteria;[1]Firmicutes;[2]Bacilli;[3]Bacillales;[4]Bacillaceae;", "94.00"),
        rdp("q1", "Bacteria", "domain", "1.0"),
    ),
    # README-style double tab after the lineage column, kept verbatim in the output
    "double_tab": (
        bl("S1", FULL, "92.61", "1435\t81\t23\t29\t1452\t1\t1421\t0.0\t2039", sep="\t\t")
        + bl("S1", BOS, "93.00", "1435\t81\t23\t29\t1452\t1\t1421\t0.0\t2039", sep="\t\t")
        + bl("S2", UNC, "91.78", "1435\t90\t26\t49\t1469\t1\t1421\t0.0\t1971", sep="\t\t"),
        rdp("S1", *RDP_BAC) + rdp("S2", *RDP_BAC),
    ),
    # a BLAST-only read in the middle is skipped with a stdout note
    "blast_only_read": (
        bl("q1", FULL, "99.00") + bl("qX", BOS, "97.00") + bl("qX", FULL, "91.00") + bl("q2", FULL, "98.00"),
        rdp("q1", *RDP_BAC) + rdp("q2", *RDP_BAC),
    ),
    # first RDP read has no BLAST line at the cursor while `found` is still undef: silently consumed
    "first_rdp_unmatched": (
        bl("q2", FULL, "99.00") + bl("q3", BOS, "97.00"),
        rdp("q1", *RDP_BAC) + rdp("q2", *RDP_BAC) + rdp("q3", "Eukaryota", "domain", "1.0"),
    ),
    # names with blanks in non-species ranks shift the (rank,name) pairing (Consensus:122)
    "space_in_name": (
        bl("q1", "[0]Bacteria;[1]Bar9 division;[5]Candidatus_Foo6;[6]Candidatus_Foo6_bar;", "88.00")
        + bl("q1", "[0]Bacteria;[1]Firmicutes;", "87.00"),
        rdp("q1", "Bacteria", "domain", "1.0", "Firmicutes", "phylum", "0.8", "division", "class", "0.5"),
    ),
    # >= 10 matches: "10" gt "9" is false as strings
    "ten_vs_nine": (
        bl("q1", "[0]Bacteria;" * 9, "90.00") + bl("q1", "[0]Bacteria;" * 10, "91.00") + bl("q1", "[0]Bacteria;" * 2, "99.00"),
        rdp("q1", "Bacteria", "domain", "1.0"),
    ),
    # undef-equals-undef: unknown BLAST rank token vs unknown RDP rank name both index to undef
    "undef_ranks": (
        bl("q1", "[9]Metazoa;[0]Eukaryota;", "90.00") + bl("q1", "[0]Eukaryota;", "95.00"),
        rdp("q1", "Metazoa", "kingdom", "1.0", "Eukaryota", "domain", "1.0"),
    ),
    # RDP names cleaned of quotes, digits, blanks and punctuation; odd token count
    "rdp_cleaning": (
        bl("q1", "[0]Bacteria;[4]Bacillaceae;[5]", "90.00") + bl("q1", "[0]Bacteria;[4]Bacillaceae;", "90.00"),
        rdp("q1", '"Bacteria"', "domain", "1.0", '"Bacillaceae 1"', "family", "0.7", "12_3", "genus", "0.1"),
    ),
    "unidentified": (
        bl("q1", "Unidentified(GI:20);", "95.00"),
        rdp("q1", *RDP_BAC),
    ),
    "pident_string_order": (
        bl("q1", FULL, "9.50") + bl("q1", FULL, "10.00") + bl("q1", FULL, "100.00") + bl("q1", FULL, "99.99")
        + bl("q2", FULL, "100.00") + bl("q2", FULL, "99.99") + bl("q2", FULL, "9.5"),
        rdp("q1", *RDP_BAC) + rdp("q2", *RDP_BAC),
    ),
    "rdp_without_five_tabs": (
        bl("q1", FULL, "99.00") + bl("q2", FULL, "98.00"),
        "q1\n" + rdp("q2", *RDP_BAC),
    ),
}


def gen_consensus(scratch):
    out = os.path.join(GOLD, "consensus")
    os.makedirs(out, exist_ok=True)
    for name, (b, r) in CONS_CASES.items():
        bp = os.path.join(scratch, name + ".blast.tsv")
        rp = os.path.join(scratch, name + ".rdp.tsv")
        op = os.path.join(scratch, name + ".out.txt")
        open(bp, "w").write(b)
        open(rp, "w").write(r)
        rc, so, se = run(["perl", PERL_CONS, "-b", bp, "-r", rp, "-o", op], scratch, timeout=60)
        assert rc == 0, (name, rc)
        for src in (bp, rp, op):
            shutil.copy(src, os.path.join(out, os.path.basename(src)))
        # stdout echoes the absolute -o path on line 3; store it with a placeholder
        open(os.path.join(out, name + ".log.txt"), "wb").write(so.replace(op.encode(), b"@OUT@"))
    # -s is opened and never read (Consensus:40-46): same output with any SOAP file
    name = "basic"
    bp, rp = (os.path.join(scratch, name + e) for e in (".blast.tsv", ".rdp.tsv"))
    sp = os.path.join(scratch, "soap_any.txt")
    open(sp, "w").write("anything at all\n")
    op = os.path.join(scratch, "basic_with_s.out.txt")
    rc, so, se = run(["perl", PERL_CONS, "-b", bp, "-r", rp, "-s", sp, "-o", op], scratch, timeout=60)
    assert open(op, "rb").read() == open(os.path.join(out, "basic.out.txt"), "rb").read()
    print("consensus:", len(CONS_CASES), "cases (+ -s ignored check)")


def main():
    what = sys.argv[1:] or ["tax", "taxcollect", "consensus", "soap"]
    if not os.path.exists(REF_TAX):
        subprocess.check_call(["make", "-C", HERE, "ref"])
    scratch = tempfile.mkdtemp(prefix="pgx_gold_")
    try:
        if "tax" in what:
            gen_tax(scratch)
        if "taxcollect" in what:
            gen_taxcollect(scratch)
        if "consensus" in what:
            gen_consensus(scratch)
        if "soap" in what:
            import gen_goldens_soap
            gen_goldens_soap.generate(scratch, GOLD, SOAP_DIR)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    main()
