#!/usr/bin/env python3
"""Generate tests/golden/megaclust and tests/golden/megaclustable by running the REFERENCE's own Perl
(/root/reference/Megaclust/megaclust2.pl, /root/reference/Megaclustable/megaclustable.pl, perl 5.34) in this
container.  oracle/ is test infrastructure; only data (inputs + the bytes the reference printed) is written to
the repo.  Runs happen in a scratch directory, under `timeout`, with PERL_HASH_SEED=0: megaclust2 prints its
table in Perl hash order, so parity for its data lines is defined as a multiset of lines (the tests sort).

Usage: python3 oracle/gen_goldens_megaclust.py
"""
import json
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("PGX_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
MC2 = os.path.join(REF, "Megaclust", "megaclust2.pl")
MCT = os.path.join(REF, "Megaclustable", "megaclustable.pl")

L_BS = "[0]Bacteria;[1]Firmicutes;[2]Bacilli;[3]Bacillales;[4]Bacillaceae;[5]Bacillus;[6]Bacillus_subtilis;"
L_BC = "[0]Bacteria;[1]Firmicutes;[2]Bacilli;[3]Bacillales;[4]Bacillaceae;[5]Bacillus;[6]Bacillus_cereus;"
L_EC = "[0]Bacteria;[1]Proteobacteria;[2]Gammaproteobacteria;[3]Enterobacterales;[4]Enterobacteriaceae;[5]Escherichia;[6]Escherichia_coli;"
L_UB = "[0]Bacteria;[5]uncultured_bacterium;[6]uncultured_bacterium;"
L_AR = "[0]Archaea;[1]Euryarchaeota;[2]Methanobacteria;"
L_UN = "Unidentified(GI:20);"


def row(q, lin, pid="99.33", alen="150", mm="1", gap="0", qs="1", qe="150", ss="11", se="160", ev="2e-70", bits="270", sep="\t"):
    return q + "\t" + lin + sep + "\t".join([pid, alen, mm, gap, qs, qe, ss, se, ev, bits]) + "\n"


def consensus_text(rows):
    out = []
    for i, r in enumerate(rows):
        out.append(r)
        out.append("#Matches found: %d\n" % (i % 7))
    return "".join(out)


MEGACLUST_CASES = {
    # name: (input text, argv after -i/-o)
    "basic": (consensus_text([row("q1", L_BS), row("q2", L_UB, pid="90.00"), row("q3", L_UN, pid="95.00"),
                              row("q4", L_BS, pid="88.00"), row("q5", L_BS, pid="97.10"), row("q6", L_EC, pid="100.00")]), []),
    "readme_80": (consensus_text([row("q1", L_BS, pid="81.00", bits="120"), row("q2", L_UB, pid="79.99", bits="150"),
                                  row("q3", L_EC, pid="80.00", bits="100", ev="1e-20"), row("q4", L_EC, pid="80.00", bits="99.9"),
                                  row("q5", L_AR, pid="85.5", bits="101", ev="1e-19"), row("q6", L_AR, pid="99", bits="2.5e+03", ev="0.0")]),
                  ["-b", "100", "-s", "80", "-e", "1e-20"]),
    "edges_equal": (consensus_text([row("a", L_BS, pid="95.00", ev="1e-20", bits="200"), row("b", L_BS, pid="94.99"),
                                    row("c", L_BS, ev="1.1e-20"), row("d", L_BS, bits="199.9"), row("e", L_BC, pid="95", ev="1E-20", bits="2e2"),
                                    row("f", L_BC, pid=" 96.0", ev="1e-21", bits="+300")]), []),
    "dup_pairs": ("".join([row("q1", L_BS), row("q1", L_BS, pid="98.00"), row("q1", L_BC), row("q2", L_BS), row("q2", L_BS),
                           row("q3", L_EC), "#x\n", row("q3", L_EC)]), []),
    "count_all": ("".join([row("q1", L_BS), row("q1", L_BS, pid="98.00"), row("q1", L_BC), row("q2", L_BS), row("q2", L_BS),
                           row("q3", L_EC), row("q3", L_EC)]), ["-c", "1"]),
    "count_zero_is_off": ("".join([row("q1", L_BS), row("q1", L_BS)]), ["-c", "0"]),
    "double_tab_and_spaces": ("".join([row("S1", L_BS, sep="\t\t"), row("S2", L_UB, sep="\t "), row("S3", L_EC, sep=" \t"),
                                       "S4\t" + L_AR + "\t\t\t99.0\t150\t1\t0\t1\t150\t11\t160\t2e-70\t270\n",
                                       "S5 x\t" + L_AR + "\t99.0\t150\t1\t0\t1\t150\t11\t160\t2e-70\t270\n"]), []),
    "short_and_odd_lines": ("q1\t" + L_BS + "\t99.0\n" + "\n" + "justoneword\n" + row("q2", L_BS) + "# comment\n" + " #notcomment\t" + L_BC +
                            "\t99\t1\t1\t0\t1\t1\t1\t1\t0\t999\n" + row("q3", L_EC, pid="abc") + row("q4", L_EC, ev="xyz") +
                            row("q5", L_EC, bits="") + row("q6", L_EC, pid="1e2") + row("q7", L_EC, pid="0x64") + row("q8", L_EC, pid=".975e2") +
                            row("q9", L_EC, ev="-1") + row("q10", L_EC, bits="inf") + row("q11", L_EC, pid="nan"), []),
    "crlf": (row("q1", L_BS).replace("\n", "\r\n") + row("q2", L_BC).replace("\n", "\r\n") + row("q3", L_BC, bits="199").replace("\n", "\r\n"), []),
    "no_trailing_newline": (row("q1", L_BS) + row("q2", L_BC).rstrip("\n"), []),
    "delimiter": (row("q1", L_BS) + row("q2", L_BC), ["-d", ";"]),
    "delimiter_zero": (row("q1", L_BS) + row("q2", L_BC), ["-d", "0"]),
    "delimiter_tab": (row("q1", L_BS) + row("q2", L_BC), ["-d", "\t"]),
    "s_zero_is_default": (row("q1", L_BS, pid="50.00") + row("q2", L_BC), ["-s", "0"]),
    "s_fraction": (row("q1", L_BS, pid="97.49") + row("q2", L_BC, pid="97.50") + row("q3", L_EC, pid="97.51"), ["-s", "97.5"]),
    "s_out_of_range": (row("q1", L_BS), ["-s", "101"]),
    "s_negative": (row("q1", L_BS), ["-s", "-1"]),
    "e_zero_is_default": (row("q1", L_BS, ev="1e-19") + row("q2", L_BC, ev="1e-21"), ["-e", "0"]),
    "e_loose": (row("q1", L_BS, ev="1e-6") + row("q2", L_BC, ev="0.001") + row("q3", L_BC, ev="1e-5"), ["-e", "1e-5"]),
    "b_loose": (row("q1", L_BS, bits="55.4") + row("q2", L_BC, bits="55.3") + row("q3", L_BC, bits="1e2"), ["-b", "55.4"]),
    "empty_input": ("", []),
    "only_comments": ("#a\n#b\n", []),
    "comma_in_subject": (row("q1", "[0]Bacteria;[6]Foo,bar;") + row("q2", "[0]Bacteria;[6]Foo,bar;"), []),
}
MEGACLUST_ARGV_ONLY = {
    "help": ["-h"],
    "no_args": [],
    "missing_output": ["-i", "x.txt"],
    "missing_input_file": ["-i", "does_not_exist.txt", "-o", "out.txt"],
}

M1 = "OTU,times_hit\n" + L_BS + ",12\n" + L_BC + ",3\n" + L_EC + ",7\n" + L_UB + ",2\n" + L_UN + ",5\n"
M2 = "OTU,times_hit\n" + L_EC + ",1\n" + L_AR + ",4\n" + L_BS + ",10\n"
M3 = "OTU,times_hit\n" + L_AR + ",6\n[0]Eukaryota;[1]Chordata,9\n[0]Eukaryota,2\n" + "[0]Bacteria;[1]Firmicutes,1\r\n" + "[1]NoDomain;,3\n" + "garbage line\n" + "[0],8\n"
MEGACLUSTABLE_CASES = {
    # name: (files {name: text}, argv)
    "domain_three_files": ({"m1.csv": M1, "m2.csv": M2, "m3.csv": M3}, ["-m", "m1.csv", "m2.csv", "m3.csv", "-t", "0", "-o", "out.txt"]),
    "phylum": ({"m1.csv": M1, "m2.csv": M2, "m3.csv": M3}, ["-m", "m1.csv", "m2.csv", "m3.csv", "-t", "1", "-o", "out.txt"]),
    "species": ({"m1.csv": M1, "m2.csv": M2}, ["-m", "m1.csv", "m2.csv", "-t", "6", "-o", "out.txt"]),
    "genus_order_of_options": ({"m1.csv": M1, "m2.csv": M2}, ["-t", "5", "-o", "out.txt", "-m", "m1.csv", "m2.csv"]),
    "file_after_t": ({"m1.csv": M1, "m2.csv": M2}, ["-m", "m1.csv", "-t", "2", "m2.csv", "-o", "out.txt"]),
    "same_file_twice": ({"m1.csv": M1}, ["-m", "m1.csv", "m1.csv", "-t", "0", "-o", "out.txt"]),
    "level_7": ({"m1.csv": M1}, ["-m", "m1.csv", "m1.csv", "-t", "7", "-o", "out.txt"]),
    "level_two_digits": ({"m1.csv": M1}, ["-m", "m1.csv", "m1.csv", "-t", "03", "-o", "out.txt"]),
    "too_few_args": ({"m1.csv": M1}, ["-m", "m1.csv", "-t", "0", "-o"]),
    "missing_file": ({"m1.csv": M1}, ["-m", "m1.csv", "nope.csv", "-t", "0", "-o", "out.txt"]),
    "no_match_at_all": ({"m1.csv": "OTU,times_hit\nUnidentified(GI:1);,4\n"}, ["-m", "m1.csv", "m1.csv", "-t", "0", "-o", "out.txt"]),
}


def run(cmd, cwd):
    env = dict(os.environ, PERL_HASH_SEED="0", PERL_PERTURB_KEYS="0")
    p = subprocess.run(["timeout", "20"] + cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout, p.stderr


def main():
    gm = os.path.join(GOLD, "megaclust")
    gt = os.path.join(GOLD, "megaclustable")
    for d in (gm, gt):
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
    manifest = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (text, extra) in MEGACLUST_CASES.items():
            open(os.path.join(tmp, "in.txt"), "w", newline="").write(text)
            if os.path.exists(os.path.join(tmp, "out.txt")):
                os.remove(os.path.join(tmp, "out.txt"))
            rc, so, se = run(["perl", MC2, "-i", "in.txt", "-o", "out.txt"] + extra, tmp)
            open(os.path.join(gm, name + ".in.txt"), "w", newline="").write(text)
            open(os.path.join(gm, name + ".stdout.txt"), "wb").write(so)
            outp = os.path.join(tmp, "out.txt")
            has_out = os.path.exists(outp)
            if has_out:
                shutil.copy(outp, os.path.join(gm, name + ".out.csv"))
            manifest[name] = {"argv": ["-i", "in.txt", "-o", "out.txt"] + extra, "rc": rc, "has_out": has_out}
        for name, argv in MEGACLUST_ARGV_ONLY.items():
            for f in ("out.txt",):
                if os.path.exists(os.path.join(tmp, f)):
                    os.remove(os.path.join(tmp, f))
            rc, so, se = run(["perl", MC2] + argv, tmp)
            open(os.path.join(gm, name + ".stdout.txt"), "wb").write(so)
            manifest[name] = {"argv": argv, "rc": rc, "has_out": os.path.exists(os.path.join(tmp, "out.txt")),
                              "stderr_nonempty": bool(se.strip())}
    json.dump(manifest, open(os.path.join(gm, "manifest.json"), "w"), indent=1, sort_keys=True)
    manifest = {}
    for name, (files, argv) in MEGACLUSTABLE_CASES.items():
        with tempfile.TemporaryDirectory() as tmp:
            for fn, text in files.items():
                open(os.path.join(tmp, fn), "w", newline="").write(text)
                open(os.path.join(gt, name + "." + fn), "w", newline="").write(text)
            rc, so, se = run(["perl", MCT] + argv, tmp)
            open(os.path.join(gt, name + ".stdout.txt"), "wb").write(so)
            outp = os.path.join(tmp, "out.txt")
            has_out = os.path.exists(outp)
            if has_out:
                shutil.copy(outp, os.path.join(gt, name + ".out.txt"))
            manifest[name] = {"argv": argv, "files": sorted(files), "rc": rc, "has_out": has_out}
    json.dump(manifest, open(os.path.join(gt, "manifest.json"), "w"), indent=1, sort_keys=True)
    print("wrote", gm, gt)


if __name__ == "__main__":
    main()
