#!/usr/bin/env python3
"""Generate tests/golden/trim_fasta by running the REFERENCE's own Perl (/root/reference/Trim/trim2.4.pl) in this
container on FASTA-format input: `-a seq -q qual` (parse_fasta, trim2.4.pl:384-465) and `-a seq1 -b seq2 -j [-g N]`
(join_fasta, :301-382).  trim2.3.pl's join_fasta is a different text (it prints to STDOUT and tests eof in the loop
conditions); the goldens are 2.4's, the file VERDICT r1 cites.  oracle/ is test infrastructure; only data (seeded inputs +
the bytes the reference printed / wrote) goes into the repo.

Usage: python3 oracle/gen_goldens_trim_fasta.py
"""
import json
import os
import random
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("PGX_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden", "trim_fasta")
TRIM24 = os.path.join(REF, "Trim", "trim2.4.pl")


def bases(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def record(rng, name, n, width=60, qual=None, qsep=" "):
    """A FASTA record and its quality record, `width` letters / numbers per line."""
    seq = bases(rng, n)
    q = qual if qual is not None else [rng.randint(2, 40) for _ in range(n)]
    s = [">" + name] + [seq[i:i + width] for i in range(0, n, width)]
    t = [">" + name] + [qsep.join(str(v) for v in q[i:i + width]) for i in range(0, n, width)]
    return "\n".join(s) + "\n", "\n".join(t) + "\n"


def build_cases():
    rng = random.Random(20240611)
    cases = {}

    def add(name, argv, a=None, b=None, q=None):
        cases[name] = {"argv": argv, "a": a, "b": b, "q": q}

    def pf(name, recs, argv=("-a", "a.txt", "-q", "q.txt"), tail_a=b"", tail_q=b""):
        a = "".join(r[0] for r in recs).encode("latin-1") + tail_a
        q = "".join(r[1] for r in recs).encode("latin-1") + tail_q
        add(name, list(argv), a, None, q)

    basic = [record(rng, "r%d some text" % i, rng.randint(30, 200)) for i in range(8)]
    pf("pf_basic", basic)
    pf("pf_no_blank_in_header", [record(rng, "h%d" % i, rng.randint(50, 130)) for i in range(4)])
    pf("pf_header_blank_first", [record(rng, " lead%d x" % i, 70) for i in range(3)])
    zero = [record(rng, "a one", 100), record(rng, "b two", 90, qual=[0] * 90), record(rng, "c three", 150, qual=[0] * 150), record(rng, "d four", 80),
            record(rng, "e five", 10)]
    pf("pf_zero_quality_records_keep_the_previous_range", zero)
    neg = [record(rng, "n%d x" % i, 120, qual=[rng.choice((-30, -5, 0, 3, 20, 40)) for _ in range(120)]) for i in range(5)]
    pf("pf_negative_qualities_move_the_start", neg + [record(rng, "last x", 60)])
    pf("pf_lines_of_40", [record(rng, "w%d x" % i, rng.randint(50, 170), width=40) for i in range(5)])
    pf("pf_lines_of_80", [record(rng, "v%d x" % i, rng.randint(90, 250), width=80) for i in range(4)])
    odd_q = [record(rng, "q%d x" % i, 70, qsep="  " if i % 2 else " ") for i in range(4)]
    pf("pf_double_blanks_in_quality", odd_q)
    a1, q1 = record(rng, "f1 x", 65)
    q1 = q1.replace(" ", " 3.5 ", 1).replace("\n", " \n")
    a2, q2 = record(rng, "f2 x", 65)
    q2b = ">f2 x\n abc 40 1e1 4x " + " ".join(["30"] * 56) + "\n\t12 7 7 7 7\n"
    pf("pf_odd_quality_fields", [(a1, q1), (a2, q2b), record(rng, "f3 x", 61)])
    pf("pf_last_line_without_newline", [record(rng, "e%d x" % i, 75) for i in range(3)], tail_a=b">end x\nACGTACGT", tail_q=b">end x\n40 40 40 40 40 40 40 40")
    short_q = [record(rng, "s%d x" % i, 90) for i in range(4)]
    a = "".join(r[0] for r in short_q).encode()
    q = "".join(r[1] for r in short_q[:2]).encode()
    add("pf_quality_file_ends_early", ["-a", "a.txt", "-q", "q.txt"], a, None, q)
    add("pf_quality_file_longer", ["-a", "a.txt", "-q", "q.txt"], "".join(r[0] for r in short_q[:2]).encode(), None, "".join(r[1] for r in short_q).encode())
    pf("pf_blank_lines_and_gt_inside", [record(rng, "g1 x", 80), ("\n>g2 > y z\nACGT>ACGT\nAC\n\n", "\n>g2 > y z\n40 40 40 40 40 40 40 40 40\n40 40\n\n"), record(rng, "g3 x", 70)])
    pf("pf_crlf", [(r[0].replace("\n", "\r\n"), r[1].replace("\n", "\r\n")) for r in basic[:4]])
    pf("pf_single_record", [record(rng, "only x", 100)])
    pf("pf_many", [record(rng, "m%d t" % i, rng.randint(20, 400), width=rng.choice((60, 60, 60, 70))) for i in range(120)])
    add("pf_no_q", ["-a", "a.txt"], basic[0][0].encode())
    add("pf_q_zero", ["-a", "a.txt", "-q", "0"], basic[0][0].encode())
    add("pf_q_missing_file", ["-a", "a.txt", "-q", "nothere.txt"], basic[0][0].encode())
    pf("pf_with_b_and_g_ignored", basic[:3], argv=("-a", "a.txt", "-q", "q.txt", "-g", "5", "-t", "3"))

    def fa(recs):
        return "".join(recs).encode("latin-1")

    def rec(name, n, width=60, nl=True):
        s = bases(rng, n)
        return ">" + name + "\n" + "\n".join(s[i:i + width] for i in range(0, n, width)) + ("\n" if nl else "")

    one = [rec("p%d/1" % i, rng.randint(40, 60)) for i in range(6)]
    two = [rec("p%d/2" % i, rng.randint(40, 60)) for i in range(6)]
    add("jf_single_line_records", ["-a", "a.txt", "-b", "b.txt", "-j"], fa(one), fa(two))
    add("jf_gap_5", ["-a", "a.txt", "-b", "b.txt", "-j", "-g", "5"], fa(one), fa(two))
    add("jf_gap_0", ["-a", "a.txt", "-b", "b.txt", "-j", "-g", "0"], fa(one), fa(two))
    add("jf_gap_text", ["-a", "a.txt", "-b", "b.txt", "-g", "2.5", "-j"], fa(one), fa(two))
    m1 = [rec("m%d/1 extra" % i, rng.randint(100, 250)) for i in range(5)]
    m2 = [rec("m%d/2 >extra" % i, rng.randint(100, 250)) for i in range(5)]
    add("jf_multi_line_records", ["-a", "a.txt", "-b", "b.txt", "-j", "-g", "3"], fa(m1), fa(m2))
    add("jf_b_shorter", ["-a", "a.txt", "-b", "b.txt", "-j", "-g", "3"], fa(m1), fa(m2[:3]))
    add("jf_a_shorter", ["-a", "a.txt", "-b", "b.txt", "-j", "-g", "3"], fa(m1[:3]), fa(m2))
    add("jf_no_trailing_newline", ["-a", "a.txt", "-b", "b.txt", "-j"], fa(one)[:-1], fa(two)[:-1])
    add("jf_multi_line_no_trailing_newline", ["-a", "a.txt", "-b", "b.txt", "-j", "-g", "2"], fa(m1)[:-1], fa(m2)[:-1])
    add("jf_one_record_each", ["-a", "a.txt", "-b", "b.txt", "-j", "-g", "4"], fa(one[:1]), fa(two[:1]))
    add("jf_header_only_at_end", ["-a", "a.txt", "-b", "b.txt", "-j"], fa(one[:3]) + b">tail\n", fa(two[:3]) + b">tail2\n")
    add("jf_empty_b", ["-a", "a.txt", "-b", "b.txt", "-j"], fa(one[:3]), b"")
    add("jf_crlf", ["-a", "a.txt", "-b", "b.txt", "-j", "-g", "2"], fa(m1[:3]).replace(b"\n", b"\r\n"), fa(m2[:3]).replace(b"\n", b"\r\n"))
    add("jf_without_b", ["-a", "a.txt", "-j"], fa(one))
    add("jf_many", ["-a", "a.txt", "-b", "b.txt", "-j", "-g", "100"], fa([rec("k%d/1" % i, rng.randint(30, 300)) for i in range(150)]),
        fa([rec("k%d/2" % i, rng.randint(30, 300)) for i in range(150)]))
    # quality lines of more than 60 values: positions are a + 60 x line number, so a later line can move $end BEFORE $first;
    # $length = $end + 1 - $start is then negative, below the bareword cut-off (0), and the record is not printed at all
    # (trim2.4.pl:404-408) -- nor is a following record whose qualities never move the range
    def wide(name, n, q):
        seq = bases(rng, n)
        return (">" + name + "\n" + "\n".join(seq[i:i + 100] for i in range(0, n, 100)) + "\n",
                ">" + name + "\n" + "\n".join(" ".join(str(v) for v in q[i:i + 100]) for i in range(0, n, 100)) + "\n")
    q_rej = [30] * 89 + [-99999, 0] + [0] * 9 + [0] * 5 + [999999] + [1] * 20
    pf("pf_wide_quality_lines_reject_a_record", [record(rng, "before x", 80), wide("wide1 x", len(q_rej), q_rej), wide("still x", 120, [0] * 120),
                                                   record(rng, "after x", 90), wide("wide2 x", len(q_rej), q_rej), record(rng, "end x", 70)])
    return cases


def run_script(info, work):
    for key in ("a", "b", "q"):
        if info[key] is not None:
            with open(os.path.join(work, key + ".txt"), "wb") as f:
                f.write(info[key])
    p = subprocess.run(["timeout", "60", "perl", TRIM24] + info["argv"], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    fasta_path = os.path.join(work, "output_files", "trim2", "a.txt_runblast.fasta")
    fasta = open(fasta_path, "rb").read() if os.path.exists(fasta_path) else None
    return p.returncode, p.stdout, fasta


def main():
    if os.path.isdir(GOLD):
        shutil.rmtree(GOLD)
    os.makedirs(GOLD)
    manifest = {}
    for name, info in sorted(build_cases().items()):
        work = tempfile.mkdtemp(prefix="pgx_trimfa_")
        try:
            rc, out, fasta = run_script(info, work)
        finally:
            shutil.rmtree(work, ignore_errors=True)
        for key in ("a", "b", "q"):
            if info[key] is not None:
                with open(os.path.join(GOLD, "%s.%s.txt" % (name, key)), "wb") as f:
                    f.write(info[key])
        with open(os.path.join(GOLD, name + ".stdout.txt"), "wb") as f:
            f.write(out)
        if fasta is not None:
            with open(os.path.join(GOLD, name + ".runblast.fasta"), "wb") as f:
                f.write(fasta)
        manifest[name] = {"argv": info["argv"], "rc": rc, "has_a": info["a"] is not None, "has_b": info["b"] is not None,
                          "has_q": info["q"] is not None, "has_fasta": fasta is not None}
        print("%-50s rc=%d stdout=%6d fasta=%s" % (name, rc, len(out), "-" if fasta is None else len(fasta)))
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
