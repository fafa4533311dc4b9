#!/usr/bin/env python3
"""Generate tests/golden/trim by running the REFERENCE's own Perl (/root/reference/Trim/trim2.4.pl, perl 5.34) in
this container (FASTQ / QSEQ ingest + quality trimming, the step before the hot path).
oracle/ is test infrastructure; only data (seeded inputs + the bytes the reference printed / wrote) goes into
the repo.  Each case runs in a scratch directory under `timeout`; the script writes
output_files/trim2/<basename of -a>_runblast.fasta relative to its working directory.

trim2.3.pl (the version README.md:34 calls) differs from trim2.4.pl only inside join_fasta (the FASTA `-j` mode,
not covered); every case below is also run through trim2.3.pl and must give the same bytes.

Usage: python3 oracle/gen_goldens_trim.py
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
GOLD = os.path.join(ROOT, "tests", "golden", "trim")
TRIM24 = os.path.join(REF, "Trim", "trim2.4.pl")
TRIM23 = os.path.join(REF, "Trim", "trim2.3.pl")


def bases(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def fq(name, seq, qual, plus="+"):
    return ("@%s\n%s\n%s\n%s\n" % (name, seq, plus, qual)).encode("latin-1")


def phred33(vals):
    return "".join(chr(33 + v) for v in vals)


def phred64(vals):
    return "".join(chr(64 + v) for v in vals)


def fastq_profiles(seed, n, lmin=40, lmax=160):
    """Records whose quality strings exercise the running-sum rule: clean, tail drop, dips, noise, all low."""
    rng = random.Random(seed)
    out = []
    for i in range(n):
        L = rng.randint(lmin, lmax)
        kind = rng.randrange(7)
        if kind == 0:
            q = [rng.randint(30, 41) for _ in range(L)]
        elif kind == 1:
            cut = rng.randint(L // 2, L)
            q = [rng.randint(30, 41) if j < cut else rng.randint(2, 12) for j in range(L)]
        elif kind == 2:
            q = [rng.randint(2, 41) for _ in range(L)]
        elif kind == 3:
            q = [2 if (j % rng.randint(7, 40)) == 0 else 38 for j in range(L)]
        elif kind == 4:
            q = [rng.randint(2, 15) for _ in range(L)]
        elif kind == 5:
            a, b = sorted((rng.randint(0, L), rng.randint(0, L)))
            q = [5 if a <= j < b else 40 for j in range(L)]
        else:
            q = [20 + rng.choice((-1, 0, 1, 2)) for _ in range(L)]
        out.append(fq("r%d/%d" % (seed, i), bases(rng, L, "ACGTN" if i % 9 == 0 else "ACGT"), phred33(q)))
    return b"".join(out)


def qseq_line(rng, mate, L, qvals, filt="1", seq=None, machine="HWI-M1", extra=None):
    s = seq if seq is not None else bases(rng, L)
    f = [machine, "7", str(rng.randint(1, 8)), str(rng.randint(1101, 2316)), str(rng.randint(1000, 20000)),
         str(rng.randint(1000, 20000)), "ACGTAC", str(mate), s, phred64(qvals), filt]
    if extra:
        f += extra
    return ("\t".join(f) + "\n").encode("latin-1")


def qseq_pairs(seed, n, lmin=95, lmax=151):
    rng = random.Random(seed)
    a, b = [], []
    for i in range(n):
        for mate, dst in ((1, a), (2, b)):
            L = rng.randint(lmin, lmax)
            kind = rng.randrange(6)
            if kind == 0:
                q = [rng.randint(30, 40) for _ in range(L)]
            elif kind == 1:
                cut = rng.randint(L // 2, L)
                q = [rng.randint(30, 40) if j < cut else 2 for j in range(L)]
            elif kind == 2:
                q = [rng.randint(2, 40) for _ in range(L)]
            elif kind == 3:
                q = [2 for _ in range(L)]
            elif kind == 4:
                q = [3 if (j % rng.randint(9, 50)) == 0 else 39 for j in range(L)]
            else:
                q = [20 + rng.choice((-2, 0, 1, 3)) for _ in range(L)]
            s = bases(rng, L)
            if rng.random() < 0.3:
                p = rng.randrange(L)
                s = s[:p] + "." * min(3, L - p) + s[p + 3:]
            dst.append(qseq_line(rng, mate, L, q, filt=rng.choice("01"), seq=s))
    return b"".join(a), b"".join(b)


def good33(L):
    return phred33([38] * L)


def build_cases():
    rng = random.Random(7)
    cases = {}

    def add(name, argv, a=None, b=None):
        cases[name] = {"argv": argv, "a": a, "b": b}

    s100, s120, s90, s60 = bases(rng, 100), bases(rng, 120), bases(rng, 90), bases(rng, 60)
    basic = (fq("r0 x y", s100, good33(100)) + fq("r1", s100, phred33([40] * 80 + [2] * 20)) + fq("r2", s100, phred33([10] * 100)) +
             fq("r3", s60, good33(60)) + fq("r4@lane@1", s120, phred33([random.Random(3).randint(2, 40) for _ in range(120)])) +
             fq("r5\tz", s90, phred33([0 if j % 30 == 0 else 40 for j in range(90)])))
    add("fq_basic", ["-a", "a.txt"], basic)
    add("fq_paired_g5", ["-a", "a.txt", "-b", "b.txt", "-g", "5"], basic, b"")
    add("fq_paired_default_gap", ["-a", "a.txt", "-b", "b.txt"], basic, b"ignored\n")
    add("fq_paired_g0_is_default", ["-a", "a.txt", "-b", "b.txt", "-g", "0"], basic, b"")
    add("fq_paired_g_text", ["-a", "a.txt", "-b", "b.txt", "-g", "abc"], basic, b"")
    add("fq_paired_g_fraction", ["-a", "a.txt", "-b", "b.txt", "-g", "2.5"], basic, b"")
    add("fq_paired_odd_records", ["-a", "a.txt", "-b", "b.txt", "-g", "3"], basic + fq("last", s100, good33(100)), b"")
    add("fq_g_without_b", ["-a", "a.txt", "-g", "4"], basic)
    add("fq_t_has_no_effect", ["-a", "a.txt", "-t", "5"], basic)
    add("fq_lc_qc_words", ["-a", "a.txt", "-lc", "80", "-qc", "30"], basic)
    add("fq_qc_first_stops_getopts", ["-qc", "30", "-a", "a.txt"], basic)
    add("fq_crlf", ["-a", "a.txt"], basic.replace(b"\n", b"\r\n"))
    add("fq_crlf_paired", ["-a", "a.txt", "-b", "b.txt", "-g", "2"], basic.replace(b"\n", b"\r\n"), b"")
    add("fq_no_trailing_newline", ["-a", "a.txt"], basic[:-1])
    add("fq_truncated_record", ["-a", "a.txt"], basic + b"@cut\n" + s100.encode() + b"\n")
    add("fq_truncated_header_only", ["-a", "a.txt", "-b", "b.txt", "-g", "2"], basic + b"@cut\n", b"")
    add("fq_last_header_zero", ["-a", "a.txt"], basic + b"0")
    add("fq_last_header_zero_newline", ["-a", "a.txt"], basic + b"0\n")
    add("fq_quality_longer", ["-a", "a.txt", "-b", "b.txt", "-g", "1"],
        fq("q1", s90, good33(120)) + fq("q2", s90, good33(120)) + fq("q3", s60 + s60[:9], good33(80)) + fq("q4", s60 + s60[:9], good33(80)), b"")
    add("fq_quality_shorter", ["-a", "a.txt"], fq("q1", s120, good33(75)) + fq("q2", s120, good33(70)) + fq("q3", s120, good33(71)))
    add("fq_length_boundary", ["-a", "a.txt"], b"".join(fq("L%d" % n, bases(rng, n), good33(n)) for n in (69, 70, 71, 72)))
    add("fq_spaces_in_sequence", ["-a", "a.txt", "-b", "b.txt", "-g", "2"],
        fq("sp1", s60[:40] + " " + s60[:40] + "\t" + s60[:10], good33(92)) + fq("sp2", s60[:40] + " " + s60[:40] + "\t" + s60[:10], good33(92)), b"")
    add("fq_empty_lines", ["-a", "a.txt"], b"\n\n\n\n" + basic + b"\n")
    add("fq_all_low", ["-a", "a.txt"], b"".join(fq("low%d" % i, s100, phred33([5] * 100)) for i in range(5)))
    add("fq_high_bytes", ["-a", "a.txt"], fq("hb", s100, "".join(chr(200 + (j % 50)) for j in range(100))) + fq("nul", s100, "\x00" * 50 + "~" * 50))
    add("fq_random_1", ["-a", "a.txt"], fastq_profiles(11, 240))
    add("fq_random_2_paired", ["-a", "a.txt", "-b", "b.txt", "-g", "100"], fastq_profiles(12, 240, 60, 260), b"")
    add("fq_random_3_long", ["-a", "a.txt"], fastq_profiles(13, 60, 200, 700))

    qa, qb = qseq_pairs(21, 160)
    add("qs_random_g7", ["-a", "a.txt", "-b", "b.txt", "-g", "7"], qa, qb)
    add("qs_default_gap", ["-a", "a.txt", "-b", "b.txt"], qa[:len(qa) // 4], qb[:len(qb) // 4])
    add("qs_single_end", ["-a", "a.txt"], qa)
    qa2, qb2 = qseq_pairs(22, 120, 100, 180)
    for t in ("5", "1", "00", "abc", "2.7", "200", "0", "30"):
        add("qs_t_" + t.replace(".", "_"), ["-a", "a.txt", "-b", "b.txt", "-g", "3", "-t", t], qa2, qb2)
    qa3, qb3 = qseq_pairs(23, 40, 5, 40)
    first = qseq_line(rng, 1, 120, [40] * 120)
    add("qs_short_reads", ["-a", "a.txt", "-b", "b.txt", "-g", "3"], first + qa3, qseq_line(rng, 2, 120, [40] * 120) + qb3)
    lines_a, lines_b = qa.split(b"\n")[:-1], qb.split(b"\n")[:-1]
    add("qs_b_shorter", ["-a", "a.txt", "-b", "b.txt", "-g", "3"], qa, b"\n".join(lines_b[:50]) + b"\n")
    add("qs_a_shorter", ["-a", "a.txt", "-b", "b.txt", "-g", "3"], b"\n".join(lines_a[:50]) + b"\n", qb)
    add("qs_no_trailing_newline", ["-a", "a.txt", "-b", "b.txt", "-g", "3"], qa[:-1], qb[:-1])
    add("qs_last_line_zero", ["-a", "a.txt", "-b", "b.txt", "-g", "3"], qa + b"0", qb)
    # file A carries read number 2: nothing is trimmed, raw field 8 is printed (trim2.4.pl:187 is false)
    ra = b"".join(qseq_line(rng, 2, 30, [40] * 30) for _ in range(4)) + qseq_line(rng, 2, 1, [40], seq="0") + qseq_line(rng, 2, 12, [40] * 12, seq="AC..GT..AC..")
    rb = b"".join(qseq_line(rng, 1, 25, [40] * 25) for _ in range(3)) + qseq_line(rng, 1, 1, [40], seq="0") + qseq_line(rng, 1, 1, [40], seq="0") + b"x\ty\n"
    add("qs_read2_in_a", ["-a", "a.txt", "-b", "b.txt", "-g", "2"], ra, rb)
    odd = (first + b"onlyonefield\n" + b"a\tb\tc\n" + b"\n" + b"a\tb\tc\td\te\tf\tg\t1\n" + b"a\tb\tc\td\te\tf\tg\t1\t" + bases(rng, 130).encode() + b"\n" +
           b"a\t\t\t\t\t\t\t1\t" + bases(rng, 130).encode() + b"\t" + phred64([40] * 130).encode() + b"\n" +
           qseq_line(rng, 1, 130, [40] * 130, extra=["more", "fields"]) + b"\t\t\t\t\t\t\t1\t" + bases(rng, 130).encode() + b"\t" + phred64([40] * 130).encode() + b"\t\t\t\n")
    oddb = b"".join(qseq_line(rng, 2, 130, [40] * 130) for _ in range(9))
    add("qs_odd_lines", ["-a", "a.txt", "-b", "b.txt", "-g", "2"], odd, oddb)
    add("qs_crlf_not_recognised", ["-a", "a.txt", "-b", "b.txt"], qa.replace(b"\n", b"\r\n"), qb.replace(b"\n", b"\r\n"))
    add("qs_filter_flag_2_not_recognised", ["-a", "a.txt", "-b", "b.txt"], qseq_line(rng, 1, 120, [40] * 120, filt="2"), qb)
    add("qs_read_number_3_not_recognised", ["-a", "a.txt", "-b", "b.txt"], qseq_line(rng, 3, 120, [40] * 120), qb)
    add("qs_high_quality_bytes", ["-a", "a.txt", "-b", "b.txt", "-g", "1"],
        first + ("\t".join(["M", "1", "1", "1", "1", "1", "A", "1", bases(rng, 120), "".join(chr(150 + j % 100) for j in range(120)), "1"]) + "\n").encode("latin-1"),
        qseq_line(rng, 2, 120, [40] * 120) + qseq_line(rng, 2, 120, [40] * 120))

    add("usage_no_a", [])
    add("usage_a_zero", ["-a", "0"])
    add("unopenable_a", ["-a", "missing.txt"])
    add("unopenable_b", ["-a", "a.txt", "-b", "missing.txt"], basic)
    add("unknown_format", ["-a", "a.txt"], b"hello world\nsecond line\n")
    add("empty_file", ["-a", "a.txt"], b"")
    add("unknown_option", ["-x", "-a", "a.txt"], basic)
    return cases


def run_script(script, info, work):
    for key in ("a", "b"):
        if info[key] is not None:
            with open(os.path.join(work, key + ".txt"), "wb") as f:
                f.write(info[key])
    p = subprocess.run(["timeout", "60", "perl", script] + info["argv"], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    fasta_path = os.path.join(work, "output_files", "trim2", "a.txt_runblast.fasta")
    fasta = open(fasta_path, "rb").read() if os.path.exists(fasta_path) else None
    single = os.path.exists(os.path.join(work, "singletons", "a.txt_single.txt"))
    return p.returncode, p.stdout, fasta, single


def main():
    if os.path.isdir(GOLD):
        shutil.rmtree(GOLD)
    os.makedirs(GOLD)
    manifest = {}
    for name, info in sorted(build_cases().items()):
        results = []
        for script in (TRIM24, TRIM23):
            work = tempfile.mkdtemp(prefix="pgx_trim_")
            try:
                results.append(run_script(script, info, work))
            finally:
                shutil.rmtree(work, ignore_errors=True)
        assert results[0] == results[1], "trim2.3 and trim2.4 differ on " + name
        rc, out, fasta, single = results[0]
        for key in ("a", "b"):
            if info[key] is not None:
                with open(os.path.join(GOLD, "%s.%s.txt" % (name, key)), "wb") as f:
                    f.write(info[key])
        with open(os.path.join(GOLD, name + ".stdout.txt"), "wb") as f:
            f.write(out)
        if fasta is not None:
            with open(os.path.join(GOLD, name + ".runblast.fasta"), "wb") as f:
                f.write(fasta)
        manifest[name] = {"argv": info["argv"], "rc": rc, "has_a": info["a"] is not None, "has_b": info["b"] is not None,
                          "has_fasta": fasta is not None, "singletons_file": single}
        print("%-34s rc=%d stdout=%6d fasta=%s" % (name, rc, len(out), "-" if fasta is None else len(fasta)))
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
