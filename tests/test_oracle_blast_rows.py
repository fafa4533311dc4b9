"""Known answers for the score columns of BLAST mode: the 10 992 `-outfmt 6` rows that real BLAST+ 2.2.26 printed and the
reference keeps in validation_dataset/Data-set_2_consensus.xlsx (extracted by oracle/gen_goldens_xlsx.py).  The
sequences behind them are not in the reference, so the alignments themselves stay unpinned; what the rows do pin is
the arithmetic and text of columns 3, 11 and 12 (spec pgx-blastn v1, S4/S5): lambda 1.28, K 0.46, reward 1,
penalty -2, bit scores above 99.9 truncated to an integer, one-digit e-value mantissas."""
import ctypes as C
import math
import os

import pytest

from conftest import GOLD, ORACLE_DIR

ROWS = os.path.join(GOLD, "blast_rows", "xlsx_blast_rows.tsv")


def xlsx_rows():
    out = []
    for line in open(ROWS):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        out.append(dict(pident=float(f[2]), length=int(float(f[3])), mismatch=int(float(f[4])), gapopen=int(float(f[5])),
                        evalue=float(f[10]), bits=float(f[11])))
    return out


def raw_score(row):
    """Raw score of a BLAST+ row under reward 1 / penalty -2 / linear gap cost 2.5 per gap column (megablast defaults):
    identities from pident * length, gap columns = length - identities - mismatches."""
    ident = round(row["pident"] * row["length"] / 100.0)
    gaps = row["length"] - ident - row["mismatch"]
    return math.floor(ident - 2 * row["mismatch"] - 2.5 * gaps)


def check_rows(bits_text, pident_text):
    rows = xlsx_rows()
    assert len(rows) == 10992
    image = {float(bits_text(s)) for s in range(1, 2200)}
    exact = 0
    for r in rows:
        # every bit score BLAST+ printed is a value our score -> text map produces (the map skips ~46 % of the integers)
        assert r["bits"] in image, r
        s = raw_score(r)
        k = next(k for k in range(0, 64) if float(bits_text(s + k)) == r["bits"])  # StopIteration = a row below its score
        exact += k == 0
        if r["gapopen"] == 0:
            assert float(pident_text(r["length"], r["mismatch"])) == round(r["pident"], 2), r
            if r["length"] < 600:  # short gap-free rows hold no ambiguity letters: the reconstruction is exact
                assert k == 0, r
        # one significant digit: "%2.0le" / "%3.0le" of the tabular writer
        if r["evalue"] != 0.0:
            mant = r["evalue"] / 10.0 ** math.floor(math.log10(r["evalue"]) + 1e-9)
            assert abs(mant - round(mant)) < 1e-6, r
    # rows whose score is reproduced exactly from their own columns; the others sit k > 0 above it (IUPAC letters in the
    # RDP sequences score better than the mismatch they are counted as), never below
    assert exact >= 8700


class _Stats(C.Structure):
    _fields_ = [("lam", C.c_double), ("K", C.c_double), ("H", C.c_double), ("db_len", C.c_int64), ("db_nseq", C.c_int64),
                ("alpha", C.c_double), ("beta", C.c_double)]


def test_oracle_score_columns_match_the_reference_blast_rows(oracle_bin):
    lib = C.CDLL(os.path.join(ORACLE_DIR, "liboracle.so"))
    lib.o_blast_bitscore.restype = C.c_double
    lib.o_blast_bitscore.argtypes = [C.POINTER(_Stats), C.c_int32]
    lib.o_blast_format_bitscore.argtypes = [C.c_double, C.c_char_p]
    st = _Stats(1.28, 0.46, 0.85, 10 ** 9, 10 ** 6, 1.5, -2.0)

    def bits_text(s):
        buf = C.create_string_buffer(32)
        lib.o_blast_format_bitscore(lib.o_blast_bitscore(C.byref(st), s), buf)
        return buf.value.decode()

    check_rows(bits_text, lambda length, mm: "%.2f" % (100.0 * (length - mm) / length))


def test_oracle_evalue_text_has_the_blast_tabular_shapes(oracle_bin):
    lib = C.CDLL(os.path.join(ORACLE_DIR, "liboracle.so"))
    lib.o_blast_format_evalue.argtypes = [C.c_double, C.c_char_p]

    def text(e):
        buf = C.create_string_buffer(32)
        lib.o_blast_format_evalue(e, buf)
        return buf.value.decode()

    # the spreadsheet's non-zero e-values, as BLAST+ wrote them before Excel parsed them ("2e-15" for row 3)
    for r in xlsx_rows():
        if r["evalue"] != 0.0:
            assert float(text(r["evalue"])) == pytest.approx(r["evalue"], rel=1e-12), r
        else:
            assert text(1e-181) == "0.0"
    assert text(2.0000000000000002e-15) == "2e-15"
