"""The product's own score-column formatter (pgx_blast_score_columns, the function that fills the table the device row
formatter indexes) against the BLAST+ rows of the reference's validation spreadsheet: see test_oracle_blast_rows.py."""
import pytest

from test_oracle_blast_rows import check_rows

pytestmark = pytest.mark.gpu


def test_product_score_columns_match_the_reference_blast_rows():
    import pangea_plus_amd as pg
    check_rows(lambda s: pg._capi.blast_score_columns(s, 1400, 10 ** 9, 10 ** 6)[1],
               lambda length, mm: "%.2f" % (100.0 * (length - mm) / length))


def test_product_and_oracle_score_columns_are_the_same_text(oracle_bin):
    import ctypes as C
    import os

    import pangea_plus_amd as pg
    from conftest import ORACLE_DIR
    from test_oracle_blast_rows import _Stats
    lib = C.CDLL(os.path.join(ORACLE_DIR, "liboracle.so"))
    lib.o_blast_bitscore.restype = C.c_double
    lib.o_blast_bitscore.argtypes = [C.POINTER(_Stats), C.c_int32]
    lib.o_blast_evalue.restype = C.c_double
    lib.o_blast_evalue.argtypes = [C.POINTER(_Stats), C.c_int64, C.c_int32]
    lib.o_blast_format_bitscore.argtypes = [C.c_double, C.c_char_p]
    lib.o_blast_format_evalue.argtypes = [C.c_double, C.c_char_p]
    for db_len, db_nseq in ((10 ** 9, 666667), (531842, 373), (3 * 10 ** 9, 2 * 10 ** 6)):
        st = _Stats()
        lib.o_blast_stats_init(C.byref(st), C.c_int64(db_len), C.c_int64(db_nseq), 1)   # spec v2: the gapped search's statistics
        for qlen in (28, 56, 150, 513, 1400, 70000):
            for score in list(range(28, 200)) + [777, 1402, 5000, 5415, 69999]:
                if score > qlen:
                    continue
                ev, bs = C.create_string_buffer(32), C.create_string_buffer(32)
                lib.o_blast_format_evalue(lib.o_blast_evalue(C.byref(st), qlen, score), ev)
                lib.o_blast_format_bitscore(lib.o_blast_bitscore(C.byref(st), score), bs)
                assert pg._capi.blast_score_columns(score, qlen, db_len, db_nseq) == (ev.value.decode(), bs.value.decode())
