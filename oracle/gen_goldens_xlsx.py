#!/usr/bin/env python3
"""Extract the BLAST+ rows of the reference's validation spreadsheet into a text fixture.

validation_dataset/Data-set_2_consensus.xlsx, sheet 1, rows 3-5498 hold, per read, the top BLAST+ hit (columns J-S) and
the hit the Consensus step picked (columns V-AE): the ten numeric `-outfmt 6` columns (pident, length, mismatch,
gapopen, qstart, qend, sstart, send, evalue, bitscore) of 10 992 rows that real BLAST+ 2.2.26 printed.  They are the
only output of the (un-vendored) BLAST dependency the reference holds, so they are the known answers the score
columns of pgx-blastn v1 are checked against (tests/test_oracle_blast_rows.py).  Only data is written: one line per
row, the cell values exactly as the workbook stores them.

Usage: python3 oracle/gen_goldens_xlsx.py
"""
import os
import xml.etree.ElementTree as ET
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("PGX_REFERENCE", "/root/reference")
XLSX = os.path.join(REF, "validation_dataset", "Data-set_2_consensus.xlsx")
OUT = os.path.join(ROOT, "tests", "golden", "blast_rows", "xlsx_blast_rows.tsv")
NS = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main"}
GROUPS = (("blast_top_hit", ("J", "K", "L", "M", "N", "O", "P", "Q", "R", "S")),
          ("consensus_pick", ("V", "W", "X", "Y", "Z", "AA", "AB", "AC", "AD", "AE")))


def main():
    z = zipfile.ZipFile(XLSX)
    shared = ["".join(t.text or "" for t in si.iter("{%s}t" % NS["m"]))
              for si in ET.fromstring(z.read("xl/sharedStrings.xml")).findall("m:si", NS)]
    rows = ET.fromstring(z.read("xl/worksheets/sheet1.xml")).find("m:sheetData", NS).findall("m:row", NS)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    n = 0
    with open(OUT, "w") as f:
        f.write("#sheet_row\tgroup\tpident\tlength\tmismatch\tgapopen\tqstart\tqend\tsstart\tsend\tevalue\tbitscore\n")
        for r in rows[2:]:
            cells = {}
            for c in r.findall("m:c", NS):
                v = c.find("m:v", NS)
                if v is not None:
                    col = "".join(ch for ch in c.get("r") if ch.isalpha())
                    cells[col] = shared[int(v.text)] if c.get("t") == "s" else v.text
            for name, cols in GROUPS:
                if all(k in cells for k in cols):
                    f.write("\t".join([r.get("r"), name] + [cells[k] for k in cols]) + "\n")
                    n += 1
    print("wrote %d rows to %s" % (n, OUT))


if __name__ == "__main__":
    main()
