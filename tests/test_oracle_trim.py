"""Oracle (oracle/o_trim.c) against the goldens the reference's own Perl produced (oracle/gen_goldens_trim.py):
Trim/trim2.4.pl == trim2.3.pl on FASTQ and QSEQ input."""
import json
import os
import shutil

import pytest

from conftest import GOLD, run_cmd

TR = os.path.join(GOLD, "trim")


def trim_cases():
    return sorted(json.load(open(os.path.join(TR, "manifest.json"))).items())


def run_trim_case(cmd_prefix, name, info, tmp_path):
    """Run one golden case in a scratch directory and compare stdout, the runblast FASTA and the side-effect files."""
    for key in ("a", "b"):
        if info["has_" + key]:
            shutil.copy(os.path.join(TR, "%s.%s.txt" % (name, key)), tmp_path / (key + ".txt"))
    rc, out, err = run_cmd(cmd_prefix + info["argv"], cwd=tmp_path)
    assert out == open(os.path.join(TR, name + ".stdout.txt"), "rb").read()
    assert rc == info["rc"]
    fasta = tmp_path / "output_files" / "trim2" / "a.txt_runblast.fasta"
    assert fasta.exists() == info["has_fasta"]
    if info["has_fasta"]:
        assert fasta.read_bytes() == open(os.path.join(TR, name + ".runblast.fasta"), "rb").read()
    assert (tmp_path / "singletons" / "a.txt_single.txt").exists() == info["singletons_file"]


@pytest.mark.parametrize("name,info", trim_cases())
def test_oracle_trim2_matches_reference(name, info, oracle_bin, tmp_path):
    run_trim_case([oracle_bin, "trim2"], name, info, tmp_path)


def test_oracle_trim2_declines_fasta_input(oracle_bin, tmp_path):
    (tmp_path / "a.txt").write_bytes(b">r1\nACGT\n")
    rc, out, err = run_cmd([oracle_bin, "trim2", "-a", "a.txt"], cwd=tmp_path)
    assert rc == 2 and b"not covered" in err
