"""Oracle (oracle/o_trim.c) against the goldens the reference's own Perl produced (oracle/gen_goldens_trim.py:
Trim/trim2.4.pl == trim2.3.pl on FASTQ and QSEQ input; oracle/gen_goldens_trim_fasta.py: trim2.4.pl on FASTA input,
parse_fasta with a quality file and join_fasta)."""
import json
import os
import shutil

import pytest

from conftest import GOLD, run_cmd

TR = os.path.join(GOLD, "trim")
TRF = os.path.join(GOLD, "trim_fasta")


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


def trim_fasta_cases():
    return sorted(json.load(open(os.path.join(TRF, "manifest.json"))).items())


def run_trim_fasta_case(cmd_prefix, name, info, tmp_path):
    """One FASTA-input case: stdout and the (often empty) runblast file."""
    for key in ("a", "b", "q"):
        if info["has_" + key]:
            shutil.copy(os.path.join(TRF, "%s.%s.txt" % (name, key)), tmp_path / (key + ".txt"))
    rc, out, err = run_cmd(cmd_prefix + info["argv"], cwd=tmp_path)
    assert out == open(os.path.join(TRF, name + ".stdout.txt"), "rb").read()
    assert rc == info["rc"]
    fasta = tmp_path / "output_files" / "trim2" / "a.txt_runblast.fasta"
    assert fasta.exists() == info["has_fasta"]
    if info["has_fasta"]:
        assert fasta.read_bytes() == open(os.path.join(TRF, name + ".runblast.fasta"), "rb").read()


@pytest.mark.parametrize("name,info", trim_fasta_cases())
def test_oracle_trim2_fasta_modes_match_reference(name, info, oracle_bin, tmp_path):
    run_trim_fasta_case([oracle_bin, "trim2"], name, info, tmp_path)
