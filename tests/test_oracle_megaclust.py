"""Oracle (oracle/o_megaclust.c) against the goldens the reference's own Perl produced
(oracle/gen_goldens_megaclust.py): Megaclust/megaclust2.pl and Megaclustable/megaclustable.pl."""
import json
import os
import shutil

import pytest

from conftest import GOLD as GOLDEN, run_cmd

MC = os.path.join(GOLDEN, "megaclust")
MT = os.path.join(GOLDEN, "megaclustable")


def table_key(data):
    """megaclust2 prints `keys %h` (Perl hash order): header first, data lines as a multiset."""
    lines = data.split(b"\n")
    return lines[0], sorted(lines[1:])


def megaclust_cases():
    return sorted(json.load(open(os.path.join(MC, "manifest.json"))).items())


def megaclustable_cases():
    return sorted(json.load(open(os.path.join(MT, "manifest.json"))).items())


def run_megaclust_case(cmd_prefix, name, info, tmp_path):
    src = os.path.join(MC, name + ".in.txt")
    if os.path.exists(src):
        shutil.copy(src, tmp_path / "in.txt")
    rc, out, err = run_cmd(cmd_prefix + info["argv"], cwd=tmp_path)
    assert out == open(os.path.join(MC, name + ".stdout.txt"), "rb").read()
    assert (rc == 0) == (info["rc"] == 0)
    assert os.path.exists(tmp_path / "out.txt") == info["has_out"]
    if info["has_out"]:
        assert table_key((tmp_path / "out.txt").read_bytes()) == table_key(open(os.path.join(MC, name + ".out.csv"), "rb").read())


def run_megaclustable_case(cmd_prefix, name, info, tmp_path):
    for fn in info["files"]:
        shutil.copy(os.path.join(MT, name + "." + fn), tmp_path / fn)
    rc, out, err = run_cmd(cmd_prefix + info["argv"], cwd=tmp_path)
    assert out == open(os.path.join(MT, name + ".stdout.txt"), "rb").read()
    assert os.path.exists(tmp_path / "out.txt") == info["has_out"]
    if info["has_out"]:
        assert (tmp_path / "out.txt").read_bytes() == open(os.path.join(MT, name + ".out.txt"), "rb").read()


@pytest.mark.parametrize("name,info", megaclust_cases())
def test_oracle_megaclust2_matches_reference(name, info, oracle_bin, tmp_path):
    run_megaclust_case([oracle_bin, "megaclust2"], name, info, tmp_path)


@pytest.mark.parametrize("name,info", megaclustable_cases())
def test_oracle_megaclustable_matches_reference(name, info, oracle_bin, tmp_path):
    run_megaclustable_case([oracle_bin, "megaclustable"], name, info, tmp_path)
