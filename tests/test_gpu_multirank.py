"""GPU, N > 1: the real multi-rank path on the one GPU of the test box.  RCCL does not accept two ranks on one
device, so the two ranks talk over gloo; everything else is what an 8-GPU run does: `Db.alloc_like` /
`device_arrays` / `finish_import` on the receiving rank, read blocks per rank, outputs concatenated in rank order."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _env(**kw):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(kw)
    return env


@pytest.mark.parametrize("whole_index", ["0", "1"])
def test_two_ranks_import_the_database_and_agree_with_one_rank(tmp_path, whole_index):
    out = tmp_path / "table.tsv"
    port = 29700 + os.getpid() % 200 + int(whole_index)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "workers", "two_rank_search.py"), str(out), "3001"]
    p = subprocess.run(cmd, env=_env(PGX_BCAST_INDEX=whole_index), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    status = open(str(out) + ".status").read()
    assert status.startswith("ok"), status
    # every read of both blocks is in the table, in file order
    names = [l.split(b"\t", 1)[0] for l in open(out, "rb").read().splitlines()]
    firsts = [int(n[1:]) for n in names]
    assert firsts == sorted(firsts) and firsts[-1] > 2900 and firsts[0] < 5


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no outer launcher: the parent starts the ranks as a child torch.distributed.run
    and relays the one JSON line, which must say n_gpus 2 (both ranks on device 0 over gloo here)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "200000",
           "--n-seq", "20000", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=_env(PGX_BENCH_ONE_DEVICE="1", PGX_BENCH_BACKEND="gloo"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["value"] > 0
    assert j["setup_s"]["index_broadcast_bytes"] > 0
    assert j["setup_s"]["broadcast_mode"].startswith("packed bases")


def test_dry_ranks_reports_every_rank_and_compares_the_database_copies():
    """`bench.py --gpus 2 --dry-ranks`: what the first multi-GPU run should be (the `mpirun -np N` of the reference's
    Scripts/submit_MPI-blast.job:24): set-up only, one line per rank -- device, free HBM, broadcast and rebuild seconds, sums
    over the database as it stands on that GPU -- and a verdict; the receiving rank rebuilt its seed index itself, so equal
    sums mean the broadcast AND the local rebuild gave rank 0's index."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-ranks", "--n-seq", "20000"]
    p = subprocess.run(cmd, env=_env(PGX_BENCH_ONE_DEVICE="1", PGX_BENCH_BACKEND="gloo"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [json.loads(l) for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 3, p.stdout
    r0, r1, verdict = lines
    assert (r0["rank"], r1["rank"]) == (0, 1) and r0["checksum"] == r1["checksum"] and all(x > 0 for x in r0["checksum"])
    assert r1["index_rebuild_s"] > 0 and r0["hbm_free_GiB"] > 1
    assert verdict == {"dry_ranks": 2, "ranks_that_differ_from_rank_0": 0, "ranks": []}


@pytest.fixture(scope="module")
def mpi_workload(tmp_path_factory, oracle_bin):
    """3 001 reads (divisible by neither 2 nor 8) against a 1.2 Mbp database; the checker's table is the expected file."""
    from conftest import run_cmd
    d = tmp_path_factory.mktemp("mpi")
    args = ["--n-seq", "2000", "--seq-len", "600", "--n-genus", "50", "--read-len", "150"]
    assert run_cmd([oracle_bin, "synth", "db", "--out", str(d / "db.fa")] + args)[0] == 0
    assert run_cmd([oracle_bin, "synth", "reads", "--out", str(d / "reads.fa"), "--count", "3001"] + args)[0] == 0
    rc, _, se = run_cmd([oracle_bin, "blastn", "-query", str(d / "reads.fa"), "-db", str(d / "db.fa"), "-outfmt", "6", "-out",
                         str(d / "oracle.tsv"), "-num_threads", "8"], timeout=600)
    assert rc == 0, se
    bin_dir = os.path.join(ROOT, "pangea-plus_amd", "bin")
    p = subprocess.run([os.path.join(bin_dir, "makeblastdb"), "-in", str(d / "db.fa"), "-out", str(d / "nt"), "-dbtype", "nucl"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    return d


@pytest.mark.parametrize("n_proc", [1, 2, 8])
def test_mpiblastn_file_verb_with_n_processes_equals_the_one_process_table(mpi_workload, tmp_path, n_proc):
    """`mpiblastn in.fasta db out N` (reference Scripts/submit_MPI-blast.job:24, submit_multiple_MPI-blast.job:24): the
    launcher starts N rank processes (device = rank mod devices: all on GPU 0 of this box, at most four at a time),
    each searches its block of the query file, the concatenation in rank order is byte for byte the checker's table."""
    exe = os.path.join(ROOT, "pangea-plus_amd", "bin", "mpiblastn")
    out = tmp_path / "hits.txt"
    p = subprocess.run([exe, str(mpi_workload / "reads.fa"), str(mpi_workload / "nt"), str(out), str(n_proc)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=_env())
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    want = open(mpi_workload / "oracle.tsv", "rb").read()
    assert len(want) > 100000 and out.read_bytes() == want
    assert not [f for f in os.listdir(tmp_path) if ".rank" in f]  # the rank files are gone


def test_mpiblastn_relays_the_worst_rank_status_and_blastn_takes_a_device(mpi_workload, tmp_path):
    exe = os.path.join(ROOT, "pangea-plus_amd", "bin", "mpiblastn")
    out = tmp_path / "hits.txt"
    p = subprocess.run([exe, str(mpi_workload / "reads.fa"), str(tmp_path / "no_such_db"), str(out), "3"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=_env())
    assert p.returncode == 2 and b"rank" in p.stderr
    # `blastn -rank r -world_size N -gpu g`: the rank no longer implies device 0; a device that does not exist is an error
    blastn = os.path.join(ROOT, "pangea-plus_amd", "bin", "blastn")
    base = [blastn, "-query", str(mpi_workload / "reads.fa"), "-db", str(mpi_workload / "nt"), "-outfmt", "6"]
    parts = b""
    for rk in range(3):
        o = tmp_path / ("b%d.tsv" % rk)
        extra = ["-gpu", "0"] if rk == 1 else []      # rank 2 without -gpu: 2 mod (1 device) = 0
        p = subprocess.run(base + ["-out", str(o), "-rank", str(rk), "-world_size", "3"] + extra, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=300, env=_env())
        assert p.returncode == 0, p.stderr
        parts += o.read_bytes()
    assert parts == open(mpi_workload / "oracle.tsv", "rb").read()
    p = subprocess.run(base + ["-out", str(tmp_path / "x.tsv"), "-gpu", "99"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300, env=_env())
    assert p.returncode == 2 and b"device 99" in p.stderr
