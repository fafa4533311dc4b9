"""The N > 1 path on CPU: 2, 4 and 8 gloo ranks run the same sharding/broadcast code bench.py runs over RCCL."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeDb:
    """Stands in for a device database: same shape()/arrays protocol, CPU tensors."""

    def __init__(self, shape, fill):
        self._shape = shape
        g = torch.Generator().manual_seed(1234)
        self.arrays = [("words", torch.randint(0, 255, (shape[1] // 4,), dtype=torch.uint8, generator=g)),
                       ("seq_off", torch.arange(shape[0] + 1, dtype=torch.int32).view(torch.uint8)),
                       ("postings", torch.randint(0, 255, (shape[4] * 4,), dtype=torch.uint8, generator=g))]
        if not fill:
            self.arrays = [(n, torch.zeros_like(t)) for n, t in self.arrays]
        self.imported = False

    def shape(self):
        return self._shape


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from pangea_plus_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shape = (100, 4000, 0, 22, 3985, 1)
        src = FakeDb(shape, fill=True) if rank == 0 else None

        def finish(d):
            d.imported = True
        db = sharding.broadcast_database(src, rank, world, dist, lambda s: FakeDb(s, fill=False),
                                         lambda d: d.arrays, finish)
        ref = FakeDb(shape, fill=True)
        ok = db.shape() == shape and all(torch.equal(a[1], b[1]) for a, b in zip(db.arrays, ref.arrays))
        ok = ok and (rank == 0 or db.imported)
        # read sharding: each rank "classifies" its block; rank 0 concatenates in rank order
        total = 1003
        lo, hi = sharding.block_range(total, rank, world)
        local = b"".join(b"r%d\thit\n" % i for i in range(lo, hi))
        whole = sharding.gather_in_rank_order(local, rank, world, dist)
        if rank == 0:
            ok = ok and whole == b"".join(b"r%d\thit\n" % i for i in range(total))
        # weak-scaling batches never overlap
        firsts = [sharding.batch_first_read(s, rank, world, 10) for s in range(3)]
        allf = [None] * world
        dist.all_gather_object(allf, firsts)
        flat = sorted(x for f in allf for x in f)
        ok = ok and flat == [10 * k for k in range(3 * world)]
        open(os.path.join(out_dir, "rank%d.ok" % rank), "w").write("1" if ok else "0")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gloo_broadcast_and_sharding(tmp_path, world):
    """World sizes 2, 4 and 8 (config 4 is 8 ranks): 1 003 reads do not divide by any of them."""
    port = 29600 + os.getpid() % 300 + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / ("rank%d.ok" % r)).read_text() == "1"


def test_block_range_matches_cli_partition():
    sys.path.insert(0, ROOT)
    from pangea_plus_amd import sharding
    for total in (0, 1, 7, 1003):
        for world in (1, 2, 3, 8):
            cuts = [sharding.block_range(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """bench.py --gpus N must run N ranks or fail: never N = 1 silently (checked before anything touches a GPU)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b"WORLD_SIZE" in p.stderr


def test_bench_parent_starts_the_ranks_as_a_child_process(tmp_path):
    """`python bench.py --gpus 2` without a launcher: the parent must start torch.distributed.run as a child (a stub
    interpreter records the command line here: no GPU in this container)."""
    import subprocess
    stub = tmp_path / "python"
    stub.write_text("#!/bin/sh\necho \"$@\" > %s/argv.txt\nexit 7\n" % tmp_path)
    stub.chmod(0o755)
    code = ("import sys, runpy; sys.executable = %r; sys.argv = ['bench.py', '--gpus', '2', '--steps', '3']; "
            "runpy.run_path(%r, run_name='__main__')" % (str(stub), os.path.join(ROOT, "bench.py")))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 7, p.stderr  # the child's exit status is relayed
    argv = (tmp_path / "argv.txt").read_text().split()
    assert argv[:2] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in argv
    assert argv[argv.index("--nproc-per-node") + 1] == "2" and argv[-4:] == ["--gpus", "2", "--steps", "3"]


def test_mpiblastn_without_a_gpu_fails_loudly_and_leaves_the_output_alone(tmp_path):
    """`mpiblastn in.fasta db out N` (reference Scripts/submit_MPI-blast.job:24): the launcher asks a child process for
    the device count; with no GPU it says so, exits non-zero and does not create (or truncate) the output file."""
    import subprocess
    exe = os.path.join(ROOT, "pangea-plus_amd", "bin", "mpiblastn")
    if not os.path.exists(exe):
        pytest.skip("CLIs not built")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    q = tmp_path / "q.fa"
    q.write_text(">a\nACGT\n")
    out = tmp_path / "out.txt"
    out.write_text("keep me\n")
    p = subprocess.run([exe, str(q), str(tmp_path / "db"), str(out), "8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 2 and b"no HIP device" in p.stderr
    assert out.read_text() == "keep me\n"
    p = subprocess.run([exe, str(q), str(tmp_path / "db"), str(out), "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 1 and b"USAGE" in p.stderr
    p = subprocess.run([exe, str(tmp_path / "nope.fa"), str(tmp_path / "db"), str(out), "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 2 and b"cannot open query file" in p.stderr
