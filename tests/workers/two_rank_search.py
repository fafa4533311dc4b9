"""Worker of tests/test_gpu_multirank.py: one of N ranks sharing the one GPU of the test box over gloo (RCCL refuses
two ranks on one device; the 8-GPU run uses the same code over RCCL).  Rank 0 builds the database; every other rank
takes the REAL import path (Db.alloc_like -> broadcast into library-owned HBM -> finish_import, which rebuilds block
tables and seed index locally unless PGX_BCAST_INDEX=1 ships them); each rank searches its block of the reads (the
`mpirun -np N` split of Scripts/submit_MPI-blast.job:24) and rank 0 compares the concatenation with its own whole run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import pangea_plus_amd as pg
    from pangea_plus_amd import _capi, sharding
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_path = sys.argv[1]
    n_reads = int(sys.argv[2])
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda:0")
    dist.init_process_group(backend="gloo")
    pg.init(0)
    dev = torch.device("cuda", 0)
    cfg = pg.SynthCfg.default(n_seq=3000, seq_len=700, n_genus=60)
    src = pg.Db.from_synth(cfg) if rank == 0 else None
    db = sharding.broadcast_database(src, rank, world, dist, pg.Db.alloc_like,
                                     lambda d: [(n, torch.as_tensor(v, device=dev)) for n, v in d.device_arrays()],
                                     lambda d: d.finish_import())
    lo, hi = sharding.block_range(n_reads, rank, world)
    reads = pg.Reads.from_synth(cfg, lo, hi - lo)
    hits = _capi.blast_search(db, reads)
    local = hits.format(db, reads)
    whole = sharding.gather_in_rank_order(local, rank, world, dist)
    ok = True
    if rank == 0:
        all_reads = pg.Reads.from_synth(cfg, 0, n_reads)
        single = _capi.blast_search(db, all_reads).format(db, all_reads)
        ok = whole == single and len(single) > 10000
        open(out_path, "wb").write(whole)
        open(out_path + ".status", "w").write("ok %d %d" % (len(whole), whole.count(b"\n")) if ok else "DIFF")
    flag = [ok]
    dist.broadcast_object_list(flag, src=0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if flag[0] else 1)


if __name__ == "__main__":
    main()
