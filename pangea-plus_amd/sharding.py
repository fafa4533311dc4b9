"""Multi-GPU plumbing of the hot path: one process per GPU, reads sharded in contiguous blocks, the database
and its seed index built on rank 0 and broadcast ONCE (RCCL on GPUs; the same code runs over gloo with CPU
tensors in the tests).  No collective touches the per-batch data path."""


def block_range(total, rank, world):
    """Contiguous block `rank` of `world` (same arithmetic as pgx_blastn_run and mpiblastn's static split):
    all hits of a read stay on one rank and concatenating rank outputs in order reproduces the file order."""
    return total * rank // world, total * (rank + 1) // world


def batch_first_read(step, rank, world, batch):
    """Weak scaling: step `s` of rank `r` takes reads [(s*world + r)*batch, +batch) of the stream."""
    return (step * world + rank) * batch


def broadcast_database(db, rank, world, dist, alloc_like, arrays_of, finish_import, src=0):
    """Rank `src` holds `db`; every other rank allocates an empty database of the same shape and receives
    its arrays.  `arrays_of(db)` yields (name, tensor) pairs whose tensors alias the database's memory."""
    if world == 1:
        return db
    shape = [db.shape() if rank == src else None]
    dist.broadcast_object_list(shape, src=src)
    if rank != src:
        db = alloc_like(shape[0])
    names = []
    for name, tensor in arrays_of(db):
        dist.broadcast(tensor, src=src)
        names.append(name)
    if rank != src:
        finish_import(db)
    return db


def gather_in_rank_order(local_bytes, rank, world, dist, dst=0):
    """Concatenate per-rank outputs in rank order on `dst` (order matters for byte-identical files)."""
    if world == 1:
        return local_bytes
    parts = [None] * world if rank == dst else None
    dist.gather_object(local_bytes, parts, dst=dst)
    return b"".join(parts) if rank == dst else None
