"""GPU: the path a non-root rank takes in bench.py — allocate an empty database of the same shape, receive
the device arrays (here a device-to-device copy through torch tensors that alias library-owned HBM, plus
a 1-rank RCCL broadcast), finish the import — gives a database that answers identically."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_imported_database_answers_identically():
    import torch
    import torch.distributed as dist
    import pangea_plus_amd as pg
    from pangea_plus_amd import _capi
    assert torch.cuda.is_available() and torch.cuda.device_count() >= 1
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda:0")
    pg.init(0)
    cfg = pg.SynthCfg.default(n_seq=1500, seq_len=400, n_genus=40)
    src = pg.Db.from_synth(cfg)
    dst = pg.Db.alloc_like(src.shape())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for (na, va), (nb, vb) in zip(src.device_arrays(), dst.device_arrays()):
            assert na == nb
            ta = torch.as_tensor(va, device="cuda:0")
            tb = torch.as_tensor(vb, device="cuda:0")
            assert ta.data_ptr() == va.__cuda_array_interface__["data"][0]  # zero-copy alias
            dist.broadcast(ta, src=0)                                        # RCCL call on library-owned memory
            tb.copy_(ta)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    dst.finish_import()
    reads = pg.Reads.from_synth(cfg, 0, 800)
    a = _capi.blast_search(src, reads)
    b = _capi.blast_search(dst, reads)
    assert len(a) > 1000
    assert a.format(src, reads) == b.format(dst, reads)
