"""World-size-2 rehearsal of the view-parallel gradient exchange on CPU (gloo)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neuralgaussiansplatting_amd import view_parallel


def _worker(rank, world, port, P, M, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = view_parallel.GradientBucket(P, M, torch.device("cpu"))
        # each rank "renders" a different view: its gradients are a known function of the rank
        for i, (name, v) in enumerate(b.views.items()):
            v.copy_(torch.arange(v.numel(), dtype=torch.float32) * (rank + 1) + i)
        b.all_reduce()
        ok = True
        for i, (name, v) in enumerate(b.views.items()):
            want = torch.arange(v.numel(), dtype=torch.float32) * sum(r + 1 for r in range(world)) + i * world
            ok = ok and torch.equal(v, want)
        sh = b.shaped()
        ok = ok and sh["dL_dsh"].shape == (P, M, 3) and sh["dL_drotations"].shape == (P, 4)
        ok = ok and sh["dL_dmeans3D"].data_ptr() == b.flat.data_ptr()  # views, not copies
        b2 = view_parallel.GradientBucket(P, M, torch.device("cpu"))
        b2.flat.fill_(float(rank))
        b2.all_reduce(average=True)
        ok = ok and torch.allclose(b2.flat, torch.full_like(b2.flat, (world - 1) / 2.0))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_bucket_all_reduce_world2():
    world, P, M = 2, 37, 16
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, P, M, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_bucket_layout_and_sharding():
    lay, total = view_parallel.bucket_layout(100, 16)
    assert total == 100 * 59  # 3 + 48 + 1 + 3 + 4 floats per Gaussian
    assert [lay[k][0] for k in view_parallel.BUCKET_FIELDS] == [0, 300, 5100, 5200, 5500]
    assert view_parallel.shard_views(8, 1, 4) == [1, 5]
    assert sorted(sum((view_parallel.shard_views(10, r, 3) for r in range(3)), [])) == list(range(10))
    b = view_parallel.GradientBucket(5, 1, torch.device("cpu"))
    assert b.all_reduce() is None  # no process group: nothing to exchange
