"""CPU, world_size 2, gloo: the N>1 path of bench.py (submap ownership + pose all_gather + max-over-ranks timing)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mipsfusion_amd import dist as mdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        K = 4
        rot = torch.full((K, 4), float(rank)) + torch.arange(K)[:, None]
        trans = torch.full((K, 3), 10.0 * rank)
        allp = mdist.exchange_poses(rot, trans)
        t = mdist.max_over_ranks(0.5 + rank, torch.device("cpu"))
        # ray-data-parallel inference: contiguous shares of 11 rays, gathered back in order
        from mipsfusion_amd.inference import share_of
        b, e = share_of(11, rank, world)
        whole = torch.arange(11 * 3, dtype=torch.float32).reshape(11, 3)
        gathered = mdist.all_gather_ragged(whole[b:e].clone(), 11, world)
        # numpy, not torch tensors: a tensor travels through the queue as a file descriptor owned by THIS process,
        # and the parent may only open it after the worker has exited (FileNotFoundError, seen once in ~50 runs)
        q.put((rank, allp.numpy().copy(), t, mdist.submaps_of_rank(8, world, rank), gathered.numpy().copy(), (b, e)))
    finally:
        dist.destroy_process_group()


def test_pose_exchange_and_ownership_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, allp, t, owned, gathered, share in res:
        allp, gathered = torch.from_numpy(allp), torch.from_numpy(gathered)
        assert torch.equal(gathered, torch.arange(33, dtype=torch.float32).reshape(11, 3))
        assert share == ((0, 6) if rank == 0 else (6, 11))
        assert allp.shape == (2, 4, 7)
        for src in range(2):
            assert torch.equal(allp[src, :, :4], torch.full((4, 4), float(src)) + torch.arange(4)[:, None])
            assert torch.equal(allp[src, :, 4:], torch.full((4, 3), 10.0 * src))
        assert t == 1.5                               # slowest rank
        assert owned == [s for s in range(8) if s % 2 == rank]
    assert sorted(res[0][3] + res[1][3]) == list(range(8))   # a partition of the submaps


def test_single_process_is_a_noop():
    out = mdist.exchange_poses(torch.zeros(3, 4), torch.ones(3, 3))
    assert out.shape == (1, 3, 7)
    assert mdist.max_over_ranks(2.0, torch.device("cpu")) == 2.0
    assert mdist.submaps_of_rank(5, 1, 0) == [0, 1, 2, 3, 4]
