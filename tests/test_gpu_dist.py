"""The RCCL entry points of the multi-GPU paths on the ONE GPU a test box has: a one-rank ``nccl`` process group with
``mipsfusion_amd.dist.FORCE_COLLECTIVES`` set drives ``exchange_poses`` (all_gather), ``all_reduce_sum_`` / ``max_over_ranks``
(all_reduce), ``gather_particle_results`` (all_gather of padded blocks) and ``ShardedFlatAdam.step`` / ``RayDataParallelStep``
(reduce_scatter_tensor, all_gather_into_tensor, flattened all_reduce) through the real collectives; every result must equal
what the same calls give without a process group.  What this cannot show is a second rank (tests/test_dist_cpu.py: gloo, world
2 and 8) or xGMI (no multi-GPU node has been available): it shows that the nccl branches run -- tensor placement, in-place
all-gather into the replica's own storage, dtype / contiguity requirements -- before the first 8-GPU run meets them.
(the unit being sharded: InactiveMap.py:203-308, mipsfusion.py:320-335, RandomOptimizer.py:196-224.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Toy(torch.nn.Module):
    """one flat table + a small decoder (the two parameter families RayDataParallelStep shards / replicates)"""

    def __init__(self, n_table, dev):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.embed_fn = torch.nn.Module()
        self.embed_fn.params = torch.nn.Parameter((torch.randn(n_table, generator=g) * 0.1).to(dev))
        self.decoder = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2)).to(dev)
        with torch.no_grad():
            for p in self.decoder.parameters():
                p.copy_((torch.randn(p.shape, generator=g) * 0.3).to(dev))
        self.ray_share_reduce = None

    def objective(self, x):
        feat = self.embed_fn.params[(x[:, 0].abs() * 1000).long() % self.embed_fn.params.numel()]
        per_ray = (self.decoder(x) * feat[:, None]).pow(2).sum(-1) + feat.pow(2)
        counts = torch.tensor([float(x.shape[0])], dtype=torch.float64, device=x.device)
        if self.ray_share_reduce is not None:
            counts = self.ray_share_reduce(counts)
        return per_ray.sum() / float(counts[0])


def _run_steps(dev, forced):
    """four optimiser steps of the toy scene through RayDataParallelStep; the parameters afterwards"""
    from mipsfusion_amd import dist as mdist
    from mipsfusion_amd.ray_dp import RayDataParallelStep
    mdist.FORCE_COLLECTIVES = forced
    m = _Toy(4096 + 11, dev)
    pose = torch.nn.Parameter(torch.tensor([0.1, -0.2, 0.3], device=dev))
    rdp = RayDataParallelStep(m, lambda shard: torch.optim.Adam([shard], lr=0.01, betas=(0.9, 0.99), eps=1e-15),
                              lambda ps: torch.optim.Adam(ps, lr=0.01, betas=(0.9, 0.99), weight_decay=1e-6), [pose],
                              lambda ps: torch.optim.Adam(ps, lr=1e-3))
    gen = torch.Generator().manual_seed(9)
    for it in range(4):
        x = torch.randn(64, 3, generator=gen).to(dev)
        m.objective(x + pose).backward()
        rdp.step(pose=(it + 1) % 2 == 0)
    rdp.close()
    mdist.FORCE_COLLECTIVES = False
    return [m.embed_fn.params.detach().cpu().numpy().copy()] + [p.detach().cpu().numpy().copy() for p in m.decoder.parameters()] + \
           [pose.detach().cpu().numpy().copy()]


def _worker(port, q):
    import torch.distributed as dist
    from mipsfusion_amd import dist as mdist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = {}
    try:
        out["plain"] = _run_steps(dev, forced=False)                 # no process group: the skipped path
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)     # (bench.py's call)
        out["backend"] = dist.get_backend()
        mdist.FORCE_COLLECTIVES = True
        rot, trans = torch.rand(5, 4, device=dev), torch.rand(5, 3, device=dev)
        allp = mdist.exchange_poses(rot, trans)                                                    # all_gather
        out["poses_ok"] = bool(allp.shape == (1, 5, 7) and torch.equal(allp[0], torch.cat([rot, trans], -1)) and allp.is_cuda)
        t = torch.arange(21, dtype=torch.float32, device=dev).reshape(3, 7)
        out["all_reduce_ok"] = bool(torch.equal(mdist.all_reduce_sum_(t.clone()), t))             # all_reduce (SUM)
        out["max_ok"] = mdist.max_over_ranks(1.25, dev) == 1.25                                    # all_reduce (MAX, fp64)
        rows = torch.rand(2000, 8, device=dev)
        full = mdist.gather_particle_results(rows, 2000)                                           # all_gather, padded blocks
        out["particles_ok"] = bool(full.shape == rows.shape and torch.equal(full, rows))
        sums = torch.rand(10, dtype=torch.float64, device=dev)                                     # the nine loss sums + a count
        out["fp64_ok"] = bool(torch.equal(mdist.all_reduce_sum_(sums.clone()), sums))
        out["forced"] = _run_steps(dev, forced=True)       # reduce_scatter_tensor + all_gather_into_tensor + flat all_reduce
        torch.cuda.synchronize()
    except Exception as e:      # noqa: BLE001
        import traceback
        out["error"] = f"{e}\n{traceback.format_exc()}"
    finally:
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:       # noqa: BLE001
            pass
    q.put(out)


def test_rccl_entry_points_run_on_a_one_rank_group_and_change_nothing():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests must run on the MI355X box")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=600)
    p.join(120)
    assert "error" not in out, out.get("error")
    assert p.exitcode == 0
    assert out["backend"] == "nccl"
    for k in ("poses_ok", "all_reduce_ok", "max_ok", "particles_ok", "fp64_ok"):
        assert out[k], k
    # one rank: the sum over the ranks is the rank's own gradient, the reduce-scatter / all-gather round trip the identity --
    # bit for bit the step without a process group
    for a, b in zip(out["plain"], out["forced"]):
        np.testing.assert_array_equal(a, b)
