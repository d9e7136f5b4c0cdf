"""CPU, gloo, world sizes 2 and 8 (BASELINE config 4 is 8 sub-maps on 8 ranks): the N>1 path of bench.py (submap ownership +
pose all_gather + max-over-ranks timing), the sharded global BA, the particle split and ray-data-parallel training."""
import os
import socket

import pytest
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


@pytest.mark.parametrize("world", [2, 8])
def test_pose_exchange_and_ownership(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    covered = []
    for rank, allp, t, owned, gathered, share in res:
        allp, gathered = torch.from_numpy(allp), torch.from_numpy(gathered)
        assert torch.equal(gathered, torch.arange(33, dtype=torch.float32).reshape(11, 3))
        if world == 2:
            assert share == ((0, 6) if rank == 0 else (6, 11))
        covered += list(range(*share))
        assert 1 <= share[1] - share[0] <= -(-11 // world)        # contiguous, ragged by at most one ray
        assert allp.shape == (world, 4, 7)
        for src in range(world):
            assert torch.equal(allp[src, :, :4], torch.full((4, 4), float(src)) + torch.arange(4)[:, None])
            assert torch.equal(allp[src, :, 4:], torch.full((4, 3), 10.0 * src))
        assert t == 0.5 + world - 1                   # slowest rank
        assert owned == [s for s in range(8) if s % world == rank]
    assert covered == list(range(11))                 # the shares partition the rays in order
    assert sorted(sum((r[3] for r in res), [])) == list(range(8))   # a partition of the submaps


def test_single_process_is_a_noop():
    out = mdist.exchange_poses(torch.zeros(3, 4), torch.ones(3, 3))
    assert out.shape == (1, 3, 7)
    assert mdist.max_over_ranks(2.0, torch.device("cpu")) == 2.0
    assert mdist.submaps_of_rank(5, 1, 0) == [0, 1, 2, 3, 4]


# ------------------------------------------------------------------------------------------------------------------
# SURVEY 8e rows 1 and 3 (VERDICT r1 missing-2): sharded cross-sub-map global BA with the (n-1) x 7 pose-gradient
# all-reduce, and the RandomOptimizer particle split.  The arithmetic core of mipsfusion_amd.global_ba is plain torch
# and its network query is injected, so the collective logic runs here on CPU tensors with the ORACLE's sub-map models.
def _gba_problem(seed=0, n_sub=3, n_iter=4, bs=48):
    """n_sub small sub-maps (hash 2^10) with perturbed anchors, pair terms (k, k + 1) + one loop-closing term (n_sub - 1, 0)."""
    import numpy as np
    from mipsfusion_amd import synth
    from oracle import path_cpu
    cfg = synth.config_plumbing()
    g = torch.Generator().manual_seed(seed)
    models = []
    for s in range(n_sub):
        torch.manual_seed(100 + s)
        m = path_cpu.CpuScene(cfg, cfg["mapping"]["bound"], cfg["mapping"]["localMLP_max_len"])
        with torch.no_grad():
            m.embed_fn.params.copy_(torch.randn(m.embed_fn.params.shape, generator=g) * 0.3)
        models.append(m)
    anchors = torch.eye(4)[None].repeat(n_sub, 1, 1)
    for s in range(1, n_sub):
        anchors[s] = synth.default_pose(cfg, yaw=0.1 * s, pitch=0.02 * s)
        anchors[s, :3, 3] = torch.tensor([0.2 * s, -0.1 * s, 0.05 * s])
    f = synth.make_frame(cfg, seed=3)
    H, W = f["depth"].shape
    batches = []
    for it in range(n_iter):
        terms = []
        for (i, j) in [(k, k + 1) for k in range(n_sub - 1)]:
            idx = torch.randint(0, H * W, (bs,), generator=g)
            r, c = idx // W, idx % W
            rays = torch.cat([f["direction"][r, c], f["rgb"][r, c], f["depth"][r, c][:, None]], -1)
            kf = anchors[j][None].repeat(bs, 1, 1).clone()
            kf[:, :3, 3] += 0.01 * torch.randn(bs, 3, generator=g)
            terms.append((i, j, rays, kf, 5.0, None))
        idx = torch.randint(0, H * W, (bs // 2,), generator=g)
        r, c = idx // W, idx % W
        rays = torch.cat([f["direction"][r, c], f["rgb"][r, c], f["depth"][r, c][:, None]], -1)
        mask = (torch.rand(bs // 2, 1, generator=g) > 0.3).float()
        terms.append((n_sub - 1, 0, rays, anchors[n_sub - 1][None].clone(), 100.0, mask))          # get_SDF_dif2-style term
        batches.append(terms)
    return cfg, models, anchors, batches


def _gba_run(models, anchors, batches, trunc, owned, accum, group=None):
    from mipsfusion_amd.global_ba import PairTerm, ShardedGlobalBA
    ba = ShardedGlobalBA(lambda sid, pts: models[sid].run_network(pts)[..., 3], owned, anchors, trunc,
                         pose_accum_step=accum, group=group)
    trace = []
    for terms in batches:
        trace.append(float(ba.iteration([PairTerm(i, j, rays, kf, w, mask) for (i, j, rays, kf, w, mask) in terms])))
    return ba.result(), trace


def _gba_worker(rank, world, port, q, n_sub, P):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1 if world > 2 else 2)
        cfg, models, anchors, batches = _gba_problem(n_sub=n_sub)
        owned = mdist.submaps_of_rank(len(models), world, rank)                 # 3 / 2: rank 0: {0, 2}, rank 1: {1}
        poses, trace = _gba_run(models, anchors, batches, cfg["training"]["trunc"], owned, accum=2)
        # particle split (row 3): each rank contributes its share of [P, 8] rows
        lo, hi = mdist.share_of(P, rank, world)
        rows = torch.arange(P * 8, dtype=torch.float32).reshape(P, 8)
        full = mdist.gather_particle_results(rows[lo:hi].clone(), P)
        q.put((rank, poses.numpy().copy(), trace, full.numpy().copy(), sorted(owned)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_sub,P", [(2, 3, 11), (8, 8, 2000)])
def test_sharded_global_ba_equals_single_process_and_oracle(world, n_sub, P):
    """Sharded global BA (prediction-table + (n-1)x7 pose-gradient all-reduces) == the single-process run of the same class ==
    oracle/global_ba_cpu.py's restatement of InactiveMap.global_BA_overlapping; two ranks with three sub-maps, and BASELINE
    config 4's shape: eight sub-maps, one per rank, eight anchors.  Beside it the RandomOptimizer's particle split: 11 rows over
    2 ranks, the reference's 2000 particles over 8."""
    from oracle import global_ba_cpu
    cfg, models, anchors, batches = _gba_problem(n_sub=n_sub)
    trunc = cfg["training"]["trunc"]
    ref_poses, ref_trace = global_ba_cpu.optimise(models, anchors, batches, trunc, pose_accum_step=2)
    one_poses, one_trace = _gba_run(models, anchors, batches, trunc, owned=range(len(models)), accum=2)
    assert torch.allclose(one_poses, ref_poses, atol=1e-6) and torch.allclose(torch.tensor(one_trace),
                                                                             torch.tensor(ref_trace), rtol=1e-5)
    assert not torch.allclose(ref_poses[1:], anchors[1:], atol=1e-4), "the anchors must actually move"
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gba_worker, args=(r, world, port, q, n_sub, P)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    if world == 2:
        assert res[0][4] == [0, 2] and res[1][4] == [1]
    else:
        assert [r[4] for r in res] == [[k] for k in range(8)]          # one sub-map per rank
    for rank, poses, trace, full, _ in res:
        assert torch.allclose(torch.from_numpy(poses), ref_poses, atol=2e-6), f"rank {rank} anchors"
        assert torch.allclose(torch.tensor(trace), torch.tensor(ref_trace), rtol=1e-5), f"rank {rank} loss trace"
        assert torch.equal(torch.from_numpy(full), torch.arange(P * 8, dtype=torch.float32).reshape(P, 8))
    for r in res[1:]:
        assert (res[0][1] == r[1]).all(), "every rank must hold bit-identical anchors (same all-reduced gradient)"


# ------------------------------------------------------------------------ ray-data-parallel training
class _ToyScene(torch.nn.Module):
    """the two parameter families of a sub-map -- one flat table (embed_fn.params) and a few small decoder tensors -- under an
    objective built like the reference's (helper_functions/utils.py:43-47, scene_rep.py:218): per-ray squared errors summed
    over the BATCH, one term weighted by 1 - n_front / n_rays with an integer count over the batch, one term averaged over the
    batch's `valid` rays only.  ``ray_share_reduce`` (set by RayDataParallelStep) sums the share's numbers over the ranks."""

    def __init__(self, n_table):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.embed_fn = torch.nn.Module()
        self.embed_fn.params = torch.nn.Parameter(torch.randn(n_table, generator=g) * 0.1)
        self.decoder = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))
        with torch.no_grad():
            for p in self.decoder.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
        self.ray_share_reduce = None

    def objective(self, x, x_raw):          # x [n, 3] (posed), x_raw: the rays as drawn (masks are decided on the data)
        feat = self.embed_fn.params[(x_raw[:, 0].abs() * 1000).long() % self.embed_fn.params.numel()]
        per_ray = (self.decoder(x) * feat[:, None]).pow(2).sum(-1) + feat.pow(2)
        front, valid = x_raw[:, 1] > 0, x_raw[:, 2] > -0.5
        s_front, s_valid = (per_ray * front).sum(), (per_ray.sqrt() * valid).sum()
        counts = torch.tensor([float(front.sum()), float(valid.sum()), float(x.shape[0])], dtype=torch.float64)
        if self.ray_share_reduce is not None:
            counts = self.ray_share_reduce(counts)          # the batch's counts; the sums stay this share's (their
        n_front, n_valid, n = (float(c) for c in counts)    # gradients are added over the ranks by step())
        return s_front * ((1.0 - n_front / n) / n) + s_valid / n_valid


def _toy_optimisers():
    return (lambda shard: torch.optim.Adam([shard], lr=0.01, betas=(0.9, 0.99), eps=1e-15),
            lambda ps: torch.optim.Adam(ps, lr=0.01, betas=(0.9, 0.99), weight_decay=1e-6),
            lambda ps: torch.optim.Adam(ps, lr=1e-3))


def _ray_dp_worker(rank, world, port, q, n_table, n_rays):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mipsfusion_amd.ray_dp import RayDataParallelStep
        torch.manual_seed(0)
        torch.set_num_threads(1)
        m = _ToyScene(n_table)
        pose = torch.nn.Parameter(torch.tensor([0.1, -0.2, 0.3]))
        g_opt, d_opt, p_opt = _toy_optimisers()
        rdp = RayDataParallelStep(m, g_opt, d_opt, [pose], p_opt)
        assert m.ray_share_reduce is not None
        gen = torch.Generator().manual_seed(9)
        for it in range(4):
            x = torch.randn(n_rays, 3, generator=gen)             # the SAME batch on every rank ...
            b, e = rdp.my_share(n_rays)                               # ... of which each renders its (ragged) share
            (m.objective(x[b:e] + pose, x[b:e])).backward()
            if it == 2:                                           # the replica is reloaded between steps (recover_initial_param)
                with torch.no_grad():
                    m.embed_fn.params.mul_(0.5)
            rdp.step(pose=(it + 1) % 2 == 0)
        q.put((rank, m.embed_fn.params.detach().numpy().copy(), [p.detach().numpy().copy() for p in m.decoder.parameters()],
               pose.detach().numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_table,n_rays", [(2, 12, 11), (2, 11, 11), (8, 16, 19), (8, 11, 19)])
def test_ray_data_parallel_step_equals_the_single_process_step(world, n_table, n_rays):
    """RayDataParallelStep over gloo ranks (reduce-scatter -> sharded Adam -> all-gather; the batch's counts summed inside the
    forward, the shares' gradients SUMMED): all ranks end with bit-identical parameters, equal to ONE process that runs the same
    objective on the WHOLE batch and the plain optimisers -- the step mipsfusion.py:325-335 takes.  Two ranks and eight; a table
    that divides by the world size and one that does not (the padded path, whose private copy must follow a reload of the
    replica); ragged ray shares."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ray_dp_worker, args=(r, world, port, q, n_table, n_rays)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, t0, d0, p0) = res[0]
    for (_, t1, d1, p1) in res[1:]:
        assert (t0 == t1).all() and all((a == b).all() for a, b in zip(d0, d1)) and (p0 == p1).all(), "ranks diverged"
    torch.manual_seed(0)
    m = _ToyScene(n_table)
    pose = torch.nn.Parameter(torch.tensor([0.1, -0.2, 0.3]))
    g_opt, d_opt, p_opt = _toy_optimisers()
    og, od, op_ = g_opt(m.embed_fn.params), d_opt(list(m.decoder.parameters())), p_opt([pose])
    gen = torch.Generator().manual_seed(9)
    for it in range(4):
        x = torch.randn(n_rays, 3, generator=gen)
        m.objective(x + pose, x).backward()                       # the whole batch, one process
        if it == 2:
            with torch.no_grad():
                m.embed_fn.params.mul_(0.5)
        og.step(), od.step()
        og.zero_grad(), od.zero_grad()
        if (it + 1) % 2 == 0:
            op_.step()
            op_.zero_grad()
    import numpy as np
    np.testing.assert_allclose(t0, m.embed_fn.params.detach().numpy(), rtol=0, atol=1e-7)
    for a, b in zip(d0, m.decoder.parameters()):
        np.testing.assert_allclose(a, b.detach().numpy(), rtol=0, atol=1e-7)
    np.testing.assert_allclose(p0, pose.detach().numpy(), rtol=0, atol=1e-7)
