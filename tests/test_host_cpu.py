"""CPU: the C-ABI library loads and exports everything include/mipsf.h declares, host-only entry points agree
with the oracle, and the host-side mirrors (samplers, pose helpers) reproduce the reference's golden vectors.
No compute kernel is launched here."""
import ctypes as C
import math
import os
import time
import random
import re

import numpy as np
import pytest
import torch

from mipsfusion_amd import _lib
from mipsfusion_amd.helper_functions import geometry_helper as gh
from mipsfusion_amd.helper_functions import sampling_helper as sh
from oracle import tcnn_cpu

from .conftest import ROOT, load_golden


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mipsf.h")).read()
    declared = set(re.findall(r"\b(mipsf_[a-z0-9_]+)\s*\(", header))
    declared -= {"mipsf_grid_meta", "mipsf_decoder_weights", "mipsf_decoder_grads", "mipsf_render_cfg"}
    assert 25 <= len(declared) <= 45, "one argument block per kernel family, not a suffix per option (review of round 4)"
    assert not [n for n in declared if re.search(r"_ex\d*$|_v$|_keep$|_tiles$", n)]
    handle = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in mipsf.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_abi_version_and_error_channel():
    lib = _lib.lib()
    assert lib.mipsf_abi_version() == 2
    m = _lib.GridMeta()
    rc = lib.mipsf_hashgrid_meta_init(C.byref(m), 16, 4, 19, 16, 1.2)
    assert rc != 0 and b"n_features" in lib.mipsf_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc, "meta")


def test_size_queries_and_failed_calls_reset_the_kept_blocks():
    """Pure host entry points: the loss kernel's scratch is sized for BOTH of its forms (8 floats per ray, or one row of nine
    doubles per 16-ray workgroup: the larger for N < 3 -- a caller with exact allocations used to be overrun), the bf16x6 operand
    buffer is the f16 one plus its plane-2 extension; a failed call empties every cache of caller-kept counter blocks."""
    lib = _lib.lib()
    assert [_lib.buffer_size(_lib.SIZE_RENDER_PARTIAL, n) for n in (1, 2, 3, 16, 17, 4096)] == [18, 18, 24, 128, 136, 32768]
    f16, bf = (_lib.buffer_size(_lib.SIZE_DECODER_PACKED16, 0, _lib.PREC[k]) for k in ("f16x3", "bf16x6"))
    assert f16 == _lib.buffer_size(_lib.SIZE_DECODER_PACKED16, 0, _lib.PREC["f16"]) and bf > f16
    with pytest.raises(RuntimeError, match="unknown buffer"):
        _lib.buffer_size(99)
    # an argument block of another size (a caller built against another header) is refused before anything is launched
    a = _lib.HashgridBwdArgs.new()
    a.struct_size += 8
    assert lib.mipsf_hashgrid_bwd(C.byref(a), None) != 0 and b"struct_size" in lib.mipsf_last_error()
    for cls, fn in ((_lib.DecoderFwd16Args, lib.mipsf_decoder_fwd16), (_lib.DecoderChain16Args, lib.mipsf_decoder_bwd_chain16),
                    (_lib.DecoderWgrad16Args, lib.mipsf_decoder_wgrad16), (_lib.RenderFwdArgs, lib.mipsf_render_fwd),
                    (_lib.RenderBwdArgs, lib.mipsf_render_bwd)):
        blk = cls.new()
        blk.struct_size -= 4
        assert fn(C.byref(blk), None) != 0 and b"struct_size" in lib.mipsf_last_error()
        assert fn(None, None) != 0
    # ... but the two render blocks are also taken in their first form (struct_size up to the field before `draw` / `flags`, added
    # later in this ABI version): an empty batch is accepted without touching a device
    for cls, fn, field in ((_lib.RenderFwdArgs, lib.mipsf_render_fwd, "draw"), (_lib.RenderBwdArgs, lib.mipsf_render_bwd, "flags")):
        blk = cls.new()
        assert fn(C.byref(blk), None) == 0
        blk.struct_size = getattr(cls, field).offset
        assert fn(C.byref(blk), None) == 0, lib.mipsf_last_error()
    from mipsfusion_amd import ops
    ops._ZEROED[("probe",)] = object()
    with pytest.raises(RuntimeError):
        _lib.check(lib.mipsf_hashgrid_meta_init(C.byref(_lib.GridMeta()), 16, 4, 19, 16, 1.2), "meta")
    assert not ops._ZEROED and not ops._SCATTER_COUNTERS and not ops._POSE_SCRATCH


@pytest.mark.parametrize("log2_t", [10, 16, 19])
def test_level_table_equals_oracle(log2_t):
    pls = float(2.0 ** (math.log2(256 / 16) / 15))
    m = _lib.make_grid_meta(16, 2, log2_t, 16, pls)
    o = tcnn_cpu.make_grid_meta(16, 2, log2_t, 16, pls)
    assert list(m.offsets[:17]) == o.offsets
    assert list(m.resolutions[:16]) == o.resolutions
    assert np.array_equal(np.array(m.scales[:16], dtype=np.float32), np.array(o.scales, dtype=np.float32))
    assert m.n_params == o.n_params
    assert np.float32(m.log2_per_level_scale) == np.float32(o.log2_per_level_scale)


def test_product_ops_refuse_cpu_tensors():
    from mipsfusion_amd.model import get_encoder
    enc, dim = get_encoder("HashGrid", log2_hashmap_size=10, desired_resolution=256)
    assert dim == 32 and enc.params.numel() == 32768
    with pytest.raises(RuntimeError, match="GPU"):
        enc(torch.rand(4, 3))
    freq, fdim = get_encoder("Frequency", n_bins=8)
    assert fdim == 48 and freq.params.numel() == 0


def test_state_dict_keys_match_reference_interface():
    from mipsfusion_amd import synth
    from mipsfusion_amd.model import JointEncoding
    cfg = synth.config_plumbing()
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
    m = JointEncoding(cfg, bb, torch.tensor([7.0, 7.0, 7.0], dtype=torch.float64))
    g = load_golden("scene_cfg1.npz")
    ref_keys = {k[2:] for k in g.files if k.startswith("w.")}
    assert set(m.state_dict().keys()) == ref_keys
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == g["w." + k].shape, k
    m.load_state_dict({k: torch.from_numpy(g["w." + k]) for k in ref_keys})     # reference checkpoint loads
    m.recover_initial_param()


# --------------------------------------------------------------------------- samplers (a1)
def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def test_pixel_samplers_bit_exact():
    g = load_golden("sampler.npz")
    depth = torch.from_numpy(g["depth"])
    H, W = depth.shape
    r, c = sh.sample_pixels_uniformly(460, 620, 16, 24)
    assert np.array_equal(r.numpy(), g["uniform_460x620_16x24_rows"])
    assert np.array_equal(c.numpy(), g["uniform_460x620_16x24_cols"])
    r, c = sh.sample_pixels_uniformly(H, W, 4, 6)
    assert np.array_equal(r.numpy(), g["uniform_small_rows"]) and np.array_equal(c.numpy(), g["uniform_small_cols"])
    seed_all(11)
    assert np.array_equal(sh.sample_pixels_random(H, W, 100).numpy(), g["random_seed11_n100"])
    seed_all(12)
    assert np.array_equal(sh.sample_valid_pixels_random(depth, 64).numpy(), g["valid_random_seed12_n64"])
    seed_all(13)
    r, c = sh.sample_pixels_mix(H, W, 4, 6, depth, 120)
    assert np.array_equal(r.numpy(), g["mix_seed13_rows"]) and np.array_equal(c.numpy(), g["mix_seed13_cols"])
    assert np.array_equal(sh.pixel_rc_to_indices(r, c, H, W).numpy(), g["mix_seed13_indices"])
    seed_all(14)
    assert np.array_equal(sh.select_samples(H, W, 50).numpy(), g["select_samples_seed14_n50"])


def test_pixel_sampler_edge_cases():
    depth = torch.zeros(8, 8)
    depth[2, 3] = 1.0
    seed_all(0)
    idx = sh.sample_valid_pixels_random(depth, 1)
    assert idx.item() == 2 * 8 + 3                      # the only valid pixel wins the top-k
    r, c = sh.sample_pixels_uniformly(8, 8, 8, 8)        # every pixel: gap 0
    assert np.array_equal(sh.pixel_rc_to_indices(r, c, 8, 8).numpy(), np.arange(64))
    rr, cc = sh.pixel_indices_to_rc(torch.tensor([0, 7, 8, 63]), 8, 8)
    assert rr.tolist() == [0, 0, 1, 7] and cc.tolist() == [0, 7, 0, 7]


def test_pose_helpers_match_golden():
    g = load_golden("quaternion.npz")
    rot = torch.from_numpy(g["rot"]).requires_grad_(True)
    trans = torch.from_numpy(g["trans"]).requires_grad_(True)
    T = gh.qt_to_transform_matrix(rot, trans)
    np.testing.assert_allclose(T.detach().numpy(), g["T"], rtol=1e-6, atol=1e-7)
    T.backward(torch.from_numpy(g["gT"]))
    np.testing.assert_allclose(rot.grad.numpy(), g["d_rot"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(trans.grad.numpy(), g["d_trans"], rtol=1e-6)
    np.testing.assert_allclose(gh.matrix_to_quaternion(torch.from_numpy(g["T"])[:, :3, :3]).numpy(), g["q_back"],
                               rtol=1e-6, atol=1e-7)


def test_reference_checkpoint_file_loads_and_round_trips(tmp_path):
    """tests/golden/ref_model_0.pth was written by the reference's own JointEncoding with
    torch.save(model.state_dict()) (Logger.py:33-34).  It must load into this package's JointEncoding with strict
    keys, and a checkpoint written here must have exactly the reference's keys, shapes and dtypes."""
    import os
    import numpy as np
    from mipsfusion_amd import checkpoint, synth
    from mipsfusion_amd.model import JointEncoding
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ref_sd = torch.load(os.path.join(here, "ref_model_0.pth"))
    assert tuple(ref_sd.keys()) == checkpoint.EXPECTED_KEYS
    cfg = synth.config_plumbing()
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
    nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
    m = JointEncoding(cfg, bb, nf)
    checkpoint.load_state_dict(m, os.path.join(here, "ref_model_0.pth"))
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref_sd[k]), k
    out = tmp_path / "model_0.pth"
    checkpoint.save_state_dict(m, out)
    mine = torch.load(out)
    assert list(mine.keys()) == list(ref_sd.keys())
    for k in mine:
        assert mine[k].shape == ref_sd[k].shape and mine[k].dtype == ref_sd[k].dtype and torch.equal(mine[k], ref_sd[k])
    m2 = JointEncoding(cfg, bb, nf)
    checkpoint.copy_parameters_(m2, m)
    for (k, a), (_, b) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), k
    with pytest.raises(RuntimeError):
        torch.save({"x": torch.zeros(1)}, tmp_path / "bad.pth")
        checkpoint.load_state_dict(m, tmp_path / "bad.pth")


def test_deepcopy_of_the_model_draws_no_random_numbers():
    """copy.deepcopy(model) (InactiveMap.py:67,81,107) must leave the CPU RNG where it was: the reference's copy
    draws nothing, and the pixel-sampling index stream depends on the generator state (found by the config-3
    sequence test)."""
    import copy
    import numpy as np
    from mipsfusion_amd import synth
    from mipsfusion_amd.model import JointEncoding
    cfg = synth.config_plumbing()
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
    nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
    torch.manual_seed(3)
    m = JointEncoding(cfg, bb, nf)
    state = torch.get_rng_state()
    m2 = copy.deepcopy(m)
    assert torch.equal(torch.get_rng_state(), state)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("iter_ro", [5, 0])
def test_sample_producer_reproduces_the_sequential_index_stream(iter_ro):
    """mipsfusion_amd.sequence.ReferenceSampleProducer runs the torch-CPU and python generator streams in worker
    threads, one frame ahead; its index sets and jitter must be bit-identical to calling the sampling functions one
    after the other in the reference's program order (tracking_render mipsfusion.py:508-534, local_BA :293-317) from
    the same seeds -- for both sampling branches (iter_RO > 0: sample_pixels_mix; iter_RO == 0: select_samples /
    sample_valid_pixels_random).  The functions themselves are pinned by sampler.npz / keyframe_rays.npz."""
    import random
    import numpy as np
    from mipsfusion_amd import sequence, synth
    cfg = synth.config_headline()
    cfg["tracking"].update(iter_RO=iter_ro, ignore_edge_H=20, ignore_edge_W=20, iter=3)
    cfg["mapping"].update(iters=4)
    H, W = 120, 160
    R = 5000
    plans = []
    for k in range(1, 6):
        g = torch.Generator().manual_seed(k)
        depth = torch.rand(H, W, generator=g) * 3
        depth[torch.rand(H, W, generator=g) < 0.05] = 0.0
        related = torch.arange(1 + k // 2) if k % 2 == 0 else None          # BA on frames 2 and 4 with K = 2, 3
        plans.append(sequence.FramePlan(k, depth, True, related, 7 * R))
    random.seed(1), np.random.seed(1), torch.manual_seed(1)
    want = sequence.sequential_reference_samples(cfg, H, W, R, plans)
    state_after = (torch.get_rng_state(), random.getstate())
    random.seed(1), np.random.seed(1), torch.manual_seed(1)
    prod = sequence.ReferenceSampleProducer(cfg, H, W, R, max_related=4, pinned=False, slots=2)
    prod.submit(plans[0])
    for k, (plan, rec) in enumerate(zip(plans, want)):
        if k + 1 < len(plans):
            prod.submit(plans[k + 1])                                        # one frame ahead, as the loop does
        s = prod.get()
        assert s.frame_id == plan.frame_id
        assert torch.equal(s.track_idx, rec["track_idx"])
        assert torch.equal(s.track_noise, rec["track_noise"])
        if plan.ba_kf_ids is not None:
            n = rec["ba_rows"].shape[1]
            assert s.n_ba == n
            rows, owner, noise = s.ba_packed()
            assert torch.equal(rows, rec["ba_rows"]) and torch.equal(owner, rec["ba_owner"])
            assert torch.equal(noise, rec["ba_noise"])
        prod.release(s)
    prod.close()
    for t in prod._threads:
        t.join(10)
    assert torch.equal(torch.get_rng_state(), state_after[0]) and random.getstate() == state_after[1], \
        "both generators must end where the sequential program leaves them"


def test_numa_confinement_picks_a_subset_of_the_allowed_cpus():
    """mipsfusion_amd.hostcpu: the chosen CPUs are allowed ones, at most max_cpus, of one node; ranks get disjoint sets."""
    import os
    import subprocess
    import sys
    code = (
        "import os, json, sys\n"
        "sys.path.insert(0, %r)\n"
        "from mipsfusion_amd import hostcpu\n"
        "allowed = sorted(os.sched_getaffinity(0))\n"
        "nodes = hostcpu.numa_nodes()\n"
        "r, w = int(sys.argv[1]), int(sys.argv[2])\n"
        "chosen = hostcpu.confine_to_numa_node(2, r, w)\n"
        "print(json.dumps({'allowed': allowed, 'chosen': chosen, 'now': sorted(os.sched_getaffinity(0)), 'nodes': [sorted(n) for n in nodes]}))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import json
    outs = []
    for r in (0, 1):
        o = json.loads(subprocess.run([sys.executable, "-c", code, str(r), "2"], capture_output=True, text=True, check=True).stdout)
        outs.append(o)
        if o["chosen"] is not None:
            assert 1 <= len(o["chosen"]) <= 2 and set(o["chosen"]) <= set(o["allowed"]) and o["now"] == sorted(o["chosen"])
            assert any(set(o["chosen"]) <= set(n) for n in o["nodes"]), "CPUs of more than one NUMA node"
    if all(o["chosen"] is not None for o in outs) and len(outs[0]["allowed"]) >= 4:
        assert not set(outs[0]["chosen"]) & set(outs[1]["chosen"]), "two ranks on the same CPUs"


def test_host_pose_algebra_in_numpy_equals_the_torch_helpers():
    """GraphedSequence converts poses on the host in numpy fp32 scalars (no GIL hand-overs per tiny torch op); same
    operations in the same order as geometry_helper's torch functions -> equal to the last bit, up to the order of the
    4-term quaternion norm (<= 1 ulp)."""
    from mipsfusion_amd.sequence import _matrix_to_quaternion_np, _qt_to_matrix_np
    gen = torch.Generator().manual_seed(0)
    worst = 0.0
    for _ in range(300):
        q = torch.randn(4, generator=gen)
        q = q / q.norm()
        t = torch.randn(3, generator=gen)
        T_ref = gh.qt_to_transform_matrix(q[None], t[None])[0]
        T_np = torch.from_numpy(_qt_to_matrix_np(torch.cat([q, t]).numpy()))
        q_ref = gh.matrix_to_quaternion(T_ref[None, :3, :3])[0]
        q_np = torch.from_numpy(_matrix_to_quaternion_np(T_ref[:3, :3].numpy()))
        worst = max(worst, float((T_np - T_ref).abs().max()), float((q_np - q_ref).abs().max()))
    assert worst <= 2.4e-7, worst


def _hostrng_or_skip():
    from mipsfusion_amd import hostrng
    if not os.path.exists(hostrng._LIB_PATH):
        pytest.skip("libmipsf_hostrng.so not built (make -C mipsfusion_amd/csrc)")
    return hostrng


@pytest.mark.parametrize("n", [16, 17, 31, 32, 100, 4099, 65539, 285200])
def test_host_rng_replica_draws_what_torch_draws(n):
    """mipsfusion_amd.hostrng: `randn_` / `rand_` fill the values torch's default CPU generator would (torch.randn_like of
    sampling_helper.py:30/:62, torch.rand of scene_rep.py:176), bit for bit, for sizes with and without the ragged tail
    of torch's 16-wide normal fill, over interleaved calls; and hand torch back the state it would be in."""
    hostrng = _hostrng_or_skip()
    assert hostrng.available(), "the replica no longer matches this torch build (tools/micro/randn_match.py)"
    for seed in (0, 1, 20240917):
        torch.manual_seed(seed)
        want = [torch.empty(n).normal_(), torch.empty(3, 77).uniform_(), torch.empty(n).normal_(),
                torch.empty(5).normal_(), torch.empty(1000).uniform_()]
        want_state = torch.get_rng_state()
        after = torch.rand(4)
        torch.manual_seed(seed)
        with hostrng.session() as g:
            assert g.native
            got = [g.randn_(torch.empty(n)), g.rand_(torch.empty(3, 77)), g.randn_(torch.empty(n)),
                   g.randn_(torch.empty(5)), g.rand_(torch.empty(1000))]
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        assert torch.equal(torch.get_rng_state(), want_state)
        assert torch.equal(torch.rand(4), after)


def test_host_rng_refuses_what_it_cannot_fill_and_falls_back_without_the_library(monkeypatch):
    hostrng = _hostrng_or_skip()
    with hostrng.session() as g:
        with pytest.raises(ValueError):
            g.rand_(torch.empty(8, 8)[:, ::2])                  # not contiguous
        with pytest.raises(ValueError):
            g.randn_(torch.empty(64, dtype=torch.float64))
        torch.manual_seed(3)
    # the same calls without the replica are torch's own functions (a torch build the self-check rejects, no library)
    monkeypatch.setattr(hostrng, "_ok", False)
    torch.manual_seed(5)
    want = (torch.empty(100).normal_(), torch.empty(10).uniform_())
    torch.manual_seed(5)
    with hostrng.session() as g:
        assert not g.native
        got = (g.randn_(torch.empty(100)), g.rand_(torch.empty(10)))
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


@pytest.mark.parametrize("n,k,p_valid,with_lattice", [(285200, 1024, 0.95, True), (285200, 4096, 0.9, True),
                                                       (70000, 1000, 0.01, True), (8192, 128, 0.5, False),
                                                       (8192, 128, 0.0, False), (8192, 129, 0.5, False)])
def test_host_topk_pass_returns_torch_topk_indices_in_torch_topk_order(n, k, p_valid, with_lattice):
    """mipsfusion_amd.hostrng.topk_valid_pixels == the second half of sample_valid_pixels_random / sample_pixels_mix
    (sampling_helper.py:24-33, :55-68): torch.topk((depth > 0, lattice zeroed) * |draw|, k)[1], index for index --
    also where scores tie (duplicated draws; fewer valid pixels than k, so that zeros fill the tail in the order
    torch's heap leaves them), and where torch leaves its partial_sort path (k * 64 > n: torch's own ops run)."""
    hostrng = _hostrng_or_skip()
    assert hostrng.topk_available()
    g = torch.Generator().manual_seed(n + k)
    for rep in range(3):
        depth = torch.rand(n, generator=g) * (torch.rand(n, generator=g) < p_valid)
        draw = torch.randn(n, generator=g)
        if rep == 1:
            m = draw[3::7].numel()
            draw[0::7][:m] = -draw[3::7]
        blocked = None
        if with_lattice:
            blocked = torch.zeros(n, dtype=torch.uint8)
            blocked[torch.arange(0, n, 293)] = 1
        valid = (depth > 0).to(torch.float32)
        if blocked is not None:
            valid[blocked.bool()] = 0
        want = torch.topk(valid * torch.abs(draw), k)[1]
        got = hostrng.topk_valid_pixels(depth, draw, k, blocked)
        assert got.dtype == torch.int64 and torch.equal(got, want)
    nan_draw = draw.clone()
    nan_draw[11] = float("nan")                                   # torch sorts NaN first: the pass hands such inputs over
    assert torch.equal(hostrng.topk_valid_pixels(depth, nan_draw, k, blocked),
                       torch.topk(valid * torch.abs(nan_draw), k)[1])


def test_host_replica_of_python_random_sample_draws_the_same_indices():
    """mipsfusion_amd.hostrng.py_session().sample_range(n, k) == torch.tensor(random.sample(range(n), k)) -- the draws of
    keyframeSet.py:386-436 / mipsfusion.py:135-138 -- for both of CPython's branches (pool for small n, rejection set
    otherwise), interleaved with python's own calls, leaving the global generator where python would."""
    hostrng = _hostrng_or_skip()
    assert hostrng.py_available()
    for seed in range(4):
        calls = [(5000, 409), (35000, 2048), (600, 500), (20, 5), (1, 1), (5000, 0), (4096, 1365), (87, 22), (285200, 1000)]
        random.seed(seed)
        want = [random.sample(range(n), k) for n, k in calls]
        mid = random.random()
        want2 = [random.sample(range(n), k) for n, k in calls[:3]]
        end = random.getstate()
        random.seed(seed)
        with hostrng.py_session() as r:
            assert r.native
            got = [r.sample_range(n, k) for n, k in calls]
        assert random.random() == mid
        with hostrng.py_session() as r:
            got2 = [r.sample_range(n, k) for n, k in calls[:3]]
        assert [g.tolist() for g in got] == want and [g.tolist() for g in got2] == want2
        assert all(g.dtype == torch.int64 for g in got)
        assert random.getstate() == end
    with hostrng.py_session() as r:
        with pytest.raises(ValueError):
            r.sample_range(10, 11)                                 # python's own error for python's own reasons


def test_host_library_exports_every_symbol_of_its_header():
    """include/mipsf_host.h (libmipsf_hostrng.so, the host generator replicas) against the built library."""
    hostrng = _hostrng_or_skip()
    header = open(os.path.join(ROOT, "include", "mipsf_host.h")).read()
    names = sorted(set(re.findall(r"\b(mipsf_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 5, names
    handle = C.CDLL(hostrng._LIB_PATH)
    for name in names:
        assert hasattr(handle, name), f"{name} declared in mipsf_host.h but not exported"
    assert handle.mipsf_hostrng_abi() == 2


def test_packed_ba_views_share_the_front_of_the_slot_buffers():
    """mipsfusion_amd.sequence.packed: the [iters, n(, S)] view of a local-BA round is the contiguous FRONT of the
    [iters, n_max(, S)] buffer (host and device sides copy it with one DMA transfer each), for every ray count."""
    from mipsfusion_amd import sequence
    rows = torch.arange(5 * 12, dtype=torch.int64).view(5, 12).clone()
    noise = torch.arange(5 * 12 * 3, dtype=torch.float32).view(5, 12, 3).clone()
    for n in (12, 8, 1):
        v, w = sequence.packed(rows, n), sequence.packed(noise, n)
        assert v.shape == (5, n) and w.shape == (5, n, 3) and v.is_contiguous() and w.is_contiguous()
        assert v.data_ptr() == rows.data_ptr() and w.data_ptr() == noise.data_ptr()
        assert torch.equal(v.reshape(-1), rows.reshape(-1)[:5 * n])
    s = sequence.FrameSamples(n_track=4, it_track=2, n_ba_max=12, it_ba=5, S=3, pinned=False)
    s.n_ba = 7
    r, o, z = s.ba_packed()
    assert r.shape == (5, 7) and o.shape == (5, 7) and z.shape == (5, 7, 3)
    r.fill_(3)
    assert int(s.ba_rows.reshape(-1)[:35].sum()) == 105


def test_backward_from_one_is_loss_backward():
    """helper_functions.utils.backward_from_one: the same gradients as loss.backward() (the root gradient comes from a
    cached tensor of ones instead of a fill per call)."""
    from mipsfusion_amd.helper_functions.utils import backward_from_one
    w1 = torch.randn(7, requires_grad=True)
    w2 = w1.detach().clone().requires_grad_(True)
    x = torch.randn(7)
    ((w1 * x).sin().sum() * 3).backward()
    for _ in range(2):                                   # second call: the cached root gradient, accumulated grads
        backward_from_one((w2 * x).sin().sum() * 3)
    assert torch.equal(w2.grad, 2 * w1.grad)


def test_submap_timeline_follows_the_switch_schedule():
    """mipsfusion_amd.sequence.submap_timeline: which sub-map is active, and with which keyframe slots, when a frame's local
    BA runs (the frame's own keyframe / switch comes after its BA, mipsfusion.py:681-712)."""
    from mipsfusion_amd.sequence import submap_timeline
    tl = submap_timeline(50, 5, {20: ("new",), 35: ("back", 0)})
    assert tl[1] == (0, [0]) and tl[5] == (0, [0]) and tl[6] == (0, [0, 1])
    assert tl[20] == (0, [0, 1, 2, 3])                 # frame 20 is still bundle-adjusted in sub-map 0 ...
    assert tl[21] == (1, [4]) and tl[26] == (1, [4, 5])    # ... then opens sub-map 1 as its first keyframe
    assert tl[35] == (1, [4, 5, 6])
    assert tl[36] == (0, [0, 1, 2, 3, 7])              # back in sub-map 0, the overlapping keyframe joined it
    assert tl[49] == (0, [0, 1, 2, 3, 7, 8, 9])
    with pytest.raises(ValueError):
        submap_timeline(30, 5, {10: ("back", 3)})


def test_no_kernel_holds_a_packed_fp32_operation_that_crosses_halves():
    """on gfx950 a packed fp32 operation whose LOW result reads the HIGH half of a source (a 1 in op_sel) lost that
    operand in lanes 48..63 while the decoder's kernels ran on the same CUs.  Kernels in which hipcc would emit one carry
    MIPSF_SINGLE_FP32 (common.h); this compiles every unit to ISA (cross-compiles without a GPU) and checks that none is left --
    and that the audit does see them when the marking is switched off."""
    import shutil
    import sys
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import audit_packed
    crossed = [(u, k, c) for u, k, _n, c, _l in audit_packed.audit_all() if c]
    assert not crossed, crossed
    assert any(c for _u, _k, _n, c, _l in audit_packed.audit_all(("-DMIPSF_KEEP_PACKED_FP32",), units=("ro",)))     # the audit is not blind


def _load_bench_module():
    """bench.py as a module, without its NUMA confinement (import-time side effect of the script)"""
    import importlib.util
    os.environ["MIPSF_NO_CONFINE"] = "1"
    spec = importlib.util.spec_from_file_location("mipsf_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("record", ["r04_f_bench.json", "r03_k_bench.json"])
def test_bench_line_is_compact_and_carries_the_contract(record):
    """The driver keeps an 8 000-character stdout tail: round 4's 23 KB line could not be parsed.  The line bench.py prints
    is built by `compact_line` from the full result; here from RECORDED full results (the largest one on file and an
    older layout), plus a worst case with every optional block inflated."""
    import json
    bench = _load_bench_module()
    out = json.load(open(os.path.join(ROOT, "profiles", record)))
    text = bench.compact_line(out)
    assert len(text) < bench.LINE_LIMIT <= 6000 and "\n" not in text
    line = json.loads(text)
    for k in bench.REQUIRED_LINE_KEYS:
        assert k in line, k
    assert line["value"] == out["value"] and line["ms_per_step"] == out["ms_per_step"]
    assert len(line["dtype"]) <= max(120, len(out["dtype"])) and "workload" in line["config"]
    assert "model" not in line["config"]
    roof = line["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_ms", "traffic"):
        assert k in roof, k
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    if "frame" in out:
        assert line["frame"]["tracking_plus_mapping_ms_per_frame"] == out["frame"]["tracking_plus_mapping_ms_per_frame"]
    if (out.get("variants") or {}).get("unchanged_caller"):
        assert line["unchanged_caller_ms_per_step"] == out["variants"]["unchanged_caller"]["ms_per_step"]
    # the board's state while the region was timed (round 6): carried when the run sampled it, null on the line otherwise
    assert "board" in line and line["board"] is None
    withb = dict(out)
    withb["board"] = {"sclk_mhz_median": 1843.0, "sclk_mhz_min": 1712.0, "power_w_median": 1398.2, "samples": 57, "source": "x" * 400}
    withb["step_ms_stats"] = {"steps": 200, "min": 0.63, "median": 0.6416, "p95": 0.66, "max": 0.7}
    withb["value_at_median_step"] = 4.0e8
    text = bench.compact_line(withb)
    assert len(text) < bench.LINE_LIMIT
    lb = json.loads(text)
    assert lb["board"] == {"sclk_mhz_median": 1843.0, "sclk_mhz_min": 1712.0, "power_w_median": 1398.2, "samples": 57}
    assert lb["value_at_median_step"] == 4.0e8
    # worst case: hundreds of kernels and variants -> optional blocks are dropped, the contract stays
    fat = dict(withb)
    fat["kernels"] = {f"kernel_with_a_long_name_{i}": {"avg_ms": 0.123456, "frac": 0.5} for i in range(400)}
    fat["variants"] = {f"variant_{i}": {"ms_per_step": 1.0} for i in range(300)}
    text = bench.compact_line(fat)
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    for k in bench.REQUIRED_LINE_KEYS:
        assert k in line, k
    assert line["board"]["power_w_median"] == 1398.2          # (the board block is not one of the droppable ones)


def test_board_sampler_without_a_gpu_reports_nothing_and_does_not_fail():
    bench = _load_bench_module()
    with bench.BoardSampler(0, period=0.001) as b:
        time.sleep(0.01)
    sm = b.summary()
    assert set(sm) >= {"sclk_mhz_median", "power_w_median", "samples", "source"}
    if b.card is None:
        assert sm["sclk_mhz_median"] is None and sm["samples"] == 0


def test_bench_launcher_process_keeps_its_affinity_mask():
    """`python bench.py --gpus N` only starts the ranks: confining THAT process would hand every child a one-node mask
    (each rank then gets 32/N CPUs of the same node instead of a node of its own)."""
    bench = _load_bench_module()
    env = os.environ.pop("WORLD_SIZE", None)
    try:
        assert bench._only_launches_ranks(["--gpus", "8", "--steps", "5"])
        assert bench._only_launches_ranks(["--steps", "5", "--gpus=2"])
        assert not bench._only_launches_ranks(["--gpus", "1"])
        assert not bench._only_launches_ranks(["--steps", "5"])
        os.environ["WORLD_SIZE"] = "8"
        assert not bench._only_launches_ranks(["--gpus", "8"])
    finally:
        os.environ.pop("WORLD_SIZE", None)
        if env is not None:
            os.environ["WORLD_SIZE"] = env


def test_host_rng_sessions_exclude_each_other_and_ray_dp_refuses_a_model_without_the_hook():
    """(1) hostrng.session takes the default CPU generator's state out of torch and puts it back: two threads doing that at
    once would overwrite each other's draws (the frame loop has a sample-producer thread).  One process-wide lock:
    draws made by two threads through sessions are, as a set, exactly the draws one thread makes in a row.  (2)
    RayDataParallelStep sums gradients over the ranks, which is only right for a model that finishes the whole batch's losses
    through `ray_share_reduce`: a model without the hook is refused, and close() takes the hook off again."""
    import threading
    import types
    from mipsfusion_amd import hostrng
    from mipsfusion_amd.ray_dp import RayDataParallelStep
    n, rounds = 4096, 40
    torch.manual_seed(123)
    want = torch.rand(2 * rounds, n)
    torch.manual_seed(123)
    got = [[], []]

    def worker(j):
        for _ in range(rounds):
            buf = torch.empty(n)
            with hostrng.session() as sess:
                sess.rand_(buf)
            got[j].append(buf)
    ts = [threading.Thread(target=worker, args=(j,)) for j in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    rows = {tuple(r[:8].tolist()) for r in want}
    seen = [tuple(b[:8].tolist()) for g in got for b in g]
    assert len(set(seen)) == 2 * rounds and set(seen) == rows, "two sessions interleaved: draws were lost or made twice"
    with pytest.raises(TypeError, match="ray_share_reduce"):
        RayDataParallelStep(types.SimpleNamespace(), None, None)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.embed_fn = types.SimpleNamespace(params=torch.nn.Parameter(torch.zeros(64)))
            self.decoder = torch.nn.Linear(2, 2)
            self.ray_share_reduce = None
    m = Tiny()
    mk = lambda ps: torch.optim.Adam([ps] if isinstance(ps, torch.Tensor) else ps)      # noqa: E731
    with RayDataParallelStep(m, mk, mk) as rdp:
        assert m.ray_share_reduce == rdp.reduce_share
    assert m.ray_share_reduce is None


def test_decoder_widths_outside_the_reference_architecture_are_refused_by_name():
    """model/decoder.py:7-16 takes any widths; every caller of the reference builds the defaults (scene_rep.py:45).  The HIP
    kernels exist for that architecture only: another width must fail at construction with a message that names
    the argument, not later inside a kernel."""
    from mipsfusion_amd.model.decoder import MLP_reg
    MLP_reg({}, input_ch=32, input_ch_pos=48)                                   # the reference's call
    with pytest.raises(ValueError, match=r"n_hidden = 256 \(kernels: 128\)"):
        MLP_reg({}, input_ch=32, input_ch_pos=48, n_hidden=256)
    with pytest.raises(ValueError, match=r"n_class = 7 .*n_class|n_class = 7"):
        MLP_reg({}, input_ch=32, input_ch_pos=48, n_class=7)
    with pytest.raises(ValueError, match=r"input_ch = 16"):
        MLP_reg({}, input_ch=16, input_ch_pos=48)
