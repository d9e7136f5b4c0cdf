"""GPU (-m gpu): BASELINE config 3 in miniature -- the online loop with two sub-maps, a switch to a new sub-map and a
switch back (tests/seq_harness.py) run over the PRODUCT's modules, against the run of the same loop over the
reference's own classes recorded in tests/golden/sequence.npz (generator: make_golden.py::gen_sequence).

Checked: the whole index stream (pixel / keyframe-ray / lattice indices, in call order) bit for bit -- i.e. the
product consumes the python-`random`, torch-CPU and numpy generators exactly as the reference does, including the
jitter draw inside ``forward`` --, the 51-entry loss trace, every frame's local pose, both sub-maps' weights after the
hand-offs (deepcopy / load_state_dict / recover_initial_param)."""
import copy
import types

import numpy as np
import pytest
import torch

from mipsfusion_amd.helper_functions import geometry_helper as gh
from mipsfusion_amd.helper_functions import sampling_helper as sh
from mipsfusion_amd.keyframe_rays import DeviceRayDB
from mipsfusion_amd.model import JointEncoding
from mipsfusion_amd.optim import FusedAdam

from . import seq_harness
from .conftest import load_golden

pytestmark = pytest.mark.gpu


class _KfSet:
    """KeyframeSet's ray side (model/keyframeSet.py:25, 76-79, 170-175) over the device-resident database."""

    def __init__(self, cfg, H, W, num_kf, dev):
        s = cfg["sampling"]
        self.rows, self.cols = sh.sample_pixels_uniformly(H, W, s["kf_n_rays_h"], s["kf_n_rays_w"])
        self.db = DeviceRayDB(num_kf, s["kf_n_rays_h"] * s["kf_n_rays_w"], dev)
        self.n = 0
        self.sample_rays_in_submap = self.db.sample_rays_in_submap
        self.sample_rays_in_given_kf = self.db.sample_rays_in_given_kf

    def add_keyframe(self, frame):
        rays = torch.cat([frame["direction"], frame["rgb"], frame["depth"][..., None]], -1)
        self.db.store(self.n, rays[self.rows, self.cols])
        self.n += 1


def product_backend(dev, fused_adam=True, in_place=False, precision="f16x3"):
    from mipsfusion_amd.RandomOptimizer import RandomOptimizer

    def _ro(ro):
        ro.decoder_precision = precision    # the 51-iteration trace runs the RandomOptimizer in the model's arithmetic
        return ro

    def make_model(cfg, bb, nf):
        m = JointEncoding(cfg, bb, nf).to(dev)
        m.accumulate_param_grads_in_place = in_place
        m.decoder_precision = precision
        return m

    return types.SimpleNamespace(
        device=dev, make_model=make_model, deepcopy=copy.deepcopy,
        Adam=FusedAdam if fused_adam else torch.optim.Adam, sh=sh,
        qt_to_transform_matrix=gh.qt_to_transform_matrix, matrix_to_quaternion=gh.matrix_to_quaternion,
        make_kfset=lambda cfg, H, W, n: _KfSet(cfg, H, W, n, dev),
        make_ro=lambda cfg, slam: _ro(RandomOptimizer(cfg, slam)),
        ro_optimize=lambda ro, model, depth, init, last, n: ro.optimize(model, depth, init, last, n_iter=n))


def rel_max(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("fused_adam,in_place,precision", [(True, False, "f32"), (True, True, "f32"), (False, False, "f32"),
                                                          (True, True, "f16x3")])
def test_two_submap_sequence_matches_reference_run(fused_adam, in_place, precision):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests must run on the MI355X box")
    dev = torch.device("cuda:0")
    g = load_golden("sequence.npz")
    out = seq_harness.run_sequence(product_backend(dev, fused_adam, in_place, precision))
    # ---- index stream: same tags in the same order, every index equal (ray-database rows included)
    assert list(out["tags"]) == [str(t) for t in g["tags"]]
    assert [t.numel() for t in out["idx"]] == list(g["idx_len"])
    got = torch.cat(out["idx"]).numpy()
    if not np.array_equal(got, g["idx_flat"]):
        off = np.concatenate([[0], np.cumsum(g["idx_len"])])
        first = next(k for k in range(len(out["idx"])) if not np.array_equal(out["idx"][k].numpy(),
                                                                             g["idx_flat"][off[k]:off[k + 1]]))
        pytest.fail(f"index stream differs from the reference's, first at record {first} ({out['tags'][first]}); "
                    f"losses so far {out['losses'][:8]} vs {g['losses'][:8]}")
    # ---- loss trace (51 model iterations across init / tracking / BA / switch phases)
    lo, lr = out["losses"], g["losses"]
    assert lo.shape == lr.shape
    worst = float(np.max(np.abs(lo - lr) / np.abs(lr)))
    print(f"sequence: worst relative loss deviation {worst:.2e} over {lo.size} iterations "
          f"(first 10: {np.max(np.abs(lo[:10] - lr[:10]) / np.abs(lr[:10])):.2e})")
    # The loop is chaotic (51 Adam steps, best-of-iterations pose selection, a particle swarm's weighted mean): round-off
    # differences grow.  Measured with tools/dbg_seq_chaos.py (MI355X, distance of the final poses / worst loss
    # deviation from the reference's run):
    #   fp32 kernels                                            0.02 - 0.05 mm   6e-4 - 2e-3   (atomics: run to run)
    #   fp32 kernels, initial decoder weights moved by one ulp  0.05, 0.05, 0.5 mm   1e-3 - 4e-3   (three seeds)
    #   fp32 kernels, weights moved by 1e-6 relative            0.08, 0.18, 3.8 mm   9e-4 - 1.5e-2
    #   f16x3 decoder, any fp32-class weight-gradient kernel    2.3 - 2.5 mm     3e-3 - 4e-3
    # i.e. the f16x3 default -- as accurate against fp64 truth as the fp32 kernels (tests/test_gpu_parity.py), but with
    # rounding errors UNCORRELATED with torch's fp32, ~1e-6 relative per step -- lands where a 1e-6 perturbation of the
    # fp32 run itself lands.  fp32 arithmetic is held to 2 mm / 5e-3, f16x3 to 5 mm / 2e-2; before the chaotic growth
    # (first 12 iterations) both to 5e-4.
    tol_l, tol_p = (5e-3, 2e-3) if precision == "f32" else (2e-2, 5e-3)
    np.testing.assert_allclose(lo[:12], lr[:12], rtol=5e-4)       # before chaotic growth: tight in both modes
    np.testing.assert_allclose(lo, lr, rtol=tol_l)
    # ---- poses: local pose of every frame (RandomOptimizer + pose Adam + BA + switch conversions)
    dt = np.abs(out["est"][:, :3, 3] - g["est"][:, :3, 3]).max()
    dr = np.abs(out["est"][:, :3, :3] - g["est"][:, :3, :3]).max()
    print(f"sequence ({precision}): max translation deviation {dt:.2e} m, rotation-matrix deviation {dr:.2e}")
    assert dt < tol_p, "translations (m)"
    assert dr < tol_p, "rotations"
    # ---- weights of both sub-maps after the hand-offs
    for sm in (0, 1):
        for k in ("decoder.pts_linear.0.weight", "decoder.sdf_linear.2.weight", "decoder.rgb_linear.0.bias"):
            e = rel_max(out["models"][sm][k].numpy(), g[f"m{sm}.{k}"])
            assert e < 5e-2, f"sub-map {sm} {k}: {e:.2e}"
        a, b = out["models"][sm]["embed_fn.params"].numpy(), g[f"m{sm}.embed_fn.params"]
        # dense Adam moves every touched entry by ~lr per step: compare where the reference moved the table
        l2 = np.linalg.norm(a - b) / np.linalg.norm(b)
        assert l2 < 5e-2, f"sub-map {sm} grid: relative L2 {l2:.2e}"
    # the InactiveMap-side copy is the active model as of the last BA round (mipsfusion.py:683)
    for k in ("decoder.pts_linear.0.weight", "decoder.sdf_linear.2.weight"):
        assert rel_max(out["active_copy"][k].numpy(), g[f"copy.{k}"]) < 5e-2
