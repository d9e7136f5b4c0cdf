"""How far does the 51-iteration sequence (tests/seq_harness.py) travel under fp32-level perturbations?  Reference point:
the golden run.  Variants: fp32 kernels; fp32 kernels with the initial decoder weights moved by one ulp (relative 6e-8,
three seeds); the f16x3 decoder with each weight-gradient kernel."""
import sys, os, copy, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import seq_harness
from tests.test_gpu_sequence import product_backend
from tests.conftest import load_golden
dev = torch.device("cuda:0")
g = load_golden("sequence.npz")

def run(precision="f32", wgrad="auto", ulp_seed=None, rel=6e-8):
    b = product_backend(dev, True, True, precision)
    mk = b.make_model
    def make_model(cfg, bb, nf):
        m = mk(cfg, bb, nf)
        m.wgrad_precision = wgrad
        if ulp_seed is not None:
            gen = torch.Generator(device="cpu").manual_seed(ulp_seed)
            with torch.no_grad():
                for p in m.decoder.parameters():
                    p.mul_(1.0 + rel * torch.randn(p.shape, generator=gen).to(p.device))
        return m
    b.make_model = make_model
    out = seq_harness.run_sequence(b)
    dt = np.abs(out["est"][:, :3, 3] - g["est"][:, :3, 3]).max()
    lo, lr = out["losses"], g["losses"]
    return dt, float(np.max(np.abs(lo - lr) / np.abs(lr))), out

base = run("f32")
print(f"fp32 kernels:                       {base[0]*1e3:.3f} mm, loss {base[1]:.1e}")
for seed in (1, 2, 3):
    r = run("f32", ulp_seed=seed)
    d = np.abs(r[2]["est"][:, :3, 3] - base[2]["est"][:, :3, 3]).max()
    print(f"fp32 kernels, weights moved 1 ulp ({seed}): {r[0]*1e3:.3f} mm vs golden, {d*1e3:.3f} mm vs unperturbed, loss {r[1]:.1e}")
for seed in (1, 2, 3):
    r = run("f32", ulp_seed=seed, rel=1e-6)
    print(f"fp32 kernels, weights moved 1e-6 relative ({seed}): {r[0]*1e3:.3f} mm vs golden, loss {r[1]:.1e}")
for wg in ("f32", "stream_f16x3", "stream_bf16x6", "stream_bf16x3"):
    r = run("f16x3", wg)
    print(f"f16x3 decoder, wgrad {wg:14s}: {r[0]*1e3:.3f} mm, loss {r[1]:.1e}")
