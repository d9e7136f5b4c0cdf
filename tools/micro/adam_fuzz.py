"""FusedAdam against torch.optim.Adam on adversarial magnitudes (parameters over 2^40, gradients over 2^140 with zeros,
subnormals and squares that overflow): non-finite patterns and the error against max(|parameter|, |update|)."""
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from mipsfusion_amd.optim import FusedAdam
dev = torch.device('cuda:0')
torch.manual_seed(0)
n = 200003
for trial in range(4):
    p0 = torch.randn(n) * torch.exp2(torch.randint(-20, 20, (n,)).float())
    a = torch.nn.Parameter(p0.clone().to(dev)); b = torch.nn.Parameter(p0.clone())
    kw = dict(lr=0.01, betas=(0.9, 0.99), eps=[1e-15, 1e-8, 1e-15, 1e-8][trial], weight_decay=[0, 1e-6, 0, 1e-2][trial])
    oa = FusedAdam([a], capturable=bool(trial & 1), **kw); ob = torch.optim.Adam([b], **kw)
    for step in range(5):
        g = torch.randn(n) * torch.exp2(torch.randint(-100, 40, (n,)).float())
        g[torch.rand(n) < 0.3] = 0.0
        g[::1001] = 1e-42          # subnormal gradients
        g[5::1001] = 3e38          # g*g overflows
        a.grad, b.grad = g.clone().to(dev), g.clone()
        prev = b.detach().numpy().copy()
        oa.step(); ob.step()
        x, y = a.detach().cpu().numpy(), b.detach().numpy()
        fin = np.isfinite(y)
        same_nonfinite = np.array_equal(np.isfinite(x), fin)
        p_before = prev if step else p0.numpy()
        scale = np.maximum(np.abs(p_before[fin]), np.abs(y[fin] - p_before[fin])) + 1e-38
        rel = np.abs(x[fin] - y[fin]) / scale          # error against the larger of the parameter and its update
        # entries whose update is exactly zero on one side
        print(f"trial {trial} step {step}: nonfinite pattern equal {same_nonfinite}; max rel err {rel.max():.3e}; entries off by > 1e-5: {int((rel > 1e-5).sum())}")
