"""How mipsfusion_amd/csrc/host/hostrng.c's Box-Muller arithmetic was pinned to this torch build's `normal_()`.

torch fills >= 16 float32 normals by (1) a uniform fill, (2) per 16 values: u1 = 1 - x[0:8], u2 = x[8:16],
r = sqrt(-2 log u1), theta = 2 pi u2, x[0:8] = r cos theta, x[8:16] = r sin theta, with the cephes-style 8-lane
log / sincos of its AVX2 build.  Which multiply-adds of those polynomials the build CONTRACTED into FMAs is not visible
from outside; this script emulates the arithmetic in numpy (an FMA = the product in float64, rounded once -- exact for
float32 operands) with a switch per candidate contraction, and searches the switch settings for the one that
reproduces 131072 values of `torch.randn` with zero mismatches.  Result (torch 2.10 CPU, AVX2 dispatch):

    log:    polynomial steps fused; (y*x)*z + e*q1 as ONE fma with e*q1 rounded first; y - z/2 and x + y plain;
            + e*q2 unfused
    sincos: range reduction x + y*DPk unfused; cos polynomial fused, (y*z)*z - z/2 as one fmsub; sin polynomial fused,
            y*z rounded then fma(., x, x)

Run it again after a torch upgrade if hostrng's start-up self-check starts reporting a mismatch (the producer then
falls back to torch's own draws; nothing breaks, the generator stage is merely 4x slower)."""
import numpy as np, torch, math, itertools
f32 = np.float32
def fma(a, b, c, use):
    c = np.float32(c) if np.isscalar(c) else c
    if use: return (a.astype(np.float64) * b.astype(np.float64) + np.asarray(c, dtype=np.float64)).astype(np.float32)
    return (a * b).astype(np.float32) + c
def fnma(a, b, c, use):   # c - a*b
    if use: return (c.astype(np.float64) - a.astype(np.float64) * b.astype(np.float64)).astype(np.float32)
    return c - (a * b).astype(np.float32)
def log256(x, F):
    x = np.maximum(x, f32(1.17549435e-38))
    xi = x.view(np.int32); imm0 = (xi >> 23)
    x = ((xi & np.int32(~0x7f800000)) | f32(0.5).view(np.int32)).view(np.float32)
    e = (imm0 - 0x7f).astype(np.float32) + f32(1)
    mask = x < f32(0.707106781186547524)
    tmp = np.where(mask, x, f32(0)); x = x - f32(1); e = e - np.where(mask, f32(1), f32(0)); x = x + tmp
    z = x * x
    ps = [7.0376836292E-2, -1.1514610310E-1, 1.1676998740E-1, -1.2420140846E-1, 1.4249322787E-1, -1.6668057665E-1, 2.0000714765E-1, -2.4999993993E-1, 3.3333331174E-1]
    y = np.full_like(x, f32(ps[0]))
    for k, p in enumerate(ps[1:]): y = fma(y, x, f32(p), F["Lp"])
    if F.get("Lalt"):
        yx = y * x
        y = fma(yx, z, (e * f32(-2.12194440e-4)).astype(np.float32), True)     # (y*x)*z + e*q1 as one fma
    else:
        y = y * x; y = y * z
        y = fma(e, np.full_like(e, f32(-2.12194440e-4)), y, F["Lq1"])
    y = fnma(z, np.full_like(z, f32(0.5)), y, F["Lh"])
    x = x + y
    x = fma(e, np.full_like(e, f32(0.693359375)), x, F["Lq2"])
    return x
def sincos256(x, F):
    xi = x.view(np.int32); sign_sin = xi & np.int32(-2**31); x = np.abs(x)
    y = x * f32(1.27323954473516)
    imm2 = y.astype(np.int32); imm2 = (imm2 + 1) & ~1; y = imm2.astype(np.float32); imm4 = imm2
    swap = (imm2 & 4) << 29; poly_mask = (imm2 & 2) == 0
    for k, dp in enumerate((-0.78515625, -2.4187564849853515625e-4, -3.77489497744594108e-8)):
        x = fma(y, np.full_like(y, f32(dp)), x, F["D%d" % k])
    sign_cos = ((~(imm4 - 2)) & 4) << 29; sign_sin = sign_sin ^ swap
    z = x * x
    y = np.full_like(x, f32(2.443315711809948E-005))
    y = fma(y, z, f32(-1.388731625493765E-003), F["C"]); y = fma(y, z, f32(4.166664568298827E-002), F["C"])
    if F.get("Calt"):
        yz = y * z
        y = fma(yz, z, -(z * f32(0.5)), True)          # (y*z)*z - z/2 as one fmsub
    else:
        y = y * z; y = y * z
        y = fnma(z, np.full_like(z, f32(0.5)), y, F["Ch"])
    y = y + f32(1)
    y2 = np.full_like(x, f32(-1.9515295891E-4))
    y2 = fma(y2, z, f32(8.3321608736E-3), F["S"]); y2 = fma(y2, z, f32(-1.6666654611E-1), F["S"])
    y2 = y2 * z
    y2 = fma(y2, x, x, F["Sx"])
    ysin2 = np.where(poly_mask, y2, f32(0)); ysin1 = np.where(poly_mask, f32(0), y)
    y2 = y2 - ysin2; y = y - ysin1
    s = ysin1 + ysin2; c = y + y2
    return (s.view(np.int32) ^ sign_sin.astype(np.int32)).view(np.float32), (c.view(np.int32) ^ sign_cos.astype(np.int32)).view(np.float32)
torch.manual_seed(7); n = 16 * 8192
ref = torch.randn(n).numpy()
torch.manual_seed(7); u = torch.rand(n).numpy().reshape(-1, 16)
u1 = f32(1) - u[:, :8]; u2 = u[:, 8:]
def run(F):
    lg = log256(u1.copy(), F)
    radius = np.sqrt(fma(np.full_like(lg, f32(-2)), lg, f32(0), False))
    theta = f32(2.0 * math.pi) * u2
    s, c = sincos256(theta.copy(), F)
    o1 = radius * c; o2 = radius * s
    out = np.concatenate([o1, o2], 1).reshape(-1)
    return int((out != ref).sum())
keys = ["Lp", "Lq1", "Lh", "Lq2", "D0", "D1", "D2", "C", "Ch", "S", "Sx", "Calt", "Lalt"]
best = None
base = dict(Lp=True, S=True, Sx=True, C=True)
free = ["Lq1", "Lh", "Lq2", "D0", "D1", "D2", "Ch", "Calt", "Lalt"]
import itertools as _it
def gen():
    for bits in _it.product([False, True], repeat=len(free)):
        F = dict(base); F.update(dict(zip(free, bits))); yield tuple(F.get(k, False) for k in keys)
for bits in gen():
    F = dict(zip(keys, bits)); m = run(F)
    if best is None or m < best[0]: best = (m, F); print(m, F)
    if m == 0: break
