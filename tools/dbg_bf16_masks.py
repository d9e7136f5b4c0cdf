import sys
import torch
sys.path.insert(0, "/root/repo")
from mipsfusion_amd import ops, _lib
from mipsfusion_amd.model import MLP_reg
dev = torch.device("cuda:0")
for M, fs in ((20000, 1e-4), (20000, 0.2), (1000, 1e-4), (262144, 1e-4)):
    torch.manual_seed(0)
    dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
    with torch.no_grad():
        dec.sdf_linear[2].weight.mul_(3.0)
    ws = dec.ordered_parameters()
    packed, p16, pbf = ops.decoder_pack(ws), ops.decoder_pack16(ws), ops.decoder_pack16(ws, precision="bf16x6")
    x = torch.rand(M, 3, device=dev)
    feat = ((torch.rand(M, 32, device=dev) * 2 - 1) * fs).contiguous()
    dout = torch.randn(M, 10, device=dev) * torch.exp(torch.empty(M, 1, device=dev).uniform_(-16.0, -6.0))
    L = _lib.FEAT_AOS
    o32, s32 = ops.decoder_fwd(packed, feat, L, x, None, M, save=True)
    o16, s16 = ops.decoder_fwd(None, feat, L, x, None, M, save=True, precision="f16x3", packed16=p16)
    ob, sb = ops.decoder_fwd(None, feat, L, x, None, M, save=True, precision="bf16x6", packed16=pbf)
    n_act = ((M + 127) // 128) * 4 * 192 * 64
    nt = (M + 31) // 32
    m32 = s32[n_act:n_act + nt * 256].view(torch.int32)
    m16 = s16[n_act:n_act + nt * 256].view(torch.int32)
    mb = sb[n_act:n_act + nt * 256].view(torch.int32)
    def bits(a, b):
        return int(sum(bin(v & 0xffffffff).count("1") for v in (a ^ b).cpu().tolist()))
    print(f"M {M} feat {fs}: mask bits differing f16x3 vs f32: {bits(m16, m32)}, bf16x6 vs f32: {bits(mb, m32)} of {nt * 256 * 32}")
    a32 = s32[:n_act].view(-1, 192 * 64)[:nt]; ab = sb[:n_act].view(-1, 192 * 64)[:nt]
    print("   act max diff", float((a32 - ab).abs().max()), "out max diff", float((o32 - ob).abs().max()))
    res = {}
    for name, (o, s, kw, pk) in {"f32": (o32, s32, {}, packed), "bf rec32": (o32, s32, dict(precision="bf16x6", packed16=pbf), None),
                                 "bf recbf": (ob, sb, dict(precision="bf16x6", packed16=pbf), None),
                                 "f16 rec16": (o16, s16, dict(precision="f16x3", packed16=p16), None)}.items():
        g = [torch.zeros_like(w) for w in ws]
        df, dx, _ = ops.decoder_bwd(pk, feat, L, x, None, o, dout, s, g, M, **kw)
        res[name] = (df, dx, g)
    for name in res:
        if name == "f32": continue
        r = lambda a, b: float((a - b).norm() / b.norm())
        print(f"   {name}: dfeat {r(res[name][0], res['f32'][0]):.2e} dx {r(res[name][1], res['f32'][1]):.2e} w_pts0 {r(res[name][2][0], res['f32'][2][0]):.2e}")
