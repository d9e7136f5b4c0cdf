"""Run each decoder kernel repeatedly on identical inputs; every output must be bit-identical run to run."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mipsfusion_amd import ops, _lib
from mipsfusion_amd.model import MLP_reg
dev = torch.device("cuda:0")
for prec in ("f32", "f16x3"):
    for M in (4096, 70000):
        torch.manual_seed(M)
        dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
        ws = dec.ordered_parameters()
        x = torch.rand(M, 3, device=dev)
        feat = (torch.randn(16, M, 2, device=dev) * 0.3).contiguous()
        L = _lib.FEAT_LEVEL_MAJOR
        dout = torch.randn(M, 10, device=dev)
        ref, bad, where = None, {}, {}
        n_tiles = (M + 31) // 32
        for it in range(40):
            packed = ops.decoder_pack(ws) if prec == "f32" else None
            packed16 = ops.decoder_pack16(ws) if prec != "f32" else None
            kw = dict(precision=prec, packed16=packed16)
            out, saved = ops.decoder_fwd(packed, feat, L, x, None, M, save=True, **kw)
            g = [torch.zeros_like(w) for w in ws]
            dact_probe = None
            dfeat, dx, _ = ops.decoder_bwd(packed, feat, L, x, None, out, dout, saved, g, M, **kw)
            n_act = ((M + 127) // 128) * 4 * 192 * 64
            cur = {"out": out, "saved_act": saved[:n_act].view(-1, 192 * 64)[:n_tiles],
                   "saved_masks": saved[n_act:].view(torch.int32).view(-1, 256)[:n_tiles],
                   "dfeat": dfeat, "dx": dx, "g_w_pts2": g[2], "g_w_sdf0": g[6], "g_b_sdf2": g[9], "g_w_rgb0": g[4]}
            if ref is None:
                ref = {k: v.clone() for k, v in cur.items()}
            else:
                for k, v in cur.items():
                    if not torch.equal(v, ref[k]):
                        bad[k] = bad.get(k, 0) + 1
                        if k not in where:
                            d = (v != ref[k]).nonzero()
                            where[k] = (len(d), d[:3].tolist(), float((v.float() - ref[k].float()).abs().max()))
        print(prec, M, "non-identical runs out of 39:", bad if bad else "none", where)
