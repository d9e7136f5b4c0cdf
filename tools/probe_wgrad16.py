"""Streaming weight-gradient kernel (csrc/wgrad16.hip) against the fp32 LDS kernel: errors per tensor and timings."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mipsfusion_amd import ops, _lib
if os.environ.get("MIPSF_LIB_VARIANT"):      # private experiment build (tools/micro)
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libmipsf_%s.so" % os.environ["MIPSF_LIB_VARIANT"])
    print("variant", os.environ["MIPSF_LIB_VARIANT"])
from mipsfusion_amd.model import MLP_reg
dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
ws = dec.ordered_parameters()
names = ["w_pts0", "b_pts0", "w_pts2", "b_pts2", "w_rgb0", "b_rgb0", "w_sdf0", "b_sdf0", "w_sdf2", "b_sdf2"]
packed16 = ops.decoder_pack16(ws)
L = _lib.FEAT_LEVEL_MAJOR
Ms = [int(v) for v in os.environ.get("W16_M", "37,5000,90112,262144").split(",")]
for M in Ms:
    x = torch.rand(M, 3, device=dev); feat = torch.randn(16, M, 2, device=dev) * 0.3
    dout = torch.randn(M, 10, device=dev) * float(os.environ.get('W16_DOUT', '1e-3'))
    if os.environ.get('W16_TAIL'):      # per-sample magnitudes over 2^14, as a mean over mixed loss terms produces
        dout = dout * torch.exp(torch.empty(M, 1, device=dev).uniform_(-10.0, 0.0))
    out, saved = ops.decoder_fwd(None, feat, L, x, None, M, save=True, precision="f16x3", packed16=packed16)
    res = {}
    for wp in ("f32", "stream_f16x3", "stream_bf16x6", "stream_bf16x3"):
        g = [torch.zeros_like(w) for w in ws]
        ops.decoder_bwd(None, feat, L, x, None, out, dout, saved, g, M, precision="f16x3", packed16=packed16, wgrad_precision=wp)
        torch.cuda.synchronize()
        res[wp] = g
    print(f"M={M}")
    for wp in ("stream_f16x3", "stream_bf16x6", "stream_bf16x3"):
        errs = []
        for n, a, b in zip(names, res[wp], res["f32"]):
            errs.append(f"{n}:{((a - b).norm() / b.norm().clamp_min(1e-30)).item():.1e}/{((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item():.1e}")
        print(f"  {wp:14s} " + " ".join(errs))
    if M >= 90112:
        for wp in ("f32", "stream_f16x3", "stream_bf16x6", "stream_bf16x3"):
            g = [torch.zeros_like(w) for w in ws]
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for it in range(3):
                if it == 1: ev[0].record()
                for _ in range(10):
                    ops.decoder_bwd(None, feat, L, x, None, out, dout, saved, g, M, precision="f16x3", packed16=packed16, wgrad_precision=wp)
            ev[1].record(); torch.cuda.synchronize()
            print(f"  {wp}: whole backward {ev[0].elapsed_time(ev[1]) / 20 * 1000:.1f} us")
