import sys, torch
sys.path.insert(0, "/root/repo")
from mipsfusion_amd import ops, _lib
from mipsfusion_amd.model import MLP_reg
dev = torch.device("cuda:0")
for M in (1000, 4800, 70000):
    torch.manual_seed(M)
    dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
    ws = dec.ordered_parameters()
    packed, packed16 = ops.decoder_pack(ws), ops.decoder_pack16(ws)
    x = torch.rand(M, 3, device=dev)
    feat = (torch.randn(M, 32, device=dev) * 0.3).contiguous()
    lay = _lib.FEAT_AOS
    o32, s32 = ops.decoder_fwd(packed, feat, lay, x, None, M, save=True)
    o16, s16 = ops.decoder_fwd(packed, feat, lay, x, None, M, save=True, precision="f16x3", packed16=packed16)
    n_blocks = (M + 127) // 128
    n_act = n_blocks * 4 * 192 * 64
    n_tiles = (M + 31) // 32
    a32 = s32[:n_act].view(-1, 3, 4, 4, 64, 4)[:n_tiles]     # [tile][mat][rt][g][lane][4]
    a16 = s16[:n_act].view(-1, 3, 4, 4, 64, 4)[:n_tiles]
    d = (a16 - a32).abs()
    print(M, "out err", float((o16 - o32).abs().max()), "per-matrix max err", [float(d[:, m].max()) for m in range(3)],
          "scale", [float(a32[:, m].abs().max()) for m in range(3)])
    bad = (d > 1e-3).nonzero()
    if len(bad):
        print("  first bad entries (tile, mat, rt, g, lane, c):", bad[:6].tolist(), "n bad", len(bad), "tiles", sorted(set(bad[:, 0].tolist()))[:10])
    m32 = s32[n_act:n_act + n_blocks * 4 * 256].view(torch.int32)
    m16 = s16[n_act:n_act + n_blocks * 4 * 256].view(torch.int32)
    print("  masks equal on live tiles:", bool(torch.equal(m32[:n_tiles * 256], m16[:n_tiles * 256])))
