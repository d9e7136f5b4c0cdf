import sys, os, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mipsfusion_amd import ops, _lib
from mipsfusion_amd._lib import lib, dptr, check, stream_ptr
from mipsfusion_amd.model import MLP_reg
dev = torch.device("cuda:0")
torch.manual_seed(1100)
M = 1000
dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
ws = dec.ordered_parameters()
packed16 = ops.decoder_pack16(ws)
x = torch.rand(M, 3, device=dev); feat = (torch.randn(M, 32, device=dev) * 0.3).contiguous()
lay = _lib.FEAT_AOS
dout = torch.randn(M, 10, device=dev) * 1e-4
out, saved = ops.decoder_fwd(None, feat, lay, x, None, M, save=True, precision="f16x3", packed16=packed16)
dfeat = torch.empty_like(feat); dx = torch.empty(M, 3, device=dev)
dacts = []
for _ in range(2):
    dact = torch.zeros(lib().mipsf_decoder_dact_floats(M), device=dev)
    check(lib().mipsf_decoder_bwd_chain16(dptr(packed16), lay, dptr(x), dptr(out), dptr(dout), dptr(saved), dptr(dfeat), dptr(dx), dptr(dact), M, stream_ptr()), "chain")
    torch.cuda.synchronize(); dacts.append(dact)
print("chain deterministic:", torch.equal(dacts[0], dacts[1]))
res = []
for rc in (0, 0, 1, 1):
    g = [torch.zeros_like(w) for w in ws]
    st = ops._decoder_struct(g, _lib.DecoderGrads)
    partial = torch.zeros(lib().mipsf_decoder_wgrad_partial_floats(), device=dev)
    check(lib().mipsf_decoder_wgrad16_ex(dptr(packed16) if rc else None, dptr(feat), lay, dptr(x), dptr(saved), dptr(dacts[0]), C.byref(st), dptr(partial), _lib.PREC["f16x3"], M, stream_ptr()), "w")
    torch.cuda.synchronize(); res.append((g, partial))
for a, b, name in ((0, 1, "stored twice"), (2, 3, "recompute twice"), (0, 2, "stored vs recompute")):
    print(name, [float((p - q).abs().max()) for p, q in zip(res[a][0], res[b][0])], "partial equal", torch.equal(res[a][1], res[b][1]))
