import json,sys
for v in sys.argv[1:]:
    d=json.load(open(f"gpurun_out/var/{v}.json"))
    print(f"{v:10s} step {d['ms_per_step']:.4f} eager {d['eager_ms_per_step']:.4f}", {k[:18]:round(x["avg_ms"]*1e3,1) for k,x in d["kernels"].items() if k.startswith("decoder") or k in ("hashgrid_dx","render_bwd","hashgrid_bwd")})
