"""Static instruction mix per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only): how many MFMA / VALU /
conversion / LDS / vector-memory / scalar instructions one pass through a kernel's code issues.  usage:
    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S file.hip -o file.s ; python tools/isa_mix.py file.s [filter]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r"\n(_Z[^:\n]*): *; @[^\n]*\n", txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split("s_endpgm")[0]
    if flt not in name:
        continue
    cnt = collections.Counter()
    for line in body.split("\n"):
        m = re.match(r"\s+([a-z_0-9]+)", line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_mfma"):
            k = "mfma"
        elif op.startswith("ds_"):
            k = "lds"
        elif op.startswith(("buffer_", "global_", "flat_")):
            k = "vmem_" + ("st" if "store" in op else "ld")
        elif op.startswith("scratch_"):
            k = "scratch"
        elif op.startswith("v_cvt"):
            k = "v_cvt"
        elif op.startswith("v_"):
            k = "valu"
        elif op.startswith("s_waitcnt"):
            k = "waitcnt"
        elif op.startswith("s_nop"):
            k = "s_nop"
        elif op.startswith("s_"):
            k = "salu"
        else:
            continue
        cnt[k] += 1
    print(name[:90], dict(sorted(cnt.items())))
