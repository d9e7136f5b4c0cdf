"""stdin: hipcc -Rpass-analysis=kernel-resource-usage output -> one line per kernel (tools/kres.sh)."""
import re
import subprocess
import sys

rows, cur = [], {}
for line in sys.stdin:
    if "error:" in line:
        print(line, end="")
    m = re.search(r"remark:\s+(Function Name|Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|"
                  r"LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k in ("Function Name", "Name"):
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception:
        name = r["name"]
    name = re.sub(r"\(.*", "", name).replace("mipsf::", "").replace("void ", "")
    print("%-64s vgpr %4s agpr %3s vspill %3s scratch %4s lds %6s occ %s" % (
        name, r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("VGPRs Spill", "?"), r.get("ScratchSize [bytes/lane]", "?"),
        r.get("LDS Size [bytes/block]", "?"), r.get("Occupancy [waves/SIMD]", "?")))
