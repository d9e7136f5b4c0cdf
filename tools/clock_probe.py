"""Which clock does the part run at under the decoder forward?  (review of round 4, item 4)

Runs the bf16x6 forward kernel alone in a loop for ~2 s per form and, from a second thread, samples the shader clock and the
package power the driver reports (sysfs pp_dpm_sclk / hwmon, the files `rocm-smi --showclocks / --showpower` reads) every
50 ms; prints min / median / max of the samples taken inside the loop, next to the kernel's time.  Also samples an idle second
before and the Adam / hash-grid kernels' loop for contrast.  No special build needed; with a -DD16_TRACE build (MIPSF_LIB) use
tools/micro/fwd_probe.py for the in-kernel s_memtime / s_memrealtime ratio of the same kernel."""
import glob
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIPSF_NO_CONFINE", "1")
import ctypes as C

import numpy as np
import torch

import bench
TRACER = None
from mipsfusion_amd import ops, synth
from mipsfusion_amd._lib import FEAT_LEVEL_MAJOR


def sysfs_cards():
    """the sysfs directory of the device torch runs on (matched by PCI address); every card with pp_dpm_sclk otherwise"""
    cards = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if os.path.exists(os.path.join(d, "pp_dpm_sclk"))]
    try:
        p = torch.cuda.get_device_properties(0)
        addr = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
        mine = [d for d in cards if addr in os.path.realpath(d)]
        print(f"device 0 PCI address {addr}: {mine}")
        if mine:
            return mine
    except Exception as e:      # noqa: BLE001
        print(f"no PCI address from torch: {e}")
    return cards


def read_sclk(card):
    """MHz of the level marked '*' in pp_dpm_sclk (MI300-class parts list the CURRENT clock as the middle level)"""
    try:
        for line in open(os.path.join(card, "pp_dpm_sclk")):
            if "*" in line:
                return float(line.split(":")[1].strip().lower().replace("mhz", "").replace("*", "").strip())
    except (OSError, ValueError, IndexError):
        pass
    return None


def read_power(card):
    for f in glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_input")):
        try:
            return int(open(f).read()) * 1e-6
        except (OSError, ValueError):
            pass
    return None


def smi_once():
    try:
        return subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:      # noqa: BLE001
        return f"rocm-smi failed: {e}"


class Sampler(threading.Thread):
    def __init__(self, cards):
        super().__init__(daemon=True)
        self.cards, self.samples, self.stop = cards, [], False
        self.smi, self.smi_samples = bool(os.environ.get("CLOCK_PROBE_SMI", "1") != "0"), []

    def run(self):
        import json
        while not self.stop:
            self.samples.append((time.perf_counter(), [read_sclk(c) for c in self.cards], [read_power(c) for c in self.cards]))
            if self.smi:        # the management interface's own view of GPU[0] (slow: a few samples per loop)
                try:
                    j = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout)
                    self.smi_samples.append((time.perf_counter(), j))
                except Exception:       # noqa: BLE001
                    pass
            else:
                time.sleep(0.05)


def stats(v):
    v = sorted(x for x in v if x is not None)
    return "n/a" if not v else f"min {v[0]:.0f} median {v[len(v) // 2]:.0f} max {v[-1]:.0f} (n={len(v)})"


def main():
    global TRACER
    import ctypes as C_
    from mipsfusion_amd import _lib
    TRACER = getattr(C_.CDLL(_lib.LIB_PATH), "mipsf_d16_trace_read", None) if os.environ.get("MIPSF_LIB") else None
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    cards = sysfs_cards()
    print(f"sysfs cards with pp_dpm_sclk: {cards}")
    for c in cards:
        try:
            print(c, "pp_dpm_sclk:", open(os.path.join(c, "pp_dpm_sclk")).read().replace("\n", " | "))
        except OSError as e:
            print(c, e)
    cfg = synth.config_headline()
    model, frames, poses = bench.build_submap(cfg, dev, seed=0)
    M = 262144
    packed16 = ops.decoder_pack16(model.decoder.ordered_parameters(), precision="bf16x6")
    x = torch.rand(M, 3, device=dev)
    feat = torch.randn(16, M, 2, device=dev) * 1e-2
    big = torch.empty(64 << 20, device=dev)

    loops = {
        "idle": None,
        "decoder forward bf16x6, evaluation form": lambda: ops.decoder_fwd(None, feat, FEAT_LEVEL_MAJOR, x, None, M, save=False, precision="bf16x6", packed16=packed16),
        "decoder forward bf16x6, training form (lean record)": lambda: ops.decoder_fwd(None, feat, FEAT_LEVEL_MAJOR, x, None, M, save="lean", precision="bf16x6", packed16=packed16),
        "decoder forward f16x3, training form": None,
        "HBM stream (torch mul_ over 256 MB)": lambda: big.mul_(1.0001),
    }
    p3 = ops.decoder_pack16(model.decoder.ordered_parameters(), precision="f16x3")
    loops["decoder forward f16x3, training form"] = lambda: ops.decoder_fwd(None, feat, FEAT_LEVEL_MAJOR, x, None, M, save="lean", precision="f16x3", packed16=p3)
    print("rocm-smi before:\n" + smi_once())
    for name, fn in loops.items():
        s = Sampler(cards)
        s.start()
        t0 = time.perf_counter()
        n = 0
        if fn is None:
            time.sleep(1.0)
        else:
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            if TRACER is not None:
                TRACER(np.zeros(4096 * 16, dtype=np.uint64).ctypes.data_as(C.c_void_p), 1)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            a.record()
            while time.perf_counter() - t0 < 2.0:
                for _ in range(200):
                    fn()
                n += 200
                torch.cuda.synchronize()
            b.record()
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        s.stop = True
        s.join()
        inside = [x for x in s.samples if t0 + 0.3 <= x[0] <= t1]
        line = f"{name:55s}"
        if n:
            line += f" {a.elapsed_time(b) / n * 1e3:8.1f} us/launch over {n} launches;"
        for i, c in enumerate(cards[:1]):
            line += f" sclk MHz {stats([x[1][i] for x in inside])}; power W {stats([x[2][i] for x in inside])}"
        print(line, flush=True)
        if n and TRACER is not None and "decoder forward" in name:
            buf = np.zeros(4096 * 16, dtype=np.uint64)
            TRACER(buf.ctypes.data_as(C.c_void_p), 1)
            t = buf.reshape(4096, 16)
            used = t[:, 15] > 0
            if used.any():
                ticks, ns = float(t[used, :11].sum()), float(t[used, 12].sum()) * 10.0
                print(f"      in-kernel (-DD16_TRACE build): {ticks / t[used, 15].sum():.0f} s_memtime ticks per tile and wave in "
                      f"{ns / t[used, 15].sum():.0f} ns of the 100 MHz wall clock => {ticks / ns:.3f} ticks/ns over {int(t[used, 15].sum())} tiles")
        for ts, j in s.smi_samples:
            if t0 + 0.3 <= ts <= t1:
                for card, v in j.items():
                    keep = {k: x for k, x in v.items() if "sclk" in k.lower() or "power" in k.lower()}
                    print(f"      rocm-smi at +{ts - t0:.2f}s {card}: {keep}")
        if name.startswith("decoder forward bf16x6, training"):
            print("rocm-smi right after that loop:\n" + smi_once())


if __name__ == "__main__":
    main()
