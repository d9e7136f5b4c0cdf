"""Build-time audit for the hazard of DESIGN.md 4h: a packed fp32 operation whose LOW result reads the HIGH half of a source
(v_pk_{add,mul,fma}_f32 with a 1 in op_sel) lost that operand in lanes 48..63 when the decoder's kernels ran beside it.  Compiles
every unit of mipsfusion_amd/csrc to ISA and lists, per kernel, the packed fp32 operations and how many of them cross.
    python tools/audit_packed.py            -> table; exit code 1 if any kernel holds a crossed one
Kernels that would hold one carry MIPSF_SINGLE_FP32 (common.h: target("no-packed-fp32-ops")); tests/test_host_cpu.py runs this."""
import collections
import concurrent.futures
import os
import re
import shlex
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mipsfusion_amd", "csrc")
UNITS = ("capi", "hashgrid", "elementwise", "render", "decoder", "decoder16", "wgrad16", "pose", "ro")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-w", "-I" + os.path.join(ROOT, "include")]


def unit_isa(unit, out_dir, extra=()):
    """ISA of one unit under the product's flags + `extra` + whatever $EXTRA holds (the variable the Makefile appends to its
    own flags: `make EXTRA=-DFOO` and `EXTRA=-DFOO python tools/audit_packed.py` see the same build)"""
    extra = tuple(extra) + tuple(shlex.split(os.environ.get("EXTRA", "")))
    out = os.path.join(out_dir, unit + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", "-o", out, os.path.join(SRC, unit + ".hip")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def audit(path):
    """{kernel: (packed fp32 ops, crossed ones, [the crossed lines])}"""
    kern, rows = None, collections.OrderedDict()
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kern = m.group(1)
        if kern and re.search(r"\bv_pk_(add|mul|fma)_f32\b", ln):
            n, c, lines = rows.get(kern, (0, 0, []))
            m2 = re.search(r"op_sel:\[([01,]+)\]", ln)
            crossed = bool(m2 and "1" in m2.group(1))
            rows[kern] = (n + 1, c + crossed, lines + [ln.strip()] if crossed else lines)
    return rows


def audit_all(extra=(), units=UNITS):
    """[(unit, kernel, packed ops, crossed ones, the crossed lines)] for every kernel, all units compiled in parallel"""
    with tempfile.TemporaryDirectory() as d:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            paths = list(ex.map(lambda u: unit_isa(u, d, extra), units))
        return [(unit, k, n, c, lines) for unit, p in zip(units, paths) for k, (n, c, lines) in audit(p).items()]


def main(extra=()):
    rows = audit_all(extra)
    bad = [r for r in rows if r[3]]
    print(f"{len(rows)} kernels hold {sum(r[2] for r in rows)} packed fp32 operations; {len(bad)} kernels hold one whose low result reads a high half")
    for unit, k, n, c, lines in bad:
        print(f"  {unit + '.hip':16s} {k[:90]:90s} {c} of {n}: {lines[0]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(tuple(sys.argv[1:])))
