"""Host CPU placement for the processes that drive a GPU (bench.py, tools/run_sequence.py, any caller's main).

The GPU hosts are two-socket machines (2 x 64 cores, 2 NUMA nodes) whose cgroup grants a quota of 16 CPUs but an affinity
mask of all 256.  Left alone, the sample-producer threads of mipsfusion_amd.sequence and torch's OpenMP pool migrate
across both sockets: the serial torch-CPU generator stream -- the stage that bounds the frame time with reference-exact
sampling -- then runs at 9 or at 14 ms per frame from one run to the next (measured, tools/micro/ab_affinity.sh), and the
frame time with it (10.7 vs 13-15 ms).  Confining the process to up to 32 CPUs of ONE node makes it 8.7-9.3 ms every
time.  (16 CPUs are too few: the python-`random` stage then queues behind the OpenMP threads.)"""
import os


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_nodes():
    """-> list of CPU sets, one per NUMA node (a single set of everything when the topology is not exposed)."""
    base = "/sys/devices/system/node"
    nodes = []
    try:
        for name in sorted(os.listdir(base)):
            if name.startswith("node") and name[4:].isdigit():
                nodes.append(_parse_cpulist(open(os.path.join(base, name, "cpulist")).read()))
    except OSError:
        pass
    return nodes or [set(range(os.cpu_count() or 1))]


def confine_to_numa_node(max_cpus=32, local_rank=0, local_world=1):
    """Restrict this process (and every thread it starts later) to <= max_cpus CPUs of one NUMA node.  With several ranks
    on a host, rank r takes the r-th slice: ranks are spread over the nodes first, then over the CPUs of a node.
    -> the CPU list chosen, or None where affinities are not supported / nothing had to change."""
    if not hasattr(os, "sched_getaffinity"):
        return None
    allowed = os.sched_getaffinity(0)
    nodes = [sorted(n & allowed) for n in numa_nodes()]
    nodes = [n for n in nodes if n]
    if not nodes:
        return None
    per_node = max(1, -(-local_world // len(nodes)))              # ranks that share a node
    node = nodes[(local_rank // per_node) % len(nodes)] if local_world > 1 else max(nodes, key=len)
    width = min(max_cpus, max(1, len(node) // per_node))
    first = (local_rank % per_node) * width
    cpus = node[first:first + width] or node[:width]
    if set(cpus) == set(allowed):
        return None
    os.sched_setaffinity(0, cpus)
    return cpus
