"""Host CPU placement for the processes that drive a GPU (bench.py, tools/run_sequence.py, any caller's main).

The GPU hosts are two-socket machines (2 x 64 cores, 2 NUMA nodes) whose cgroup grants a quota of 16 CPUs but an affinity
mask of all 256 (128 cores x 2 hardware threads).  Left alone, the sample-producer threads of mipsfusion_amd.sequence and torch's OpenMP pool migrate
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


def _siblings(cpu):
    """hardware threads of `cpu`'s physical core (itself alone when the topology is not exposed)"""
    try:
        return _parse_cpulist(open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list").read()) or {cpu}
    except OSError:
        return {cpu}


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


def cpu_busy_fractions(interval=0.1):
    """-> {cpu: fraction of `interval` seconds it was not idle} from two readings of /proc/stat ({} if unreadable)."""
    import time

    def read():
        out = {}
        try:
            for line in open("/proc/stat"):
                if line.startswith("cpu") and line[3].isdigit():
                    f = line.split()
                    v = [int(x) for x in f[1:9]]
                    out[int(f[0][3:])] = (sum(v), v[3] + v[4])          # total, idle + iowait
        except (OSError, ValueError, IndexError):
            return {}
        return out
    a = read()
    time.sleep(interval)
    b = read()
    busy = {}
    for c, (tot1, idle1) in b.items():
        if c in a:
            dt, di = tot1 - a[c][0], idle1 - a[c][1]
            busy[c] = 0.0 if dt <= 0 else max(0.0, 1.0 - di / dt)
    return busy


def confine_to_numa_node(max_cpus=32, local_rank=0, local_world=1, avoid_busy=True):
    """Restrict this process (and every thread it starts later) to <= max_cpus CPUs of one NUMA node.  With several ranks
    on a host, rank r takes the r-th slice: ranks are spread over the nodes first, then over the CPUs of a node.  A single
    rank picks, of all nodes, the `max_cpus` CPUs that were the LEAST BUSY over the last 0.1 s (the hosts are shared:
    everybody's default is CPU 0 upwards, and a neighbour on the same cores costs the generator stage 30 %).
    -> the CPU list chosen, or None where affinities are not supported / nothing had to change."""
    if not hasattr(os, "sched_getaffinity"):
        return None
    allowed = os.sched_getaffinity(0)
    nodes = [sorted(n & allowed) for n in numa_nodes()]
    nodes = [n for n in nodes if n]
    if not nodes:
        return None
    per_node = max(1, -(-local_world // len(nodes)))              # ranks that share a node
    if local_world > 1:
        # ranks sharing a node split its PHYSICAL cores (both hardware threads of a core go to the same rank)
        node = nodes[(local_rank // per_node) % len(nodes)]
        cores = {}
        for c in node:
            cores.setdefault(min(_siblings(c)), []).append(c)
        keys = sorted(cores)
        n_cores = max(1, len(keys) // per_node)
        first = (local_rank % per_node) * n_cores
        mine = keys[first:first + n_cores] or keys[:n_cores]
        cpus = sorted(c for k in mine for c in cores[k])[:max(max_cpus, 1)]
    else:
        # one logical CPU per PHYSICAL core (two busy threads on SMT siblings run ~35 % slower each), the cores whose
        # both hardware threads were least busy
        busy = cpu_busy_fractions() if avoid_busy else {}
        best = None
        for node in nodes:
            cores = {}
            for c in node:
                sib = _siblings(c)
                key = min(sib)
                if key not in cores:
                    cores[key] = (max((busy.get(t, 0.0) for t in sib), default=0.0), min(t for t in sib if t in node))
            ranked = sorted(cores.values())
            pick = [cpu for _, cpu in ranked[:max_cpus]]
            score = (sum(b for b, _ in ranked[:max_cpus]), -len(pick))
            if pick and (best is None or score < best[0]):
                best = (score, sorted(pick))
        cpus = best[1]
    if set(cpus) == set(allowed):
        return None
    os.sched_setaffinity(0, cpus)
    return cpus
