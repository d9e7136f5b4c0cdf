import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
print("loadavg", open("/proc/loadavg").read().strip())
print("torch threads default", torch.get_num_threads())
a = torch.randn(2000, 2000)
idx = torch.randint(0, 1000000, (2000000,))
src = torch.randn(2000000, 2)
for nt in (1, 4, 8, 16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    t = time.time(); (a @ a).sum().item(); t1 = time.time() - t
    t = time.time(); torch.zeros(1000000, 2).index_add_(0, idx, src); t2 = time.time() - t
    t = time.time(); torch.sin(src).sum().item(); t3 = time.time() - t
    print(f"threads {nt:4d}: matmul {t1*1e3:8.1f} ms  index_add {t2*1e3:8.1f} ms  sin {t3*1e3:8.1f} ms", flush=True)
    if t1 > 20: break
