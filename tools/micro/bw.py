import torch, time
dev = torch.device("cuda:0")
n = 850 * 1024 * 1024 // 4
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
us = t(lambda: x.sum())
print(f"sum  (read 850 MB): {us:.1f} us  {x.numel()*4/us/1e6:.2f} TB/s")
us = t(lambda: x.max())
print(f"max  (read 850 MB): {us:.1f} us  {x.numel()*4/us/1e6:.2f} TB/s")
us = t(lambda: y.copy_(x))
print(f"copy (read+write): {us:.1f} us  {2*x.numel()*4/us/1e6:.2f} TB/s")
us = t(lambda: y.fill_(1.0))
print(f"fill (write): {us:.1f} us  {x.numel()*4/us/1e6:.2f} TB/s")
