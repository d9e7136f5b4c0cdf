"""``FusedAdam``: drop-in for ``torch.optim.Adam`` as the reference configures it (mipsfusion.py:580-584,
InactiveMap.py:53-57): per-group lr / eps / weight_decay (L2 added to the gradient, not AdamW), betas, DENSE
semantics (moments decay and parameters move where the gradient is zero).  One HIP kernel per parameter tensor
streams p, g, m, v once (28 B per parameter with the fused zero-grad)."""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False):
        """zero_grad=True (extension) clears each gradient in the same pass (the reference calls
        ``zero_grad()`` right after ``step()``, mipsfusion.py:330-335)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            small = []
            for p in group["params"]:
                if p.grad is None or p.numel() == 0:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam runs on GPU parameters only (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                if p.numel() < (1 << 18):
                    small.append(p)
                    continue
                ops.adam_step(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"],
                              group["weight_decay"], st["step"], zero_grad)
            # small tensors of a group (the decoder's ten nn.Linear tensors) share one launch per step count
            by_step = {}
            for p in small:
                by_step.setdefault(self.state[p]["step"], []).append(p)
            for step, ps in by_step.items():
                ops.adam_step_multi([p.data for p in ps], [p.grad for p in ps],
                                    [self.state[p]["exp_avg"] for p in ps], [self.state[p]["exp_avg_sq"] for p in ps],
                                    group["lr"], b1, b2, group["eps"], group["weight_decay"], step, zero_grad)
        return loss
