"""``FusedAdam``: drop-in for ``torch.optim.Adam`` as the reference configures it (mipsfusion.py:580-584,
InactiveMap.py:53-57): per-group lr / eps / weight_decay (L2 added to the gradient, not AdamW), betas, DENSE
semantics (moments decay and parameters move where the gradient is zero).  One HIP kernel per large tensor streams
p, g, m, v once (28 B per parameter with the fused zero-grad); a group's small tensors share one launch.

``capturable=True`` keeps the step counter and the two step-dependent scalars in device memory so that
``step()`` can be captured into a hipGraph (``torch.cuda.graph``) and replayed."""
import torch

from . import _lib, ops


import os

_SPLIT_LAUNCHES = os.environ.get("MIPSF_ADAM_SPLIT", "0") == "1"      # A/B: advance + per-tensor launches instead of one


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = capturable
        self._dev = {}          # group index -> (step int32[1], hyper float[2]) when capturable

    def _ensure_state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    @staticmethod
    def _all_small(work):
        n_t = sum(len(live) for _, _, live, _ in work)
        return (len(work) <= _lib.ADAM_MAX_GROUPS and n_t <= _lib.ADAM_MAX_TENSORS and
                all(p.numel() <= _lib.ADAM_SMALL_MAX_NUMEL for _, _, live, _ in work for p in live))

    @torch.no_grad()
    def reset(self):
        """Back to the state of a freshly constructed optimiser (moments and step counters zero), IN PLACE: the
        reference builds a new pose optimiser for every frame / BA round (mipsfusion.py:472-475, 300-303); a
        captured iteration keeps reading the same state tensors, so they are cleared instead of replaced."""
        tensors = []
        for st in self.state.values():
            if st:
                st["step"] = 0
                tensors += [st["exp_avg"], st["exp_avg_sq"]]
        tensors += [step_dev for step_dev, _ in self._dev.values()]
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault((t.dtype, t.device), []).append(t)
        for group in by_dtype.values():
            torch._foreach_zero_(group)         # one launch per dtype instead of one per tensor

    # ---- checkpointing: with capturable=True the live step counters are the per-group device tensors (graph replays
    # advance them without the host seeing it), so they are read back / re-seeded around (load_)state_dict
    def _sync_steps_from_device(self):
        for gi, (step_dev, _) in self._dev.items():
            n = int(step_dev.item())
            for p in self.param_groups[gi]["params"]:
                if self.state.get(p):
                    self.state[p]["step"] = n

    def state_dict(self):
        self._sync_steps_from_device()
        return super().state_dict()

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():          # torch casts a saved int step to a tensor of the param's dtype
            if st and torch.is_tensor(st.get("step")):
                st["step"] = int(st["step"].item())
        for gi, (step_dev, _) in self._dev.items():     # keep the tensors a captured graph reads, refresh their value
            steps = {self.state[p]["step"] for p in self.param_groups[gi]["params"] if self.state.get(p)}
            if len(steps) > 1:
                raise RuntimeError(f"capturable FusedAdam: group {gi} has parameters at different steps {sorted(steps)}")
            if steps:
                step_dev.fill_(steps.pop())

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False):
        """zero_grad=True (extension) clears each gradient in the same pass (the reference calls
        ``zero_grad()`` right after ``step()``, mipsfusion.py:330-335)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work = []
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            live = [p for p in group["params"] if p.grad is not None and p.numel() > 0]
            if not live:
                continue
            for p in live:
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam runs on GPU parameters only (no CPU fallback)")
                self._ensure_state(p)["step"] += 1
            hyper = None
            if self.capturable:
                # one device-resident counter per group (all its tensors step together)
                if len({self.state[p]["step"] for p in live}) > 1:
                    raise RuntimeError(f"capturable FusedAdam: the parameters of group {gi} are at different steps "
                                       "(some had no gradient in an earlier step); use capturable=False for that")
                if gi not in self._dev:
                    dev = live[0].device
                    start = self.state[live[0]]["step"] - 1
                    self._dev[gi] = (torch.full((1,), start, dtype=torch.int32, device=dev),
                                     torch.zeros(2, dtype=torch.float32, device=dev))
                hyper = self._dev[gi][1]
            work.append((gi, group, live, hyper))
        if self.capturable and work and self._all_small(work):
            # an optimiser of tiny tensors only (pose optimisers): counters, bias corrections and every update in ONE
            # launch instead of advance + one multi-tensor step per group
            ops.adam_step_small([(self._dev[gi][0], self._dev[gi][1], group["lr"], group["betas"][0], group["betas"][1],
                                  group["eps"], group["weight_decay"],
                                  [(p.data, p.grad, self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for p in live])
                                 for gi, group, live, _ in work], zero_grad)
            return loss
        n_t = sum(len(live) for _, _, live, _ in work)
        if (self.capturable and work and not _SPLIT_LAUNCHES and len(work) <= _lib.ADAM_MAX_GROUPS and n_t <= _lib.ADAM_MAX_TENSORS and
                all(p.numel() < (1 << 32) and p.is_contiguous() for _, _, live, _ in work for p in live)):
            # any optimiser that fits one descriptor (the map optimiser: table + ten decoder tensors): ONE launch instead of
            # advance + one launch per large tensor + one multi-tensor launch per group
            if getattr(self, "_ticket", None) is None or self._ticket.device != work[0][2][0].device:
                self._ticket = torch.zeros(576, dtype=torch.int32, device=work[0][2][0].device)    # MIPSF_ADAM_TICKET_WORDS
            try:
                ops.adam_step_all([(self._dev[gi][0], self._dev[gi][1], group["lr"], group["betas"][0], group["betas"][1],
                                    group["eps"], group["weight_decay"],
                                    [(p.data, p.grad, self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for p in live])
                                   for gi, group, live, _ in work], self._ticket, zero_grad)
            except RuntimeError:
                self._ticket = None            # a failed launch may have left tickets behind: start from a zeroed block
                raise
            return loss
        if self.capturable and work:       # all groups' step counters / bias corrections in ONE launch
            adv = [(self._dev[gi][0], self._dev[gi][1], group["lr"], group["betas"][0], group["betas"][1])
                   for gi, group, _, _ in work]
            for i in range(0, len(adv), 8):
                ops.adam_advance_n(adv[i:i + 8])
        for gi, group, live, hyper in work:
            b1, b2 = group["betas"]
            small = []
            for p in live:
                st = self.state[p]
                if p.numel() < (1 << 18):
                    small.append(p)
                    continue
                ops.adam_step(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"],
                              group["weight_decay"], st["step"], zero_grad, hyper_dev=hyper)
            # small tensors of a group (the decoder's ten nn.Linear tensors) share one launch per step count
            by_step = {}
            for p in small:
                by_step.setdefault(self.state[p]["step"], []).append(p)
            for step, ps in by_step.items():
                ops.adam_step_multi([p.data for p in ps], [p.grad for p in ps],
                                    [self.state[p]["exp_avg"] for p in ps], [self.state[p]["exp_avg_sq"] for p in ps],
                                    group["lr"], b1, b2, group["eps"], group["weight_decay"], step, zero_grad,
                                    hyper_dev=hyper)
        return loss
