"""hipGraph capture of whole optimisation iterations.

The hot path is ~25 kernel launches + autograd bookkeeping per iteration: ~0.8 ms of host time on this stack, which
bounds small iterations (tracking: 1000 rays) and leaves gaps between kernels in large ones.  Everything on the path
is capture-safe -- kernels are enqueued on torch's current stream, scratch comes from torch's allocator, the
optimiser's step-dependent scalars live in device memory (``FusedAdam(capturable=True)``) -- so ``n_inner``
consecutive iterations are recorded once with ``torch.cuda.graph`` and replayed with one launch.

Usage: write ``step_fn(k)`` so that it only reads *static* device tensors (copy fresh inputs into them before
``replay()``) and follows the same Python control flow every time it is called with the same ``k``.

Stream rule: autograd runs a leaf's ``AccumulateGrad`` on the stream that was current when the leaf first took part
in autograd.  If that was a different stream than the capture stream, autograd forks a cross-stream wait out of the
capture and ``hipStreamEndCapture`` dies on the unjoined fork (ROCm 7.2 segfaults instead of returning an error).
So: do ALL work that touches the optimised tensors -- setup iterations, warm-up, capture -- on ONE non-default stream
(``work_stream()`` below) and pass it here.
"""
import torch


def work_stream(device=None) -> torch.cuda.Stream:
    """Make a fresh non-default stream current (for the rest of the process) and return it."""
    s = torch.cuda.Stream(device=device)
    s.wait_stream(torch.cuda.current_stream(device))
    torch.cuda.set_stream(s)
    return s


class GraphedSteps:
    def __init__(self, step_fn, n_inner: int = 1, warmup: int = 2, stream: torch.cuda.Stream = None):
        self.n_inner = n_inner
        stream = stream if stream is not None else torch.cuda.current_stream()
        if stream == torch.cuda.default_stream():
            raise RuntimeError("capture needs a non-default stream: call mipsfusion_amd.graph.work_stream() first "
                               "and create/optimise the model on it")
        with torch.cuda.stream(stream):
            for _ in range(warmup):                         # allocator / lazy-init warm-up on the capture stream
                for k in range(n_inner):
                    step_fn(k)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=stream):
                for k in range(n_inner):
                    step_fn(k)

    def replay(self):
        self.graph.replay()
