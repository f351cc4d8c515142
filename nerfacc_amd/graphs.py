"""hipGraph replay of fixed-shape steps ("hipGraph capture of the fixed-shape PropNet path").

The batched path -- ``PropNetEstimator.sampling``'s level loop, the batched transmittance, the proposal loss and their
backward passes -- has fixed shapes and no host synchronisation, so a whole step captures into ONE graph launch.
``CapturedStep`` is the product object for that: warm-up on a side stream (allocator pools, lazily built library
state), capture, then ``step()`` replays the graph and returns the same output tensors every time (torch's
static-tensor contract: copy what must outlive the next replay; inputs the step reads are read again from the same
tensors, so write new data INTO them).

When it pays: steps bound by launches.  Measured on MI355X with torch-elementwise proposal networks
(``scripts/graph_cfg3.py``): BASELINE cfg 3 at 2^20 rays is bound by its kernels -- eager 6.16 ms, replay 6.12 ms (0.5 %);
the same step at 4096 rays is bound by launches -- eager 1.18 ms, replay 0.25 ms (4.6 x).  The occupancy-grid sampler does not capture: its output shapes are
data-dependent (one size read per call).
"""
from __future__ import annotations

from typing import Any, Callable

import torch


class CapturedStep:
    """``CapturedStep(fn, warmup=3)``: run ``fn()`` ``warmup`` times on a side stream, capture one more run into a
    ``torch.cuda.CUDAGraph`` (hipGraph), and replay it on every call.

    ``fn`` takes no arguments (close over the tensors it reads), must not synchronise with the host, and returns a tensor
    or a (nested) tuple / list / dict of tensors: the captured run's outputs, returned again -- refreshed in place -- by each
    call.  Gradients computed inside ``fn`` with ``torch.autograd.grad`` (or ``backward`` into ``.grad`` buffers that
    exist before the capture) are part of the graph.
    """

    def __init__(self, fn: Callable[[], Any], warmup: int = 3, device=None) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("nerfacc_amd.CapturedStep needs a ROCm device")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()
        self.replays = 0

    def __call__(self) -> Any:
        self.graph.replay()
        self.replays += 1
        return self.outputs

    step = __call__
