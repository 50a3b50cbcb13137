"""HIP-graph replay of a whole call (the restart pipelines are ~150 launches of ~25 us of Python
each; the model-level one is bound by exactly that).

Every launch of this package goes to torch's current stream and every buffer comes from torch's allocator, so a call can be
captured as it is -- the second stream of the surface categories included (it forks from and joins the capturing stream) --
and replayed with no Python between its kernels.  The contract is that of any stream capture: the call must not read device
results on the host, its inputs are the device tensors it closed over AT THEIR ADDRESSES (new data is copied into them, e.g.
by the H2D copy of the next set of restart files), and its outputs are the objects the capture returned, refreshed in place by
every replay.
"""
from typing import Any, Callable

import torch


class GraphedCall:
    """``GraphedCall(fn)`` runs ``fn()`` a few times eagerly (allocations settle, lazily built tables exist), captures one
    more run, and ``replay()`` re-executes it.  ``result`` is what the captured run returned (any structure of this package's
    Datasets / DataArrays / tensors); it is updated in place by each replay."""

    def __init__(self, fn: Callable[[], Any], warmup: int = 2, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedCall needs a 'cuda' (ROCm) device")
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(self._device):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # (capture needs a non-default stream; warm up on one, as torch's recipe does)
                for _ in range(max(1, int(warmup))):
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self.result = fn()

    def replay(self):
        """Re-execute the captured call on the current stream (asynchronously, like any launch); returns ``result``."""
        with torch.cuda.device(self._device):
            self._graph.replay()
        return self.result
