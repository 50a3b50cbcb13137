"""Batch inference over many snapshots that start and end in host memory (BASELINE configs[3]: a year of
3-hourly C384 snapshots, tile t on GPU t): host -> device copies, the fused MLP kernel and device -> host
copies of consecutive snapshots run concurrently on three HIP streams over a ring of buffers, so the
sustained rate is that of the slowest of the three stages (PCIe, for this network) instead of their sum.

The reference processes one snapshot at a time, synchronously (``predict_on_dataset`` per time step,
external/fv3fit/fv3fit/_shared/xr_prediction.py:111-139; offline loops in
workflows/diagnostics/fv3net/diagnostics/offline/compute.py).  torch supplies pinned host memory, streams
and events only.
"""
from typing import Callable, Dict, List, Mapping, Optional

import numpy as np
import torch

from ..mlp import MlpModel

HostArrays = Dict[str, np.ndarray]


class SnapshotStream:
    """``SnapshotStream(model, n_samples).run(n, fill, consume)``.

    ``fill(i, inputs)`` writes snapshot ``i``'s source arrays (``[feature, sample]``, or ``[sample]`` for
    single-level variables) into the pinned host arrays ``inputs``; ``consume(i, outputs)`` receives the
    outputs of snapshot ``i`` as pinned host arrays that stay valid until it returns.  Both run on the
    calling thread, in snapshot order, overlapped with the device work of the neighbouring snapshots.
    """

    def __init__(self, model: MlpModel, n_samples: int, in_dtype=np.float32, out_dtype=np.float32, n_buffers: int = 3,
                 device: Optional[torch.device] = None):
        if n_buffers < 2:
            raise ValueError("at least two buffers are needed to overlap anything")
        self.model = model
        self.n_samples = int(n_samples)
        self.device = torch.device(device) if device is not None else model.device
        spec = model.spec
        need = spec.source_nfeat()
        nfeat = {o.name: o.nfeat for o in spec.outputs}
        for r in spec.residuals:
            nfeat[r.name] = nfeat[r.output]
        tin = torch.float64 if np.dtype(in_dtype) == np.float64 else torch.float32
        tout = torch.float64 if np.dtype(out_dtype) == np.float64 else torch.float32
        self._host_in: List[Dict[str, torch.Tensor]] = []
        self._dev_in: List[Dict[str, torch.Tensor]] = []
        self._host_out: List[Dict[str, torch.Tensor]] = []
        self._dev_out: List[Dict[str, torch.Tensor]] = []
        for _ in range(n_buffers):
            self._host_in.append({k: torch.empty((need[k], self.n_samples), dtype=tin).pin_memory() for k in spec.sources})
            self._dev_in.append({k: torch.empty((need[k], self.n_samples), dtype=tin, device=self.device) for k in spec.sources})
            self._host_out.append({k: torch.empty((nfeat[k], self.n_samples), dtype=tout).pin_memory() for k in spec.output_names})
            self._dev_out.append({k: torch.empty((nfeat[k], self.n_samples), dtype=tout, device=self.device)
                                  for k in spec.output_names})
        self.n_buffers = n_buffers
        self._s_in = torch.cuda.Stream(self.device)
        self._s_run = torch.cuda.Stream(self.device)
        self._s_out = torch.cuda.Stream(self.device)
        self.bytes_in = sum(t.numel() * t.element_size() for t in self._host_in[0].values())
        self.bytes_out = sum(t.numel() * t.element_size() for t in self._host_out[0].values())

    def host_inputs(self, slot: int) -> HostArrays:
        return {k: t.numpy() for k, t in self._host_in[slot].items()}

    def run(self, n_snapshots: int, fill: Callable[[int, HostArrays], None],
            consume: Callable[[int, Mapping[str, np.ndarray]], None]) -> None:
        nb = self.n_buffers
        uploaded = [torch.cuda.Event() for _ in range(nb)]    # H2D of the slot's snapshot done
        computed = [torch.cuda.Event() for _ in range(nb)]    # kernel done: inputs free, outputs ready
        downloaded = [torch.cuda.Event() for _ in range(nb)]  # D2H done: device outputs free, host outputs ready
        in_flight: List[int] = []

        def retire(i):
            slot = i % nb
            downloaded[slot].synchronize()
            consume(i, {k: t.numpy() for k, t in self._host_out[slot].items()})

        for i in range(n_snapshots):
            slot = i % nb
            if len(in_flight) == nb:  # the slot's previous snapshot must have left the buffers
                retire(in_flight.pop(0))
            fill(i, self.host_inputs(slot))
            with torch.cuda.stream(self._s_in):
                if i >= nb:
                    self._s_in.wait_event(computed[slot])  # the kernel that read these device inputs
                for k, t in self._host_in[slot].items():
                    self._dev_in[slot][k].copy_(t, non_blocking=True)
                uploaded[slot].record(self._s_in)
            with torch.cuda.stream(self._s_run):
                self._s_run.wait_event(uploaded[slot])
                if i >= nb:
                    self._s_run.wait_event(downloaded[slot])  # the copy that read these device outputs
                self.model.predict(self._dev_in[slot], out=self._dev_out[slot])
                computed[slot].record(self._s_run)
            with torch.cuda.stream(self._s_out):
                self._s_out.wait_event(computed[slot])
                for k, t in self._dev_out[slot].items():
                    self._host_out[slot][k].copy_(t, non_blocking=True)
                downloaded[slot].record(self._s_out)
            in_flight.append(i)
        while in_flight:
            retire(in_flight.pop(0))
