"""Host <-> device plumbing for the labelled-array entry points (torch is memory only)."""
import os
from typing import Hashable, List

import numpy as np
import torch

from .. import _lib
from ..xr_compat import DataArray


def compute_device() -> torch.device:
    _lib.load()  # fail loudly if the HIP extension is not built
    if not torch.cuda.is_available():
        raise RuntimeError(
            "fv3net_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback"
        )
    return torch.device("cuda", torch.cuda.current_device())


def on_device(data) -> torch.Tensor:
    """numpy (host) or torch data -> contiguous-able torch tensor on the compute device."""
    if isinstance(data, torch.Tensor):
        return data if data.is_cuda else data.to(compute_device())
    a = np.asarray(data)
    if a.dtype.byteorder not in ("=", "|"):
        a = a.astype(a.dtype.newbyteorder("="))
    return torch.from_numpy(np.ascontiguousarray(a)).to(compute_device())


def like_input(result: torch.Tensor, original):
    """Device results go back to where the caller's data lived: numpy in -> numpy out."""
    if isinstance(original, torch.Tensor) and original.is_cuda:
        return result
    if isinstance(original, torch.Tensor):
        return result.cpu()
    return result.cpu().numpy()


def horizontal_last(da: DataArray, y_dim: Hashable, x_dim: Hashable):
    """Transpose so the dims are (*outer, y, x); returns (device tensor, outer dim names)."""
    outer: List[Hashable] = [d for d in da.dims if d not in (y_dim, x_dim)]
    t = on_device(da.transpose(*outer, y_dim, x_dim).data)
    return t, outer


def float_tensor(t: torch.Tensor) -> torch.Tensor:
    """numpy promotion of `field * float weights`: integer fields become float64."""
    return t if t.dtype in (torch.float32, torch.float64) else t.to(torch.float64)


def download_all(outputs):
    """{name: device tensor or anything else} -> {name: numpy array or the value as it was}, with one device-to-host
    copy (one synchronisation) per dtype instead of one per array: for small arrays the copies' fixed cost dominates."""
    result = {name: t for name, t in outputs.items() if not (isinstance(t, torch.Tensor) and t.is_cuda)}
    by_dtype = {}
    for name, t in outputs.items():
        if name not in result:
            by_dtype.setdefault(t.dtype, []).append((name, t))
    for items in by_dtype.values():
        if len(items) == 1:
            result[items[0][0]] = items[0][1].cpu().numpy()
            continue
        flat = torch.cat([t.reshape(-1) for _, t in items]).cpu().numpy()
        pos = 0
        for name, t in items:
            result[name] = flat[pos:pos + t.numel()].reshape(tuple(t.shape))
            pos += t.numel()
    return {name: result[name] for name in outputs}


_SIDE_STREAMS = {}


def side_streams(dev: torch.device, n: int = 2):
    """``n`` (<= 2) streams of the device for work beside the calling stream -- streams whose kernels the hardware really
    starts beside the caller's.  The HIP runtime multiplexes streams onto four hardware queues, round-robin in the order
    streams are created, and not every pair of queues runs side by side: a kernel whose stream shares the caller's queue waits
    for the caller's kernel to END, one whose queue shares the caller's dispatch pipe waits until the caller's grid has been
    DISPATCHED -- for a sweep of 13 824 long-lived waves that is nearly the same thing.  Measured on the pressure-level
    pipeline: 6.9 or 7.5 ms for the same call depending on how many streams the process happened to create before
    (``benchmarks/stream_queue_probe.py``).  So, once per device: five candidate streams (consecutive creations cycle through
    the queues) and, for each, a 20-microsecond idle wavefront on it beside a grid of 32 768 idle wavefronts (four rounds of
    50 microseconds) on the calling stream; two candidates of the kind described below are kept.  Cached per device (by the
    first eager call); costs about three milliseconds, once."""
    main = torch.cuda.current_stream(dev)
    key = dev.index
    picked = _SIDE_STREAMS.get(key)
    if picked is None and torch.cuda.is_current_stream_capturing():
        return [torch.cuda.Stream(device=dev) for _ in range(n)]   # (no timing inside a capture; an eager call calibrates)
    if picked is None:
        import ctypes

        candidates = [torch.cuda.Stream(device=dev) for _ in range(5)]   # (stream priorities changed nothing measurable)
        spin = lambda stream, us, wgs: _lib.call_on(dev, "fv3hip_spin", us, wgs, ctypes.c_void_p(stream.cuda_stream))
        for s in candidates + [main]:   # first use (the runtime binds a stream to its queue lazily)
            spin(s, 20, 1)
        torch.cuda.synchronize(dev)
        cost = []
        for s in candidates:
            best = None
            for _ in range(2):
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(main)
                s.wait_event(t0)
                spin(main, 50, 32768)
                spin(s, 20, 1)
                t1.record(s)
                torch.cuda.synchronize(dev)
                ms = t0.elapsed_time(t1)
                best = ms if best is None else min(best, ms)
            cost.append(best)
        if os.environ.get("FV3NET_AMD_DEBUG_STREAMS"):
            print("side-stream calibration (ms until a 0.02 ms kernel beside a 0.2 ms grid has finished):", [round(c, 3) for c in cost], flush=True)
        # Three kinds of answer (MI355X): ~0.23 ms -- the candidate shares the caller's queue; ~0.075 -- its wavefront got a slot
        # when the grid's first round retired (a neighbour that takes what the caller's kernel leaves); ~0.03 -- it started at
        # once, AHEAD of the grid's own wavefronts.  The pipelines want the second kind: with a side stream of the third kind the
        # same pressure-level call took 7.45 instead of 7.03 ms (its short kernels keep cutting in on the sweeps, which are the
        # critical path).
        order = sorted(range(len(candidates)), key=lambda i: abs(cost[i] - 0.075))
        picked = _SIDE_STREAMS[key] = [candidates[i] for i in order[:2]]
    return picked[:n]
