"""Host <-> device plumbing for the labelled-array entry points (torch is memory only)."""
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
