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
