"""Dataset <-> [sample, feature] adapters (external/fv3fit/fv3fit/_shared/stacking.py:7-52,
xr_prediction.py:51-72,111-139).

The reference stacks every non-vertical dim into ``_fv3fit_sample`` (a transposing copy plus a
MultiIndex), calls Keras on ``[sample, z]`` arrays and unstacks.  The HIP kernel reads arrays
through (feature stride, sample stride), so here "stacking" is only a reshape of the native
``[z, ...]`` (or ``[..., z]``) memory; a copy happens only for dim orders that interleave the
vertical dim with the sample dims.
"""
from typing import Hashable, List, Sequence, Tuple

from ..xr_compat import DataArray, Dataset

SAMPLE_DIM_NAME = "_fv3fit_sample"
DATASET_DIM_NAME = "dataset"
Z_DIM_NAMES = ["z", "pfull"]


def _infer_dimension_order(ds: Dataset) -> Tuple:
    dim_order: List[Hashable] = []
    for variable in ds:
        for dim in ds[variable].dims:
            if dim not in dim_order:
                dim_order.append(dim)
    return tuple(dim_order)


def stack(ds: Dataset, unstacked_dims: Sequence[str] = None) -> Dataset:
    """Return a Dataset whose variables are ``[_fv3fit_sample, *unstacked_dims]`` (a host-side
    utility kept for API parity; ``predict`` itself does not need it)."""
    import numpy as np

    unstacked = [d for d in _infer_dimension_order(ds) if d in (unstacked_dims or [])]
    unstacked.sort()
    stack_dims = [d for d in _infer_dimension_order(ds) if d not in unstacked]
    out = Dataset(attrs=ds.attrs)
    for name in ds:
        da = ds[name]
        missing = [d for d in stack_dims if d not in da.dims]
        if missing:
            raise ValueError(f"variable {name!r} lacks sample dims {missing}")
        keep = [d for d in unstacked if d in da.dims]
        t = da.transpose(*stack_dims, *keep)
        n = int(np.prod([da.sizes[d] for d in stack_dims])) if stack_dims else 1
        data = t.values.reshape((n,) + tuple(da.sizes[d] for d in keep))
        out[name] = DataArray(data, dims=(SAMPLE_DIM_NAME,) + tuple(keep), attrs=da.attrs)
    return out


def match_prediction_to_input_coords(input: Dataset, prediction: Dataset) -> Dataset:
    """Same coords as the input and the input's dimension order (stacking.py:40-52)."""
    input_coords = input.coords
    out = Dataset(attrs=prediction.attrs)
    dim_order = _infer_dimension_order(input)
    for name in prediction:
        da = prediction[name]
        coords = {k: input_coords[k] for k in da.dims if k in input_coords}
        order = [d for d in dim_order if d in da.dims]
        order += [d for d in da.dims if d not in order]
        out[name] = da._replace(coords=coords).transpose(*order)
    return out


def column_sources(x: Dataset, names: Sequence[Hashable], unstacked_dims: Sequence[str]):
    """The variables ``names`` of ``x`` as device arrays ``[feature, sample]`` over one common sample order -- what the
    reference gets from ``stack`` + ``pack`` (stacking.py:7-37), here views of the native ``[z, ...]`` (or ``[..., z]``)
    memory wherever the dim order allows.  Returns (name -> tensor, sample dims, dim sizes, the vertical dim's name, the
    first variable's own data -- the caller returns host arrays for host input)."""
    import numpy as np

    from ..cubedsphere._device import on_device

    arrays = {name: x[name] for name in names}  # KeyError for a missing variable
    zdims = set(unstacked_dims)
    order = _infer_dimension_order(Dataset({k: v for k, v in arrays.items()}))
    sample_dims = [d for d in order if d not in zdims]
    sizes = {}
    for da in arrays.values():
        for d, n in da.sizes.items():
            if sizes.setdefault(d, n) != n:
                raise ValueError(f"conflicting sizes for dimension {d!r}")
    n_samples = int(np.prod([sizes[d] for d in sample_dims])) if sample_dims else 1
    zname = next((d for d in order if d in zdims), unstacked_dims[0] if unstacked_dims else "z")
    host_input = None
    sources = {}
    for name, da in arrays.items():
        zs = [d for d in da.dims if d in zdims]
        if len(zs) > 1:
            raise ValueError(f"variable {name!r} has more than one unstacked dim: {zs}")
        own_samples = [d for d in da.dims if d not in zdims]
        if set(own_samples) != set(sample_dims):
            raise ValueError(
                f"variable {name!r} has sample dims {own_samples}, expected {sample_dims} "
                "(broadcasting inputs over sample dims is not supported)"
            )
        if host_input is None:
            host_input = da.data
        nfeat = da.sizes[zs[0]] if zs else 1
        t = on_device(da.data)
        if own_samples == sample_dims and (not zs or da.dims[0] == zs[0]):
            t2 = t.contiguous().reshape(nfeat, n_samples)            # native [z, ...]: a view
        elif own_samples == sample_dims and da.dims[-1] == zs[0]:
            t2 = t.contiguous().reshape(n_samples, nfeat).t()        # [..., z]: a strided view
        else:
            t2 = on_device(da.transpose(*zs, *sample_dims).data).contiguous().reshape(nfeat, n_samples)
        sources[name] = t2
    return sources, sample_dims, sizes, zname, host_input
