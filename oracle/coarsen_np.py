"""ORACLE -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's horizontal coarse-graining and of the helpers the
pressure-level path composes.  Arrays are plain numpy with the horizontal dims LAST, (y, x);
the xarray name handling of the reference lives in the product's host layer, not here.

Follows (paths relative to the reference checkout):
  external/vcm/vcm/cubedsphere/coarsen.py:183-218   weighted_block_average
  external/vcm/vcm/cubedsphere/coarsen.py:221-273   edge_weighted_block_average
  external/vcm/vcm/cubedsphere/coarsen.py:591-683   block_edge_sum / block_edge_coarsen
  external/vcm/vcm/cubedsphere/coarsen.py:795-840   block_coarsen
  external/vcm/vcm/cubedsphere/coarsen.py:557-588, 686-786   block_median, _mode, _block_mode
  external/vcm/vcm/cubedsphere/_skimage.py:125-202  block_reduce (view_as_blocks + func)
  external/vcm/vcm/cubedsphere/coarsen.py:843-938   block_upsample(_like)
  external/vcm/vcm/cubedsphere/coarsen.py:109-132   coarsen_coords_coord_func
  external/vcm/vcm/calc/thermo/vertically_dependent.py:41-66,153-179  pressure_at_interface,
      pressure_at_midpoint_log
  external/vcm/vcm/cubedsphere/regridz.py:149-220   _regrid_given_delp, _mask_weights

Third-party arithmetic restated (not vendored in the reference):
  * xarray==0.19.0 ``DataArray.coarsen(...).sum()`` (constraints.txt:315): windows are made by
    reshaping to [..., Y, f, X, f] and reduced with a NaN-skipping sum (NaN -> 0, no min_count);
    min/max/mean are the NaN-skipping numpy reductions.
  * scipy==1.7.3 ``scipy.stats.mode`` (constraints.txt:249): most frequent value, smallest
    value on ties; NaNs never match anything so they are never counted.
Pinned by the reference's known answers in external/vcm/tests/test_cubedsphere.py and the
regression JSONs under external/vcm/tests/_coarsen_restarts_regression_tests/reference
(tests/test_oracle_coarsen.py, tests/golden/).
"""
import numpy as np


def _blocks(a, fy, fx):
    """View [..., ny, nx] as [..., ny/fy, fy, nx/fx, fx] (xarray's coarsen reshape)."""
    ny, nx = a.shape[-2:]
    if ny % fy or nx % fx:
        raise ValueError(f"shape {(ny, nx)} is not divisible by the window {(fy, fx)}")
    return a.reshape(a.shape[:-2] + (ny // fy, fy, nx // fx, fx))


def _nansum_blocks(a, fy, fx):
    b = _blocks(a, fy, fx)
    if np.issubdtype(b.dtype, np.floating):
        b = np.where(np.isnan(b), 0, b)
    return b.sum(axis=(-3, -1))


def weighted_block_average(obj, weights, factor):
    """obj [..., ny, nx]; weights broadcastable to obj with numpy rules."""
    num = _nansum_blocks(obj * weights, factor, factor)
    den = _nansum_blocks(np.asarray(weights), factor, factor)
    with np.errstate(invalid="ignore", divide="ignore"):
        return num / den


def edge_weighted_block_average(obj, spacing, factor, edge="x"):
    """edge='x': mean along x windows, every factor-th row kept; edge='y': the transpose."""
    if edge == "x":
        fy, fx = 1, factor
    elif edge == "y":
        fy, fx = factor, 1
    else:
        raise ValueError(f"'edge' most be either 'x' or 'y'; got {edge}.")
    num = _nansum_blocks(obj * spacing, fy, fx)
    den = _nansum_blocks(np.asarray(spacing), fy, fx)
    with np.errstate(invalid="ignore", divide="ignore"):
        coarsened = num / den
    return coarsened[..., ::factor, :] if edge == "x" else coarsened[..., :, ::factor]


def block_coarsen(a, factor, method="sum"):
    b = _blocks(np.asarray(a), factor, factor)
    isfloat = np.issubdtype(b.dtype, np.floating)
    with np.errstate(invalid="ignore"), np.testing.suppress_warnings() as sup:
        sup.filter(RuntimeWarning)
        if method == "sum":
            return (np.where(np.isnan(b), 0, b) if isfloat else b).sum(axis=(-3, -1))
        if method == "mean":
            return np.nanmean(b, axis=(-3, -1)) if isfloat else b.mean(axis=(-3, -1))
        if method == "min":
            return np.nanmin(b, axis=(-3, -1)) if isfloat else b.min(axis=(-3, -1))
        if method == "max":
            return np.nanmax(b, axis=(-3, -1)) if isfloat else b.max(axis=(-3, -1))
        if method == "median":
            return np.median(b, axis=(-3, -1))
        if method == "mode":
            return block_mode(a, factor)
    raise ValueError(f"unknown method {method}")


def block_edge_coarsen(a, factor, edge="x", method="sum"):
    a = np.asarray(a)
    fy, fx = (1, factor) if edge == "x" else (factor, 1)
    b = _blocks(a, fy, fx)
    isfloat = np.issubdtype(b.dtype, np.floating)
    if method == "sum":
        red = (np.where(np.isnan(b), 0, b) if isfloat else b).sum(axis=(-3, -1))
    elif method == "min":
        red = np.nanmin(b, axis=(-3, -1)) if isfloat else b.min(axis=(-3, -1))
    elif method == "max":
        red = np.nanmax(b, axis=(-3, -1)) if isfloat else b.max(axis=(-3, -1))
    elif method == "mean":
        red = np.nanmean(b, axis=(-3, -1))
    else:
        raise ValueError(method)
    return red[..., ::factor, :] if edge == "x" else red[..., :, ::factor]


def _mode_1d(v, nan_policy="propagate"):
    """scipy.stats.mode 1.7.3 for one window.  'propagate' runs the generic algorithm
    (np.unique scores, counts by ==, strict > keeps the smallest on ties; an all-NaN window
    never beats the initial (0, count 0)); 'omit' drops NaNs first (masked path)."""
    v = np.asarray(v).ravel()
    if nan_policy == "omit":
        v = v[~np.isnan(v)] if np.issubdtype(v.dtype, np.floating) else v
        if v.size == 0:
            return np.nan
    best, best_count = 0, 0
    for score in np.unique(v):
        count = int(np.sum(v == score))
        if count > best_count:
            best, best_count = score, count
    return best


def block_mode(a, factor, nan_policy="propagate"):
    a = np.asarray(a)
    b = _blocks(a, factor, factor)
    b = np.moveaxis(b, -3, -2)  # [..., Y, X, fy, fx]
    flat = b.reshape(b.shape[:-2] + (-1,))
    out = np.empty(flat.shape[:-1], dtype=a.dtype)
    for idx in np.ndindex(out.shape):
        out[idx] = _mode_1d(flat[idx], nan_policy)
    return out


def _upsample_axis(a, factor, axis):
    n = a.shape[axis]
    if n % 2 == 1:  # staggered: last point not repeated
        head = np.repeat(np.take(a, range(n - 1), axis=axis), factor, axis=axis)
        tail = np.take(a, [n - 1], axis=axis)
        return np.concatenate([head, tail], axis=axis)
    return np.repeat(a, factor, axis=axis)


def block_upsample(a, factor):
    a = np.asarray(a)
    return _upsample_axis(_upsample_axis(a, factor, -1), factor, -2)


def coarsen_coords(coord, factor):
    """coarsen_coords_coord_func: ((c0 - 1) // f + 1) as int then float32, c0 = first of each window."""
    c = np.asarray(coord)
    first = c.reshape(-1, factor)[:, 0]
    return ((first - 1) // factor + 1).astype(int).astype(np.float32)


def pressure_at_interface(delp, toa_pressure, z_axis):
    delp = np.asarray(delp)
    top_shape = list(delp.shape)
    top_shape[z_axis] = 1
    top = np.full(top_shape, toa_pressure, dtype=delp.dtype)
    return np.concatenate([top, delp], axis=z_axis).cumsum(axis=z_axis)


def pressure_at_midpoint_log(delp, toa_pressure, z_axis):
    pi = pressure_at_interface(delp, toa_pressure, z_axis)
    dlogp = np.diff(np.log(pi), axis=z_axis)
    return delp / dlogp


def mask_weights(weights, phalf_coarse_on_fine, phalf_fine, z_axis, pfull_coarse_on_fine=None,
                 extrapolate=False):
    """weights broadcastable against the pressure arrays without their z axis."""
    ps = np.take(phalf_fine, [phalf_fine.shape[z_axis] - 1], axis=z_axis)
    if extrapolate:
        cond = pfull_coarse_on_fine < ps
    else:
        n = phalf_coarse_on_fine.shape[z_axis]
        cond = np.take(phalf_coarse_on_fine, range(1, n), axis=z_axis) < ps
    w = np.expand_dims(np.asarray(weights), z_axis % phalf_fine.ndim)
    return np.where(cond, w, 0.0).astype(np.result_type(weights, np.float32) if False else np.asarray(weights).dtype)


def regrid_to_area_weighted_pressure(fields, delp, area, toa_pressure, factor, mappm_fn,
                                     extrapolate=False):
    """regridz.py:31-78 + 149-197 for arrays laid out [tile, z, y, x] (area [tile, y, x]).
    ``mappm_fn(p_in, f_in, p_out)`` works on [ncol, level] arrays.  Returns (dict of regridded
    fields, masked area [tile, z, y, x])."""
    z_axis = 1
    area_b = area[:, None, :, :]
    delp_coarse = weighted_block_average(delp, area_b, factor)
    delp_c_on_f = block_upsample(delp_coarse, factor)
    phalf_c = pressure_at_interface(delp_c_on_f, toa_pressure, z_axis)
    phalf_f = pressure_at_interface(delp, toa_pressure, z_axis)

    def cols(a):  # [tile, z, y, x] -> [ncol, z]
        return np.moveaxis(a, z_axis, -1).reshape(-1, a.shape[z_axis])

    out = {}
    for name, f in fields.items():
        r = mappm_fn(cols(phalf_f), cols(f), cols(phalf_c))
        nt, _, ny, nx = f.shape
        out[name] = np.moveaxis(r.reshape(nt, ny, nx, -1), -1, z_axis)
    pfull = pressure_at_midpoint_log(delp_c_on_f, toa_pressure, z_axis) if extrapolate else None
    masked = mask_weights(area, phalf_c, phalf_f, z_axis, pfull, extrapolate)
    return out, masked


# ---------------------------------------------------------------------------------------------
# Cell centres -> cell edges across the cube, and the edge-weighted pressure-level regrid
# ---------------------------------------------------------------------------------------------
# external/vcm/vcm/cubedsphere/xgcm.py:7-34 (FV3_FACE_CONNECTIONS, data): for tile t and axis a,
# ((left neighbour tile, its axis), (right neighbour tile, its axis)).
FV3_FACE_CONNECTIONS = {
    0: {"x": ((4, "y"), (1, "x")), "y": ((5, "y"), (2, "x"))},
    1: {"x": ((0, "x"), (3, "y")), "y": ((5, "x"), (2, "y"))},
    2: {"x": ((0, "y"), (3, "x")), "y": ((1, "y"), (4, "x"))},
    3: {"x": ((2, "x"), (5, "y")), "y": ((1, "x"), (4, "y"))},
    4: {"x": ((2, "y"), (5, "x")), "y": ((3, "y"), (0, "x"))},
    5: {"x": ((4, "x"), (1, "y")), "y": ((3, "x"), (0, "y"))},
}


def interp_center_to_outer(a, axis):
    """What ``xgcm.Grid.interp(da, axis)`` (xgcm 0.6.1, pinned in constraints.txt:316; not in the
    reference tree) computes for a cell-centred [tile, z, y, x] array on the FV3 cube
    (regridz.py:123-135): pad each tile with its neighbours' adjacent row -- the left neighbour's
    LAST line along its connecting axis, the right neighbour's FIRST one; a neighbour connected
    through its other axis contributes that line transposed and reversed -- then
    ``0.5 * (left + right)``.  Pinned by the u / v arrays of the reference's pressure-level
    regression fixtures (tests/test_oracle_coarsen.py), which touch all 12 cube edges."""
    a = np.asarray(a)
    outs = []
    for t in range(6):
        (ln, la), (rn, ra) = FV3_FACE_CONNECTIONS[t][axis]
        if axis == "x":
            left = a[ln][:, :, -1] if la == "x" else a[ln][:, -1, ::-1]
            right = a[rn][:, :, 0] if ra == "x" else a[rn][:, 0, ::-1]
            ext = np.concatenate([left[:, :, None], a[t], right[:, :, None]], axis=2)
            outs.append(0.5 * (ext[:, :, :-1] + ext[:, :, 1:]))
        else:
            left = a[ln][:, -1, :] if la == "y" else a[ln][:, ::-1, -1]
            right = a[rn][:, 0, :] if ra == "y" else a[rn][:, ::-1, 0]
            ext = np.concatenate([left[:, None, :], a[t], right[:, None, :]], axis=1)
            outs.append(0.5 * (ext[:, :-1, :] + ext[:, 1:, :]))
    return np.stack(outs).astype(a.dtype)


def regrid_to_edge_weighted_pressure(fields, delp, length, toa_pressure, factor, mappm_fn, edge="x",
                                     extrapolate=False):
    """regridz.py:81-146 + 149-197 for edge-valued fields [tile, z, y(+1), x(+1)] with edge
    lengths ``length`` [tile, y(+1), x(+1)].  Returns (dict of regridded fields, masked lengths)."""
    z_axis = 1
    delp_staggered = interp_center_to_outer(delp, "x" if edge == "y" else "y")
    delp_coarse = edge_weighted_block_average(delp_staggered, length[:, None], factor, edge)
    delp_c_on_f = block_upsample(delp_coarse, factor)
    phalf_c = pressure_at_interface(delp_c_on_f, toa_pressure, z_axis)
    phalf_f = pressure_at_interface(delp_staggered, toa_pressure, z_axis)

    def cols(a):
        return np.moveaxis(a, z_axis, -1).reshape(-1, a.shape[z_axis])

    out = {}
    for name, f in fields.items():
        r = mappm_fn(cols(phalf_f), cols(f), cols(phalf_c))
        nt, _, ny, nx = f.shape
        out[name] = np.moveaxis(r.reshape(nt, ny, nx, -1), -1, z_axis)
    pfull = pressure_at_midpoint_log(delp_c_on_f, toa_pressure, z_axis) if extrapolate else None
    masked = mask_weights(length, phalf_c, phalf_f, z_axis, pfull, extrapolate)
    return out, masked


# ---------------------------------------------------------------------------------------------
# humidity limiters (external/vcm/vcm/calc/thermo/non_negative_sphum.py:6-45, local.py:317-360)
# ---------------------------------------------------------------------------------------------
_HEAT_CAPACITY = 1004 - 287.05
_LV0 = 2.5e6


def non_negative_sphum(sphum, dq1, dq2, dt):
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = (-sphum) / (dt * dq2)
        ok = sphum + dq2 * dt >= 0
        return np.where(ok, dq1, ratio * dq1), np.where(ok, dq2, ratio * dq2)


def non_negative_sphum_mse_conserving(sphum, q2, dt, q1=None):
    q2_new = np.where(sphum + q2 * dt >= 0, q2, -sphum / dt)
    if q1 is None:
        return q2_new, None
    mse = _HEAT_CAPACITY * q1 + _LV0 * q2
    return q2_new, (mse - _LV0 * q2_new) / _HEAT_CAPACITY


# ---------------------------------------------------------------------------------------------
# blended pressure-level / model-level coarse-graining (coarsen_restarts.py:559-676)
# ---------------------------------------------------------------------------------------------
SIGMA_BLEND = 0.9


def surface_pressure_from_delp(delp, toa_pressure, z_axis):
    """vertically_dependent.py:189-208: delp.sum(z) + p_toa."""
    return np.asarray(delp).sum(axis=z_axis) + toa_pressure


def compute_blending_weights(blending_pressure, ps_coarse, pfull_coarse, z_axis):
    """coarsen_restarts.py:559-576; ``pfull_coarse`` has the z axis, the other two do not."""
    ps = np.expand_dims(ps_coarse, z_axis)
    pb = np.expand_dims(blending_pressure, z_axis)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = (ps - pfull_coarse) / (ps - pb)
    return np.where(pfull_coarse > pb, w, 1.0)


def blending_weights_agrid(delp, area, toa_pressure, factor):
    """coarsen_restarts.py:579-622 for delp [tile, z, y, x], area [tile, y, x]."""
    delp_c = weighted_block_average(delp, area[:, None], factor)
    pfull_c = pressure_at_midpoint_log(delp_c, toa_pressure, 1)
    ps = surface_pressure_from_delp(delp, toa_pressure, 1)
    ps_c = surface_pressure_from_delp(delp_c, toa_pressure, 1)
    pb = SIGMA_BLEND * block_coarsen(ps, factor, "min")
    return compute_blending_weights(pb, ps_c, pfull_c, 1)


def blending_weights_dgrid(delp, length, toa_pressure, factor, edge):
    """coarsen_restarts.py:625-661: the same on the cell edges the ``edge`` wind component lives on."""
    delp_e = interp_center_to_outer(delp, "x" if edge == "y" else "y")
    delp_ec = edge_weighted_block_average(delp_e, length[:, None], factor, edge)
    pfull_c = pressure_at_midpoint_log(delp_ec, toa_pressure, 1)
    ps = surface_pressure_from_delp(delp_e, toa_pressure, 1)
    ps_c = surface_pressure_from_delp(delp_ec, toa_pressure, 1)
    pb = SIGMA_BLEND * block_edge_coarsen(ps, factor, edge, "min")
    return compute_blending_weights(pb, ps_c, pfull_c, 1)


def blend(weights, pressure_level, model_level):
    """coarsen_restarts.py:664-676."""
    return weights * pressure_level + (1 - weights) * model_level
