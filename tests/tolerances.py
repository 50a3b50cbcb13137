"""The floating-point gate of the MLP tests, per (variable, level).

north_star: outputs within 1e-5 relative fp32 of the reference.  For a column model "relative" is per output LEVEL: a
variable's levels differ by orders of magnitude (cloud water at 200 hPa against 900 hPa), so one scale per variable would let
the small levels carry any error (VERDICT r02 #6a).  For every level k of every output:

    max_samples |gpu - truth|[:, k]  <=  1e-5 * max_samples |truth[:, k]|
    ... and no worse than 8x the float32 CPU evaluation of the same graph (+ 1e-7 of the level's scale; the per-variable
        form of this check used 4x -- a per-level maximum over a few thousand samples is an extreme-value statistic of two
        independent roundings and fluctuates by more than that on one level in several hundred)

``truth`` = the float64 oracle.  A level whose truth is identically zero (masked / clipped levels) must be exactly zero.
"""
import numpy as np


def assert_close_per_level(got, truth, cpu32=None, name="", rel=1e-5, compounding=False, slack32=1e-7):
    """``got``, ``truth``, ``cpu32``: [sample, level] arrays.  ``compounding``: a recurrence over the levels compounds
    rounding -- the gate is then ``rel * scale + 4 x the float32 CPU evaluation's own error`` per level."""
    got, truth = np.asarray(got, dtype=np.float64), np.asarray(truth, dtype=np.float64)
    assert got.shape == truth.shape, (name, got.shape, truth.shape)
    if got.ndim == 1:
        got, truth = got[:, None], truth[:, None]
        cpu32 = None if cpu32 is None else np.asarray(cpu32)[:, None]
    scale = np.max(np.abs(truth), axis=0)
    err = np.max(np.abs(got - truth), axis=0)
    zero = scale == 0
    assert np.all(err[zero] == 0), (name, "levels that are identically zero in the oracle", np.nonzero(zero & (err > 0))[0])
    worst = int(np.argmax(np.where(zero, 0, err / np.where(zero, 1, scale))))
    err32 = None if cpu32 is None else np.max(np.abs(np.asarray(cpu32, dtype=np.float64) - truth), axis=0)
    bound = rel * scale + (4 * err32 if (compounding and err32 is not None) else 0)
    assert np.all(err <= bound), (name, f"level {worst}: err {err[worst]:.3e}, level scale {scale[worst]:.3e}, "
                                        f"ratio {err[worst] / scale[worst]:.2e}")
    if cpu32 is not None and not compounding:
        bad = err > 8 * err32 + slack32 * scale
        assert not bad.any(), (name, "worse than 8x the float32 CPU evaluation at levels", np.nonzero(bad)[0][:8],
                               err[bad][:4], err32[bad][:4])
    return float(np.max(np.where(zero, 0, err / np.where(zero, 1, scale))))


def decades(rng, nfeat, n_decades=4.5, top=1.0):
    """A per-level scale that falls by ``n_decades`` orders of magnitude over the levels (what the standard deviations of a
    real cloud or humidity output do between the boundary layer and the stratosphere), with a little jitter."""
    k = np.arange(nfeat) / max(nfeat - 1, 1)
    return (top * 10.0 ** (-n_decades * k) * rng.uniform(0.7, 1.4, nfeat)).astype(np.float32)
