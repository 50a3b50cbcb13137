"""The online schedule of the hooks (external/emulation/emulation/_emulate/microphysics.py:23-47, _time.py:6-12):
``IntervalSchedule`` selects the Fortran physics (weight 1) in the first half of every ``period`` counted from
``initial_time`` and the emulator (weight 0) in the second half; ``TimeMask`` blends the two with that weight,
``state * alpha + emulator * (1 - alpha)`` over the keys they share -- on the device for device arrays.

The reference measures time with ``cftime.DatetimeJulian`` (cftime is not part of its tree and not installed): only
differences enter, so a Julian-calendar day number (every fourth year a leap year) is all that is needed."""
import dataclasses
import datetime
from typing import Callable, Sequence, Union

TimeLike = Union[datetime.datetime, Sequence[int]]


def _julian_day_number(year: int, month: int, day: int) -> int:
    """Day number in the Julian calendar (the standard formula; 1 January 4713 BC is day 0)."""
    a = (14 - month) // 12
    y = year + 4800 - a
    m = month + 12 * a - 3
    return day + (153 * m + 2) // 5 + 365 * y + y // 4 - 32083


def julian_seconds(time: TimeLike) -> int:
    """Seconds since the Julian-calendar epoch of a datetime-like (``year month day hour minute second`` attributes, e.g. a
    ``datetime.datetime`` or a ``cftime.DatetimeJulian``) or a (year, month, day, hour, minute, second) sequence."""
    if hasattr(time, "year"):
        y, mo, d, h, mi, s = time.year, time.month, time.day, time.hour, time.minute, time.second
    else:
        y, mo, d, h, mi, s = (list(time) + [0, 0, 0])[:6]
    return ((_julian_day_number(int(y), int(mo), int(d)) * 24 + int(h)) * 60 + int(mi)) * 60 + int(s)


def translate_time(time: Sequence[int]):
    """The model's 6-integer time -> (year, month, day, hour, minute): fields 0, 1, 2, 4, 5 of the tuple the Fortran
    passes (_time.py:6-12; field 3 is not used there either)."""
    return (time[0], time[1], time[2], time[4], time[5], 0)


@dataclasses.dataclass
class IntervalSchedule:
    """Select the left value (1.0) in the first half of an interval of ``period`` counted from ``initial_time``."""

    period: datetime.timedelta
    initial_time: TimeLike

    def __call__(self, time: TimeLike) -> float:
        elapsed = julian_seconds(time) - julian_seconds(self.initial_time)
        fraction_of_interval = (elapsed / self.period.total_seconds()) % 1
        return 1.0 if fraction_of_interval < 0.5 else 0.0

    @staticmethod
    def from_dict(d) -> "IntervalSchedule":
        period = d["period"]
        if not isinstance(period, datetime.timedelta):
            period = datetime.timedelta(seconds=period)  # the reference's type hook (config.py:252-258)
        return IntervalSchedule(period, d["initial_time"])


@dataclasses.dataclass
class TimeMask:
    schedule: Callable[[TimeLike], float]

    def __call__(self, state, emulator):
        model_time = state["model_time"]
        if hasattr(model_time, "is_cuda"):  # (a caller that uploaded it anyway: one transfer, not one per element)
            model_time = model_time.cpu().tolist()
        alpha = self.schedule(translate_time([int(x) for x in model_time]))
        common_keys = set(state) & set(emulator)
        return {key: _blend(state[key], emulator[key], alpha) for key in common_keys}


def _blend(left, right, alpha: float):
    import torch

    if isinstance(left, torch.Tensor) or isinstance(right, torch.Tensor):
        from .. import ops
        from ..cubedsphere._device import on_device

        a, b = on_device(left), on_device(right)
        if a.dtype != b.dtype:  # numpy promotion
            a, b = a.double(), b.double()
        # alpha is exactly 0 or 1 with an IntervalSchedule: the selected side as it is, without a pass over the data.
        # (One difference from `x * 1 + y * 0`: a NaN or infinity of the UNSELECTED side does not leak into the result.)
        if alpha == 1.0 or alpha == 0.0:
            return (a if alpha == 1.0 else b).contiguous()
        return ops.ew("add", ops.ew("mul_s", a.contiguous(), scalar=alpha), ops.ew("mul_s", b.contiguous(), scalar=1 - alpha))
    return left * alpha + right * (1 - alpha)
