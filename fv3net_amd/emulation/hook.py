"""``MicrophysicsHook``: applies the emulator to the Fortran state in place
(external/emulation/emulation/_emulate/microphysics.py:19-100)."""
import gc
from typing import Callable, MutableMapping

import numpy as np

FortranState = MutableMapping[str, np.ndarray]
Mask = Callable[[FortranState, FortranState], FortranState]


HOST_ENTRIES = ("model_time", "rank")  # (_emulate/microphysics.py:92-93 drops them from the model inputs)


def _is_field(name, value) -> bool:
    return (isinstance(value, np.ndarray) and value.ndim >= 1 and value.dtype.kind == "f" and name not in HOST_ENTRIES)


def always_emulator(state: FortranState, emulator: FortranState):
    return emulator


class MicrophysicsHook:
    """Object that applies a ML model to the fortran state."""

    def __init__(self, model: Callable[[FortranState], FortranState], mask: Mask = always_emulator,
                 garbage_collection_interval: int = 1000) -> None:
        self.name = "microphysics emulator"
        self.garbage_collection_interval = garbage_collection_interval
        self.mask = mask
        self._calls_since_last_collection = 0
        self.model = model

    def _maybe_garbage_collect(self):
        """A collection every ``garbage_collection_interval`` calls.  (The reference's counter logic,
        _emulate/microphysics.py:74-79, collects on every second call whatever the interval; a full ``gc.collect()`` with
        torch loaded costs 10-25 ms -- a hundred times the hook's device work at the column counts of one rank -- and
        has no effect on the results, so the evident intent is implemented instead.)"""
        self._calls_since_last_collection += 1
        if self._calls_since_last_collection >= self.garbage_collection_interval:
            gc.collect()
            self._calls_since_last_collection = 0

    def microphysics(self, state: FortranState) -> None:
        """Hook called from Fortran through call_py_fort.  ``state`` holds ``[feature, sample]``
        (or ``[sample]``) arrays plus scalar entries; it is updated in place with the
        emulator's ``[feature, sample]`` outputs.

        With a device-resident model (``HipEmulator``) the state is uploaded once, the network and
        every mask run on the device, and only the final outputs come back: one PCIe round trip per
        call instead of one per mask."""
        if getattr(self.model, "device_resident", False):
            from ..cubedsphere._device import download_all, on_device

            # only the floating-point fields go to the device: the bookkeeping entries (`model_time`, `rank`; integers and
            # scalars, whatever container call_py_fort hands them over in) stay on the host, where TimeMask reads them
            dev_state = {name: on_device(v) if _is_field(name, v) else v for name, v in state.items()}
            inputs = {name: v.t() if v.dim() == 2 else v for name, v in dev_state.items() if hasattr(v, "dim")}
            predictions = self.model(inputs)
            # (a model wrapped in tensor transforms returns its inputs along with its outputs, models.py:56-65: an entry that
            # IS the input array has nothing to write back)
            predictions = {name: t for name, t in predictions.items() if t is not inputs.get(name)}
            # numpy's .T: reverse all axes ([sample, z, class] logits -> [class, z, sample])
            model_outputs = {name: t.permute(*reversed(range(t.dim()))) for name, t in predictions.items()}
            model_outputs.update(self.mask(dev_state, model_outputs))
            state.update(download_all(model_outputs))
            self._maybe_garbage_collect()
            return
        inputs = {name: state[name].T for name in state if hasattr(state[name], "T")}
        predictions = self.model(inputs)
        predictions = {name: t for name, t in predictions.items() if t is not inputs.get(name)}
        # transpose back to FV3 conventions
        model_outputs = {name: np.asarray(tensor).T for name, tensor in predictions.items()}
        model_outputs.update(self.mask(state, model_outputs))
        state.update(model_outputs)
        self._maybe_garbage_collect()
