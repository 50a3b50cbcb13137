"""The emulator as a callable on the Fortran state, replacing
``emulation.models.ModelWithClassifier`` + ``transform_model``
(external/emulation/emulation/models.py:14-65) for regressors with the "dense" architecture.

The reference wraps a Keras model: log / difference transforms forward, ``model.predict`` in
Python-level mini-batches of ``batch_size`` columns, transforms backward.  Here the transforms,
the normalisation, the network and the residual outputs are one fused HIP launch over all the
columns of the call.
"""
import os
from typing import Dict, Mapping, Sequence

import numpy as np
import torch
import yaml

from ..cubedsphere._device import compute_device, on_device
from ..mlp import MlpModel, MlpSpec


class HipEmulator:
    """``emulator(inputs) -> outputs`` on dicts of ``[sample, feature]`` (or ``[sample]``) arrays,
    like the reference's model adapters.  Entries the network does not read (``rank``,
    ``model_time``, unused fields) are ignored."""

    device_resident = True  # MicrophysicsHook keeps the state and the masks on the device around this model
    _SPEC_FILENAME = "spec.yaml"
    _WEIGHTS_FILENAME = "weights.npz"

    def __init__(self, spec: MlpSpec, inputs_to_ignore: Sequence[str] = ("rank", "model_time")):
        self.spec = spec
        self.inputs_to_ignore = tuple(inputs_to_ignore)
        self._model = None

    @property
    def model(self) -> MlpModel:
        if self._model is None:
            self._model = MlpModel(self.spec, device=compute_device())
        return self._model

    @property
    def input_variables(self):
        return self.spec.sources

    @property
    def output_variables(self):
        return self.spec.output_names

    def __call__(self, state: Mapping[str, np.ndarray]) -> Dict[str, np.ndarray]:
        sources = {}
        on_gpu = False
        for name in self.spec.sources:
            a = state[name]
            if isinstance(a, torch.Tensor):
                on_gpu = on_gpu or a.is_cuda
                t = on_device(a)
                sources[name] = t.t() if t.dim() == 2 else t
                continue
            a = np.asarray(a)
            if a.ndim == 2 and a.T.flags.c_contiguous:
                # a transposed view of call_py_fort's [feature, sample] array: upload it as it is
                sources[name] = on_device(a.T)
            elif a.ndim == 2:
                sources[name] = on_device(np.ascontiguousarray(a)).t()
            else:
                sources[name] = on_device(a)
        outs = self.model.predict(sources, layout="feature_sample")
        result = {}
        for name, t in outs.items():
            if on_gpu:
                result[name] = t.t()
            else:
                result[name] = t.cpu().numpy().T  # [sample, feature] view of the [feature, sample] buffer
        return result

    def dump(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        meta, arrays = self.spec.to_arrays()
        np.savez(os.path.join(path, self._WEIGHTS_FILENAME), **arrays)
        with open(os.path.join(path, self._SPEC_FILENAME), "w") as f:
            yaml.safe_dump(meta, f)

    @classmethod
    def load(cls, path: str) -> "HipEmulator":
        with open(os.path.join(path, cls._SPEC_FILENAME)) as f:
            meta = yaml.safe_load(f)
        with np.load(os.path.join(path, cls._WEIGHTS_FILENAME), allow_pickle=False) as z:
            arrays = {k: z[k] for k in z.files}
        return cls(MlpSpec.from_arrays(meta, arrays))
