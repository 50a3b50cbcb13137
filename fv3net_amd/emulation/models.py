"""The emulator as a callable on the Fortran state, replacing
``emulation.models.ModelWithClassifier`` + ``transform_model``
(external/emulation/emulation/models.py:14-65) for models with the "dense" architecture
(``HipEmulator``) and the "dense-local" one (``HipLocalEmulator``: the reference's production gscond
regressor and its classifier).

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
from ..local_mlp import HybridRnnModel, HybridRnnSpec, LocalMlpModel, LocalMlpSpec, RnnModel, RnnSpec
from ..mlp import MlpModel, MlpModelSplitBf16, MlpSpec
from . import zhao_carr


class HipEmulator:
    """``emulator(inputs) -> outputs`` on dicts of ``[sample, feature]`` (or ``[sample]``) arrays,
    like the reference's model adapters.  Entries the network does not read (``rank``,
    ``model_time``, unused fields) are ignored."""

    device_resident = True  # MicrophysicsHook keeps the state and the masks on the device around this model
    _SPEC_FILENAME = "spec.yaml"
    _WEIGHTS_FILENAME = "weights.npz"
    # "fp32": the product kernel (fp32 MFMA).  "split-bf16": OPT-IN, experimental -- the same graph with the contraction on
    # the bf16 matrix cores, every operand split into three bf16 pieces (fp32-level accuracy, ~1.4x faster on the Zhao-Carr
    # network).  A network that kernel does not implement raises at construction, nothing falls back.
    ARITHMETIC_ENV = "FV3NET_AMD_EMULATOR_ARITHMETIC"

    def __init__(self, spec: MlpSpec, inputs_to_ignore: Sequence[str] = ("rank", "model_time"), arithmetic: str = None):
        self.spec = spec
        self.inputs_to_ignore = tuple(inputs_to_ignore)
        self.arithmetic = arithmetic or os.environ.get(self.ARITHMETIC_ENV, "fp32")
        if self.arithmetic not in ("fp32", "split-bf16"):
            raise ValueError(f"arithmetic must be 'fp32' or 'split-bf16', got {self.arithmetic!r}")
        self._model = None

    @property
    def model(self):
        if self._model is None:
            cls = MlpModelSplitBf16 if self.arithmetic == "split-bf16" else MlpModel
            self._model = cls(self.spec, device=compute_device())
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
        if self.arithmetic == "split-bf16":  # (that kernel reads float32 [feature, sample] rows with unit sample stride only)
            from .. import ops

            for name, t in sources.items():
                t = t.unsqueeze(0) if t.dim() == 1 else t
                if t.stride(1) != 1:
                    t = t.contiguous()
                sources[name] = t if t.dtype == torch.float32 else ops.cast(t, torch.float32)
            outs = self.model.predict(sources)
        else:
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


class HipLocalEmulator:
    """A "dense-local" (``LocalMlpSpec``) or RNN (``RnnSpec``: "rnn-v1-shared-weights", the production precpd
    architecture) model with the interface of ``HipEmulator``: dicts of ``[sample, feature]`` (or ``[sample]``)
    arrays in and out; multi-channel outputs (classifier logits) come back as ``[sample, feature, channel]``
    like the Keras model's."""

    device_resident = True
    _SPEC_FILENAME = HipEmulator._SPEC_FILENAME
    _WEIGHTS_FILENAME = HipEmulator._WEIGHTS_FILENAME

    def __init__(self, spec, inputs_to_ignore: Sequence[str] = ("rank", "model_time")):
        self.spec = spec
        self.inputs_to_ignore = tuple(inputs_to_ignore)
        self._model = None

    @property
    def model(self):
        if self._model is None:
            # (both follow FV3NET_AMD_EMULATOR_ARITHMETIC, like HipEmulator)
            cls = RnnModel if isinstance(self.spec, RnnSpec) else HybridRnnModel if isinstance(self.spec, HybridRnnSpec) else LocalMlpModel
            self._model = cls(self.spec, device=compute_device())
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
            else:
                a = np.asarray(a)
                # [sample, feature] views of call_py_fort's [feature, sample] arrays go up as they are
                t = on_device(a.T).t() if a.ndim == 2 and a.T.flags.c_contiguous else on_device(a)
            sources[name] = t.t() if t.dim() == 2 else t
        outs = self.model.predict(sources)
        result = {}
        for name, t in outs.items():
            t = t.permute(*reversed(range(t.dim())))  # [sample, feature(, channel)] view
            result[name] = t if on_gpu else t.cpu().numpy()
        return result

    def dump(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        meta, arrays = self.spec.to_arrays()
        np.savez(os.path.join(path, self._WEIGHTS_FILENAME), **arrays)
        with open(os.path.join(path, self._SPEC_FILENAME), "w") as f:
            yaml.safe_dump(meta, f)

    @classmethod
    def load(cls, path: str) -> "HipLocalEmulator":
        return load_emulator(path, expect=cls)


def load_emulator(path: str, expect=None):
    """Load a saved emulator directory; ``spec.yaml``'s ``architecture`` key ("dense" when absent,
    "dense-local") selects the class -- the counterpart of ``tf.keras.models.load_model`` in
    external/emulation/emulation/config.py:39-44."""
    with open(os.path.join(path, HipEmulator._SPEC_FILENAME)) as f:
        meta = yaml.safe_load(f)
    with np.load(os.path.join(path, HipEmulator._WEIGHTS_FILENAME), allow_pickle=False) as z:
        arrays = {k: z[k] for k in z.files}
    arch = meta.get("architecture", "dense")
    if arch == "dense-local":
        model = HipLocalEmulator(LocalMlpSpec.from_arrays(meta, arrays))
    elif arch in ("rnn-v1-shared-weights", "rnn-v1"):  # (both build Conv1D heads, architecture.py:483-506)
        model = HipLocalEmulator(RnnSpec.from_arrays(meta, arrays))
    elif arch == "rnn":  # HybridRNN (architecture.py:78-147)
        model = HipLocalEmulator(HybridRnnSpec.from_arrays(meta, arrays))
    elif arch in ("dense", "linear"):  # ("linear", architecture.py:285-302 with MLPBlock(depth=0): a dense model without hidden layers)
        model = HipEmulator(MlpSpec.from_arrays(meta, arrays))
    else:
        raise NotImplementedError(f"architecture {arch!r} is not implemented on the device "
                                  "(dense, linear, dense-local, rnn, rnn-v1, rnn-v1-shared-weights are)")
    if expect is not None and not isinstance(model, expect):
        raise TypeError(f"{path} holds a {type(model).__name__}, not a {expect.__name__}")
    return model


def _get_classify_output(logit_classes, one_hot_axis: int = 0) -> Dict[str, object]:
    """One-hot decode of the class logits (external/emulation/emulation/zhao_carr.py:193-198): every class
    whose logit equals the maximum is hot; plus ``nontrivial_tendency``.  Device arrays stay on the device."""
    names = zhao_carr.CLASS_NAMES  # sorted
    if isinstance(logit_classes, torch.Tensor) and logit_classes.is_cuda:
        from .. import _lib
        from ..ops import _ptr, _require_device, _stream

        moved = torch.movedim(logit_classes, one_hot_axis, 0)  # [class, *plane]
        # the plane's dims in memory order: a [sample, z] view of [z, sample] memory is used as it lies
        order = sorted(range(1, moved.dim()), key=lambda d: -moved.stride(d))
        logits = moved.permute(0, *order)
        if logits.dtype not in (torch.float32, torch.float64):
            logits = logits.to(torch.float32)
        logits = logits.contiguous()
        n_class = int(logits.shape[0])
        if n_class != len(names):
            raise ValueError(f"expected {len(names)} classes along axis {one_hot_axis}, got {n_class}")
        dev = _require_device(logits)
        onehot = torch.empty(logits.shape, dtype=torch.uint8, device=dev)
        both = torch.empty(logits.shape[1:], dtype=torch.uint8, device=dev)
        _lib.call_on(dev, "fv3hip_classify_onehot", _ptr(logits), _lib.F64 if logits.dtype == torch.float64 else _lib.F32, n_class,
                  both.numel(), _ptr(onehot), _ptr(both), names.index(zhao_carr.POSITIVE_TENDENCY),
                  names.index(zhao_carr.NEGATIVE_TENDENCY), _stream(dev))
        inverse = [order.index(d) for d in range(1, moved.dim())]  # back to the plane's own dim order (views)
        d = {name: onehot[i].view(torch.bool).permute(*inverse) for i, name in enumerate(names)}
        d["nontrivial_tendency"] = both.view(torch.bool).permute(*inverse)
        return d
    logits = np.asarray(logit_classes)
    one_hot = logits == np.max(logits, axis=one_hot_axis, keepdims=True)
    d = {name: np.take(one_hot, i, one_hot_axis) for i, name in enumerate(names)}
    d["nontrivial_tendency"] = d[zhao_carr.POSITIVE_TENDENCY] | d[zhao_carr.NEGATIVE_TENDENCY]
    return d


class ModelWithClassifier:
    """``model`` preceded by an optional ``classifier`` whose decoded classes are fed to the model and
    returned with its outputs (external/emulation/emulation/models.py:14-53).  Both are callables on
    dicts of ``[sample, feature]`` arrays (``HipEmulator`` / ``HipLocalEmulator``)."""

    def __init__(self, model, classifier=None, class_key: str = "gscond_classes", batch_size: int = 1024,
                 inputs_to_ignore: Sequence[str] = ("rank", "model_time")):
        self.model = model
        self.classifier = classifier
        self._class_key = class_key
        self._batch_size = batch_size  # accepted and ignored: a call is one launch over all columns
        self.inputs_to_ignore = inputs_to_ignore

    @property
    def device_resident(self) -> bool:
        return all(getattr(m, "device_resident", False) for m in (self.model, self.classifier) if m is not None)

    def __call__(self, state):
        state = {k: v for k, v in state.items() if k not in self.inputs_to_ignore}
        if self.classifier is not None:
            classifier_outputs = dict(self.classifier(state))
            classifier_outputs.update(_get_classify_output(classifier_outputs[self._class_key], one_hot_axis=-1))
        else:
            classifier_outputs = {}
        model_outputs = dict(self.model({**classifier_outputs, **state}))
        model_outputs.update(classifier_outputs)
        return model_outputs


def combine_classifier_and_regressor(classifier, regressor, batch_size: int = 1024) -> ModelWithClassifier:
    """external/emulation/emulation/models.py (used at config.py:146-148)."""
    return ModelWithClassifier(regressor, classifier, batch_size=batch_size)


def transform_model(model, transform):
    """external/emulation/emulation/models.py:56-65."""

    def combined(x):
        x_transformed = transform.forward(x)
        x_transformed.update(model(x_transformed))
        return transform.backward(x_transformed)

    # (the hook keeps the state on the device around a device-resident model; the transforms work on device tensors too)
    combined.device_resident = getattr(model, "device_resident", False)
    return combined
