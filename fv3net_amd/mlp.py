"""Column MLP: the network description shared by the fv3fit-style dense predictor and the
Zhao-Carr microphysics emulator, and its device handle.

``MlpSpec`` is a plain description of the fused predict graph (see ``fv3hip_mlp_desc_t`` in
``include/fv3hip.h`` for the reference lines each field restates).  ``MlpModel`` uploads it
once (``fv3hip_mlp_create``) and runs ``fv3hip_mlp_predict`` on device arrays.
"""
import ctypes
import dataclasses
from typing import Dict, List, Mapping, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .ops import _require_device, _stream


@dataclasses.dataclass
class InputSpec:
    """One network input variable: features [start, start+nfeat) of array ``source``,
    optionally log-transformed, then ``(x - center) / scale``."""

    source: str
    nfeat: int
    start: int = 0
    transform: str = "none"  # "none" | "log"
    eps: float = 0.0
    center: Optional[np.ndarray] = None  # [nfeat]; None = 0
    scale: Optional[np.ndarray] = None   # [nfeat]; None = 1 (already includes any epsilon)


@dataclasses.dataclass
class OutputSpec:
    """One output head: ``y = yhat * scale + center``, limited to [min, max], times ``mask``."""

    name: str
    nfeat: int
    scale: Optional[np.ndarray] = None
    center: Optional[np.ndarray] = None
    min: Optional[float] = None
    max: Optional[float] = None
    mask: Optional[np.ndarray] = None  # [nfeat] of 0/1


@dataclasses.dataclass
class ResidualSpec:
    """Derived output ``name = source + output`` (fv3fit Difference.backward)."""

    name: str
    source: str
    output: str


@dataclasses.dataclass
class MlpSpec:
    inputs: List[InputSpec]
    hidden_kernels: List[np.ndarray]  # Keras layout [in, out]
    hidden_biases: List[np.ndarray]
    outputs: List[OutputSpec]
    out_kernel: np.ndarray            # [width, sum(out nfeat)]
    out_bias: np.ndarray
    residuals: List[ResidualSpec] = dataclasses.field(default_factory=list)
    activation: str = "relu"
    # name under which the last hidden layer's activations ([width] features) are returned as an output of
    # their own (``outputs`` may then be empty): the cell of a recurrent layer is such a model
    hidden_output: Optional[str] = None

    @property
    def sources(self) -> List[str]:
        seen: List[str] = []
        for name in [i.source for i in self.inputs] + [r.source for r in self.residuals]:
            if name not in seen:
                seen.append(name)
        return seen

    @property
    def n_in_features(self) -> int:
        return sum(i.nfeat for i in self.inputs)

    @property
    def n_out_features(self) -> int:
        return sum(o.nfeat for o in self.outputs)

    @property
    def width(self) -> int:
        """Width of the hidden layers; a network without hidden layers ("linear") feeds its inputs to the output layer."""
        return int(self.hidden_kernels[0].shape[1]) if self.hidden_kernels else int(self.out_kernel.shape[0])

    @property
    def output_names(self) -> List[str]:
        return [o.name for o in self.outputs] + [r.name for r in self.residuals] + ([self.hidden_output] if self.hidden_output else [])

    def source_nfeat(self) -> Dict[str, int]:
        """Minimum number of features each source array must have."""
        need: Dict[str, int] = {}
        for i in self.inputs:
            need[i.source] = max(need.get(i.source, 0), i.start + i.nfeat)
        outs = {o.name: o.nfeat for o in self.outputs}
        for r in self.residuals:
            need[r.source] = max(need.get(r.source, 0), outs[r.output])
        return need

    def validate(self):
        k = self.n_in_features
        if len(self.hidden_kernels) != len(self.hidden_biases):
            raise ValueError("hidden_kernels and hidden_biases differ in length")
        if self.activation not in ("relu", "linear"):
            raise ValueError(f"activation must be 'relu' or 'linear', got {self.activation!r}")
        if self.hidden_output and not self.hidden_kernels:
            raise ValueError("hidden_output needs a hidden layer")
        w = self.width
        if self.hidden_kernels and tuple(self.hidden_kernels[0].shape) != (k, w):
            raise ValueError(f"first kernel has shape {self.hidden_kernels[0].shape}, expected {(k, w)}")
        if tuple(self.hidden_biases[0].shape if self.hidden_biases else (w,)) != (w,):
            raise ValueError(f"first bias has shape {self.hidden_biases[0].shape}, expected {(w,)}")
        for kern, b in zip(self.hidden_kernels[1:], self.hidden_biases[1:]):
            if tuple(kern.shape) != (w, w) or tuple(b.shape) != (w,):
                raise ValueError("hidden layers must all have the same width")
        f = self.n_out_features
        if tuple(self.out_kernel.shape) != (w, f) or tuple(self.out_bias.shape) != (f,):
            raise ValueError(f"output kernel has shape {self.out_kernel.shape}, expected {(w, f)}")
        names = [o.name for o in self.outputs]
        for r in self.residuals:
            if r.output not in names:
                raise ValueError(f"residual {r.name!r} refers to unknown output {r.output!r}")

    # -- flat npz (de)serialisation: the artifact the predictors dump -----------------------
    def to_arrays(self) -> Tuple[dict, Dict[str, np.ndarray]]:
        meta = {
            "activation": self.activation,
            "inputs": [
                {"source": i.source, "nfeat": i.nfeat, "start": i.start, "transform": i.transform,
                 "eps": float(i.eps)} for i in self.inputs
            ],
            "outputs": [
                {"name": o.name, "nfeat": o.nfeat, "min": o.min, "max": o.max} for o in self.outputs
            ],
            "residuals": [dataclasses.asdict(r) for r in self.residuals],
            "n_hidden": len(self.hidden_kernels),
            "hidden_output": self.hidden_output,
        }
        arrays: Dict[str, np.ndarray] = {}
        for n, i in enumerate(self.inputs):
            if i.center is not None:
                arrays[f"in{n}_center"] = np.asarray(i.center, np.float32)
            if i.scale is not None:
                arrays[f"in{n}_scale"] = np.asarray(i.scale, np.float32)
        for n, (kern, b) in enumerate(zip(self.hidden_kernels, self.hidden_biases)):
            arrays[f"hidden{n}_kernel"] = np.asarray(kern, np.float32)
            arrays[f"hidden{n}_bias"] = np.asarray(b, np.float32)
        arrays["out_kernel"] = np.asarray(self.out_kernel, np.float32)
        arrays["out_bias"] = np.asarray(self.out_bias, np.float32)
        for n, o in enumerate(self.outputs):
            for key in ("scale", "center", "mask"):
                val = getattr(o, key)
                if val is not None:
                    arrays[f"out{n}_{key}"] = np.asarray(val, np.float32)
        return meta, arrays

    @classmethod
    def from_arrays(cls, meta: Mapping, arrays: Mapping[str, np.ndarray]) -> "MlpSpec":
        inputs = [
            InputSpec(
                source=m["source"], nfeat=int(m["nfeat"]), start=int(m.get("start", 0)),
                transform=m.get("transform", "none"), eps=float(m.get("eps", 0.0)),
                center=arrays.get(f"in{n}_center"), scale=arrays.get(f"in{n}_scale"),
            )
            for n, m in enumerate(meta["inputs"])
        ]
        outputs = [
            OutputSpec(
                name=m["name"], nfeat=int(m["nfeat"]), min=m.get("min"), max=m.get("max"),
                scale=arrays.get(f"out{n}_scale"), center=arrays.get(f"out{n}_center"),
                mask=arrays.get(f"out{n}_mask"),
            )
            for n, m in enumerate(meta["outputs"])
        ]
        nh = int(meta["n_hidden"])
        return cls(
            inputs=inputs,
            hidden_kernels=[np.asarray(arrays[f"hidden{n}_kernel"]) for n in range(nh)],
            hidden_biases=[np.asarray(arrays[f"hidden{n}_bias"]) for n in range(nh)],
            outputs=outputs,
            out_kernel=np.asarray(arrays["out_kernel"]),
            out_bias=np.asarray(arrays["out_bias"]),
            residuals=[ResidualSpec(**r) for r in meta.get("residuals", [])],
            activation=meta.get("activation", "relu"),
            hidden_output=meta.get("hidden_output"),
        )


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _fptr(a: Optional[np.ndarray]):
    if a is None:
        return ctypes.POINTER(ctypes.c_float)()
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _iptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


# Default of ``MlpModel(small_limit=...)`` for models built without saying (the emulators build theirs internally): None =
# the library's rule.  Tests set it to 0 / a large number to pin one kernel or the other.
DEFAULT_SMALL_LIMIT: Optional[int] = None


class MlpModel:
    """Device handle of a fused MLP (``fv3hip_mlp_t``)."""

    def __init__(self, spec: MlpSpec, device="cuda", small_limit: Optional[int] = None):
        """``small_limit``: calls of at most that many samples run on the feature-split kernel for small sample counts
        (``fv3hip_mlp_set_small_limit``); None = the library's rule (while the 128-sample tiles would leave CUs idle),
        0 = never."""
        spec.validate()
        self.spec = spec
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MlpModel needs a 'cuda' (ROCm) device; there is no CPU fallback")
        with torch.cuda.device(self.device):
            _require_device(torch.empty(1, device=self.device))
            self._handle = self._create(spec)
        self.flops_per_sample = int(_lib.load().fv3hip_mlp_flops_per_sample(self._handle))
        if small_limit is None:
            small_limit = DEFAULT_SMALL_LIMIT
        if small_limit is not None:
            self.set_small_limit(small_limit)

    def set_small_limit(self, max_samples: Optional[int]) -> None:
        _lib.call("fv3hip_mlp_set_small_limit", self._handle, -1 if max_samples is None else int(max_samples))

    @staticmethod
    def _create(spec: MlpSpec, entry_point: str = "fv3hip_mlp_create"):
        sources = spec.sources
        keep = []  # keep numpy buffers alive during the call

        def arr_i(vals):
            a = np.ascontiguousarray(np.asarray(vals, dtype=np.int32))
            keep.append(a)
            return a

        def arr_f(vals):
            a = _f32(vals)
            keep.append(a)
            return a

        in_center = np.concatenate(
            [np.zeros(i.nfeat, np.float32) if i.center is None else np.broadcast_to(_f32(i.center), (i.nfeat,))
             for i in spec.inputs]
        )
        in_scale = np.concatenate(
            [np.ones(i.nfeat, np.float32) if i.scale is None else np.broadcast_to(_f32(i.scale), (i.nfeat,))
             for i in spec.inputs]
        )
        transforms = {"none": _lib.TRANSFORM_NONE, "log": _lib.TRANSFORM_LOG}
        F = spec.n_out_features

        def per_feature(attr, default):
            parts = []
            for o in spec.outputs:
                v = getattr(o, attr)
                parts.append(
                    np.full(o.nfeat, default, np.float32) if v is None
                    else np.broadcast_to(_f32(v), (o.nfeat,))
                )
            return np.concatenate(parts) if parts else np.zeros(0, np.float32)

        any_min = any(o.min is not None for o in spec.outputs)
        any_max = any(o.max is not None for o in spec.outputs)
        any_mask = any(o.mask is not None for o in spec.outputs)
        hk = [arr_f(k) for k in spec.hidden_kernels]
        hb = [arr_f(b) for b in spec.hidden_biases]
        PF = ctypes.POINTER(ctypes.c_float)
        hk_ptrs = (PF * len(hk))(*[_fptr(k) for k in hk])
        hb_ptrs = (PF * len(hb))(*[_fptr(b) for b in hb])
        out_names = [o.name for o in spec.outputs]

        desc = _lib.MlpDesc()
        desc.n_sources = len(sources)
        desc.n_inputs = len(spec.inputs)
        desc.in_source = _iptr(arr_i([sources.index(i.source) for i in spec.inputs]))
        desc.in_feat_start = _iptr(arr_i([i.start for i in spec.inputs]))
        desc.in_nfeat = _iptr(arr_i([i.nfeat for i in spec.inputs]))
        desc.in_transform = _iptr(arr_i([transforms[i.transform] for i in spec.inputs]))
        desc.in_eps = _fptr(arr_f([i.eps for i in spec.inputs]))
        desc.in_center = _fptr(arr_f(in_center))
        desc.in_scale = _fptr(arr_f(in_scale))
        desc.n_hidden = len(hk)
        desc.width = spec.width
        desc.hidden_activation = {"relu": _lib.ACT_RELU, "linear": _lib.ACT_LINEAR}[spec.activation]
        desc.hidden_kernels = ctypes.cast(hk_ptrs, ctypes.POINTER(PF))
        desc.hidden_biases = ctypes.cast(hb_ptrs, ctypes.POINTER(PF))
        desc.n_outputs = len(spec.outputs)
        desc.out_nfeat = _iptr(arr_i([o.nfeat for o in spec.outputs]))
        desc.out_kernel = _fptr(arr_f(spec.out_kernel))
        desc.out_bias = _fptr(arr_f(spec.out_bias))
        desc.out_scale = _fptr(arr_f(per_feature("scale", 1.0)))
        desc.out_center = _fptr(arr_f(per_feature("center", 0.0)))
        if any_min:
            desc.out_min = _fptr(arr_f(np.concatenate(
                [np.full(o.nfeat, -np.inf if o.min is None else o.min, np.float32) for o in spec.outputs])))
        if any_max:
            desc.out_max = _fptr(arr_f(np.concatenate(
                [np.full(o.nfeat, np.inf if o.max is None else o.max, np.float32) for o in spec.outputs])))
        if any_mask:
            desc.out_mask = _fptr(arr_f(per_feature("mask", 1.0)))
        desc.hidden_output = 1 if spec.hidden_output else 0
        desc.n_residual = len(spec.residuals)
        if spec.residuals:
            desc.res_source = _iptr(arr_i([sources.index(r.source) for r in spec.residuals]))
            desc.res_output = _iptr(arr_i([out_names.index(r.output) for r in spec.residuals]))
        assert F == sum(o.nfeat for o in spec.outputs)
        handle = ctypes.c_void_p()
        _lib.call(entry_point, ctypes.byref(desc), ctypes.byref(handle))
        return handle

    def predict(
        self,
        sources: Mapping[str, torch.Tensor],
        layout: str = "feature_sample",
        out_dtype: torch.dtype = torch.float32,
        out: Optional[Mapping[str, torch.Tensor]] = None,
    ) -> Dict[str, torch.Tensor]:
        """Run the network on 2-D device arrays.

        ``layout='feature_sample'``: every source is ``[feature, sample]`` (or ``[sample]`` for a
        single-feature variable) -- the model's native [z, (y, x)] arrays and call_py_fort's
        arrays; outputs come back as ``[feature, sample]``.
        ``layout='sample_feature'``: ``[sample, feature]`` in and out (what ``stack`` produces).
        Arbitrary strides are honoured, nothing is copied.  ``out``: preallocated output arrays (all of
        ``spec.output_names``, float32 or float64 alike) to write into instead of allocating.
        """
        spec = self.spec
        names = spec.sources
        need = spec.source_nfeat()
        tensors = []
        n_samples = None
        for name in names:
            t = sources[name]
            if t.dim() == 1:
                t = t.unsqueeze(0) if layout == "feature_sample" else t.unsqueeze(1)
            if t.dim() != 2:
                raise ValueError(f"source {name!r} must be 1-D or 2-D, got shape {tuple(t.shape)}")
            nf, ns = (t.shape[0], t.shape[1]) if layout == "feature_sample" else (t.shape[1], t.shape[0])
            if nf < need[name]:
                raise ValueError(f"source {name!r} has {nf} features, the model needs {need[name]}")
            if n_samples is None:
                n_samples = int(ns)
            elif int(ns) != n_samples:
                raise ValueError("sources differ in their number of samples")
            tensors.append(t)
        dev = _require_device(*tensors)
        dtypes = {t.dtype for t in tensors}
        if dtypes - {torch.float32, torch.float64}:
            raise TypeError(f"sources must be float32 or float64, got {dtypes}")
        if len(dtypes) > 1:  # the kernel wants one source dtype per call
            tensors = [t.to(torch.float64) for t in tensors]
        src_code = _lib.F64 if tensors[0].dtype == torch.float64 else _lib.F32
        fs_ax, ss_ax = (0, 1) if layout == "feature_sample" else (1, 0)

        outs: Dict[str, torch.Tensor] = {}
        nfeat = {o.name: o.nfeat for o in spec.outputs}
        for r in spec.residuals:
            nfeat[r.name] = nfeat[r.output]
        if spec.hidden_output:
            nfeat[spec.hidden_output] = spec.width
        out_list = []
        for name in spec.output_names:
            shape = (nfeat[name], n_samples) if layout == "feature_sample" else (n_samples, nfeat[name])
            if out is not None:
                t = out[name]
                if tuple(t.shape) != shape or t.device != dev:
                    raise ValueError(f"out[{name!r}] must have shape {shape} on {dev}")
                out_dtype = t.dtype
            else:
                t = torch.empty(shape, dtype=out_dtype, device=dev)
            outs[name] = t
            out_list.append(t)

        if len({t.dtype for t in out_list}) > 1 or out_dtype not in (torch.float32, torch.float64):
            raise TypeError("outputs must all be float32 or all float64")
        n_src, n_out = len(tensors), len(out_list)
        src_ptrs = (ctypes.c_void_p * n_src)(*[t.data_ptr() for t in tensors])
        src_dt = (ctypes.c_int * n_src)(*([src_code] * n_src))
        src_fs = (ctypes.c_int64 * n_src)(*[t.stride(fs_ax) for t in tensors])
        src_ss = (ctypes.c_int64 * n_src)(*[t.stride(ss_ax) for t in tensors])
        out_ptrs = (ctypes.c_void_p * n_out)(*[t.data_ptr() for t in out_list])
        out_fs = (ctypes.c_int64 * n_out)(*[t.stride(fs_ax) for t in out_list])
        out_ss = (ctypes.c_int64 * n_out)(*[t.stride(ss_ax) for t in out_list])
        _lib.call_on(dev,
            "fv3hip_mlp_predict", self._handle, src_ptrs, src_dt, src_fs, src_ss, n_samples, out_ptrs,
            _lib.F64 if out_dtype == torch.float64 else _lib.F32, out_fs, out_ss, _stream(dev),
        )
        return outs

    @property
    def last_variant(self) -> str:
        """Kernel instantiation and epilogue flavour of the last ``predict`` (``fv3hip_mlp_last_variant``)."""
        return _lib.load().fv3hip_mlp_last_variant(self._handle).decode()

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                _lib.load().fv3hip_mlp_destroy(h)
            except Exception:
                pass


class MlpModelSplitBf16:
    """EXPERIMENTAL: the same network through ``fv3hip_mlp3_*`` -- the contraction on the bf16 matrix cores with every fp32
    operand split into three bf16 pieces (csrc/mlp_bf16x3.hip).  float32 ``[feature, sample]`` sources
    with unit sample stride only; ``MlpModel`` (fp32 MFMA) is the product path."""

    def __init__(self, spec: MlpSpec, device="cuda"):
        spec.validate()
        self.spec = spec
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MlpModelSplitBf16 needs a 'cuda' (ROCm) device; there is no CPU fallback")
        with torch.cuda.device(self.device):
            _require_device(torch.empty(1, device=self.device))
            self._handle = MlpModel._create(spec, "fv3hip_mlp3_create")
        self.flops_per_sample = int(_lib.load().fv3hip_mlp3_flops_per_sample(self._handle))
        self.last_variant = "mlp3_kernel"

    def predict(self, sources: Mapping[str, torch.Tensor], out: Optional[Mapping[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """``sources``: name -> float32 ``[feature, sample]`` device array with unit sample stride.  ``out``: preallocated
        float32 ``[feature, sample]`` arrays (unit sample stride) for any of the outputs -- the hidden output included, as
        ``MlpModel.predict`` takes them; the others are allocated."""
        spec = self.spec
        need = spec.source_nfeat()
        tensors = []
        n = None
        # the C entry point sees pointers, strides and n only: a short or mismatched source would be read past its end
        for name in spec.sources:
            t = sources[name]
            t = t.unsqueeze(0) if t.dim() == 1 else t
            if t.dim() != 2:
                raise ValueError(f"source {name!r} must be 1-D or 2-D, got shape {tuple(t.shape)}")
            if t.dtype != torch.float32 or (t.shape[1] > 1 and t.stride(1) != 1):
                raise TypeError("the split-bf16 kernel takes float32 [feature, sample] sources with unit sample stride")
            if int(t.shape[0]) < need[name]:
                raise ValueError(f"source {name!r} has {int(t.shape[0])} features, the model needs {need[name]}")
            if n is None:
                n = int(t.shape[1])
            elif int(t.shape[1]) != n:
                raise ValueError("sources differ in their number of samples")
            tensors.append(t)
        dev = _require_device(*tensors)
        nfeat = {o.name: o.nfeat for o in spec.outputs}
        for r in spec.residuals:
            nfeat[r.name] = nfeat[r.output]
        if spec.hidden_output:
            nfeat[spec.hidden_output] = spec.width
        outs = {}
        for name in spec.output_names:
            given = None if out is None else out.get(name)
            if given is None:
                outs[name] = torch.empty((nfeat[name], n), dtype=torch.float32, device=dev)
                continue
            g2 = given.unsqueeze(0) if given.dim() == 1 else given
            if g2.dtype != torch.float32 or tuple(g2.shape) != (nfeat[name], n) or g2.stride(1) != 1 or g2.device != dev:
                raise TypeError(f"out[{name!r}] must be a float32 [{nfeat[name]}, {n}] array with unit sample stride on {dev}")
            outs[name] = g2
        ol = [outs[name] for name in spec.output_names]
        ns, no = len(tensors), len(ol)
        _lib.call_on(dev, "fv3hip_mlp3_predict", self._handle, (ctypes.c_void_p * ns)(*[t.data_ptr() for t in tensors]),
                     (ctypes.c_int64 * ns)(*[t.stride(0) for t in tensors]), n, (ctypes.c_void_p * no)(*[t.data_ptr() for t in ol]),
                     (ctypes.c_int64 * no)(*[t.stride(0) for t in ol]), _stream(dev))
        return outs

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                _lib.load().fv3hip_mlp3_destroy(h)
            except Exception:
                pass
