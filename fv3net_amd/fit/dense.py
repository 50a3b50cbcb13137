"""The dense column predictor: ``fv3fit``'s ``PureKerasModel`` + ``DenseModel`` training,
re-done for MI355X.

* ``HipDenseModel`` is the drop-in for ``fv3fit.keras.PureKerasModel``
  (external/fv3fit/fv3fit/keras/_models/shared/pure_keras.py:22-102): same constructor meaning,
  ``predict(X) -> Dataset``, ``dump`` / ``load`` through the ``name``-file registry.  The network
  runs in the fused HIP kernel; the dataset is consumed in its native ``[z, ...]`` layout, there
  is no stack / unstack copy (xr_prediction.py:111-139 does two).
* ``train_dense_model`` is the offline training step (dense.py:90-107,165-310) in PyTorch-ROCm:
  the only place PyTorch computes anything.
* The artifact is ``name`` ("hip-dense") + ``config.yaml`` (the reference's keys) +
  ``spec.yaml`` + ``weights.npz``; TensorFlow SavedModels cannot be read without TensorFlow.
"""
import dataclasses
import os
from typing import Dict, Hashable, Iterable, List, Mapping, Optional, Sequence

import numpy as np
import torch
import yaml

from ..cubedsphere._device import compute_device, download_all, on_device
from ..mlp import InputSpec, MlpModel, MlpSpec, OutputSpec
from ..xr_compat import DataArray, Dataset, from_compat, to_compat
from . import io
from .predictor import Predictor
from .stacking import column_sources, match_prediction_to_input_coords


@io.register("hip-dense")
class HipDenseModel(Predictor):
    """Column MLP predictor running on the MI355X fused kernel."""

    _CONFIG_FILENAME = "config.yaml"
    _SPEC_FILENAME = "spec.yaml"
    _WEIGHTS_FILENAME = "weights.npz"

    def __init__(self, input_variables: Iterable[Hashable], output_variables: Iterable[Hashable], model: MlpSpec,
                 unstacked_dims: Sequence[str] = ("z",), n_halo: int = 0):
        super().__init__(list(input_variables), list(output_variables))
        if n_halo != 0:
            raise ValueError("dense column models take no halo (dense.py:233); n_halo must be 0")
        self.spec = model
        self._n_halo = n_halo
        self._unstacked_dims = list(unstacked_dims) if unstacked_dims is not None else []
        names = model.output_names
        missing = [v for v in self.output_variables if v not in names]
        if missing:
            raise ValueError(f"output variables {missing} are not produced by the network ({names})")
        self._model: Optional[MlpModel] = None  # created on first predict (needs the GPU)

    @property
    def model(self) -> MlpModel:
        if self._model is None:
            self._model = MlpModel(self.spec, device=compute_device())
        return self._model

    # -- prediction -------------------------------------------------------------------------
    def predict(self, X):
        """Predict an output dataset from an input dataset.  Does not mutate ``X``."""
        x = to_compat(X)
        sources, sample_dims, sizes, zname, host_input = column_sources(x, self.spec.sources, self._unstacked_dims)
        outs = self.model.predict(sources, layout="feature_sample")

        result = Dataset()
        nfeat_out = {o.name: o.nfeat for o in self.spec.outputs}
        for r in self.spec.residuals:
            nfeat_out[r.name] = nfeat_out[r.output]
        shaped, dims_of = {}, {}
        for name in self.output_variables:
            t = outs[name]
            if nfeat_out[name] == 1:
                shaped[name], dims_of[name] = t.reshape([sizes[d] for d in sample_dims]), tuple(sample_dims)
            else:
                shaped[name] = t.reshape([nfeat_out[name]] + [sizes[d] for d in sample_dims])
                dims_of[name] = (zname,) + tuple(sample_dims)
        if not (isinstance(host_input, torch.Tensor) and host_input.is_cuda):
            shaped = download_all(shaped)  # host data in -> host data out, one copy for all outputs
        for name in self.output_variables:
            result[name] = DataArray(shaped[name], dims=dims_of[name])
        return from_compat(match_prediction_to_input_coords(x, result), X)

    # -- serialisation ----------------------------------------------------------------------
    def dump(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        meta, arrays = self.spec.to_arrays()
        np.savez(os.path.join(path, self._WEIGHTS_FILENAME), **arrays)
        with open(os.path.join(path, self._SPEC_FILENAME), "w") as f:
            yaml.safe_dump(meta, f)
        with open(os.path.join(path, self._CONFIG_FILENAME), "w") as f:
            yaml.safe_dump(
                {"input_variables": list(self.input_variables), "output_variables": list(self.output_variables),
                 "unstacked_dims": list(self._unstacked_dims), "n_halo": self._n_halo}, f)

    @classmethod
    def load(cls, path: str) -> "HipDenseModel":
        with open(os.path.join(path, cls._CONFIG_FILENAME)) as f:
            config = yaml.safe_load(f)
        with open(os.path.join(path, cls._SPEC_FILENAME)) as f:
            meta = yaml.safe_load(f)
        with np.load(os.path.join(path, cls._WEIGHTS_FILENAME), allow_pickle=False) as z:
            arrays = {k: z[k] for k in z.files}
        return cls(config["input_variables"], config["output_variables"], MlpSpec.from_arrays(meta, arrays),
                   unstacked_dims=config.get("unstacked_dims", None), n_halo=config.get("n_halo", 0))


# ---------------------------------------------------------------------------------------------
# building a spec from arrays (an exporter run where TensorFlow exists produces exactly these)
# ---------------------------------------------------------------------------------------------
def spec_from_arrays(
    input_variables: Sequence[str],
    input_means: Sequence[np.ndarray],
    input_stds: Sequence[np.ndarray],
    hidden_kernels: Sequence[np.ndarray],
    hidden_biases: Sequence[np.ndarray],
    output_variables: Sequence[str],
    output_kernels: Sequence[np.ndarray],
    output_biases: Sequence[np.ndarray],
    output_means: Sequence[np.ndarray],
    output_stds: Sequence[np.ndarray],
    clip: Optional[Mapping[str, slice]] = None,
    limits: Optional[Mapping[str, Sequence[Optional[float]]]] = None,
    epsilon: float = 1e-7,
) -> MlpSpec:
    """Assemble the predict graph of dense.py:239-310 from its weights.  ``input_stds`` are the
    fitted standard deviations; the forward scale is ``std + epsilon`` in float32
    (normalization.py:39-46).  ``clip[name]`` keeps a slice of an input's levels (clip.py:48-62)
    and zero-masks the complement of an output's (clip.py:33-46); means/stds of clipped inputs
    are those of the kept levels."""
    clip = dict(clip or {})
    limits = dict(limits or {})
    inputs = []
    for name, mean, std in zip(input_variables, input_means, input_stds):
        mean, std = np.atleast_1d(np.asarray(mean, np.float32)), np.atleast_1d(np.asarray(std, np.float32))
        start = 0
        if name in clip:
            start = clip[name].start or 0
        inputs.append(InputSpec(source=name, nfeat=int(mean.shape[0]), start=int(start), center=mean,
                                scale=std + np.float32(epsilon)))
    outputs = []
    for name, kern, mean, std in zip(output_variables, output_kernels, output_means, output_stds):
        nf = int(np.asarray(kern).shape[1])
        mask = None
        if name in clip:
            s = clip[name]
            mask = np.zeros(nf, np.float32)
            mask[(s.start or 0):(s.stop if s.stop is not None else nf)] = 1.0
        lo, hi = limits.get(name, (None, None))
        outputs.append(OutputSpec(name=name, nfeat=nf, scale=np.broadcast_to(np.asarray(std, np.float32), (nf,)).copy(),
                                  center=np.broadcast_to(np.asarray(mean, np.float32), (nf,)).copy(), min=lo, max=hi,
                                  mask=mask))
    return MlpSpec(
        inputs=inputs,
        hidden_kernels=[np.asarray(k, np.float32) for k in hidden_kernels],
        hidden_biases=[np.asarray(b, np.float32) for b in hidden_biases],
        outputs=outputs,
        out_kernel=np.concatenate([np.asarray(k, np.float32) for k in output_kernels], axis=1),
        out_bias=np.concatenate([np.asarray(b, np.float32) for b in output_biases]),
    )


# ---------------------------------------------------------------------------------------------
# offline training (PyTorch-ROCm)
# ---------------------------------------------------------------------------------------------
@dataclasses.dataclass
class DenseHyperparameters:
    """The subset of fv3fit's ``DenseHyperparameters`` (dense.py:39-87) that shapes the network."""

    input_variables: List[str]
    output_variables: List[str]
    width: int = 8            # DenseNetworkConfig.width   (dense_network.py:31)
    depth: int = 3            # DenseNetworkConfig.depth: hidden layers + 1 (dense_network.py:32)
    epochs: int = 3           # TrainingLoopConfig.epochs
    batch_size: int = 512
    learning_rate: float = 1e-3
    normalization_fit_samples: int = 500_000   # dense.py:211-217
    clip: Dict[str, slice] = dataclasses.field(default_factory=dict)
    limits: Dict[str, Sequence[Optional[float]]] = dataclasses.field(default_factory=dict)
    unstacked_dims: Sequence[str] = ("z",)
    seed: int = 0


def _stack_for_training(batches: Sequence[Dataset], names: Sequence[str], unstacked_dims) -> List[np.ndarray]:
    cols: List[List[np.ndarray]] = [[] for _ in names]
    for ds in batches:
        ds = to_compat(ds)
        for i, name in enumerate(names):
            da = ds[name]
            zs = [d for d in da.dims if d in unstacked_dims]
            rest = [d for d in da.dims if d not in unstacked_dims]
            a = da.transpose(*rest, *zs).values
            cols[i].append(a.reshape(-1, a.shape[-1]) if zs else a.reshape(-1, 1))  # ensure_nd (dense.py:146)
    return [np.concatenate(c, axis=0).astype(np.float32) for c in cols]


def train_dense_model(hyperparameters: DenseHyperparameters, train_batches: Sequence[Dataset],
                      device: Optional[str] = None) -> HipDenseModel:
    """Fit normalisation on up to ``normalization_fit_samples`` samples, train
    ``Dense(width, relu) x (depth-1) -> Dense per output`` with Adam on the std-scaled MSE of the
    (clipped) outputs, and return the predictor (dense.py:165-310, training_loop.py:86-138)."""
    hp = hyperparameters
    torch.manual_seed(hp.seed)
    dev = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
    X = _stack_for_training(train_batches, hp.input_variables, hp.unstacked_dims)
    y = _stack_for_training(train_batches, hp.output_variables, hp.unstacked_dims)
    nfit = hp.normalization_fit_samples

    def clipped(arr, name):
        return arr[..., hp.clip[name]] if name in hp.clip else arr

    Xc = [clipped(a, n) for a, n in zip(X, hp.input_variables)]
    yc = [clipped(a, n) for a, n in zip(y, hp.output_variables)]
    x_mean = [a[:nfit].mean(axis=0).astype(np.float32) for a in Xc]
    x_std = [a[:nfit].std(axis=0).astype(np.float32) for a in Xc]
    y_mean = [a[:nfit].mean(axis=0).astype(np.float32) for a in y]
    y_std = [a[:nfit].std(axis=0).astype(np.float32) for a in y]
    yc_std = [np.std(a[:nfit], axis=0, dtype=np.float32) for a in yc]

    eps = np.float32(1e-7)
    xin = torch.from_numpy(np.concatenate([(a - m) / (s + eps) for a, m, s in zip(Xc, x_mean, x_std)], axis=1)).to(dev)
    k = xin.shape[1]
    layers: List[torch.nn.Module] = []
    fan = k
    for _ in range(hp.depth - 1):
        lin = torch.nn.Linear(fan, hp.width)
        torch.nn.init.xavier_uniform_(lin.weight)  # Keras Dense default: glorot_uniform, zero bias
        torch.nn.init.zeros_(lin.bias)
        layers += [lin, torch.nn.ReLU()]
        fan = hp.width
    trunk = torch.nn.Sequential(*layers).to(dev)
    heads = torch.nn.ModuleList()
    for a in y:
        lin = torch.nn.Linear(fan, a.shape[1])
        torch.nn.init.xavier_uniform_(lin.weight)
        torch.nn.init.zeros_(lin.bias)
        heads.append(lin)
    heads = heads.to(dev)
    targets = [torch.from_numpy(a).to(dev) for a in yc]
    t_mean = [torch.from_numpy(m).to(dev) for m in y_mean]
    t_std = [torch.from_numpy(s).to(dev) for s in y_std]
    t_cstd = [torch.from_numpy(np.maximum(s, 1e-12)).to(dev) for s in yc_std]
    opt = torch.optim.Adam(list(trunk.parameters()) + list(heads.parameters()), lr=hp.learning_rate)
    n = xin.shape[0]
    g = torch.Generator().manual_seed(hp.seed)
    for _ in range(hp.epochs):
        perm = torch.randperm(n, generator=g).to(dev)
        for i in range(0, n, hp.batch_size):
            idx = perm[i:i + hp.batch_size]
            hidden = trunk(xin[idx])
            loss = 0.0
            for j, name in enumerate(hp.output_variables):
                pred = heads[j](hidden) * t_std[j] + t_mean[j]
                lo, hi = hp.limits.get(name, (None, None))
                if lo is not None or hi is not None:
                    pred = torch.clamp(pred, min=lo, max=hi)
                if name in hp.clip:
                    pred = pred[..., hp.clip[name]]
                loss = loss + torch.mean(((pred - targets[j][idx]) / t_cstd[j]) ** 2)
            opt.zero_grad()
            loss.backward()
            opt.step()

    lin_layers = [m for m in trunk if isinstance(m, torch.nn.Linear)]
    spec = spec_from_arrays(
        hp.input_variables, x_mean, x_std,
        [m.weight.detach().cpu().numpy().T.copy() for m in lin_layers],
        [m.bias.detach().cpu().numpy().copy() for m in lin_layers],
        hp.output_variables,
        [h.weight.detach().cpu().numpy().T.copy() for h in heads],
        [h.bias.detach().cpu().numpy().copy() for h in heads],
        y_mean, y_std, clip=hp.clip, limits=hp.limits,
    )
    return HipDenseModel(hp.input_variables, hp.output_variables, spec, unstacked_dims=hp.unstacked_dims)
