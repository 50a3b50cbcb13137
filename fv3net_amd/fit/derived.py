"""``DerivedMapping`` for the variables that ML predictions are turned into right after the network
(external/vcm/vcm/derived_mapping.py:8-112 for the mechanics; :197-250 the surface shortwave fluxes via albedo /
transmissivity, :417-448 Q1 / Q2 / pQ1 / pQ2) and ``DerivedModel`` (external/fv3fit/fv3fit/_shared/models.py:110-220).
The element-wise arithmetic runs on the device (``fv3hip_ew``).  The rest of the reference's mapping (winds, cos zenith
angle, land / sea masks, EAMXX radiation splits, column integrals, ...) registers itself from ``derived_more``."""
import os
from typing import Callable, Hashable, Iterable, List, Mapping, MutableMapping, Sequence

import numpy as np
import yaml

from .. import ops
from ..cubedsphere._device import like_input, on_device
from ..xr_compat import DataArray, Dataset, from_compat, merge, to_compat
from . import io
from .predictor import Predictor


def _binary(op: str, a: DataArray, b: DataArray) -> DataArray:
    """Element-wise ``a op b`` of two arrays with the same dims (any order), on the device."""
    if set(a.dims) != set(b.dims):
        raise ValueError(f"derived variables combine arrays over the same dimensions, got {a.dims} and {b.dims}")
    b = b.transpose(*a.dims)
    ta, tb = on_device(a.data), on_device(b.data)
    if ta.dtype != tb.dtype:  # numpy promotion
        ta, tb = ta.double(), tb.double()
    return a._replace(data=like_input(ops.ew(op, ta.contiguous(), tb.contiguous()), a.data), name=None)


def _scalar(op: str, a: DataArray, s: float) -> DataArray:
    return a._replace(data=like_input(ops.ew(op, on_device(a.data).contiguous(), scalar=s), a.data), name=None)


class DerivedMapping(Mapping):
    """A uniform mapping-like interface for both existing and derived variables (derived_mapping.py:8-112)."""

    VARIABLES: MutableMapping[Hashable, Callable[..., DataArray]] = {}
    REQUIRED_INPUTS: MutableMapping[Hashable, Iterable[Hashable]] = {}
    USE_NONDERIVED_IF_EXISTS: List[Hashable] = []

    def __init__(self, mapper):
        self._mapper = to_compat(mapper)

    @classmethod
    def register(cls, name: Hashable, required_inputs: Iterable[Hashable] = None, use_nonderived_if_exists: bool = False):
        def decorator(func):
            cls.VARIABLES[name] = func
            if required_inputs:
                cls.REQUIRED_INPUTS[name] = required_inputs
            if use_nonderived_if_exists is True:
                cls.USE_NONDERIVED_IF_EXISTS.append(name)
            return func

        return decorator

    def __getitem__(self, key: Hashable) -> DataArray:
        if key in self.VARIABLES:
            if key in self.USE_NONDERIVED_IF_EXISTS:
                try:
                    return self._mapper[key]
                except KeyError:
                    return self.VARIABLES[key](self)
            return self.VARIABLES[key](self)
        return self._mapper[key]

    def keys(self):
        return set(self._mapper) | set(self.VARIABLES)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def dataset(self, keys: Iterable[Hashable]) -> Dataset:
        return Dataset({key: self[key] for key in keys})

    @classmethod
    def find_all_required_inputs(cls, derived_variables: Iterable[Hashable]) -> List[Hashable]:
        """All non-derived inputs of ``derived_variables``, through intermediate derived ones (derived_mapping.py:85-112)."""
        def recurse(variables, deps):
            with_deps = [v for v in variables if v in cls.REQUIRED_INPUTS]
            if not with_deps:
                return
            new_deps = []
            for v in with_deps:
                new_deps += list(cls.REQUIRED_INPUTS[v])
            deps += new_deps
            recurse(new_deps, deps)

        deps: List[Hashable] = []
        recurse(list(derived_variables), deps)
        nonderived = list(set(d for d in deps if d not in cls.VARIABLES))
        maybe_nonderived = list(set(d for d in deps if d in cls.USE_NONDERIVED_IF_EXISTS))
        return nonderived + maybe_nonderived


def _net_sfc_shortwave_flux_via_albedo(downward_sfc_shortwave_flux: DataArray, albedo: DataArray) -> DataArray:
    # (1 - albedo) * downward flux (derived_mapping.py:194-195)
    return _binary("mul", _scalar("add_s", _scalar("mul_s", albedo, -1.0), 1.0), downward_sfc_shortwave_flux)


@DerivedMapping.register("net_shortwave_sfc_flux_derived", required_inputs=[
    "surface_diffused_shortwave_albedo", "override_for_time_adjusted_total_sky_downward_shortwave_flux_at_surface"])
def net_shortwave_sfc_flux_derived(self):
    return _net_sfc_shortwave_flux_via_albedo(
        self["override_for_time_adjusted_total_sky_downward_shortwave_flux_at_surface"], self["surface_diffused_shortwave_albedo"])


@DerivedMapping.register("downward_shortwave_sfc_flux_via_transmissivity", required_inputs=[
    "total_sky_downward_shortwave_flux_at_top_of_atmosphere", "shortwave_transmissivity_of_atmospheric_column"])
def downward_shortwave_sfc_flux_via_transmissivity(self):
    return _binary("mul", self["shortwave_transmissivity_of_atmospheric_column"],
                   self["total_sky_downward_shortwave_flux_at_top_of_atmosphere"])


@DerivedMapping.register("net_shortwave_sfc_flux_via_transmissivity", required_inputs=[
    "surface_diffused_shortwave_albedo", "downward_shortwave_sfc_flux_via_transmissivity"])
def net_shortwave_sfc_flux_via_transmissivity(self):
    return _net_sfc_shortwave_flux_via_albedo(self["downward_shortwave_sfc_flux_via_transmissivity"],
                                              self["surface_diffused_shortwave_albedo"])


def _zeros_like(a: DataArray) -> DataArray:
    import torch

    return a._replace(data=like_input(torch.zeros_like(on_device(a.data)), a.data), name=None)


@DerivedMapping.register("Q1", required_inputs=["pQ1"], use_nonderived_if_exists=True)
def Q1(self):
    return _add(self["dQ1"], self["pQ1"]) if "dQ1" in self.keys() else self["pQ1"]


@DerivedMapping.register("Q2", required_inputs=["pQ2"], use_nonderived_if_exists=True)
def Q2(self):
    return _add(self["dQ2"], self["pQ2"]) if "dQ2" in self.keys() else self["pQ2"]


@DerivedMapping.register("pQ1", required_inputs=["pressure_thickness_of_atmospheric_layer"], use_nonderived_if_exists=True)
def pQ1(self):
    return _zeros_like(self["pressure_thickness_of_atmospheric_layer"])


@DerivedMapping.register("pQ2", required_inputs=["pressure_thickness_of_atmospheric_layer"], use_nonderived_if_exists=True)
def pQ2(self):
    return _zeros_like(self["pressure_thickness_of_atmospheric_layer"])


def _add(a: DataArray, b: DataArray) -> DataArray:
    return _binary("add", a, b)


@io.register("derived_model")
class DerivedModel(Predictor):
    _CONFIG_FILENAME = "derived_model.yaml"
    _BASE_MODEL_SUBDIR = "base_model_data"

    def __init__(self, model: Predictor, derived_output_variables: Sequence[Hashable]):
        # a DerivedModel of a DerivedModel wraps the underlying base model once (models.py:130-140)
        if isinstance(model, DerivedModel):
            self.base_model: Predictor = model.base_model
            self._derived_output_variables = list(model._derived_output_variables) + list(derived_output_variables)
        else:
            self.base_model = model
            self._derived_output_variables = list(derived_output_variables)
        self._additional_input_variables = self.get_additional_inputs()
        full_inputs = sorted(set(list(model.input_variables) + list(self._additional_input_variables)))
        full_outputs = sorted(set(list(model.output_variables) + list(derived_output_variables)))
        self._check_derived_predictions_supported()
        super().__init__(full_inputs, full_outputs)

    def get_additional_inputs(self):
        derived_variable_inputs = DerivedMapping.find_all_required_inputs(self._derived_output_variables)
        return [name for name in derived_variable_inputs if name not in self.base_model.output_variables]

    def predict(self, X):
        x = to_compat(X)
        self._check_additional_inputs_present(x)
        base_prediction = to_compat(self.base_model.predict(X))
        required_inputs = Dataset({name: x[name] for name in self._additional_input_variables})
        derived_mapping = DerivedMapping(merge([required_inputs, base_prediction]))
        derived_prediction = derived_mapping.dataset(self._derived_output_variables)
        return from_compat(merge([base_prediction, derived_prediction]), X)

    def dump(self, path: str):
        base_model_path = os.path.join(path, self._BASE_MODEL_SUBDIR)
        io.dump(self.base_model, base_model_path)
        with open(os.path.join(path, self._CONFIG_FILENAME), "w") as f:
            yaml.safe_dump({"derived_output_variables": list(self._derived_output_variables), "model": base_model_path}, f)

    @classmethod
    def load(cls, path: str) -> "DerivedModel":
        with open(os.path.join(path, cls._CONFIG_FILENAME)) as f:
            config = yaml.safe_load(f)
        return cls(io.load(config["model"]), config["derived_output_variables"])

    def _check_additional_inputs_present(self, X):
        missing = np.setdiff1d(list(self._additional_input_variables), list(X))
        if len(missing) > 0:
            raise KeyError(f"Missing additional inputs {missing} in input dataset needed to compute derived prediction "
                           "variables. Make sure these are present in the data and included in the DerivedModel config "
                           "under additional_input_variables.")

    def _check_derived_predictions_supported(self):
        invalid = np.setdiff1d(list(self._derived_output_variables), list(DerivedMapping.VARIABLES))
        if len(invalid) > 0:
            raise ValueError(f"Invalid variables {invalid} provided in init arg derived_output_variables. Variables in this "
                             "arg must be available as derived variables in vcm.DerivedMapping.")


from . import derived_more  # noqa: E402,F401  (registers the rest of the reference's derived variables)
