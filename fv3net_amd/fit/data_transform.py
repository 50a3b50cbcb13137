"""``vcm.DataTransform`` / ``ChainedDataTransform`` (external/vcm/vcm/data_transform.py:1-367) -- the registry of named
dataset transforms that ``TransformedPredictor`` applies to a model's predictions every timestep
(external/fv3fit/fv3fit/_shared/models.py:279-337) -- with the arithmetic on the device: element-wise steps through
``fv3hip_ew`` in the reference's order of operations (the float64 results are the same bits as numpy's), the flux-form
transforms through ``fv3hip_tendency_to_flux`` / ``fv3hip_flux_to_tendency``, the tapering through ``fv3hip_level_scale``."""
import dataclasses
from typing import Callable, MutableMapping, Sequence, Set

from .. import ops
from ..cubedsphere._device import like_input, on_device
from ..xr_compat import DataArray, to_compat
from .models import vertical_tapering_scale_factors

# vcm/calc/thermo/constants.py:2-14
_RDGAS = 287.05
_LATENT_HEAT_VAPORIZATION_0_C = 2.5e6
_SPECIFIC_ENTHALPY_LIQUID = 4185.5
_SPECIFIC_ENTHALPY_VAP0R = 1846
_SPECIFIC_HEAT_CONST_PRESSURE = 1004
_FREEZING_TEMPERATURE = 273.15
_DEFAULT_SURFACE_TEMPERATURE = _FREEZING_TEMPERATURE + 15

DELP = "pressure_thickness_of_atmospheric_layer"
DLW_SFC = "total_sky_downward_longwave_flux_at_surface"
DSW_SFC = "total_sky_downward_shortwave_flux_at_surface"
DSW_TOA = "total_sky_downward_shortwave_flux_at_top_of_atmosphere"
ULW_SFC = "total_sky_upward_longwave_flux_at_surface"
ULW_TOA = "total_sky_upward_longwave_flux_at_top_of_atmosphere"
USW_SFC = "total_sky_upward_shortwave_flux_at_surface"
USW_TOA = "total_sky_upward_shortwave_flux_at_top_of_atmosphere"
COL_T_NUDGE = "storage_of_internal_energy_path_due_to_fine_res_temperature_nudging"
LHF = "latent_heat_flux"
SHF = "sensible_heat_flux"


@dataclasses.dataclass
class DataTransformRegistryEntry:
    func: Callable
    inputs: Sequence[str]
    outputs: Sequence[str]


DATA_TRANSFORM_REGISTRY: MutableMapping[str, DataTransformRegistryEntry] = {}


def register(inputs: Sequence[str], outputs: Sequence[str]):
    def decorator(func):
        name = func.__name__
        if name in DATA_TRANSFORM_REGISTRY:
            raise ValueError(f"Function {name} has already been added to registry.")
        DATA_TRANSFORM_REGISTRY[name] = DataTransformRegistryEntry(func=func, inputs=inputs, outputs=outputs)
        return func

    return decorator


# ---- device arithmetic on DataArrays (same dims in any order; results take the first operand's dims) ----
def _dev(a: DataArray):
    return on_device(a.data).contiguous()


def _wrap(res, like: DataArray, attrs=None) -> DataArray:
    return like._replace(data=like_input(res, like.data), name=None, attrs=attrs or {})


def _binary(op: str, a: DataArray, b: DataArray) -> DataArray:
    if set(a.dims) != set(b.dims):
        raise ValueError(f"data transforms combine arrays over the same dimensions, got {a.dims} and {b.dims}")
    ta, tb = _dev(a), _dev(b.transpose(*a.dims))
    if ta.dtype != tb.dtype:  # numpy promotion
        ta, tb = ta.double(), tb.double()
    return _wrap(ops.ew(op, ta, tb), a)


def _scalar(op: str, a: DataArray, s: float) -> DataArray:
    return _wrap(ops.ew(op, _dev(a), scalar=s), a)


def latent_heat_vaporization(temperature):
    """vcm/calc/thermo/local.py:25-28, a scalar or an array."""
    slope = _SPECIFIC_ENTHALPY_LIQUID - _SPECIFIC_ENTHALPY_VAP0R
    if isinstance(temperature, DataArray):
        return _scalar("add_s", _scalar("mul_s", _scalar("add_s", temperature, -_FREEZING_TEMPERATURE), slope), _LATENT_HEAT_VAPORIZATION_0_C)
    return _LATENT_HEAT_VAPORIZATION_0_C + slope * (temperature - _FREEZING_TEMPERATURE)


def _times_latent_heat(q2: DataArray, temperature) -> DataArray:
    lv = latent_heat_vaporization(temperature)
    return _binary("mul", lv, q2) if isinstance(lv, DataArray) else _scalar("mul_s", q2, lv)


def moist_static_energy_tendency(q1: DataArray, q2: DataArray, temperature=_FREEZING_TEMPERATURE) -> DataArray:
    """vcm/calc/thermo/local.py:317-337: (cp - Rd) Q1 + Lv(T) Q2."""
    out = _binary("add", _scalar("mul_s", q1, _SPECIFIC_HEAT_CONST_PRESSURE - _RDGAS), _times_latent_heat(q2, temperature))
    return out.assign_attrs(units="W/kg", long_name="tendency of moist static energy")


def temperature_tendency(qm: DataArray, q2: DataArray, temperature=_FREEZING_TEMPERATURE) -> DataArray:
    """vcm/calc/thermo/local.py:340-364: (Qm - Lv(T) Q2) / (cp - Rd)."""
    out = _scalar("div_s", _binary("sub", qm, _times_latent_heat(q2, temperature)), _SPECIFIC_HEAT_CONST_PRESSURE - _RDGAS)
    return out.assign_attrs(units="K/s", long_name="tendency of air temperature")


def latent_heat_flux_to_evaporation(lhf: DataArray, surface_temperature: float = _DEFAULT_SURFACE_TEMPERATURE) -> DataArray:
    """vcm/calc/thermo/local.py:69-82."""
    return _scalar("div_s", lhf, latent_heat_vaporization(surface_temperature))


def _sum(first: DataArray, *rest: DataArray) -> DataArray:
    out = first
    for r in rest:
        out = _binary("add", out, r)
    return out


def _tendency_to_flux(tendency: DataArray, toa, up: DataArray, delp: DataArray, dim="z", rectify=True, closure_only=False):
    """vcm/calc/flux_form.py:7-75 on DataArrays: ``tendency`` / ``delp`` over the same dims including ``dim``, the surface
    arrays over the others (``toa`` None = zero)."""
    axis = tendency.get_axis_num(dim)
    flat_dims = tuple(d for d in tendency.dims if d != dim)
    flux, down = ops.tendency_to_flux(_dev(tendency), _dev(delp.transpose(*tendency.dims)),
                                      None if toa is None else _dev(toa.transpose(*flat_dims)), _dev(up.transpose(*flat_dims)), axis,
                                      rectify=rectify, closure_only=closure_only)
    sfc = DataArray(like_input(down, tendency.data), dims=flat_dims, coords={k: v for k, v in up.coords.items() if k in flat_dims})
    return (None if flux is None else _wrap(flux, tendency)), sfc


def _flux_to_tendency(net_flux: DataArray, down: DataArray, up: DataArray, delp: DataArray, dim="z") -> DataArray:
    axis = net_flux.get_axis_num(dim)
    flat_dims = tuple(d for d in net_flux.dims if d != dim)
    res = ops.flux_to_tendency(_dev(net_flux), _dev(down.transpose(*flat_dims)), _dev(up.transpose(*flat_dims)),
                               _dev(delp.transpose(*net_flux.dims)), axis)
    return _wrap(res, net_flux)


def _tapered(ds, name: str, cutoff: int, rate: float) -> DataArray:
    da = ds[name]
    scaling = vertical_tapering_scale_factors(n_levels=ds.dims["z"], cutoff=cutoff, rate=rate)
    return _wrap(ops.level_scale(_dev(da), on_device(scaling), da.get_axis_num("z")), da)


# ---- the registry (data_transform.py:68-330); every function updates and returns the dataset, as the reference's do ----
@register(["dQ1"], ["tapered_dQ1"])
def tapered_dQ1(ds, cutoff: int, rate: float):
    ds["tapered_dQ1"] = _tapered(ds, "dQ1", cutoff, rate)
    return ds


@register(["dQ2"], ["tapered_dQ2"])
def tapered_dQ2(ds, cutoff: int, rate: float):
    ds["tapered_dQ2"] = _tapered(ds, "dQ2", cutoff, rate)
    return ds


@register(["Q1", "Q2"], ["Qm"])
def Qm_from_Q1_Q2(ds):
    ds["Qm"] = moist_static_energy_tendency(ds["Q1"], ds["Q2"])
    return ds


@register(["Qm", "Q2"], ["Q1"])
def Q1_from_Qm_Q2(ds):
    ds["Q1"] = temperature_tendency(ds["Qm"], ds["Q2"])
    return ds


@register(["Q1", "Q2", "air_temperature"], ["Qm"])
def Qm_from_Q1_Q2_temperature_dependent(ds):
    ds["Qm"] = moist_static_energy_tendency(ds["Q1"], ds["Q2"], temperature=ds["air_temperature"])
    return ds


@register(["Qm", "Q2", "air_temperature"], ["Q1"])
def Q1_from_Qm_Q2_temperature_dependent(ds):
    ds["Q1"] = temperature_tendency(ds["Qm"], ds["Q2"], temperature=ds["air_temperature"])
    return ds


@register(["dQ1", "pQ1"], ["Q1"])
def Q1_from_dQ1_pQ1(ds):
    ds["Q1"] = _binary("add", ds["dQ1"], ds["pQ1"])
    return ds


@register(["dQ2", "pQ2"], ["Q2"])
def Q2_from_dQ2_pQ2(ds):
    ds["Q2"] = _binary("add", ds["dQ2"], ds["pQ2"])
    return ds


_QM_FLUX_INPUTS = ["Qm", DELP, DLW_SFC, DSW_SFC, DSW_TOA, ULW_SFC, ULW_TOA, USW_SFC, USW_TOA, LHF, SHF, COL_T_NUDGE]


def _toa_net_flux(ds, include_temperature_nudging: bool) -> DataArray:
    toa = _binary("sub", _binary("sub", ds[DSW_TOA], ds[USW_TOA]), ds[ULW_TOA])
    return _binary("add", toa, ds[COL_T_NUDGE]) if include_temperature_nudging else toa


@register(_QM_FLUX_INPUTS, ["Qm_flux", "implied_downward_radiative_flux_at_surface"])
def Qm_flux_from_Qm_tendency(ds, rectify_downward_radiative_flux=True, include_temperature_nudging=True):
    up = _sum(ds[LHF], ds[SHF], ds[USW_SFC], ds[ULW_SFC])
    net_flux, down = _tendency_to_flux(ds["Qm"], _toa_net_flux(ds, include_temperature_nudging), up, ds[DELP], dim="z",
                                       rectify=rectify_downward_radiative_flux)
    ds["Qm_flux"] = net_flux.assign_attrs(units="W/m**2", long_name="Net flux of MSE")
    ds["implied_downward_radiative_flux_at_surface"] = down.assign_attrs(
        units="W/m**2", long_name="Implied downward radiative flux from <Qm> budget closure")
    return ds


@register(["Q2", DELP, LHF], ["Q2_flux", "implied_surface_precipitation_rate"])
def Q2_flux_from_Q2_tendency(ds, rectify_surface_precipitation_rate=True):
    net_flux, down = _tendency_to_flux(ds["Q2"], None, latent_heat_flux_to_evaporation(ds[LHF]), ds[DELP], dim="z",
                                       rectify=rectify_surface_precipitation_rate)
    ds["Q2_flux"] = net_flux.assign_attrs(units="kg/s/m**2", long_name="Net flux of moisture")
    ds["implied_surface_precipitation_rate"] = down.assign_attrs(
        units="kg/s/m**2", long_name="Implied surface precipitation rate computed as E-<Q2>")
    return ds


@register(["Qm_flux", "implied_downward_radiative_flux_at_surface", DELP, ULW_SFC, USW_SFC, LHF, SHF], ["Qm"])
def Qm_tendency_from_Qm_flux(ds):
    up = _sum(ds[LHF], ds[SHF], ds[USW_SFC], ds[ULW_SFC])
    ds["Qm"] = _flux_to_tendency(ds["Qm_flux"], ds["implied_downward_radiative_flux_at_surface"], up, ds[DELP]).assign_attrs(units="W/kg")
    return ds


@register(["Q2_flux", "implied_surface_precipitation_rate", DELP, LHF], ["Q2"])
def Q2_tendency_from_Q2_flux(ds):
    ds["Q2"] = _flux_to_tendency(ds["Q2_flux"], ds["implied_surface_precipitation_rate"], latent_heat_flux_to_evaporation(ds[LHF]),
                                 ds[DELP]).assign_attrs(units="kg/kg/s")
    return ds


@register(_QM_FLUX_INPUTS, ["implied_downward_radiative_flux_at_surface"])
def implied_downward_radiative_flux_at_surface(ds, rectify=True, include_temperature_nudging=True):
    """Assuming <Qm> = SHF + LHF + R_net + <T_nudge>."""
    up = _sum(ds[LHF], ds[SHF], ds[USW_SFC], ds[ULW_SFC])
    _, down = _tendency_to_flux(ds["Qm"], _toa_net_flux(ds, include_temperature_nudging), up, ds[DELP], dim="z", rectify=rectify,
                                closure_only=True)
    ds["implied_downward_radiative_flux_at_surface"] = down.assign_attrs(
        units="W/m**2", long_name="Implied downward radiative flux from <Qm> budget closure")
    return ds


@register(["Q2", DELP, LHF], ["implied_surface_precipitation_rate"])
def implied_surface_precipitation_rate(ds, rectify=True):
    """Assuming <Q2> = E-P."""
    _, down = _tendency_to_flux(ds["Q2"], None, latent_heat_flux_to_evaporation(ds[LHF]), ds[DELP], dim="z", rectify=rectify,
                                closure_only=True)
    ds["implied_surface_precipitation_rate"] = down.assign_attrs(
        units="kg/s/m**2", long_name="Implied surface precipitation rate computed as E-<Q2>")
    return ds


def _incloud_to_gridcell(cloud_fraction: DataArray, incloud: DataArray) -> DataArray:
    """vcm/calc/clouds.py:40-66 with its default limits."""
    res = _binary("incloud_to_gridcell", cloud_fraction, incloud)
    return res.transpose(*incloud.dims)


@register(["cloud_amount", "incloud_water_mixing_ratio"], ["cloud_water_mixing_ratio"])
def cloud_water_mixing_ratio_from_incloud(ds):
    ds["cloud_water_mixing_ratio"] = _incloud_to_gridcell(ds["cloud_amount"], ds["incloud_water_mixing_ratio"]).assign_attrs(
        long_name="cloud water mixing ratio", units="kg/kg")
    return ds


@register(["cloud_amount", "incloud_ice_mixing_ratio"], ["cloud_ice_mixing_ratio"])
def cloud_ice_mixing_ratio_from_incloud(ds):
    ds["cloud_ice_mixing_ratio"] = _incloud_to_gridcell(ds["cloud_amount"], ds["incloud_ice_mixing_ratio"]).assign_attrs(
        long_name="cloud ice mixing ratio", units="kg/kg")
    return ds


@dataclasses.dataclass
class DataTransform:
    name: str
    kwargs: dict = dataclasses.field(default_factory=dict)

    def __post_init__(self):
        if self.name not in DATA_TRANSFORM_REGISTRY:
            raise ValueError(f"unknown data transform {self.name!r}; known: {sorted(DATA_TRANSFORM_REGISTRY)}")

    def apply(self, ds):
        return DATA_TRANSFORM_REGISTRY[self.name].func(to_compat(ds), **self.kwargs)

    @property
    def input_variables(self) -> Sequence[str]:
        return DATA_TRANSFORM_REGISTRY[self.name].inputs

    @property
    def output_variables(self) -> Sequence[str]:
        return DATA_TRANSFORM_REGISTRY[self.name].outputs


@dataclasses.dataclass
class ChainedDataTransform:
    transforms: Sequence[DataTransform]

    def apply(self, ds):
        for transform in self.transforms:
            ds = transform.apply(ds)
        return ds

    @property
    def input_variables(self) -> Sequence[str]:
        inputs: Set[str] = set()
        for transform in self.transforms[::-1]:
            inputs.update(transform.input_variables)
            for output in transform.output_variables:
                inputs.discard(output)
        return sorted(list(inputs))

    @property
    def output_variables(self) -> Sequence[str]:
        outputs: Set[str] = set()
        for transform in self.transforms:
            outputs.update(transform.output_variables)
        return sorted(list(outputs))
