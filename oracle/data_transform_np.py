"""ORACLE -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the output transforms ``TransformedPredictor`` applies and of what ``OutOfSampleModel`` computes
around its base model.  Arrays are plain numpy: column arrays ``[z, ...]`` (the vertical axis first), surface arrays
``[...]``.  Follows (paths relative to the reference checkout):
  external/vcm/vcm/data_transform.py:68-330                 the registered transforms
  external/vcm/vcm/calc/flux_form.py:7-104                  tendencies <-> interface fluxes, budget closure
  external/vcm/vcm/calc/thermo/vertically_dependent.py:18-38, thermo/local.py:25-28,69-82,317-364, constants.py:2-14
  external/vcm/vcm/calc/clouds.py:40-66, calc/calc.py:52-56
  external/fv3fit/fv3fit/_shared/taper_function.py:23-64
  external/fv3fit/fv3fit/sklearn/_min_max_novelty_detector.py:94-121 (score from MinMaxScaler.transform's output)
Pinned by the known answers the reference's tests hold (tests/test_oracle_data_transform.py: vcm/tests/test_calc_clouds.py,
fv3fit/tests/test_taper.py) and by its round-trip properties (vcm/tests/test_flux_form.py, test_data_transform.py); the
sklearn scalers / OneClassSVM are installed and serve as their own oracle in the tests."""
import numpy as np

GRAVITY = 9.80665
RDGAS = 287.05
LV0 = 2.5e6
C_LIQUID = 4185.5
C_VAPOR = 1846
CP = 1004
T_FREEZE = 273.15
T_SURFACE_DEFAULT = T_FREEZE + 15
CLIMIT1, CLIMIT2 = 1.0e-3, 5.0e-2


def vertical_tapering_scale_factors(n_levels, cutoff, rate):
    z = np.arange(n_levels)
    return np.hstack([np.exp((z[slice(None, cutoff)] - cutoff) / rate), np.ones(n_levels - cutoff)])


def latent_heat_vaporization(t):
    return LV0 + (C_LIQUID - C_VAPOR) * (t - T_FREEZE)


def moist_static_energy_tendency(q1, q2, temperature=T_FREEZE):
    return (CP - RDGAS) * q1 + latent_heat_vaporization(temperature) * q2


def temperature_tendency(qm, q2, temperature=T_FREEZE):
    return (qm - latent_heat_vaporization(temperature) * q2) / (CP - RDGAS)


def latent_heat_flux_to_evaporation(lhf, surface_temperature=T_SURFACE_DEFAULT):
    return lhf / latent_heat_vaporization(surface_temperature)


def tendency_to_flux(tendency, toa_net_flux, surface_upward_flux, delp, rectify=True):
    flux = -np.cumsum(tendency * delp / GRAVITY, axis=0)
    flux = np.concatenate([np.zeros_like(flux[:1]), flux], axis=0)
    flux = flux + toa_net_flux
    down = flux[-1] + surface_upward_flux
    if rectify:
        down = np.where(down >= 0, down, 0)
    return flux[:-1], down


def tendency_to_implied_surface_downward_flux(tendency, toa_net_flux, surface_upward_flux, delp, rectify=True):
    down = toa_net_flux + surface_upward_flux - (tendency * delp / GRAVITY).sum(axis=0)
    if rectify:
        down = np.where(down >= 0, down, 0)
    return down


def flux_to_tendency(net_flux, surface_downward_flux, surface_upward_flux, delp):
    full = np.concatenate([net_flux, (surface_downward_flux - surface_upward_flux)[None]], axis=0)
    return -(GRAVITY * np.diff(full, axis=0) / delp)


def incloud_to_gridcell_condensate(cloud_fraction, incloud, climit1=CLIMIT1, climit2=CLIMIT2):
    rectified = np.where(cloud_fraction > climit2, cloud_fraction, climit2)
    return np.where(cloud_fraction <= climit1, incloud, incloud * rectified)


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


def _toa(ds, nudging):
    toa = ds[DSW_TOA] - ds[USW_TOA] - ds[ULW_TOA]
    return toa + ds[COL_T_NUDGE] if nudging else toa


def _up(ds):
    return ds[LHF] + ds[SHF] + ds[USW_SFC] + ds[ULW_SFC]


def apply(name, ds, **kw):
    """The registered transform ``name`` on a dict of arrays; returns the dict of its outputs."""
    if name in ("tapered_dQ1", "tapered_dQ2"):
        src = name[len("tapered_"):]
        scaling = vertical_tapering_scale_factors(ds[src].shape[0], kw["cutoff"], kw["rate"])
        return {name: scaling.reshape((-1,) + (1,) * (ds[src].ndim - 1)) * ds[src]}
    if name == "Qm_from_Q1_Q2":
        return {"Qm": moist_static_energy_tendency(ds["Q1"], ds["Q2"])}
    if name == "Q1_from_Qm_Q2":
        return {"Q1": temperature_tendency(ds["Qm"], ds["Q2"])}
    if name == "Qm_from_Q1_Q2_temperature_dependent":
        return {"Qm": moist_static_energy_tendency(ds["Q1"], ds["Q2"], ds["air_temperature"])}
    if name == "Q1_from_Qm_Q2_temperature_dependent":
        return {"Q1": temperature_tendency(ds["Qm"], ds["Q2"], ds["air_temperature"])}
    if name == "Q1_from_dQ1_pQ1":
        return {"Q1": ds["dQ1"] + ds["pQ1"]}
    if name == "Q2_from_dQ2_pQ2":
        return {"Q2": ds["dQ2"] + ds["pQ2"]}
    if name == "Qm_flux_from_Qm_tendency":
        flux, down = tendency_to_flux(ds["Qm"], _toa(ds, kw.get("include_temperature_nudging", True)), _up(ds), ds[DELP],
                                      kw.get("rectify_downward_radiative_flux", True))
        return {"Qm_flux": flux, "implied_downward_radiative_flux_at_surface": down}
    if name == "Q2_flux_from_Q2_tendency":
        flux, down = tendency_to_flux(ds["Q2"], np.zeros_like(ds[LHF]), latent_heat_flux_to_evaporation(ds[LHF]), ds[DELP],
                                      kw.get("rectify_surface_precipitation_rate", True))
        return {"Q2_flux": flux, "implied_surface_precipitation_rate": down}
    if name == "Qm_tendency_from_Qm_flux":
        return {"Qm": flux_to_tendency(ds["Qm_flux"], ds["implied_downward_radiative_flux_at_surface"], _up(ds), ds[DELP])}
    if name == "Q2_tendency_from_Q2_flux":
        return {"Q2": flux_to_tendency(ds["Q2_flux"], ds["implied_surface_precipitation_rate"], latent_heat_flux_to_evaporation(ds[LHF]), ds[DELP])}
    if name == "implied_downward_radiative_flux_at_surface":
        return {name: tendency_to_implied_surface_downward_flux(ds["Qm"], _toa(ds, kw.get("include_temperature_nudging", True)), _up(ds),
                                                                ds[DELP], kw.get("rectify", True))}
    if name == "implied_surface_precipitation_rate":
        return {name: tendency_to_implied_surface_downward_flux(ds["Q2"], np.zeros_like(ds[LHF]), latent_heat_flux_to_evaporation(ds[LHF]),
                                                                ds[DELP], kw.get("rectify", True))}
    if name == "cloud_water_mixing_ratio_from_incloud":
        return {"cloud_water_mixing_ratio": incloud_to_gridcell_condensate(ds["cloud_amount"], ds["incloud_water_mixing_ratio"])}
    if name == "cloud_ice_mixing_ratio_from_incloud":
        return {"cloud_ice_mixing_ratio": incloud_to_gridcell_condensate(ds["cloud_amount"], ds["incloud_ice_mixing_ratio"])}
    raise KeyError(name)


def taper_mask(score, cutoff=0):
    return np.where(score > cutoff, 0, 1)


def taper_ramp(score, ramp_min=0, ramp_max=1):
    return np.clip((ramp_max - score) / (ramp_max - ramp_min), 0, 1)


def taper_decay(score, threshold=0, rate=0.5):
    return np.minimum(rate ** (score - threshold), 1)


def minmax_score(scaled_x):
    """``scaled_x``: MinMaxScaler.transform's [sample, feature] output."""
    return np.maximum(scaled_x.max(axis=1) - 1, 0) + np.maximum(-1 * scaled_x.min(axis=1), 0)
