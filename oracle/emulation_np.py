"""CPU restatement (numpy) of the Zhao-Carr emulator's post-processing -- TEST INFRASTRUCTURE ONLY
(imported by tests/ and nothing else).

Follows external/emulation/emulation/masks.py:23-76 (RangeMask, LevelMask) and
external/emulation/emulation/zhao_carr.py:60-344 (squash, conservation limits, phase-dependent latent
heat with the ice/water flag scan, strict TOA-to-surface precipitation scan, simple column budget,
classifier masks).  State and emulator are dicts of [feature(z), sample] arrays as the Fortran hook
passes them (emulation/_typing.py:4-8).  Pinned by the literal known answers of
external/emulation/tests/test_zhao_carr.py:15-147, test_mask.py:7-47 and the three regtest scalars
(tests/test_oracle_emulation.py).

One reference quirk is kept on purpose: ``ice_water_flag`` (zhao_carr.py:114-138) reads its input as
[n, z] and scans along the LAST axis, while ``apply_condensation_phase_dependent`` (:147-151) hands it
the [z, sample] state arrays unchanged -- so in the reference the scan runs along the sample axis.
"""
import numpy as np

GRAVITY = 9.80665      # zhao_carr.py:33-37 (physcons.f)
CP = 1.0046e3
LV = 2.5e6
RHO_WATER = 1000.0
HFUS = 3.3358e5

CLOUD_IN = "cloud_water_mixing_ratio_input"
QV_IN = "specific_humidity_input"
T_IN = "air_temperature_input"
DELP = "pressure_thickness_of_atmospheric_layer"
CLOUD_G, QV_G, T_G = (f"{v}_after_gscond" for v in ("cloud_water_mixing_ratio", "specific_humidity", "air_temperature"))
CLOUD_P, QV_P, T_P = (f"{v}_after_precpd" for v in ("cloud_water_mixing_ratio", "specific_humidity", "air_temperature"))
PRECIP = "total_precipitation"
CLASSES = ["negative_tendency", "positive_tendency", "zero_cloud", "zero_tendency"]  # sorted(CLASS_NAMES)


def range_mask(emulator, key, lo=None, hi=None):
    out = dict(emulator)
    if lo is not None:
        out[key] = np.maximum(out[key], lo)
    if hi is not None:
        out[key] = np.minimum(out[key], hi)
    return out


def level_mask(state, emulator, key, start, stop, fill_value=None):
    field = np.array(emulator[key], dtype=np.float64)  # the reference casts the emulator field to float64
    sl = slice(start, stop)
    if fill_value is None:
        field[sl] = state[key][sl]
    elif isinstance(fill_value, str):
        field[sl] = state[fill_value][sl]
    else:
        field[sl] = fill_value
    return {**emulator, key: field}


def squash(cloud, humidity, bound):
    cloud_out = np.where(cloud < bound, 0, cloud)
    return cloud_out, humidity + (cloud - cloud_out)


def infer_gscond_cloud_from_conservation(state, emulator):
    return {**emulator, CLOUD_G: state[CLOUD_IN] - (emulator[QV_G] - state[QV_IN])}


def limit_net_condensation(state, net):
    cond = np.where(net > 0, net, 0.0)
    evap = np.where(net < 0, net, 0.0)
    return np.maximum(evap, -state[CLOUD_IN]) + np.minimum(cond, state[QV_IN])


def ice_water_flag(t_celsius, cloud):
    n, z = t_celsius.shape
    iw = np.zeros_like(t_celsius, dtype=np.result_type(t_celsius, np.float32) if t_celsius.dtype.kind != "f" else t_celsius.dtype)
    for i in range(n):
        for k in range(z - 1, -1, -1):
            t = t_celsius[i, k]
            if t < -15:
                iw[i, k] = 1.0
            elif t > 0.0:
                iw[i, k] = 0.0
            elif k < z - 1 and iw[i, k + 1] == 1 and cloud[i, k] > 1e-20:
                iw[i, k] = 1.0
    return iw


def latent_heat_phase_dependent(iw):
    return LV + iw * HFUS


def apply_condensation(state, net, lv):
    return {CLOUD_G: state[CLOUD_IN] + net, QV_G: state[QV_IN] - net, T_G: state[T_IN] + lv * net / CP}


def update_with_net_condensation(cloud_out, state, emulator, phase_dependent=False):
    net = limit_net_condensation(state, cloud_out - state[CLOUD_IN])
    if phase_dependent:
        lv = latent_heat_phase_dependent(ice_water_flag(state[T_IN] - 273.16, state[CLOUD_IN]))
    else:
        lv = LV
    return {**emulator, **apply_condensation(state, net, lv)}


def classify(logits):
    one_hot = logits == np.max(logits, axis=0, keepdims=True)
    return {name: one_hot[i] for i, name in enumerate(CLASSES)}


def gscond_cloud_choice(state, emulator, mode):
    """The cloud the gscond masks put in before conservation (zhao_carr.py:180-246)."""
    c = emulator[CLOUD_G]
    if mode == "fortran_vanishes":
        return np.where(state[CLOUD_G] < 1e-15, 0, c)
    if mode == "fortran_identical":
        return np.where(state[CLOUD_G] == state[CLOUD_IN], state[CLOUD_IN], c)
    if mode == "class_zero_cloud":
        return np.where(classify(emulator["gscond_classes"])["zero_cloud"], 0, c)
    if mode == "class_zero_tend":
        return np.where(classify(emulator["gscond_classes"])["zero_tendency"], state[CLOUD_IN], c)
    return c


def mask_zero_cloud_classifier_precpd(state, emulator):
    return {**emulator, CLOUD_P: np.where(classify(emulator["precpd_classes"])["zero_cloud"], 0, emulator[CLOUD_P])}


def strict_precip_scan(c_to_p, p_to_v):
    lim_c = np.maximum(c_to_p, 0)
    lim_v = np.maximum(p_to_v, 0)
    total = np.zeros(p_to_v.shape[1])
    for k in range(p_to_v.shape[0] - 1, -1, -1):
        total = total + lim_c[k]
        evap = np.minimum(total, lim_v[k])
        total = total - evap
        lim_v[k, :] = evap
    return lim_c, lim_v, total


def enforce_conservative_precpd(state, emulator):
    delp = state[DELP]
    src = -1 * (emulator[CLOUD_P] - state[CLOUD_G]) * delp / GRAVITY
    sink = (emulator[QV_P] - state[QV_G]) * delp / GRAVITY
    src_l, sink_l, total = strict_precip_scan(src, sink)
    evap = sink_l / delp * GRAVITY
    return {**emulator,
            CLOUD_P: state[CLOUD_G] + (-1 * src_l) / delp * GRAVITY,
            QV_P: state[QV_G] + evap,
            T_P: state[T_G] + LV / CP * -1 * evap,
            PRECIP: total / RHO_WATER}


def conservative_precip_simple(state, emulator):
    delp = state[DELP]
    before = np.sum((state[QV_G] + state[CLOUD_G]) * delp / GRAVITY, axis=0)
    after = np.sum((emulator[QV_P] + emulator[CLOUD_P]) * delp / GRAVITY, axis=0)
    return {**emulator, PRECIP: (before - after) / RHO_WATER}
