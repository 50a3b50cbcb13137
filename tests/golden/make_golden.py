"""Generates the golden fixtures under tests/golden/.  Run in the build container only
(needs /root/reference and oracle/_ref/libmappm_ref.so):

    python tests/golden/make_golden.py

What it writes (data only -- inputs and expected outputs, never reference source):

* mappm_reference.npz -- random and edge-case columns pushed through the REFERENCE's own
  mappm.f90 compiled by `make -C oracle ref` (oracle/_ref/libmappm_ref.so), with the outputs.
* coarsen_restarts_reference.npz -- the expected outputs held by the reference's own regression
  fixtures external/vcm/tests/_coarsen_restarts_regression_tests/reference/*.json, for the
  (tag, category, variable) triples the hot path covers.  The INPUTS are not stored: they are
  regenerated in the test by the reference's rule `np.random.seed(0);
  np.random.uniform(lo, hi, shape).astype(dtype)` (external/synth/synth/core.py:52-67) from the
  shapes/dtypes/ranges recorded here (schemas/*.json, test_coarsen_restarts.py:25-30).
* normalization_regtest.json -- the printed float32 arrays of
  external/fv3fit/tests/keras/_regtest_outputs/test_shared_utils.test_standard_{normalize,denormalize}.out
  (input arange(20).reshape(10, 2)).
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def make_mappm():
    from oracle import mappm_c

    assert mappm_c.have_reference(), "run `make -C oracle ref` first"
    rng = np.random.default_rng(2024)
    cases = {}
    n = 0

    def add(pe1, q, pe2, iv, kord):
        nonlocal n
        out = mappm_c.reference_mappm(pe1, q, pe2, iv, kord)
        cases[f"case{n}_pe1"], cases[f"case{n}_q1"], cases[f"case{n}_pe2"] = pe1, q, pe2
        cases[f"case{n}_ivkord"] = np.array([iv, kord])
        cases[f"case{n}_q2"] = out
        n += 1

    for km, kn, ncol in [(79, 79, 96), (63, 63, 32), (7, 7, 48), (4, 9, 16), (79, 40, 16), (20, 90, 16)]:
        for iv, kord in [(1, 1), (0, 1), (-1, 4), (2, 6), (0, 7), (-2, 7), (1, 3), (1, 5), (0, 2)]:
            dp1 = rng.uniform(300, 1500, (ncol, km))
            pe1 = np.concatenate([np.full((ncol, 1), 300.0), 300 + np.cumsum(dp1, 1)], 1).astype(np.float32)
            dp2 = rng.uniform(300, 1500, (ncol, kn))
            top = rng.choice([100.0, 300.0, 500.0])
            pe2 = np.concatenate([np.full((ncol, 1), top), 300 + np.cumsum(dp2, 1)], 1).astype(np.float32)
            q = rng.uniform(-1000, 1000, (ncol, km)).astype(np.float32)
            if iv == 0:
                q = np.abs(q)
            add(pe1, q, pe2, iv, kord)
    # ties, zero-thickness target layers, NaNs, flat fields
    for trial in range(4):
        ncol, km, kn = 64, 30, 30
        dp1 = rng.integers(1, 4, (ncol, km)).astype(np.float32)
        pe1 = np.concatenate([np.full((ncol, 1), 3.0, np.float32), 3 + np.cumsum(dp1, 1)], 1).astype(np.float32)
        dp2 = rng.integers(0, 4, (ncol, kn)).astype(np.float32)
        pe2 = np.concatenate([np.full((ncol, 1), float(rng.integers(0, 6)), np.float32), 3 + np.cumsum(dp2, 1)], 1).astype(np.float32)
        q = rng.uniform(-10, 10, (ncol, km)).astype(np.float32)
        if trial >= 1:
            q[rng.random((ncol, km)) < 0.05] = np.nan
        if trial >= 2:
            q = np.round(q)
        for iv, kord in [(1, 1), (0, 7), (-1, 4)]:
            add(pe1, q, pe2, iv, kord)
    # kord > 7: cs_profile / cs_limiters, every scheme, noise / smooth / tied / NaN-holed columns (its own generator, so that
    # the cases above stay what they were)
    rng = np.random.default_rng(2025)
    for kord in range(8, 19):
        for iv in (-1, 0, 1, 2):
            km, kn, ncol = [(79, 79, 24), (30, 41, 24), (7, 12, 16), (4, 9, 16)][(kord + iv) % 4]
            kind = ["noise", "smooth", "ties", "nans"][(kord + 2 * iv) % 4]
            dp1 = rng.uniform(300, 1500, (ncol, km))
            pe1 = np.concatenate([np.full((ncol, 1), 300.0), 300 + np.cumsum(dp1, 1)], 1).astype(np.float32)
            pe2 = np.concatenate([np.full((ncol, 1), rng.choice([100.0, 300.0, 500.0])), 300 + np.cumsum(rng.uniform(300, 1500, (ncol, kn)), 1)],
                                 1).astype(np.float32)
            z = np.linspace(0, 1, km)[None, :]
            if kind == "noise":
                q = rng.uniform(-1000, 1000, (ncol, km))
            elif kind == "smooth":
                q = 300 * np.sin(2 * np.pi * (z * rng.uniform(0.5, 3, (ncol, 1)) + rng.uniform(0, 1, (ncol, 1))))
                q = q + rng.normal(0, 1, (ncol, km)) * (rng.random((ncol, km)) < 0.1)
            else:
                q = np.round(rng.uniform(-3, 3, (ncol, km)))
                if kind == "nans":
                    q[rng.random((ncol, km)) < 0.03] = np.nan
            if iv == 0:
                q = np.abs(q) if kind != "smooth" else q + 250
            add(pe1, q.astype(np.float32), pe2, iv, kord)
    cases["n_cases"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "mappm_reference.npz"), **cases)
    print("mappm cases:", n)


def make_interpolate_2d():
    """Random and edge-case rows through the REFERENCE's interpolate_2d.f90 (compiled into
    oracle/_ref/libmappm_ref.so), with its outputs."""
    from oracle import mappm_c

    assert mappm_c.have_reference(), "run `make -C oracle ref` first"
    rng = np.random.default_rng(77)
    cases = {}
    for n, (m, n_in, n_out) in enumerate([(40, 79, 30), (16, 5, 9), (8, 2, 7), (12, 33, 1)]):
        x = np.cumsum(rng.uniform(0.1, 1, (m, n_in)), axis=1)
        if n_in > 6:
            x[3, 5] = x[3, 4]          # a repeated collocation point
        y = rng.normal(0, 1, (m, n_in))
        xp = rng.uniform(-0.5, x.max() + 0.5, (m, n_out))
        xp[:, 0] = x[:, 0]
        if n_out > 2:
            xp[:, 1] = x[:, -1]
            xp[:, 2] = x[:, n_in // 2]
        if n_out > 3:
            xp[0, 3] = np.nan
        cases[f"case{n}_xp"], cases[f"case{n}_x"], cases[f"case{n}_y"] = xp, x, y
        cases[f"case{n}_out"] = mappm_c.reference_interpolate_2d(xp, x, y)
    cases["n_cases"] = np.array(4)
    np.savez_compressed(os.path.join(HERE, "interpolate_2d_reference.npz"), **cases)
    print("interpolate_2d cases: 4")


# every configuration of the reference's regression test (test_coarsen_restarts.py:32-61): function, kwargs
CONFIGS = {
    "mass-weighted-model-level-with-agrid-winds": ("sigma", {"coarsen_agrid_winds": True, "mass_weighted": True}),
    "area-weighted-model-level-without-agrid-winds": ("sigma", {"coarsen_agrid_winds": False, "mass_weighted": False}),
    "pressure-level-with-agrid-winds": ("pressure", {"coarsen_agrid_winds": True, "extrapolate": False}),
    "pressure-level-without-agrid-winds": ("pressure", {"coarsen_agrid_winds": False}),
    "blended-mass-weighted-with-agrid-winds": ("blended", {"coarsen_agrid_winds": True}),
    "blended-area-weighted-without-agrid-winds": ("blended", {"coarsen_agrid_winds": False, "mass_weighted": False}),
    "pressure-level-extrapolate-with-agrid-winds": ("pressure", {"coarsen_agrid_winds": False, "extrapolate": True}),
}
CATEGORIES = ["fv_core.res", "fv_tracer.res", "fv_srf_wnd.res", "sfc_data"]


def make_coarsen_restarts():
    base = os.path.join(REF, "external/vcm/tests/_coarsen_restarts_regression_tests")
    arrays = {}
    meta = {"factor": 2, "toa_pressure": 300.0, "default_range": [-1000, 1000],
            "ranges": {"delp": [3, 5], "area": [0.5, 1], "dx": [0.5, 1], "dy": [0.5, 1]},
            "inputs": {}, "expected": []}
    for category in ["fv_core.res", "fv_tracer.res", "fv_srf_wnd.res", "sfc_data", "grid"]:
        with open(os.path.join(base, "schemas", f"{category}-schema.json")) as f:
            schema = json.load(f)["schema"]
        meta["inputs"][category] = {
            name: {"dims": v["dims"], "shape": v["array"]["shape"], "dtype": v["array"]["dtype"]}
            for name, v in schema["variables"].items()
        }
    meta["configs"] = {tag: {"method": m, "kwargs": kw} for tag, (m, kw) in CONFIGS.items()}
    sfc_seen = None
    for tag in CONFIGS:
        for category in CATEGORIES:
            with open(os.path.join(base, "reference", f"{tag}-{category}.json")) as f:
                ref = json.load(f)
            if category == "sfc_data":  # identical for every configuration: stored once
                blob = json.dumps({k: v["data"] for k, v in ref["data_vars"].items()}, sort_keys=True)
                if sfc_seen is None:
                    sfc_seen = blob
                else:
                    assert blob == sfc_seen, tag
                    continue
            for var, entry in ref["data_vars"].items():
                key = f"{tag}|{category}|{var}"
                arrays[key] = np.array(entry["data"], dtype=np.float64)
                meta["expected"].append({"key": key, "tag": tag, "category": category, "variable": var,
                                         "dims": entry["dims"]})
    arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "coarsen_restarts_reference.npz"), **arrays)
    print("coarsen-restarts expected arrays:", len(meta["expected"]))


def make_normalization():
    out = {}
    base = os.path.join(REF, "external/fv3fit/tests/keras/_regtest_outputs")
    for name in ("normalize", "denormalize"):
        text = open(os.path.join(base, f"test_shared_utils.test_standard_{name}.out")).read()
        first = text.split("array(")[1].split(", dtype")[0]
        vals = [float(x) for x in re.findall(r"-?\d+\.\d*(?:e-?\d+)?", first)]
        out[name] = np.array(vals, dtype=np.float32).reshape(10, 2).tolist()
    with open(os.path.join(HERE, "normalization_regtest.json"), "w") as f:
        json.dump({"input": "np.arange(20).reshape(10, 2).astype(np.float32), three identical variables", **out}, f)
    print("normalization regtest arrays written")


if __name__ == "__main__":
    make_mappm()
    make_interpolate_2d()
    make_coarsen_restarts()
    make_normalization()
