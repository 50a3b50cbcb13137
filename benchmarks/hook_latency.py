"""Wall time of one MicrophysicsHook.microphysics(state) call as the Fortran model makes it: float64 [79, ncol] numpy
arrays in, the state updated in place (host -> device, network, masks, device -> host), for the column counts of one
rank: 2 304 (a 48 x 48 subdomain) and 9 216 (96 x 96)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import bench
from fv3net_amd.emulation import HipEmulator, MicrophysicsHook
from fv3net_amd.emulation.config import ModelConfig
from fv3net_amd.mlp import ResidualSpec

spec = bench.zc_spec(0)
spec.residuals = [
    ResidualSpec("air_temperature_after_precpd", "air_temperature_input", "temperature_precpd_difference"),
    ResidualSpec("specific_humidity_after_precpd", "specific_humidity_input", "humidity_precpd_difference"),
    ResidualSpec("cloud_water_mixing_ratio_after_precpd", "cloud_water_mixing_ratio_input", "cloud_precpd_difference"),
]
cfg = ModelConfig.from_dict({"cloud_squash": 1e-8, "ranges": {"total_precipitation": {"min": 0.0}},
                             "mask_emulator_levels": {"air_temperature_after_precpd": {"start": 74, "fill_value": "air_temperature_input"}}})
hook = MicrophysicsHook(model=HipEmulator(spec), mask=cfg._build_mask())
for n in (2304, 9216):
    src = bench.zc_inputs_numpy(np.random.default_rng(5), n)
    base = {k: np.ascontiguousarray(v.T.astype(np.float64)) for k, v in src.items()}
    base["model_time"], base["rank"] = [2016, 8, 1, 0, 15, 0], 0
    for _ in range(3):
        hook.microphysics(dict(base))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        hook.microphysics(dict(base))
    print(f"ncol={n}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per hook call "
          f"({sum(v.nbytes for v in base.values() if isinstance(v, np.ndarray)) / 1e6:.1f} MB of float64 state in)", flush=True)
