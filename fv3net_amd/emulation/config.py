"""Configuration of the online hooks (external/emulation/emulation/config.py:120-324), reduced
to what the hot path implements: a regressor at ``path`` in the ``HipEmulator`` format.  The
reference's post-processing options (range / level masks, Zhao-Carr conservation fixes, the
classifier, the zarr monitor) are not implemented yet and are rejected loudly rather than
silently ignored."""
import dataclasses
import logging
from typing import Optional

import yaml

from .hook import MicrophysicsHook
from .models import HipEmulator

logger = logging.getLogger("emulation")

_UNIMPLEMENTED = (
    "classifier_path", "tensor_transform", "ranges", "mask_emulator_levels", "cloud_squash", "gscond_cloud_conservative",
    "mask_gscond_identical_cloud", "mask_gscond_zero_cloud", "enforce_conservative", "enforce_conservative_phase_dependent",
    "mask_gscond_zero_cloud_classifier", "mask_gscond_no_tend_classifier", "mask_precpd_zero_cloud_classifier",
    "enforce_strict_precpd_conservative", "simple_precip_conservative", "online_schedule",
)


def do_nothing(state):
    pass


@dataclasses.dataclass
class ModelConfig:
    """``path``: directory holding ``spec.yaml`` + ``weights.npz``.  ``batch_size`` is accepted for
    compatibility and ignored: the fused kernel takes all columns of a call at once."""

    path: Optional[str] = None
    batch_size: int = 512

    @staticmethod
    def from_dict(d: dict) -> "ModelConfig":
        bad = [k for k in d if k in _UNIMPLEMENTED and d[k] not in (None, False, {}, [])]
        if bad:
            raise NotImplementedError(f"zhao_carr_emulation options not implemented on the device yet: {bad}")
        unknown = [k for k in d if k not in ("path", "batch_size") and k not in _UNIMPLEMENTED]
        if unknown:
            raise ValueError(f"unknown ModelConfig keys: {unknown}")
        return ModelConfig(path=d.get("path"), batch_size=int(d.get("batch_size", 512)))

    def build(self) -> MicrophysicsHook:
        if self.path:
            model = HipEmulator.load(self.path)
        else:
            def model(x):
                return x
        return MicrophysicsHook(model=model)


@dataclasses.dataclass
class EmulationConfig:
    model: Optional[ModelConfig] = None
    gscond: Optional[ModelConfig] = None

    @staticmethod
    def _build_model(model: Optional[ModelConfig]):
        if model is None:
            logger.info("No model configured.")
            return do_nothing
        return model.build().microphysics

    def build_model_hook(self):
        return self._build_model(self.model)

    def build_gscond_hook(self):
        return self._build_model(self.gscond)

    def build_storage_hook(self):
        return do_nothing

    @staticmethod
    def from_dict(dict_: dict) -> "EmulationConfig":
        unknown = [k for k in dict_ if k not in ("model", "gscond", "storage")]
        if unknown:
            raise ValueError(f"unknown zhao_carr_emulation keys: {unknown}")
        if dict_.get("storage"):
            raise NotImplementedError("the storage hook (zarr/netCDF monitor) is outside this build")
        return EmulationConfig(
            model=ModelConfig.from_dict(dict_["model"]) if dict_.get("model") else None,
            gscond=ModelConfig.from_dict(dict_["gscond"]) if dict_.get("gscond") else None,
        )


def get_hooks(path: str = "fv3config.yml"):
    """(gscond, microphysics, store) callables built from ``./fv3config.yml`` (config.py:309-324)."""
    config_key = "zhao_carr_emulation"
    try:
        with open(path) as f:
            dict_ = yaml.safe_load(f) or {}
    except FileNotFoundError:
        logging.warning("Config not found...using defaults.")
        dict_ = {}
    config = EmulationConfig.from_dict(dict_.get(config_key, {}) or {})
    return config.build_gscond_hook(), config.build_model_hook(), config.build_storage_hook()
