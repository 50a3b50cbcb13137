"""Configuration of the online hooks (external/emulation/emulation/config.py:120-324): a regressor
at ``path`` in the ``HipEmulator`` format plus the reference's post-processing options -- range and
level masks, cloud squashing and the Zhao-Carr conservation fixes -- composed in the reference's
order (config.py:175-221) and run on the device.  ``path`` / ``classifier_path`` name saved "dense" or
"dense-local" emulators (``models.load_emulator``); their tensor transforms are part of the saved
model, as they are part of the reference's SavedModel graph.  Config-level tensor transforms are not part of this build and are rejected loudly rather than silently ignored; a ``storage``
section (the zarr / netCDF monitor) is accepted with a warning and its hook does nothing."""
import dataclasses
import logging
import os
from typing import Dict, Iterable, List, Mapping, Optional

import yaml

from . import zhao_carr
from .hook import MicrophysicsHook
from .masks import LevelMask, Mask, RangeMask, compose_masks
from .models import combine_classifier_and_regressor, load_emulator
from .schedule import IntervalSchedule, TimeMask

logger = logging.getLogger("emulation")

_UNIMPLEMENTED = ()
_FLAGS = (
    "gscond_cloud_conservative", "mask_gscond_identical_cloud", "mask_gscond_zero_cloud", "enforce_conservative",
    "enforce_conservative_phase_dependent", "mask_gscond_zero_cloud_classifier", "mask_gscond_no_tend_classifier",
    "mask_precpd_zero_cloud_classifier", "enforce_strict_precpd_conservative", "simple_precip_conservative",
)


@dataclasses.dataclass
class Range:
    min: Optional[float] = None
    max: Optional[float] = None


@dataclasses.dataclass
class LevelSlice:
    start: Optional[int] = None
    stop: Optional[int] = None
    fill_value: Optional[object] = None


def do_nothing(state):
    pass


@dataclasses.dataclass
class ModelConfig:
    """``path``: directory holding ``spec.yaml`` + ``weights.npz``.  ``batch_size`` is accepted for
    compatibility and ignored: the fused kernel takes all columns of a call at once.  The other
    attributes are the reference's (config.py:62-136)."""

    path: Optional[str] = None
    classifier_path: Optional[str] = None
    online_schedule: Optional[object] = None  # IntervalSchedule, or any callable time -> weight of the Fortran physics
    ranges: Mapping[str, Range] = dataclasses.field(default_factory=dict)
    mask_emulator_levels: Mapping[str, LevelSlice] = dataclasses.field(default_factory=dict)
    cloud_squash: Optional[float] = None
    gscond_cloud_conservative: bool = False
    mask_gscond_identical_cloud: bool = False
    mask_gscond_zero_cloud: bool = False
    enforce_conservative: bool = False
    enforce_conservative_phase_dependent: bool = False
    mask_gscond_zero_cloud_classifier: bool = False
    mask_gscond_no_tend_classifier: bool = False
    mask_precpd_zero_cloud_classifier: bool = False
    enforce_strict_precpd_conservative: bool = False
    simple_precip_conservative: bool = False
    batch_size: int = 512
    tensor_transform: List[object] = dataclasses.field(default_factory=list)  # emulation.transforms objects (config.py:120)

    def __post_init__(self):
        if self.enforce_conservative and self.enforce_conservative_phase_dependent:
            raise ValueError("These options are mutually exclusive.")
        if self.enforce_strict_precpd_conservative and self.simple_precip_conservative:
            raise ValueError("Conservative precip flags should not both be true.")

    @staticmethod
    def from_dict(d: dict) -> "ModelConfig":
        bad = [k for k in d if k in _UNIMPLEMENTED and d[k] not in (None, False, {}, [])]
        if bad:
            raise NotImplementedError(f"zhao_carr_emulation options not implemented on the device yet: {bad}")
        known = {f.name for f in dataclasses.fields(ModelConfig)} | set(_UNIMPLEMENTED)
        unknown = [k for k in d if k not in known]
        if unknown:
            raise ValueError(f"unknown ModelConfig keys: {unknown}")
        kwargs: Dict[str, object] = {k: bool(d[k]) for k in _FLAGS if k in d}
        if d.get("cloud_squash") is not None:
            kwargs["cloud_squash"] = float(d["cloud_squash"])
        kwargs["ranges"] = {k: Range(**v) for k, v in (d.get("ranges") or {}).items()}
        kwargs["mask_emulator_levels"] = {k: LevelSlice(**v) for k, v in (d.get("mask_emulator_levels") or {}).items()}
        if d.get("online_schedule"):
            kwargs["online_schedule"] = IntervalSchedule.from_dict(d["online_schedule"])
        from .transforms import transform_from_dict

        kwargs["tensor_transform"] = [transform_from_dict(e) for e in (d.get("tensor_transform") or [])]
        return ModelConfig(path=d.get("path"), classifier_path=d.get("classifier_path"), batch_size=int(d.get("batch_size", 512)), **kwargs)

    def build(self) -> MicrophysicsHook:
        if self.path:
            regressor = load_emulator(self.path)
            classifier = load_emulator(self.classifier_path) if self.classifier_path is not None else None
            model = combine_classifier_and_regressor(classifier, regressor, self.batch_size)
        else:
            def model(x):
                return x
        if self.tensor_transform:  # config.py:145-161: forward on the inputs, the model, backward on everything
            from .models import transform_model
            from .transforms import ComposedTransform

            model = transform_model(model, ComposedTransform(self.tensor_transform))
        return MicrophysicsHook(model=model, mask=self._build_mask())

    def _build_mask(self) -> Mask:
        return compose_masks(self._build_masks())

    def _build_masks(self) -> Iterable[Mask]:
        """The reference's order (config.py:175-221)."""
        if self.online_schedule:
            yield TimeMask(self.online_schedule)
        for key, rng in self.ranges.items():
            yield RangeMask(key, min=rng.min, max=rng.max)
        if self.gscond_cloud_conservative:
            yield zhao_carr.infer_gscond_cloud_from_conservation
        if self.cloud_squash is not None:
            yield lambda x, y: zhao_carr.squash_gscond(x, y, self.cloud_squash)
            yield lambda x, y: zhao_carr.squash_precpd(x, y, self.cloud_squash)
        if self.mask_gscond_identical_cloud:
            yield zhao_carr.mask_where_fortran_cloud_identical
        if self.mask_gscond_zero_cloud:
            yield zhao_carr.mask_where_fortran_cloud_vanishes_gscond
        if self.mask_gscond_no_tend_classifier:
            yield zhao_carr.mask_zero_tend_classifier
        if self.mask_gscond_zero_cloud_classifier:
            yield zhao_carr.mask_zero_cloud_classifier
        if self.mask_precpd_zero_cloud_classifier:
            yield zhao_carr.mask_zero_cloud_classifier_precpd
        if self.enforce_conservative:
            yield zhao_carr.enforce_conservative_gscond
        elif self.enforce_conservative_phase_dependent:
            yield zhao_carr.enforce_conservative_phase_dependent
        if self.simple_precip_conservative:
            yield zhao_carr.conservative_precip_simple
        elif self.enforce_strict_precpd_conservative:
            yield zhao_carr.enforce_conservative_precpd
        for key, sl in self.mask_emulator_levels.items():
            yield LevelMask(key, start=sl.start, stop=sl.stop, fill_value=sl.fill_value)


@dataclasses.dataclass
class EmulationConfig:
    model: Optional[ModelConfig] = None
    gscond: Optional[ModelConfig] = None
    storage: Optional[object] = None  # monitor.StorageConfig (config.py:224-229 of the reference)

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
        if self.storage is None:
            return do_nothing
        from .monitor import StorageHook

        return StorageHook.from_config(self.storage, n_ranks=int(os.environ.get("FV3NET_AMD_N_RANKS", "1"))).store

    @staticmethod
    def from_dict(dict_: dict) -> "EmulationConfig":
        from .monitor import StorageConfig

        unknown = [k for k in dict_ if k not in ("model", "gscond", "storage")]
        if unknown:
            raise ValueError(f"unknown zhao_carr_emulation keys: {unknown}")
        storage = None
        if dict_.get("storage"):
            known = {f.name for f in dataclasses.fields(StorageConfig)}
            bad = [k for k in dict_["storage"] if k not in known]
            if bad:
                raise ValueError(f"unknown zhao_carr_emulation.storage keys: {bad}")
            storage = StorageConfig(**dict_["storage"])
        return EmulationConfig(
            model=ModelConfig.from_dict(dict_["model"]) if dict_.get("model") else None,
            gscond=ModelConfig.from_dict(dict_["gscond"]) if dict_.get("gscond") else None,
            storage=storage,
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
