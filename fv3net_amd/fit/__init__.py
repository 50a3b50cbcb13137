"""``fv3fit``-shaped model API for the column MLP: Predictor ABC, the name-file io registry,
the dataset <-> array adapter, and the HIP dense predictor (registered as ``"hip-dense"``)."""
from . import io
from .io import dump, load
from .predictor import Predictor
from .stacking import SAMPLE_DIM_NAME, match_prediction_to_input_coords, stack
from .dense import DenseHyperparameters, HipDenseModel, spec_from_arrays, train_dense_model
from .testing import ConstantOutputNoveltyDetector, ConstantOutputPredictor
from .derived import DerivedMapping, DerivedModel
from .models import (CombinedOutputModel, EnsembleModel, SquashedOutputConfig, SquashedOutputModel, TaperConfig, TaperedModel,
                     vertical_tapering_scale_factors)
from .data_transform import DATA_TRANSFORM_REGISTRY, ChainedDataTransform, DataTransform
from .novelty import (MinMaxNoveltyDetector, NoveltyDetector, OCSVMNoveltyDetector, get_taper_function, taper_decay, taper_mask,
                      taper_ramp)
from .transformed import OutOfSampleModel, TransformedPredictor

__all__ = [
    "ChainedDataTransform", "ConstantOutputNoveltyDetector", "DATA_TRANSFORM_REGISTRY", "DataTransform", "MinMaxNoveltyDetector", "NoveltyDetector", "OCSVMNoveltyDetector",
    "OutOfSampleModel", "TransformedPredictor", "get_taper_function", "taper_decay", "taper_mask", "taper_ramp",
    "CombinedOutputModel", "ConstantOutputPredictor", "DenseHyperparameters", "DerivedMapping", "DerivedModel", "EnsembleModel", "SquashedOutputConfig",
    "SquashedOutputModel", "TaperConfig", "TaperedModel", "vertical_tapering_scale_factors", "HipDenseModel", "Predictor", "SAMPLE_DIM_NAME", "dump", "io",
    "load", "match_prediction_to_input_coords", "spec_from_arrays", "stack", "train_dense_model",
]
