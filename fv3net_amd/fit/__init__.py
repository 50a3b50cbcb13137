"""``fv3fit``-shaped model API for the column MLP: Predictor ABC, the name-file io registry,
the dataset <-> array adapter, and the HIP dense predictor (registered as ``"hip-dense"``)."""
from . import io
from .io import dump, load
from .predictor import Predictor
from .stacking import SAMPLE_DIM_NAME, match_prediction_to_input_coords, stack
from .dense import DenseHyperparameters, HipDenseModel, spec_from_arrays, train_dense_model
from .testing import ConstantOutputPredictor
from .derived import DerivedMapping, DerivedModel
from .models import (CombinedOutputModel, EnsembleModel, SquashedOutputConfig, SquashedOutputModel, TaperConfig, TaperedModel,
                     vertical_tapering_scale_factors)

__all__ = [
    "CombinedOutputModel", "ConstantOutputPredictor", "DenseHyperparameters", "DerivedMapping", "DerivedModel", "EnsembleModel", "SquashedOutputConfig",
    "SquashedOutputModel", "TaperConfig", "TaperedModel", "vertical_tapering_scale_factors", "HipDenseModel", "Predictor", "SAMPLE_DIM_NAME", "dump", "io",
    "load", "match_prediction_to_input_coords", "spec_from_arrays", "stack", "train_dense_model",
]
