"""Predictor interface (external/fv3fit/fv3fit/_shared/predictor.py:13-95)."""
import abc
from typing import Hashable, Iterable


class Dumpable(abc.ABC):
    @abc.abstractmethod
    def dump(self, path: str) -> None:
        """Serialize to a directory."""


class Loadable(abc.ABC):
    @classmethod
    def load(cls, path: str):
        """Load from a directory."""
        ...


class Reloadable(Dumpable, Loadable):
    pass


class Predictor(Reloadable):
    """Has ``predict(X: Dataset) -> Dataset`` over the variables named by ``input_variables`` /
    ``output_variables``; can be dumped to and loaded from a directory."""

    def __init__(self, input_variables: Iterable[Hashable], output_variables: Iterable[Hashable], **kwargs):
        super().__init__()
        if len(kwargs.keys()) > 0:
            raise TypeError(f"received unexpected keyword arguments: {tuple(kwargs.keys())}")
        self.input_variables = input_variables
        self.output_variables = output_variables

    @abc.abstractmethod
    def predict(self, X):
        """Predict an output dataset from an input dataset."""

    @abc.abstractmethod
    def dump(self, path: str) -> None:
        """Serialize to a directory."""

    @classmethod
    @abc.abstractmethod
    def load(cls, path: str) -> "Predictor":
        """Load a serialized model from a directory."""

    def input_sensitivity(self, stacked_sample):
        raise NotImplementedError(
            f"input_sensitivity is not implemented for Predictor subclass {self.__class__.__name__}."
        )
