"""Model io registry keyed by the ``name`` file of a model directory
(external/fv3fit/fv3fit/_shared/io.py:17-100): ``fv3fit.load(path)`` reads ``<path>/name`` and
dispatches to the class registered under that name; ``dump`` writes the name then the model."""
import os
import warnings
from functools import partial
from typing import Callable, MutableMapping, Type

from .predictor import Reloadable

_NAME_PATH = "name"
_NAME_ENCODING = "UTF-8"

# names of reference artifacts this build cannot read (TensorFlow SavedModels etc.)
UNREADABLE_REFERENCE_NAMES = ("all-keras", "all-keras-dict", "packed-keras", "sklearn", "sklearn_random_forest")


class _Register:
    def __init__(self) -> None:
        self._model_types: MutableMapping[str, Type[Reloadable]] = {}

    def __call__(self, name: str) -> Callable:
        if name in self._model_types:
            raise ValueError(f"{name} is already registered by {self._model_types[name]}.")
        return partial(self._register_class, name=name)

    def _register_class(self, cls, name: str):
        self._model_types[name] = cls
        return cls

    def _load_by_name(self, name: str, path: str) -> Reloadable:
        if name in UNREADABLE_REFERENCE_NAMES and name not in self._model_types:
            raise ValueError(
                f"model artifact of type '{name}' needs the reference's TensorFlow/sklearn stack to read; export it "
                "to the 'hip-dense' format (weights.npz + spec.yaml, see INTEGRATION.md) where that stack exists."
            )
        return self._model_types[name].load(path)

    def get_name(self, obj: Reloadable) -> str:
        return_name, name_cls = None, None
        for name, cls in self._model_types.items():
            if isinstance(obj, cls):
                if name_cls is None or issubclass(cls, name_cls):
                    return_name, name_cls = name, cls
        if return_name is None:
            raise ValueError(f"{type(obj)} is not registered. Consider decorating with @io.register(\"name\")")
        return return_name

    @staticmethod
    def _get_name_from_path(path: str) -> str:
        import fsspec

        return fsspec.get_mapper(path)[_NAME_PATH].decode(_NAME_ENCODING).strip()

    def _dump_class_name(self, obj: Reloadable, path: str):
        import fsspec

        fsspec.get_mapper(path)[_NAME_PATH] = self.get_name(obj).encode(_NAME_ENCODING)

    def load(self, path: str) -> Reloadable:
        """Load a serialized Reloadable from `path`."""
        try:
            name = self._get_name_from_path(path)
        except KeyError as e:
            warnings.warn(
                f"Model type is not located at {os.path.join(path, _NAME_PATH)}. Trying all known models one-by-one.",
                UserWarning,
            )
            for name in self._model_types:
                try:
                    return self._load_by_name(name, path)
                except Exception:  # noqa
                    pass
            raise e
        else:
            return self._load_by_name(name, path)

    def dump(self, obj: Reloadable, path: str):
        """Dump a Reloadable to a path"""
        os.makedirs(path, exist_ok=True) if "://" not in path else None
        self._dump_class_name(obj, path)
        obj.dump(path)


register = _Register()
dump = register.dump
load = register.load
