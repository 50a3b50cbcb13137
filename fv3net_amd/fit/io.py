"""Saving and loading predictors by the ``name`` file of a model directory.

The contract kept from the reference (external/fv3fit/fv3fit/_shared/io.py:17-100; ``fv3fit.load`` is what
``machine_learning.py:156`` calls): a model directory holds a UTF-8 file ``name`` whose content selects the class that
reads the rest of the directory; ``@register("name")`` binds a class (with ``dump(path)`` / ``load(path)``) to such a name;
``dump(obj, path)`` writes the name and then the object; ``load(path)`` dispatches on it.  Everything else here is this
package's own: a plain table and three functions.
"""
import os

NAME_FILE = "name"

# directory name -> class, in registration order
_TABLE = {}

# artifact types of the reference that need its TensorFlow / sklearn stack to be read
_FOREIGN = frozenset({"all-keras", "all-keras-dict", "packed-keras", "sklearn", "sklearn_random_forest"})


def register(name):
    """Class decorator: objects of the class are saved to and loaded from directories named ``name``."""
    if name in _TABLE:
        raise ValueError(f"{name} is already registered by {_TABLE[name]}.")

    def bind(cls):
        _TABLE[name] = cls
        return cls

    return bind


def registered_name(obj):
    """The name ``obj`` is saved under: that of the most derived registered class it is an instance of."""
    best = None
    for name, cls in _TABLE.items():
        if isinstance(obj, cls) and (best is None or issubclass(cls, _TABLE[best])):
            best = name
    if best is None:
        raise ValueError(f"{type(obj)} is not registered. Consider decorating with @io.register(\"name\")")
    return best


def _open(path, leaf, mode):
    """Local directories directly; anything with a protocol (gs://, memory://) through fsspec when it is installed."""
    if "://" not in str(path):
        return open(os.path.join(path, leaf), mode)
    import fsspec

    return fsspec.open(str(path).rstrip("/") + "/" + leaf, mode).open()


def dump(obj, path):
    """Write ``<path>/name`` and let the object write the rest."""
    name = registered_name(obj)
    if "://" not in str(path):
        os.makedirs(path, exist_ok=True)
    with _open(path, NAME_FILE, "wb") as f:
        f.write(name.encode("utf-8"))
    obj.dump(path)


def load(path):
    """Read ``<path>/name`` and hand the directory to the class registered under it.  A directory without the file is
    offered to every registered class in turn (the reference does the same for artifacts older than the name file)."""
    try:
        with _open(path, NAME_FILE, "rb") as f:
            name = f.read().decode("utf-8").strip()
    except (FileNotFoundError, KeyError) as missing:
        for cls in _TABLE.values():
            try:
                return cls.load(path)
            except Exception:  # noqa: BLE001  (not this class's artifact)
                continue
        raise FileNotFoundError(f"no '{NAME_FILE}' file in {path} and no registered class can read it") from missing
    if name not in _TABLE:
        if name in _FOREIGN:
            raise ValueError(
                f"model artifact of type '{name}' needs the reference's TensorFlow/sklearn stack to read; export it "
                "to the 'hip-dense' format (weights.npz + spec.yaml, see INTEGRATION.md) where that stack exists."
            )
        raise ValueError(f"unknown model type '{name}' in {path}; registered: {sorted(_TABLE)}")
    return _TABLE[name].load(path)
