"""Drop-in for the ``emulation`` package the Fortran physics driver calls through call_py_fort
(external/emulation/emulation/__init__.py:1-18): module-level ``gscond``, ``microphysics`` and
``store`` callables built at import time from ``./fv3config.yml``."""
import sys

if not hasattr(sys, "argv"):  # the embedded interpreter under call_py_fort has none
    sys.argv = [""]

from .config import EmulationConfig, ModelConfig, get_hooks  # noqa: E402
from .hook import MicrophysicsHook, always_emulator  # noqa: E402
from .models import HipEmulator  # noqa: E402

gscond, microphysics, store = get_hooks()
