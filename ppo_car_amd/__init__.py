"""Import shim: the implementation lives in the directory `ppo-car_amd/` (the name the project
layout prescribes, which is not a valid Python identifier).  This package only extends its
search path to that directory, so `import ppo_car_amd.env` loads `ppo-car_amd/env.py`."""
import os as _os

_impl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ppo-car_amd")
if not _os.path.isdir(_impl):
    raise ImportError(f"ppo_car_amd: implementation directory missing: {_impl}")
__path__.append(_impl)

from ._capi import PpoCarError, lib_path  # noqa: E402,F401
from .env import Track, VecCarEnv  # noqa: E402,F401
from .buffer import Buffer  # noqa: E402,F401
from .model import Agent, PolicyRangeError, layer_init  # noqa: E402,F401
