"""Import alias: the package lives in the directory `robust-nonlinear-mpc_amd/` (hyphenated, as the
project is named); Python cannot import a hyphenated name, so this stub points `__path__` at it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "robust-nonlinear-mpc_amd")]
__path__[0] = _os.path.normpath(__path__[0])
from ._init import *  # noqa: F401,F403,E402
