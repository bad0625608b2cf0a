"""Import the reference's own, unmodified hot-path files from /root/reference.

Build-container only (the reference never travels to the GPU box); used by
`oracle/make_golden.py` to emit the fixtures under `tests/golden/` and by the
optional cross-check in `tests/test_oracle_vs_reference.py` (skipped when
/root/reference is absent).

The reference imports `torchvision`, `trajdata`, ... at module import time
although the sampling path never calls them.  A meta-path finder appended
AFTER the real finders fabricates inert stub modules for those absent
third-party names only, so real packages always win and no reference file is
modified or copied (recipe: SURVEY.md section 8(c)).
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import importlib.util
import io
import os
import sys
from contextlib import redirect_stdout
from unittest.mock import MagicMock

REF = os.environ.get("CLD_REFERENCE", "/root/reference")
_CANDIDATES = ("torchvision", "trajdata", "seaborn", "pytorch_lightning", "wandb", "zarr", "cv2",
               "shapely", "memory_profiler", "h5py", "l5kit", "pymap3d", "protobuf", "imageio", "numba")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "models", "dm"))


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self, absent):
        self.absent = set(absent)

    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in self.absent:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__ = []
        m.__spec__ = spec
        m.__name__ = spec.name
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REF}")
    sys.dont_write_bytecode = True
    absent = []
    for n in _CANDIDATES:
        try:
            if importlib.util.find_spec(n) is None:
                absent.append(n)
        except (ImportError, ValueError):
            absent.append(n)
    sys.meta_path.append(_StubFinder(absent))
    sys.path[:0] = [REF, os.path.join(REF, "src")]
    _installed = True


def load():
    """Returns a namespace with the reference classes/functions of the hot path."""
    install()
    import types

    import yaml
    ns = types.SimpleNamespace()
    with redirect_stdout(io.StringIO()):
        from configs.custom_config import ConfigBase, dict_to_config
        from models.dm.dm_model import DmModel
        from models.vae.lstm_vae import LSTMVAE
        from models.vae.vae_model import VaeModel
        import tbsim.dynamics as dynamics
        from tbsim.models.diffuser_helpers import unicyle_forward_dynamics
    ns.DmModel, ns.LSTMVAE, ns.VaeModel = DmModel, LSTMVAE, VaeModel
    ns.dynamics, ns.unicyle_forward_dynamics = dynamics, unicyle_forward_dynamics
    with open(os.path.join(REF, "config.yaml")) as f:
        ns.algo = dict_to_config(ConfigBase, yaml.safe_load(f)["algo"])
    return ns


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)
