"""Importable alias for the product package.

The product lives in `controllable-latent-diffusion-for-traffic-simulation_amd/`
(the directory name the build contract asks for); hyphens make that name
un-importable with a plain `import`, so this shim package points its
`__path__` at that directory and runs its `__init__`.  `import cld_amd` and
`import cld_amd.dm_model` therefore load the files of the hyphenated package.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "controllable-latent-diffusion-for-traffic-simulation_amd")
__path__.insert(0, _PKG_DIR)
_init = _os.path.join(_PKG_DIR, "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
