"""Import shim: registers the directory ``midi-vae_amd/`` as the package ``midi_vae_amd``.

``import midi_vae_amd`` (this file) replaces itself in ``sys.modules`` with the real package, so
``from midi_vae_amd.packers import ...`` works although the directory name is not an identifier.
"""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "midi-vae_amd")
_spec = _u.spec_from_file_location("midi_vae_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_pkg = _u.module_from_spec(_spec)
_sys.modules["midi_vae_amd"] = _pkg
_spec.loader.exec_module(_pkg)
