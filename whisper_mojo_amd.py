"""Import shim: the package directory is named `whisper.mojo_amd/` (not an importable identifier), so
`import whisper_mojo_amd` loads it from there under this name."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "whisper.mojo_amd")
_spec = _u.spec_from_file_location("whisper_mojo_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["whisper_mojo_amd"] = _mod
_spec.loader.exec_module(_mod)
