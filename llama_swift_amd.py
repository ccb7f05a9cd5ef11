"""Import shim: the package directory is literally ``llama.swift_amd/`` (the reference's name plus
``_amd``), which is not an importable identifier; this module loads it under the name
``llama_swift_amd`` and replaces itself in ``sys.modules``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "llama.swift_amd")
_spec = importlib.util.spec_from_file_location(
    "llama_swift_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["llama_swift_amd"] = _mod
_spec.loader.exec_module(_mod)
