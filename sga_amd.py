"""Import alias: the product package lives in the directory
`improving-inference-for-neural-image-compression_amd/` (name fixed by the repo layout, not a
valid Python identifier).  `import sga_amd` loads that directory as the package `sga_amd`."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "improving-inference-for-neural-image-compression_amd")
_spec = importlib.util.spec_from_file_location(
    "sga_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sga_amd"] = _mod
_spec.loader.exec_module(_mod)
