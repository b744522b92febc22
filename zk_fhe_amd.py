"""Import shim: the package directory is `zk-fhe_amd/` (hyphen, mirrors the reference's name), which
Python cannot import by name.  `import zk_fhe_amd` loads it under this module name."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zk-fhe_amd")
_spec = importlib.util.spec_from_file_location("zk_fhe_amd", os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["zk_fhe_amd"] = _mod
_spec.loader.exec_module(_mod)
