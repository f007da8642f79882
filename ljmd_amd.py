"""Import alias: the package directory name required by the build contract
(`molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd/`) is not a valid
Python identifier, so `import ljmd_amd` loads that directory under this name."""
import importlib.util
import sys
from pathlib import Path

_PKG_DIR = Path(__file__).resolve().parent / "molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd"
_spec = importlib.util.spec_from_file_location(
    "ljmd_amd", _PKG_DIR / "__init__.py", submodule_search_locations=[str(_PKG_DIR)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ljmd_amd"] = _mod
_spec.loader.exec_module(_mod)
