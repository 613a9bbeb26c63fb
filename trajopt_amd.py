"""Import shim: the package directory is ``trajectoryoptimization.jl_amd/`` (a dot is not a legal
module name), so it is loaded under the module name ``trajectoryoptimization_jl_amd`` and re-exported
here as ``trajopt_amd``."""
import importlib.util
import sys
from pathlib import Path

_NAME = "trajectoryoptimization_jl_amd"
if _NAME not in sys.modules:
    _dir = Path(__file__).resolve().parent / "trajectoryoptimization.jl_amd"
    _spec = importlib.util.spec_from_file_location(_NAME, _dir / "__init__.py", submodule_search_locations=[str(_dir)])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    _spec.loader.exec_module(_mod)
_pkg = sys.modules[_NAME]
globals().update({k: v for k, v in vars(_pkg).items() if not k.startswith("__")})
capi = _pkg.capi
internal = _pkg.internal
