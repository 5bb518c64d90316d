"""Import shim: exposes the package directory ``link-prediction-gnn_amd/`` as ``lpgnn_amd``
(the directory name required by the layout contract contains a hyphen)."""
import importlib.util
import pathlib
import sys

_dir = pathlib.Path(__file__).resolve().parent / "link-prediction-gnn_amd"
_spec = importlib.util.spec_from_file_location(__name__, _dir / "__init__.py",
                                               submodule_search_locations=[str(_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
