"""Import shim: makes the hyphenated directory ``quickvc-official_amd/`` importable as the
package ``quickvc_official_amd`` (``import quickvc_official_amd`` from the repo root)."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quickvc-official_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_module = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _module
_spec.loader.exec_module(_module)
