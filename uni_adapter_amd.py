"""Import shim: the package directory is `uni-adapter_amd/` (not a valid Python identifier), so
`import uni_adapter_amd` loads that directory as the package `uni_adapter_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'uni-adapter_amd')
_spec = importlib.util.spec_from_file_location(
    'uni_adapter_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules['uni_adapter_amd'] = _pkg
_spec.loader.exec_module(_pkg)
