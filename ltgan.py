"""Import alias for the package directory `long-tail-gan_amd/` (a hyphen is not importable).

`import ltgan` yields that package; `import ltgan.engine`, `from ltgan import generator` work.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "long-tail-gan_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
