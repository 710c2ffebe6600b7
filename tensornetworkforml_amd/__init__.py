"""MI355X-native backend of the MPS two-site sweep optimiser.

The reference's drivers import its modules by bare name (`import Network_class as tn`,
training_diagonals.py:25-26) and its pickles name `Network_class.Network` / `Tensor_class.Tensor`.
Importing this package therefore registers its modules under those bare names as well, one module
object per name, so that such code (and `pickle.load` of the shipped models) resolves to this
backend.  If a different module of the same name is already imported (e.g. the reference itself)
the import fails instead of silently mixing the two.
"""
import importlib.util
import os
import sys

__version__ = '0.1'
_HERE = os.path.dirname(os.path.abspath(__file__))


def _load_bare(name):
    mod = sys.modules.get(name)
    if mod is not None:
        origin = os.path.abspath(getattr(mod, '__file__', '') or '')
        if os.path.dirname(origin) != _HERE:
            raise ImportError("a different module named %r is already imported from %s; "
                              "tensornetworkforml_amd cannot register its own under that name" % (name, origin))
        return mod
    spec = importlib.util.spec_from_file_location(name, os.path.join(_HERE, name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(name, None)
        raise
    return mod


for _n in ('Tensor_class', 'custom_linalg_tools', 'data_generator', 'Network_class'):
    _m = _load_bare(_n)
    sys.modules[__name__ + '.' + _n] = _m
    globals()[_n] = _m

from Network_class import Network          # noqa: E402
from Tensor_class import Tensor             # noqa: E402
from custom_linalg_tools import contract    # noqa: E402
