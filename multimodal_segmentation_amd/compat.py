"""The reference's TOP-LEVEL module names as aliases of this package's modules.

The reference is run from its repository root, so its callers import `model_components.decoder`, `models.dafnet`,
`model_executors.dafnet_executor`, `configuration.dafnet_config_chaos`, `layers.spade`, `loaders.loader_factory`,
`callbacks.swa`, `costs`, `utils.data_utils`, `model_tester` (experiment.py:113-124 resolves `'models.' + conf.model` and
`'model_executors.' + conf.executor` with importlib; models/dafnet.py:11-16 does `from model_components import ...`).

    import multimodal_segmentation_amd.compat as compat
    compat.install()
    from model_components import decoder               # IS multimodal_segmentation_amd.model_components.decoder
    importlib.import_module('models.dafnet').DAFNet    # what experiment.py:115-118 does

`install()` puts a finder at the END of `sys.meta_path` that answers exactly those names by importing the
package-qualified module and registering the same module object under the short name -- no second copy of any module
exists, so class identities, module-level caches and the kernel library handle are shared.  Because it is the LAST
finder, a real top-level module or package of one of these generic names (`utils`, `models`, ...) that the regular path
finders can locate keeps precedence: the aliases never shadow user or third-party code, they only answer names nobody
else provides.  (A second finder at the FRONT answers `<top>.<sub>` only when `<top>` already is one of these aliases -- the alias
shares the real package's `__path__`, and the path finder would otherwise load the sub-module a second time under the short name.)
Opt-in: nothing is claimed before `install()`; `uninstall()` removes the finders and the aliases.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

_PKG = __name__.rsplit('.', 1)[0]
TOP_LEVEL = ('model_components', 'models', 'model_executors', 'configuration', 'layers', 'loaders', 'callbacks', 'costs',
             'utils', 'model_tester')


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real_name):
        self.real_name = real_name

    def create_module(self, spec):
        real = importlib.import_module(self.real_name)      # the one and only module object
        self.real_spec = real.__spec__
        return real

    def exec_module(self, module):
        # already executed under its package-qualified name; importlib has just pointed __spec__ at the alias -- put the real
        # one back so that relative imports inside the module keep resolving against the package
        module.__spec__ = self.real_spec


def _alias_spec(fullname):
    real_name = _PKG + '.' + fullname
    try:
        real = importlib.import_module(real_name)
    except ModuleNotFoundError as exc:
        if exc.name == real_name:                        # no such module in the package: let the other finders answer
            return None
        raise
    return importlib.machinery.ModuleSpec(fullname, _AliasLoader(real_name), is_package=hasattr(real, '__path__'))


class _AliasFinder(importlib.abc.MetaPathFinder):
    """LAST on sys.meta_path: answers the TOP-LEVEL names only, and only when no regular finder located a module of that name"""

    def find_spec(self, fullname, path=None, target=None):
        if '.' in fullname or fullname not in TOP_LEVEL:
            return None
        return _alias_spec(fullname)


class _SubmoduleFinder(importlib.abc.MetaPathFinder):
    """FIRST on sys.meta_path, but it only answers `<top>.<sub>` when `<top>` in sys.modules IS one of this package's modules (an
    alias made by _AliasFinder): the alias shares the real package's __path__, so the regular path finder would otherwise load
    `<top>/<sub>.py` a second time under the short name (and its relative imports would fail).  Children of anybody else's
    `utils`, `models`, ... are never touched."""

    def find_spec(self, fullname, path=None, target=None):
        top = fullname.split('.', 1)[0]
        if '.' not in fullname or top not in TOP_LEVEL:
            return None
        parent = sys.modules.get(top)
        if parent is None or getattr(parent, '__name__', '') != _PKG + '.' + top:
            return None
        return _alias_spec(fullname)


_finder = _AliasFinder()
_sub_finder = _SubmoduleFinder()


def install():
    """claim the reference's top-level module names (idempotent)"""
    if _finder not in sys.meta_path:
        sys.meta_path.append(_finder)
    if _sub_finder not in sys.meta_path:
        sys.meta_path.insert(0, _sub_finder)
    return _finder


def uninstall():
    for f in (_finder, _sub_finder):
        if f in sys.meta_path:
            sys.meta_path.remove(f)
    for name in [n for n in sys.modules if n.split('.', 1)[0] in TOP_LEVEL]:
        mod = sys.modules[name]
        if getattr(mod, '__name__', '').startswith(_PKG + '.'):
            del sys.modules[name]
