"""Top-level module names of the reference tree (`networks.*`, `utils.*`) for callers that put THIS directory on
`sys.path` the way image-matching-toolbox puts `third_party/patch2pix` there:

    sys.path.append('<repo>/patch2pix_amd')
    from utils.eval.model_helper import load_model, estimate_matches
    from networks.patch2pix import Patch2Pix

`networks/__init__.py` and `utils/__init__.py` notice that they are being imported under the bare name, load the real
package (`patch2pix_amd`, by file location -- `sys.path` is not touched) and call `install()`.  From then on
`networks[.x]` / `utils[.x]` are the SAME module objects as `patch2pix_amd.networks[.x]` / `patch2pix_amd.utils[.x]`
(one copy of every module, one load of libp2p_hip.so), so both spellings can be mixed in one process.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

_ROOTS = ("networks", "utils")


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real):
        self.real = real
        self.real_spec = real.__spec__

    def create_module(self, spec):
        return self.real

    def exec_module(self, module):      # already executed under its real name; the import system has just
        module.__spec__ = self.real_spec   # re-pointed __spec__ at the alias -- put the real one back


class _AliasFinder(importlib.abc.MetaPathFinder):
    """Answers only for SUBMODULES of `networks` / `utils`, and only while the bare top-level name is bound to this
    package's module: the top-level names themselves are found by the ordinary path finder (whose __init__.py calls
    `bootstrap`), so another project's `utils` package is importable again as soon as sys.modules['utils'] is not ours
    (e.g. after `uninstall()` or `del sys.modules['utils']`)."""

    def find_spec(self, fullname, path=None, target=None):
        head, _, rest = fullname.partition(".")
        if head not in _ROOTS or not rest:
            return None
        if getattr(sys.modules.get(head), "__name__", None) != "patch2pix_amd." + head:
            return None
        try:
            real = importlib.import_module("patch2pix_amd." + fullname)
        except ModuleNotFoundError as e:
            if e.name == "patch2pix_amd." + fullname:
                return None
            raise
        spec = importlib.machinery.ModuleSpec(fullname, _AliasLoader(real), is_package=hasattr(real, "__path__"))
        return spec


def install():
    if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder())


def uninstall():
    """Remove the finder and every bare alias from sys.modules (the patch2pix_amd.* modules stay loaded)."""
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, _AliasFinder)]
    for name in list(sys.modules):
        if name.partition(".")[0] in _ROOTS and getattr(sys.modules[name], "__name__", "").startswith("patch2pix_amd."):
            del sys.modules[name]


def bootstrap(bare_name, init_file):
    """Called from networks/__init__.py / utils/__init__.py when imported as a top-level package."""
    import importlib.util
    import os
    if "patch2pix_amd" not in sys.modules:
        pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(init_file)))
        spec = importlib.util.spec_from_file_location("patch2pix_amd", os.path.join(pkg_dir, "__init__.py"),
                                                      submodule_search_locations=[pkg_dir])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["patch2pix_amd"] = mod
        spec.loader.exec_module(mod)
    install()
    real = importlib.import_module("patch2pix_amd." + bare_name)
    sys.modules[bare_name] = real
    return real
