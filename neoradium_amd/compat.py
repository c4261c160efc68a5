"""Opt-in import alias: lets ``from neoradium import Carrier, PDSCH, ...`` resolve to this build, so the Playground
notebooks run with their import lines unchanged (reference neoradium/__init__.py:4-21 is the name list mirrored here).

    import neoradium_amd.compat; neoradium_amd.compat.install()        # at the top of a notebook / script
    python -m neoradium_amd.compat script.py [args ...]                 # or: run a script with the alias installed

``install()`` registers ``sys.modules['neoradium']`` and every ``neoradium.<submodule>`` this build has (``utils``,
``random``, ``harq``, ``ldpc``, ``pdsch``, ...) as THE SAME module objects as ``neoradium_amd[.<submodule>]`` -- an alias,
not a copy, so class identities, the global ``random`` generator and ``isinstance`` checks are shared.  It refuses to
shadow a real ``neoradium`` that is already imported (pass ``force=True`` to replace it) and ``uninstall()`` removes exactly
what it added.  Submodules of the reference that are out of scope here (``trjchan``, ``deepmimo``) are
not aliased: importing them raises ``ModuleNotFoundError`` naming this build.
"""
import importlib
import importlib.abc
import importlib.machinery
import runpy
import sys

# submodules of the reference package (neoradium/*.py) this build has a counterpart for
SUBMODULES = ('antenna', 'carrier', 'cdl', 'chancodebase', 'channelmodel', 'csifeedback', 'csirs', 'dmrs', 'grid', 'harq',
              'ldpc', 'modulation', 'pdsch', 'polar', 'random', 'snrhelper', 'tdl', 'utils', 'waveform')
OUT_OF_SCOPE = ('trjchan', 'deepmimo')
_installed = []


class _OutOfScopeFinder(importlib.abc.MetaPathFinder):
    """`import neoradium.trjchan` must fail with a message that names this build, not fall through to some other package."""

    def find_spec(self, name, path=None, target=None):
        if name.startswith('neoradium.') and name.split('.', 1)[1] in OUT_OF_SCOPE and 'neoradium' in sys.modules \
                and getattr(sys.modules['neoradium'], '__name__', '') == 'neoradium_amd':
            raise ModuleNotFoundError(f"{name} is outside the scope of neoradium_amd (the MI355X PDSCH hot path); "
                                      f"'neoradium' is aliased to neoradium_amd by neoradium_amd.compat.install()")
        return None


_finder = _OutOfScopeFinder()


def install(force=False):
    """Alias ``neoradium`` (and its submodules) to ``neoradium_amd``.  Returns the package module."""
    import neoradium_amd as pkg
    cur = sys.modules.get('neoradium')
    if cur is not None and cur is not pkg:
        if not force:
            raise ImportError("a different 'neoradium' package is already imported; call install(force=True) to replace it")
        for k in [k for k in sys.modules if k == 'neoradium' or k.startswith('neoradium.')]:
            del sys.modules[k]
    if cur is pkg:
        return pkg
    sys.modules['neoradium'] = pkg
    _installed.append('neoradium')
    for sub in SUBMODULES:
        mod = importlib.import_module('neoradium_amd.' + sub)
        sys.modules['neoradium.' + sub] = mod
        _installed.append('neoradium.' + sub)
    if _finder not in sys.meta_path:
        sys.meta_path.insert(0, _finder)
    return pkg


def uninstall():
    """Remove the aliases install() added."""
    while _installed:
        sys.modules.pop(_installed.pop(), None)
    if _finder in sys.meta_path:
        sys.meta_path.remove(_finder)


def installed():
    import neoradium_amd as pkg
    return sys.modules.get('neoradium') is pkg


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        print(__doc__)
        return 2
    install()
    sys.argv = argv
    runpy.run_path(argv[0], run_name='__main__')
    return 0


if __name__ == '__main__':
    sys.exit(main())
