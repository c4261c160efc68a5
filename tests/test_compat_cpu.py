"""`neoradium_amd.compat.install()`: the Playground notebooks' import statements resolve to this build unchanged
(reference neoradium/__init__.py:4-21; the statements below are the distinct import lines of the reference's notebooks,
written out here -- nothing is read from /root/reference)."""
import subprocess
import sys
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NOTEBOOK_IMPORTS = [
    "import neoradium as nr",
    "from neoradium import Carrier, PDSCH, Grid",
    "from neoradium import Carrier, PDSCH, CdlChannel, AntennaPanel, Grid, random",
    "from neoradium import LdpcEncoder, HarqEntity, random, Modem",
    "from neoradium import Carrier, PDSCH, CdlChannel, AntennaPanel, Grid, random, LdpcEncoder",
    "from neoradium import LdpcEncoder, LdpcDecoder",
    "from neoradium import Carrier, PDSCH, LdpcEncoder, HarqEntity, random",
    "from neoradium import Carrier, PDSCH, Grid, CsiRsConfig, CsiRsSet",
    "from neoradium import Carrier, PDSCH, CdlChannel, AntennaPanel, LdpcEncoder, random, HarqEntity, SnrScheduler",
    "from neoradium import Carrier, PDSCH, CdlChannel, AntennaPanel, LdpcEncoder, LdpcDecoder, Grid",
    "from neoradium import Carrier, PDSCH, CdlChannel, AntennaPanel, LdpcEncoder, Grid, random, Waveform",
    "from neoradium import Carrier, PDSCH, CdlChannel, AntennaPanel, LdpcEncoder, Grid, random, SnrScheduler",
    "from neoradium import Carrier, PDSCH, CdlChannel, AntennaPanel, Grid, random, LdpcEncoder, HarqEntity, SnrScheduler",
    "from neoradium import Carrier, Modem, TdlChannel, Grid, random",
    "from neoradium import Carrier, Modem, CdlChannel, AntennaPanel, Grid, Waveform, random",
    "from neoradium import Carrier, CsiRsConfig, CsiRsSet, CdlChannel, Grid, AntennaPanel",
    "from neoradium import Carrier, CdlChannel, AntennaPanel, random, Waveform",
    "from neoradium import PolarEncoder, PolarDecoder",
    "from neoradium import CsiReport, ChannelModel, AntennaElement, AntennaArray, DMRS, PTRS",
    "from neoradium.utils import getNmse",
    "from neoradium.utils import toLinear",
    "from neoradium.utils import toDb, toLinear, getNmse",
    "from neoradium.utils import getNmse, getMse",
    "from neoradium.harq import HarqProcess, HarqCW",
]


@pytest.fixture
def alias():
    import neoradium_amd.compat as compat
    assert 'neoradium' not in sys.modules or compat.installed()
    compat.install()
    yield compat
    compat.uninstall()
    assert 'neoradium' not in sys.modules and 'neoradium.utils' not in sys.modules


def test_notebook_import_lines_resolve_to_this_build(alias):
    import neoradium_amd as pkg
    for stmt in NOTEBOOK_IMPORTS:
        ns = {}
        exec(stmt, ns)
        for name, obj in ns.items():
            if name == '__builtins__':
                continue
            if name == 'nr':
                assert obj is pkg
                continue
            mod = stmt.split()[1]
            home = pkg if mod == 'neoradium' else sys.modules[mod.replace('neoradium', 'neoradium_amd', 1)]
            assert getattr(home, name) is obj, (stmt, name)
            owner = getattr(obj, '__module__', None) or type(obj).__module__
            assert owner.startswith('neoradium_amd'), (stmt, name, owner)
    # one shared global generator, not a copy: seeding through either name is seen through the other
    from neoradium import random as r1
    from neoradium_amd import random as r2
    assert r1 is r2
    r1.setSeed(5)
    a = r2.bits(8)
    r2.setSeed(5)
    assert (r1.bits(8) == a).all()
    import neoradium.utils
    import neoradium_amd.utils
    assert neoradium.utils is neoradium_amd.utils and neoradium.__version__ == pkg.__version__


def test_out_of_scope_submodules_fail_by_name(alias):
    for stmt in ("from neoradium import DeepMimoData", "from neoradium import TrjChannel, Trajectory"):
        with pytest.raises(ImportError):
            exec(stmt, {})
    with pytest.raises(ModuleNotFoundError, match="outside the scope of neoradium_amd"):
        exec("import neoradium.trjchan", {})
    with pytest.raises(ModuleNotFoundError, match="outside the scope of neoradium_amd"):
        exec("from neoradium.deepmimo import DeepMimoData", {})


def test_install_does_not_shadow_a_real_package(alias):
    import types
    alias.uninstall()
    sys.modules['neoradium'] = types.ModuleType('neoradium')
    try:
        with pytest.raises(ImportError):
            alias.install()
        pkg = alias.install(force=True)
        assert sys.modules['neoradium'] is pkg
    finally:
        alias.uninstall()
        sys.modules.pop('neoradium', None)


def test_run_a_script_unchanged_through_the_module_entry(tmp_path):
    script = tmp_path / "nb.py"
    script.write_text("from neoradium import Carrier, PDSCH, CdlChannel, AntennaPanel, LdpcEncoder, Grid, random, SnrScheduler\n"
                      "from neoradium.utils import toLinear, getNmse, getMse\n"
                      "import sys\n"
                      "car = Carrier(numRbs=25, spacing=15)\n"
                      "print(Carrier.__module__, toLinear(10.0), car.curBwp.numRbs, sys.argv[1])\n")
    r = subprocess.run([sys.executable, '-m', 'neoradium_amd.compat', str(script), 'x7'], cwd=ROOT, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ['neoradium_amd.carrier', '10.0', '25', 'x7']
