"""neoradium_amd -- MI355X-native PDSCH link-level hot path behind the NeoRadium class surface.

    from neoradium_amd import Carrier, PDSCH, CdlChannel, AntennaPanel, LdpcEncoder, Grid, random, SnrScheduler

mirrors ``from neoradium import ...`` (reference neoradium/__init__.py:4-21) for the classes on the PDSCH / HARQ
link path; ``neoradium_amd.engine.PdschLink`` is the batched, device-resident Monte-Carlo loop.

The compute path is libnrx.so (hand-written HIP for gfx950, C ABI in include/nrx.h).  There is no CPU fallback:
calling any operator without the built library or without a GPU raises.
"""
__version__ = '0.1.0'

from .channelmodel import ChannelModel                         # noqa: F401
from .modulation import Modem                                  # noqa: F401
from .harq import HarqEntity                                   # noqa: F401
from .ldpc import LdpcEncoder, LdpcDecoder                     # noqa: F401
from .carrier import Carrier, BandwidthPart                    # noqa: F401
from .cdl import CdlChannel                                    # noqa: F401
from .tdl import TdlChannel                                    # noqa: F401
from .grid import Grid                                         # noqa: F401
from .waveform import Waveform                                 # noqa: F401
from .antenna import AntennaElement, AntennaPanel, AntennaArray  # noqa: F401
from .pdsch import PDSCH, DMRS, PTRS                           # noqa: F401
from .csirs import CsiRsConfig, CsiRsSet, CsiRs                # noqa: F401
from .csifeedback import CsiReport                             # noqa: F401
from .random import random                                     # noqa: F401
from .snrhelper import SnrScheduler                            # noqa: F401
from .engine import PdschLink, run_sweep, run_harq_sharded     # noqa: F401

try:                                                           # polar codec (control channel path)
    from .polar import PolarEncoder, PolarDecoder              # noqa: F401
    from .pdcch import PDCCH                                   # noqa: F401
except ImportError:                                            # pragma: no cover
    pass
