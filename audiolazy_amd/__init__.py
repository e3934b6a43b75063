"""audiolazy_amd -- an MI355X-native blocked stream-filter engine behind
AudioLazy's ZFilter / Stream operator surface.

Only the hot path of the reference is here: linear filter
execution (LinearFilter/ZFilter.__call__, CascadeFilter, resonator/comb,
lowpass/highpass, the gammatone bank) and lpc.kautocor, executed by hand-written
HIP kernels for gfx950 in libalzhip.so (C ABI: include/alz.h).  Filter design
and the z**-1 algebra stay on the host in float64.
"""
from ._ffi import ParCorError, load as load_library, device_count  # noqa: F401
from .stream import Stream, blocks  # noqa: F401
from .bank import FilterBank, memory_to_hist, sections_of  # noqa: F401

__version__ = "0.1.0"
