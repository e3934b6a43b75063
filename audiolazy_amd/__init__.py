"""audiolazy_amd -- an MI355X-native blocked stream-filter engine behind
AudioLazy's ZFilter / Stream operator surface.

Only the hot path of the reference is here: linear filter execution
(LinearFilter/ZFilter.__call__ incl. Stream-valued coefficients, CascadeFilter, ParallelFilter,
resonator/comb, lowpass/highpass, the gammatone bank), lpc.kautocor, and the pieces either side
of it (WavStream / chunks sample formats, Streamix mixing), executed by hand-written HIP kernels
for gfx950 in libalzhip.so (C ABI: include/alz.h).  Filter design, the z**-1 algebra and the lazy
Stream type stay on the host in float64 and follow the reference's semantics.
"""
from ._ffi import ParCorError, load as load_library, device_count, last_kernel  # noqa: F401
from .stream import (Stream, ControlStream, Streamix, StreamTeeHub, MemoryLeakWarning, blocks, thub, tostream,  # noqa: F401
                     avoid_stream, cycle, repeat, count, chain, zero_pad, rint)
from .bank import FilterBank, memory_to_hist, sections_of, block_size, mix_tracks, mix_sets  # noqa: F401
from .poly import Poly, x, lagrange, resample  # noqa: F401
from .strategy import StrategyDict  # noqa: F401
from .filters import (LinearFilter, LinearFilterProperties, ZFilter, z, FilterList, CascadeFilter, ParallelFilter,  # noqa: F401
                      comb, resonator, lowpass, highpass)
from .auditory import erb, gammatone_erb_constants, gammatone, gammatone_bank, erb_space  # noqa: F401
from .lpc import (acorr, levinson_durbin, lpc, kautocor_frames, acorr_frames, lag_matrix, lag_matrix_frames,  # noqa: F401
                  toeplitz, parcor, parcor_stable, lsf, lsf_stable)
from .synth import white_noise, zeros, zeroes, ones, karplus_strong  # noqa: F401
from .analysis import envelope, envelope_block, maverage, amdf, clip  # noqa: F401
from . import maps  # noqa: F401
from .pcm import WavStream, chunks, decode_pcm, encode_pcm  # noqa: F401
from .misc import (dB10, dB20, freq2lag, lag2freq, freq_to_lag, lag_to_freq, almost_eq, line, cached, elementwise,  # noqa: F401
                   DEFAULT_SAMPLE_RATE)


def sHz(rate):
  """(s, Hz) unit pair for a sample rate: ``440 * Hz`` is 440 Hz in rad/sample and
  ``0.5 * s`` is half a second in samples (reference audiolazy/lazy_misc.py:300-320)."""
  import math
  return float(rate), 2 * math.pi / rate

__version__ = "0.2.0"
