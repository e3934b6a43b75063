#!/usr/bin/env python3
"""Wall-clock rate of the filter *call protocol* (Python iterables in, Stream out) at several
block sizes: LTI biquad and a time-varying resonator steered by coefficient Streams."""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiolazy_amd as al

s, Hz = al.sHz(44100)
N = 1 << 17
data = list(al.white_noise(N))
for block in (64, 1024, 4096, 65536):
  al.block_size(block)
  filt = al.resonator.z_exp(1000 * Hz, 100 * Hz)
  t0 = time.perf_counter(); out = list(filt(data)); dt = time.perf_counter() - t0
  freqs = al.Stream(itertools.cycle([(900 + k) * Hz for k in range(200)]))
  tv = al.resonator.z_exp(freqs, 100 * Hz)
  t1 = time.perf_counter(); out = tv(data).take(N); dt2 = time.perf_counter() - t1
  print("block %6d: LTI %.2f Msamples/s   time-varying (design per sample on the host) %.3f Msamples/s"
        % (block, N / dt / 1e6, N / dt2 / 1e6))
