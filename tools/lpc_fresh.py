"""Where do 57 us per step go on a fresh box?  (the driver's `lpc_bit_identical` 0.134 ms per step, two
rounds running, against 0.081 behind a test suite and 0.077 under rocprofv3.)  Run as the FIRST GPU process of a lease:
  python tools/lpc_fresh.py
For the bit-identical kernel (k_acorr_stage<17, dense Levinson>), the default one and the FMA one, in this order of
experiments: per-launch HIP-event durations of the first 48 launches; the old bench protocol (5 warm-ups + 20 steps, one
event pair); the same after 2 s of host sleep; 400 steps; 20 steps again; and host-side enqueue time per call."""
import ctypes
import subprocess
import sys
import time

sys.path.insert(0, '.')
t_imp = time.perf_counter()
import torch
from audiolazy_amd import _ffi

L = _ffi.load()
print("import + load: %.1f s" % (time.perf_counter() - t_imp))
N, order, F = 480, 16, 65536


def smi(tag):
  try:
    out = subprocess.run(["rocm-smi", "--showclocks", "--showperflevel"], capture_output=True, text=True, timeout=20).stdout
    keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Performance Level"))]
    print("[smi %s] %s" % (tag, " | ".join(keep)[:400]))
  except Exception as exc:
    print("[smi %s] unavailable: %s" % (tag, exc))


smi("start")
g = torch.Generator(device='cuda').manual_seed(20260927)
sig = torch.empty(F * N, dtype=torch.float64, device='cuda').uniform_(-1, 1, generator=g)
coefs = torch.empty((F, order + 1), dtype=torch.float64, device='cuda')
err = torch.empty(F, dtype=torch.float64, device='cuda')
st = torch.empty(F, dtype=torch.int32, device='cuda')
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def make(flags):
  def call():
    _ffi.check(L.alz_lpc_kautocor_dev_ex(sig.data_ptr(), F, N, N, order, coefs.data_ptr(), err.data_ptr(), st.data_ptr(),
                                         flags, 0, stream))
  return call


def protocol(call, warm, steps):
  for _ in range(warm):
    call()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  e0.record()
  for _ in range(steps):
    call()
  t_enq = time.perf_counter() - t0
  e1.record()
  torch.cuda.synchronize()
  wall = time.perf_counter() - t0
  return e0.elapsed_time(e1) / steps * 1e3, wall / steps * 1e6, t_enq / steps * 1e6


for name, flags in (("bit-identical (dense Levinson)", _ffi.LPC_DENSE), ("default", 0), ("fma", _ffi.LPC_FUSED)):
  call = make(flags)
  torch.cuda.synchronize()
  evs = []
  for _ in range(48):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); call(); b.record()
    evs.append((a, b))
  torch.cuda.synchronize()
  d = [a.elapsed_time(b) * 1e3 for a, b in evs]
  print("%s: first 48 launches, us each (own event pair): %s" % (name, " ".join("%.0f" % v for v in d)))
  print("   old protocol 5 + 20:        events %.1f us/step, wall %.1f, host enqueue %.1f" % protocol(call, 5, 20))
  time.sleep(2.0)
  print("   after 2 s of idle, 5 + 20:  events %.1f us/step, wall %.1f, host enqueue %.1f" % protocol(call, 5, 20))
  time.sleep(2.0)
  print("   after 2 s of idle, 0 + 20:  events %.1f us/step, wall %.1f, host enqueue %.1f" % protocol(call, 0, 20))
  print("   400 steps:                  events %.1f us/step, wall %.1f, host enqueue %.1f" % protocol(call, 5, 400))
  print("   5 + 20 right after:         events %.1f us/step, wall %.1f, host enqueue %.1f" % protocol(call, 5, 20))
  smi("after " + name)
# the bench's own order: a CPU-side pause the length of its parity leg, then the workload
call = make(_ffi.LPC_DENSE)
import numpy as np
t0 = time.perf_counter()
x = np.random.default_rng(1).uniform(-1, 1, 1 << 24)
while time.perf_counter() - t0 < 3.0:
  x = np.sort(x)[::-1].copy()
print("after 3 s of host CPU work: 3 + 20 (round 4's secondary protocol): events %.1f us/step, wall %.1f, host enqueue %.1f"
      % protocol(call, 3, 20))
