"""Throughput of the streaming kernels for the common tap patterns (4096 channels x 2**18, time-major)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import audiolazy_amd as al
C, N = (int(sys.argv[2]) if len(sys.argv) > 2 else 4096), 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 18)
LAYOUT = sys.argv[3] if len(sys.argv) > 3 else "time"
ONLY = sys.argv[4] if len(sys.argv) > 4 else ""
s, Hz = al.sHz(48000)
fc = np.geomspace(50., 20000., C)
designs = {
  "lowpass.pole (b0/a1)": lambda f: al.lowpass.pole(f * Hz),
  "highpass.z (b0 b1/a1)": lambda f: al.highpass.z(f * Hz),
  "lowpass.pole**2 (b0/a1 a2)": lambda f: al.lowpass.pole(f * Hz) ** 2,
  "resonator.z_exp (b0 b2/a1 a2)": lambda f: al.resonator.z_exp(f * Hz, f / 10 * Hz),
  "general biquad (b0 b1 b2/a1 a2)": lambda f: al.ZFilter([.2, .3, .1], [1., -1.2 * np.cos(f * Hz), .5]),
  "gain a0 != 1": lambda f: al.ZFilter([.2, .3, .1], [2., -1.2 * np.cos(f * Hz), .5]),
}
x = torch.rand((N, C) if LAYOUT == "time" else (C, N), dtype=torch.float64, device="cuda") * 2 - 1
print("%d channels x 2^%d, %s-major" % (C, N.bit_length() - 1, LAYOUT))
y = torch.empty_like(x)
for name, d in designs.items():
  if ONLY and ONLY not in name: continue
  bank = al.FilterBank.from_filters([d(f) for f in fc])
  bank.reset()
  bank.process(x, layout=LAYOUT, out=y); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(3): bank.process(x, layout=LAYOUT, out=y)
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 3
  print("%-34s %-12s %7.1f Gsamples/s" % (name, bank.last_kernel, C * N / ms / 1e6))
  if name.startswith("lowpass.pole (b0"):
    bank.set_input_map("abs")
    bank.reset()
    bank.process(x, layout=LAYOUT, out=y); torch.cuda.synchronize()
    e0.record()
    for _ in range(3): bank.process(x, layout=LAYOUT, out=y)
    e1.record(); torch.cuda.synchronize()
    print("%-34s %-12s %7.1f Gsamples/s" % (name + " after |x|", bank.last_kernel, C * N / (e0.elapsed_time(e1) / 3) / 1e6))
