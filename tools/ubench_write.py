"""Write-only HBM ceiling on this box: torch fill_/zero_ on an 8 GiB float64 tensor, a read-only sum,
and a copy, for the roofline of kernels whose traffic is (almost) only stores (configs[3]: 8 B written
per output, input served from L2)."""
import torch
n = 1 << 30
y = torch.empty(n, dtype=torch.float64, device="cuda")
x = torch.empty(n, dtype=torch.float64, device="cuda")


def timed(fn, reps=10):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / reps


ms = timed(lambda: y.fill_(1.5))
print("fill_   8 GiB: %.3f ms  %.2f TB/s written" % (ms, n * 8 / ms / 1e9))
ms = timed(lambda: y.zero_())
print("zero_   8 GiB: %.3f ms  %.2f TB/s written" % (ms, n * 8 / ms / 1e9))
ms = timed(lambda: y.copy_(x))
print("copy_   8 GiB: %.3f ms  %.2f TB/s read + written" % (ms, 2 * n * 8 / ms / 1e9))
ms = timed(lambda: x.sum())
print("sum     8 GiB: %.3f ms  %.2f TB/s read" % (ms, n * 8 / ms / 1e9))
