#!/usr/bin/env python3
"""Generates and builds tools/variants/ubench_dep2: a lone wave per SIMD running loops of FP64 instructions with chosen dependency
patterns (round 6: what does a dependent v_mul_f64 / v_add_f64 really cost, by distance and by kind?).  usage: python tools/ubench_dep2.py"""
import os, subprocess
tests = []
def chains(k, ops):      # k interleaved chains; ops: per-position opcode pattern
  body = []
  for i in range(k):
    op = ops[i % len(ops)]
    body.append("v_%s_f64 %%%d, %%%d, %%8" % (op, i, i))
  return body
for k in (1, 2, 3, 4, 5, 6, 7, 8):
  tests.append(("%d interleaved v_mul_f64 chains (operands %d instructions old)" % (k, k), chains(k, ["mul"])))
tests.append(("1 chain alternating v_mul_f64 / v_add_f64", ["v_mul_f64 %0, %0, %8", "v_add_f64 %0, %0, %9"]))
tests.append(("8 chains, v_mul_f64 and v_add_f64 alternating", chains(8, ["mul", "add"])))
# biquad recurrence step, y in %0, previous y in %1, t1 %2, t2 (carried) %3, t2n %4; p = %9 (constant stand-in), A = %8, B = %10
tests.append(("recurrence step as compiled: t1=A y; t2n=B y; t1+=p; y=t2+t1; (t2<-t2n by renaming: two steps)",
              ["v_mul_f64 %2, %8, %0", "v_mul_f64 %4, %10, %0", "v_add_f64 %2, %2, %9", "v_add_f64 %0, %3, %2",
               "v_mul_f64 %2, %8, %0", "v_mul_f64 %3, %10, %0", "v_add_f64 %2, %2, %9", "v_add_f64 %0, %4, %2"]))
tests.append(("recurrence step: t1=A y; t1+=p; y=t2+t1; t2n=B yprev (two steps)",
              ["v_mul_f64 %2, %8, %0", "v_add_f64 %2, %2, %9", "v_mov_b64 %1, %0", "v_add_f64 %0, %3, %2", "v_mul_f64 %3, %10, %1"]))
tests.append(("recurrence step with v_fma: y = fma(A, y, fma(B, yprev, p)) as two dependent fmas",
              ["v_fma_f64 %2, %10, %1, %9", "v_mov_b64 %1, %0", "v_fma_f64 %0, %8, %0, %2"]))
src = ['#include <hip/hip_runtime.h>', '#include <stdio.h>', '#include <stdlib.h>',
       '#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e)); exit(1);} } while (0)']
for n, (name, body) in enumerate(tests):
  rept = max(1, 240 // len(body))
  asm = ".rept %d\\n\\t" % rept + " \\n\\t".join(body) + " \\n\\t.endr"
  src.append('__global__ __launch_bounds__(64) void k%d(double *out, long long *cyc, int iters, double a, double b, double c) {' % n)
  src.append('  double v[8]; for (int i = 0; i < 8; ++i) v[i] = 1.0 + threadIdx.x * 1e-3 + i;')
  src.append('  const long long t0 = __builtin_readcyclecounter();')
  src.append('  for (int i = 0; i < iters; ++i) asm volatile("%s" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(a), "v"(b), "v"(c));' % asm)
  src.append('  const long long t1 = __builtin_readcyclecounter();')
  src.append('  double s = 0; for (int i = 0; i < 8; ++i) s += v[i]; out[blockIdx.x * 64 + threadIdx.x] = s; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; }')
src.append('int main() { double *out; long long *cyc; const int waves = 1024, iters = 4000; CK(hipMalloc(&out, waves * 64 * 8)); CK(hipMalloc(&cyc, waves * 8)); long long h[8];')
for n, (name, body) in enumerate(tests):
  rept = max(1, 240 // len(body))
  src.append('  k%d<<<waves, 64>>>(out, cyc, 4, 1.0000001, 1e-9, 0.9999999); CK(hipDeviceSynchronize()); k%d<<<waves, 64>>>(out, cyc, iters, 1.0000001, 1e-9, 0.9999999); CK(hipDeviceSynchronize());' % (n, n))
  src.append('  CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost)); printf("%%-100s %%.2f cycles per instruction (%%.1f per pattern of %d)\\n", "%s", (double)h[3] / (iters * %d.0), (double)h[3] / (iters * %d.0));'
             % (len(body), name, rept * len(body), rept))
src.append('  return 0; }')
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, "tools", "variants"), exist_ok=True)
path = os.path.join(root, "tools", "variants", "ubench_dep2.hip")
open(path, "w").write("\n".join(src) + "\n")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", path, "-o", os.path.join(root, "tools", "variants", "ubench_dep2")], check=True)
print("built", len(tests), "tests")
