#!/usr/bin/env python3
"""k_fir_ring requests a block's taps with a hand-written s_load_dwordx8 and waits for them a block later.  The compiler does not
know that the destination registers are in flight in between: this script reads a device assembly listing of alz_fir.hip
(hipcc -S --cuda-device-only) and reports any instruction that touches the destination SGPRs of such a load before the next
`s_waitcnt lgkmcnt(0)` along the fall-through path (conditional branches are scanned straight through, an unconditional one ends
the scan).   usage: tools/check_asm_prefetch.py fir.s"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
bad = n = open_ends = 0
for i, l in enumerate(lines):
  m = re.match(r"\s+s_load_dwordx8 s\[(\d+):(\d+)\], s\[\d+:\d+\], 0x0", l)
  if not (m and ";;#ASMSTART" in lines[i - 1]):
    continue
  lo, hi = int(m.group(1)), int(m.group(2))
  n += 1
  for j in range(i + 1, min(len(lines), i + 4000)):
    t = lines[j]
    if "s_waitcnt lgkmcnt(0)" in t:
      break
    if re.match(r"\s+(s_branch|s_endpgm|s_setpc_b64)\b", t):   # the listing's next line is another path: not followed
      open_ends += 1
      break
    if t.strip().startswith(";") or "s_load_dwordx8" in t:
      continue
    hit = any(not (int(b) < lo or int(a) > hi) for a, b in re.findall(r"s\[(\d+):(\d+)\]", t)) or \
          any(lo <= int(r) <= hi for r in re.findall(r"\bs(\d+)\b", t))
    if hit:
      bad += 1
      print("touched before the wait: line %d: %s" % (j + 1, t.strip()))
      break
print("hand-written tap loads: %d, violations: %d (scans that ended at an unconditional branch: %d)" % (n, bad, open_ends))
sys.exit(1 if bad or n == 0 else 0)
