#!/usr/bin/env python3
"""k_fir_ring (alz_fir.hip) and k_cdot3 (alz_scan.hip) request scalar operands with hand-written s_load_dwordx8 / x16 and wait
for them a block of arithmetic later.  The compiler does not know that the destination registers are in flight in between: this
script reads a device assembly listing (hipcc -S --cuda-device-only) and reports any instruction that touches the destination
SGPRs of such a request before the next `s_waitcnt lgkmcnt(0)` along the fall-through path (conditional branches are scanned
straight through, an unconditional one ends the scan).   usage: tools/check_asm_prefetch.py file.s"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
bad = n = open_ends = 0
i = 0
while i < len(lines):
  if ";;#ASMSTART" not in lines[i]:
    i += 1
    continue
  j = i + 1
  dst = []
  while j < len(lines) and ";;#ASMEND" not in lines[j]:
    m = re.match(r"\s+s_load_dwordx(?:8|16) s\[(\d+):(\d+)\], s\[\d+:\d+\], 0x0", lines[j])
    if m:
      dst.append((int(m.group(1)), int(m.group(2))))
    j += 1
  i = j + 1
  if not dst:
    continue
  n += 1
  for k in range(j + 1, min(len(lines), j + 6000)):
    t = lines[k]
    if "s_waitcnt lgkmcnt(0)" in t:
      break
    if re.match(r"\s+(s_branch|s_endpgm|s_setpc_b64)\b", t):   # the listing's next line is another path: not followed
      open_ends += 1
      break
    if t.strip().startswith(";") or not t.startswith("\t"):
      continue
    hit = False
    for lo, hi in dst:
      if any(not (int(b) < lo or int(a) > hi) for a, b in re.findall(r"s\[(\d+):(\d+)\]", t)) or \
         any(lo <= int(r) <= hi for r in re.findall(r"\bs(\d+)\b", t)):
        hit = True
    if hit:
      bad += 1
      print("touched before the wait: line %d: %s" % (k + 1, t.strip()))
      break
print("hand-written scalar requests: %d, violations: %d (scans that ended at an unconditional branch: %d)" % (n, bad, open_ends))
sys.exit(1 if bad or n == 0 else 0)
