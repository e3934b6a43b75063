"""Hand-written scalar prefetches: k_fir_ring (csrc/alz_fir.hip, round 6) requests the taps of block k + 1 with an s_load_dwordx8
while block k runs and waits for them a block later; k_cdot3 (csrc/alz_scan.hip, round 5) does the same with pairs of
s_load_dwordx16.  The compiler does not know that the destination SGPRs are in flight in between.  This test compiles the files to
device listings and checks (tools/check_asm_prefetch.py) that nothing touches those registers before the wait, for the shipped
optimisation level and for the -O1 of the sanitizer builds."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc on this machine")
@pytest.mark.parametrize("src,opt", [("alz_fir.hip", "-O3"), ("alz_fir.hip", "-O1"), ("alz_scan.hip", "-O3"), ("alz_scan.hip", "-O1")])
def test_no_instruction_touches_the_tap_registers_in_flight(tmp_path, src, opt):
  out = tmp_path / "listing.s"
  cmd = [HIPCC, "--offload-arch=gfx950", opt, "-std=c++17", "-ffp-contract=off", "-fPIC", "-fno-fast-math", "-S",
         "--cuda-device-only", os.path.join(ROOT, "audiolazy_amd", "csrc", src), "-o", str(out)]
  subprocess.run(cmd, check=True, capture_output=True, timeout=600)
  r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_prefetch.py"), str(out)], capture_output=True, text=True)
  assert r.returncode == 0, r.stdout + r.stderr
  assert "violations: 0" in r.stdout
