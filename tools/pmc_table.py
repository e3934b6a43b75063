"""Merge the per-workload counter sums of tools/pmc_workloads.sh into profiles/r0N_pmc_traffic_table.json, the table
bench.py fills every ``roofline.traffic`` from (labelled "not measured in this run").
usage: python tools/pmc_table.py gpurun_out/<tag> [previous table] > profiles/r0N_pmc_traffic_table.json
(a previous table supplies the workloads this pass did not measure again)"""
import glob, json, os, sys
d = sys.argv[1]
ALG = {  # algorithmic bytes per step, as bench.py defines them
  "headline": 16.0 * 4096 * 2 ** 20, "fir256_bit_exact": 16.0 * 8192 * 2 ** 18, "fir256_fma": 16.0 * 8192 * 2 ** 18,
  "gammatone": (8 + 8 / 256.) * 256 * 64 * 2 ** 16, "gammatone_one_stream": (8 + 8 / 256.) * 256 * 2 ** 20,
  "gammatone_one_stream_time_parallel": (8 + 8 / 256.) * 256 * 2 ** 20,
  "gammatone_one_stream_time_parallel_tm": (8 + 8 / 256.) * 256 * 2 ** 20, "narrow512_time_parallel_chan": 16.0 * 512 * 2 ** 20, "lpc": 3984.0 * 65536, "lpc_bit_identical": 3984.0 * 65536,
  "lpc_fma": 3984.0 * 65536, "lpc_1m": 3984.0 * 2 ** 20, "lpc_1m_bit_identical": 3984.0 * 2 ** 20,
  "gammatone_fma": (8 + 8 / 256.) * 256 * 64 * 2 ** 16, "envelope_abs": 16.0 * 4096 * 2 ** 20, "timevar_shared": 16.0 * 4096 * 2 ** 18,
  "timevar_per_channel": 40.0 * 4096 * 2 ** 18, "narrow512_bit_exact": 16.0 * 512 * 2 ** 20, "narrow512_time_parallel": 16.0 * 512 * 2 ** 20,
  "narrow512_time_parallel_three_launch": 16.0 * 512 * 2 ** 20,
  "comb_fb": 16.0 * 4096 * 2 ** 18, "comb_fb_chan": 16.0 * 4096 * 2 ** 18, "karplus_one_string": 16.0 * 2 ** 22,
  "iir_order6": 16.0 * 4096 * 2 ** 18, "maverage_recursive_256": 16.0 * 4096 * 2 ** 18,
  "biquad_chan": 16.0 * 4096 * 2 ** 20, "biquad_chan_fma": 16.0 * 4096 * 2 ** 20, "biquad_fma": 16.0 * 4096 * 2 ** 20, "biquad_8192": 16.0 * 8192 * 2 ** 19}
# FETCH_SIZE x 2 per the guide's gfx950 note -- for every kernel: round 5 read a known byte count in k_fir_ring's shape
# (8 B / lane buffer loads, 512-byte row pieces 64 KiB apart: tools/ubench_fetch8.hip, profiles/r05_fetch8_calibration.log)
# and the counter reported half of it, as it does for 16 B / lane streams (TCC_EA0_RDREQ x 64 B, the requests are 128 B).
# The tables of rounds 3 - 4 carried the FIR rows raw ("uncalibrated"): their 1.95 x was 3.4 x.
WIDE = lambda key: True
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes + kernel trace only, summed over this library's kernels of "
               "`python bench.py --no-secondary --no-parity-check --steps 3 --warmup 1 <workload>` and divided by the 4 steps; "
               "FETCH_SIZE doubled (guide: gfx950 counts half of a stream; calibrated here on k_duo: 1.00001 x algorithmic, and in round 5 "
               "on a known byte count read in k_fir_ring's 8 B/lane shape: tools/ubench_fetch8.hip, profiles/r05_fetch8_calibration.log "
               "-- the FIR rows of the round 3 - 4 tables were raw, i.e. half); WRITE_SIZE as reported", "workloads": {}}
if len(sys.argv) > 2:
  out["workloads"] = {k: v for k, v in json.load(open(sys.argv[2]))["workloads"].items() if k in ALG}
for key, alg in ALG.items():
  try:
    f = json.load(open(os.path.join(d, "pmc_%s_FETCH_SIZE.json" % key)))
    w = json.load(open(os.path.join(d, "pmc_%s_WRITE_SIZE.json" % key)))
  except (OSError, ValueError):
    continue
  fetch = f["kb_per_step"] * 1024 * (2.0 if WIDE(key) else 1.0)
  write = w["kb_per_step"] * 1024
  out["workloads"][key] = {"fetch_bytes_per_step": fetch, "write_bytes_per_step": write, "traffic_bytes_per_step": fetch + write,
                           "algorithmic_bytes_per_step": alg, "traffic_over_algorithmic": (fetch + write) / alg,
                           "fetch_correction": 2.0 if WIDE(key) else 1.0,
                           "kernels": {k: {"launches_per_4_steps": v[0], "FETCH_KB": v[1], "WRITE_KB": w["kernels"].get(k, [0, 0])[1]}
                                       for k, v in f["kernels"].items()}}
print(json.dumps(out, indent=1))
