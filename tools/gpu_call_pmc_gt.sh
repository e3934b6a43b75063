#!/bin/bash
# HBM bytes of configs[3]'s k_pipe per launch (separate FETCH_SIZE / WRITE_SIZE passes, counters + kernel trace only)
mkdir -p gpurun_out/r02pmc
CMD="python $PWD/bench.py --workload gammatone --no-cpu-baseline --no-parity-check --steps 3 --warmup 1"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 200 bash tools/pmc_run.sh gt_$ctr $ctr -- $CMD
done
python - <<'PY' | tee gpurun_out/r02pmc/pmc_gammatone_traffic.txt
import csv, glob, collections
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
  for f in glob.glob("gpurun_out/pmc_gt_%s/**/*counter_collection.csv" % ctr, recursive=True):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_pipe" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    if vals: res[ctr] = (len(vals), sum(vals) / len(vals))
alg_w = 256 * 64 * 65536 * 8.0
alg_r = 64 * 65536 * 8.0
print("k_pipe, 256 bands x 64 streams x 2^16 samples, channel-major; per launch (mean over dispatches):")
for ctr, (n, kb) in res.items():
  corr = 2.0 if ctr == "FETCH_SIZE" else 1.0     # the guide's gfx950 correction for FETCH_SIZE on 16 B/lane streams (WRITE_SIZE uncalibrated)
  print("  %s n=%d mean=%.5g KB  -> %.4f GB (x%.0f correction: %.4f GB)" % (ctr, n, kb, kb * 1024 / 1e9, corr, kb * 1024 * corr / 1e9))
print("  algorithmic: %.4f GB written (8 B per output), %.4f GB read once (the 64 input streams; every band re-reads them from L2)" % (alg_w / 1e9, alg_r / 1e9))
PY
rm -rf gpurun_out/pmc_gt_FETCH_SIZE gpurun_out/pmc_gt_WRITE_SIZE
