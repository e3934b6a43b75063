#!/usr/bin/env python3
"""HBM traffic per launch of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE,
WRITE_SIZE; kernel trace only), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes
for gfx950 (FETCH_SIZE counts half of a 16 B/lane stream: doubled).

usage: tools/pmc_traffic.py <dir of FETCH pass> <dir of WRITE pass> <kernel substring> <channels> <samples> > json
"""
import collections, csv, glob, json, sys


def mean_counter(d, counter, kernel):
  vals = []
  for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
      if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
        per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals.extend(per_dispatch.values())
  return sum(vals) / len(vals), len(vals)


fetch_dir, write_dir, kernel, C, N = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
fetch_kb, nf = mean_counter(fetch_dir, "FETCH_SIZE", kernel)
write_kb, nw = mean_counter(write_dir, "WRITE_SIZE", kernel)
alg = 8.0 * C * N
fetch_b, write_b = 2.0 * fetch_kb * 1024, write_kb * 1024
print(json.dumps({
  "kernel": kernel, "block": "%d channels x %d samples, time-major" % (C, N), "dispatches": [nf, nw],
  "algorithmic_read_bytes": alg, "algorithmic_write_bytes": alg,
  "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
  "fetch_bytes_corrected": fetch_b, "write_bytes": write_b, "traffic_bytes": fetch_b + write_b,
  "traffic_over_algorithmic": (fetch_b + write_b) / (2 * alg),
  "note": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) + --kernel-trace only; FETCH_SIZE doubled per "
          "the guide's gfx950 correction (it reads 1/2 of a 16 B/lane stream), WRITE_SIZE uncalibrated"}, indent=1))
