#!/usr/bin/env python3
"""Summarise rocprofv3 counter CSVs: per kernel name, mean of each counter over dispatches."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
  for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
      acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
      if "k_wave" not in k and "k_small" not in k and "alz" not in k and "k_fir" not in k:
        continue
      print(d, k)
      for c, v in sorted(cs.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
