"""A markdown table of a bench.py full record (profiles/r0N_bench_full.json): headline + every secondary workload with
value, median ms per step, fastest / slowest batch, roofline fraction, traffic ratio, kernel, parity.
usage: python tools/bench_table.py profiles/r05_bench_full.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("| workload | value | ms per step (median; min – max of the batches) | roofline frac | traffic / algorithmic | kernel | parity |")
print("|---|---|---|---|---|---|---|")
print("| **configs[1]** headline (K = %d timed steps) | **%.1f %s** | %.4f | %.4f (kernel %.4f ms) | %s | `%s` | %s |" % (
    d["steps"], d["value"], d["unit"], d["ms_per_step"], r["frac"], r["kernel_ms_avg"],
    "%.5f" % r["traffic_ratio"] if r.get("traffic_ratio") else "–", d["config"].get("kernel"), str(d["config"].get("parity_spot_check"))[:70]))
for k, e in d.get("secondary", {}).items():
  if "roofline" not in e:
    continue
  t = e.get("timing") or {}
  rr = e["roofline"]
  print("| `%s` | %.4g %s | %.4f (%s) | %.3f | %s | `%s` | %s |" % (
      k, e["value"], e["unit"], e["ms_per_step"],
      "%.4f – %.4f, %d × %d" % (t["ms_per_step_min"], t["ms_per_step_max"], t["batches"], t["steps_per_batch"]) if t else "–",
      rr["frac"], "%.4f" % rr["traffic_ratio"] if rr.get("traffic_ratio") else "–", str(e["kernel"])[:44], str(e["parity"])[:60]))
c = d.get("cpu_baseline")
if c:
  print("\nCPU (%s, %d cores used of %s logical): %s" % (c["kind"], c["cores"], c.get("host_logical_cpus"),
        ", ".join("%s %.4g %s" % (k, v["value"] * 1e3, "Msamples/s") for k, v in c.get("legs", {}).items())))
