"""Print the headline and the secondary entries of a bench.py JSON line (file argument) compactly."""
import json, sys
seen = set()
for ln in open(sys.argv[1]):
  if ln.startswith("k_") and ln not in seen:        # device-side printf of an -DALZ_ABLATE build (first of each)
    seen.add(ln)
    if len(seen) <= 6:
      print("  " + ln.rstrip())
try:
  d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
except Exception as exc:
  print("no JSON line in", sys.argv[1], exc); sys.exit(0)
r = d["roofline"]
print("  %.2f %s  n_gpus %d  ms/step %.4f  kernel_ms %.4f  frac %.4f | %s | %s" % (
    d["value"], d["unit"], d["n_gpus"], d["ms_per_step"], r.get("kernel_ms_avg", 0), r["frac"], d["config"].get("kernel"),
    str(d["config"].get("parity_spot_check"))[:90]))
for k, v in d.get("secondary", {}).items():
  if "frac" in v:                                   # the compact line (round 4): slim secondary entries
    print("    %-38s %9.3f %-10s frac %.3f  ms/step %.4f %s %s traffic x%s | %s" % (k, v["value"], v["unit"], v["frac"],
          v["ms_per_step"], v.get("ms_min_max", ""), v.get("n", ""), v.get("traffic_ratio"), str(v["parity"])[:90]))
  elif "roofline" in v:
    print("    %-36s %9.3f %-10s frac %.3f (step %.3f) %-34s | %s" % (k, v["value"], v["unit"], v["roofline"]["frac"],
          v["roofline"].get("frac_from_ms_per_step", 0), str(v["kernel"])[:34], str(v["parity"])[:80]))
  else:
    print("    %-36s %s" % (k, json.dumps(v)[:200]))
