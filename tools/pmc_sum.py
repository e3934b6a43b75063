"""Sum one rocprofv3 counter over this library's kernels in a counter_collection CSV tree.
usage: pmc_sum.py <dir> <counter> <launch groups = steps + warmup>  ->  JSON {counter_kb_per_step, kernels: {name: [launches, kb]}}"""
import collections, csv, glob, json, re, sys
d, counter, groups = sys.argv[1], sys.argv[2], int(sys.argv[3])
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != counter or "alz::" not in r["Kernel_Name"]:
      continue
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("(anonymous namespace)::", "")
    per[name][r["Dispatch_Id"]] += float(r["Counter_Value"])
kernels = {k: [len(v), sum(v.values())] for k, v in per.items()}
total = sum(v[1] for v in kernels.values())
print(json.dumps({"counter": counter, "kb_per_step": total / groups, "steps_counted": groups, "kernels": kernels}))
