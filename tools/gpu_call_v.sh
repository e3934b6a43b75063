#!/bin/bash
# PMC passes over configs[3] (k_pipe, shipped build): where do the CU's cycles go?
mkdir -p gpurun_out/r02v
CMD="python $PWD/bench.py --workload gammatone --no-cpu-baseline --no-parity-check --steps 3 --warmup 1"
i=0
for set in "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 bash tools/pmc_run.sh gt$i $set -- $CMD
  mv gpurun_out/pmc_gt$i gpurun_out/pmc_gt$i.log gpurun_out/r02v/ 2>/dev/null
done
python - <<'PY' > gpurun_out/r02v/pmc_gammatone.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r02v/pmc_gt*")):
  for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
      acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
      if "k_pipe" not in k: continue
      for c, v in sorted(cs.items()):
        print("%-28s n=%d mean=%.5g" % (c, len(v), sum(v) / len(v)))
PY
cat gpurun_out/r02v/pmc_gammatone.txt
rm -rf gpurun_out/r02v/pmc_gt?   # raw CSVs are large
tail -3 gpurun_out/r02v/pmc_gt4.log
