#!/bin/bash
# combs on a few channels of time-major rows: the test, then throughput against round 1's kernel on stereo rows
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_gpu_bank.py -x -q -k "few_channels" 2>&1 | tail -30
for a in "--channels 2 --log2-samples 22 --comb-delay 441" "--channels 2 --log2-samples 22 --comb-delay 441 --comb-linearized"; do
  python bench.py --workload comb $a --no-cpu-baseline --no-secondary --full-json - > /tmp/nc.json 2> /tmp/nc.err || tail -5 /tmp/nc.err
  python tools/show_line.py /tmp/nc.json | head -1 | cut -c1-200
done
ALZ_COMB_OFF=1 ALZ_LIBRARY=tools/variants/libalzhip_tuning.so python bench.py --workload comb --channels 2 --log2-samples 18 --comb-delay 441 --no-cpu-baseline --no-secondary --no-parity-check --full-json - > /tmp/nc.json 2> /tmp/nc.err || tail -5 /tmp/nc.err
python tools/show_line.py /tmp/nc.json | head -1 | cut -c1-200
