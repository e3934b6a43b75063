#!/bin/bash
# A/B builds of libalzhip.so (tools/variants/*.so, same ABI) on the cfg2 bench, both layouts.
for f in tools/variants/*.so; do
  for lay in time chan; do
    v=$(ALZ_LIBRARY=$PWD/$f timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --layout $lay 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f %s' % (d['value'], d['config']['parity_spot_check']))")
    echo "$(basename $f) $lay $v"
  done
done
