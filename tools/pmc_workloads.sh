#!/bin/bash
# HBM traffic of every workload bench.py reports, by rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in separate passes,
# kernel trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes): one run per workload and counter, the
# counter summed over this library's kernels and divided by the number of steps.  Output: gpurun_out/<tag>/pmc_*.json,
# merged by tools/pmc_table.py into profiles/r0N_pmc_traffic_table.json.
#   usage: tools/pmc_workloads.sh <tag> [regex: only the workloads whose key matches]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-secondary --no-parity-check --steps 3 --warmup 1"
ONLY=${2:-.}
run() {  # key, bench args
  key=$1; shift
  [[ $key =~ $ONLY ]] || return 0
  for ctr in FETCH_SIZE WRITE_SIZE; do
    cd /tmp
    timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/raw_${key}_$ctr -o p -- python $R/bench.py $COMMON "$@" > $O/raw_${key}_$ctr.log 2>&1
    python $R/tools/pmc_sum.py $O/raw_${key}_$ctr $ctr 4 > $O/pmc_${key}_$ctr.json 2>> $O/errors.log
    rm -rf $O/raw_${key}_$ctr
  done
  echo "$key: $(cat $O/pmc_${key}_FETCH_SIZE.json | head -c 200)"
}
run headline
run fir256_bit_exact --workload fir
run fir256_fma --workload fir --fused
run gammatone --workload gammatone
run gammatone_fma --workload gammatone --fused
run gammatone_one_stream --workload gammatone --streams 1
run gammatone_one_stream_time_parallel --workload gammatone --streams 1 --time-parallel 1
run gammatone_one_stream_time_parallel_tm --workload gammatone --streams 1 --time-parallel 1 --bank-layout time
run lpc --workload lpc
run lpc_bit_identical --workload lpc --lpc-exact
run lpc_fma --workload lpc --fused
run lpc_1m --workload lpc --lpc-frames 1048576
run lpc_1m_bit_identical --workload lpc --lpc-frames 1048576 --lpc-exact
run envelope_abs --workload envelope
run timevar_shared --workload timevar
run timevar_per_channel --workload timevar --streams 0
run narrow512_bit_exact --channels 512 --time-parallel 0
run narrow512_time_parallel --channels 512 --time-parallel 1
run narrow512_time_parallel_chan --channels 512 --time-parallel 1 --layout chan
run narrow512_time_parallel_three_launch --channels 512 --time-parallel 8192
run comb_fb --workload comb
run comb_fb_chan --workload comb --layout chan
run karplus_one_string --workload comb --channels 1 --log2-samples 22 --comb-delay 109 --comb-linearized --layout chan
run iir_order6 --workload butter6
run maverage_recursive_256 --workload maverage256
run biquad_chan --layout chan
run biquad_chan_fma --layout chan --fused
run biquad_fma --fused
run biquad_8192 --channels 8192 --log2-samples 19
