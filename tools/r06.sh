#!/bin/bash
# Round 6: ONE runner for every GPU-box call of the round (round 5 had 37 one-off r05_call*.sh).
#   gpurun --timeout T -- 'bash tools/r06.sh <call> [args]'
# Each call writes under gpurun_out/r06_<call>/; tools/README.md has the table "figure -> call".
R=${GRAFT_REPO_ROOT:-$(pwd)}
call=$1; shift
O=$R/gpurun_out/r06_$call
mkdir -p $O
cd $R
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
suite() { timeout ${1:-1500} python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" $O/pytest_gpu.log | tail -3; }
smoke() { timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1; }
driver() { timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json $O/bench_driver_full.json > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; python tools/show_line.py $O/bench_driver.json | cut -c1-170; }
stats() { cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --full-json $O/bench_under_rocprof.json "$@" > $O/stats.log 2>&1
  find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
  find $O/stats -name "*kernel_trace.csv" -exec sh -c 'grep -E "Kernel_Name|alz::" "$1" > '$O'/kernel_dispatches.csv' _ {} \;
  rm -rf $O/stats; head -12 $O/kernel_stats.csv | cut -c1-170; cd $R; }
case $call in
  look)    # the one-pass kernel: the race demonstrated on the ablation build, the soak, the scan tests, then the whole suite
    for dbg in 3072 1024; do
      ALZ_LIBRARY=$R/tools/variants/libalzhip_lookrace.so ALZ_WAVE_DEBUG=$dbg timeout 300 python tools/look_race_demo.py 20 2>&1 | tee -a $O/look_race_demo.log
    done
    ALZ_LIBRARY=$R/tools/variants/libalzhip_lookrace.so ALZ_WAVE_DEBUG=4096 timeout 300 python tools/look_check_demo.py 2>&1 | tee $O/look_check_demo.log
    timeout 600 python -m pytest tests/test_gpu_look_soak.py -x -q -s > $O/soak.log 2>&1; echo "soak rc=$?"; tail -25 $O/soak.log
    timeout 600 python -m pytest tests/test_gpu_scan.py -x -q > $O/scan.log 2>&1; echo "scan rc=$?"; tail -5 $O/scan.log
    suite; smoke; driver ;;
  comb)    # the comb step kernels: their tests, the karplus goldens, timings by shape, round 1's k_sparse beside them
    timeout 900 python -m pytest tests/test_gpu_bank.py tests/test_gpu_filters_api.py -x -q -k "comb or string or karplus" > $O/comb_tests.log 2>&1; echo "comb tests rc=$?"; tail -5 $O/comb_tests.log
    B="--no-cpu-baseline --no-secondary --steps 10 --warmup 2 --full-json -"
    for a in "--layout time" "--layout chan" "--layout time --in-place" "--layout chan --in-place" "--layout time --comb-linearized" "--layout chan --comb-linearized" \
             "--layout time --comb-delay 64" "--layout chan --comb-delay 64" "--channels 1 --log2-samples 22 --comb-delay 109 --comb-linearized --layout chan" \
             "--channels 1 --log2-samples 22 --comb-delay 441 --layout chan" "--channels 64 --log2-samples 20 --comb-delay 109 --comb-linearized --layout chan" "--channels 16384 --log2-samples 16 --layout time" "--channels 16384 --log2-samples 16 --layout chan"; do
      echo "== comb $a"; timeout 300 python bench.py --workload comb $a $B > $O/tmp.json 2> $O/tmp.err || tail -3 $O/tmp.err; python tools/show_line.py $O/tmp.json | head -1 | cut -c1-230
    done 2>&1 | tee $O/comb_shapes.log
    echo "== one string through k_comb_cm (tuning build, ALZ_STRING_OFF=1)" | tee -a $O/comb_shapes.log
    for a in "--channels 1 --log2-samples 22 --comb-delay 109 --comb-linearized --layout chan" "--channels 1 --log2-samples 22 --comb-delay 441 --layout chan"; do
      echo "== k_comb_cm $a"; ALZ_STRING_OFF=1 ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so timeout 300 python bench.py --workload comb $a $B --no-parity-check > $O/tmp.json 2> $O/tmp.err || tail -3 $O/tmp.err; python tools/show_line.py $O/tmp.json | head -1 | cut -c1-230
    done 2>&1 | tee -a $O/comb_shapes.log
    echo "== round 1's k_sparse on the same shapes (tuning build, ALZ_COMB_OFF=1)" | tee -a $O/comb_shapes.log
    for a in "--layout time" "--channels 1 --log2-samples 20 --comb-delay 109 --comb-linearized --layout time"; do
      echo "== k_sparse $a"; ALZ_COMB_OFF=1 ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so timeout 300 python bench.py --workload comb $a $B --no-parity-check > $O/tmp.json 2> $O/tmp.err || tail -3 $O/tmp.err; python tools/show_line.py $O/tmp.json | head -1 | cut -c1-230
    done 2>&1 | tee -a $O/comb_shapes.log ;;
  pipe)    # configs[3] k_pipe: SQ counters of the shipped kernel (one pass per set), per-wave cycle accounting and the shader clock
           # of THIS box under the kernel's load (-DALZ_ABLATE -DALZ_PIPE_TIMING variant), the same for the headline kernel's box rate
    G="--workload gammatone --no-cpu-baseline --no-secondary --no-parity-check --steps 6 --warmup 2 --full-json -"
    timeout 300 python bench.py --workload gammatone --no-cpu-baseline --no-secondary --steps 10 --warmup 2 --full-json - > $O/g.json 2> $O/g.err; python tools/show_line.py $O/g.json | head -1 | cut -c1-200 | tee $O/pipe_rate.log
    ALZ_LIBRARY=$R/tools/variants/libalzhip_pipetiming.so timeout 300 python bench.py $G 2>&1 | grep "k_pipe block" | sort | uniq -c | sort -rn | head -12 | tee $O/pipe_timing.log
    for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
               "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
               "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVES"; do
      tag=$(echo $set | cut -d' ' -f1)
      (cd /tmp; timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python $R/bench.py $G > $O/pmc_$tag.log 2>&1)
      python tools/pmc_summary.py $O/pmc_$tag 2>&1 | grep -A12 "k_pipe" | head -14 | tee -a $O/pipe_pmc.txt; rm -rf $O/pmc_$tag
    done ;;
  mid)     # k_mid: its tests, the two workloads, the lane-per-channel kernels on the same shapes (tuning build, ALZ_MID_OFF=1)
    timeout 900 python -m pytest tests/test_gpu_mid.py -x -q > $O/mid_tests.log 2>&1; echo "mid tests rc=$?"; tail -5 $O/mid_tests.log
    B="--no-cpu-baseline --no-secondary --steps 10 --warmup 2 --full-json -"
    for a in "--workload butter6" "--workload maverage256" "--workload butter6 --channels 16384 --log2-samples 16" "--workload maverage256 --channels 16384 --log2-samples 16"; do
      echo "== $a"; timeout 300 python bench.py $a $B > $O/tmp.json 2> $O/tmp.err || tail -3 $O/tmp.err; python tools/show_line.py $O/tmp.json | head -1 | cut -c1-230
      echo "== round 5's kernels: $a"; ALZ_MID_OFF=1 ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so timeout 600 python bench.py $a $B --no-parity-check --steps 3 --warmup 1 > $O/tmp.json 2> $O/tmp.err || tail -3 $O/tmp.err; python tools/show_line.py $O/tmp.json | head -1 | cut -c1-230
    done 2>&1 | tee $O/mid_shapes.log ;;
  pipelong) # configs[3]: the same kernel timed as 12 launches on a cold chip and as 90, alternating (NOTES_r06.md 3: the "box spread")
    for k in 1 2 3; do
      for a in "--steps 10 --warmup 2" "--steps 60 --warmup 30"; do
        timeout 300 python bench.py --workload gammatone --no-cpu-baseline --no-secondary --no-parity-check $a --full-json - > $O/g.json 2> $O/g.err
        echo "$a: $(python tools/show_line.py $O/g.json | head -1 | cut -c1-90)"
      done
    done 2>&1 | tee $O/pipe_long.log ;;
  asan)    bash tools/asan_check.sh gpu 2>&1 | tee $O/asan_gpu.log ;;
  firvar)  # configs[2]: register tile x occupancy of k_fir_ring (variant builds of alz_fir.hip over the tuning objects)
    for v in ${FIRVARS:-tuning fir_i32 tuning fir_i32}; do
      for m in "" "--fused"; do
        ALZ_LIBRARY=$R/tools/variants/libalzhip_$v.so timeout 300 python bench.py --workload fir $m --no-cpu-baseline --steps 6 --warmup 2 --full-json - > $O/f.json 2> $O/f.err || tail -2 $O/f.err
        echo "$v $m: $(python tools/show_line.py $O/f.json | head -1 | cut -c1-150)"
      done
    done 2>&1 | tee $O/fir_variants.log ;;
  pmc)     # HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the workloads named by the regex $1 (default: round 6's new ones)
    bash tools/pmc_workloads.sh r06_pmc "${1:-comb|karplus|iir_order6|maverage|narrow512_time_parallel}" 2>&1 | tee $O/pmc_workloads.log
    python tools/pmc_table.py $R/gpurun_out/r06_pmc profiles/r05_pmc_traffic_table.json > $O/r06_pmc_traffic_table.json; python - <<PYEOF
import json
t = json.load(open("$O/r06_pmc_traffic_table.json"))["workloads"]
for k, v in sorted(t.items()):
  print("%-42s traffic / algorithmic %.4f" % (k, v["traffic_over_algorithmic"]))
PYEOF
    ;;
  fuzz)    # the GPU-side differential fuzzers against the oracle (round 6's new one first)
    for f in "fuzz_sparse.py 300 606" "fuzz_timeparallel.py 100 1604" "fuzz_bank.py 200 1605" "fuzz_outer.py 100 1606" "fuzz_stream.py 120 1607"; do
      echo "== $f"; timeout 900 python tools/$f 2>&1 | tail -6 | cut -c1-400
    done 2>&1 | tee $O/fuzz_gpu.log ;;
  lpcab)   # LPC modes through variant libraries, interleaved (LIBS="shipped lev_old ..."; MODES overrides the mode list), then the LPC tests
    IFS=';' read -ra modes <<< "${MODES:---lpc-exact;--lpc-frames 1048576 --lpc-exact;;--lpc-frames 1048576}"
    for rep in 1 2; do
      for lib in ${LIBS:-shipped lev_old}; do
        for m in "${modes[@]}"; do
          if [ $lib = shipped ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so; fi
          timeout 300 python bench.py --workload lpc $m --no-cpu-baseline --no-secondary --steps 200 --warmup 50 --full-json - > $O/l.json 2> $O/l.err || tail -3 $O/l.err
          echo "$lib [$m]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-120)"
        done
      done
    done 2>&1 | tee $O/lpc_ab.log
    unset ALZ_LIBRARY
    timeout 900 python -m pytest tests/test_gpu_lpc.py tests/test_gpu_fullwidth.py -x -q -k "lpc or LPC or frames or levinson" 2>&1 | tail -3 | tee -a $O/lpc_ab.log ;;
  duofma)  # configs[1] in the opt-in FMA mode: k_duo's fused instantiations with the storing wave / non-temporal tiles / paced pass (variant
           # builds of alz_wave.hip: -DALZ_DUO_FMA3=1 [-DALZ_DUO_FMA_ORDER=1] [-DALZ_PACE_ALL=1]) against the shipped rule (default kernel)
    for rep in 1 2; do
      for lib in ${LIBS:-shipped duo_fma3 duo_fma3o duo_fma3op}; do
        for m in "--fused" "--fused --layout chan"; do
          if [ $lib = shipped ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so; fi
          timeout 300 python bench.py --workload biquad $m --no-cpu-baseline --no-secondary --steps 10 --warmup 3 --full-json - > $O/l.json 2> $O/l.err || tail -3 $O/l.err
          echo "$lib [$m]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-170)"
        done
      done
    done 2>&1 | tee $O/duo_fma.log ;;
  pipeph)  # configs[3]: where in a group each stage wave of k_pipe issues its hand-over's LDS operations (variant builds of alz_casc.hip,
           # -DALZ_PIPE_PHMAP=0x....; libalzhip_pipe_ph<map>.so), warm-chip protocol, interleaved
    for rep in 1 2; do
      for lib in ${LIBS:-shipped pipe_ph0222 pipe_ph0202 pipe_ph2222 pipe_ph0212 pipe_ph0321 pipe_ph0220}; do
        for m in "" ${MODES:-"--fused"}; do
          if [ $lib = shipped ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so; fi
          timeout 300 python bench.py --workload gammatone $m --no-cpu-baseline --no-secondary --steps 60 --warmup 30 --full-json - > $O/l.json 2> $O/l.err || tail -3 $O/l.err
          echo "$lib [$m]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-140)"
        done
      done
    done 2>&1 | tee $O/pipe_phase.log ;;
  cascskew) # k_casc: the sections of a cascade in one skewed sweep (shipped) against section after section (-DALZ_CASC_SKEW=0): the
           # one-stream filterbank in time-parallel mode (both layouts) and a bank wide enough for the single-wave cascade (256 streams)
    timeout 900 python -m pytest tests -x -q -m gpu -k "casc or cscan or outer or cfg4 or gammatone or cascade" > $O/casc_tests.log 2>&1; echo "cascade tests rc=$?"; tail -3 $O/casc_tests.log
    for rep in 1 2; do
      for lib in ${LIBS:-shipped casc_noskew}; do
        for m in "--streams 1 --log2-samples 20 --time-parallel 1 --bank-layout chan" "--streams 1 --log2-samples 20 --time-parallel 1 --bank-layout time" "--streams 256 --bank-layout chan" "--streams 256 --bank-layout time"; do
          if [ $lib = shipped ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so; fi
          timeout 300 python bench.py --workload gammatone $m --no-cpu-baseline --no-secondary --steps 30 --warmup 10 --full-json - > $O/l.json 2> $O/l.err || tail -3 $O/l.err
          echo "$lib [$m]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-170)"
        done
      done
    done 2>&1 | tee $O/casc_skew.log ;;
  suite)   suite; smoke ;;
  final)   suite; smoke; driver; stats ;;
  py)      timeout ${T:-900} python "$@" 2>&1 | tee $O/py_$(basename $1 .py).log | tail -${TAIL:-40} ;;
  sh)      timeout ${T:-900} bash "$@" 2>&1 | tee $O/sh_$(basename $1 .sh).log | tail -${TAIL:-40} ;;
  *) echo "unknown call $call" ;;
esac
