#!/bin/bash
# The library's HOST code under AddressSanitizer + UBSan (make -C audiolazy_amd/csrc asan -> tools/variants/libalzhip_asan.so):
#   tools/asan_check.sh cpu    the CPU suite's library tests (exports, signatures, argument checks, error mapping) -- build container
#   tools/asan_check.sh gpu    smoke(), the one-pass / scan tests, the comb and k_mid tests, one bench line -- on a GPU box, through the
#                              UBSan-only build (make ubsan): with ASan's allocator interposed the ROCm runtime cannot reserve its
#                              address space (hipInit aborts with "AddressSanitizer: out-of-memory" inside libamdhip64: call 7)
# Any report of the sanitizers fails the run (halt_on_error).  protect_shadow_gap=0: the ROCm runtime maps memory where ASan's
# shadow gap lies; detect_leaks=0: the interpreter and the runtime keep their allocations until exit.
R=$(cd $(dirname $0)/.. && pwd)
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export ALZ_LIBRARY=$R/tools/variants/libalzhip_asan.so LD_PRELOAD=$RT
cd $R
case ${1:-cpu} in
  cpu) python -m pytest tests/test_cabi_cpu.py tests/test_bench_cpu.py -x -q -m "not gpu" 2>&1 | tail -5 ;;
  gpu) unset LD_PRELOAD; export ALZ_LIBRARY=$R/tools/variants/libalzhip_ubsan.so
       python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok under host UBSan')" 2>&1 | tail -3
       python -m pytest tests/test_gpu_scan.py tests/test_gpu_mid.py tests/test_gpu_bank.py tests/test_gpu_host_path.py -x -q -m gpu 2>&1 | tail -5
       python -m pytest tests/test_gpu_fullwidth.py -x -q -m gpu -k "clock" 2>&1 | tail -3     # (round 6, third session: the launchers of the clocked shapes)
       python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --channels 512 --log2-samples 14 --full-json - 2>&1 | tail -2 | cut -c1-300 ;;
esac
