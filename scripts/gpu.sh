#!/bin/bash
# Everything that is run on the GPU box, one mode per lease (through gpurun: `gpurun --timeout N -- 'bash scripts/gpu.sh <mode> ...'`).
# Output lands in gpurun_out/<tag>_*; what is to be judged is copied into profiles/ by hand.
#
#   tests [tag] [pytest targets...]   the -m gpu suite (default: all of tests/), slowest durations
#   ab VAR [tag] [pytest targets...]  a parity subset, then both workloads' step time with VAR=1 / VAR=0 interleaved
#                                     (VAR: any switch api.hip reads at context creation -- FG_WINO, FG_WINO_UP, FG_WINO_5X5, FG_WINO_WGRAD,
#                                      FG_FUSE_PRELU, FG_THIN_SLAB, FG_THIN_BIAS, FG_DEFER_WFINISH, FG_ADAM_PACK, ...)
#   conv [tag]                        per-layer micro-benchmark (scripts/bench_conv.py) of both workloads' layer shapes
#   kstats [tag]                      rocprofv3 --kernel-trace --stats tables of both workloads
#   trace-wino [tag] [dbg]            s_memtime rows of wino_kernel (FG_WINO_TRACE=1; dbg = FG_WINO_DBG, 100 = every 8 MFMA slots)
#   trace-wgrad [tag]                 s_memtime rows of wino_wgrad_kernel (FG_WINO_WGRAD_TRACE=1 and 2)
#   ubench                            scripts/ubench/issue: cost of one non-MFMA instruction next to v_mfma_f32_32x32x2_f32
#   evidence [tag]                    scripts/collect_profiles.sh: the default bench line + kernel tables + PMC passes
#   pmc-tail [tag]                    FETCH / WRITE / SQ wait-active / L2 hit-miss / instruction-mix passes over a few cfg2 iterations for the
#                                     tail kernels (pack_jobs, wgrad_finish_jobs, actpool_*, adam, bn_apply) -> gpurun_out/<tag>.txt
#   store-width                       scripts/ubench/store_width: what a store's width / shape costs (64 MB, 5 shapes x distances)
set -u
MODE=${1:-tests}; shift 1 || true
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
T0=$(date +%s)
step_ms() {   # step_ms <workload> <extra bench args...>: prints "ms_per_step value"
  local w=$1; shift
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-alt-math --no-roofline "$@" 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value'],1))"
}
case $MODE in
tests)
  TAG=${1:-tests}; shift 1 || true
  timeout 1700 python -m pytest ${@:-tests} -m gpu -q --durations=8 > $OUT/${TAG}.log 2>&1
  echo "rc=$? t=$(( $(date +%s) - T0 ))"; grep -n "^FAILED\|^ERROR" $OUT/${TAG}.log | head -30; tail -14 $OUT/${TAG}.log ;;
ab)
  VAR=${1:?VAR}; TAG=${2:-ab}; shift 2 || true
  timeout 900 python -m pytest ${@:-tests/test_gpu_wino.py tests/test_gpu_ops.py tests/test_gpu_net.py} -m gpu -q > $OUT/${TAG}_tests.log 2>&1
  echo "tests rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
  grep -n "^FAILED\|^ERROR\|passed\|failed" $OUT/${TAG}_tests.log | tail -12
  for i in 1 2; do for v in 1 0; do echo "cfg2 $VAR=$v $(env $VAR=$v bash -c "$(declare -f step_ms); step_ms cfg2")" | tee -a $OUT/${TAG}_summary.txt; done; done
  for v in 1 0; do echo "c2f $VAR=$v $(env $VAR=$v bash -c "$(declare -f step_ms); step_ms c2f --steps 10 --warmup 3")" | tee -a $OUT/${TAG}_summary.txt; done ;;
conv)
  TAG=${1:-conv}
  timeout 300 python scripts/bench_conv.py 20 > $OUT/${TAG}_cfg2.txt 2>&1; timeout 400 python scripts/bench_conv.py 10 c2f > $OUT/${TAG}_c2f.txt 2>&1
  cat $OUT/${TAG}_cfg2.txt $OUT/${TAG}_c2f.txt ;;
kstats)
  TAG=${1:-ks}
  rocprofv3 --kernel-trace --stats -d $OUT/p1 -o run -- python bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-roofline > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
  python scripts/rocpd_stats.py $OUT/p1/run_results.db auto --by-grid > $OUT/${TAG}_bench_kernel_stats.md
  rocprofv3 --kernel-trace --stats -d $OUT/p2 -o run -- python bench.py --workload c2f --steps 6 --warmup 2 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
  python scripts/rocpd_stats.py $OUT/p2/run_results.db auto > $OUT/${TAG}_c2f_kernel_stats.md
  rm -rf $OUT/p1 $OUT/p2
  head -45 $OUT/${TAG}_bench_kernel_stats.md; head -40 $OUT/${TAG}_c2f_kernel_stats.md ;;
trace-wino)
  TAG=${1:-wtr}; DBG=${2:-0}
  python -m face_generator_amd.build --measure > /dev/null 2>&1; export FACEGEN_HIP_LIB=$PWD/face_generator_amd/libfacegen_hip_measure.so   # the trace kernels exist only there
  rm -f $OUT/${TAG}_trace.txt
  for shape in "128 16 16 64 128 3 0" "128 32 32 128 256 3 0" "128 16 16 256 128 5 1" "128 64 64 128 256 5 0"; do
    FG_WINO_DBG=$DBG FG_WINO_TRACE=1 FG_WS_TRACE_FILE=$OUT/${TAG}_trace.txt timeout 120 python scripts/bench_one.py fwd 2 0 $shape > /dev/null 2>&1
  done
  if [ "$DBG" -ge 100 ]; then python scripts/wino_trace2_report.py $OUT/${TAG}_trace.txt | tee $OUT/${TAG}_trace_report.txt | head -70
  else python scripts/ws_trace_report.py $OUT/${TAG}_trace.txt 2>&1 | grep "launch\|per block\|MFMA-pipe\|wall" | cut -c1-330 | tee $OUT/${TAG}_trace_report.txt; fi
  gzip -f $OUT/${TAG}_trace.txt ;;
trace-wgrad)
  TAG=${1:-wwtr}
  python -m face_generator_amd.build --measure > /dev/null 2>&1; export FACEGEN_HIP_LIB=$PWD/face_generator_amd/libfacegen_hip_measure.so
  rm -f $OUT/${TAG}_t1.txt $OUT/${TAG}_t2.txt
  for shape in "128 16 16 256 128 5 1" "128 8 8 128 256 5 1" "128 64 64 64 128 5 0" "128 32 32 128 256 3 0"; do
    FG_WINO_WGRAD_TRACE=1 FG_WS_TRACE_FILE=$OUT/${TAG}_t1.txt timeout 120 python scripts/bench_one.py wgrad 2 0 $shape > /dev/null 2>&1
    FG_WINO_WGRAD_TRACE=2 FG_WS_TRACE_FILE=$OUT/${TAG}_t2.txt timeout 120 python scripts/bench_one.py wgrad 2 0 $shape > /dev/null 2>&1
  done
  python scripts/ws_trace_report.py $OUT/${TAG}_t1.txt 2>&1 | grep "launch\|per block\|MFMA-pipe\|wall" | cut -c1-330 | tee $OUT/${TAG}_t1_report.txt
  python scripts/wino_trace2_report.py $OUT/${TAG}_t2.txt 2>&1 | tee $OUT/${TAG}_t2_report.txt | grep -v "chunk  [3-9]\|chunk 1[0-2]" | head -40
  gzip -f $OUT/${TAG}_t1.txt $OUT/${TAG}_t2.txt ;;
ubench)
  timeout 200 scripts/ubench/issue | tee $OUT/ubench_issue.txt ;;
evidence)
  bash scripts/collect_profiles.sh ${1:-r06} ;;
pmc-tail)
  TAG=${1:-pmc_tail}
  rm -f $OUT/${TAG}.txt
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
    d=$OUT/pmc_$(echo $pmc | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $d -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
    echo "## $pmc" >> $OUT/${TAG}.txt
    for k in pack_jobs wgrad_finish_jobs actpool_bwd actpool_fwd adam_kernel bn_apply; do python scripts/pmc_summary.py $d $k >> $OUT/${TAG}.txt 2>&1; done
    rm -rf $d
  done
  cat $OUT/${TAG}.txt ;;
store-width)
  [ -x scripts/ubench/store_width ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/store_width scripts/ubench/store_width.hip 2>/dev/null
  scripts/ubench/store_width | tee $OUT/store_width.txt ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
echo "done t=$(( $(date +%s) - T0 ))"
