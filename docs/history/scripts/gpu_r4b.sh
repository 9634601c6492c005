#!/bin/bash
# round 4, second call: parity of the changed pieces, then A/B pairs inside one box (each setting twice, interleaved)
set -u
OUT=gpurun_out; TAG=${1:-r04b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_net.py tests/test_gpu_c2f.py tests/test_gpu_ops.py tests/test_gpu_step_abi.py tests/test_golden.py -m gpu -q -x > $OUT/${TAG}_tests.log 2>&1
echo "parity rc=$?" | tee $OUT/${TAG}_summary.txt; tail -4 $OUT/${TAG}_tests.log
run() {  # label, workload, env...
  local label=$1 wl=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $wl --steps $([ $wl = c2f ] && echo 10 || echo 50) --warmup $([ $wl = c2f ] && echo 3 || echo 10) --no-cpu-baseline --no-alt-math --no-clock-probe --no-live-traffic > $OUT/${TAG}_b.json 2>/dev/null
  python - "$label" "$wl" <<'P'
import json,sys
d=json.loads(open("gpurun_out/%s_b.json" % __import__("os").environ.get("TAG","r04b")).read().strip().splitlines()[-1])
t=d["roofline"].get("hbm_tail",[])
pick=lambda n: " ".join("%s=%.1fus" % (x["kernel"], x["us"]) for x in t if n in x["kernel"])
print("[%s] %s %.1f img/s %.4f ms | tail %.3f ms | %s | %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["roofline"].get("hbm_tail_total",{}).get("ms_per_iter",0), pick("thin_in<3"), pick("thin_wgrad<3")))
P
}
export TAG
for rep in 1 2; do
  run "default" cfg2 FG_X=0
  run "thin_bias=0" cfg2 FG_THIN_BIAS=0
  run "tpw=2" cfg2 FG_THIN_TPW=2
  run "tpw=1" cfg2 FG_THIN_TPW=1
  run "nt=1" cfg2 FG_THIN_NT=1
  run "tpw=2 nt=1" cfg2 FG_THIN_TPW=2 FG_THIN_NT=1
done 2>&1 | tee $OUT/${TAG}_ab_cfg2.txt
for rep in 1 2; do
  run "default" c2f FG_X=0
  run "thin_bias=0" c2f FG_THIN_BIAS=0
done 2>&1 | tee $OUT/${TAG}_ab_c2f.txt
# per-kernel view of the small kernels after the multi_final change
rocprofv3 --kernel-trace --stats -d $OUT/p1 -o run -- python bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p1/run_results.db auto > $OUT/${TAG}_bench_kernel_stats.md
rocprofv3 --kernel-trace --stats -d $OUT/p2 -o run -- python bench.py --workload c2f --steps 6 --warmup 2 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p2/run_results.db auto > $OUT/${TAG}_c2f_kernel_stats.md
rm -rf $OUT/p1 $OUT/p2
grep -E "multi_final|wgrad_finish|colsum|thin_in_mfma_kernel|thin_wgrad|iterations" $OUT/${TAG}_bench_kernel_stats.md $OUT/${TAG}_c2f_kernel_stats.md
