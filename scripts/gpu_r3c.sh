#!/bin/bash
# round 3, third GPU call: whole suite (default switches), then the wave-specialised fp32 weight gradient: parity + A/B timing
set -u
OUT=gpurun_out
TAG=${1:-r3c}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x --durations=6 > $OUT/${TAG}_tests.log 2>&1
echo "suite rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
FG_WGRAD_WS=1 timeout 900 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q -k "cfg2 or S64_forward_backward or S64_full_steps or conv or G_forward" > $OUT/${TAG}_ws_tests.log 2>&1
echo "ws parity rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
for ws in 0 1; do
  FG_WGRAD_WS=$ws timeout 200 python scripts/bench_conv.py 10 > $OUT/${TAG}_conv_ws$ws.txt 2>&1
  FG_WGRAD_WS=$ws timeout 200 python bench.py --workload cfg2 --no-cpu-baseline --no-alt-math --no-live-traffic > $OUT/${TAG}_bench_cfg2_ws$ws.json 2>/dev/null
  FG_WGRAD_WS=$ws timeout 200 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > $OUT/${TAG}_bench_c2f_ws$ws.json 2>/dev/null
done
echo "all done t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
tail -4 $OUT/${TAG}_tests.log; tail -4 $OUT/${TAG}_ws_tests.log
grep -h "wgrad" $OUT/${TAG}_conv_ws0.txt $OUT/${TAG}_conv_ws1.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), round(d["ms_per_step"],4))
        for k,v in d["kernels"].items():
            if "wgrad" in k: print("    %-50s %6.3f ms %6.1f TF"%(k,v["ms_per_iter"],v["executed_tflops"]))
    except Exception as e: print(f, "ERR", e)
PY
