#!/bin/bash
# quick A/B of the wave-specialised weight gradient: parity subset, per-layer timing, one SQ pass
set -u
OUT=gpurun_out
TAG=${1:-r3e}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
FG_WGRAD_WS=1 timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_ops.py -m gpu -q -x -k "cfg2 or S64_forward_backward or conv" > $OUT/${TAG}_ws_tests.log 2>&1
echo "ws parity rc=$?" | tee $OUT/${TAG}_summary.txt
for ws in 0 1; do
  FG_WGRAD_WS=$ws timeout 200 python scripts/bench_conv.py 10 > $OUT/${TAG}_conv_ws$ws.txt 2>&1
done
d=$OUT/pmc_tmp; rm -rf $d
FG_WGRAD_WS=1 timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $d -- python scripts/bench_one.py wgrad 4 > /dev/null 2>&1
python scripts/pmc_summary.py $d wgrad_ws > $OUT/${TAG}_pmc.txt 2>&1; rm -rf $d
FG_WGRAD_WS=1 timeout 200 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > $OUT/${TAG}_bench_c2f_ws1.json 2>/dev/null
FG_WGRAD_WS=1 timeout 200 python bench.py --workload cfg2 --no-cpu-baseline --no-alt-math --no-live-traffic > $OUT/${TAG}_bench_cfg2_ws1.json 2>/dev/null
tail -3 $OUT/${TAG}_ws_tests.log
grep -h "wgrad" $OUT/${TAG}_conv_ws0.txt $OUT/${TAG}_conv_ws1.txt
cat $OUT/${TAG}_pmc.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), round(d["ms_per_step"],4))
        for k,v in d["kernels"].items():
            if "wgrad" in k: print("    %-50s %6.3f ms %6.1f TF"%(k,v["ms_per_iter"],v["executed_tflops"]))
    except Exception as e: print(f, "ERR", e)
PY
