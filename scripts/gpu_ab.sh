#!/bin/bash
# ws wgrad default (bias partials in the loaders) + 512 x 64 igemm_ws tiles: parity subset, then both bench lines with / without
set -u
OUT=gpurun_out
TAG=${1:-r3g}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_c2f.py tests/test_gpu_baseline_sizes.py tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not batch_64_two" > $OUT/${TAG}_tests.log 2>&1
echo "parity rc=$?" | tee $OUT/${TAG}_summary.txt
timeout 200 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > $OUT/${TAG}_bench_c2f_new.json 2>/dev/null
FG_WGRAD_WS=0 timeout 200 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > $OUT/${TAG}_bench_c2f_nowgradws.json 2>/dev/null
timeout 200 python bench.py --workload cfg2 --no-cpu-baseline --no-alt-math --no-live-traffic > $OUT/${TAG}_bench_cfg2_new.json 2>/dev/null
tail -3 $OUT/${TAG}_tests.log
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), round(d["ms_per_step"],4))
        for k,v in d["kernels"].items():
            print("    %-50s %4.1f %6.3f ms %6.1f TF"%(k,v["calls_per_iter"],v["ms_per_iter"],v["executed_tflops"]))
    except Exception as e: print(f, "ERR", e)
PY
