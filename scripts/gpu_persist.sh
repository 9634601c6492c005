#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3p}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
# correctness first, each under its own timeout (a barrier mismatch would hang)
FG_WS_PERSIST=3 timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_c2f.py -m gpu -q -x > $OUT/${TAG}_tests1.log 2>&1; echo "small parity rc=$?"
FG_WS_PERSIST=3 timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not batch_64_two" > $OUT/${TAG}_tests2.log 2>&1; echo "full-size parity rc=$?"
tail -3 $OUT/${TAG}_tests1.log; tail -3 $OUT/${TAG}_tests2.log
for pv in 0 1 3; do
  FG_WS_PERSIST=$pv timeout 200 python scripts/bench_conv.py 5 c2f 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_conv_c2f_p$pv.txt
  FG_WS_PERSIST=$pv timeout 200 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > $OUT/${TAG}_bench_c2f_p$pv.json 2>/dev/null
  FG_WS_PERSIST=$pv timeout 200 python bench.py --workload cfg2 --no-cpu-baseline --no-alt-math --no-live-traffic > $OUT/${TAG}_bench_cfg2_p$pv.json 2>/dev/null
done
paste -d'|' $OUT/${TAG}_conv_c2f_p0.txt $OUT/${TAG}_conv_c2f_p3.txt | grep -E "fwd|dgrad|^[GD]" | cut -c1-210
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), round(d["ms_per_step"],4))
        for k,v in d["kernels"].items():
            if "igemm_ws" in k: print("    %-50s %4.1f %6.3f ms %6.1f TF"%(k,v["calls_per_iter"],v["ms_per_iter"],v["executed_tflops"]))
    except Exception as e: print(f,"ERR",e)
PY
