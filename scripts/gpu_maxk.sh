#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3o}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
for mk in 0 1152; do
  FG_WS64_MAXK=$mk timeout 300 python scripts/bench_conv.py 5 c2f 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_conv_c2f_mk$mk.txt
  FG_WS64_MAXK=$mk timeout 200 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > $OUT/${TAG}_bench_c2f_mk$mk.json 2>/dev/null
done
FG_WS64_MAXK=1152 timeout 600 python -m pytest tests/test_gpu_c2f.py tests/test_gpu_baseline_sizes.py -m gpu -q -x -k "S64_forward_backward or S64_full_steps" > $OUT/${TAG}_tests.log 2>&1; echo "parity rc=$?"
paste -d'|' $OUT/${TAG}_conv_c2f_mk0.txt $OUT/${TAG}_conv_c2f_mk1152.txt | grep -E "fwd|dgrad|^[GD]" | grep -A2 "^D" | cut -c1-200
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), round(d["ms_per_step"],4))
    for k,v in d["kernels"].items():
        if "igemm_ws" in k: print("    %-50s %4.1f %6.3f ms %6.1f TF"%(k,v["calls_per_iter"],v["ms_per_iter"],v["executed_tflops"]))
PY
