#!/bin/bash
# round 4, call i: PReLU + MaxPool [+ Dropout] as one stage (FG_ACTMAXPOOL=0 switches back) -- c2f parity, A/B twice
set -u
OUT=gpurun_out; TAG=${1:-r04i}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_c2f.py tests/test_gpu_fusion.py tests/test_gpu_net.py "tests/test_gpu_baseline_sizes.py" tests/test_gpu_train_epoch.py -m gpu -q -k "c2f or fusion or S64 or prelu or maxpool or evaluate" > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$?" | tee $OUT/${TAG}_summary.txt; tail -4 $OUT/${TAG}_tests.log
for rep in 1 2 3; do
 for amp in 1 0; do
    FG_ACTMAXPOOL=$amp timeout 300 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-live-traffic --no-clock-probe > $OUT/${TAG}_b.json 2>/dev/null
    python - $TAG $amp <<'P'
import json,sys
d=json.loads(open("gpurun_out/%s_b.json" % sys.argv[1]).read().strip().splitlines()[-1])
t={x["kernel"]: x for x in d["roofline"].get("hbm_tail",[])}
k=d["kernels"]
fw=" ".join("%s=%.3f" % (n.split("/")[0][-22:], v["ms_per_iter"]) for n,v in k.items() if ("conv_fwd" in n and ("ws64x3" in n or "ws_act" in n or "igemm_ws_kernel" in n)))
print("actmaxpool=%s c2f %.1f img/s %.4f ms | tail %.3f ms | amp %s | %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["hbm_tail_total"]["ms_per_iter"], "%.1fus" % t["actmaxpool_fwd_kernel"]["us"] if "actmaxpool_fwd_kernel" in t else "-", fw))
P
 done
done 2>&1 | tee $OUT/${TAG}_bench.txt
