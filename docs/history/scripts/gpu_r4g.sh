#!/bin/bash
# round 4, call g: re-pack patches split over several blocks (FG_PACK_SPLIT=1 switches back) -- parity subset, A/B twice
set -u
OUT=gpurun_out; TAG=${1:-r04g}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_step_abi.py tests/test_gpu_net.py tests/test_gpu_c2f.py tests/test_golden.py tests/test_gpu_conv_upsample.py -m gpu -q > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$?" | tee $OUT/${TAG}_summary.txt; tail -3 $OUT/${TAG}_tests.log
for rep in 1 2; do
 for sp in -1 1 2; do
  for wl in cfg2 c2f; do
    FG_PACK_SPLIT=$sp timeout 300 python bench.py --workload $wl --steps $([ $wl = c2f ] && echo 10 || echo 50) --warmup $([ $wl = c2f ] && echo 3 || echo 10) --no-cpu-baseline --no-alt-math --no-live-traffic --no-clock-probe > $OUT/${TAG}_b.json 2>/dev/null
    python - $wl $TAG $sp <<'P'
import json,sys
d=json.loads(open("gpurun_out/%s_b.json" % sys.argv[2]).read().strip().splitlines()[-1])
t=[x for x in d["roofline"].get("hbm_tail",[]) if "pack" in x["kernel"]]
print("split=%s %s %.1f img/s %.4f ms | pack %s" % (sys.argv[3], sys.argv[1], d["value"], d["ms_per_step"], ["%.1fus %.2fTB/s" % (x["us"], x["tb_s"]) for x in t]))
P
  done
 done
done 2>&1 | tee $OUT/${TAG}_bench.txt
