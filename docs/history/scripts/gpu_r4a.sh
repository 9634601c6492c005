#!/bin/bash
# round 4, first call: the whole -m gpu suite (new tests: fold precondition, Adam t=2 at B=128), then the default bench line
set -u
OUT=gpurun_out; TAG=${1:-r04a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
T0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q -x --durations=6 > $OUT/${TAG}_gpu_tests.log 2>&1
echo "suite rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
tail -12 $OUT/${TAG}_gpu_tests.log
timeout 400 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print("cfg2", round(d["value"],1), round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), "exec", d["step_roofline"].get("executed_frac"))
print("also", d["config"].get("also")); print("roofline.c2f", json.dumps(d["roofline"].get("c2f"))[:600])
for t in d["roofline"].get("hbm_tail", []): print("  tail %-26s x%.0f %8.1f us %6.2f TB/s %.2f" % (t["kernel"], t["launches_per_iter"], t["us"], t["tb_s"], t["frac_of_8TBs"]))
print(d["roofline"].get("hbm_tail_total"))
for t in d["c2f"].get("roofline",{}).get("hbm_tail", []): print("  c2f tail %-26s x%.0f %8.1f us %6.2f TB/s %.2f" % (t["kernel"], t["launches_per_iter"], t["us"], t["tb_s"], t["frac_of_8TBs"]))
PY
