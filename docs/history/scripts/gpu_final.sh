#!/bin/bash
# round-end GPU call: the whole -m gpu suite, smoke(), then the evidence set (scripts/collect_profiles.sh)
set -u
OUT=gpurun_out; TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/${TAG}_gpu_tests.log 2>&1
echo "suite rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
timeout 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1
echo "smoke rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
bash scripts/collect_profiles.sh $TAG > $OUT/${TAG}_collect.log 2>&1
echo "collect rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
tail -4 $OUT/${TAG}_gpu_tests.log; tail -2 $OUT/${TAG}_smoke.log
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print("cfg2", round(d["value"],1), round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), d["roofline"]["traffic"], d["roofline"]["traffic_freshness"])
c=d["c2f"]; print("c2f", c.get("value"), c.get("ms_per_step"), c.get("error"), c.get("roofline",{}).get("frac"), c.get("step_roofline",{}).get("algorithmic_frac_of_f32_mfma_peak"))
PY
