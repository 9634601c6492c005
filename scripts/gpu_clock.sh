#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3v}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 200 python scripts/clock_by_kernel.py cfg2 2>&1 | tee $OUT/${TAG}_clock_cfg2.txt
timeout 200 python scripts/clock_by_kernel.py c2f 2>&1 | tee $OUT/${TAG}_clock_c2f.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-math --no-live-traffic > $OUT/${TAG}_bench.json 2>$OUT/${TAG}_bench.err
python - <<P
import json
d=json.loads(open("$OUT/${TAG}_bench.json").read().strip().splitlines()[-1])
print("cfg2", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("granted_clock_ghz"), d["roofline"].get("frac_at_granted_clock"))
c=d.get("c2f",{}); print("c2f", c.get("value"), c.get("ms_per_step"), c.get("step_roofline",{}).get("granted_clock_ghz"), c.get("roofline",{}).get("frac"), c.get("roofline",{}).get("frac_at_granted_clock"))
P
# the long-K shapes in the trace
rm -f $OUT/${TAG}_trace.txt
for shape in "128 64 64 128 256 5 0" "128 16 16 256 128 5 1"; do
  FG_WS_TRACE=1 FG_WS_TRACE_FILE=$OUT/${TAG}_trace.txt timeout 120 python scripts/bench_one.py fwd 3 0 $shape > /dev/null 2>&1
done
python scripts/ws_trace_report.py $OUT/${TAG}_trace.txt | grep -v "calib" | tee $OUT/${TAG}_trace_report.txt
gzip -f $OUT/${TAG}_trace.txt
