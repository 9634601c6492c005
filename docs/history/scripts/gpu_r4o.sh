#!/bin/bash
# round 4, call o: Adam inside the re-pack launch (FG_ADAM_PACK=1, measured slower in round 3) again, now that the pack runs 6+ blocks per CU
set -u
OUT=gpurun_out; TAG=${1:-r04o}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
for rep in 1 2; do
 for ap in 1 0; do
  for wl in cfg2 c2f; do
    FG_ADAM_PACK=$ap timeout 300 python bench.py --workload $wl --steps $([ $wl = c2f ] && echo 10 || echo 50) --warmup $([ $wl = c2f ] && echo 3 || echo 10) --no-cpu-baseline --no-alt-math --no-live-traffic --no-clock-probe --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('adam_pack=$ap $wl %.1f img/s %.4f ms' % (d['value'], d['ms_per_step']))"
  done
 done
done 2>&1 | tee $OUT/${TAG}_bench.txt
