#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3q}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
for pf in 0 1; do
  FG_WS_PREFETCH=$pf timeout 300 python scripts/bench_conv.py 5 c2f 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_conv_c2f_pf$pf.txt
  FG_WS_PREFETCH=$pf timeout 200 python scripts/bench_conv.py 10 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_conv_cfg2_pf$pf.txt
done
paste -d'|' $OUT/${TAG}_conv_c2f_pf0.txt $OUT/${TAG}_conv_c2f_pf1.txt | grep -E "fwd|dgrad|^[GD]" | cut -c1-210
paste -d'|' $OUT/${TAG}_conv_cfg2_pf0.txt $OUT/${TAG}_conv_cfg2_pf1.txt | grep -E "fwd|dgrad|^[gd]" | cut -c1-210
