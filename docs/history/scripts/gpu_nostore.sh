#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3n}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
for ns in 0 1; do
  FG_DEBUG_NOSTORE=$ns timeout 300 python scripts/bench_conv.py 5 c2f 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_conv_c2f_ns$ns.txt
  FG_DEBUG_NOSTORE=$ns timeout 200 python scripts/bench_conv.py 10 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_conv_cfg2_ns$ns.txt
done
paste -d'|' $OUT/${TAG}_conv_c2f_ns0.txt $OUT/${TAG}_conv_c2f_ns1.txt | grep -E "fwd|dgrad|^[GD]" | cut -c1-200
paste -d'|' $OUT/${TAG}_conv_cfg2_ns0.txt $OUT/${TAG}_conv_cfg2_ns1.txt | grep -E "fwd|dgrad|^[gd]" | cut -c1-200
