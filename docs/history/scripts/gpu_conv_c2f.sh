#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3i}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 300 python scripts/bench_conv.py 5 c2f > $OUT/${TAG}_conv_c2f.txt 2>&1
timeout 200 python scripts/bench_conv.py 10 > $OUT/${TAG}_conv_cfg2.txt 2>&1
cat $OUT/${TAG}_conv_c2f.txt $OUT/${TAG}_conv_cfg2.txt
