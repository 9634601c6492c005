#!/bin/bash
# micro-bench of the contraction kernels on the hot-path layer shapes under environment variants
set -u
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rm -f $OUT/conv_variants.txt
run() { L=$1; shift; echo "== $L" >> $OUT/conv_variants.txt; env "$@" timeout 120 python scripts/bench_conv.py 10 2>&1 | grep -v "^$" >> $OUT/conv_variants.txt; }
run "default" FG_X=0
run "wgrad slots 1024" FG_WGRAD_SLOTS=1024
run "wgrad slots 768" FG_WGRAD_SLOTS=768
run "wgrad slots 256" FG_WGRAD_SLOTS=256
cat $OUT/conv_variants.txt
