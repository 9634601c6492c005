#!/bin/bash
# SQ instruction / wait counters on the weak c2f layers (one pass per group)
set -u
OUT=gpurun_out; TAG=${1:-r3j}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
: > $OUT/${TAG}_pmc.txt
run() { label=$1; which=$2; shape=$3; shift 3
  d=$OUT/pmc_tmp; rm -rf $d
  timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -- python scripts/bench_one.py $which 3 0 $shape > /dev/null 2>&1
  echo "## $label : $*" >> $OUT/${TAG}_pmc.txt
  python scripts/pmc_summary.py $d igemm >> $OUT/${TAG}_pmc.txt 2>&1
  python scripts/pmc_summary.py $d wgrad_ >> $OUT/${TAG}_pmc.txt 2>&1
  rm -rf $d; }
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
Bc="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
for spec in "G2fwd fwd 128_64_64_64_64_3_0" "D4fwd fwd 128_32_32_128_256_3_0" "G4fwd fwd 128_64_64_128_256_5_0" "G3wgrad wgrad 128_64_64_64_128_5_0" "G2wgrad wgrad 128_64_64_64_64_3_0"; do
  set -- $spec; shape=$(echo $3 | tr '_' ' ')
  run $1 $2 "$shape" $A
  run $1 $2 "$shape" $Bc
done
cat $OUT/${TAG}_pmc.txt
