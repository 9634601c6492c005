#!/bin/bash
# round 4, call l: SQ counters on D's mid-size convolutions (cfg2): instructions per MFMA, matrix-pipe busy share, waits
set -u
OUT=gpurun_out; TAG=${1:-r04l}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
: > $OUT/${TAG}_pmc.txt
run() { label=$1; which=$2; shape=$3; shift 3
  d=$OUT/pmc_tmp; rm -rf $d
  timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -- python scripts/bench_one.py $which 3 0 $shape > /dev/null 2>&1
  echo "## $label : $*" >> $OUT/${TAG}_pmc.txt
  python scripts/pmc_summary.py $d igemm >> $OUT/${TAG}_pmc.txt 2>&1
  rm -rf $d; }
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
Bc="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
Cc="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
for spec in "d13fwd fwd 128_4_4_256_512_3_0" "d9fwd fwd 128_8_8_128_256_3_0" "d13dgrad dgrad 128_4_4_256_512_3_0" "d5dgrad dgrad 128_16_16_64_128_3_0"; do
  set -- $spec; shape=$(echo $3 | tr '_' ' ')
  run $1 $2 "$shape" $A
  run $1 $2 "$shape" $Bc
  run $1 $2 "$shape" $Cc
done
cat $OUT/${TAG}_pmc.txt
