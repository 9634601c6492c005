#!/bin/bash
# round 3, fourth GPU call: where do the wave cycles of the weight-gradient kernels go?  SQ counter passes (PMC + kernel-trace only)
# on the dominant layer (scripts/bench_one.py: g9 shape) for the forward kernel (reference point), the symmetric wgrad kernel and
# the wave-specialised one.
set -u
OUT=gpurun_out
TAG=${1:-r3d}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rocprofv3 -L > $OUT/${TAG}_counters_avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $OUT/${TAG}_counters_avail.txt | sort -u | tr '\n' ' ' > $OUT/${TAG}_sq_names.txt
: > $OUT/${TAG}_pmc.txt
run() {   # which, ws, label, counters...
  which=$1; ws=$2; label=$3; shift 3
  d=$OUT/pmc_tmp
  rm -rf $d
  FG_WGRAD_WS=$ws timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -- python scripts/bench_one.py $which 4 > /dev/null 2>&1
  echo "## $label : $*" >> $OUT/${TAG}_pmc.txt
  python scripts/pmc_summary.py $d igemm_ws >> $OUT/${TAG}_pmc.txt 2>&1
  python scripts/pmc_summary.py $d wgrad >> $OUT/${TAG}_pmc.txt 2>&1
  rm -rf $d
}
for spec in "fwd 0 forward" "wgrad 0 wgrad_symmetric" "wgrad 1 wgrad_ws"; do
  set -- $spec
  run $1 $2 $3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
  run $1 $2 $3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
  run $1 $2 $3 SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES
done
cat $OUT/${TAG}_pmc.txt
