# PMC passes over a few iterations of the cfg2 step; per-kernel means for the tail kernels named in $1 (default: pack_jobs wgrad_finish_jobs)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
OUT=gpurun_out; TAG=${1:-pmc_tail}
rm -f $OUT/${TAG}.txt
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
  d=$OUT/pmc_$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $d -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
  echo "## $pmc" >> $OUT/${TAG}.txt
  for k in pack_jobs wgrad_finish_jobs actpool_bwd actpool_fwd adam_kernel bn_apply; do python scripts/pmc_summary.py $d $k >> $OUT/${TAG}.txt 2>&1; done
  rm -rf $d
done
cat $OUT/${TAG}.txt
