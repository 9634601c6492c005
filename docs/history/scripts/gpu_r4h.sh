#!/bin/bash
# round 4, call h: loads-in-flight in the split / partial sums, Linear weight gradients on 64 x 64 tiles -- parity subset, A/B
set -u
OUT=gpurun_out; TAG=${1:-r04h}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_step_abi.py tests/test_gpu_net.py tests/test_gpu_c2f.py tests/test_golden.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -q > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$?" | tee $OUT/${TAG}_summary.txt; tail -3 $OUT/${TAG}_tests.log
for rep in 1 2; do
 for lw in 1 0; do
    FG_LINEAR_WGRAD64=$lw timeout 300 python bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-live-traffic --no-clock-probe > $OUT/${TAG}_b.json 2>/dev/null
    python - $TAG $lw <<'P'
import json,sys
d=json.loads(open("gpurun_out/%s_b.json" % sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels"]
lin=" ".join("%s=%.1fus" % (n.split("/")[0][:22], 1e3*v["ms_per_iter"]/v["calls_per_iter"]) for n,v in k.items() if "linear_wgrad" in n)
print("linear64=%s cfg2 %.1f img/s %.4f ms | %s" % (sys.argv[2], d["value"], d["ms_per_step"], lin))
P
 done
done 2>&1 | tee $OUT/${TAG}_bench.txt
timeout 300 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-live-traffic --no-clock-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2f %.1f img/s %.4f ms' % (d['value'], d['ms_per_step']))" | tee -a $OUT/${TAG}_bench.txt
rocprofv3 --kernel-trace --stats -d $OUT/p1 -o run -- python bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p1/run_results.db auto > $OUT/${TAG}_bench_kernel_stats.md; rm -rf $OUT/p1
grep -E "wgrad_finish|wgrad_kernel|actpool|sum_splits|pack_jobs|iterations" $OUT/${TAG}_bench_kernel_stats.md
