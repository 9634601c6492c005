#!/bin/bash
# round 4, call e: the generator lookahead (FG_FUSE_G_LOOKAHEAD): parity / bit-identity, then A/B on cfg2 (interleaved, 3 pairs)
set -u
OUT=gpurun_out; TAG=${1:-r04e}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_step_abi.py tests/test_gpu_c_host.py tests/test_gpu_train_epoch.py tests/test_golden.py tests/test_gpu_dist.py tests/test_gpu_dist2.py "tests/test_gpu_baseline_sizes.py::test_cfg2_full_step_at_batch_128" -m gpu -q > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$?" | tee $OUT/${TAG}_summary.txt; tail -5 $OUT/${TAG}_tests.log
export TAG
for rep in 1 2 3; do
  for la in 1 0; do
    FG_G_LOOKAHEAD=$la timeout 300 python bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-live-traffic > $OUT/${TAG}_b.json 2>/dev/null
    python - $la <<'P'
import json,sys,os
d=json.loads(open("gpurun_out/%s_b.json" % os.environ["TAG"]).read().strip().splitlines()[-1])
k=d["kernels"]
dom=[v for n,v in k.items() if n.startswith("igemm_ws_kernel<128>")]
print("lookahead=%s cfg2 %.1f img/s %.4f ms exec %.4f dom %.4f clock %s enq %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], d["step_roofline"].get("executed_frac",0), d["roofline"]["frac"], d["step_roofline"].get("granted_clock_ghz"), d["host_enqueue_ms_per_step"]))
P
  done
done 2>&1 | tee $OUT/${TAG}_ab.txt
