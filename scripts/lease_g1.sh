set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 300 python scripts/wino_error_hist.py > gpurun_out/r06_wino_error.txt 2>&1; echo "errhist rc=$? t=$(( $(date +%s) - T0 ))"; cat gpurun_out/r06_wino_error.txt | tail -20
timeout 600 python -m pytest tests/test_gpu_image_scale.py tests/test_gpu_t7_fixture.py tests/test_gpu_properties.py tests/test_gpu_wino.py -m gpu -q -x > gpurun_out/g1_new.log 2>&1; echo "new tests rc=$? t=$(( $(date +%s) - T0 ))"; tail -30 gpurun_out/g1_new.log
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/g1_all.log 2>&1; echo "all rc=$? t=$(( $(date +%s) - T0 ))"; grep -n "^FAILED\|^ERROR" gpurun_out/g1_all.log | head; tail -12 gpurun_out/g1_all.log
timeout 300 python bench.py --no-cpu-baseline --no-alt-math > gpurun_out/g1_bench.json 2> gpurun_out/g1_bench.err; echo "bench rc=$? t=$(( $(date +%s) - T0 ))"
python -c "
import json; d=json.loads(open('gpurun_out/g1_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); r=d['roofline']; print({k:r.get(k) for k in ('frac','achieved','granted_clock_ghz','iteration_average_clock_ghz','frac_at_granted_clock','survey_8d_frac','dominant_launch','traffic_over_algorithmic')})
print(r['hbm_tail_total'])
for t in r['hbm_tail'][:12]: print(t)
"
bash scripts/gpu.sh kstats g1 > gpurun_out/g1_kstats.txt 2>&1; head -60 gpurun_out/g1_bench_kernel_stats.md
echo "done t=$(( $(date +%s) - T0 ))"
