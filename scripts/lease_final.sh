set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 1700 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r06_tests.log 2>&1; echo "tests rc=$? t=$(( $(date +%s) - T0 ))"; grep -n "^FAILED\|^ERROR" gpurun_out/r06_tests.log | head; tail -4 gpurun_out/r06_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1; echo "collect rc=$? t=$(( $(date +%s) - T0 ))"
python - <<'PY'
import json
for f in ("gpurun_out/r06_bench.json", "gpurun_out/r06_bench_c2f.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline", {})
        print(f, round(d["value"], 1), round(d["ms_per_step"], 4), {k: r.get(k) for k in ("frac", "achieved", "granted_clock_ghz", "frac_at_granted_clock", "survey_8d_frac", "traffic_over_algorithmic")})
        print("  dominant_launch", r.get("dominant_launch"))
        print("  hbm_tail_total", r.get("hbm_tail_total"))
        print("  cpu_baseline", d.get("cpu_baseline"))
        if "c2f" in d: print("  c2f", {k: d["c2f"].get(k) for k in ("value", "ms_per_step")})
        print("  step_roofline", {k: d["step_roofline"].get(k) for k in ("executed_frac", "algorithmic_frac_of_f32_mfma_peak", "granted_clock_ghz")})
    except Exception as e:
        print(f, "ERR", e)
PY
head -50 gpurun_out/r06_bench_kernel_stats.md
echo "done t=$(( $(date +%s) - T0 ))"
