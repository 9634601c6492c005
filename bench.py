#!/usr/bin/env python
"""bench.py -- GAN train images/sec (G+D step) at 32x32x3, batch 128 per GPU (BASELINE.json config 2 / 3).

One "step" = one iteration of adversarial.lua:54-288: D_iterations x D-step (G forward on B/2 noises in train
mode -> D forward/BCE/backward on B/2 real || B/2 fake -> L2 penalty -> clamp -> Adam on D) + G_iterations x
G-step (G forward on B noises -> D forward/BCE(target 1)/backward-to-input -> G backward -> clamp -> Adam on G).
Synthetic inputs resident in HBM before the timed region (real ~ U[0,1), noise ~ U(-1,1) drawn on device by
Philox inside the step, dropout masks drawn on device), reference init N(0,0.005^2)/N(0,0.001^2) (train.lua:137).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  The default run (no --workload) times the headline workload (configs[1], 32x32) and then, in the
same process and the same JSON line, BASELINE configs[3] / [4] (c2f 64x64: B = 128, D_it = 1 on one GPU; 64 per GPU with D_it = 2 when
N > 1) as the sub-record "c2f" with its own value / ms_per_step / roofline / step_roofline / cpu_baseline.  N > 1: weak scaling (B=128 per GPU; --strong: B=128 globally), RCCL all-reduce (sum)
of the flat D / G gradient vectors each update through the library's own communicator (fg_comm_*), replicas identical.

Roofline keys (dominant kernel, HIP events on the launch stream, live in this run):
  roofline.achieved / frac      EXECUTED MFMA TFLOP/s of that kernel and its fraction of the fp32 MFMA peak (<= 1)
  roofline.algorithmic_*        the same time priced with the reference-formulation FLOPs of SURVEY 8(d) (un-folded 5x5
                                taps; nearest-x2 tap folding executes 9/25 of them, so this can exceed the peak)
  step_roofline.*               the whole iteration: algorithmic and executed FLOPs over the measured step time
"""
import argparse
import re
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_MFMA_TFLOPS = 2516.6   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense, spec (16x the fp32 form)
NOMINAL_CLOCK_GHZ = 2.4       # the clock the peaks below are quoted at (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec
PEAK_HBM_TBS = 8.0               # MI355X_MICROARCH.md: HBM3E, spec
# algorithmic FLOPs per image for cfg2/3: 7.962 GFLOP/img, 1019.19 GFLOP per B=128 iteration
MG, MD, G1, D1 = 1052934144, 59703808, 819200, 1769472
C2F_FLOP_PER_IMAGE = 37.220e9    # configs[3]: 4764.19 GFLOP per B=128 iteration


def alg_flops_per_iter(B, d_it=1, g_it=1):
    return 2.0 * (d_it * (B / 2 * MG + 3 * B * MD - B * D1) + g_it * (3 * B * MG - B * G1 + 2 * B * MD))


def parse_prof(text):
    rows = {}
    for line in text.strip().splitlines():
        parts = line.rsplit(None, 5)        # the label may contain blanks (template argument lists)
        if len(parts) != 6:
            continue
        name = parts[0].replace(" ", "")
        rows[name] = dict(calls=int(parts[1]), ms=float(parts[2]), alg=float(parts[3]), exe=float(parts[4]), bytes=float(parts[5]))
    return rows


def cpu_baselines(workload, batch):
    """The iteration restated on the host cores, two ways: primary = PyTorch-CPU fp32 (ATen / oneDNN, all
    cores), secondary = the numpy oracle (im2col + sgemm per layer, the THNN SpatialConvolutionMM algorithm).  The
    reference's own Lua/Torch `nn` path cannot run here (no Lua) -- both are restatements ("port")."""
    import numpy as np
    from oracle import torch_cpu as TC
    out = {}
    cores = os.cpu_count() or 1
    try:
        # The thread count matters more than the core count: oneDNN's small convolutions stop scaling (and then collapse)
        # long before a 256-thread pool is full -- 256 threads measured 2.5 img/s where 8 give ~65.  A short sweep
        # (bounded: ~4 s of timed work per point) picks the best setting; every point is reported.
        Bc = batch if workload == "cfg2" else min(batch, 32)
        avail = TC.usable_cores()
        pts = []
        for th in sorted(set(t for t in (8, 16, 32, 64) if t <= avail) | ({avail} if avail <= 96 else set())):
            r = TC.time_iterations(workload, Bc, min_seconds=4.0, max_iters=20, threads=th)
            pts.append(r)
            if len(pts) >= 2 and r["images_per_sec"] < 0.5 * max(p["images_per_sec"] for p in pts):
                break                                   # past the knee: more threads only get slower (and cost minutes)
        best = max(pts, key=lambda p: p["images_per_sec"])
        out["cpu_baseline"] = dict(value=best["images_per_sec"], unit="images/sec", cores=best["threads"], kind="port",
                                   sample="PyTorch-CPU fp32 (oracle/torch_cpu.py: ATen/oneDNN), full iterations (D-step + G-step) at "
                                          "batch %d, >= 4 s per point after 1 warm-up; threads -> img/s: %s; %d usable cores; restated CPU "
                                          "baseline -- the reference's Lua/Torch nn path is not executable in this environment"
                                          % (Bc, ", ".join("%d -> %.1f" % (p["threads"], p["images_per_sec"]) for p in pts), avail))
    except Exception as e:
        out["cpu_baseline"] = dict(value=None, unit="images/sec", cores=cores, kind="port", sample="failed: %s" % str(e)[:200])
    if workload == "cfg2":
        from oracle import torch7_nn as O
        rng = np.random.default_rng(1)
        G = O.create_G32((3, 32, 32), 100, rng, weight_init_=False)
        D = O.create_D32b((3, 32, 32), rng)
        O.initialize_weights(G, rng=rng)
        O.initialize_weights(D, rng=rng)
        st = O.GanState(G, D)
        B = 128 if cores >= 64 else 32
        real = rng.uniform(0, 1, (B // 2, 3, 32, 32)).astype(np.float32)
        t0 = time.time()
        O.step_D(st, real, rng.uniform(-1, 1, (B // 2, 100)).astype(np.float32))
        O.step_G(st, rng.uniform(-1, 1, (B, 100)).astype(np.float32))
        dt = time.time() - t0
        out["cpu_baseline_port"] = dict(value=B / dt, unit="images/sec", cores=cores, kind="port",
                                        sample="1 iteration at batch %d, numpy/OpenBLAS oracle (oracle/torch7_nn.py), %.1f s" % (B, dt))
    return out


def _sha16(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except Exception:
        return None


KERNEL_SOURCE = os.path.join(ROOT, "face_generator_amd", "csrc", "wino.hip")
# the dominant launch of the dominant kernel: nearest-x2 + 5x5 conv 256 -> 128 forward at B = 128 (models.lua:68-69), since round 5 a
# Winograd F(2x2, 3x3) contraction over the four output parities: reads the 33.5 MB input + 8.4 MB of transformed weights
# (4 parities x 16 positions x 256 x 128), writes the 67.1 MB output
DOMINANT_LAUNCH = dict(kernel="wino_kernel", launch="nearest-x2 + 5x5 conv 256->128 forward, B=128 (models.lua:68-69), Winograd F(2x2,3x3) "
                       "over four output parities",
                       algorithmic_bytes_per_launch=128 * 16 * 16 * 256 * 4 + 4 * 16 * 256 * 128 * 4 + 128 * 32 * 32 * 128 * 4 + 128 * 4,
                       bench_one=["fwd", "4"], trace_prefix="wino_kernel<0>")
# configs[3]: the 5x5 conv 128 -> 256 at 64x64 (models_c2f.lua:126) -- in the step it runs as wino_kernel<1> (the PReLU behind it in
# the epilogue: the pre-activation AND prelu(x) are stored); the module-level launch measured here is the same loop with the plain
# epilogue (wino_kernel<0>, one store of the output), so both byte counts are given
DOMINANT_LAUNCH_C2F = dict(kernel="wino_kernel", launch="5x5 conv 128->256 forward at 64x64, B=128 (models_c2f.lua:126): Winograd F(2x2,3x3) over "
                           "four 3x3 sub-kernels; measured on the plain-epilogue instantiation wino_kernel<0> (module-level launch); the "
                           "step's launch (wino_kernel<1>) also stores prelu(x): + 536.9 MB of writes",
                           algorithmic_bytes_per_launch=128 * 64 * 64 * 128 * 4 + 4 * 16 * 128 * 256 * 4 + 128 * 64 * 64 * 256 * 4 + 256 * 4,
                           algorithmic_bytes_with_fused_prelu_store=128 * 64 * 64 * 128 * 4 + 4 * 16 * 128 * 256 * 4 + 2 * 128 * 64 * 64 * 256 * 4 + 256 * 4,
                           bench_one=["fwd", "3", "0", "128", "64", "64", "128", "256", "5", "0"], trace_prefix="wino_kernel<0>")


def live_traffic(kernel, timeout=150, spec=None):
    """HBM bytes per launch of the dominant kernel, measured NOW: two rocprofv3 PMC passes (FETCH_SIZE, then WRITE_SIZE -- they do
    not fit one pass; --kernel-trace only, as MI355X_MICROARCH.md prescribes) over scripts/bench_one.py, which launches exactly the
    dominant launch.  FETCH_SIZE is doubled (gfx950: 128-byte requests tallied at 64 bytes for 16-byte-per-lane loads); both are KiB.
    Returns None when rocprofv3 is unavailable or a pass fails (the caller then labels a committed file as stale or not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    spec = spec or DOMINANT_LAUNCH
    if not kernel.startswith(spec["kernel"]) or shutil.which("rocprofv3") is None:
        return None
    vals = {}
    for pmc in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="fg_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(["rocprofv3", "--pmc", pmc, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                            sys.executable, os.path.join(ROOT, "scripts", "bench_one.py")] + spec["bench_one"],
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            tot, launches = 0.0, set()
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Kernel_Name"].startswith("void " + spec["trace_prefix"]) or r["Kernel_Name"].startswith(spec["trace_prefix"]):
                        if r["Counter_Name"] == pmc:
                            tot += float(r["Counter_Value"])
                            launches.add(r["Dispatch_Id"])
            if not launches:
                return None
            vals[pmc] = tot / len(launches)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    tj = {k: v for k, v in spec.items() if k not in ("bench_one", "trace_prefix")}
    tj.update(fetch_size_kb_raw=vals["FETCH_SIZE"], fetch_correction=2.0, write_size_kb=vals["WRITE_SIZE"],
              hbm_bytes_per_launch=2.0 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024,
              kernel_source_sha16=_sha16(KERNEL_SOURCE), freshness="live",
              source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over scripts/bench_one.py fwd, spawned by this "
                     "bench.py run")
    return tj


def load_traffic(kernel, live=True, spec=None):
    """HBM bytes per launch of the dominant kernel: measured by this run (live_traffic) when rocprofv3 is there, otherwise the newest
    committed PMC summary, labelled `stale` unless it was taken from the kernel source this run executes (sha of igemm.hip)."""
    if live:
        tj = live_traffic(kernel, spec=spec)
        if tj is not None:
            return tj
    if spec is not None and spec is not DOMINANT_LAUNCH:
        return None
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if tj.get("kernel") and kernel.startswith(tj["kernel"]):
            same = tj.get("kernel_source_sha16") is not None and tj.get("kernel_source_sha16") == _sha16(KERNEL_SOURCE)
            tj["freshness"] = "committed file profiles/%s, same kernel source as this run" % name if same else \
                              "stale: committed file profiles/%s from an earlier kernel source" % name
            return tj
    return None


def dominant_launch_clock(ctx, torch, spec, cover_ms=40.0):
    """The clock the chip grants the DOMINANT LAUNCH itself: the one-wave s_memtime / s_memrealtime probe on its own
    stream while nothing but that launch runs back to back (module-level entry, same shape and data distribution as in the step; its
    on-the-fly weight pack is < 2 % of the time).  The iteration-average clock (step_roofline.granted_clock_ghz) mixes in the
    low-power tail of the step and understates what the big launches get taken away.  -> (GHz, ms covered) or (None, None)."""
    import ctypes
    from face_generator_amd import ops
    a = spec["bench_one"]
    B, H, W, Cin, Cout, k, up = (128, 16, 16, 256, 128, 5, 1) if len(a) < 10 else tuple(int(v) for v in a[3:10])
    g = torch.Generator().manual_seed(0)
    d = ctx.device
    x = torch.randn(B, H, W, Cin, generator=g).to(d)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(d)
    b = torch.randn(Cout, generator=g).to(d)
    try:
        for _ in range(3):
            ops.conv2d_forward(x, w, b, upsample2x=bool(up), ctx=ctx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ops.conv2d_forward(x, w, b, upsample2x=bool(up), ctx=ctx)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / 5 * 1e3
        n = max(8, int(cover_ms / max(per, 1e-3)) + 4)
        ctx.check(ctx.lib.fg_prof_clock_start(ctx.h, ctypes.c_double(max(1.0, 0.85 * per * n))))
        for _ in range(n):
            ops.conv2d_forward(x, w, b, upsample2x=bool(up), ctx=ctx)
        torch.cuda.synchronize()
        ghz, cov = ctypes.c_double(0.0), ctypes.c_double(0.0)
        ctx.check(ctx.lib.fg_prof_clock_read(ctx.h, ctypes.byref(ghz), ctypes.byref(cov)))
        return (ghz.value, cov.value) if ghz.value > 0 else (None, None)
    except Exception:
        return None, None


STAGE = {"name": "start", "t0": time.time()}


def stage(name):
    """Where the run is (read by the watchdog of an N > 1 job: a partial JSON line names the stage that did not finish)."""
    STAGE["name"], STAGE["t0"] = name, time.time()


def measure(args, ctx, tr, iteration, torch, dist, world, rank, B, flops_per_iter, out, workload, steps, warmup, alt_math=True):
    """Timed region + roofline leg shared by both workloads.  Fills `out` in place."""
    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(v):
        if world > 1:
            t = torch.tensor([v], dtype=torch.float64, device=ctx.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return v

    stage(workload + ": warm-up")
    for _ in range(warmup):
        iteration()
    tr.finish_pending()
    sync_all()
    stage(workload + ": timed steps")
    t0 = time.perf_counter()
    for _ in range(steps):
        iteration()
    tr.finish_pending()          # N > 1: the last D update is deferred behind the next G forward -- complete it in-region
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0          # this rank's own clock, before the closing barrier: a straggler shows here
    sync_all()
    dt = max_over_ranks(time.perf_counter() - t0)
    ms = 1000.0 * dt / steps
    out.update(value=world * B * steps / dt, ms_per_step=ms)
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=ctx.device)
        t[rank] = 1000.0 * dt_local / steps
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out["per_rank_ms_per_step"] = [round(v, 4) for v in t.tolist()]
    stage(workload + ": roofline / baseline legs")
    # host cost of a step (outside the timed region): wall time to ENQUEUE a short burst of steps on an idle queue, no
    # sync inside -- short enough that the launch queue never back-pressures, so it is the pure host-side cost
    nb = min(4, steps)
    tb = time.perf_counter()
    for _ in range(nb):
        iteration()
    t_enq = time.perf_counter() - tb
    tr.finish_pending()
    sync_all()
    out["host_enqueue_ms_per_step"] = 1000.0 * max_over_ranks(t_enq) / nb
    out["step_roofline"] = {"algorithmic_gflop_per_iter": flops_per_iter / 1e9,
                            "algorithmic_tflops_per_gpu": flops_per_iter / (ms * 1e-3) / 1e12,
                            "algorithmic_frac_of_f32_mfma_peak": flops_per_iter / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}

    if not args.no_roofline:
        # every rank runs the extra iterations (they contain collectives); only rank 0 records HIP events
        import ctypes
        if rank == 0:
            ctx.check(ctx.lib.fg_prof_enable(ctx.h, 1))
        for _ in range(args.prof_iters):
            iteration()
        tr.finish_pending()
        sync_all()
        if not args.no_clock_probe:
            # the clock the chip grants this workload, in a leg of its own (beside the HIP-event leg the probe's queue cost the
            # dominant kernel 3 %): a one-wave probe on its own stream, asleep while the same iterations run again
            if rank == 0:
                ctx.check(ctx.lib.fg_prof_enable(ctx.h, 0))
                ctx.check(ctx.lib.fg_prof_clock_start(ctx.h, ctypes.c_double(max(1.0, 0.9 * ms * args.prof_iters))))
            for _ in range(args.prof_iters):
                iteration()
            tr.finish_pending()
            sync_all()
            if rank == 0:
                ghz, cov = ctypes.c_double(0.0), ctypes.c_double(0.0)
                ctx.check(ctx.lib.fg_prof_clock_read(ctx.h, ctypes.byref(ghz), ctypes.byref(cov)))
                if ghz.value > 0:
                    out["step_roofline"].update({
                        "granted_clock_ghz": ghz.value, "nominal_clock_ghz": NOMINAL_CLOCK_GHZ, "clock_probe_ms": cov.value,
                        "clock_note": "shader cycles (s_memtime) / 100 MHz ticks (s_memrealtime) read by a sleeping one-wave probe while "
                                      "prof_iters more iterations run; the part clocks to its power budget, the peaks here are at 2.4 GHz"})
    if rank == 0 and not args.no_roofline:
        buf = ctypes.create_string_buffer(1 << 16)
        ctx.check(ctx.lib.fg_prof_report(ctx.h, buf, len(buf), 1))
        ctx.check(ctx.lib.fg_prof_enable(ctx.h, 0))
        rows = parse_prof(buf.value.decode())
        sym = {}                                   # aggregate by kernel symbol (what rocprofv3 --stats reports)
        for name, r in rows.items():
            k = name.split("/")[0]
            a = sym.setdefault(k, dict(calls=0, ms=0.0, alg=0.0, exe=0.0, bytes=0.0))
            for f in ("calls", "ms", "alg", "exe", "bytes"):
                a[f] += r[f]
        if sym:
            sym = {k: v for k, v in sym.items() if v["exe"] > 0} or sym     # contraction launches (the tail is reported apart)
            exe_iter = sum(v["exe"] for v in sym.values()) / args.prof_iters
            mfma_ms_iter = sum(v["ms"] for v in sym.values() if v["exe"] > 0) / args.prof_iters
            out["step_roofline"].update({
                "executed_gflop_per_iter": exe_iter / 1e9,
                "executed_tflops_per_gpu": exe_iter / (ms * 1e-3) / 1e12,
                "executed_frac": exe_iter / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "contraction_kernel_ms_per_iter": mfma_ms_iter,
                "note": "executed = MFMA FLOPs the contraction kernels issue (igemm / wgrad launches; the 3-channel thin "
                        "convolutions and the 512->1 head, < 1 % of the FLOPs, are not counted)"})
            dom = max(sym, key=lambda k: sym[k]["ms"])
            a = sym[dom]
            alg = a["alg"] / (a["ms"] * 1e-3) / 1e12
            exe = a["exe"] / (a["ms"] * 1e-3) / 1e12
            tj = None
            spec_dom = DOMINANT_LAUNCH if workload == "cfg2" else DOMINANT_LAUNCH_C2F
            if world == 1:
                tj = load_traffic(dom, live=not args.no_live_traffic, spec=spec_dom)
            # the dominant kernel's biggest launch (by time) and the clock the chip grants THAT launch
            dom_rows = {n: r for n, r in rows.items() if n.split("/")[0] == dom and r["exe"] > 0 and r["ms"] > 0}
            big = max(dom_rows, key=lambda n: dom_rows[n]["ms"] / dom_rows[n]["calls"]) if dom_rows else None
            kghz, kcov = (None, None)
            if not args.no_clock_probe and world == 1:
                kghz, kcov = dominant_launch_clock(ctx, torch, spec_dom)
            out["roofline"] = {"bound": "mfma", "achieved": exe, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": exe / PEAK_F32_MFMA_TFLOPS,
                               "traffic": tj["hbm_bytes_per_launch"] if tj else None,
                               "algorithmic_bytes": tj.get("algorithmic_bytes_per_launch") if tj else None,
                               "traffic_over_algorithmic": (tj["hbm_bytes_per_launch"] / tj["algorithmic_bytes_per_launch"])
                               if tj and tj.get("algorithmic_bytes_per_launch") else None,
                               "traffic_note": (tj["launch"] + "; " + tj["source"]) if tj else None,
                               "algorithmic_bytes_with_fused_prelu_store": tj.get("algorithmic_bytes_with_fused_prelu_store") if tj else None,
                               "traffic_freshness": tj.get("freshness") if tj else None,
                               # SURVEY 8(d)'s own definition next to it: reference-formulation FLOPs of the WHOLE step / wall time /
                               # 157.3 -- above 1 since Winograd (36 -> 16 multiplies) and the x2-upsample tap fold (100 -> 16): an
                               # algorithmic reduction, not pipe utilisation (`frac` is that)
                               "survey_8d_frac": out["step_roofline"]["algorithmic_frac_of_f32_mfma_peak"],
                               "granted_clock_ghz": kghz,
                               "granted_clock_source": ("one-wave s_memtime probe while ONLY the dominant launch (%s) runs back to back, %.0f ms"
                                                        % (spec_dom["launch"].split(",")[0], kcov)) if kghz else None,
                               "iteration_average_clock_ghz": out["step_roofline"].get("granted_clock_ghz"),
                               "frac_at_granted_clock": (exe / (PEAK_F32_MFMA_TFLOPS * kghz / NOMINAL_CLOCK_GHZ)) if kghz else None,
                               "dominant_launch": ({"label": big, "launches_per_iter": dom_rows[big]["calls"] / args.prof_iters,
                                                    "avg_launch_ms": dom_rows[big]["ms"] / dom_rows[big]["calls"],
                                                    "executed_tflops": dom_rows[big]["exe"] / (dom_rows[big]["ms"] * 1e-3) / 1e12,
                                                    "frac": dom_rows[big]["exe"] / (dom_rows[big]["ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                                    "frac_at_its_own_clock": (dom_rows[big]["exe"] / (dom_rows[big]["ms"] * 1e-3) / 1e12
                                                                              / (PEAK_F32_MFMA_TFLOPS * kghz / NOMINAL_CLOCK_GHZ)) if kghz else None}
                                                   if big else None),
                               "kernel": dom, "avg_launch_ms": a["ms"] / a["calls"],
                               "launches_per_iter": a["calls"] / args.prof_iters,
                               "algorithmic_tflops": alg, "algorithmic_frac": alg / PEAK_F32_MFMA_TFLOPS,
                               "note": "achieved / frac = MFMA FLOPs this kernel executes on LIVE tiles x LIVE channels (padding not "
                                       "credited) / its HIP-event time (all its launches of the iteration); algorithmic_* prices the "
                                       "same time with the reference-formulation (un-folded 5x5) FLOPs of SURVEY 8(d); frac_at_granted_clock "
                                       "uses the clock granted to the dominant launch itself, not the iteration average"}
            if "ws6" in dom:      # bf16x6 kernels issue 6 bf16 MFMA flops per fp32-equivalent flop: price against the bf16 pipe too
                out["roofline"].update({"bf16_issued_tflops": 6.0 * exe, "bf16_dense_peak": PEAK_BF16_MFMA_TFLOPS,
                                        "bf16_issued_frac": 6.0 * exe / PEAK_BF16_MFMA_TFLOPS})
            # the HBM-bound tail, separately -- algorithmic bytes (every operand once) of each
            # pointwise / thin launch over its HIP-event time, against the 8 TB/s HBM peak
            def padded(k):       # thin_in<7,3> / thin_out<5,3> / thin_wgrad<7,4>: the 5x5 / 7x7 layers with <= 4 channels on one side
                m = re.match(r"thin_(?:in|out|wgrad)<(\d+),", k)
                return bool(m) and int(m.group(1)) >= 5
            pad = [{"kernel": k, "launches_per_iter": round(v["calls"] / args.prof_iters, 2), "us": round(1e3 * v["ms"] / v["calls"], 2),
                    "ms_per_iter": round(v["ms"] / args.prof_iters, 4), "useful_gflop_per_launch": round(v["alg"] / v["calls"] / 1e9, 3),
                    "useful_tflops": round(v["alg"] / (v["ms"] * 1e-3) / 1e12, 2),
                    "useful_frac_of_f32_mfma_peak": round(v["alg"] / (v["ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
                   for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]) if padded(k) and v["ms"] > 0]
            if pad:
                out["roofline"]["mfma_padded"] = pad
                out["roofline"]["mfma_padded_note"] = ("bound: the fp32 matrix pipe on tiles padded to the MFMA shape (a 7x7x3 window = 147 of 160 "
                                                       "K columns, 21 of 32 N columns live): priced by useful FLOPs / time / 157.3, not in TB/s")
            tail = [{"kernel": k, "launches_per_iter": round(v["calls"] / args.prof_iters, 2),
                     "bytes_per_launch": round(v["bytes"] / v["calls"]), "us": round(1e3 * v["ms"] / v["calls"], 2),
                     "ms_per_iter": round(v["ms"] / args.prof_iters, 4),
                     "tb_s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e12, 3),
                     "frac_of_8TBs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e12 / PEAK_HBM_TBS, 4)}
                    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]) if v["bytes"] > 0 and v["ms"] > 0 and not padded(k)]
            if tail:
                tb, tm = sum(t["bytes_per_launch"] * t["launches_per_iter"] for t in tail), sum(t["ms_per_iter"] for t in tail)
                out["roofline"]["hbm_tail"] = tail
                out["roofline"]["hbm_tail_total"] = {"ms_per_iter": tm, "gb_per_iter": tb / 1e9, "tb_s": tb / (tm * 1e-3) / 1e12,
                                                     "frac_of_8TBs": tb / (tm * 1e-3) / 1e12 / PEAK_HBM_TBS, "bound": "hbm", "peak_tb_s": PEAK_HBM_TBS,
                                                     "note": "launches shorter than ~8 us sit on the launch floor, not on HBM"}
            out["kernels"] = {k: {"calls_per_iter": v["calls"] / args.prof_iters, "ms_per_iter": v["ms"] / args.prof_iters,
                                  "executed_tflops": v["exe"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0}
                              for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]) if v["exe"] > 0 or v["bytes"] == 0}
    # (the supplementary bf16x6 leg runs AFTER the roofline leg: it is power-bound and leaves the part at a lower clock for the
    # next few hundred ms -- measured right behind it the fp32 kernels read 6-8 % slow)
    alt = None
    if args.math == "f32" and not args.no_alt_math and alt_math:
        # supplementary, never the headline: the same K steps with the large contractions in bf16x6 (fp32 emulated with
        # six exact split-bf16 plane products, include/facegen_hip.h fg_set_math); every rank runs it (collectives)
        try:
            ctx.set_math(6)
            for _ in range(min(warmup, 4)):
                iteration()
            tr.finish_pending()
            sync_all()
            t1 = time.perf_counter()
            for _ in range(steps):
                iteration()
            tr.finish_pending()
            sync_all()
            dt6 = max_over_ranks(time.perf_counter() - t1)
            alt = {"math": "bf16x6 (fp32 emulated: 6 exact bf16 split-plane products, fp32 accumulate; fg_set_math(ctx, 6))",
                   "value": world * B * steps / dt6, "unit": "images/sec", "ms_per_step": 1000.0 * dt6 / steps,
                   "note": "opt-in mode, parity-tested at the same tolerances (FG_MATH=6 pytest -m gpu); not the headline value"}
        except Exception as e:      # supplementary only: never let it take the headline measurement down
            alt = {"math": "bf16x6", "error": str(e)[:200]}
        finally:
            ctx.set_math(0)

    if alt is not None:
        out["alt_math"] = alt
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out.update(cpu_baselines(workload, B))


def timed_leg(ctx, torch, dist, world, rank, tr, iteration, steps, warmup):
    """`steps` iterations bracketed like the headline (barrier + synchronize on both sides, MAX over ranks) plus every rank's own
    step time; no roofline leg.  Returns (ms_per_step, [per-rank ms])."""
    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    for _ in range(warmup):
        iteration()
    tr.finish_pending()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        iteration()
    tr.finish_pending()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    sync_all()
    dt = time.perf_counter() - t0
    per = [1000.0 * dt_local / steps]
    if world > 1:
        t = torch.zeros(world + 1, dtype=torch.float64, device=ctx.device)
        t[rank] = 1000.0 * dt_local / steps
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        m = torch.tensor([dt], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        dt = float(m.item())
        per = [round(v, 4) for v in t.tolist()[:world]]
    return 1000.0 * dt / steps, per


def multi_gpu_extras(args, ctx, torch, dist, coll, world, rank, B, out):
    """N > 1 only: what a first SCALE record needs to be decisive, in the same driver-timed run --
      * `compute_only`: the SAME per-rank iteration with no communicator attached (no exchange, no deferred update): every rank's
        N = 1-equivalent compute time; compute_over_step = max over ranks / the headline's ms_per_step is the scaling efficiency
        the exchange leaves, independent of any N = 1 run on another box;
      * `sync_bn` (or `per_gpu_bn` when the headline ran --sync-bn): the other BatchNorm mode -- per-GPU statistics are the
        throughput mode, sync-BN is the reference's B_global semantics; both numbers in one line;
      * `strong`: BASELINE configs[1]'s global batch of 128 split over the ranks (the weak line is the headline).
    Each leg builds its own nets and is bracketed like the headline; a failure is recorded, never raised."""
    ex = {}
    steps, warm = max(3, min(args.steps, 20)), max(1, min(args.warmup, 3))

    def leg(name, build, with_coll, note):
        stage("multi-GPU extra: " + name)
        try:
            w = build(coll if with_coll else None)
            ms, per = timed_leg(ctx, torch, dist, world, rank, w["tr"], w["iteration"], steps, warm)
            bpg = w["config"]["batch_per_gpu"]
            ex[name] = {"value": world * bpg * 1000.0 / ms, "unit": "images/sec", "ms_per_step": ms, "per_rank_ms_per_step": per,
                        "batch_per_gpu": bpg, "steps": steps, "warmup": warm, "note": note}
            del w
            torch.cuda.empty_cache()
        except Exception as e:
            ex[name] = {"error": str(e)[:300]}

    class A(object):       # args with one field changed
        def __init__(self, **kw):
            self.__dict__.update(vars(args)); self.__dict__.update(kw)
    leg("compute_only", lambda c: build_cfg2(args, ctx, torch, None, 1, rank, B), False,
        "no communicator: each rank's own iteration at its per-GPU batch (replicas diverge -- timing only); `value` is the "
        "no-exchange aggregate N x B / max-rank time")
    if "ms_per_step" in ex.get("compute_only", {}) and out.get("ms_per_step"):
        c = ex["compute_only"]
        out["per_rank_compute_ms"] = c["per_rank_ms_per_step"]
        out["compute_over_step"] = max(c["per_rank_ms_per_step"]) / out["ms_per_step"]
        out["exchange_exposed_ms_per_step"] = out["ms_per_step"] - max(c["per_rank_ms_per_step"])
    other = "per_gpu_bn" if args.sync_bn else "sync_bn"
    leg(other, lambda c: build_cfg2(A(sync_bn=not args.sync_bn), ctx, torch, c, world, rank, B), True,
        "the other BatchNorm mode of the same weak-scaling workload (headline: %s)" % ("sync-BN" if args.sync_bn else "per-GPU statistics"))
    if not args.strong and 128 % (2 * world) == 0:
        leg("strong", lambda c: build_cfg2(A(strong=True), ctx, torch, c, world, rank, 128 // world), True,
            "strong scaling: the global batch of 128 (BASELINE configs[1]) split over the ranks, %d per GPU" % (128 // world))
    out["multi_gpu"] = ex


C2F_D2_FLOP_PER_IMAGE = 45.130e9   # configs[4]: D_iterations = 2 -> 2888.30 GFLOP per B=64 iteration


def build_cfg2(args, ctx, torch, coll, world, rank, B):
    """configs[1] / [2]: G32 + D32b at 32x32x3 (models.lua:57-81, 382-416), reference init, Adam, D_it = G_it = 1."""
    from face_generator_amd import models, nn_utils, adversarial
    from face_generator_amd.state import S
    C = 3
    gen = torch.Generator().manual_seed(1)              # identical initial replicas on every rank
    G = models.create_G((C, 32, 32), 100)
    D = models.create_D((C, 32, 32))
    nn_utils.initializeWeights(D, gen=gen)
    nn_utils.initializeWeights(G, gen=gen)
    G.cuda(ctx, max_batch=B)
    D.cuda(ctx, max_batch=B)
    S.OPT.update(batchSize=B, noiseDim=100)
    S.noise_seed = 1 + rank                             # each rank draws its own shard of the global batch
    G.device_net.mask_seed = D.device_net.mask_seed = 1000 + rank
    S.OPT["sync_bn"] = bool(args.sync_bn)
    tr = adversarial.Trainer(ctx, G, D, S.OPT, dist=coll)
    real = ctx.uniform((B // 2, 32, 32, C), 0.0, 1.0, seed=77 + rank)

    def iteration():
        if tr.gan is not None:        # one C call per closure (fg_step_D / fg_step_G); noise + masks drawn inside it
            tr.step_D(real, None)
            tr.step_G(B)
        else:
            tr.step_D(real, S.next_noise(ctx, B // 2, 100))
            tr.step_G(S.next_noise(ctx, B, 100))
    return dict(tr=tr, iteration=iteration, flops=alg_flops_per_iter(B), real=real,
                data="synthetic (U[0,1) images, U(-1,1) noise, reference init N(0,.005^2)/N(0,.001^2))",
                config={"workload": "configs[1]: 32x32 color, noiseDim=100, batch %d per GPU, Adam, D_it=G_it=1" % B
                                    + ("" if world == 1 else "; configs[2]-style %s scaling, RCCL grad all-reduce"
                                       % ("strong" if args.strong else "weak")),
                        "batch_per_gpu": B, "global_batch": world * B, "parallelism": "dp%d" % world,
                        "batchnorm": "sync (global-batch statistics)" if (args.sync_bn and world > 1) else "per-GPU statistics"})


def build_c2f(args, ctx, torch, coll, world, rank, B, d_it):
    """configs[3] / [4]: coarse-to-fine G_d + D_c at 64x64x3 (models_c2f.lua:113-145, 237-278), Torch default init, Adam."""
    from face_generator_amd import models_c2f, adversarial_c2f
    from face_generator_amd.state import S
    Sz = 64
    gen = torch.Generator().manual_seed(1)
    G = models_c2f.create_G((3, Sz, Sz), gen=gen).cuda(ctx, max_batch=B)
    D = models_c2f.create_D((3, Sz, Sz), gen=gen).cuda(ctx, max_batch=B)
    S.noise_seed = 1 + rank
    G.inner.device_net.mask_seed = D.inner.device_net.mask_seed = 1000 + rank
    tr = adversarial_c2f.TrainerC2F(ctx, G, D, dict(batchSize=B), dist=coll)
    fine = ctx.uniform((B, Sz, Sz, 3), 0.0, 1.0, seed=70 + rank)
    from face_generator_amd import dataset_c2f
    coarse, diff = dataset_c2f.toResultDevice(fine, Sz // 2, ctx=ctx)    # dataset_c2f.lua:49-61 through fg_c2f_coarse_diff (image.scale)
    h = B // 2
    diff_r, coarse_r, coarse_f = diff[:h].contiguous(), coarse[:h].contiguous(), coarse[h:].contiguous()

    def iteration():
        for _ in range(d_it):             # adversarial_c2f.lua:123-160: D_iterations closures per G closure
            if tr.gan is not None:        # one C call per closure; noise planes and dropout masks drawn inside it
                tr.step_D(diff_r, coarse_r, None, coarse_f)
            else:
                tr.step_D(diff_r, coarse_r, S.next_noise(ctx, h, Sz * Sz).view(h, Sz, Sz, 1), coarse_f)
        if tr.gan is not None:
            tr.step_G(None, coarse)
        else:
            tr.step_G(S.next_noise(ctx, B, Sz * Sz).view(B, Sz, Sz, 1), coarse)
    which = "configs[3]" if (world == 1 and d_it == 1) else "configs[4]-style"
    return dict(tr=tr, iteration=iteration, flops=(C2F_FLOP_PER_IMAGE if d_it == 1 else C2F_D2_FLOP_PER_IMAGE) * B,
                data="synthetic (U[0,1) fine images, coarse = image.scale down to 32 and back up to 64 (fg_c2f_coarse_diff), diff = fine - coarse, U(-1,1) noise planes)",
                config={"workload": "%s: 64x64 color coarse-to-fine G_d/D_c, batch %d per GPU, Adam, D_it=%d, G_it=1" % (which, B, d_it),
                        "batch_per_gpu": B, "global_batch": world * B, "parallelism": "dp%d" % world})


def dry_collective(args):
    """--dry-collective: everything a `--gpus N` job does on the HOST side of the gradient exchange, with no kernel launched and
    no collective issued (planning-only context + dry communicators, include/facegen_hip.h).  For every rank of the job and every
    combination of {maxAccuracyD gate: none / passes / holds} x {sync_bn} x {overlap}, one iteration (D closure, G closure, the
    deferred D update) runs through fg_step_D / fg_step_G and the schedule the library recorded is printed: order, dtype, element
    count and stream of every all-reduce.  All ranks must agree; under torch.distributed.run (gloo) every process walks its own
    rank and rank 0 compares, otherwise one process walks all N ranks."""
    import torch
    import torch.distributed as dist
    from face_generator_amd import models, nn_utils, adversarial, distributed
    from face_generator_amd.runtime import get_context
    from face_generator_amd.state import S
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    N = max(args.gpus, world_env)
    if world_env > 1:
        dist.init_process_group("gloo")
        my_ranks = [int(os.environ["RANK"])]
    else:
        my_ranks = list(range(N))
    ctx = get_context(-1)
    B = min(args.batch, 8)                 # the schedule does not depend on the batch; keep the host buffers small
    combos = [(g, sb, ov) for g in ("none", "pass", "hold") for sb in (0, 1) for ov in (0, 1)]
    result = {}
    for r in my_ranks:
        coll = distributed.DryCollective(ctx, r, N)
        gen = torch.Generator().manual_seed(1)
        G = models.create_G((3, 32, 32), 100)
        D = models.create_D((3, 32, 32))
        nn_utils.initializeWeights(D, gen=gen)
        nn_utils.initializeWeights(G, gen=gen)
        G.cuda(ctx, max_batch=B)
        D.cuda(ctx, max_batch=B)
        real = ctx.zeros(B // 2, 32, 32, 3)
        per = {}
        for (gate, sb, ov) in combos:
            opt = dict(batchSize=B, noiseDim=100, sync_bn=bool(sb))
            tr = adversarial.Trainer(ctx, G, D, opt, dist=coll)
            assert tr.gan is not None, "the step-level entries must carry the closures"
            tr.gan.set_comm(coll, sync_bn=bool(sb), overlap=ov)
            coll.schedule(reset=True)
            g = None if gate == "none" else ((lambda acc: True) if gate == "pass" else (lambda acc: False))
            tr.step_D(real, None, gate=g)
            tr.step_G(B)
            tr.finish_pending()
            per["gate=%s sync_bn=%d overlap=%d" % (gate, sb, ov)] = coll.schedule(reset=True)
            del tr
        result[r] = per
        coll.close()
    if world_env > 1:
        gathered = [None] * world_env
        dist.all_gather_object(gathered, result)
        result = {}
        for g in gathered:
            result.update(g)
        dist.barrier()
        dist.destroy_process_group()
        if int(os.environ["RANK"]) != 0:
            return
    ranks = sorted(result)
    agree = all(result[r] == result[ranks[0]] for r in ranks)
    out = {"dry_collective": True, "n_gpus": N, "ranks_walked": ranks, "ranks_agree": agree, "schedule": result[ranks[0]],
           "note": "one iteration (D closure, G closure, deferred D update) per combination; lines are <seq> <op> <dtype> <count> <stream>"}
    if not agree:
        out["per_rank"] = {str(r): result[r] for r in ranks}
    print(json.dumps(out))
    if not agree:
        raise SystemExit(3)


def step_entry(tr):
    return ("fg_step_D / fg_step_G (C ABI, one call per closure)" if tr.gan is not None
            else "net-level entries driven from the host loop")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (OPT.batchSize); with --strong the GLOBAL batch")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --batch is the global batch, split over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--prof-iters", type=int, default=3)
    ap.add_argument("--sync-bn", action="store_true",
                    help="N > 1: all-reduce the BatchNorm sums (exact B_global statistics) instead of per-GPU statistics")
    ap.add_argument("--math", choices=["f32", "bf16x6"], default="f32",
                    help="arithmetic of the large conv contractions: f32 = native fp32 MFMA (default, the headline); "
                         "bf16x6 = fp32 emulated on the bf16 matrix pipe with six exact split-plane products (fg_set_math)")
    ap.add_argument("--no-alt-math", action="store_true", help="skip the supplementary bf16x6 timing")
    ap.add_argument("--collective", choices=["auto", "fg_comm", "torch"], default="auto",
                    help="N > 1: gradient all-reduce through the library's own RCCL communicator (fg_comm_*, default) or "
                         "torch.distributed; auto = fg_comm, falling back (and saying so) if its self-test fails")
    ap.add_argument("--workload", choices=["both", "cfg2", "c2f"], default="both",
                    help="both (default): the headline cfg2 line plus the c2f sub-record; cfg2: 32x32 G32+D32b only (BASELINE "
                         "configs[1]); c2f: 64x64 coarse-to-fine (configs[3]) as the line itself")
    ap.add_argument("--c2f-steps", type=int, default=10, help="timed steps of the c2f sub-record (3 warm-up steps)")
    ap.add_argument("--c2f-timeout", type=float, default=240.0,
                    help="N > 1: if the c2f sub-record has not finished after this many seconds, rank 0 prints the line without it")
    ap.add_argument("--no-clock-probe", action="store_true", help="do not run the one-wave clock probe beside the roofline leg")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not spawn the two rocprofv3 PMC passes for roofline.traffic (a committed summary is used and labelled)")
    ap.add_argument("--watchdog", type=float, default=360.0,
                    help="N > 1: seconds the headline leg may take from process start (rendezvous, RCCL init, self-test, warm-up, timed "
                         "steps); past it rank 0 prints a PARTIAL JSON line naming the stage that did not finish and every rank exits 4")
    ap.add_argument("--no-multi-gpu-extras", action="store_true",
                    help="N > 1: skip the compute-only / other-BatchNorm-mode / strong-scaling legs behind the headline (multi_gpu record)")
    ap.add_argument("--extras-timeout", type=float, default=240.0, help="N > 1: seconds the multi_gpu legs may take in total")
    ap.add_argument("--no-dry-check", action="store_true",
                    help="N > 1: do not walk the collective schedule of all N ranks on the CPU first (rank 0, a subprocess, ~5 s)")
    ap.add_argument("--dry-collective", action="store_true",
                    help="no GPU needed: walk the gradient-exchange path of a --gpus N job for every rank with planning-only contexts "
                         "and print the collective schedule (order, dtype, count, stream) every rank would issue; exit 3 on a mismatch")
    args = ap.parse_args()
    if args.dry_collective:
        return dry_collective(args)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (one process per GPU)" % args.gpus)
    # test hook (tests/test_gpu_dist2.py): FG_BENCH_TEST_GLOO=1 runs every rank on device 0 with gloo collectives, so the
    # multi-rank control flow of this file can be exercised on a one-GPU box (RCCL refuses two ranks on one device)
    test_gloo = os.environ.get("FG_BENCH_TEST_GLOO") == "1"
    if test_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    from face_generator_amd import _lib as _fg_lib
    out = {"library": _fg_lib.lib_path()}          # the shared object this run loads (in-tree; FACEGEN_HIP_LIB overrides)
    watchdog = None
    dry = None
    if world > 1:
        import threading

        def bark():
            # an N > 1 job that hangs (rendezvous, ncclCommInitRank, a collective no other rank joins) must not run to the
            # driver's limit without a word: say where it stopped, with whatever was measured so far
            if rank == 0:
                part = dict(out)
                part.update(error="stage '%s' did not finish: %.1f s in it, watchdog %.1f s from process start"
                                  % (STAGE["name"], time.time() - STAGE["t0"], args.watchdog),
                            stage=STAGE["name"], partial=True, n_gpus=world, dry_collective=dry)
                part.setdefault("metric", "GAN train images/sec (G+D step) at 32x32x3 bs128")
                part.setdefault("value", None)
                # what a reader of a PARTIAL line needs to tell "RCCL never came up" from "a collective hung later": None = the run
                # did not get that far (the stage says where it stopped)
                for k in ("collective", "collective_fallback", "rccl_ranks_seen", "library"):
                    part.setdefault(k, None)
                print(json.dumps(part), flush=True)
            os._exit(4)
        watchdog = threading.Timer(args.watchdog, bark)
        watchdog.daemon = True
        watchdog.start()
        if rank == 0 and not args.no_dry_check:
            # the one thing that can hang an N-GPU run -- ranks issuing different collectives -- checked first, without a GPU:
            # every rank's fg_step_D / fg_step_G walked by planning-only contexts in a CPU subprocess
            stage("dry collective schedule (CPU)")
            import hashlib
            import subprocess
            try:
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                                                                        "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                                        "TORCHELASTIC_RUN_ID", "FG_BENCH_TEST_GLOO")}
                env["HIP_VISIBLE_DEVICES"] = ""
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--dry-collective"], env=env,
                                   capture_output=True, text=True, timeout=90)
                dj = json.loads(r.stdout.strip().splitlines()[-1])
                key = "gate=none sync_bn=%d overlap=1" % (1 if args.sync_bn else 0)
                dry = {"ranks_agree": dj["ranks_agree"], "ranks_walked": len(dj["ranks_walked"]),
                       "schedule_sha16": hashlib.sha256(json.dumps(dj["schedule"], sort_keys=True).encode()).hexdigest()[:16],
                       "this_run": dj["schedule"].get(key), "combination": key}
            except Exception as e:          # a diagnostic: never take the measurement down
                dry = {"error": str(e)[:200]}
        stage("torch.distributed rendezvous (init_process_group)")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if test_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from face_generator_amd import distributed
    from face_generator_amd.runtime import get_context

    ctx = get_context(local_rank)
    ctx.set_math(6 if args.math == "bf16x6" else 0)
    if args.strong:
        if args.batch % (2 * world):
            raise SystemExit("--strong: the global batch %d must split into even per-GPU batches over %d ranks" % (args.batch, world))
        B = args.batch // world
    else:
        B = args.batch
    coll = None
    if world > 1:
        stage("fg_comm create + self-test (RCCL communicator of the library)")
        coll = distributed.make_collective(ctx, dist, prefer="torch" if (test_gloo or args.collective == "torch") else "fg_comm",
                                           strict=args.collective == "fg_comm")
    headline = "c2f" if args.workload == "c2f" else "cfg2"
    out.update({
        "metric": "GAN train images/sec (G+D step) at 32x32x3 bs128" if headline == "cfg2"
                  else "GAN train images/sec (G+D step), c2f 64x64x3 bs128",
        "value": None, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": None, "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
        "dtype": "f32" if args.math == "f32" else "f32 (emulated: 6 exact bf16 split-plane products, fp32 accumulate)",
    })
    if world > 1:
        out["dry_collective"] = dry
        out["collective"] = coll.describe()
        # loud, top-level: the gradient exchange is NOT the library's own communicator (Trainer then drives the closures from
        # the host instead of fg_step_D / fg_step_G -- config.step_entry says so too)
        out["collective_fallback"] = bool(coll.fallback)
        out["rccl_ranks_seen"] = coll.ranks_seen
    stage(headline + ": build nets")
    if headline == "c2f":
        w = build_c2f(args, ctx, torch, coll, world, rank, B, 1)
    else:
        w = build_cfg2(args, ctx, torch, coll, world, rank, B)
    tr = w["tr"]
    out["data"] = w["data"]
    out["config"] = w["config"]
    out["config"]["step_entry"] = step_entry(tr)
    measure(args, ctx, tr, w["iteration"], torch, dist, world, rank, B, w["flops"], out, headline, args.steps, args.warmup)
    if headline == "cfg2" and world == 1:
        # the reference's boundary hands over HOST tensors (dataset[i]:clone() into a FloatTensor, adversarial.lua:244-249):
        # the same K steps with the real half coming from host memory every iteration (NCHW FloatTensor -> H2D -> NHWC);
        # never `value` -- the PCIe-inclusive rate of the contract
        real_host = w["real"].permute(0, 3, 1, 2).contiguous().cpu().pin_memory()
        for _ in range(3):
            tr.step_D(ctx.to_device_nhwc(real_host), None); tr.step_G(B)
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(args.steps):
            tr.step_D(ctx.to_device_nhwc(real_host), None); tr.step_G(B)
        torch.cuda.synchronize()
        out["host_input_images_per_sec"] = B * args.steps / (time.perf_counter() - th)
    out["reference_accounting_images_per_sec"] = out["value"] / 2   # adversarial.lua:305 counts B/2 per iteration
    if watchdog is not None:
        watchdog.cancel()          # the headline is measured; the supplementary legs have their own timers below
    if world > 1 and headline == "cfg2" and not args.no_multi_gpu_extras:
        import threading
        ex_done = threading.Event()

        def ex_bail():             # a stalled extra leg must not cost the headline: print what exists and leave (every rank)
            if not ex_done.is_set():
                if rank == 0:
                    out.setdefault("multi_gpu", {})["error"] = "stage '%s' did not finish within %.0f s" % (STAGE["name"], args.extras_timeout)
                    print(json.dumps(out), flush=True)
                os._exit(0)
        ex_timer = threading.Timer(args.extras_timeout, ex_bail)
        ex_timer.daemon = True
        ex_timer.start()
        del w, tr
        torch.cuda.empty_cache()
        multi_gpu_extras(args, ctx, torch, dist, coll, world, rank, B, out)
        ex_done.set()
        ex_timer.cancel()
        w = tr = None

    if args.workload == "both":
        # BASELINE configs[3] (one GPU: B = 128, D_it = 1) / configs[4] (N > 1: 64 per GPU, D_it = 2) in the same driver-timed run
        import threading
        done = threading.Event()
        if world > 1:
            def bail():
                # the headline must not be lost to a stall of the supplementary leg: print it and leave (every rank has this timer)
                if not done.is_set():
                    if rank == 0:
                        out["c2f"] = {"error": "the c2f leg did not finish within %.0f s" % args.c2f_timeout}
                        print(json.dumps(out), flush=True)
                    os._exit(0)
            timer = threading.Timer(args.c2f_timeout, bail)
            timer.daemon = True
            timer.start()
        sub = {}
        try:
            w = tr = None
            torch.cuda.empty_cache()
            cB, d_it = (B, 1) if world == 1 else (max(2, B // 2), 2)
            cw = build_c2f(args, ctx, torch, coll, world, rank, cB, d_it)
            sub = {"metric": "GAN train images/sec (G+D step), c2f 64x64x3", "value": None, "unit": "images/sec",
                   "steps": args.c2f_steps, "warmup": 3, "ms_per_step": None, "dtype": out["dtype"], "data": cw["data"],
                   "config": cw["config"]}
            sub["config"]["step_entry"] = step_entry(cw["tr"])
            measure(args, ctx, cw["tr"], cw["iteration"], torch, dist, world, rank, cB, cw["flops"], sub, "c2f", args.c2f_steps, 3,
                    alt_math=False)
            if world > 1 and not args.no_multi_gpu_extras:
                del cw
                torch.cuda.empty_cache()
                stage("c2f: compute-only leg")
                cl = build_c2f(args, ctx, torch, None, 1, rank, cB, d_it)
                _, per = timed_leg(ctx, torch, dist, world, rank, cl["tr"], cl["iteration"], max(1, min(args.c2f_steps, 5)), 1)
                sub["per_rank_compute_ms"] = per
                sub["compute_over_step"] = max(per) / sub["ms_per_step"]
                sub["exchange_exposed_ms_per_step"] = sub["ms_per_step"] - max(per)
        except Exception as e:          # supplementary: never let it take the headline measurement down
            sub["error"] = str(e)[:300]
        done.set()
        out["c2f"] = sub
        # the same numbers where a driver that keeps only `config` / `roofline` / `cpu_baseline` still sees them
        if sub.get("value") is not None:
            sr, rr = sub.get("step_roofline", {}), sub.get("roofline", {})
            out["config"]["also"] = "%s -> roofline.c2f" % sub["config"]["workload"]
            if isinstance(out.get("roofline"), dict):
                out["roofline"]["c2f"] = {
                    "workload": sub["config"]["workload"], "value": sub["value"], "unit": "images/sec", "ms_per_step": sub["ms_per_step"],
                    "steps": sub["steps"], "warmup": sub["warmup"],
                    "algorithmic_frac": sr.get("algorithmic_frac_of_f32_mfma_peak"), "executed_frac": sr.get("executed_frac"),
                    "granted_clock_ghz": sr.get("granted_clock_ghz"),
                    "target_80pct": {"images_per_sec": 0.8 * PEAK_F32_MFMA_TFLOPS * 1e12 / C2F_FLOP_PER_IMAGE if d_it == 1 else None,
                                     "ms_per_step": (cB * C2F_FLOP_PER_IMAGE / (0.8 * PEAK_F32_MFMA_TFLOPS * 1e12) * 1e3) if d_it == 1 else None},
                    "dominant_kernel": {k: rr.get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms", "launches_per_iter",
                                                               "frac_at_granted_clock")},
                    # HBM bytes of the dominant launch from this run's own PMC passes
                    "traffic": rr.get("traffic"), "algorithmic_bytes": rr.get("algorithmic_bytes"),
                    "algorithmic_bytes_with_fused_prelu_store": rr.get("algorithmic_bytes_with_fused_prelu_store"),
                    "traffic_over_algorithmic": rr.get("traffic_over_algorithmic"), "traffic_freshness": rr.get("traffic_freshness"),
                    "traffic_note": rr.get("traffic_note"),
                    "mfma_padded": rr.get("mfma_padded"), "mfma_padded_note": rr.get("mfma_padded_note"),
                    "hbm_tail_total": rr.get("hbm_tail_total")}
            if isinstance(out.get("cpu_baseline"), dict) and isinstance(sub.get("cpu_baseline"), dict):
                out["cpu_baseline"]["c2f"] = {k: sub["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind")}
    if world > 1:
        dist.barrier()
        if coll is not None:
            coll.close()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
