"""GPU experiment (needs a -DM3_ABL_PHASES build loaded through M3P2I_HIP_LIB, tools/flag_variants.sh):
where the point_env rollout kernel's time goes, per wave and per phase (shader-clock counter around every
phase of every time step; the counter reads themselves cost ~10 % of the kernel).

    tools/flag_variants.sh phases "-DM3_ABL_PHASES"            (build container)
    M3P2I_HIP_LIB=gpurun_variants/phases.so python tools/phase_breakdown.py push northstar   (GPU box)
"""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

PH = ["action assembly + prefetch", "broad-phase mask + dispatch", "forces + contact detection", "solver passes",
      "net force + integration", "task cost", "stores + accumulation", "-"]
out = {}
for name in sys.argv[1:] or ["push"]:
    env, task, goal, mm, K, T = bench.CONFIGS[name]
    pl, sim, obj, cfg = bench.build_tamp(env, task, goal, mm, K, 0, 1, T, "cuda:0")
    state = sim._dof_state[0]
    lib = ctypes.CDLL(os.environ["M3P2I_HIP_LIB"])
    buf = (ctypes.c_ulonglong * (1024 * 8))()
    for it in range(20):
        pl.command(state)
    torch.cuda.synchronize()
    lib.m3_dbg_phases(buf, 1)
    N = 200
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for it in range(N):
        pl.command(state)
    t1.record()
    torch.cuda.synchronize()
    lib.m3_dbg_phases(buf, 0)
    a = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(1024, 8) / N
    a = a[a.sum(1) > 0]
    tot = a.sum(1)
    worst = int(np.argmax(tot))
    res = {"K": K, "T": T, "waves": int(a.shape[0]), "command_ms_instrumented": t0.elapsed_time(t1) / N,
           "clock_ticks_per_command": {"mean_wave": float(tot.mean()), "slowest_wave": float(tot[worst])},
           "share_slowest_wave_pct": {PH[q]: round(100 * a[worst, q] / tot[worst], 1) for q in range(7)},
           "share_mean_wave_pct": {PH[q]: round(100 * a[:, q].mean() / tot.mean(), 1) for q in range(7)},
           "ticks_per_step_slowest_wave": {PH[q]: round(a[worst, q] / T, 1) for q in range(7)}}
    out[name] = res
    print(name, json.dumps(res, indent=1))
    pl._engine.close()
if os.path.isdir(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")):
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "phase_breakdown.json"), "w"), indent=1)
