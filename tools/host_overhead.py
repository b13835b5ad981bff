import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
for K, T in ((64, 12), (2000, 30)):
    pl, sim, obj, _cfg = bench.build_tamp("point_env", "push", (-1.0, -1.0), False, K, 0, 1, T, "cuda:0")
    from m3p2i_aip_amd import sampling
    pl._ensure_noise()
    state = sim._dof_state[0]
    for _ in range(50): pl.command(state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): pl.command(state)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    e = pl._engine
    t0 = time.perf_counter()
    for _ in range(500): e.command()
    t_issue_e = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all_e = time.perf_counter() - t0
    print(f"K={K} T={T}: planner.command issue {t_issue/500*1e6:.1f} us, with sync {t_all/500*1e6:.1f} us | engine.command issue {t_issue_e/500*1e6:.1f} us, with sync {t_all_e/500*1e6:.1f} us")
