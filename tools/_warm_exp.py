import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
env, task, goal, mm, K, T = bench.CONFIGS["push"]
pl, sim, obj, cfg = bench.build_tamp(env, task, goal, mm, K, 0, 1, T, "cuda:0")
state = sim._dof_state[0]
if pre:
    x = torch.randn(4096, 4096, device="cuda:0")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < pre * 1e-3:
        y = x @ x
    torch.cuda.synchronize()
for _ in range(5): pl.command(state)
out = []
for b in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pl.command(state)
    torch.cuda.synchronize(); out.append(round((time.perf_counter() - t0) / 20 * 1e3, 4))
print("pre", pre, out)
