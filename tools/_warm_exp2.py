import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
mode = sys.argv[1]
env, task, goal, mm, K, T = bench.CONFIGS["push"]
pl, sim, obj, cfg = bench.build_tamp(env, task, goal, mm, K, 0, 1, T, "cuda:0")
state = sim._dof_state[0]
for _ in range(5): pl.command(state)
out = []
for b in range(8):
    if mode == "gc": t=time.perf_counter(); gc.collect(); gct=time.perf_counter()-t
    if mode == "sleep": time.sleep(0.05)
    if mode == "sleep5": time.sleep(0.005)
    if mode == "warm2":
        time.sleep(0.05)
        for _ in range(2): pl.command(state)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pl.command(state)
    torch.cuda.synchronize(); out.append(round((time.perf_counter() - t0) / 20 * 1e3, 4))
print(mode, out, round(gct*1e3,1) if mode=="gc" else "")
