import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode in ("spin", "yield", "block"):
    hip = ctypes.CDLL("libamdhip64.so")
    flag = {"spin": 1, "yield": 2, "block": 4}[mode]
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(ctypes.c_uint(flag)))
import torch, bench
env, task, goal, mm, K, T = bench.CONFIGS["push"]
pl, sim, obj, cfg = bench.build_tamp(env, task, goal, mm, K, 0, 1, T, "cuda:0")
state = sim._dof_state[0]
for _ in range(30): pl.command(state)
out = []
for b in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pl.command(state)
    torch.cuda.synchronize(); out.append(round((time.perf_counter() - t0) / 20 * 1e3, 4))
# sync cost alone
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(200): torch.cuda.synchronize()
idle=(time.perf_counter()-t0)/200*1e6
# one command + sync latency
lat=[]
for _ in range(50):
    torch.cuda.synchronize(); t0=time.perf_counter(); pl.command(state); torch.cuda.synchronize(); lat.append((time.perf_counter()-t0)*1e3)
print(mode, out, "idle sync %.1f us"%idle, "single cmd+sync p50 %.4f ms"%sorted(lat)[25])
