"""GPU: the bit-exactness fuzz tests of tests/test_hip_parity_point.py / test_hip_parity_panda.py (random worlds:
bodies overlapping each other, the obstacle and the walls, rotated, moving; random tasks and controls) for many more
seeds than the test suite runs.  Every state / action / cost of the HIP rollout must equal the oracle's bit for bit.
    python tools/fuzz_parity.py [first_seed=16] [n_point=400] [n_panda=200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tests import test_hip_parity_point as tp, test_hip_parity_panda as tq

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n2 = int(sys.argv[3]) if len(sys.argv) > 3 else 200
oracle.build()
bad = []
t0 = time.time()
def panda(o, seed):     # the three forms of the Panda kernel in turn (lanes per sample: world spec v3)
    tq.test_panda_rollout_bit_exact_on_random_worlds(o, seed, (16, 8, 1)[seed % 3])


for name, fn, n in (("point", tp.test_rollout_bit_exact_on_random_worlds, n1), ("panda", panda, n2)):
    for seed in range(s0, s0 + n):
        try:
            fn(oracle, seed)
        except AssertionError as e:
            bad.append((name, seed, str(e)[:300]))
            print("MISMATCH", name, seed, str(e)[:300], flush=True)
    print(f"{name}: seeds {s0}..{s0 + n - 1} done, {len([b for b in bad if b[0] == name])} mismatches, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
