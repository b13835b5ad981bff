"""Debug helper: a panda closed-loop episode, then the planner's view at the last tick (joint positions against
their limits, contact forces in the real world, statistics of the rollouts' costs)."""
import sys, json, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import closed_loop
from m3p2i_aip_amd import _lib as L
e = int(sys.argv[1]); sc = sys.argv[2]; ticks = int(sys.argv[3])
rng = np.random.default_rng([77, e]); j = dict(cube=(0.0, 0.0) if e == 0 else tuple(rng.uniform(-0.02, 0.02, 2).tolist()))
keep = {}
orig_close = closed_loop.Tamp.close
def close(self):
    pl = self.motion_planner
    J = pl.cost_total.cpu().numpy()
    ch = pl._engine.cost_horizon.cpu().numpy()
    keep["J"] = (float(J.min()), float(np.median(J)), float(J.max()))
    keep["frac_ge_1000_any_step"] = float((ch >= 1000).any(axis=1).mean())
    keep["cost_h_best"] = ch[np.argmin(J)].round(3).tolist()
    keep["cost_h_null"] = ch[-1].round(3).tolist()
    keep["beta"] = pl.beta
    keep["mean_action0"] = pl.mean_action[0].cpu().numpy().round(3).tolist()
    keep["q"] = self.sim._dof_state[0, 0::2].cpu().numpy().round(4).tolist()
    keep["goal"] = self.task_planner.curr_goal.cpu().numpy().round(4).tolist()
    for a in ("table", "shelf_stand", "cubeB"):
        keep["F_" + a] = self.sim.get_actor_contact_forces_by_name(a, "box")[0].cpu().numpy().round(3).tolist()
    keep["cubeA"] = self.sim.get_actor_link_by_name("cubeA", "box")[0, :7].cpu().numpy().round(4).tolist()
    keep["cubeB"] = self.sim.get_actor_link_by_name("cubeB", "box")[0, :7].cpu().numpy().round(4).tolist()
    orig_close(self)
closed_loop.Tamp.close = close
r = closed_loop.run("config_panda", ["mppi.num_samples=4000", "mppi.horizon=20", f"mppi.halton_scramble={sc}"], ticks=ticks, jitter=j)
print(r["success"], r["timeline"], r["cube_to_goal_xy"], r["cube_height_above_goal"])
for k, v in keep.items():
    print(k, v)
