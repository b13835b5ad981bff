"""CPU BASELINE "B2" -- test / bench infrastructure, NOT product code (same rules as oracle/__init__.py).

The command() of the point_env planner in the REFERENCE'S LOOP SHAPE, on the host cores: a Python loop
over the horizon in which every quantity is its own small torch-CPU tensor op (what the reference does
on its GPU: ~180 aten launches per time step for push, ~540 for push_pull), wrapped
around one batched simulator step per time step (the oracle's C/OpenMP integrator in the role Isaac
Gym's PhysX step plays there: `gym.simulate` on all K environments, isaacgym_wrapper.py:354-360), then
the importance-weight update as tensor ops with the beta searches as Python `while` loops
(mppi.py:430-456, m3p2i.py:24-64).  It is what BASELINE.md section 3 calls the "reference-shaped CPU"
baseline: same algorithm and numbers as oracle.OraclePointPlanner (tests/test_refshaped_cpu.py compares
them), organised as the reference organises it, so that its time shows what that organisation costs on
this box -- next to the fused C port (`cpu_baseline.kind == "port"`).

Written against the formulas in SURVEY.md section 8(a) (A4, A5, A7, A8, A10-A12), not transcribed.
"""
from __future__ import annotations

import time

import numpy as np
import torch

import oracle as O


class RefShapedPointPlanner:
    def __init__(self, task, goal, multi_modal, K, T, delta, kp_suction=400.0, u_lim=3.0, sigma=3.0, gamma=0.95):
        self.task, self.multi_modal, self.K, self.T = task, bool(multi_modal), K, T
        self.half = K // 2
        self.goal = torch.tensor(list(goal)[:2], dtype=torch.float32)
        self.kp = float(kp_suction)
        self.sc = O.default_scene()
        self.delta = torch.from_numpy(np.array(delta, np.float32, copy=True))       # [K, T, 2]
        self.scale = torch.sqrt(torch.tensor([sigma, sigma]))
        self.u_min, self.u_max = torch.full((2,), -u_lim), torch.full((2,), u_lim)
        z = lambda: torch.zeros(T, 2)
        self.mean, self.mean_1, self.mean_2, self.best_1, self.best_2, self.best = z(), z(), z(), z(), z(), z()
        self.gamma_seq = torch.cumprod(torch.tensor([1.0] + [gamma] * (T - 1)), 0).view(1, T)
        self.beta = 1.0
        self.worlds = np.zeros((K, O.WORLD_FLOATS), np.float32)
        self.W = torch.from_numpy(self.worlds)          # zero-copy view, like gymtorch.wrap_tensor
        self.pending = torch.zeros(K, 4)                # suction staged by the last cost evaluation (Q5)

    # ---- pieces --------------------------------------------------------------------------------
    @staticmethod
    def _shift(seq):
        last = seq[-1].clone()
        seq = torch.roll(seq, -1, dims=0)
        seq[-1] = last
        return seq

    def _assemble(self):
        d = self.delta.clone()
        d[-1] = 0.0
        scaled = d * self.scale.view(1, 1, 2)
        if self.multi_modal:
            a1 = self.mean_1.unsqueeze(0) + scaled[:self.half]
            a2 = self.mean_2.unsqueeze(0) + scaled[self.half:]
            act = torch.cat([a1, a2], 0)
        else:
            act = self.mean.unsqueeze(0) + scaled
        act = torch.max(torch.min(act, self.u_max), self.u_min)
        if self.multi_modal:
            act[0] = self.best_1
            act[self.half] = self.best_2
        return act

    def _dist_terms(self):
        W = self.W
        robot = torch.stack([W[:, O.W_R], W[:, O.W_R + 1]], 1)
        box = torch.stack([W[:, O.W_B], W[:, O.W_B + 1]], 1)
        r2b = robot - box
        b2g = self.goal.view(1, 2) - box
        d_rb = torch.linalg.norm(r2b, dim=1)
        d_bg = torch.linalg.norm(b2g, dim=1)
        cos = torch.sum(r2b * b2g, 1) / (d_rb * d_bg)
        return robot, box, d_rb + 10.0 * d_bg, cos

    def _push_cost(self):
        _, _, dist, cos = self._dist_terms()
        return 3.0 * dist + torch.clamp(cos, min=0.0)

    def _pull_cost(self):
        W = self.W
        robot, box, dist, cos = self._dist_terms()
        vel = torch.stack([W[:, O.W_R + 4], W[:, O.W_R + 5]], 1)
        to_box = box - robot
        gap = torch.linalg.norm(to_box, dim=1)
        toward = torch.sum(vel * to_box, 1) > 0
        inv = (1.0 / gap).view(-1, 1)
        unit = to_box * inv
        active = (inv.view(-1) > (1.5 if self.K == 1 else 1.8))
        f_box = torch.zeros(self.K, 2)
        f_box[active] = -self.kp * unit[active]
        f_rob = torch.zeros(self.K, 2)
        f_rob[active] = self.kp * unit[active]
        forces = torch.clamp(torch.cat([f_rob, f_box], 1), -500.0, 500.0)
        forces[toward] = 0.0
        if self.multi_modal:
            forces[:self.half] = 0.0
        self.pending = forces                      # acts during the NEXT step
        near = toward & (gap <= 0.5)
        return 3.0 * dist + 3.0 * (0.6 * near.float()) + 7.0 * torch.clamp(-cos, min=0.0)

    def _cost(self):
        if self.task == "push":
            return self._push_cost()
        if self.task == "pull":
            return self._pull_cost()
        if self.task == "push_pull":
            return torch.cat([self._push_cost()[:self.half], self._pull_cost()[self.half:]])
        raise ValueError(self.task)

    @staticmethod
    def _search(J, lo=3.0, hi=10.0):
        J = J - torch.min(J)
        beta, n = 1.0, 1
        e = torch.exp(-J / beta)
        eta = float(torch.sum(e))
        while (eta > hi or eta < lo) and n < 1000:
            beta = beta * 0.9 if eta > hi else beta * 1.2
            e = torch.exp(-J / beta)
            eta = float(torch.sum(e))
            n += 1
        return e / eta

    # ---- command -------------------------------------------------------------------------------
    def command(self, world0):
        K, T = self.K, self.T
        self.mean = self._shift(self.mean)
        if self.multi_modal:
            self.mean_1, self.mean_2 = self._shift(self.mean_1), self._shift(self.mean_2)
            self.best_1, self.best_2 = self._shift(self.best_1), self._shift(self.best_2)
        act = self._assemble()
        self.worlds[:] = np.asarray(world0, np.float32).reshape(1, -1)[:, :O.WORLD_FLOATS]
        self.W[:, O.W_FEXT_R:O.W_FEXT_R + 4] = self.pending
        cost_h = torch.zeros(K, T)
        states, actions = [], []
        for t in range(T):
            u = act[:, t].clone()
            u[K - 1] = 0.0                                   # null action on the zero-noise sample
            O.step_batch(self.sc, self.worlds, u.numpy())    # "gym.simulate": all K environments, one call
            state = torch.stack([self.W[:, O.W_R], self.W[:, O.W_R + 4], self.W[:, O.W_R + 1], self.W[:, O.W_R + 5]], 1)
            c = self._cost()
            if self.task != "push":
                self.W[:, O.W_FEXT_R:O.W_FEXT_R + 4] = self.pending
            cost_h[:, t] = c
            states.append(state)
            actions.append(u)
        actions = torch.stack(actions, 1)
        states = torch.stack(states, 1)
        J = torch.sum(cost_h * self.gamma_seq, 1)
        if self.multi_modal:
            w = self._search(J)
            w1, w2 = self._search(J[:self.half]), self._search(J[self.half:])
            self.best_1 = actions[torch.argmax(w1)].clone()
            self.best_2 = actions[self.half + torch.argmax(w2)].clone()
            self.mean_1 = torch.sum(w1.view(-1, 1, 1) * actions[:self.half], 0)
            self.mean_2 = torch.sum(w2.view(-1, 1, 1) * actions[self.half:], 0)
        else:
            Js = J - torch.min(J)
            e = torch.exp(-Js / self.beta)
            w = e / torch.sum(e)
            self.best = actions[torch.argmax(w)].clone()
        new_mean = torch.sum(w.view(-1, 1, 1) * actions, 0)
        self.mean = 0.02 * self.mean + 0.98 * new_mean
        top = torch.topk(w, 20).indices
        self.top_trajs = states[top][:, :, [0, 2]]
        from scipy.signal import savgol_filter
        return savgol_filter(self.mean.numpy(), 9, 2, axis=0).astype(np.float32)


def time_commands(task, goal, multi_modal, K, T, delta, budget_s=7.0, threads=1):
    pl = RefShapedPointPlanner(task, goal, multi_modal, K, T, delta)
    w0 = O.init_world(1)[0]
    pl.command(w0)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and n < 500:
        pl.command(w0)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": K * T * n / dt, "unit": "state-steps/s", "ms_per_command": dt / n * 1e3, "calls": n,
            "cores": threads,
            "what": "per-t Python loop of per-op torch-CPU tensors around one batched simulator step per time step "
                    "(oracle/refshaped.py): the reference's loop structure on the host cores"}
