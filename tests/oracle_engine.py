"""Test infrastructure: an object with the HipEngine phase interface (rollout / update /
finalize + buffers) whose arithmetic is the CPU oracle.  It lets the world_size-2 gloo tests
exercise the HOST-side sharding logic (planner.py, distributed.py: sample offsets, the
all-gather of trajectory costs, the packed all-reduce, owner-only rows) on a machine without
a GPU.  It is never importable from the product package."""
import numpy as np
import torch

import oracle as O
from m3p2i_aip_amd import _lib as L


def _t(a):
    return torch.from_numpy(a)


class OracleEngine:
    def __init__(self, cfg: L.Config):
        self.cfg = cfg
        c = cfg
        self.Kg, self.Kl, self.k0, self.T, self.nu = c.K_global, c.K_local, c.k_offset, c.T, c.nu
        T, nu, Kl, Kg = self.T, self.nu, self.Kl, self.Kg
        f = np.float32
        self.np = {
            L.BUF_STATES: np.zeros((T, Kl, 4), f), L.BUF_ACTIONS: np.zeros((T, Kl, nu), f),
            L.BUF_COST_HORIZON: np.zeros((T, Kl), f), L.BUF_TRAJ_COST: np.zeros(Kl, f),
            L.BUF_TRAJ_COST_ALL: np.zeros(Kg, f), L.BUF_WEIGHTS: np.zeros(Kg, f),
            L.BUF_WEIGHTS_1: np.zeros(Kg // 2, f), L.BUF_WEIGHTS_2: np.zeros(Kg - Kg // 2, f),
            L.BUF_TOP_IDX: np.zeros(L.TOPK, np.int32), L.BUF_TOP_TRAJS: np.zeros((L.TOPK, T, 2), f),
            L.BUF_REDUCE: np.zeros(6 * T * nu + L.TOPK * T * 2, f),
            L.BUF_PENDING_FORCE: np.zeros((4, Kl), f),
            L.BUF_COV: np.array([[c.noise_sigma_diag[j] for j in range(nu)],
                                 [np.sqrt(f(c.noise_sigma_diag[j])) for j in range(nu)]], f),   # cov_action | scale_tril
        }
        if c.mode_simple or c.sampling_random:
            raise NotImplementedError("OracleEngine: halton-spline planners with an explicit noise table only")
        self.cov_active = bool(c.update_cov) and not c.multi_modal
        for b in range(L.BUF_MEAN, L.BUF_ACTION_OUT + 1):
            self.np[b] = np.zeros((T, nu), f)
        self.regen = bool(c.shard_mix) and Kl != Kg and bool(c.multi_modal) and not c.mode_simple
        self.mix = bool(c.shard_mix) and Kl != Kg and not self.regen
        self.HDR, self.RL = 48, 48 + 6 * T * nu + L.TOPK * T * 2   # record layout: m3_internal.hpp
        if self.regen:   # {costs of the shard | top-k costs | top-k global indices | top-k trajectories}, padded to 4
            self.RL = (Kl + 2 * L.TOPK + L.TOPK * T * 2 + 4 + 96 * 3 + 3) // 4 * 4   # (+ minima, ladder table: shard_mix 2)
            self.np[L.BUF_RECORD] = np.zeros(self.RL, f)
            self.np[L.BUF_RECORDS_ALL] = np.zeros((Kg // Kl, self.RL), f)
            self.np[L.BUF_TRAJ_COST] = self.np[L.BUF_RECORD][:Kl]     # alias, as in the library
        if self.mix:
            self.np[L.BUF_RECORD] = np.zeros(self.RL, f)
            self.np[L.BUF_RECORDS_ALL] = np.zeros((Kg // Kl, self.RL), f)
        self.p3 = self.regen and int(c.shard_mix) == 3     # two small exchanges (m3_update_b in between)
        if self.p3:
            self.RLB = (8 + 6 * T * nu + 3) // 4 * 4        # record B: m3_internal.hpp recb_length
            self.np[L.BUF_RECORD_B] = np.zeros(self.RLB, f)
            self.np[L.BUF_RECORDS_B_ALL] = np.zeros((Kg // Kl, self.RLB), f)
        self.t = {k: _t(v) for k, v in self.np.items()}
        self.sc = O.default_scene()
        self.device = torch.device("cpu")
        self.delta = None
        self.beta = 1.0
        self._info = None
        self.task, self.goal, self.grip = "push", (0.0, 0.0), 0
        self.bound = None
        self.calls = 0

    # ---- interface used by planner.py / distributed.py ----
    def use_torch_stream(self, stream=None):
        pass

    def buffer(self, which):
        return self.t[which]

    @property
    def needs_global_noise(self):
        return self.regen

    def set_noise(self, delta):
        d = np.ascontiguousarray(delta.cpu().numpy() if torch.is_tensor(delta) else delta, dtype=np.float32)
        if self.regen:   # every rank holds the rows of all samples
            assert d.shape == (self.Kg, self.T, self.nu)
            self.delta_all = d
            d = np.ascontiguousarray(d[self.k0:self.k0 + self.Kl])
        self.delta = d
        assert self.delta.shape == (self.Kl, self.T, self.nu)
        self.np[L.BUF_NOISE] = np.ascontiguousarray(self.delta.transpose(1, 0, 2))
        self.t[L.BUF_NOISE] = _t(self.np[L.BUF_NOISE])

    def set_noise_knots(self, knots, degree=2, smoothing=0.5):
        """The device sampler's role, on the host: scipy's FITPACK spline per series (sampling.smoothing_spline)."""
        from m3p2i_aip_amd import sampling
        k = np.asarray(knots.detach().cpu().numpy() if torch.is_tensor(knots) else knots, np.float32)
        assert smoothing == 0.5
        self.set_noise(sampling._spline_rows(k, self.T, degree))

    def relabel_samples(self):
        pass   # a wavefront-placement matter of the HIP kernels; labels stay as generated here

    def set_objective(self, task, goal, gripper_cmd=0):
        self.task, self.goal, self.grip = task, tuple(goal), gripper_cmd

    def bind_sim_point(self, dof, root, box_actor, dyn_actor):
        self.bound = (dof, root, box_actor, dyn_actor)

    def set_beta(self, beta):
        self.beta = float(beta)

    def set_call_count(self, calls):
        self.calls = int(calls)

    def reset(self):
        for b in list(range(L.BUF_MEAN, L.BUF_ACTION_OUT + 1)) + [L.BUF_PENDING_FORCE]:
            self.np[b][...] = 0
        self.beta, self.calls = 1.0, 0

    def _ocfg(self):
        c = self.cfg
        o = O.make_cfg(self.Kg, self.T, self.nu, multi_modal=bool(c.multi_modal), task=self.task,
                       goal=self.goal, u_min=list(c.u_min)[:self.nu], u_max=list(c.u_max)[:self.nu],
                       noise_sigma_diag=list(c.noise_sigma_diag)[:self.nu], u_scale=c.u_scale,
                       gamma=c.gamma, lambda_=c.lambda_, sample_null_action=bool(c.sample_null_action),
                       filter_u=bool(c.filter_u), kp_suction=c.kp_suction)
        for j in range(self.nu):
            o.scale_tril[j] = float(self.np[L.BUF_COV][1, j])    # (update_cov rewrites it: mppi.py:516)
        return o

    def _world0(self):
        dof, root, bi, di = self.bound
        d, r = dof[0].numpy(), root[0].numpy()
        w = O.init_world(1)[0]
        w[0], w[1], w[4], w[5] = d[0], d[2], d[1], d[3]
        for base, a in ((O.W_B, bi), (O.W_D, di)):
            qz, qw = r[a, 5], r[a, 6]
            w[base:base + 7] = (r[a, 0], r[a, 1], 1 - 2 * qz * qz, 2 * qz * qw, r[a, 7], r[a, 8], r[a, 12])
        return w

    def rollout(self):
        cfg = self._ocfg()
        k0, k1, T, nu = self.k0, self.k0 + self.Kl, self.T, self.nu
        full = np.zeros((self.Kg, T, nu), np.float32)
        full[k0:k1] = self.delta
        sh = np.minimum(np.arange(1, T + 1), T - 1)
        n = self.np
        act = O.assemble_actions(cfg, full, n[L.BUF_MEAN][sh], n[L.BUF_MEAN_1][sh], n[L.BUF_MEAN_2][sh],
                                 n[L.BUF_BEST_1][sh], n[L.BUF_BEST_2][sh], k0, k1)
        pend = np.ascontiguousarray(n[L.BUF_PENDING_FORCE].T)
        r = O.point_rollout(cfg, self.sc, self._world0(), act, pend, k0, k1)
        n[L.BUF_PENDING_FORCE][...] = pend.T
        n[L.BUF_STATES][...] = r["states"].transpose(1, 0, 2)
        n[L.BUF_ACTIONS][...] = r["actions"].transpose(1, 0, 2)
        n[L.BUF_COST_HORIZON][...] = r["cost_h"].T
        n[L.BUF_TRAJ_COST][...] = r["J"]

    def _update_mix(self):
        """One-collective sharding (m3_config.shard_mix): softmin over the LOCAL shard, f32."""
        n, T, nu, k0, k1, f = self.np, self.T, self.nu, self.k0, self.k0 + self.Kl, np.float32
        c = self.cfg
        J = n[L.BUF_TRAJ_COST]
        beta = f(c.lambda_ if c.mode_simple else self.beta)
        m_r = J.min()
        e = np.exp((f(-1.0) / beta) * (J - m_r)).astype(f)
        eta_r = e.sum(dtype=f)
        w = (e * (f(1.0) / eta_r)).astype(f)
        n[L.BUF_WEIGHTS][k0:k1] = w
        actions = n[L.BUF_ACTIONS]                                  # [T, Kl, nu]
        rec = n[L.BUF_RECORD]
        rec[...] = 0
        kk = np.arange(k0, k1)
        best = int(np.argmin(J))
        rec[0], rec[1] = m_r, eta_r
        rec[2], rec[3] = w[kk < self.Kg // 2].sum(dtype=f), w[kk >= self.Kg // 2].sum(dtype=f)
        rec[4:5] = np.array([k0 + best], np.int32).view(f)
        order = np.lexsort((kk, J))[:L.TOPK]
        rec[8:28] = J[order]
        rec[28:48] = (k0 + order).astype(np.int32).view(f)
        body = rec[self.HDR:]
        body[:T * nu] = np.einsum("k,tkj->tj", w, actions).astype(f).reshape(-1)
        body[3 * T * nu:4 * T * nu] = actions[:, best, :].reshape(-1)
        body[6 * T * nu:] = n[L.BUF_STATES][:, order, :][:, :, [0, 2]].transpose(1, 0, 2).reshape(-1)

    def _finalize_mix(self):
        n, T, nu, f = self.np, self.T, self.nu, np.float32
        c = self.cfg
        R = n[L.BUF_RECORDS_ALL]
        N = R.shape[0]
        beta = f(c.lambda_ if c.mode_simple else self.beta)
        m = R[:, 0].min()
        s = (np.exp((f(-1.0) / beta) * (R[:, 0] - m)).astype(f) * R[:, 1]).astype(f)
        rho = (s / s.sum(dtype=f)).astype(f)
        br = int(np.argmin(R[:, 0]))
        red = n[L.BUF_REDUCE]
        red[...] = 0
        B = R[:, self.HDR:]
        red[:T * nu] = (rho[:, None] * B[:, :T * nu]).sum(axis=0, dtype=f)
        red[3 * T * nu:4 * T * nu] = B[br, 3 * T * nu:4 * T * nu]
        Jc = R[:, 8:28].reshape(-1)
        Ic = R[:, 28:48].copy().view(np.int32).reshape(-1)
        order = np.lexsort((Ic, Jc))[:L.TOPK]
        n[L.BUF_TOP_IDX][...] = Ic[order]
        top = red[6 * T * nu:].reshape(L.TOPK, T, 2)
        for slot, cnd in enumerate(order):
            top[slot] = B[cnd // L.TOPK, 6 * T * nu:].reshape(L.TOPK, T, 2)[cnd % L.TOPK]
        n[L.BUF_WEIGHTS][self.k0:self.k0 + self.Kl] *= rho[self.k0 // self.Kl]

        class _I:
            pass
        i = _I()
        i.wsum_push = float((rho * R[:, 2]).sum(dtype=f))
        i.wsum_pull = float((rho * R[:, 3]).sum(dtype=f))
        i.beta = float(self.beta)
        self._info = i

    def _update_regen(self):
        """Before the all-gather: the shard's costs are the head of the record already (alias); add its
        own top-k (costs, global indices, trajectories)."""
        n, T, Kl, k0 = self.np, self.T, self.Kl, self.k0
        rec = n[L.BUF_RECORD]
        J = rec[:Kl]
        order = np.lexsort((np.arange(Kl), J))[:L.TOPK]
        rec[Kl:Kl + L.TOPK] = J[order]
        rec[Kl + L.TOPK:Kl + 2 * L.TOPK] = (k0 + order).astype(np.int32).view(np.float32)
        rec[Kl + 2 * L.TOPK:Kl + 2 * L.TOPK + L.TOPK * T * 2] = \
            n[L.BUF_STATES][:, order, :][:, :, [0, 2]].transpose(1, 0, 2).reshape(-1)

    def _finalize_regen(self):
        """After the all-gather: all K costs are here; the other shards' actions are re-generated from the
        replicated noise table + plan; then the unsharded update."""
        cfg = self._ocfg()
        n, T, nu, Kl, Kg = self.np, self.T, self.nu, self.Kl, self.Kg
        R = n[L.BUF_RECORDS_ALL]
        J = np.ascontiguousarray(R[:, :Kl].reshape(-1))
        n[L.BUF_TRAJ_COST_ALL][...] = J
        w, w1, w2, info = O.update_weights(cfg, J, self.beta)
        n[L.BUF_WEIGHTS][...], n[L.BUF_WEIGHTS_1][...], n[L.BUF_WEIGHTS_2][...] = w, w1, w2
        sh = np.minimum(np.arange(1, T + 1), T - 1)
        act = O.assemble_actions(cfg, self.delta_all, n[L.BUF_MEAN][sh], n[L.BUF_MEAN_1][sh], n[L.BUF_MEAN_2][sh],
                                 n[L.BUF_BEST_1][sh], n[L.BUF_BEST_2][sh], 0, Kg)
        if cfg.sample_null_action:
            act[Kg - 1] = 0.0                     # what the rollout stores for the null sample
        ps = O.partial_sums(cfg, w, w1, w2, act, 0, Kg)
        red = n[L.BUF_REDUCE]
        red[...] = 0
        red[:3 * T * nu] = ps.reshape(-1)
        for i, g in enumerate([info.best_idx, info.best_idx_1, Kg // 2 + info.best_idx_2]):
            red[(3 + i) * T * nu:(4 + i) * T * nu] = act[g].reshape(-1)
        order = np.lexsort((np.arange(Kg), J))[:L.TOPK]
        n[L.BUF_TOP_IDX][...] = order
        top = red[6 * T * nu:].reshape(L.TOPK, T, 2)
        for slot, g in enumerate(order):
            rec = R[g // Kl]
            ids = rec[Kl + L.TOPK:Kl + 2 * L.TOPK].copy().view(np.int32)
            q = int(np.nonzero(ids == g)[0][0])
            top[slot] = rec[Kl + 2 * L.TOPK:Kl + 2 * L.TOPK + L.TOPK * T * 2].reshape(L.TOPK, T, 2)[q]
        self._info = info

    def update_b(self):
        """shard_mix = 3 between the two exchanges: the searches on all gathered costs (the library walks the mixture
        of the shards' ladder tables: the same decisions), then the weights and the weighted action sums of THIS rank's
        own samples and its best rows into its second record."""
        assert self.p3
        cfg = self._ocfg()
        n, T, nu, Kl, Kg, k0 = self.np, self.T, self.nu, self.Kl, self.Kg, self.k0
        J = np.ascontiguousarray(n[L.BUF_RECORDS_ALL][:, :Kl].reshape(-1))
        w, w1, w2, info = O.update_weights(cfg, J, self.beta)
        n[L.BUF_WEIGHTS][...], n[L.BUF_WEIGHTS_1][...], n[L.BUF_WEIGHTS_2][...] = w, w1, w2
        actions = np.ascontiguousarray(n[L.BUF_ACTIONS].transpose(1, 0, 2))
        rb = n[L.BUF_RECORD_B]
        rb[...] = 0
        rb[8:8 + 3 * T * nu] = O.partial_sums(cfg, w, w1, w2, actions, k0, k0 + Kl).reshape(-1)
        h = Kg // 2
        sets = [(w, 0, Kg, 0), (w1, 0, h, 0), (w2, h, Kg, h)]          # (weights, first / last global index, offset)
        for i, (ws, lo, hi, off) in enumerate(sets):
            a, b = max(lo, k0), min(hi, k0 + Kl)
            rb[2 * i], rb[2 * i + 1] = np.inf, -1.0
            if a < b:
                loc = ws[a - off:b - off]
                g = a + int(np.argmax(loc))                              # first maximum, as the unsharded argmax
                rb[2 * i], rb[2 * i + 1] = -loc[g - a], float(g)
                rb[8 + (3 + i) * T * nu:8 + (4 + i) * T * nu] = actions[g - k0].reshape(-1)
        rb[6], rb[7] = w[max(0, k0):min(h, k0 + Kl)].sum(dtype=np.float32) if k0 < h else 0.0, \
            w[max(h, k0):k0 + Kl].sum(dtype=np.float32) if k0 + Kl > h else 0.0
        self._info = info

    def _finalize_p3(self):
        n, T, nu, Kl, Kg = self.np, self.T, self.nu, self.Kl, self.Kg
        RB, R = n[L.BUF_RECORDS_B_ALL], n[L.BUF_RECORDS_ALL]
        red = n[L.BUF_REDUCE]
        red[...] = 0
        acc = np.zeros(3 * T * nu, np.float32)
        for r in range(RB.shape[0]):                                     # rank order
            acc = (acc + RB[r, 8:8 + 3 * T * nu]).astype(np.float32)
        red[:3 * T * nu] = acc
        for i in range(3):
            keys = [(RB[r, 2 * i], int(RB[r, 2 * i + 1]), r) for r in range(RB.shape[0]) if RB[r, 2 * i + 1] >= 0]
            win = min(keys)[2]
            red[(3 + i) * T * nu:(4 + i) * T * nu] = RB[win, 8 + (3 + i) * T * nu:8 + (4 + i) * T * nu]
        J = np.ascontiguousarray(R[:, :Kl].reshape(-1))
        order = np.lexsort((np.arange(Kg), J))[:L.TOPK]
        n[L.BUF_TOP_IDX][...] = order
        top = red[6 * T * nu:].reshape(L.TOPK, T, 2)
        for slot, g in enumerate(order):
            rec = R[g // Kl]
            ids = rec[Kl + L.TOPK:Kl + 2 * L.TOPK].copy().view(np.int32)
            q = int(np.nonzero(ids == g)[0][0])
            top[slot] = rec[Kl + 2 * L.TOPK:Kl + 2 * L.TOPK + L.TOPK * T * 2].reshape(L.TOPK, T, 2)[q]

    def update(self):
        if self.regen:
            return self._update_regen()
        if self.mix:
            return self._update_mix()
        cfg = self._ocfg()
        n = self.np
        if self.Kl == self.Kg:
            n[L.BUF_TRAJ_COST_ALL][...] = n[L.BUF_TRAJ_COST]
        J = n[L.BUF_TRAJ_COST_ALL]
        w, w1, w2, info = O.update_weights(cfg, J, self.beta)
        self.beta = info.beta
        n[L.BUF_WEIGHTS][...], n[L.BUF_WEIGHTS_1][...], n[L.BUF_WEIGHTS_2][...] = w, w1, w2
        k0, k1, T, nu = self.k0, self.k0 + self.Kl, self.T, self.nu
        actions = np.ascontiguousarray(n[L.BUF_ACTIONS].transpose(1, 0, 2))
        ps = O.partial_sums(cfg, w, w1, w2, actions, k0, k1)
        red = n[L.BUF_REDUCE]
        red[...] = 0
        red[:3 * T * nu] = ps.reshape(-1)
        best = [info.best_idx, info.best_idx_1, self.Kg // 2 + info.best_idx_2] if cfg.multi_modal \
            else [info.best_idx, -1, -1]
        for i, g in enumerate(best):
            if k0 <= g < k1:
                red[(3 + i) * T * nu:(4 + i) * T * nu] = actions[g - k0].reshape(-1)
        order = np.lexsort((np.arange(self.Kg), J))[:L.TOPK]
        n[L.BUF_TOP_IDX][...] = order
        top = red[6 * T * nu:].reshape(L.TOPK, T, 2)
        for r_, g in enumerate(order):
            if k0 <= g < k1:
                top[r_] = n[L.BUF_STATES][:, g - k0, :][:, [0, 2]]
        self._info = info

    def finalize(self):
        if self.p3:
            self._finalize_p3()
        elif self.regen:
            self._finalize_regen()
        if self.mix:
            self._finalize_mix()
        cfg = self._ocfg()
        n, T, nu = self.np, self.T, self.nu
        red = n[L.BUF_REDUCE]
        sh = np.minimum(np.arange(1, T + 1), T - 1)
        ps = red[:3 * T * nu].reshape(3, T, nu)
        n[L.BUF_MEAN][...] = O.mean_update(cfg, n[L.BUF_MEAN][sh], ps[0])
        if cfg.multi_modal:
            n[L.BUF_MEAN_1][...], n[L.BUF_MEAN_2][...] = ps[1], ps[2]
            n[L.BUF_BEST_1][...] = red[4 * T * nu:5 * T * nu].reshape(T, nu)
            n[L.BUF_BEST_2][...] = red[5 * T * nu:6 * T * nu].reshape(T, nu)
        else:
            n[L.BUF_BEST][...] = red[3 * T * nu:4 * T * nu].reshape(T, nu)
        a = n[L.BUF_MEAN].copy()
        n[L.BUF_ACTION_OUT][...] = O.savgol9(a) if cfg.filter_u else a
        if getattr(self, "_action_out", None) is not None:
            self._action_out.copy_(self.t[L.BUF_ACTION_OUT])
        n[L.BUF_TOP_TRAJS][...] = red[6 * T * nu:].reshape(L.TOPK, T, 2)
        if self.cov_active:   # mppi.py:508-516 (as oracle.OraclePointPlanner)
            assert self.Kl == self.Kg
            f = np.float32
            d = (n[L.BUF_ACTIONS] - n[L.BUF_MEAN][:, None, :]).astype(np.float64)      # [T, K, nu]
            upd = (n[L.BUF_WEIGHTS].astype(np.float64)[None, :, None] * d * d).sum(axis=1).mean(axis=0).astype(f)
            cov = (f(1.0 - 0.7) * n[L.BUF_COV][0] + f(0.7) * upd).astype(f)
            n[L.BUF_COV][0] = (cov + f(0.005)).astype(f)
            n[L.BUF_COV][1] = np.sqrt(n[L.BUF_COV][0])
        self.calls += 1

    def set_action_out(self, tensor):
        self._action_out = tensor

    def command(self, sync_host=False):
        self.rollout()
        self.update()
        self.finalize()
        return self.t[L.BUF_ACTION_OUT]

    def info(self):
        i = L.Info()
        i.calls = self.calls
        s = self._info
        if s is not None:
            i.wsum_push, i.wsum_pull, i.beta = s.wsum_push, s.wsum_pull, s.beta
            i.pull_preference = int(s.wsum_pull > s.wsum_push)
        return i

    @property
    def states(self):
        return self.t[L.BUF_STATES].permute(1, 0, 2)

    @property
    def actions(self):
        return self.t[L.BUF_ACTIONS].permute(1, 0, 2)

    def close(self):
        pass
