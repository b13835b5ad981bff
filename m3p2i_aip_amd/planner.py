"""MPPI / M3P2I with the reference's constructor and command() API on the HIP engine.

Mirror of
  src/m3p2i_aip/planners/motion_planner/mppi.py    MPPIConfig :9-59, MPPI :61-517
  src/m3p2i_aip/planners/motion_planner/m3p2i.py   M3P2I :5-92

    M3P2I(cfg, dynamics=callable, running_cost=callable).command(state) -> Tensor[T, nu]

Two execution modes behind the same API:

* FUSED   one C call: rollout+cost kernel, weight kernel, weighted-sum kernel, finalize
          kernel.  The Python callbacks are never invoked.  Used when the callbacks are the
          standard plugin of scripts/reactive_tamp.py:63-73 (step the wrapper, return
          Objective.compute_cost) -- which the planner CHECKS, not assumes: on the first
          command it runs both modes from the same state and keeps the fused path only if
          the trajectory costs agree (``cfg.mppi.fused`` = True/False overrides the probe).
* STEP    the reference's per-t loop (mppi.py:296-315) through the user's callables, with
          the wrapper's step()/getters backed by single-step kernels; the importance-weight
          update still runs in the HIP update kernels.

Quirks of the reference that change numbers are kept (bug-compatible) and listed in
DESIGN.md: Q1 cost_total aliasing, Q2 lambda_ unused in halton mode, Q3 multi-modal beta
restarts at 1, Q4 double noise scaling for halton-spline+random, Q5 suction acts one step
late, Q6 null action on the zero-noise sample, Q7 push_pull evaluates both costs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np
import torch

from . import _lib as L
from . import sampling, scenes
from .engine import HipEngine, make_config

# The engine class the planner instantiates.  The product always uses HipEngine (HIP kernels,
# no fallback); the world_size-2 gloo tests substitute an oracle-backed stand-in to exercise
# the host-side sharding logic on a machine without a GPU (tests/oracle_engine.py).
ENGINE_CLS = HipEngine
SHARD_MIX3_FROM = 524288     # K_global from which the default multi-modal sharding protocol is shard_mix = 3


@dataclass
class MPPIConfig(object):
    """Same fields and defaults as the reference's MPPIConfig (mppi.py:9-59); the last three
    are additions of this build (None = automatic)."""
    num_samples: int = 200
    horizon: int = 12
    nx: int = 4
    mppi_mode: str = 'halton-spline'
    sampling_method: str = "halton"
    noise_sigma: Optional[List[List[float]]] = None
    noise_mu: Optional[List[float]] = None
    device: str = "cuda:0"
    lambda_: float = 1.0
    update_lambda: bool = False
    update_cov: bool = False
    u_min: Optional[List[float]] = None
    u_max: Optional[List[float]] = None
    u_init: float = 0.0
    U_init: Optional[List[List[float]]] = None
    u_scale: float = 1
    u_per_command: int = 1
    rollout_var_discount: float = 0.95
    sample_null_action: bool = False
    sample_previous_plan: bool = True
    sample_other_priors: bool = False
    noise_abs_cost: bool = False
    filter_u: bool = False
    use_priors: bool = False
    seed_val: int = 0
    eta_u_bound: int = 10
    eta_l_bound: int = 5
    # --- additions ---
    fused: Optional[bool] = None      # None: probe the callbacks on the first command
    rank: int = 0                     # sample sharding: this process owns
    world_size: int = 1               #   num_samples/world_size consecutive samples
    shard_mix: Optional[int] = None   # None: one-collective protocol whenever it applies (multi-modal: 2); 1 / 2 / False
    relabel_samples: bool = True      # generated noise rows into wavefront-coherent order (same sample set)
    action_ring: int = 0              # 0: command() returns a fresh tensor each call (mppi.py:238-246 does); n > 0:
                                      # slot (call % n) of a ring the planner owns (no allocator call per command)
    device_knots: bool = False        # Halton + erfinv knots on the device too (~1e-6 from the host sampler's)
    halton_scramble: str = "none"     # "none": plain Halton (mppi_utils.py:81-87, pinned by golden G8); "faure": the
                                      # generalized Halton structure of the ghalton branch the reference's planner
                                      # takes (mppi.py:465-471, mppi_utils.py:89-95) with Faure's (1992) permutations


def _get(cfg, name, default=None):
    try:
        return getattr(cfg, name)
    except Exception:
        return default


class MPPI():
    def __init__(self, cfg, dynamics: Callable = None, running_cost: Callable = None):
        self.env_type = cfg.env_type
        self.multi_modal = bool(cfg.multi_modal)
        self._top_cfg = cfg
        m = cfg.mppi
        self.mppi_mode = m.mppi_mode
        self.sampling_method = m.sampling_method
        if self.mppi_mode not in ("simple", "halton-spline"):
            raise ValueError(f"unknown mppi_mode {self.mppi_mode!r}")
        if self.sampling_method not in ("halton", "random"):
            raise ValueError(f"unknown sampling_method {self.sampling_method!r}")
        # U_init / u_init are accepted and ignored: the reference stores both and never reads them
        # (mppi.py:122-123, :132-133 -- the one use is commented out; tests/golden: g9_opt_dead).  A warm start
        # is installed with planner.mean_action = ... (planner.U in simple mode), as with the reference.
        self.noise_abs_cost = bool(_get(m, "noise_abs_cost", False))     # mppi.py:116, :366-367 (simple mode)
        self.update_cov = bool(_get(m, "update_cov", False))             # mppi.py:201, :508-516 ("!! weird")
        self.step_size_cov, self.kappa = 0.7, 0.005                      # mppi.py:202-203

        self.K = int(m.num_samples)
        self.half_K = int(self.K / 2)
        self.T = int(m.horizon)
        self.filter_u = bool(m.filter_u)
        self.lambda_ = float(m.lambda_)
        self.sample_null_action = bool(m.sample_null_action)
        self.u_per_command = int(m.u_per_command)
        self.device = m.device
        self.tensor_args = {'device': m.device, 'dtype': torch.float32}
        self.nx = int(m.nx)

        noise_sigma = m.noise_sigma
        if not noise_sigma:
            noise_sigma = np.identity(int(m.nx / 2)).tolist()
        noise_sigma = [list(map(float, r)) for r in noise_sigma]
        self.nu = len(noise_sigma)
        sig = torch.tensor(noise_sigma, dtype=torch.float32)
        # a non-diagonal noise_sigma: the halton-spline path scales the Halton noise with sqrt(diag(noise_sigma))
        # and never reads the off-diagonal entries (mppi.py:175-176, :394); MultivariateNormal sampling
        # (sampling_method='random', mppi_mode='simple': mppi.py:129-131, :340, :481) and the action cost
        # (mppi.py:128, :366-372) use the whole matrix -- the library takes it as noise_sigma_full
        noise_mu = _get(m, "noise_mu", None)
        noise_mu = [0.0] * self.nu if not noise_mu else list(map(float, noise_mu))     # mppi.py:120-121
        u_max, u_min = m.u_max, m.u_min
        if u_max and not u_min:
            u_min = [-float(x) for x in u_max]
        if u_min and not u_max:
            u_max = [-float(x) for x in u_min]
        if not u_max:
            raise ValueError("u_min/u_max are required (mppi.py:135-136)")
        self.noise_sigma = sig.to(m.device)
        self.noise_mu = torch.tensor(noise_mu, dtype=torch.float32, device=m.device)
        self.noise_sigma_inv = torch.inverse(sig).to(m.device)
        self.u_max = torch.tensor(list(map(float, u_max)), device=m.device)
        self.u_min = torch.tensor(list(map(float, u_min)), device=m.device)
        self.u_scale = float(m.u_scale)
        self._cov0 = torch.diagonal(self.noise_sigma, 0)
        self.knot_scale, self.degree, self.seed_val = 4, 2, int(_get(m, "seed_val", 0) or 0)
        self.n_knots = self.T // self.knot_scale
        self.step_size_mean = 0.98
        self.gamma = float(m.rollout_var_discount)
        self.gamma_seq = torch.cumprod(torch.tensor([1.0] + [self.gamma] * (self.T - 1)), dim=0
                                       ).reshape(1, self.T).to(**self.tensor_args)
        self.sgf_window, self.sgf_order = 9, 2

        self.F = dynamics
        self.running_cost = running_cost
        self.terminal_state_cost = None
        self.state = None
        self._have_noise = False
        self.ee_states = 'None'

        rank, world = int(_get(m, "rank", 0) or 0), int(_get(m, "world_size", 1) or 1)
        if self.K % world:
            raise ValueError("num_samples must be divisible by world_size")
        self.rank, self.world_size = rank, world
        self.transport, self.ranks_seen = None, 1        # set when a transport is attached (distributed.py)
        self.K_local = self.K // world
        self.k_offset = rank * self.K_local
        dev = torch.device(m.device)
        isaac = _get(cfg, "isaacgym", None)
        # sharded protocol, ONE collective per command (an all-gather of per-rank records) whenever it
        # applies: single-mode MPPI (beta fixed during a command) mixes per-rank softmins; the multi-modal
        # beta search gets every rank's costs in the records and re-generates the other ranks' actions
        # from the replicated noise table (needs an explicit table: not sampling_method='random').
        # shard_mix=False forces all-gather of J + all-reduce of the packed sums.
        sm = _get(m, "shard_mix", None)
        single = not (self.multi_modal and self.mppi_mode != "simple")
        can_mix = single or self.sampling_method != "random"
        if sm and not can_mix:
            raise ValueError("shard_mix=True with multi_modal needs sampling_method='halton' (a noise table)")
        self.shard_mix = bool(world > 1 and can_mix and (True if sm is None else sm))
        # multi-modal: shard_mix=2 adds per-shard ladder tables to the records -- half the per-rank work after the
        # collective, equal to the unsharded run up to f32 rounding; the INTEGER 1 keeps the bit-identical variant (all K
        # costs re-evaluated on every rank); None / True (the bool: "use the one-collective family") = chosen by size, below
        # shard_mix=3: two small exchanges and O(K_local) work per rank after the first (more ranks / samples than the
        # one-collective protocols are meant for: their post-gather work grows with K_global)
        # The default picks by size: `2` (one collective, O(K_global) work per rank after it) up to SHARD_MIX3_FROM
        # samples, `3` (two small exchanges, O(K_local) work) from there on -- the crossover of the emulated
        # rank-0-of-8 measurements (profiles/r03/protocols_rank0_of_8_K*.json: K_global 64 000 / 256 000: 2 is faster,
        # 0.181 vs 0.205 and 0.144 vs 0.230 ms; 1 M: 3 is, 0.402 vs 0.441) -- when 3 applies at all (T * nu <= 2048, at
        # most 256 workgroups of the shard's weights pass: K_local <= 262 144).
        if not self.shard_mix:
            level = 0
        elif single:
            level = 1
        elif sm is None or sm is True:
            fits3 = self.T * self.nu <= 2048 and self.K_local <= 262144
            level = 3 if (self.K >= SHARD_MIX3_FROM and fits3) else 2
        else:
            level = int(sm) if int(sm) in (1, 2, 3) else 1
        self._shard_mix_level = level
        self.protocol = {0: "gather+reduce (two collectives)", 1: "one collective (records; weights of all samples re-evaluated)",
                         2: "one collective (records with ladder tables)", 3: "two small exchanges"}[level] if world > 1 else "unsharded"
        self.relabel_samples = bool(_get(m, "relabel_samples", True))
        self.action_ring = int(_get(m, "action_ring", 0) or 0)
        self._engine = ENGINE_CLS(make_config(
            K=self.K, K_local=self.K_local, k_offset=self.k_offset, T=self.T, nu=self.nu,
            env_type=self.env_type, multi_modal=self.multi_modal,
            mode_simple=self.mppi_mode == "simple",
            sampling_random=self.sampling_method == "random" or self.mppi_mode == "simple",
            sample_null_action=self.sample_null_action, filter_u=self.filter_u,
            u_per_command=self.u_per_command, u_min=list(map(float, u_min)),
            u_max=list(map(float, u_max)), noise_sigma_diag=[noise_sigma[j][j] for j in range(self.nu)],
            u_scale=self.u_scale, gamma=self.gamma, lambda_=self.lambda_,
            kp_suction=float(_get(cfg, "kp_suction", 0) or 0),
            pre_height_diff=float(_get(cfg, "pre_height_diff", 0) or 0),
            dt=float(_get(isaac, "dt", 0.05 if self.env_type == "point_env" else 0.01)),
            substeps=int(_get(isaac, "substeps", 2)), seed=self.seed_val, device=dev.index or 0,
            cube_on_shelf=bool(_get(cfg, "cube_on_shelf", False)), shard_mix=self._shard_mix_level,
            noise_mu=noise_mu, noise_sigma=noise_sigma, noise_abs_cost=self.noise_abs_cost,
            update_cov=self.update_cov))
        if self.mppi_mode == "simple":
            # mppi.py:129-134: U starts as T draws of N(noise_mu, noise_sigma) from torch's global generator
            dist = torch.distributions.MultivariateNormal(self.noise_mu.cpu(), covariance_matrix=sig)
            self.U = dist.sample((self.T,)).to(**self.tensor_args)
        self._fused = _get(m, "fused", None)
        self._sim = None
        self._objective = None
        self._discover_plugin()
        self.gripper_command = None
        self.collective = None  # set by m3p2i_aip_amd.distributed for world_size > 1

    # ------------------------------------------------------------------ plugin discovery
    def _discover_plugin(self):
        """Find the wrapper + Objective behind the callbacks (reactive_tamp.py keeps them as
        attributes of the object whose bound methods it passes in)."""
        from .cost_functions import Objective, live_objectives
        from .isaacgym_wrapper import IsaacGymWrapper, live_wrappers
        owners = [getattr(f, "__self__", None) for f in (self.F, self.running_cost)]
        for o in owners:
            if o is None:
                continue
            for v in list(vars(o).values()) if hasattr(o, "__dict__") else []:
                if isinstance(v, IsaacGymWrapper) and self._sim is None:
                    self._sim = v
                if isinstance(v, Objective) and self._objective is None:
                    self._objective = v
        if self._sim is None:
            c = [w for w in live_wrappers() if w.num_envs == self.K_local and w.env_type == self.env_type]
            if len(c) == 1:
                self._sim = c[0]
        if self._objective is None:
            c = [o for o in live_objectives() if o.num_samples == self.K]
            if len(c) == 1:
                self._objective = c[0]

    def attach(self, sim=None, objective=None):
        """Explicitly name the simulator wrapper / Objective used by the fused path."""
        if sim is not None:
            self._sim = sim
        if objective is not None:
            self._objective = objective
        return self

    # ------------------------------------------------------------------ reference attributes
    def _buf(self, which):
        return self._engine.buffer(which)

    mean_action = property(lambda s: s._buf(L.BUF_MEAN), lambda s, v: s._buf(L.BUF_MEAN).copy_(v))
    mean_action_1 = property(lambda s: s._buf(L.BUF_MEAN_1), lambda s, v: s._buf(L.BUF_MEAN_1).copy_(v))
    mean_action_2 = property(lambda s: s._buf(L.BUF_MEAN_2), lambda s, v: s._buf(L.BUF_MEAN_2).copy_(v))
    best_traj = property(lambda s: s._buf(L.BUF_BEST), lambda s, v: s._buf(L.BUF_BEST).copy_(v))
    best_traj_1 = property(lambda s: s._buf(L.BUF_BEST_1), lambda s, v: s._buf(L.BUF_BEST_1).copy_(v))
    best_traj_2 = property(lambda s: s._buf(L.BUF_BEST_2), lambda s, v: s._buf(L.BUF_BEST_2).copy_(v))
    U = property(lambda s: s._buf(L.BUF_MEAN), lambda s, v: s._buf(L.BUF_MEAN).copy_(v))
    # [K] softmin weights.  Sharded single-mode planners (shard_mix) materialise their OWN shard's entries
    # plus the global top-20's (top_values); the multi-modal protocols fill all K on every rank.
    weights = property(lambda s: s._buf(L.BUF_WEIGHTS))
    weights_1 = property(lambda s: s._buf(L.BUF_WEIGHTS_1))
    weights_2 = property(lambda s: s._buf(L.BUF_WEIGHTS_2))
    states = property(lambda s: s._engine.states)        # [K_local, T, 4] strided view
    @property
    def actions(self):
        """[K_local, T, nu] controls of the last rollout as the reference leaves them in `self.actions`: the stack
        handed to the dynamics divided by u_scale (mppi.py:353, :420; the library keeps the scaled stack, which is
        what the distribution update consumes, mppi.py:313-331)."""
        a = self._engine.actions
        return a if self.u_scale == 1.0 else a / self.u_scale

    top_trajs = property(lambda s: s._buf(L.BUF_TOP_TRAJS))
    top_idx = property(lambda s: s._buf(L.BUF_TOP_IDX).to(torch.int64))
    top_values = property(lambda s: s.weights[s.top_idx])
    cost_total = property(lambda s: s._buf(L.BUF_TRAJ_COST))

    # cov_action / scale_tril (mppi.py:175-176): constants unless update_cov rewrites them after every command of
    # a single-mode halton-spline planner (mppi.py:508-516; the library keeps both in M3_BUF_COV)
    @property
    def cov_action(self):
        return self._buf(L.BUF_COV)[0] if self.update_cov else self._cov0

    @property
    def scale_tril(self):
        return self._buf(L.BUF_COV)[1] if self.update_cov else torch.sqrt(self._cov0)

    @property
    def beta(self):
        return self._engine.info().beta

    @property
    def total_costs(self):
        J = self._buf(L.BUF_TRAJ_COST_ALL)
        return J - J.min()

    # ------------------------------------------------------------------ command
    def _dynamics(self, state, u, t=None):
        return self.F(state, u, t=None)

    def _running_cost(self, state):
        return self.running_cost(state)

    def _ensure_noise(self):
        if self.mppi_mode == "simple" or self.sampling_method == "random":
            return
        if self._have_noise:
            return
        e = self._engine
        # (a one-collective multi-modal shard holds the rows of ALL samples: engine.needs_global_noise)
        k0, k1 = (0, self.K) if e.needs_global_noise else (self.k_offset, self.k_offset + self.K_local)
        scramble = str(_get(self._top_cfg.mppi, "halton_scramble", "none") or "none")
        if scramble not in sampling.SCRAMBLES:
            raise ValueError(f"unknown halton_scramble {scramble!r} (one of {sampling.SCRAMBLES})")
        # device sampler: Halton knots on the host (K*nu*n_knots values, vectorised), the K*nu
        # spline fits -- where the reference's ~1 s init goes -- one GPU thread each
        if bool(_get(self._top_cfg.mppi, "device_knots", False)):
            if self.n_knots <= self.degree:
                raise ValueError(f"horizon T={self.T} gives n_knots={self.n_knots}: the spline needs T >= {self.knot_scale * (self.degree + 1)}")
            e.set_noise_halton(self.n_knots, self.degree, 0.5, scramble)     # ... and the knots on the device as well
        else:
            e.set_noise_knots(sampling.halton_knots(self.K, self.T, self.nu, self.knot_scale, self.degree, k0, k1,
                                                    scramble=scramble), self.degree, 0.5)
        if self.relabel_samples:
            e.relabel_samples()   # the sampler's row labels are arbitrary: wavefront-coherent ones
        self._have_noise = True

    @property
    def delta(self):
        """[K_local, T, nu] noise (mppi.py:466-483), a strided view of the library's time-major buffer."""
        return self._engine.buffer(L.BUF_NOISE).permute(1, 0, 2) if self._have_noise else None

    def _push_objective(self):
        o = self._objective
        if o is None or o.task is None or o.goal is None:
            raise RuntimeError("fused command() needs an Objective with update_objective(task, goal) done; "
                               "use planner.attach(sim, objective)")
        grip = {"open": 1, "close": 2}.get(self.gripper_command, 0)
        self._engine.set_objective(o.task, o.goal_list(), grip)
        avoid = bool(getattr(o, "avoid_dyn_obs", False))
        if avoid != getattr(self, "_avoid_dyn_obs", False):     # (extension, off by default: cost_functions.Objective)
            self._engine.set_avoid_dyn_obs(avoid)
            self._avoid_dyn_obs = avoid

    def _bind_world(self):
        s = self._sim
        if s is None:
            raise RuntimeError("fused command() needs the HIP IsaacGymWrapper; use planner.attach(sim, objective)")
        if getattr(self, "_bound_sim", None) is not s:
            if self.env_type == "point_env":
                self._engine.bind_sim_point(s._dof_state, s._root_state,
                                            scenes.actor_index(self.env_type, "box"),
                                            scenes.actor_index(self.env_type, "dyn-obs"))
            else:
                self._engine.bind_sim_panda(s._dof_state, s._root_state,
                                            scenes.actor_index(self.env_type, "cubeA"),
                                            scenes.actor_index(self.env_type, "cubeB"),
                                            scenes.actor_index(self.env_type, "dyn-obs"))
            self._bound_sim = s

    def _exchange(self, phase):
        if self.collective is None:
            return
        times = getattr(self, "collective_times", None)
        if times is None:
            self.collective(self, phase)
            return
        # bench.py's `collective_ms`: events on the stream the collective is enqueued on
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.collective(self, phase)
        e1.record()
        times.append((phase, e0, e1))

    def _next_action_slot(self):
        """The tensor this command's plan is written into by the finalize kernel itself (`m3_set_action_out`): a fresh
        one per call, as the reference returns (no copy: torch's caching allocator hands out the block, stream-ordered
        on the stream the kernels run on, ~2 us of host time hidden behind the rollout) -- or, with
        `MPPIConfig.action_ring = n`, slot (call % n) of a ring."""
        rows = self.u_per_command if self.mppi_mode == "simple" else self.T
        if not self.action_ring:
            out = torch.empty(rows, self.nu, **self.tensor_args)
        else:
            if getattr(self, "_action_ring", None) is None:
                self._action_ring = torch.zeros(self.action_ring, rows, self.nu, **self.tensor_args)
                self._action_slot = 0
            out = self._action_ring[self._action_slot]
            self._action_slot = (self._action_slot + 1) % self.action_ring
        self._engine.set_action_out(out)
        return out

    def _command_fused(self):
        self._push_objective()
        self._bind_world()
        e = self._engine
        out = self._next_action_slot()
        if self.world_size == 1 and self.collective is None:
            e.command()
        else:
            e.rollout()
            self._update_exchange_finalize()
        return out

    def _update_exchange_finalize(self):
        e = self._engine
        if self.world_size > 1 and self.collective is None:
            raise RuntimeError("planner built with world_size > 1 but no collectives installed: call "
                               "m3p2i_aip_amd.distributed.attach_collectives(planner) first (without the "
                               "exchange the update would run on zero / stale remote slices)")
        if self.shard_mix and self._shard_mix_level == 3:
            e.update()                      # the shard's costs, top-k, minima and ladder table -> its record
            self._exchange("records")
            e.update_b()                    # searches on the mixed tables; weights + sums of the OWN samples
            self._exchange("records_b")     # ~6 T nu floats per rank
            e.finalize()
        elif self.shard_mix:
            e.update()                      # softmin over the local shard -> this rank's record
            self._exchange("records")       # the one collective
            e.finalize()                    # mix the ranks' records, then the usual finalize
        elif self.world_size == 1 and self.collective is None:
            e.update_finalize()             # unsharded: one or two launches
        else:
            self._exchange("gather")
            e.update()
            self._exchange("reduce")
            e.finalize()

    def _assemble_torch(self):
        """mppi.py:381-416 / :335-347 in torch ops (STEP mode only; the fused kernel does this
        in registers).  Same f32 operations in the same order => identical bits."""
        T, nu, Kl, k0 = self.T, self.nu, self.K_local, self.k_offset
        e = self._engine
        ks = torch.arange(k0, k0 + Kl, device=self.device)
        if self.mppi_mode == "simple" or self.sampling_method == "random":
            # MultivariateNormal(noise_mu, noise_sigma).sample((K, T)) (mppi.py:340 / :481): this command's draws
            # of the stream the fused kernel generates in registers
            delta = e.sample_noise().permute(1, 0, 2).clone()
        else:
            delta = self.delta.clone()
        if self.mppi_mode == "simple":      # mppi.py:341-350
            U = torch.roll(self.U, -1, dims=0)                                   # :221
            act = torch.max(torch.min(U.unsqueeze(0) + delta, self.u_max), self.u_min)
            if self.env_type == "panda_env" and self.gripper_command in ("open", "close"):
                act[:, :, 7:] = 1.5 if self.gripper_command == "open" else -1.5
            return act
        delta[ks == self.K - 1] = 0.0
        shift = torch.clamp(torch.arange(1, T + 1, device=self.device), max=T - 1)
        scaled = delta * self.scale_tril.view(1, 1, nu)
        if self.multi_modal:
            m1, m2 = self.mean_action_1[shift], self.mean_action_2[shift]
            first = (ks < self.half_K).view(Kl, 1, 1)
            act = torch.where(first, m1.unsqueeze(0) + scaled, m2.unsqueeze(0) + scaled)
        else:
            act = self.mean_action[shift].unsqueeze(0) + scaled
        act = torch.max(torch.min(act, self.u_max), self.u_min)
        if self.multi_modal:
            act[ks == 0] = self.best_traj_1[shift]
            act[ks == self.half_K] = self.best_traj_2[shift]
        if self.env_type == "panda_env":
            if self.gripper_command == "open":
                act[:, :, 7:] = 1.5
            elif self.gripper_command == "close":
                act[:, :, 7:] = -1.5
        return act

    def _command_step(self):
        """mppi.py:275-332 through the user's callables."""
        e = self._engine
        T, Kl = self.T, self.K_local
        act = self._assemble_torch()
        self.perturbed_action = act
        A, S, C = self._buf(L.BUF_ACTIONS), self._buf(L.BUF_STATES), self._buf(L.BUF_COST_HORIZON)
        state = self.state.view(1, -1).repeat(Kl, 1) if self.state.shape != (Kl, self.nx) else self.state
        J = torch.zeros(Kl, **self.tensor_args)
        Ssum = torch.zeros(Kl, **self.tensor_args)
        gs = torch.ones((), **self.tensor_args)
        last = self.K - 1 - self.k_offset
        for t in range(T):
            u = self.u_scale * act[:, t]
            if self.sample_null_action and 0 <= last < Kl:
                u[last] = 0.0
            state, u = self._dynamics(state, u, t)
            c = self._running_cost(state)
            A[t].copy_(u)        # mppi.py:313: the scaled controls, as the update consumes them
            S[t].copy_(state[:, :4])
            C[t].copy_(c)
            J = J + gs * c
            Ssum = Ssum + c                             # mppi.py:309
            gs = gs * self.gamma
        if self.mppi_mode == "simple":
            # mppi.py:355-362: noise = perturbed_action (the scaled controls, :311) - U; cost_total += sum U * action_cost
            # (cost_total also carries + mean(S) through the aliasing of :284, a shift the softmin cancels)
            U = torch.roll(self.U, -1, dims=0)
            noise = A.permute(1, 0, 2) - U.unsqueeze(0)
            if self.noise_abs_cost:
                noise = noise.abs()                                              # :366-367
            J = Ssum + torch.sum(U.unsqueeze(0) * ((self.lambda_ * noise) @ self.noise_sigma_inv), dim=(1, 2))
        self._buf(L.BUF_TRAJ_COST).copy_(J)
        out = self._next_action_slot()
        self._update_exchange_finalize()
        return out

    def command(self, state):
        if not torch.is_tensor(state):
            state = torch.tensor(state)
        self.state = state.to(**self.tensor_args)
        self._engine.use_torch_stream()
        self._ensure_noise()
        if self._fused is None:
            return self._probe_and_command()
        return self._command_fused() if self._fused else self._command_step()

    def _probe_and_command(self):
        """First call with fused='auto': run the fused path, rewind, run the step path through
        the user's callbacks, and keep the fused path only if both produced the same costs."""
        can_fuse = (self._sim is not None and self._objective is not None and self.F is not None
                    and self.running_cost is not None and self._objective.task is not None)
        can_step = self.F is not None and self.running_cost is not None
        if can_fuse and not can_step:
            self._fused = True
            return self._command_fused()
        if not can_fuse:
            self._fused = False
            return self._command_step()
        # both legs must start from the same warm start (means, best trajectories, pending suction
        # forces, adapted beta): snapshot it, run the fused leg, put it back, run the step leg
        keep = [L.BUF_MEAN, L.BUF_MEAN_1, L.BUF_MEAN_2, L.BUF_BEST, L.BUF_BEST_1, L.BUF_BEST_2,
                L.BUF_PENDING_FORCE, L.BUF_COV]
        saved = [self._buf(b).clone() for b in keep]
        info0 = self._engine.info()
        beta0, calls0 = info0.beta, info0.calls
        self._command_fused()
        Jf = self._buf(L.BUF_TRAJ_COST).clone()
        for b, v in zip(keep, saved):
            self._buf(b).copy_(v)
        self._engine.set_beta(beta0)
        self._engine.set_call_count(calls0)    # (the in-kernel noise stream is keyed by the command index)
        out = self._command_step()
        Js = self._buf(L.BUF_TRAJ_COST)
        same = bool(torch.allclose(Jf, Js, rtol=1e-5, atol=1e-4))
        self._fused = same
        self.probe_result = dict(fused=same, max_abs_diff=float((Jf - Js).abs().max()))
        if same and self.env_type == "point_env":
            # carry the suction force staged by the last cost evaluation into the fused state
            simw = self._sim._engine.buffer(L.BUF_SIM_WORLD)
            self._buf(L.BUF_PENDING_FORCE).copy_(simw[18:22])
        return out

    def set_noise(self, delta):
        """Explicit noise [K_local, T, nu] (e.g. a recorded sample set); [K, T, nu] -- all samples -- for a
        sharded multi-modal planner with shard_mix (engine.needs_global_noise)."""
        self._engine.set_noise(torch.as_tensor(delta, dtype=torch.float32).to(self.device))
        self._have_noise = True

    def _shift_action(self, action_seq):
        saved = action_seq[-1].clone()
        action_seq = torch.roll(action_seq, -1, dims=0)
        action_seq[-1] = saved
        return action_seq


class M3P2I(MPPI):
    def __init__(self, cfg, dynamics=None, running_cost=None):
        super().__init__(cfg, dynamics, running_cost)
        self.suction_active = _get(cfg, "suction_active", False)

    def update_gripper_command(self, task):
        if task in ["reach", "place"]:
            self.gripper_command = "open"
        elif task == "pick":
            self.gripper_command = "close"

    def pull_preference_tensor(self):
        """The pull preference of the last command as a 1-element int32 DEVICE tensor (a view of the
        library's m3_info), for consumers on the same GPU: no host synchronisation.  1 = pull."""
        if not self.multi_modal:
            return torch.full((1,), int(bool(self.suction_active)), dtype=torch.int32, device=self.device)
        w = L.INFO_PULL_PREFERENCE
        return self._engine.buffer(L.BUF_INFO)[w:w + 1]

    def get_pull_preference(self):
        if self.multi_modal:
            i = self._engine.info()   # one host sync, like the reference's two .item() calls
            return int(i.wsum_pull > i.wsum_push)
        return self.suction_active
