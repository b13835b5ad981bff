"""HipEngine: thin object over one ``m3_handle`` of libm3p2i_hip.so.

PyTorch is plumbing here: it provides the HIP stream and wraps the library-owned device
buffers as zero-copy tensors (``__cuda_array_interface__``).  All arithmetic of the hot
path happens inside the HIP kernels.
"""
from __future__ import annotations

import os

import ctypes as C

import numpy as np
import torch

from . import _lib as L


class _DevArray:
    """Minimal __cuda_array_interface__ carrier for a library-owned device pointer."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}
        self._owner = owner  # keeps the handle alive while views exist


def make_config(K, T, nu=2, env_type="point_env", multi_modal=False, mode_simple=False,
                sampling_random=False, sample_null_action=True, filter_u=True,
                u_per_command=None, u_min=None, u_max=None, noise_sigma_diag=None, u_scale=1.0,
                gamma=0.95, lambda_=1.0, kp_suction=400.0, pre_height_diff=0.05, dt=None,
                substeps=2, solver_iters=6, seed=0, device=0, K_local=None, k_offset=0,
                cube_on_shelf=False, sim_only=False, shard_mix=False, noise_mu=None, noise_sigma=None,
                noise_abs_cost=False, update_cov=False) -> L.Config:
    """noise_sigma: the full [nu][nu] matrix (sets noise_sigma_diag too; full_sigma when it has off-diagonal
    entries)."""
    lib = L.load()
    c = L.Config()
    env = L.ENV_POINT if env_type in ("point_env", 0) else L.ENV_PANDA
    lib.m3_default_config(C.byref(c), env)
    c.device = int(device)
    c.K_global = int(K)
    c.K_local = int(K if K_local is None else K_local)
    c.k_offset = int(k_offset)
    c.T, c.nu = int(T), int(nu)
    c.multi_modal = int(bool(multi_modal))
    c.mode_simple = int(bool(mode_simple))
    c.sampling_random = int(bool(sampling_random))
    c.sample_null_action = int(bool(sample_null_action))
    c.filter_u = int(bool(filter_u))
    c.u_per_command = int(T if u_per_command is None else u_per_command)
    for name, vals in (("u_min", u_min), ("u_max", u_max), ("noise_sigma_diag", noise_sigma_diag)):
        if vals is not None:
            arr = getattr(c, name)
            for j in range(nu):
                arr[j] = float(vals[j])
    c.u_scale, c.gamma, c.lambda_ = float(u_scale), float(gamma), float(lambda_)
    c.kp_suction, c.pre_height_diff = float(kp_suction), float(pre_height_diff)
    if dt is not None:
        c.dt = float(dt)
    c.substeps, c.solver_iters = int(substeps), int(solver_iters)
    c.cube_on_shelf = int(bool(cube_on_shelf))
    c.sim_only = int(bool(sim_only))
    if noise_sigma is not None:
        for i in range(nu):
            c.noise_sigma_diag[i] = float(noise_sigma[i][i])
            for j in range(nu):
                c.noise_sigma_full[i * nu + j] = float(noise_sigma[i][j])
                if i != j and float(noise_sigma[i][j]) != 0.0:
                    c.full_sigma = 1
    if noise_mu is not None:
        for j in range(nu):
            c.noise_mu[j] = float(noise_mu[j])
    c.noise_abs_cost, c.update_cov = int(bool(noise_abs_cost)), int(bool(update_cov))
    c.shard_mix = int(shard_mix)   # 0: gather + reduce; 1 (True): one collective; 2: ... with ladder tables (multi-modal);
                                   # 3: two small exchanges, O(K_local) work per rank (multi-modal)
    c.seed = int(seed)
    return c


class HipEngine:
    def __init__(self, cfg: L.Config):
        if not torch.cuda.is_available():
            raise L.M3Error("HipEngine needs a HIP device (torch.cuda.is_available() is False); "
                            "there is no CPU fallback")
        self.lib = L.load()
        self.cfg = cfg
        self._h = C.c_void_p()
        torch.cuda.set_device(cfg.device)
        torch.zeros(1, device=f"cuda:{cfg.device}")  # make sure the HIP context exists
        L.check(self.lib.m3_create(C.byref(cfg), C.byref(self._h)))
        self.device = torch.device(f"cuda:{cfg.device}")
        self._views = {}
        self._action_out = None
        self.use_torch_stream()
        if os.environ.get("M3P2I_WAVE_ORDER", "1") == "0":   # experiments (tools/): samples to wavefronts by index
            self.set_wave_order(False)
        if os.environ.get("M3P2I_ROLLOUT_LANES"):            # experiments (tools/panda_lanes_sweep.py)
            self.set_rollout_lanes(int(os.environ["M3P2I_ROLLOUT_LANES"]))

    # ---- lifetime ----
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._views.clear()
            self.lib.m3_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        L.check(rc, self._h)

    # ---- configuration ----
    def use_torch_stream(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._ck(self.lib.m3_set_stream(self._h, C.c_void_p(s.cuda_stream)))

    def set_rollout_lanes(self, lanes=0):
        self._ck(self.lib.m3_set_rollout_lanes(self._h, int(lanes)))

    def set_panda_lanes_per_sample(self, lps=0):
        """panda_env rollout: 1 = a lane per sample, 16 = the sixteen lanes of a DPP row share a sample (the contact rows run
        across them), 0 = by size; same results bit for bit."""
        self._ck(self.lib.m3_set_panda_lanes_per_sample(self._h, int(lps)))

    def panda_lanes_per_sample_used(self):
        """The form the last panda rollout ran in (automatic choice: by size; for reach, by whether the last commands' rollouts
        brought the gripper within reach of a box)."""
        return int(self.lib.m3_panda_lanes_per_sample_used(self._h))

    def set_panda_reach_cost_kernel(self, on=True):
        """reach, unsharded, K <= 8192, default sampler: the cost in a kernel of its own behind a rollout without shadow slots
        (default) or the shadow slots; same bits."""
        self._ck(self.lib.m3_set_panda_reach_cost_kernel(self._h, int(bool(on))))

    def panda_near_share(self):
        """1/1000 of the last finished panda rollout's (sample, substep) pairs with the gripper within reach of a box (-1: none yet)."""
        return int(self.lib.m3_panda_near_share(self._h))

    def set_update_launches(self, launches=0):
        """0 / 3: the three-launch multi-modal update beyond k_update_small's range; 5: round 3's five launches."""
        self._ck(self.lib.m3_set_update_launches(self._h, int(launches)))

    def set_ladder_spins(self, spins=-1):
        """Bound of the multi-modal update's in-launch waits: -1 default, 0 = every workgroup takes the give-up branch."""
        self._ck(self.lib.m3_set_ladder_spins(self._h, int(spins)))

    def set_wave_order(self, on=True):
        """Samples sorted into coherent wavefronts (default) or assigned by index; same results."""
        self._ck(self.lib.m3_set_wave_order(self._h, int(bool(on))))

    def relabel_samples(self):
        """Permute the noise rows once into wavefront order (labels of a generated sample set carry
        no meaning): same sample set, coalesced stores.  See include/m3p2i_hip.h."""
        self._ck(self.lib.m3_relabel_samples(self._h))

    def enable_timing(self, on=True):
        self._ck(self.lib.m3_enable_timing(self._h, int(on)))

    @property
    def needs_global_noise(self):
        """One-collective multi-modal shard: the noise rows of ALL K_global samples live on every rank."""
        c = self.cfg
        return bool(c.shard_mix and c.K_local != c.K_global and c.multi_modal and not c.mode_simple)

    def set_noise(self, delta):
        """delta: [K_local, T, nu] (reference layout), torch (cpu/cuda) or numpy -- [K_global, T, nu] for a
        handle with `needs_global_noise`."""
        c = self.cfg
        rows, fn = (c.K_global, self.lib.m3_set_noise_global) if self.needs_global_noise else (c.K_local, self.lib.m3_set_noise)
        if isinstance(delta, torch.Tensor) and delta.is_cuda:
            d = delta.to(torch.float32).contiguous()
            assert tuple(d.shape) == (rows, c.T, c.nu), d.shape
            self._ck(fn(self._h, d.data_ptr(), 1))
            torch.cuda.current_stream(self.device).synchronize()
        else:
            d = np.ascontiguousarray(delta.cpu().numpy() if isinstance(delta, torch.Tensor)
                                     else delta, dtype=np.float32)
            assert d.shape == (rows, c.T, c.nu), d.shape
            self._ck(fn(self._h, d.ctypes.data, 0))

    def set_noise_knots(self, knots, degree=2, smoothing=0.5):
        """Halton-spline sampler on the device: knots [K_local, nu, n_knots] (Gaussian Halton values
        of this shard's samples; [K_global, ...] with `needs_global_noise`) -> smoothing-spline noise in
        the library's noise buffer."""
        c = self.cfg
        rows, fn = (c.K_global, self.lib.m3_set_noise_knots_global) if self.needs_global_noise \
            else (c.K_local, self.lib.m3_set_noise_knots)
        k = np.ascontiguousarray(knots.detach().cpu().numpy() if isinstance(knots, torch.Tensor) else knots,
                                 dtype=np.float32)
        assert k.ndim == 3 and k.shape[:2] == (rows, c.nu), k.shape
        self._ck(fn(self._h, k.ctypes.data, int(k.shape[2]), int(degree), float(smoothing), 0))

    def set_call_count(self, calls):
        self._ck(self.lib.m3_set_call_count(self._h, int(calls)))

    def sample_noise(self):
        """sampling_random / simple mode: this command's N(noise_mu, noise_sigma) draws into BUF_NOISE (what the
        fused rollout generates in registers) -- for the planner's STEP mode."""
        self._ck(self.lib.m3_sample_noise(self._h))
        return self.buffer(L.BUF_NOISE)

    def set_noise_halton(self, n_knots, degree=2, smoothing=0.5, scramble="none"):
        """The whole Halton-spline sampler on the device (knots included): include/m3p2i_hip.h.
        scramble: "none" (plain van der Corput, pinned by golden G8) or "faure" (generalized Halton)."""
        code = {"none": L.HALTON_PLAIN, "faure": L.HALTON_FAURE}[scramble]
        self._ck(self.lib.m3_set_noise_halton_scrambled(self._h, int(n_knots), int(degree), float(smoothing), code))

    def set_objective(self, task, goal, gripper_cmd=0):
        t = L.TASKS[task] if isinstance(task, str) else int(task)
        g = [float(x) for x in (goal.detach().cpu().reshape(-1).tolist()
                                if isinstance(goal, torch.Tensor) else np.ravel(goal))]
        arr = (C.c_float * len(g))(*g)
        self._ck(self.lib.m3_set_objective(self._h, t, arr, len(g), int(gripper_cmd)))

    def set_avoid_dyn_obs(self, on):
        """Extension (off = the reference): push / pull / push_pull add get_motion_cost like navigation does."""
        self._ck(self.lib.m3_set_avoid_dyn_obs(self._h, int(bool(on))))

    def set_multi_modal(self, mm):
        self._ck(self.lib.m3_set_multi_modal(self._h, int(bool(mm))))

    def set_plan(self, which, values):
        v = np.ascontiguousarray(values, dtype=np.float32)
        assert v.shape == (self.cfg.T, self.cfg.nu)
        self._ck(self.lib.m3_set_plan(self._h, which, v.ctypes.data))

    def set_beta(self, beta):
        self._ck(self.lib.m3_set_beta(self._h, float(beta)))

    def reset(self):
        self._ck(self.lib.m3_reset(self._h))

    def set_action_out(self, tensor):
        """Redirect the returned plan into a caller-owned device tensor (None: library buffer)."""
        self._action_out = tensor
        if tensor is None:
            self._ck(self.lib.m3_set_action_out(self._h, None))
            return
        rows = self.cfg.u_per_command if self.cfg.mode_simple else self.cfg.T
        assert tensor.is_cuda and tensor.dtype == torch.float32 and tensor.is_contiguous()
        assert tuple(tensor.shape) == (rows, self.cfg.nu)
        self._ck(self.lib.m3_set_action_out(self._h, C.c_void_p(tensor.data_ptr())))

    def set_world_point(self, robot, box, dyn_obs):
        w = L.PointWorld()
        for dst, src, n in ((w.robot, robot, 4), (w.box, box, 7), (w.dyn_obs, dyn_obs, 7)):
            for i in range(n):
                dst[i] = float(src[i])
        self._ck(self.lib.m3_set_world_point(self._h, C.byref(w)))

    def set_world_point_raw(self, w18):
        arr = (C.c_float * 18)(*[float(x) for x in w18])
        self._ck(self.lib.m3_set_world_point_raw(self._h, arr))

    def set_world_panda_raw(self, w57):
        """q9 qd9 | cubeA | cubeB | dyn-obs, each pos3 quat4(xyzw) linvel3 angvel3 (include/m3p2i_hip.h)"""
        if len(w57) != 57:
            raise ValueError("set_world_panda_raw: 57 floats (q9 qd9 cubeA13 cubeB13 dyn-obs13), got %d" % len(w57))
        arr = (C.c_float * 57)(*[float(x) for x in w57])
        self._ck(self.lib.m3_set_world_panda_raw(self._h, arr))

    def bind_sim_panda(self, dof_state, root_state, cubeA_actor, cubeB_actor, obs_actor):
        assert dof_state.is_cuda and root_state.is_cuda and dof_state.dtype == torch.float32
        self._bound = (dof_state, root_state)
        self._ck(self.lib.m3_bind_sim_panda(self._h, dof_state.data_ptr(), root_state.data_ptr(),
                                            root_state.shape[-2], cubeA_actor, cubeB_actor, obs_actor))

    def bind_sim_point(self, dof_state, root_state, box_actor, dyn_actor):
        assert dof_state.is_cuda and root_state.is_cuda and dof_state.dtype == torch.float32
        self._bound = (dof_state, root_state)  # keep alive
        self._ck(self.lib.m3_bind_sim_point(self._h, dof_state.data_ptr(), root_state.data_ptr(),
                                            root_state.shape[-2], box_actor, dyn_actor))

    # ---- the hot path ----
    def command(self, sync_host=False):
        """One MPPI iteration.  Returns the device tensor [T, nu] of the filtered plan."""
        if sync_host:
            rows = self.cfg.u_per_command if self.cfg.mode_simple else self.cfg.T
            out = np.zeros((rows, self.cfg.nu), np.float32)
            self._ck(self.lib.m3_command(self._h, out.ctypes.data))
            return out
        self._ck(self.lib.m3_command(self._h, None))
        return self._action_out if self._action_out is not None else self.buffer(L.BUF_ACTION_OUT)

    def rollout(self):
        self._ck(self.lib.m3_rollout(self._h))

    def update(self):
        self._ck(self.lib.m3_update(self._h))

    def update_b(self):
        """shard_mix = 3, between the two exchanges: searches on the mixed tables, local weights and sums."""
        self._ck(self.lib.m3_update_b(self._h))

    def finalize(self):
        self._ck(self.lib.m3_finalize(self._h))

    # ---- device-side exchange of the records (include/m3p2i_hip.h: m3_p2p_*) ----
    def p2p_export(self) -> bytes:
        """This rank's exchange block as an IPC handle (64 bytes) for the other ranks' p2p_connect."""
        buf = C.create_string_buffer(L.IPC_HANDLE_BYTES)
        self._ck(self.lib.m3_p2p_export(self._h, C.cast(buf, C.c_void_p)))
        return buf.raw

    def p2p_connect(self, handles):
        """handles: the p2p_export() of every rank, in rank order (other processes)."""
        blob = b"".join(bytes(x) for x in handles)
        assert len(blob) == L.IPC_HANDLE_BYTES * len(handles)
        buf = C.create_string_buffer(blob, len(blob))
        self._ck(self.lib.m3_p2p_connect(self._h, C.cast(buf, C.c_void_p), len(handles)))

    def p2p_connect_local(self, engines):
        """engines: the HipEngine of every rank, in rank order, all living in this process."""
        arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
        self._p2p_peers = list(engines)     # keep the peers' blocks alive as long as this handle writes into them
        self._ck(self.lib.m3_p2p_connect_local(self._h, arr, len(engines)))

    def p2p_put(self, channel=0):
        """channel 0: M3_BUF_RECORD; 1: M3_BUF_RECORD_B (the second exchange of shard_mix = 3)."""
        self._ck(self.lib.m3_p2p_put_ch(self._h, int(channel)))

    def p2p_wait(self, channel=0):
        self._ck(self.lib.m3_p2p_wait_ch(self._h, int(channel)))

    def p2p_exchange(self, channel=0):
        """put + wait in one launch (one process per GPU; a process driving several handles on one stream uses
        p2p_put / p2p_wait, every put before any wait)."""
        self._ck(self.lib.m3_p2p_exchange(self._h) if channel == 0 else self.lib.m3_p2p_exchange_b(self._h))

    def p2p_status(self):
        """(missing_rank or -1, memory kind: 1 uncached / 2 fine-grained / 3 plain); synchronises the stream."""
        miss, kind = C.c_int(), C.c_int()
        self._ck(self.lib.m3_p2p_status(self._h, C.byref(miss), C.byref(kind)))
        return miss.value, kind.value

    def p2p_set_timeout_ms(self, first_ms=30000, ms=500):
        """How long a wait spins for a missing peer: a channel's first exchange (start-up skew between the ranks) / later ones."""
        self._ck(self.lib.m3_p2p_set_timeout_ms(self._h, int(first_ms), int(ms)))

    def p2p_clear_error(self):
        """Collective re-arm after a wait that gave up: own flags, error word and sequence numbers back to zero (the caller
        holds a barrier before and after: distributed.p2p_recover)."""
        self._ck(self.lib.m3_p2p_clear_error(self._h))

    def p2p_detach(self):
        """This handle stops using the device-side exchange: finalize no longer reads its error word."""
        self._ck(self.lib.m3_p2p_detach(self._h))

    def p2p_set_memory_kind(self, first_kind=1):
        """Where the exchange block's allocation chain starts (1 uncached, 2 fine-grained, 3 plain); before export / connect."""
        self._ck(self.lib.m3_p2p_set_memory_kind(self._h, int(first_kind)))

    def update_finalize(self):
        """update + finalize of an unsharded handle in as few launches as possible."""
        self._ck(self.lib.m3_update_finalize(self._h))

    # ---- views ----
    def _shape(self, which):
        c = self.cfg
        Kl, Kg, T, nu = c.K_local, c.K_global, c.T, c.nu
        return {
            L.BUF_STATES: ((T, Kl, 4), "<f4"), L.BUF_ACTIONS: ((T, Kl, nu), "<f4"),
            L.BUF_COST_HORIZON: ((T, Kl), "<f4"), L.BUF_TRAJ_COST: ((Kl,), "<f4"),
            L.BUF_TRAJ_COST_ALL: ((Kg,), "<f4"), L.BUF_WEIGHTS: ((Kg,), "<f4"),
            L.BUF_WEIGHTS_1: ((Kg // 2,), "<f4"), L.BUF_WEIGHTS_2: ((Kg - Kg // 2,), "<f4"),
            L.BUF_MEAN: ((T, nu), "<f4"), L.BUF_MEAN_1: ((T, nu), "<f4"),
            L.BUF_MEAN_2: ((T, nu), "<f4"), L.BUF_BEST: ((T, nu), "<f4"),
            L.BUF_BEST_1: ((T, nu), "<f4"), L.BUF_BEST_2: ((T, nu), "<f4"),
            L.BUF_ACTION_OUT: ((T, nu), "<f4"), L.BUF_TOP_IDX: ((L.TOPK,), "<i4"),
            L.BUF_TOP_TRAJS: ((L.TOPK, T, 2), "<f4"),
            L.BUF_REDUCE: ((self.lib.m3_reduce_len(self._h),), "<f4"),
            L.BUF_NOISE: ((T, Kl, nu), "<f4"), L.BUF_PENDING_FORCE: ((4, Kl), "<f4"),
            L.BUF_SIM_WORLD: ((28 if c.env_type == L.ENV_POINT else 77, Kl), "<f4"),
            L.BUF_INFO: ((L.INFO_WORDS,), "<i4"),
            L.BUF_NOISE_ALL: ((Kg // Kl, T, Kl, nu), "<f4"),
            L.BUF_COV: ((2, nu), "<f4"),
            L.BUF_RECORD: ((self.lib.m3_record_len(self._h),), "<f4"),
            L.BUF_RECORDS_ALL: ((Kg // Kl, self.lib.m3_record_len(self._h)), "<f4"),
            L.BUF_RECORD_B: ((self.lib.m3_record_b_len(self._h),), "<f4"),
            L.BUF_RECORDS_B_ALL: ((Kg // Kl, self.lib.m3_record_b_len(self._h)), "<f4"),
        }[which]

    def buffer(self, which) -> torch.Tensor:
        """Zero-copy torch view of a library-owned device buffer."""
        if which in self._views:
            return self._views[which]
        p = C.c_void_p()
        n = C.c_longlong()
        self._ck(self.lib.m3_get_buffer(self._h, which, C.byref(p), C.byref(n)))
        shape, typestr = self._shape(which)
        t = torch.as_tensor(_DevArray(p.value, shape, typestr, self), device=self.device)
        self._views[which] = t
        return t

    def info(self) -> L.Info:
        i = L.Info()
        self._ck(self.lib.m3_get_info(self._h, C.byref(i)))
        return i

    def timing(self) -> L.Timing:
        t = L.Timing()
        self._ck(self.lib.m3_get_timing(self._h, C.byref(t)))
        return t

    # reference-layout conveniences ([K, T, c] strided views, no copies)
    @property
    def states(self):
        return self.buffer(L.BUF_STATES).permute(1, 0, 2)

    @property
    def actions(self):
        return self.buffer(L.BUF_ACTIONS).permute(1, 0, 2)

    @property
    def cost_horizon(self):
        return self.buffer(L.BUF_COST_HORIZON).permute(1, 0)

    # ---- step mode ----
    def sim_bind_views(self, dof_state, root_state, rigid_body_state, net_contact_force):
        self._simviews = (dof_state, root_state, rigid_body_state, net_contact_force)
        self._ck(self.lib.m3_sim_bind_views(
            self._h, dof_state.data_ptr(), root_state.data_ptr(), rigid_body_state.data_ptr(),
            net_contact_force.data_ptr(), root_state.shape[1], rigid_body_state.shape[1]))

    def sim_pull_state(self):
        self._ck(self.lib.m3_sim_pull_state(self._h))

    def sim_shift_actor(self, actor, dx, dy, dz=0.0):
        """root position of one actor of every environment += (dx, dy, dz) in the bound view, then the state
        upload -- one launch (update_dyn_obs)."""
        self._ck(self.lib.m3_sim_shift_actor(self._h, int(actor), float(dx), float(dy), float(dz)))

    def sim_push_state(self):
        self._ck(self.lib.m3_sim_push_state(self._h))

    def sim_set_velocity_target(self, u, zero_copy=False):
        """Velocity targets for the next sim_step().  Default: COPIED at this point into an engine-owned tensor (one
        small device copy on the stream) -- Isaac Gym's set_dof_velocity_target_tensor semantics: what the caller does
        with `u` afterwards does not matter.  zero_copy=True (the closed-loop tools, which own their tensors): the
        pointer is handed to the step kernel instead (one launch less per tick); the caller must leave the tensor alone
        until step() -- an in-place torch op in between is detected and refused there, a write by another library
        call (e.g. the planner's action ring being rewritten eight commands later) is not."""
        u = u.to(torch.float32).contiguous()
        assert u.is_cuda and tuple(u.shape) == (self.cfg.K_local, self.cfg.nu), u.shape
        if not zero_copy:
            if getattr(self, "_u_target", None) is None:
                self._u_target = torch.empty(self.cfg.K_local, self.cfg.nu, device=self.device, dtype=torch.float32)
            self._u_target.copy_(u)
            u = self._u_target
        self._pending_u = (u, u._version)

    def sim_apply_body_forces(self, f):
        f = f.to(torch.float32).contiguous()
        assert f.is_cuda
        self._ck(self.lib.m3_sim_apply_body_forces(self._h, f.data_ptr()))

    def sim_step(self):
        pending = getattr(self, "_pending_u", None)
        if pending is None:
            self._ck(self.lib.m3_sim_step(self._h))      # the targets of the last set call stay in force
            return
        u, version = pending
        self._pending_u = None      # (cleared on every path: a failed step must not leave a stale target behind)
        if u._version != version:
            raise RuntimeError("the velocity-target tensor was modified in place between "
                               "set_dof_velocity_target_tensor() and step(): pass a clone")
        self._ck(self.lib.m3_sim_step_with_target(self._h, u.data_ptr()))

    def sim_suction_forces(self, kp_suction):
        """calculate_suction on the device: [K_local, n_bodies, 3] body forces (skill_utils.py:59-94)."""
        nb = self._simviews[2].shape[1]
        f = torch.empty(self.cfg.K_local, nb, 3, device=self.device, dtype=torch.float32)
        self._ck(self.lib.m3_sim_suction_forces(self._h, float(kp_suction), f.data_ptr()))
        return f

    def sim_check_and_apply_suction(self, action, kp_suction, apply=True, want_flags=False, enabled=None):
        """check_suction_condition (+ apply_rigid_body_force_tensors(calculate_suction) where it holds)
        on the device, no host sync (skill_utils.py:36-56).  enabled: optional 1-element int32 device
        tensor that gates the suction (the planner's pull preference).  Returns the per-env condition
        (int32 device tensor) if want_flags."""
        a = action.to(device=self.device, dtype=torch.float32).reshape(self.cfg.K_local, 2).contiguous()
        flags = torch.empty(self.cfg.K_local, device=self.device, dtype=torch.int32) if want_flags else None
        if enabled is not None:
            assert enabled.is_cuda and enabled.dtype == torch.int32 and enabled.numel() == 1
            self._gate = enabled   # keep alive until the kernel ran
        self._ck(self.lib.m3_sim_check_and_apply_suction(
            self._h, a.data_ptr(), float(kp_suction), int(bool(apply)),
            C.c_void_p(flags.data_ptr()) if want_flags else None,
            C.c_void_p(enabled.data_ptr()) if enabled is not None else None))
        return flags

    def cost(self, out=None):
        if out is None:
            out = torch.empty(self.cfg.K_local, device=self.device, dtype=torch.float32)
        self._ck(self.lib.m3_cost(self._h, out.data_ptr()))
        return out
