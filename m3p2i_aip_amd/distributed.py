"""Sample sharding across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI).

The rollouts of different samples are independent for the whole horizon (mppi.py:296-315);
only the importance-weight update couples them.  Default protocol
(planner.shard_mix): ONE collective per command(), an all-gather of small per-rank records.

  single-mode MPPI (beta is fixed during a command, mppi.py:430-456)
    rollout -> update (softmin over the LOCAL shard: m_r, eta_r, normalised sums, best, top-20)
      -> all_gather  record[~1.6 K floats]
    finalize (k_mix: rho_r = exp(-(m_r - m)/beta) eta_r / Z; sums = sum_r rho_r S_r; merge of the
              ranks' best / top-20; then the usual finalize) -- identical on every rank, equal to the
              unsharded result up to f32 rounding.

  multi-modal M3P2I (the on-the-fly beta search needs eta(beta) over ALL samples at every pass,
  m3p2i.py:24-64 -- a reduce of weight sums would only be exact for a fixed beta)
    rollout -> update (the shard's own top-20, minima and ladder sums eta_r(beta_j) into its record)
      -> all_gather  record = {J of the shard [K/N] | top-20 costs, indices, trajectories | minima, ladder table}
    finalize (the searches walk the MIXTURE of the shards' ladder tables -- passes over the gathered costs
              only if a search leaves its ladder --, then one kernel forms the weights of all K samples, the
              weighted action sums with the other ranks' actions RE-GENERATED from the replicated noise
              table and plan (a_k[t] is a function of the global sample index, mppi.py:381-416) instead of
              communicated, the best rows and the plan.)
    Every rank holds the noise rows of all K samples (15 MB at K = 64000, init only).  Equal to the
    single-GPU run of the same K up to f32 rounding (same iteration counts, plan <= 3e-5), identical on every
    rank; MPPIConfig.shard_mix = 1 selects the variant that re-evaluates all K costs with the unsharded kernels
    and is bit-identical to the single-GPU run (tests/test_c5_sharded_gpu.py).

shard_mix = 3 (multi-modal, opt-in: more ranks / samples per GPU than BASELINE's): TWO small exchanges,
    rollout -> update (as above: costs | top-20 | minima, ladder table) -> all_gather record
      -> update_b (searches on the mixed tables; weights of the rank's OWN samples; their weighted action sums from
                   its own action buffer -- nothing re-generated, nothing of size K_global touched)
      -> all_gather record_b (~6 T nu floats) -> finalize (sums in rank order, best rows of the winning rank, plan).

Fallback protocol (shard_mix=False; also the multi-modal path with sampling_method='random', whose
in-kernel noise has no table): two collectives,
    rollout -> all_gather J[K] -> update (weights for all K, weighted sums over the local shard)
      -> all_reduce(sum) packed buffer (6*T*nu + 40*T floats) -> finalize.

Transport of the records phase: RCCL (`attach_collectives`, a host-issued all_gather on the stream) or the library's
own device-side exchange over peer-mapped memory (`attach_p2p`: every rank stores its record straight into every
peer's block -- one hop over xGMI -- and acquires the peers' flags; two small kernels on the handle's stream, no
library call, csrc/p2p.hip).

All messages are latency-bound (<= 300 KB at K = 64000): bucket size and ring bandwidth over the 7
xGMI links are irrelevant here; the number of dependent collectives per command is what counts.
"""
from __future__ import annotations

import torch.distributed as dist

from . import _lib as L


def attach_collectives(planner, group=None):
    """Install the collective(s) on a planner built with cfg.mppi.rank/world_size."""
    if planner.world_size != dist.get_world_size(group):
        raise ValueError("planner.world_size does not match the process group")

    def exchange(pl, phase):
        e = pl._engine
        if phase == "gather":
            out, loc = e.buffer(L.BUF_TRAJ_COST_ALL), e.buffer(L.BUF_TRAJ_COST)
            try:
                dist.all_gather_into_tensor(out, loc, group=group)
            except (RuntimeError, NotImplementedError):  # backends without the flat variant
                dist.all_gather(list(out.chunk(pl.world_size)), loc, group=group)
        elif phase == "reduce":
            dist.all_reduce(e.buffer(L.BUF_REDUCE), op=dist.ReduceOp.SUM, group=group)
        elif phase == "records":
            out, loc = e.buffer(L.BUF_RECORDS_ALL), e.buffer(L.BUF_RECORD)
            try:
                dist.all_gather_into_tensor(out, loc, group=group)
            except (RuntimeError, NotImplementedError):
                dist.all_gather(list(out.unbind(0)), loc, group=group)
        elif phase == "records_b":       # shard_mix = 3: the second, small exchange
            out, loc = e.buffer(L.BUF_RECORDS_B_ALL), e.buffer(L.BUF_RECORD_B)
            try:
                dist.all_gather_into_tensor(out, loc, group=group)
            except (RuntimeError, NotImplementedError):
                dist.all_gather(list(out.unbind(0)), loc, group=group)
        else:
            raise ValueError(phase)

    planner.collective = exchange
    planner.transport = dist.get_backend(group)      # "nccl" is RCCL on ROCm
    planner.ranks_seen = dist.get_world_size(group)  # the communicator's size as the backend reports it
    return planner


P2P_POLL_EVERY = 64      # commands between two reads of the exchange's error word (a stream sync + 4 bytes: ~10 us)


def attach_p2p(planner, group=None, poll_every=P2P_POLL_EVERY):
    """The records exchange of a shard_mix planner through peer-mapped device memory instead of RCCL (one process
    per GPU; the IPC handles of the exchange blocks travel once, at set-up, through the process group's object
    gather).  Planners without the one-collective protocol (shard_mix=False) keep their two RCCL collectives.

    Skew bound.  A wait spins for at most the handle's time-out -- 30 s for a channel's first exchange, 0.5 s
    afterwards (`engine.p2p_set_timeout_ms`) -- so the ranks must reach every command within that of each other (a rank
    stalled longer: garbage collection, rendering, a failed bench row).  A wait that gives up does NOT hang the GPU and
    does NOT pass stale data on: the missing rank's slot is filled with NaN and a sticky error word is set; while it is set
    the finalize kernels hand out NaN plans (this command's and every later one's) and leave the warm-start state -- means,
    best trajectories -- untouched, so nothing can act on a plan built from a part of the samples and nothing is poisoned.
    The host learns of it by reading the word, which synchronises the stream: this transport does so every `poll_every`
    commands (an asynchronous command loop; `poll_every=1` for a loop that reads every action back anyway, where the read
    is a synchronisation point already), in `planner.p2p_check()`, and in `detach_p2p(planner)`; each raises RuntimeError
    naming the rank that never arrived -- where the RCCL transport would have blocked."""
    if planner.world_size != dist.get_world_size(group):
        raise ValueError("planner.world_size does not match the process group")
    if not planner.shard_mix:
        return attach_collectives(planner, group)
    e = planner._engine
    # Preflight, agreed on by ALL ranks (a transport chosen per rank would deadlock): every rank's device must be able
    # to map every other rank's -- hipDeviceCanAccessPeer for each pair this process can see (two ranks on one device:
    # the single-GPU emulation of the tests) -- and the IPC handles must open.  Anything else: the RCCL transport, with
    # the reason logged once by rank 0.
    import torch
    my_dev = torch.device(planner.device).index or 0
    devs = [None] * planner.world_size
    dist.all_gather_object(devs, my_dev, group=group)
    reason = None
    n_vis = torch.cuda.device_count()
    for d in devs:
        if d != my_dev and d < n_vis and my_dev < n_vis and not torch.cuda.can_device_access_peer(my_dev, d):
            reason = f"device {my_dev} cannot access device {d} (hipDeviceCanAccessPeer)"
            break
    if reason is None:
        try:
            mine = e.p2p_export()
        except RuntimeError as ex:
            mine, reason = None, f"exchange block / IPC export failed: {ex}"
    else:
        mine = None
    handles = [None] * planner.world_size
    dist.all_gather_object(handles, (mine, reason), group=group)
    if all(r is None for _, r in handles):
        try:
            e.p2p_connect([h for h, _ in handles])
        except RuntimeError as ex:
            reason = f"hipIpcOpenMemHandle failed: {ex}"
    verdicts = [None] * planner.world_size
    dist.all_gather_object(verdicts, reason if reason is not None else next((r for _, r in handles if r), None), group=group)
    bad = next((v for v in verdicts if v), None)
    if bad is not None:
        if planner.rank == 0:
            import sys
            print(f"m3p2i_aip_amd.distributed: p2p exchange unavailable ({bad}); using the RCCL collectives", file=sys.stderr)
        attach_collectives(planner, group)
        planner.p2p_fallback_reason = bad
        return planner
    dist.barrier(group=group)      # nobody starts exchanging before every rank has mapped every block

    state = {"n": 0}

    def check(pl):
        missing, _kind = pl._engine.p2p_status()
        if missing >= 0:
            raise RuntimeError("p2p exchange: rank %d did not arrive within the wait's time-out (engine.p2p_set_timeout_ms); "
                               "this rank's plans since then are NaN" % missing)

    def exchange(pl, phase):
        if phase not in ("records", "records_b"):
            raise ValueError(phase)
        pl._engine.p2p_exchange(0 if phase == "records" else 1)
        if phase == "records":
            state["n"] += 1
            if poll_every and state["n"] % poll_every == 0:
                check(pl)

    planner.collective = exchange
    planner.transport = "p2p"
    planner.ranks_seen = dist.get_world_size(group)
    planner.p2p_check = lambda: check(planner)     # explicit poll (tools, tests, before shutting a rank down)
    planner._p2p_group = group
    return planner


def p2p_recover(planner):
    """Collective recovery after a wait that gave up (every rank calls it, e.g. from the handler of the RuntimeError that
    `p2p_check` raised -- the ranks that did not time out see nothing wrong and must be told by the application): barrier
    (nobody is still exchanging) -> every rank zeroes its own block's flags and error word and restarts its sequence
    numbers (`m3_p2p_clear_error`) -> barrier (nobody puts into a block about to be zeroed).  The planners continue from
    their last good plans: the finalize kernels never overwrote the warm-start state with the NaN plan."""
    if getattr(planner, "transport", None) != "p2p":
        raise RuntimeError("p2p_recover: the planner does not use the p2p transport")
    group = getattr(planner, "_p2p_group", None)
    dist.barrier(group=group)
    planner._engine.p2p_clear_error()
    dist.barrier(group=group)
    return planner


def detach_p2p(planner):
    """Back to the RCCL collectives; reads the exchange's error word first (raises if a wait ever gave up).  Either way the
    handle stops reading the word (`m3_p2p_detach`): after the exception the planner hands out finite plans again over
    RCCL, from its last good warm-start state."""
    chk = getattr(planner, "p2p_check", None)
    if chk is None:
        return planner
    try:
        chk()
    finally:
        planner.p2p_check = None
        planner._engine.p2p_detach()
        attach_collectives(planner, getattr(planner, "_p2p_group", None))
    return planner
