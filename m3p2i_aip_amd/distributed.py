"""Sample sharding across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI).

The rollouts of different samples are independent for the whole horizon (mppi.py:296-315);
only the importance-weight update couples them.  Per command():

    rollout (local K/N samples)
      -> all_gather  J[K]                 K floats: every rank then runs the SAME min /
                                          beta search / normalisation on all K costs, so the
                                          weights are bit-identical on every rank and equal
                                          to the single-GPU ones (the reference's on-the-fly
                                          beta search needs eta(beta) over the whole set at
                                          every pass -- one reduce of weight sums would only
                                          be exact for a fixed beta)
    update  (weights for all K, weighted sums over the local shard)
      -> all_reduce(sum) packed buffer    3 x [T,nu] weighted sums, 3 x [T,nu] best rows
                                          (zero except on the owning rank), 20 x [T,2] top
                                          trajectories (same) : 6*T*nu + 40*T floats
    finalize (identical on every rank)

Both messages are latency-bound (<= 256 KB and ~6 KB at K=64000): bucket size and ring
bandwidth over the 7 xGMI links are irrelevant here, hop count is what matters.

Single-mode MPPI (cfg.multi_modal False; beta is fixed during a command, mppi.py:430-456) needs
only ONE collective (planner.shard_mix, the default there):

    rollout -> update (softmin over the LOCAL shard: m_r, eta_r, normalised sums, best, top-20)
      -> all_gather  record[~1.6 K floats]
    finalize (k_mix: rho_r = exp(-(m_r - m)/beta) eta_r / Z; sums = sum_r rho_r S_r; merge of the
              ranks' best / top-20; then the usual finalize) -- identical on every rank, equal to
              the two-collective result up to f32 rounding.
"""
from __future__ import annotations

import torch.distributed as dist

from . import _lib as L


def attach_collectives(planner, group=None):
    """Install the two collectives on a planner built with cfg.mppi.rank/world_size."""
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
        else:
            raise ValueError(phase)

    planner.collective = exchange
    return planner
