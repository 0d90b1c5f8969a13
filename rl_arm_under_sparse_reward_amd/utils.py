"""sync_networks / sync_grads -- mirrors of the reference's utils.py on RCCL.

Reference semantics (utils.py:6-69), preserved:
  * sync_networks(net): every rank ends up with rank 0's parameters (MPI Bcast, utils.py:13);
  * sync_grads(net):   gradients are SUMMED over ranks, not averaged (MPI Allreduce SUM,
                       utils.py:47), so the effective learning rate scales with world size.
The transport is torch.distributed ("nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU
tests), one process per GPU.  When a network is attached to a `ddpg_agent`, the exchange runs
directly on the library's device vectors (zero copy); otherwise on the module's host tensors.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


class Communicator:
    """Thin view of the default torch.distributed process group (or a single rank)."""

    def __init__(self, device_id=None, force=False):
        import torch.distributed as dist

        self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.device_id = device_id
        self.force = bool(force) and self._dist is not None   # run the collectives even in a 1-rank group (tests)

    @property
    def active(self):
        """True when the data-parallel exchange steps must run (more than one rank, or forced)."""
        return self.world_size > 1 or self.force

    @property
    def world_size(self):
        return self._dist.get_world_size() if self._dist else 1

    @property
    def rank(self):
        return self._dist.get_rank() if self._dist else 0

    # ---- tensor-level collectives (host or device tensors)
    def allreduce_sum_(self, t: torch.Tensor):
        if self._dist:
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t

    def allreduce_mean_(self, t: torch.Tensor):
        """normalizer.py:60-64: Allreduce(SUM) then divide by the number of ranks."""
        if self._dist:
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
            t /= self.world_size
        return t

    def broadcast_(self, t: torch.Tensor, root=0):
        if self._dist:
            self._dist.broadcast(t, src=root)
        return t

    # ---- zero-copy collectives on library-owned device vectors
    def _view(self, address, n):
        dev = self.device_id if self.device_id is not None else torch.cuda.current_device()
        return torch.as_tensor(_lib.DevicePointer(address, n), device=f"cuda:{dev}")

    def allreduce_sum_device(self, address, n):
        if self.active:
            self.allreduce_sum_(self._view(address, n))

    def allreduce_mean_device(self, address, n):
        if self.active:
            self.allreduce_mean_(self._view(address, n))

    def broadcast_device(self, address, n, root=0):
        if self.active:
            self.broadcast_(self._view(address, n), root)


def _flat(tensors):
    return torch.cat([t.detach().reshape(-1) for t in tensors])


def sync_networks(network, comm: Communicator | None = None):
    """utils.py:6-15."""
    comm = comm or Communicator()
    learner = getattr(network, "_learner", None)
    if learner is not None:
        learner._broadcast_params(comm)
        return
    params = [p for _, p in network.named_parameters()]
    flat = _flat(params)
    comm.broadcast_(flat, 0)
    off = 0
    with torch.no_grad():
        for p in params:
            p.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()


def sync_grads(network, comm: Communicator | None = None):
    """utils.py:43-48 (SUM)."""
    comm = comm or Communicator()
    learner = getattr(network, "_learner", None)
    if learner is not None:
        learner._allreduce_grads(comm)
        return
    params = [p for _, p in network.named_parameters()]
    flat = _flat([p.grad for p in params])
    comm.allreduce_sum_(flat)
    off = 0
    for p in params:
        p.grad.copy_(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
