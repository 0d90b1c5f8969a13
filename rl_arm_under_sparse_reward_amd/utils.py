"""sync_networks / sync_grads -- mirrors of the reference's utils.py on RCCL.

Reference semantics (utils.py:6-69), preserved:
  * sync_networks(net): every rank ends up with rank 0's parameters (MPI Bcast, utils.py:13);
  * sync_grads(net):   gradients are SUMMED over ranks, not averaged (MPI Allreduce SUM,
                       utils.py:47), so the effective learning rate scales with world size.
The transport is RCCL over xGMI, one process per GPU.  torch.distributed provides the process group
(rendezvous, barriers, the "gloo" CPU tests); on GPUs the collectives of the hot path are issued by the
library itself on its own stream (`hp_comm_*`, csrc/comm.hip) so that a whole training cycle -- including
the per-update gradient all-reduce -- is one hipGraph with no host round trip per update.  The native
communicator is bootstrapped from the torch group (rank 0's RCCL id is broadcast through it);
RLARM_COMM=torch keeps every collective on torch.distributed instead.  When a network is attached to a
`ddpg_agent`, the exchange runs directly on the library's device vectors (zero copy); otherwise on the
module's host tensors.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib


class Communicator:
    """Thin view of the default torch.distributed process group (or a single rank)."""

    def __init__(self, device_id=None, force=False, gate=None):
        import torch.distributed as dist

        self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.device_id = device_id
        self.force = bool(force) and self._dist is not None   # run the collectives even in a 1-rank group (tests)
        self.native = None        # library-side RCCL communicator (hp_comm *), see attach_native
        self._native_lib = None
        self.peer = None          # library-side peer-memory exchange (hp_peer *), see attach_peer
        self.shared_device = False   # two or more ranks on one physical device (set by attach_peer)
        self.gate = gate             # peer exchange: waits in gate kernels (hp_peer_set_gate); None = exactly when ranks share a device
        self.peer_refused = None     # why attach_peer did not return an exchange although it was tried (for the run's record)

    def agree(self, flag, ctx=None):
        """True iff `flag` is true on EVERY rank (collective).  Transport decisions must be taken by all ranks together."""
        if self._dist is None:
            return bool(flag)
        if self._dist.get_backend() == "nccl":
            dev_id = ctx.device_id if ctx is not None else (self.device_id if self.device_id is not None else torch.cuda.current_device())
            dev = torch.device("cuda", dev_id)
        else:
            dev = torch.device("cpu")
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def drop_peer(self):
        """Destroy the peer-memory exchange (every rank, together: e.g. the agent's engine cannot use it), so that the
        next transport down is attached and the exchange memory / IPC mappings do not stay open."""
        if self.peer is not None:
            self._native_lib.hp_peer_destroy(self.peer)
            self.peer = None

    def attach_native(self, ctx):
        """Create (once) the library's own RCCL communicator for this rank.  Collective: every rank of the
        group must call it.  Returns the handle, or None when the group is not on GPUs / RLARM_COMM=torch."""
        if self.native is not None:
            return self.native
        if not self.active or os.environ.get("RLARM_COMM", "auto") not in ("auto", "native", "rccl"):
            return None
        if self._dist.get_backend() != "nccl":
            return None
        lib = ctx.lib
        ident = torch.zeros(128, dtype=torch.uint8, device=f"cuda:{ctx.device_id}")
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            _lib.check(lib.hp_comm_unique_id(buf))
            ident.copy_(torch.tensor(list(buf), dtype=torch.uint8))
        self._dist.broadcast(ident, src=0)
        raw = (C.c_uint8 * 128)(*ident.cpu().tolist())
        h = C.c_void_p()
        status = lib.hp_comm_create(ctx.h, raw, self.rank, self.world_size, C.byref(h))
        # all ranks must agree on the transport: if the communicator could not be created anywhere (RCCL not loadable,
        # two ranks on one GPU, ...) everybody falls back to torch.distributed
        ok = torch.tensor([1 if status == 0 else 0], dtype=torch.int32, device=ident.device)
        self._dist.all_reduce(ok, op=self._dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if status == 0:
                lib.hp_comm_destroy(h)
            return None
        self.native, self._native_lib = h, lib
        return h

    def attach_peer(self, ctx, n_grad_floats):
        """Create (once) the one-shot peer-memory exchange (csrc/peer.hip) for this rank: every rank exports a block of
        device memory, the 64-byte IPC handles travel through the torch group, every rank maps the others' blocks, and a
        self-check all-reduce must give the exact expected sums on every rank.  Collective.  Returns the handle, or None
        (RLARM_COMM selects another transport, ranks cannot map each other's memory, self-check failed on any rank)."""
        if self.peer is not None:
            return self.peer
        want = os.environ.get("RLARM_COMM", "auto")
        if not self.active or want not in ("auto", "peer") or self.world_size > 16:
            return None
        lib, dist = ctx.lib, self._dist
        dev = torch.device("cuda", ctx.device_id) if dist.get_backend() == "nccl" else torch.device("cpu")

        def agree(flag):   # every rank must take the same decision
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1

        h = C.c_void_p()
        handle = (C.c_uint8 * 64)()
        ok = lib.hp_peer_create(ctx.h, self.rank, self.world_size, int(n_grad_floats), C.byref(h), handle) == 0
        mine = torch.tensor(list(handle) if ok else [0] * 64, dtype=torch.uint8, device=dev)
        every = [torch.zeros(64, dtype=torch.uint8, device=dev) for _ in range(self.world_size)]
        dist.all_gather(every, mine)
        if not agree(ok):
            self.peer_refused = "hp_peer_create failed" + ("" if not ok else " on another rank") + (
                ": " + _lib.last_error(lib) if not ok else "")
            if ok:
                lib.hp_peer_destroy(h)
            return None
        raw = (C.c_uint8 * (64 * self.world_size))(*[int(b) for t in every for b in t.cpu().tolist()])
        # ranks that share one physical device (an N-rank rehearsal on fewer GPUs) depend on each other's kernels being
        # co-resident: their waits go into one-wavefront gate kernels (hp_peer_set_gate) so that waiting ranks cannot
        # starve a computing one of registers / LDS.  One rank per GPU (the real job): no gate, no extra kernel boundary
        bus = C.create_string_buffer(32)
        _lib.check(lib.hp_ctx_pci_bus_id(ctx.h, bus, 32))
        mine_id = torch.tensor(list(bus.raw), dtype=torch.uint8, device=dev)
        ids = [torch.zeros(32, dtype=torch.uint8, device=dev) for _ in range(self.world_size)]
        dist.all_gather(ids, mine_id)
        self.shared_device = len({bytes(t.cpu().tolist()) for t in ids}) < self.world_size
        _lib.check(lib.hp_peer_set_gate(h, 1 if (self.shared_device if self.gate is None else self.gate) else 0))
        ok = lib.hp_peer_connect(h, raw) == 0
        if not agree(ok):      # before any collective kernel: a rank that could not map its peers must not leave the others waiting
            self.peer_refused = ("mapping the peers' exchange memory failed" + (" on another rank" if ok else ": " + _lib.last_error(lib)))
            lib.hp_peer_destroy(h)
            return None
        # self-check: rank r contributes (r + 1) * (i + 1); the rank-ordered sum is exact in float32.  Every rank runs EVERY
        # collective step whatever its own intermediate results (a rank that stopped early would leave the others spinning
        # in the remaining ones until the device timeout); the verdicts are combined afterwards
        n, w = 257, self.world_size
        probe = torch.arange(1, n + 1, dtype=torch.float32, device=f"cuda:{ctx.device_id}") * float(self.rank + 1)
        torch.cuda.synchronize(ctx.device_id)
        for mean in (0, 1):
            v = probe.clone()
            torch.cuda.synchronize(ctx.device_id)
            launched = lib.hp_peer_allreduce_f32(h, C.c_void_p(v.data_ptr()), n, mean) == 0
            ctx.synchronize()
            expect = torch.arange(1, n + 1, dtype=torch.float32) * float(w * (w + 1) // 2)
            if mean:
                expect = expect / float(w)
            ok = bool(torch.equal(v.cpu(), expect)) and launched and ok
        bad, err = C.c_uint32(), C.c_uint32()
        # ... and the gradient channel itself: both buffer parities, 16-byte system-scope loads of every peer's vector
        checked = lib.hp_peer_selfcheck(h, C.byref(bad)) == 0
        ok = ok and checked and bad.value == 0
        ok = lib.hp_peer_status(h, C.byref(err)) == 0 and err.value == 0 and ok
        if not agree(ok):
            self.peer_refused = ("attach-time self-check failed" + (" on another rank" if ok else
                                 f" (gradient channel: {bad.value} mismatching elements, error word 0x{err.value:x})"))
            lib.hp_peer_destroy(h)
            return None
        self.peer, self._native_lib = h, lib
        return h

    def close(self):
        if self.peer is not None:
            self._native_lib.hp_peer_destroy(self.peer)
            self.peer = None
        if self.native is not None:
            self._native_lib.hp_comm_destroy(self.native)
            self.native = None

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
    def _staged(self, t: torch.Tensor):
        """A gloo group moves host memory: device tensors go through a host copy (test / debugging transport)."""
        return t.is_cuda and self._dist.get_backend() == "gloo"

    def allreduce_sum_(self, t: torch.Tensor):
        if self._dist:
            if self._staged(t):
                h = t.cpu()
                self._dist.all_reduce(h, op=self._dist.ReduceOp.SUM)
                t.copy_(h)
            elif not t.is_cuda and self._dist.get_backend() == "nccl":
                d = t.to(f"cuda:{self.device_id if self.device_id is not None else torch.cuda.current_device()}")
                self._dist.all_reduce(d, op=self._dist.ReduceOp.SUM)      # RCCL moves device memory: host scalars go up and back
                t.copy_(d.cpu())
            else:
                self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t

    def allreduce_mean_(self, t: torch.Tensor):
        """normalizer.py:60-64: Allreduce(SUM) then divide by the number of ranks."""
        if self._dist:
            self.allreduce_sum_(t)
            t /= self.world_size
        return t

    def broadcast_(self, t: torch.Tensor, root=0):
        if self._dist:
            if self._staged(t):
                h = t.cpu()
                self._dist.broadcast(h, src=root)
                t.copy_(h)
            else:
                self._dist.broadcast(t, src=root)
        return t

    # ---- zero-copy collectives on library-owned device vectors
    def _view(self, address, n):
        dev = self.device_id if self.device_id is not None else torch.cuda.current_device()
        return torch.as_tensor(_lib.DevicePointer(address, n), device=f"cuda:{dev}")

    def allreduce_sum_device(self, address, n):
        if not self.active:
            return
        if self.native is not None:
            _lib.check(self._native_lib.hp_comm_allreduce_sum_f32(self.native, C.c_void_p(address), int(n)))
        else:
            self.allreduce_sum_(self._view(address, n))

    def allreduce_mean_device(self, address, n):
        if not self.active:
            return
        if self.peer is not None and n <= 1024:
            _lib.check(self._native_lib.hp_peer_allreduce_f32(self.peer, C.c_void_p(address), int(n), 1))
        elif self.native is not None:
            _lib.check(self._native_lib.hp_comm_allreduce_mean_f32(self.native, C.c_void_p(address), int(n)))
        else:
            self.allreduce_mean_(self._view(address, n))

    def broadcast_device(self, address, n, root=0):
        if not self.active:
            return
        if self.native is not None:
            _lib.check(self._native_lib.hp_comm_broadcast_f32(self.native, C.c_void_p(address), int(n), int(root)))
        else:
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
