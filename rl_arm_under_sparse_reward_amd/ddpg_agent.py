"""ddpg_agent -- the learner half of the reference's ddpg_agent.py on the MI355X.

Same wiring as ddpg_agent.py:18-53 (actor/critic + targets, Adam x2, her_sampler,
replay_buffer, two normalizers, optional demo preload) and the same method names for the
pieces on the hot path:

    _update_network()                 ddpg_agent.py:225-277   one sample + DDPG update
    _soft_update_target_network()     :220-222 (both nets at once, as :149-150 calls it twice)
    _update_normalizer(episode_batch) :187-212
    _preproc_og(o, g)                 :214-217
    _init_demo_buffer()               :82-90
    save_checkpoint()                 :158-161  (same 5-element list, same state_dict keys)

plus `train_cycle(episode_batch)`: lines :143-150 (store -> normalizer -> n_batches updates ->
polyak) as ONE cached hipGraph launch.  The rollout half (`learn`, :92-161) is host-side glue
around any gym-GoalEnv-like object and lives in `learn()`.

Data never leaves the device between sampling and the optimizer step; the methods that
return numpy (`buffer.sample`, normalizer attributes, `state_dict`) copy out on demand.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from datetime import datetime

import numpy as np
import torch

from . import _lib
from . import random as _random
from .her import her_sampler
from .models import actor, critic
from .normalizer import normalizer
from .replay_buffer import replay_buffer
from .utils import Communicator

NET_ACTOR, NET_CRITIC, NET_ACTOR_TARGET, NET_CRITIC_TARGET = 0, 1, 2, 3


class ddpg_agent:
    def __init__(self, args, env, env_params, comm: Communicator | None = None, ctx=None, rng=None):
        self.savetime = 0
        self.args = args
        # one environment, or a list of them: the rollout / evaluation loops step a list in lockstep with one batched
        # policy call per timestep (the host feeder of SURVEY 8f N1)
        self.envs = list(env) if isinstance(env, (list, tuple)) else [env]
        self.env = self.envs[0]
        self.env_params = env_params
        self.ctx = ctx or _lib.Context.default()
        self.lib = self.ctx.lib
        self.comm = comm or Communicator(self.ctx.device_id)
        if rng is None:
            # the reference draws from the process-global np.random, which train.py:36 seeds with seed + rank before it
            # builds the agent.  The device twin of that stream is seeded the same way unless the launch script already
            # did it (random.seed / random.set_state), so ranks never share a sampler stream by accident
            rng = _random.global_state()
            if not rng.seeded:
                rng.seed(int(args.seed) + self.comm.rank)
        self.rng = rng
        # networks: host containers initialised like the reference (consumes the torch RNG identically)
        self.actor_network = actor(env_params)
        self.critic_network = critic(env_params)
        self.actor_target_network = actor(env_params)       # re-initialised from the online nets below
        self.critic_target_network = critic(env_params)
        cfg = _lib.AgentCfg(
            obs_dim=env_params['obs'], goal_dim=env_params['goal'], act_dim=env_params['action'], hidden=256,
            batch=int(args.batch_size), grad_world_size=self.comm.world_size,
            max_action=float(env_params['action_max']), gamma=float(args.gamma), action_l2=float(args.action_l2),
            lr_actor=float(args.lr_actor), lr_critic=float(args.lr_critic), polyak=float(args.polyak),
            clip_obs=float(args.clip_obs), clip_range=float(args.clip_range),
            adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8)
        self.h = C.c_void_p()
        _lib.check(self.lib.hp_agent_create(self.ctx.h, C.byref(cfg), C.byref(self.h)))
        for slot, net in ((NET_ACTOR, self.actor_network), (NET_CRITIC, self.critic_network),
                          (NET_ACTOR_TARGET, self.actor_target_network),
                          (NET_CRITIC_TARGET, self.critic_target_network)):
            net.attach(self, slot)
        self.actor_network.push()
        self.critic_network.push()
        # data-parallel ranks on GPUs: the library exchanges gradients and normalizer sums itself, inside the cycle graph.
        # First choice: one-shot all-reduce over peer memory fused with the optimizer (csrc/peer.hip); else RCCL
        # (csrc/comm.hip); else torch.distributed from a host-driven loop.
        self._native_comm = None
        self._peer = None
        if self.comm.active:
            n = C.c_int64()
            p = C.c_void_p()
            _lib.check(self.lib.hp_agent_grad_buffer(self.h, C.byref(p), C.byref(n)))
            self._peer = self.comm.attach_peer(self.ctx, n.value)
            if self._peer is not None:
                # e.g. the layer-per-launch engine cannot use it.  The ranks decide TOGETHER, and an exchange nobody uses is
                # destroyed (not merely forgotten): `comm.peer` must not outlive it, or the fallback below would take the
                # peer start-up path while its gradients go through another transport
                ok = self.lib.hp_agent_set_peer(self.h, self._peer) == 0
                if not self.comm.agree(ok, self.ctx):
                    _lib.check(self.lib.hp_agent_set_peer(self.h, None))
                    self.comm.drop_peer()
                    self._peer = None
        if self._peer is None:
            self._native_comm = self.comm.attach_native(self.ctx)
            if self._native_comm is not None:
                _lib.check(self.lib.hp_agent_set_comm(self.h, self._native_comm))
        self._grad_mean = str(getattr(args, "grad_reduce", "sum")).lower() == "mean"
        if getattr(args, "grad_reduce", "sum") not in ("sum", "mean"):
            raise ValueError("grad_reduce must be 'sum' (reference semantics) or 'mean'")
        _lib.check(self.lib.hp_agent_set_grad_reduce(self.h, int(self._grad_mean)))
        self._broadcast_params(self.comm)                       # sync_networks x2 (ddpg_agent.py:27-28)
        _lib.check(self.lib.hp_agent_sync_targets(self.h))      # targets := online (ddpg_agent.py:33-34)
        # her sampler + replay buffer (ddpg_agent.py:45-47)
        reward_func = getattr(self.env, "compute_reward", None)
        self.her_module = her_sampler(args.replay_strategy, args.replay_k, reward_func,
                                      distance_threshold=None if reward_func is not None and hasattr(
                                          getattr(reward_func, "__self__", None), "distance_threshold")
                                      else getattr(args, "distance_threshold", 0.05),
                                      reward_type=None if reward_func is not None and hasattr(
                                          getattr(reward_func, "__self__", None), "reward_type")
                                      else getattr(args, "reward_type", "sparse"), rng=self.rng,
                                      goal_dim=env_params['goal'])
        self.buffer = replay_buffer(self.env_params, self.args.buffer_size, self.her_module.sample_her_transitions,
                                    rng=self.rng, ctx=self.ctx)
        if getattr(self.args, "add_demo", False):
            self._init_demo_buffer()
        # normalizers (ddpg_agent.py:52-53)
        self.o_norm = normalizer(size=env_params['obs'], default_clip_range=self.args.clip_range, ctx=self.ctx,
                                 comm=self.comm)
        self.g_norm = normalizer(size=env_params['goal'], default_clip_range=self.args.clip_range, ctx=self.ctx,
                                 comm=self.comm)
        self._norm_stage = None          # staging buffer of _update_normalizer(episode_batch)
        import threading
        self._pending_updates = 0              # argument-less _update_network() calls not issued yet (_lib.py: deferred updates)
        self._pending_parked = False           # their last flush failed: not retried until the caller asks (see _flush_updates)
        self._defer_updates = os.environ.get("RLARM_DEFER_UPDATES", "1") != "0"
        self._stage_lock = threading.RLock()   # orders "store a wave, then sample ITS staged episodes" across feeder threads
        self.success_rates = []
        self.model_path = os.path.join(self.args.save_dir, self.args.env_name)

    def close_comm(self):
        """Detach and destroy the library-side RCCL communicator (call on every rank before
        torch.distributed.destroy_process_group / interpreter exit)."""
        self._flush_updates()         # deferred updates still exchange gradients: issue them while the transport exists
        if self._native_comm is not None or self._peer is not None or self.comm.peer is not None or self.comm.native is not None:
            self.ctx.synchronize()
            _lib.check(self.lib.hp_agent_set_comm(self.h, None))
            _lib.check(self.lib.hp_agent_set_peer(self.h, None))
            self._native_comm = None
            self._peer = None
            self.comm.close()

    def check_exchange(self):
        """Raise if the peer-memory exchange died (a rank late by more than RLARM_PEER_TIMEOUT_S, or gone): the optimizer
        steps from that update on were skipped, so the replicas are no longer in step.  train_cycle / _update_network raise
        the same at their next call without synchronising; learn() asks at every epoch boundary."""
        if self._peer is not None:
            err = C.c_uint32()
            _lib.check(self.lib.hp_peer_status(self._peer, C.byref(err)))
            if err.value:
                raise RuntimeError("peer-memory exchange timed out: a rank was late by more than RLARM_PEER_TIMEOUT_S or is "
                                   "gone; updates were skipped on this rank and the replicas are no longer in step")

    # ------------------------------------------------------------------ parameter plumbing
    def _count(self, slot):
        return int(self.lib.hp_agent_param_count(self.h, slot))

    def _get_flat(self, slot):
        out = np.empty(self._count(slot), np.float32)
        _lib.check(self.lib.hp_agent_get_params(self.h, slot, _lib.ptr(out, C.c_float), out.size))
        return out

    def _set_flat(self, slot, flat):
        flat = _lib.as_f32(flat)
        _lib.check(self.lib.hp_agent_set_params(self.h, slot, _lib.ptr(flat, C.c_float), flat.size))

    def get_flat_grads(self, slot):
        out = np.empty(self._count(slot), np.float32)
        _lib.check(self.lib.hp_agent_get_grads(self.h, slot, _lib.ptr(out, C.c_float), out.size))
        return out

    def get_adam_state(self, slot):
        n = self._count(slot)
        m, v, step = np.empty(n, np.float32), np.empty(n, np.float32), C.c_int64()
        _lib.check(self.lib.hp_agent_get_adam(self.h, slot, _lib.ptr(m, C.c_float), _lib.ptr(v, C.c_float), n,
                                              C.byref(step)))
        return m, v, step.value

    def set_adam_state(self, slot, m, v, step):
        """Load optimizer state (test hook, hp_agent_set_adam): torch.optim.Adam's exp_avg / exp_avg_sq in the flat order of
        utils.py:18-27 and the number of steps already taken (both optimizers step together)."""
        m, v = _lib.as_f32(m), _lib.as_f32(v)
        _lib.check(self.lib.hp_agent_set_adam(self.h, slot, _lib.ptr(m, C.c_float), _lib.ptr(v, C.c_float), m.size,
                                              int(step)))

    def _broadcast_params(self, comm):
        if not comm.active:
            return
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(self.lib.hp_agent_param_buffer(self.h, C.byref(p), C.byref(n)))
        if comm.native is None and self._peer is not None and comm.peer is not None:
            # the library exchanges gradients itself over peer memory: only this one-off broadcast goes through
            # torch.distributed; the library keeps its own (capturable) stream, so order the two sides with full
            # synchronisations instead of moving the library onto torch's stream
            self.ctx.synchronize()
            comm.broadcast_device(p.value, n.value, 0)
            torch.cuda.synchronize(self.ctx.device_id)
            return
        if comm.native is None:
            # every exchange of this run goes through torch.distributed (host-driven loop): the library's kernels must be
            # stream-ordered with the collectives torch enqueues
            self.ctx.use_torch_stream()
        comm.broadcast_device(p.value, n.value, 0)

    def _allreduce_grads(self, comm):
        if not comm.active:
            return
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(self.lib.hp_agent_grad_buffer(self.h, C.byref(p), C.byref(n)))
        if self._grad_mean:
            comm.allreduce_mean_device(p.value, n.value)
        else:
            comm.allreduce_sum_device(p.value, n.value)                  # utils.py:47

    def _actor_forward(self, slot, x):
        x = _lib.as_f32(x)
        x2 = x.reshape(-1, x.shape[-1])
        out = np.empty((x2.shape[0], self.env_params['action']), np.float32)
        _lib.check(self.lib.hp_agent_actor_forward(self.h, slot, _lib.ptr(x2, C.c_float), x2.shape[0],
                                                   _lib.ptr(out, C.c_float)))
        return out.reshape(x.shape[:-1] + (out.shape[-1],))

    def _critic_forward(self, slot, x, actions):
        x, actions = _lib.as_f32(x), _lib.as_f32(actions)
        x2, a2 = x.reshape(-1, x.shape[-1]), actions.reshape(-1, actions.shape[-1])
        if x2.shape[0] != a2.shape[0]:
            raise ValueError("critic forward: inputs and actions differ in length")
        out = np.empty((x2.shape[0], 1), np.float32)
        _lib.check(self.lib.hp_agent_critic_forward(self.h, slot, _lib.ptr(x2, C.c_float), _lib.ptr(a2, C.c_float),
                                                    x2.shape[0], _lib.ptr(out, C.c_float)))
        return out.reshape(x.shape[:-1] + (1,))

    # ------------------------------------------------------------------ hot path
    def _handles(self):
        return (self.h, self.buffer._dev.h, self.o_norm.h, self.g_norm.h, self.rng.h)

    def _update_network(self, n_updates=None):
        """ddpg_agent.py:225-277.  `_update_network(n)`: n updates back to back, issued now.  `_update_network()` -- the
        reference's call form, once per minibatch in its inner loop (:145-147) -- is only counted and issued together with the
        calls that follow it, as soon as anything else touches the library or `n_batches` of them have come (_lib.py,
        "deferred updates"): the same updates bit for bit, 40.5 instead of 51 us each at batch 256."""
        if n_updates is None:
            if self._defer_updates and (not self.comm.active or self._native_comm is not None or self._peer is not None):
                with _lib.pending_lock:
                    if self._pending_updates == 0:      # where the first of these deferred calls was made: named if their flush fails
                        import sys
                        f = sys._getframe(1)
                        self._pending_site = f"{f.f_code.co_filename}:{f.f_lineno}"
                    self._pending_updates += 1
                    self._pending_parked = False            # a new update call is the natural point to try the owed ones again
                    if self._pending_updates >= int(self.args.n_batches):
                        self._flush_updates()
                    else:
                        _lib.register_pending(self)
                return
            n_updates = 1
        with _lib.pending_lock:
            # owed updates that a failed flush parked come FIRST, like in the reference's sequential loop: an explicit call is a
            # point to try them again, exactly as the argument-less form above does (if they fail again this call raises and
            # issues nothing of its own)
            self._pending_parked = False
        self._flush_updates()
        self._issue_updates(int(n_updates))

    def _flush_updates(self):
        """Issue the counted `_update_network()` calls.  In chunks whose lengths are powers of two (at most n_batches): a loop
        that is flushed at irregular points would otherwise ask for a new sequence length -- a freshly captured graph -- every
        time and thrash the library's 8-entry graph cache.

        A failing flush raises ONCE, like the reference's `_update_network()` does at the same cause (e.g. `ValueError: high <=
        0` on an empty buffer, ddpg_agent.py:227): the error names the deferred call it belongs to, what could not be issued
        stays owed (`pending_updates`) but is PARKED -- no later library call retries it by itself, so every object stays usable
        and the caller can remove the cause (e.g. store an episode).  The owed updates are tried again by the next
        `_update_network()` / `_update_network(n)` call (in front of that call's own updates: the order of updates never
        changes), by `retry_pending_updates()`, or dropped by `discard_pending_updates()`.  Every OTHER caller of this method
        (train_cycle, weight reads, close_comm) leaves parked updates parked."""
        with _lib.pending_lock:
            if self._pending_parked:
                return
            n, self._pending_updates = self._pending_updates, 0
            _lib.unregister_pending(self)
            cap = max(1, int(self.args.n_batches))
            while n:
                chunk = n if n == cap else 1 << (min(n, cap).bit_length() - 1)
                try:
                    self._issue_updates(chunk)
                except Exception as e:
                    self._pending_updates += n                 # nothing of this chunk was enqueued: still owed ...
                    self._pending_parked = True                # ... but not retried behind the caller's back
                    note = (f"deferred _update_network() x {n}, first called at {getattr(self, '_pending_site', '?')} "
                            "(issued by a later library call, _lib.py 'deferred updates')")
                    if hasattr(e, "add_note"):
                        e.add_note(note)
                        raise
                    raise type(e)(f"{e} [{note}]") from e
                n -= chunk

    @property
    def pending_updates(self):
        """Argument-less `_update_network()` calls counted but not issued yet (parked ones included)."""
        return self._pending_updates

    def retry_pending_updates(self):
        """Issue the owed updates now (after a failed flush parked them and its cause has been removed).  Raises like the
        flush did if they fail again."""
        with _lib.pending_lock:
            self._pending_parked = False
            self._flush_updates()

    def discard_pending_updates(self):
        """Give up on the owed updates (what a caller that catches the reference's error and moves on does); returns how many."""
        with _lib.pending_lock:
            n, self._pending_updates, self._pending_parked = self._pending_updates, 0, False
            _lib.unregister_pending(self)
            return n

    def _issue_updates(self, n_updates):
        fp, sq = float(self.her_module.future_p), float(self.her_module.sq_threshold)
        if not self.comm.active or self._native_comm is not None or self._peer is not None:
            # single rank, or the library exchanges the gradients itself between backward and Adam
            _lib.check(self.lib.hp_agent_sample_and_update(*self._handles(), fp, sq, int(n_updates)))
            return
        for _ in range(int(n_updates)):          # data-parallel ranks: grads are SUMmed between backward and Adam
            _lib.check(self.lib.hp_agent_forward_backward(*self._handles(), fp, sq))
            self._allreduce_grads(self.comm)
            _lib.check(self.lib.hp_agent_apply(self.h))

    def update_on_minibatch(self, x, x_next, actions, r):
        """One update on a caller-supplied normalised minibatch (parity hook); returns (actor_loss, critic_loss)."""
        x, xn, a = _lib.as_f32(x), _lib.as_f32(x_next), _lib.as_f32(actions)
        r = _lib.as_f32(r).reshape(-1)
        B = int(self.args.batch_size)
        if x.shape[0] != B or xn.shape != x.shape or a.shape[0] != B or r.shape[0] != B:
            raise ValueError("minibatch shapes do not match args.batch_size")
        losses = np.empty(2, np.float32)
        f = C.c_float
        _lib.check(self.lib.hp_agent_update_minibatch(self.h, _lib.ptr(x, f), _lib.ptr(xn, f), _lib.ptr(a, f),
                                                      _lib.ptr(r, f), _lib.ptr(losses, f)))
        return float(losses[0]), float(losses[1])

    def last_losses(self, n=1):
        """(actor_loss, critic_loss) of the n most recent updates, oldest first.  Synchronises."""
        out = np.empty(2 * n, np.float32)
        _lib.check(self.lib.hp_agent_get_losses(self.h, _lib.ptr(out, C.c_float), n))
        return out.reshape(n, 2)

    def _soft_update_target_network(self, target=None, source=None):
        """ddpg_agent.py:220-222.  The reference calls it once per net (:149-150); the device pass covers
        both nets, so it acts when called for the actor pair (or with no arguments) and is a no-op for the
        critic pair."""
        if target is None or target is self.actor_target_network:
            _lib.check(self.lib.hp_agent_soft_update(self.h))

    def _preproc_og(self, o, g):
        o = np.clip(o, -self.args.clip_obs, self.args.clip_obs)
        g = np.clip(g, -self.args.clip_obs, self.args.clip_obs)
        return o, g

    def _update_normalizer(self, episode_batch=None):
        """ddpg_agent.py:187-212: HER-sample T transitions out of `episode_batch`, clip, update both normalizers,
        recompute.  The given episodes are uploaded into a private staging buffer (the reference's `buffer_temp`), so
        the statistics come from exactly these episodes whatever was stored in between (e.g. by a feeder thread).
        `episode_batch=None` reuses the episodes the replay buffer staged in its most recent store_episode (no upload;
        the benchmark's cycle boundary) and raises if nothing was ever stored."""
        fp = float(self.her_module.future_p)
        if episode_batch is not None:
            if self._norm_stage is None:
                from .replay_buffer import DeviceEpisodeBuffer
                ep = self.env_params
                self._norm_stage = DeviceEpisodeBuffer(1, ep['max_timesteps'], ep['obs'], ep['goal'], ep['action'],
                                                       ctx=self.ctx)
            self._norm_stage.stage(episode_batch)
            src = self._norm_stage
        else:
            src = self.buffer._dev
        _lib.check(self.lib.hp_norm_update_from_staged(src.h, self.rng.h, self.o_norm.h, self.g_norm.h, fp,
                                                       float(self.args.clip_obs)))
        self.o_norm.recompute_stats()
        self.g_norm.recompute_stats()

    def train_cycle(self, episode_batch, n_batches=None):
        """ddpg_agent.py:143-150 as one hipGraph: store_episode, _update_normalizer, n_batches x
        _update_network, soft update of both targets.  Asynchronous.  With data-parallel ranks the graph
        contains the RCCL all-reduces (gradients every update, normalizer sums once) when the library owns the
        communicator; otherwise the loop is driven from the host."""
        n_batches = int(n_batches or self.args.n_batches)
        if self.comm.active and self._native_comm is None and self._peer is None:   # collectives on torch.distributed: host-driven loop
            self.buffer.store_episode(episode_batch)
            self._update_normalizer(episode_batch)
            self._update_network(n_batches)
            self._soft_update_target_network()
            return
        # same shape validation as store_episode (a wrong T or dimension raises ValueError like numpy's broadcast at
        # replay_buffer.py:39-42 instead of letting the library read past the arrays)
        obs, ag, g, act, n_new = self.buffer._dev._checked(episode_batch)
        d = C.c_double
        _lib.check(self.lib.hp_agent_train_cycle(
            *self._handles(), _lib.ptr(obs, d), _lib.ptr(ag, d), _lib.ptr(g, d), _lib.ptr(act, d), n_new,
            float(self.her_module.future_p), float(self.her_module.sq_threshold), n_batches))

    def train_cycle_from_feeder(self, feeder, slot, n_batches=None):
        """ddpg_agent.py:143-150 on the episodes of a feeder wave: store_episode is an asynchronous DMA out of the feeder's
        registered shared-memory slot (no CPU copy), the normalizer samples the staged episodes, the updates replay their
        cached graph.  Asynchronous; ranks of a data-parallel group call it in step like train_cycle."""
        n_batches = int(n_batches or self.args.n_batches)
        if not self.comm.active or self._native_comm is not None or self._peer is not None:
            # the whole cycle as one cached graph behind the DMA (hp_agent_train_cycle_pinned): same launches as train_cycle
            with self._stage_lock:
                feeder.train_cycle_wave(slot, n_batches)
            return
        with self._stage_lock:       # store + normalizer update as one unit: another thread's store_wave (e.g. a rollout
            feeder.store_wave(slot)  # thread) must not replace the staged episodes in between
            self._update_normalizer()
        self._update_network(n_batches)
        self._soft_update_target_network()

    def update_kernels(self, n_updates=None):
        """The kernels a sequence of `n_updates` updates enqueues on this agent as it is NOW -- engine, switches, attached
        transport -- read off the library's own launch logic (hp_agent_update_kernels: the logic runs under a discarded stream
        capture, nothing executes).  {'open': [...], 'prologue': [...], 'updates': [[...] per update], 'close': [...]}."""
        self._flush_updates()
        n = int(n_updates or self.args.n_batches)
        host_loop = self.comm.active and self._native_comm is None and self._peer is None
        buf = C.create_string_buffer(1 << 16)
        _lib.check(self.lib.hp_agent_update_kernels(*self._handles(), float(self.her_module.future_p),
                                                    float(self.her_module.sq_threshold), n, 1 if host_loop else 0, buf, len(buf)))
        out, cur = {"open": [], "prologue": [], "updates": [], "close": []}, None
        for tok in buf.value.decode().split(","):
            if tok == "#update":
                out["updates"].append([])
                cur = out["updates"][-1]
            elif tok.startswith("#"):
                cur = out[tok[1:]]
            elif tok:
                cur.append(tok)
        return out

    def engine(self):
        """Kernels this agent's updates run (hp_agent_engine): e.g. {'engine': 'slab8', 'slab_rows': 4, 'weight_grad':
        'gemm_lds 32x32'} at the reference batch, {'engine': 'slab32', 'slab_rows': 32, 'weight_grad': 'dw64 split 3'} at 4096.
        `kernels_per_update`: the launches of one steady-state update of an n_batches sequence, by name, as the library
        enqueues them with the transport attached NOW (ask before close_comm)."""
        e, r, d, f = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.hp_agent_engine(self.h, C.byref(e), C.byref(r), C.byref(d)))
        _lib.check(self.lib.hp_agent_update_form(self.h, int(self.args.n_batches), C.byref(f)))
        out = {"engine": {0: "layers", 8: "slab8", 32: "slab32"}[e.value], "slab_rows": r.value,
               "weight_grad": f"dw64 split {d.value}" if d.value else "gemm_lds 32x32",
               "launches_per_update": {0: "chains | weight gradients (+ Adam, or exchange + Adam)",
                                       1: "split: chains (targets one update ahead) + critic weight gradients (+ exchange + Adam) | "
                                          "actor weight gradients (+ exchange + Adam)"}[f.value]}
        try:
            k = self.update_kernels()
            ups = k["updates"]
            out["kernels_per_update"] = ups[min(2, len(ups) - 1)]
            out["kernels_per_sequence_extra"] = k["open"] + k["prologue"] + k["close"]
        except _lib.HpError as err:      # e.g. the legacy stream cannot be captured: the form above still stands
            out["kernels_per_update"] = None
            out["kernels_error"] = str(err)
        return out

    def policy_snapshot(self):
        """Publish the current actor + normalizer statistics to feeders that call the policy while cycles run
        (hp_agent_act_snapshot): stream-ordered with the updates, no host wait."""
        _lib.check(self.lib.hp_agent_policy_snapshot(self.h, self.o_norm.h, self.g_norm.h))

    # ------------------------------------------------------------------ demos / checkpoints (formats preserved)
    def plot_success_rate(self):
        """ddpg_agent.py:73-80: dump the per-epoch success rates where the reference does; the plot itself needs
        matplotlib and a display, so it is drawn only when both are there."""
        saved_dir = 'test_rates/'
        os.makedirs(saved_dir, exist_ok=True)
        np.save(saved_dir + str(self.args.seed) + '_' + str(self.args.add_demo) + '_success_rates.npy',
                np.array(self.success_rates))
        try:
            import matplotlib.pyplot as plt
        except Exception:
            return
        plt.plot(self.success_rates)
        if os.environ.get("DISPLAY") or os.environ.get("MPLBACKEND"):
            plt.show()

    def _init_demo_buffer(self):
        """ddpg_agent.py:82-90: keys obs, acs, ag, g of a get_demo_data_*.py file (info is ignored)."""
        demo = np.load(self.args.demo_name, allow_pickle=True)
        self.buffer.store_episode([np.array(demo['obs']), np.array(demo['ag']), np.array(demo['g']),
                                   np.array(demo['acs'])])

    def checkpoint_payload(self):
        """The 5-element list the reference saves (ddpg_agent.py:158-161)."""
        return [self.o_norm.mean, self.o_norm.std, self.g_norm.mean, self.g_norm.std,
                self.actor_network.state_dict()]

    def save_checkpoint(self, path=None):
        if path is None:
            os.makedirs(self.model_path, exist_ok=True)
            self.savetime += 1
            path = os.path.join(self.model_path, f"{self.args.seed}_{self.args.add_demo}{self.savetime}_model.pt")
        torch.save(self.checkpoint_payload(), path)
        return path

    def load_checkpoint(self, path):
        """Warm start from a reference-format checkpoint (the block commented out at ddpg_agent.py:54-62)."""
        o_mean, o_std, g_mean, g_std, model = torch.load(path, map_location="cpu", weights_only=False)
        self.actor_network.load_state_dict(model)
        self.o_norm.set_stats(o_mean, o_std)
        self.g_norm.set_stats(g_mean, g_std)

    # ------------------------------------------------------------------ rollout side (host glue)
    def _preproc_inputs(self, obs, g):
        inputs = np.concatenate([self.o_norm.normalize(obs), self.g_norm.normalize(g)])
        return torch.tensor(inputs, dtype=torch.float32).unsqueeze(0)

    def act(self, obs, g, target=False, clip_obs=0.0):
        """_preproc_inputs (:163-171) + actor (:114-116) for a stack of environments in ONE device call: obs [n, obs] and
        g [n, goal] float64 (or single rows) -> actions [n, action] float32.  clip_obs > 0 also clips the raw values to
        +-clip_obs first, which is what the checkpoint reader demo_push.py:15-22 does (the rollouts do not)."""
        obs, g = _lib.as_f64(obs), _lib.as_f64(g)
        o2, g2 = obs.reshape(-1, obs.shape[-1]), g.reshape(-1, g.shape[-1])
        if o2.shape[0] != g2.shape[0]:
            raise ValueError("act: observation and goal stacks differ in length")
        out = np.empty((o2.shape[0], self.env_params['action']), np.float32)
        d = C.c_double
        _lib.check(self.lib.hp_agent_act(self.h, self.o_norm.h, self.g_norm.h, NET_ACTOR_TARGET if target else NET_ACTOR,
                                         _lib.ptr(o2, d), _lib.ptr(g2, d), o2.shape[0], float(clip_obs),
                                         _lib.ptr(out, C.c_float)))
        return out.reshape(obs.shape[:-1] + (out.shape[-1],))

    def _select_actions(self, pi):
        """ddpg_agent.py:174-184: Gaussian noise, clip, epsilon-random (numpy global RNG, like the reference).  `action`
        stays the float32 array the policy returned and is updated in place, so every step rounds to float32 exactly
        where the reference's `action += ...` does."""
        amax = self.env_params['action_max']
        action = (pi.cpu().numpy() if isinstance(pi, torch.Tensor) else np.array(pi, dtype=np.float32)).squeeze()
        action += self.args.noise_eps * amax * np.random.randn(*action.shape)
        action = np.clip(action, -amax, amax)
        random_actions = np.random.uniform(low=-amax, high=amax, size=self.env_params['action'])
        action += np.random.binomial(1, self.args.random_eps, 1)[0] * (random_actions - action)
        return action

    def collect_episodes(self, n_rollouts, epoch=0, explore=True):
        """Rollout half of learn() (:101-137) for `n_rollouts` episodes; returns the four episode arrays.  The
        environments in self.envs are stepped in lockstep, one batched policy call per timestep; with a single
        environment this is the reference's loop, including the order of its draws from numpy's global stream."""
        T = int(self.env_params['max_timesteps'])
        mb = ([], [], [], [])
        done = 0
        while done < n_rollouts:
            envs = self.envs[:n_rollouts - done]
            k = len(envs)
            first = [env.reset() for env in envs]
            obs = [o['observation'] for o in first]
            ag = [o['achieved_goal'] for o in first]
            g = [o['desired_goal'] for o in first]
            ep = [([], [], [], []) for _ in envs]
            for _t in range(T):
                pi = self.act(np.stack(obs), np.stack(g))
                for i, env in enumerate(envs):
                    action = self._select_actions(pi[i]) if explore else pi[i].astype(np.float64)
                    if epoch >= 100:
                        action = np.clip(action, -0.15, 0.15)            # ddpg_agent.py:118-119
                    observation_new, _, _, info = env.step(action)
                    for dst, v in zip(ep[i], (obs[i], ag[i], g[i], action)):
                        dst.append(np.array(v, dtype=np.float64))
                    obs[i], ag[i] = observation_new['observation'], observation_new['achieved_goal']
            for i in range(k):
                ep[i][0].append(np.array(obs[i], dtype=np.float64)); ep[i][1].append(np.array(ag[i], dtype=np.float64))
                for dst, src in zip(mb, ep[i]):
                    dst.append(src)
            done += k
        return [np.array(a) for a in mb]

    def learn(self):
        """ddpg_agent.py:92-161 with the learner half on the device.

        The reference draws exploration noise (:177-183), overflow slots (replay_buffer.py:64,67) and HER indices
        (her.py:24-31) from ONE stream, numpy's global one.  With `args.share_numpy_stream` (default) that single stream
        is handed to the device for the learner phase of every cycle and taken back before the next rollout, so a run
        seeded like train.py:34-39 consumes exactly the reference's random words in the reference's order (with one
        environment; a list of environments is stepped in lockstep, which reorders the exploration draws).  The
        hand-back synchronises, which the next rollout needs anyway: it evaluates the updated actor."""
        share = bool(getattr(self.args, "share_numpy_stream", True))
        print("initial buffer size:", self.buffer.current_size)                  # :97
        for epoch in range(self.args.n_epochs):
            start = time.time()
            for _ in range(self.args.n_cycles):
                episodes = self.collect_episodes(self.args.num_rollouts_per_mpi, epoch)
                if share:
                    self.rng.set_state(np.random.get_state())
                self.train_cycle(episodes)
                if share:
                    np.random.set_state(self.rng.get_state())
            self.ctx.synchronize()
            self.check_exchange()
            print(str(time.time() - start))
            rate = self._eval_agent()
            self.success_rates.append(rate)
            if self.comm.rank == 0:
                print('[{}] epoch is: {}, eval success rate is: {:.3f}'.format(datetime.now(), epoch, rate))
                self.save_checkpoint()

    def _eval_agent(self):
        """ddpg_agent.py:280-304: success at the last step of n_test_rollouts noise-free episodes, averaged over ranks;
        the environments in self.envs run in lockstep with one batched policy call per timestep."""
        wins = []
        remaining = int(self.args.n_test_rollouts)
        while remaining > 0:
            envs = self.envs[:remaining]
            first = [env.reset() for env in envs]
            obs = [o['observation'] for o in first]
            g = [o['desired_goal'] for o in first]
            last = [0.0] * len(envs)
            for _t in range(int(self.env_params['max_timesteps'])):
                actions = self.act(np.stack(obs), np.stack(g))
                for i, env in enumerate(envs):
                    observation_new, _, _, info = env.step(actions[i])
                    obs[i], g[i] = observation_new['observation'], observation_new['desired_goal']
                    last[i] = float(info.get('is_success', last[i]))
            wins.extend(last)
            remaining -= len(envs)
        local = torch.tensor([float(np.mean(wins))], dtype=torch.float64)
        if self.comm.world_size > 1:
            local = local.to(f"cuda:{self.ctx.device_id}")
        self.comm.allreduce_mean_(local)
        return float(local.item())

    def __del__(self):
        try:
            self._flush_updates()     # counted `_update_network()` calls are owed even if nobody looks at the result
        except Exception:
            pass
        try:
            _lib.unregister_pending(self)
            self.lib.hp_agent_destroy(self.h)
        except Exception:
            pass
