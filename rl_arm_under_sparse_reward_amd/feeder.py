"""Host experience feeder for many environments (BASELINE config 5: 64 environments feeding one shard).

The reference collects experience with ONE environment per MPI rank inside learn() (ddpg_agent.py:101-142): reset, then
T times {normalise, actor forward at batch 1, exploration noise, env.step}, and hands the episode arrays to
store_episode.  Here the same loop runs for K environments at once:

  * the environments live in `n_workers` worker PROCESSES (PyBullet is single-threaded and holds the GIL), each stepping
    its share of the environments and applying the reference's exploration (_select_actions :174-184, the +-0.15 clip
    from epoch 100 :118-119) with its own RandomState(seed + worker), like the reference's per-rank seeds (train.py:36);
  * per timestep the trainer process makes ONE batched policy call for all K rows (hp_agent_act, or
    hp_agent_act_snapshot on a second stream when collection overlaps training);
  * workers write every step straight into a slot of a shared-memory staging ring laid out like the library's staging
    block ([obs K x (T+1) x obs | ag | g | actions], float64); the trainer registered the ring with the device
    (hp_host_register), so store_episode of a finished wave is one asynchronous DMA out of that slot with no CPU copy
    (hp_buffer_store_pinned); the slot is reused once its ticket is done.

Workers import numpy and the environment only (never torch or the HIP library).
"""
from __future__ import annotations

import ctypes as C
import multiprocessing as mp
from multiprocessing import shared_memory

import numpy as np

CMD_STEP, CMD_RESET, CMD_EXIT = 0, 1, 2


class _Layout:
    """Offsets (in float64 elements) inside one staging slot and inside the 'current observation' board."""

    def __init__(self, n_envs, T, obs, goal, act):
        self.n, self.T, self.obs, self.goal, self.act = n_envs, T, obs, goal, act
        self.o_obs = 0
        self.o_ag = self.o_obs + n_envs * (T + 1) * obs
        self.o_g = self.o_ag + n_envs * (T + 1) * goal
        self.o_act = self.o_g + n_envs * T * goal
        self.slot_elems = self.o_act + n_envs * T * act
        self.now_elems = n_envs * (obs + 2 * goal)

    def slot_views(self, buf, slot):
        base = np.ndarray((self.slot_elems,), np.float64, buffer=buf, offset=slot * self.slot_elems * 8)
        n, T = self.n, self.T
        return (base[self.o_obs:self.o_ag].reshape(n, T + 1, self.obs), base[self.o_ag:self.o_g].reshape(n, T + 1, self.goal),
                base[self.o_g:self.o_act].reshape(n, T, self.goal), base[self.o_act:].reshape(n, T, self.act))

    def now_views(self, buf):
        a = np.ndarray((self.now_elems,), np.float64, buffer=buf)
        n = self.n
        o = a[:n * self.obs].reshape(n, self.obs)
        ag = a[n * self.obs:n * (self.obs + self.goal)].reshape(n, self.goal)
        g = a[n * (self.obs + self.goal):].reshape(n, self.goal)
        return o, ag, g


def _worker(w, lo, hi, env_specs, dims, names, go, done, seed, noise_eps, random_eps, action_max):
    """One worker process: environments [lo, hi)."""
    n_envs, T, obs_d, goal_d, act_d, n_slots = dims
    lay = _Layout(n_envs, T, obs_d, goal_d, act_d)
    shm_ring, shm_now, shm_act, shm_ctl = (shared_memory.SharedMemory(name=n) for n in names)
    now_o, now_ag, now_g = lay.now_views(shm_now.buf)
    act_in = np.ndarray((n_envs, act_d), np.float32, buffer=shm_act.buf)
    ctl = np.ndarray((8,), np.int64, buffer=shm_ctl.buf)        # cmd, slot, epoch, t, explore
    envs = [cls(**kw) for cls, kw in env_specs]
    rs = np.random.RandomState(seed + w)
    try:
        while True:
            go.acquire()
            cmd, slot, epoch, t, explore = (int(x) for x in ctl[:5])
            if cmd == CMD_EXIT:
                break
            s_obs, s_ag, s_g, s_act = lay.slot_views(shm_ring.buf, slot)
            if cmd == CMD_RESET:
                for i, env in zip(range(lo, hi), envs):
                    o = env.reset()
                    now_o[i], now_ag[i], now_g[i] = o['observation'], o['achieved_goal'], o['desired_goal']
                    s_obs[i, 0], s_ag[i, 0] = now_o[i], now_ag[i]
            else:
                for i, env in zip(range(lo, hi), envs):
                    action = act_in[i].copy()                   # float32, updated in place like ddpg_agent.py:176-183
                    if explore:
                        action += noise_eps * action_max * rs.randn(*action.shape)
                        action = np.clip(action, -action_max, action_max)
                        random_actions = rs.uniform(low=-action_max, high=action_max, size=act_d)
                        action += rs.binomial(1, random_eps, 1)[0] * (random_actions - action)
                    if epoch >= 100:
                        action = np.clip(action, -0.15, 0.15)   # :118-119
                    o, _, _, _info = env.step(action)
                    s_g[i, t], s_act[i, t] = now_g[i], action
                    now_o[i], now_ag[i] = o['observation'], o['achieved_goal']
                    s_obs[i, t + 1], s_ag[i, t + 1] = now_o[i], now_ag[i]
            done.release()
    finally:
        for s in (shm_ring, shm_now, shm_act, shm_ctl):
            s.close()


class EpisodeFeeder:
    """K environments in worker processes feeding one replay shard through a registered shared-memory ring."""

    def __init__(self, agent, env_specs, n_workers=8, n_slots=3, seed=0, snapshot_policy=False):
        from . import _lib

        self._lib = _lib
        self.agent = agent
        p = agent.env_params
        self.n_envs, self.T = len(env_specs), int(p['max_timesteps'])
        self.n_workers = max(1, min(int(n_workers), self.n_envs))
        self.n_slots = int(n_slots)
        self.lay = _Layout(self.n_envs, self.T, p['obs'], p['goal'], p['action'])
        self.snapshot_policy = bool(snapshot_policy)
        self._shm = [shared_memory.SharedMemory(create=True, size=max(8, sz)) for sz in (
            self.n_slots * self.lay.slot_elems * 8, self.lay.now_elems * 8, self.n_envs * p['action'] * 4, 64)]
        ring, now, act, ctl = self._shm
        self.now_o, self.now_ag, self.now_g = self.lay.now_views(now.buf)
        self.act_in = np.ndarray((self.n_envs, p['action']), np.float32, buffer=act.buf)
        self.ctl = np.ndarray((8,), np.int64, buffer=ctl.buf)
        self.ctl[:] = 0
        # the ring is DMA-able from here on: store_episode of a wave reads the workers' writes in place
        self._ring_addr = C.addressof(C.c_char.from_buffer(ring.buf))
        _lib.check(agent.lib.hp_host_register(agent.ctx.h, C.c_void_p(self._ring_addr), self.n_slots * self.lay.slot_elems * 8))
        self._tickets = [0] * self.n_slots
        self._next_slot = 0
        ctx = mp.get_context("spawn")
        self._go = [ctx.Semaphore(0) for _ in range(self.n_workers)]
        self._done = [ctx.Semaphore(0) for _ in range(self.n_workers)]
        bounds = np.linspace(0, self.n_envs, self.n_workers + 1).astype(int)
        dims = (self.n_envs, self.T, p['obs'], p['goal'], p['action'], self.n_slots)
        names = tuple(s.name for s in self._shm)
        a = agent.args
        self._procs = []
        for w in range(self.n_workers):
            lo, hi = int(bounds[w]), int(bounds[w + 1])
            pr = ctx.Process(target=_worker, daemon=True,
                             args=(w, lo, hi, env_specs[lo:hi], dims, names, self._go[w], self._done[w], int(seed),
                                   float(a.noise_eps), float(a.random_eps), float(p['action_max'])))
            pr.start()
            self._procs.append(pr)
        self.waves = 0

    # ---------------------------------------------------------------- one wave = n_envs episodes
    def _round(self, cmd, slot, epoch, t, explore):
        self.ctl[:5] = (cmd, slot, epoch, t, int(explore))
        for s in self._go:
            s.release()
        for s in self._done:
            s.acquire()

    def _policy(self, obs, g):
        ag = self.agent
        if not self.snapshot_policy:
            return ag.act(obs, g)
        out = np.empty((obs.shape[0], ag.env_params['action']), np.float32)
        d = C.c_double
        self._lib.check(ag.lib.hp_agent_act_snapshot(ag.h, self._lib.ptr(obs, d), self._lib.ptr(g, d), obs.shape[0], 0.0,
                                                     self._lib.ptr(out, C.c_float)))
        return out

    def _wait_slot(self, slot):
        t = self._tickets[slot]
        if t:
            done = C.c_int32()
            self._lib.check(self.agent.lib.hp_buffer_store_done(self.agent.buffer._dev.h, t, 1, C.byref(done)))
            self._tickets[slot] = 0

    def collect_wave(self, epoch=0, explore=True):
        """ddpg_agent.py:105-137 for all environments at once; returns the slot holding the finished episodes."""
        slot = self._next_slot
        self._next_slot = (slot + 1) % self.n_slots
        self._wait_slot(slot)                      # the DMA of the wave that last used this slot has finished
        self._round(CMD_RESET, slot, epoch, 0, explore)
        for t in range(self.T):
            self.act_in[:] = self._policy(np.ascontiguousarray(self.now_o), np.ascontiguousarray(self.now_g))
            self._round(CMD_STEP, slot, epoch, t, explore)
        self.waves += 1
        return slot

    def episodes(self, slot):
        """numpy views [n, T+1, obs], [n, T+1, goal], [n, T, goal], [n, T, action] of a slot (valid until it is reused)."""
        return self.lay.slot_views(self._shm[0].buf, slot)

    def store_wave(self, slot):
        """replay_buffer.store_episode of the slot's episodes: asynchronous DMA straight out of the shared ring."""
        ag = self.agent
        ticket = C.c_uint64()
        with ag._stage_lock:     # the device staging of "the most recent store" is what _update_normalizer(None) samples
            self._lib.check(ag.lib.hp_buffer_store_pinned(ag.buffer._dev.h, ag.rng.h,
                                                          C.c_void_p(self._ring_addr + slot * self.lay.slot_elems * 8),
                                                          self.n_envs, C.byref(ticket)))
        self._tickets[slot] = ticket.value
        return ticket.value

    def train_cycle_wave(self, slot, n_batches):
        """ddpg_agent.py:143-150 on the slot's episodes as ONE cached graph behind the asynchronous DMA out of the shared ring
        (hp_agent_train_cycle_pinned): store, normalizer update, n_batches updates, soft update."""
        ag = self.agent
        ticket = C.c_uint64()
        self._lib.check(ag.lib.hp_agent_train_cycle_pinned(
            *ag._handles(), C.c_void_p(self._ring_addr + slot * self.lay.slot_elems * 8), self.n_envs,
            float(ag.her_module.future_p), float(ag.her_module.sq_threshold), int(n_batches), C.byref(ticket)))
        self._tickets[slot] = ticket.value
        return ticket.value

    def close(self):
        if not self._procs:
            return
        self.ctl[0] = CMD_EXIT
        for s in self._go:
            s.release()
        for pr in self._procs:
            pr.join(timeout=5)
            if pr.is_alive():
                pr.terminate()
        self._procs = []
        for slot in range(self.n_slots):
            try:
                self._wait_slot(slot)
            except Exception:
                pass
        try:
            self.agent.ctx.synchronize()
            self._lib.check(self.agent.lib.hp_host_unregister(self.agent.ctx.h, C.c_void_p(self._ring_addr)))
        finally:
            self.now_o = self.now_ag = self.now_g = self.act_in = self.ctl = None
            for s in self._shm:
                try:
                    s.close()
                    s.unlink()
                except Exception:
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
