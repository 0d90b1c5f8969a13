"""ctypes binding of include/rlarm_hip.h (librlarm_hip.so, gfx950 HIP kernels).

There is deliberately no fallback: if the shared library is missing or no MI355X is
visible, every entry point of this package raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "librlarm_hip.so")

c_void_pp = C.POINTER(C.c_void_p)
f64p = C.POINTER(C.c_double)
f32p = C.POINTER(C.c_float)
i64p = C.POINTER(C.c_int64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


class HpError(RuntimeError):
    pass


class SampleOut(C.Structure):
    _fields_ = [("obs", f64p), ("ag", f64p), ("g", f64p), ("actions", f64p), ("obs_next", f64p), ("ag_next", f64p),
                ("r", f32p), ("e", i64p), ("t", i64p), ("future_t", i64p), ("her", u8p), ("r64", f64p)]


class SampleDevOut(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_next", C.c_void_p), ("actions", C.c_void_p), ("r", C.c_void_p),
                ("e", C.c_void_p), ("t", C.c_void_p), ("future_t", C.c_void_p), ("her", C.c_void_p)]


class AgentCfg(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("goal_dim", C.c_int32), ("act_dim", C.c_int32), ("hidden", C.c_int32),
                ("batch", C.c_int32), ("grad_world_size", C.c_int32),
                ("max_action", C.c_double), ("gamma", C.c_double), ("action_l2", C.c_double),
                ("lr_actor", C.c_double), ("lr_critic", C.c_double), ("polyak", C.c_double),
                ("clip_obs", C.c_double), ("clip_range", C.c_double),
                ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double)]


ABI_VERSION = 4     # HP_ABI_VERSION of include/rlarm_hip.h this table binds

# entry points declared in include/rlarm_hip_debug.h: diagnostics and test hooks, outside the stable surface
DEBUG_SYMBOLS = {"hp_ctx_launch_floor", "hp_ctx_event_pair_us", "hp_ctx_clock_mhz", "hp_ctx_calibrate", "hp_buffer_sample_device_us",
                 "hp_buffer_sample_dev_us", "hp_buffer_sample_dev_fast_us",
                 "hp_agent_set_adam", "hp_agent_debug_chain", "hp_agent_debug_timeline", "hp_agent_update_kernels"}

# name -> (restype, argtypes); every symbol declared in include/rlarm_hip.h and include/rlarm_hip_debug.h
PROTOTYPES = {
    "hp_abi_version": (C.c_int, []),
    "hp_last_error": (C.c_char_p, []),
    "hp_ctx_create": (C.c_int, [C.c_int, c_void_pp]),
    "hp_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hp_ctx_get_stream": (C.c_int, [C.c_void_p, c_void_pp]),
    "hp_ctx_borrow_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hp_ctx_return_stream": (C.c_int, [C.c_void_p]),
    "hp_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "hp_ctx_device_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "hp_ctx_pci_bus_id": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "hp_ctx_launch_floor": (C.c_int, [C.c_void_p, C.c_int, C.c_int, f64p]),
    "hp_ctx_event_pair_us": (C.c_int, [C.c_void_p, C.c_int, f64p]),
    "hp_ctx_clock_mhz": (C.c_int, [C.c_void_p, f64p]),
    "hp_ctx_calibrate": (C.c_int, [C.c_void_p, f64p]),
    "hp_ctx_destroy": (None, [C.c_void_p]),
    "hp_rng_create": (C.c_int, [C.c_void_p, c_void_pp]),
    "hp_rng_seed": (C.c_int, [C.c_void_p, C.c_uint32]),
    "hp_rng_set_state": (C.c_int, [C.c_void_p, u32p, C.c_int32]),
    "hp_rng_get_state": (C.c_int, [C.c_void_p, u32p, C.POINTER(C.c_int32)]),
    "hp_rng_randint": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, i64p]),
    "hp_rng_uniform": (C.c_int, [C.c_void_p, C.c_int64, f64p]),
    "hp_rng_destroy": (None, [C.c_void_p]),
    "hp_buffer_create": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_void_pp]),
    "hp_buffer_store": (C.c_int, [C.c_void_p, C.c_void_p, f64p, f64p, f64p, f64p, C.c_int64]),
    "hp_buffer_stage": (C.c_int, [C.c_void_p, f64p, f64p, f64p, f64p, C.c_int64]),
    "hp_host_register": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "hp_host_unregister": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hp_buffer_store_pinned": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_uint64)]),
    "hp_buffer_store_done": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int32, C.POINTER(C.c_int32)]),
    "hp_buffer_info": (C.c_int, [C.c_void_p, i64p, i64p, i64p, C.POINTER(C.c_int32)]),
    "hp_buffer_last_slots": (C.c_int, [C.c_void_p, i64p, C.c_int64]),
    "hp_buffer_read": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, f64p]),
    "hp_buffer_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.POINTER(SampleOut)]),
    "hp_buffer_sample_device_us": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int32, f64p,
                                             f64p]),
    "hp_buffer_sample_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                       C.c_double, C.POINTER(SampleDevOut)]),
    "hp_buffer_sample_dev_us": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                          C.c_double, C.c_int32, C.c_int32, f64p, f64p]),
    "hp_buffer_enable_f32_rows": (C.c_int, [C.c_void_p]),
    "hp_buffer_sample_dev_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                           C.c_double, C.POINTER(SampleDevOut)]),
    "hp_buffer_sample_dev_fast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double,
                                            C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(SampleDevOut)]),
    "hp_buffer_sample_dev_fast_us": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double,
                                               C.c_int32, C.c_int32, f64p]),
    "hp_buffer_destroy": (None, [C.c_void_p]),
    "hp_compute_reward_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int32,
                                        C.c_void_p, C.c_void_p]),
    "hp_is_success_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p]),
    "hp_compute_reward": (C.c_int, [C.c_void_p, f64p, f64p, C.c_int64, C.c_int32, C.c_double, C.c_int32, f32p, f64p]),
    "hp_is_success": (C.c_int, [C.c_void_p, f64p, f64p, C.c_int64, C.c_int32, C.c_double, f32p]),
    "hp_norm_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32, c_void_pp]),
    "hp_norm_update": (C.c_int, [C.c_void_p, f64p, C.c_int64]),
    "hp_norm_recompute_begin": (C.c_int, [C.c_void_p, c_void_pp, i64p]),
    "hp_norm_recompute_end": (C.c_int, [C.c_void_p]),
    "hp_norm_recompute": (C.c_int, [C.c_void_p]),
    "hp_norm_get": (C.c_int, [C.c_void_p, f32p, f64p, f32p, f32p, f32p, f32p, f32p, f32p]),
    "hp_norm_set_stats": (C.c_int, [C.c_void_p, f32p, f64p]),
    "hp_norm_normalize": (C.c_int, [C.c_void_p, f64p, C.c_int64, C.c_double, f64p]),
    "hp_norm_update_from_staged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]),
    "hp_norm_destroy": (None, [C.c_void_p]),
    "hp_agent_create": (C.c_int, [C.c_void_p, C.POINTER(AgentCfg), c_void_pp]),
    "hp_agent_param_count": (C.c_int64, [C.c_void_p, C.c_int32]),
    "hp_agent_set_params": (C.c_int, [C.c_void_p, C.c_int32, f32p, C.c_int64]),
    "hp_agent_get_params": (C.c_int, [C.c_void_p, C.c_int32, f32p, C.c_int64]),
    "hp_agent_get_grads": (C.c_int, [C.c_void_p, C.c_int32, f32p, C.c_int64]),
    "hp_agent_get_adam": (C.c_int, [C.c_void_p, C.c_int32, f32p, f32p, C.c_int64, i64p]),
    "hp_agent_set_adam": (C.c_int, [C.c_void_p, C.c_int32, f32p, f32p, C.c_int64, C.c_int64]),
    "hp_agent_update_minibatch": (C.c_int, [C.c_void_p, f32p, f32p, f32p, f32p, f32p]),
    "hp_agent_sample_and_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                             C.c_double, C.c_int32]),
    "hp_agent_get_losses": (C.c_int, [C.c_void_p, f32p, C.c_int32]),
    "hp_agent_soft_update": (C.c_int, [C.c_void_p]),
    "hp_agent_actor_forward": (C.c_int, [C.c_void_p, C.c_int32, f32p, C.c_int64, f32p]),
    "hp_agent_critic_forward": (C.c_int, [C.c_void_p, C.c_int32, f32p, f32p, C.c_int64, f32p]),
    "hp_agent_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, f64p, f64p, C.c_int64, C.c_double, f32p]),
    "hp_agent_policy_snapshot": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hp_agent_act_snapshot": (C.c_int, [C.c_void_p, f64p, f64p, C.c_int64, C.c_double, f32p]),
    "hp_agent_forward_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                            C.c_double]),
    "hp_agent_grad_buffer": (C.c_int, [C.c_void_p, c_void_pp, i64p]),
    "hp_agent_param_buffer": (C.c_int, [C.c_void_p, c_void_pp, i64p]),
    "hp_agent_apply": (C.c_int, [C.c_void_p]),
    "hp_agent_sync_targets": (C.c_int, [C.c_void_p]),
    "hp_comm_unique_id": (C.c_int, [u8p]),
    "hp_comm_create": (C.c_int, [C.c_void_p, u8p, C.c_int32, C.c_int32, c_void_pp]),
    "hp_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hp_comm_allreduce_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "hp_comm_allreduce_mean_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "hp_comm_broadcast_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "hp_comm_destroy": (None, [C.c_void_p]),
    "hp_agent_set_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hp_peer_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, c_void_pp, u8p]),
    "hp_peer_connect": (C.c_int, [C.c_void_p, u8p]),
    "hp_peer_allreduce_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "hp_peer_selfcheck": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "hp_peer_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "hp_peer_phases": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "hp_peer_set_gate": (C.c_int, [C.c_void_p, C.c_int32]),
    "hp_peer_destroy": (None, [C.c_void_p]),
    "hp_agent_set_peer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hp_agent_cycle_mode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "hp_agent_set_grad_reduce": (C.c_int, [C.c_void_p, C.c_int32]),
    "hp_agent_train_cycle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, f64p, f64p, f64p,
                                       f64p, C.c_int64, C.c_double, C.c_double, C.c_int32]),
    "hp_agent_train_cycle_pinned": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                              C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_uint64)]),
    "hp_agent_debug_chain": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, f64p]),
    "hp_agent_debug_timeline": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "hp_agent_update_kernels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                          C.c_int32, C.c_int32, C.c_char_p, C.c_int32]),
    "hp_agent_engine": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hp_agent_status": (C.c_int, [C.c_void_p, u32p]),
    "hp_agent_update_form": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "hp_agent_profile": (C.c_int, [C.c_void_p, C.c_int32]),
    "hp_agent_profile_read": (C.c_int, [C.c_void_p, f64p, C.c_int32]),
    "hp_agent_destroy": (None, [C.c_void_p]),
}

_lib = None
_lock = threading.Lock()

# ---- deferred updates --------------------------------------------------------------------------------------------------
# The reference's inner loop calls `_update_network()` once per minibatch (ddpg_agent.py:145-147).  Issued one by one, every call
# is a graph of its own that draws its index plan in a launch in front and gathers its minibatch inside the chain kernel: 51 us
# per update at batch 256.  The mirror therefore only COUNTS argument-less `_update_network()` calls and issues them together --
# one hp_agent_sample_and_update(n): 40.5 us per update -- as soon as anything else touches the library: every entry point goes
# through the proxy below, which first issues what is pending.  n updates in one call are bit for bit n calls of one update
# (tests/test_gpu_update.py).  RLARM_DEFER_UPDATES=0 switches it off.
#
# What a caller CAN observe, and must know (tests/test_gpu_update.py::test_deferred_updates_*):
#   * a zero-copy view of library memory held across calls (`DevicePointer`, e.g. the gradient / parameter arena handed to
#     torch) shows the state as of the LAST LIBRARY CALL, not as of the last `_update_network()`: touch the library (any entry
#     point, or `flush_pending()`) before reading such a view;
#   * an error of a deferred update (an empty buffer: "high <= 0", a dead rank exchange, a hand-off fault) is raised by the call
#     that triggers the flush, possibly from another thread -- the exception then names the deferred call it belongs to
#     ("deferred _update_network() x n").  It is raised ONCE, like the reference's own error at `_update_network()`: the updates
#     that could not be issued stay owed but parked -- no later library call retries them by itself, every object stays usable --
#     until the next `_update_network()` call or `agent.retry_pending_updates()` tries them again, or
#     `agent.discard_pending_updates()` drops them.
pending_lock = threading.RLock()
_pending = []            # objects with a `_flush_updates()` method and work outstanding
_NO_FLUSH = {"hp_last_error", "hp_abi_version"}


def register_pending(obj):
    with pending_lock:
        if not any(o is obj for o in _pending):
            _pending.append(obj)


def unregister_pending(obj):
    with pending_lock:
        _pending[:] = [o for o in _pending if o is not obj]


def flush_pending():
    """Issue every deferred update now (called by the library proxy in front of any other entry point).  If one object's
    flush raises, the objects behind it stay registered (their updates are still owed) and the exception carries on."""
    if not _pending:
        return
    with pending_lock:
        todo, _pending[:] = list(_pending), []
        for i, o in enumerate(todo):
            try:
                o._flush_updates()
            except BaseException:
                for rest in todo[i + 1:]:
                    register_pending(rest)
                raise


class _Library:
    """The ctypes library with one addition: pending deferred updates are issued in front of any other call."""

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)

    def __getattr__(self, name):
        fn = getattr(object.__getattribute__(self, "_cdll"), name)     # AttributeError for unknown symbols, like ctypes
        if name in _NO_FLUSH or not name.startswith("hp_"):
            return fn

        def call(*args):
            if _pending:
                flush_pending()
            return fn(*args)

        call.__name__ = name
        object.__setattr__(self, name, call)       # next access is a plain attribute hit
        return call


def load(path: str | None = None):
    """dlopen the in-tree HIP library and attach prototypes.  Raises if it is not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        p = path or os.environ.get("RLARM_LIB") or LIB_PATH    # RLARM_LIB: A/B builds of the same ABI
        if not os.path.exists(p):
            raise HpError(
                f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
        lib = C.CDLL(p)
        missing = []
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        if missing:    # header/library mismatch (tests/test_abi.py asserts this list is empty)
            raise HpError("librlarm_hip.so does not export: " + ", ".join(missing) + " -- rebuild it")
        if lib.hp_abi_version() != ABI_VERSION:
            raise HpError(f"librlarm_hip.so has ABI version {lib.hp_abi_version()}, this package binds version {ABI_VERSION} "
                          "(include/rlarm_hip.h: HP_ABI_VERSION) -- rebuild it")
        _lib = _Library(lib)
        return _lib


def last_error(lib=None):
    """hp_last_error() of the calling thread as text (for records of failures that are handled, not raised)."""
    return ((lib or load()).hp_last_error() or b"").decode("utf-8", "replace")


def check(status: int):
    """Translate an hp_status into the exception the reference would raise at the same spot."""
    if status == 0:
        return
    msg = (load().hp_last_error() or b"").decode("utf-8", "replace")
    if status in (-1, -2):          # HP_ERR_INVALID / HP_ERR_EMPTY -> numpy raises ValueError there
        raise ValueError(msg)
    raise HpError(f"[hp_status {status}] {msg}")


def ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Context:
    """Process-wide device context (one process per GPU, like one process per MPI rank)."""

    _default = None

    def __init__(self, device_id: int | None = None):
        lib = load()
        if device_id is None:
            device_id = int(os.environ.get("LOCAL_RANK", "0"))
        h = C.c_void_p()
        check(lib.hp_ctx_create(int(device_id), C.byref(h)))
        self.lib, self.h, self.device_id = lib, h, int(device_id)

    @classmethod
    def default(cls):
        if cls._default is None:
            cls._default = cls()
        return cls._default

    def set_stream(self, stream_ptr):
        check(self.lib.hp_ctx_set_stream(self.h, C.c_void_p(stream_ptr or 0)))
        self._bound_stream = stream_ptr or 0

    def use_torch_stream(self):
        import torch

        torch.cuda.set_device(self.device_id)
        # torch's default stream has handle 0, which hp_ctx_set_stream reads as "the context's own stream": name the
        # legacy default stream explicitly (hipStreamLegacy == 1), or the two sides would run unordered
        self.set_stream(torch.cuda.current_stream(self.device_id).cuda_stream or 1)

    def torch_bridge(self):
        """Run ONE library call that writes torch-owned device memory ON torch's current stream, without rebinding the context
        (hp_ctx_borrow_stream / hp_ctx_return_stream): `with ctx.torch_bridge(): ...`.  The launches inside land in torch's
        stream order -- the caching allocator's assumption for memory it handed out, and what the consumer of the outputs runs
        in -- behind what the context's own stream held; the own stream catches up lazily when it is next used, so a fused
        learner on the same context keeps its stream and its cached graphs.  No host synchronisation either way."""
        import contextlib

        import torch

        @contextlib.contextmanager
        def bridge():
            cur = C.c_void_p(torch.cuda.current_stream(self.device_id).cuda_stream or None)
            check(self.lib.hp_ctx_borrow_stream(self.h, cur))
            try:
                yield
            finally:
                self.lib.hp_ctx_return_stream(self.h)
        return bridge()

    def synchronize(self):
        check(self.lib.hp_ctx_synchronize(self.h))

    @property
    def name(self):
        buf = C.create_string_buffer(160)
        check(self.lib.hp_ctx_device_name(self.h, buf, 160))
        return buf.value.decode()


class DevicePointer:
    """Zero-copy view of a library-owned device vector for torch (`torch.as_tensor(obj, device=...)`)."""

    def __init__(self, address: int, n: int, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(address), False),
                                         "version": 2}
