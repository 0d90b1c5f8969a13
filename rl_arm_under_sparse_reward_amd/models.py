"""actor / critic -- state-dict-compatible mirrors of the reference's models.py.

The reference networks are (models.py:11-44)
    actor : Linear(obs+goal,256)-ReLU-Linear(256,256)-ReLU-Linear(256,256)-ReLU-Linear(256,action), max_action*tanh
    critic: same trunk on cat[x, actions/max_action], Linear(256,1) head
with parameter names fc1, fc2, fc3, action_out | q_out -- the names the `.pt` checkpoints carry
(ddpg_agent.py:158-161, read back by demo_push.py:28,41).

Here the modules are *parameter containers*: torch.nn.Linear layers are created (so the default
initialisation consumes the torch RNG exactly like the reference and `state_dict()` /
`load_state_dict()` interoperate with reference checkpoints), but the arithmetic runs in
csrc/agent.hip on the device copy owned by a `ddpg_agent`.  Calling a module that is not
attached to a learner raises: there is no host forward path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

_TRUNK = ("fc1", "fc2", "fc3")


class _DeviceMLP(nn.Module):
    head_name = ""

    def __init__(self, env_params, in_features, out_features, hidden=256):
        super().__init__()
        self.max_action = env_params['action_max']
        widths = [in_features, hidden, hidden, hidden]
        for name, w_in in zip(_TRUNK, widths):
            setattr(self, name, nn.Linear(w_in, hidden))
        setattr(self, self.head_name, nn.Linear(hidden, out_features))
        self._learner = None      # set by ddpg_agent
        self._slot = None         # HP_NET_* id inside the learner

    # ---- flat views in named_parameters() order (utils.py:18-40 of the reference)
    def flat_parameters(self) -> np.ndarray:
        return np.concatenate([p.detach().cpu().numpy().ravel() for _, p in self.named_parameters()]).astype(np.float32)

    def load_flat_parameters(self, flat):
        flat = np.asarray(flat, dtype=np.float32)
        off = 0
        with torch.no_grad():
            for _, p in self.named_parameters():
                n = p.numel()
                p.copy_(torch.from_numpy(flat[off:off + n].reshape(tuple(p.shape)).copy()))
                off += n
        if off != flat.size:
            raise ValueError(f"flat vector has {flat.size} values, network needs {off}")

    def attach(self, learner, slot):
        self._learner, self._slot = learner, slot

    def pull(self):
        """Refresh the host tensors from the device copy (before state_dict()/checkpointing)."""
        if self._learner is not None:
            self.load_flat_parameters(self._learner._get_flat(self._slot))
        return self

    def push(self):
        """Upload the host tensors to the device copy (after load_state_dict())."""
        if self._learner is not None:
            self._learner._set_flat(self._slot, self.flat_parameters())
        return self

    def state_dict(self, *a, **kw):
        self.pull()
        return super().state_dict(*a, **kw)

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        self.push()
        return out

    def _require_learner(self):
        if self._learner is None:
            raise RuntimeError(
                f"{type(self).__name__} is a parameter container; its forward pass runs on the MI355X through "
                "ddpg_agent (no host fallback).  Attach it to a learner first.")
        return self._learner


class actor(_DeviceMLP):
    head_name = "action_out"

    def __init__(self, env_params):
        super().__init__(env_params, env_params['obs'] + env_params['goal'], env_params['action'])

    def forward(self, x):
        """models.py:19-26 on the device: x [rows, obs+goal] float32 -> actions [rows, action]."""
        learner = self._require_learner()
        arr = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        out = learner._actor_forward(self._slot, arr)
        return torch.from_numpy(out) if isinstance(x, torch.Tensor) else out


class critic(_DeviceMLP):
    head_name = "q_out"

    def __init__(self, env_params):
        super().__init__(env_params, env_params['obs'] + env_params['goal'] + env_params['action'], 1)

    def forward(self, x, actions):
        """models.py:36-44 on the device: x [rows, obs+goal] float32, actions [rows, action] -> q [rows, 1].  (Inside the
        update the Q-values are produced and consumed by the fused kernels; this stand-alone forward serves callers that
        evaluate the critic themselves.)"""
        learner = self._require_learner()
        to_np = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        out = learner._critic_forward(self._slot, to_np(x), to_np(actions))
        return torch.from_numpy(out) if isinstance(x, torch.Tensor) else out
