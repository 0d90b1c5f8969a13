"""Hyper-parameters of the hot path, named like the reference's live config
(arguments.py:74-106 `Args`; the argparse variant there is dead code)."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class Args:
    n_epochs: int = 200
    n_cycles: int = 50
    n_batches: int = 40
    seed: int = 125
    replay_strategy: str = "future"
    clip_return: float = 50.0          # unused by the reference too (it uses 1/(1-gamma))
    save_dir: str = "saved_models/"
    noise_eps: float = 0.01
    random_eps: float = 0.3
    buffer_size: float = 1e6 * 1 / 2
    replay_k: int = 4
    clip_obs: float = 200
    batch_size: int = 256
    gamma: float = 0.98
    action_l2: float = 1
    lr_actor: float = 0.001
    lr_critic: float = 0.001
    polyak: float = 0.95
    n_test_rollouts: int = 25
    clip_range: float = 5
    cuda: bool = True                   # informational: the learner always runs on the MI355X
    num_rollouts_per_mpi: int = 2
    add_demo: bool = False
    demo_name: str = "bmirobot_1000_push_demo.npz"
    train_type: str = "push"
    env_name: str = "bmirobot_push seed125"
    distance_threshold: float = 0.05    # bmirobot_push_F.py:20 / bmirobot_pickandplace_v2.py:19
    reward_type: str = "sparse"         # bmirobot_push_F.py:9; "dense" = -distance (compute_reward :89-90)
    share_numpy_stream: bool = True     # learn(): one random stream for exploration + sampler, like the reference's np.random
    grad_reduce: str = "sum"            # data-parallel gradient exchange: "sum" = utils.py:47 (reference), or "mean"
