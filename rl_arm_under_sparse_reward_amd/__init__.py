"""rl_arm_under_sparse_reward_amd -- MI355X-native HER replay + DDPG update.

Drop-in mirrors of the reference's hot-path objects (PiggyCh/RL_arm_under_sparse_reward):

    her.her_sampler             <- her.py
    replay_buffer.replay_buffer <- replay_buffer.py
    normalizer.normalizer       <- normalizer.py
    models.actor / models.critic<- models.py
    utils.sync_networks / sync_grads <- utils.py   (RCCL instead of mpi4py)
    ddpg_agent.ddpg_agent       <- ddpg_agent.py (learner half + thin rollout glue)
    random                      <- the numpy global RandomState the reference samples from

All arithmetic runs in librlarm_hip.so (csrc/*.hip, gfx950).  Importing this package does not
load the library; the first object that needs the device does, and fails loudly if it cannot.
"""
__version__ = "0.1.0"
