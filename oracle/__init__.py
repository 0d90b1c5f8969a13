"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path (HER 'future' relabel + episodic replay
sampling + running normalizer + DDPG actor/critic update).  It exists so that the HIP
path can be *checked*; it is never the thing shipped or measured.

Who may import this package (enforced by tests/test_abi.py::test_product_never_imports_oracle):
  * tests/
  * __graft_entry__.smoke()
  * bench.py's `cpu_baseline` leg
Nothing under rl_arm_under_sparse_reward_amd/ imports it; the product fails loudly when
its HIP library is missing instead of falling back to this code.

Pinning status: the reference has NO tests, golden vectors or fixtures of its own
(SURVEY.md section 4), so parity is pinned by running the reference's own Python here
(tools/gen_golden.py imports /root/reference/{her,replay_buffer,normalizer,models,
utils,ddpg_agent}.py with an in-memory single-rank mpi4py stub) and committing its
outputs under tests/golden/.  tests/test_oracle_*.py check every function below
against those fixtures bit-for-bit (integers, rewards, float32 normalizer bits) or to
the tolerance written in the test (float32 network update).

Third-party arithmetic the reference leans on and that is therefore *used*, not
restated, here (same call sites as the reference): numpy legacy RandomState
(her.py:24-31), numpy.linalg.norm (bmirobot_env_push_F.py:23), torch.nn.functional
linear/relu/tanh + autograd (models.py:15-24,32-42), torch.optim.Adam
(ddpg_agent.py:42-43).  The MT19937 stream is additionally restated from scratch in
mt19937_legacy.c so the device generator has an independent, readable twin.
"""
