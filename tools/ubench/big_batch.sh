BATCH=256 RLARM_LIB=$PWD/rl_arm_under_sparse_reward_amd/librlarm_tl.so python tools/ubench/notorch_cycle.py 2>&1 | grep "timeline\|us/step"
BATCH=256 RLARM_FB_PREFETCH=0 RLARM_LIB=$PWD/rl_arm_under_sparse_reward_amd/librlarm_tl.so python tools/ubench/notorch_cycle.py 2>&1 | grep "timeline\|us/step"
