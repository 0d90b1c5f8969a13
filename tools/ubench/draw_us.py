"""Device time of the stand-alone index-draw kernel (k_draw_plan: MT19937 legacy stream, masked rejection, 53-bit uniforms) and of
the two gather kernels by batch size.  RLARM_LIB selects the build."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_arm_under_sparse_reward_amd import _lib
from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
from rl_arm_under_sparse_reward_amd.replay_buffer import DeviceEpisodeBuffer
from rl_arm_under_sparse_reward_amd.normalizer import normalizer
from rl_arm_under_sparse_reward_amd.her import squared_threshold
from rl_arm_under_sparse_reward_amd.synthetic import make_episodes
ctx = _lib.Context(0); lib = ctx.lib
rng = DeviceRandomState(125, ctx=ctx)
buf = DeviceEpisodeBuffer(5000, 100, 27, 3, 4, ctx=ctx)
buf.store(rng, make_episodes(5000, seed=1))
on, gn = normalizer(27, default_clip_range=5, ctx=ctx), normalizer(3, default_clip_range=5, ctx=ctx)
for B in [int(x) for x in os.environ.get("BATCHES", "100,256,1024,4096,65536").split(",")]:
    d, g, d2, g2 = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    reps = 200 if B <= 4096 else 20
    _lib.check(lib.hp_buffer_sample_device_us(buf.h, rng.h, B, 0.8, squared_threshold(0.05), reps, C.byref(d), C.byref(g)))
    _lib.check(lib.hp_buffer_sample_dev_us(buf.h, rng.h, on.h, gn.h, B, 0.8, squared_threshold(0.05), 200.0, reps, 0, C.byref(d2), C.byref(g2)))
    print(f"batch {B:6d}: index draw {d.value:8.2f} us | gather (float64 dict) {g.value:7.2f} us | fused gather (float32 x, x', a, r) {g2.value:7.2f} us "
          f"= {B / g2.value:8.1f} transitions/us, {812 * B / g2.value / 1e3:7.1f} GB/s of this build's bytes")
