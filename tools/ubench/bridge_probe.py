import os, sys, time, contextlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("BATCH", "256")
import runpy
import torch
from rl_arm_under_sparse_reward_amd import _lib
mode = os.environ.get("BRIDGE", "both")
orig = _lib.Context.torch_bridge
def patched(self):
    @contextlib.contextmanager
    def bridge():
        cur = C.c_void_p(torch.cuda.current_stream(self.device_id).cuda_stream or None)
        if mode in ("both", "entry"): _lib.check(self.lib.hp_ctx_fence_stream(self.h, cur, 0))
        yield
        if mode in ("both", "exit"): _lib.check(self.lib.hp_ctx_fence_stream(self.h, cur, 1))
        if mode == "sync": self.synchronize()
    return bridge()
_lib.Context.torch_bridge = patched
print("BRIDGE =", mode)
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "level1_gpu.py"), run_name="__main__")
