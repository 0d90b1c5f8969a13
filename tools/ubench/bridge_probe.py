"""level1_gpu.py with the device-output sampler ordered against torch in different ways (measurement helper, round 6):
  BRIDGE=borrow  (default, the product: hp_ctx_borrow_stream -- the sampler's launches go to torch's stream for that call)
  BRIDGE=own     the sampler on the context's OWN stream, torch's stream made to wait for it by an event (exit fence only)
  BRIDGE=sync    the sampler on the context's own stream, the host waits for it
The variants with an event recorded on torch's stream for the way IN (entry fence, both fences) were measured with a round-6
build that had hp_ctx_fence_stream: 2269 / 2206 us per update against 2077 (exit only) and 2083 (borrow); DESIGN.md section 8."""
import contextlib
import ctypes as C
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from rl_arm_under_sparse_reward_amd import _lib

mode = os.environ.get("BRIDGE", "borrow")
if mode != "borrow":
    def patched(self):
        @contextlib.contextmanager
        def bridge():
            yield
            if mode == "sync":
                self.synchronize()
            else:   # torch's current stream waits for the context's stream
                mine = C.c_void_p()
                _lib.check(self.lib.hp_ctx_get_stream(self.h, C.byref(mine)))
                ext = torch.cuda.ExternalStream(mine.value, device=torch.device("cuda", self.device_id))
                torch.cuda.current_stream(self.device_id).wait_stream(ext)
        return bridge()
    _lib.Context.torch_bridge = patched
print("BRIDGE =", mode)
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "level1_gpu.py"), run_name="__main__")
