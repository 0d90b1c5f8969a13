"""profiles/r06_pmc_traffic_sample_fused.json from the raw counter averages of a measurement visit (tools/gpu_round6.sh writes
gpurun_out/r06/pmc_sample_raw.json: FETCH_SIZE / WRITE_SIZE in KiB per gather-kernel launch, grouped by kernel and grid).
python tools/sample_traffic_json.py gpurun_out/r06/pmc_sample_raw.json profiles/r06_pmc_traffic_sample_fused.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import csrc_sha16

raw = json.load(open(sys.argv[1]))
out = {"note": "HBM traffic of hp_buffer_sample_dev's / hp_buffer_sample_dev_f32's gather kernels from separate rocprofv3 passes "
               "(--kernel-trace --pmc FETCH_SIZE / WRITE_SIZE; tools/gpu_round6.sh, tools/ubench/sample_fused.py), averaged over "
               "launches at 5000 and 10000 episodes; counters in KiB; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
               "(MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts wide reads at half their bytes)",
       "csrc_sha16": csrc_sha16(), "kernels": {}}
for key, c in sorted(raw.items()):
    name, grid = key.split("|grid")
    batch = int(grid) // 32 if "fused2" in name or "packed" in name else int(grid)   # 32 lanes per transition
    if batch >= 4096:
        batch = int(grid) // 8                                                        # FLIGHT 4: 8 transitions per wavefront
    f, w = c["FETCH_SIZE"], c["WRITE_SIZE"]
    total = (2 * f + w) * 1024
    out["kernels"][f"{name} batch {batch}"] = {
        "batch": batch, "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes_per_launch": int(round(total)),
        "bytes_per_transition": round(total / batch, 1), "read_bytes_per_transition_x2": round(2 * f * 1024 / batch, 1),
        "written_bytes_per_transition": round(w * 1024 / batch, 1),
        "algorithmic_bytes_per_transition": {"survey_8d_f32_storage": 528, "this_kernel": 572 if "packed" in name else 812}}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v["bytes_per_transition"] for k, v in out["kernels"].items()}))
