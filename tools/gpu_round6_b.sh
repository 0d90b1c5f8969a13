#!/bin/bash
# Round-6 visit B: the fused sampler (k_gather_fused2) -- parity tests, timing over shard sizes (5000 episodes = 149 MB, Infinity-
# Cache resident; 10000 / 20000 = 298 / 597 MB, HBM resident), and its HBM traffic from separate FETCH_SIZE / WRITE_SIZE passes.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_her.py tests/test_gpu_level1.py -q -rs > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_update.py tests/test_gpu_teacher_forced.py -q -rs -k "torch or tilessplit" > $O/pytest2.log 2>&1; echo "pytest2 rc $?"; tail -6 $O/pytest2.log
EPISODES=5000,10000,20000 BATCHES=256,4096,65536,262144,1048576 python tools/ubench/sample_fused.py > $O/sample_fused_sweep.txt 2>&1; cat $O/sample_fused_sweep.txt | grep -v amdgpu.ids
for c in FETCH_SIZE WRITE_SIZE; do
  d=$O/pmc_$c; rm -rf $d; mkdir -p $d
  EPISODES=5000,10000 BATCHES=256,262144 REPS=10 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python tools/ubench/sample_fused.py > $d/log.txt 2>&1
done
python - <<'PY'
import sqlite3, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = glob.glob(f"gpurun_out/r06b/pmc_{c}/**/*.db", recursive=True)
    if not dbs:
        print("no db for", c); continue
    cur = sqlite3.connect(dbs[0]).cursor()
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    print(c, cols)
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    gx = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
    q = f"select {name_col}, {gx if gx else 0}, count(*), avg(value) from counters_collection where counter_name='{c}' and {name_col} like '%gather_fused%' group by {name_col}, {gx if gx else 0}"
    for r in cur.execute(q):
        print(c, r)
        out.setdefault(f"{r[0].split('(')[0]}|grid{r[1]}", {})[c] = r[3]
json.dump(out, open("gpurun_out/r06b/pmc_sample_fused_raw.json", "w"), indent=1)
PY
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- env EPISODES=5000 BATCHES=256,262144 REPS=20 python tools/ubench/sample_fused.py > $O/trace_log.txt 2>&1
python tools/trace_summary.py $(find $O/trace -name "*.db" | head -1) "EPISODES=5000 BATCHES=256,262144 REPS=20 python tools/ubench/sample_fused.py" > $O/kernel_trace_sample_fused.txt 2>&1; head -12 $O/kernel_trace_sample_fused.txt
rm -rf $O/trace $O/pmc_FETCH_SIZE/*/ $O/pmc_WRITE_SIZE/*/ 2>/dev/null; ls $O
