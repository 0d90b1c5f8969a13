set -u; export TMPDIR=/tmp; O=gpurun_out/r02d; rm -rf $O; mkdir -p $O
B=512; K=8; S=800; W=80
CMD="python bench.py --batch $B --replay-k $K --steps $S --warmup $W --no-cpu-baseline --no-profile"
d=$O/trace_b$B; mkdir -p $d
rocprofv3 --kernel-trace --stats -d $d -o trace -- $CMD > $d/log.txt 2>&1
python tools/trace_summary.py $d/trace_results.db "$CMD" > $O/kernel_trace_b${B}_k$K.txt
d2=$O/csv_b$B; mkdir -p $d2
rocprofv3 --kernel-trace --stats --output-format csv -d $d2 -o trace -- $CMD > $d2/log.txt 2>&1
cp $d2/trace_kernel_stats.csv $O/rocprofv3_kernel_stats_b${B}_k$K.csv 2>/dev/null
cp $O/kernel_trace_b${B}_k$K.txt profiles/r02_kernel_trace_b${B}_k$K.txt
python bench.py --batch 512 --replay-k 8 --steps 2000 --warmup 200 --cpu-seconds 10 2>/dev/null | tail -1 > $O/bench_n1_b512_k8.json
rm -rf $d $d2; head -5 $O/kernel_trace_b512_k8.txt
