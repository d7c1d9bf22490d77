cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4tl
for w in c0 notebook; do
rocprofv3 --kernel-trace -d gpurun_out/r4tl/$w -o t --output-format csv -- python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r4tl/$w.json 2> gpurun_out/r4tl/$w.err
python3 tools/trace_one_eval.py gpurun_out/r4tl/$w/t_kernel_trace.csv > gpurun_out/r4tl/timeline_$w.txt
rm -rf gpurun_out/r4tl/$w
done
cat gpurun_out/r4tl/timeline_c0.txt gpurun_out/r4tl/timeline_notebook.txt
for w in c0 notebook; do python3 bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline --no-roofline > gpurun_out/r4tl/bench_$w.json 2>/dev/null; python3 -c "import json;d=json.load(open('gpurun_out/r4tl/bench_$w.json'));print('$w', d['ms_per_step'])"; done
