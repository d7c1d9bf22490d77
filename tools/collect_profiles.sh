#!/bin/bash
# Runs ON THE GPU BOX (gpurun): kernel-trace stats + the two PMC passes of the default bench command; writes under gpurun_out/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r05}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
python3 tools/analyze_trace.py $OUT/trace/bench_kernel_trace.csv > $OUT/trace_analysis.txt
# the roofline leg alone (one batch of latents on one stream): rocprofv3's per-kernel average must agree with bench.py's
rocprofv3 --kernel-trace --stats -d $OUT/trace_serial -o bench --output-format csv -- python3 bench.py --steps 0 --warmup 0 --no-cpu-baseline > $OUT/bench_serial_pass_under_rocprof.json 2> $OUT/trace_serial.log
python3 tools/analyze_trace.py $OUT/trace_serial/bench_kernel_trace.csv > $OUT/trace_serial_analysis.txt
rm -f $OUT/trace_serial/bench_kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2> $OUT/pmc_write.log
python3 tools/pmc_to_json.py $OUT/pmc_fetch/pmc_counter_collection.csv $OUT/pmc_write/pmc_counter_collection.csv c2 $OUT/pmc_traffic.json > $OUT/pmc_traffic_main.txt
python3 tools/pmc_summary.py $OUT/pmc_fetch/pmc_counter_collection.csv $OUT/pmc_write/pmc_counter_collection.csv > $OUT/pmc_traffic_summary.txt
rm -f $OUT/trace/bench_kernel_trace.csv $OUT/pmc_fetch/pmc_counter_collection.csv $OUT/pmc_write/pmc_counter_collection.csv   # large; summaries kept
ls -la $OUT
