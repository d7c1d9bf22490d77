#!/bin/bash
# Runs ON THE GPU BOX, part A of the round's evidence at HEAD: the full GPU suite, then rocprofv3 stats + PMC passes of the default bench command.
mkdir -p gpurun_out/final; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/final/pytest_gpu.log 2>&1; tail -3 gpurun_out/final/pytest_gpu.log
bash tools/collect_profiles.sh final > gpurun_out/collect_final.log 2>&1; tail -3 gpurun_out/collect_final.log
