#!/bin/bash
# Build tools/ab/liblmm_hip_prev.so from the library sources of a git revision (default HEAD), for same-box A/B runs (tools/ab_lib.sh).
REV=${1:-HEAD}
set -e
T=/tmp/prev_build; rm -rf $T; mkdir -p $T/linearmixingmodels.jl_amd/csrc $T/include
for f in lmm_kernels.hip lmm_api.hip lmm_internal.h lmm_work_item.h lmm_kernels_f32w.hip; do git show $REV:linearmixingmodels.jl_amd/csrc/$f > $T/linearmixingmodels.jl_amd/csrc/$f; done
git show $REV:include/lmm_hip.h > $T/include/lmm_hip.h
cd $T/linearmixingmodels.jl_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -c lmm_kernels.hip -o k.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -c lmm_api.hip -o a.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c lmm_kernels_f32w.hip -o f.o
wait
mkdir -p /root/repo/tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared k.o a.o f.o -o /root/repo/tools/ab/liblmm_hip_prev.so -lrccl
echo built prev from $REV
