// fence_probe: what do agent-scope release / acquire fences and a cross-workgroup flag hand-off cost on gfx950 while the L2 holds
// dirty tiles?  (round 3: sizing the in-kernel dependency flags of a persistent panel kernel.)
//   hipcc --offload-arch=gfx950 -O3 tools/fence_probe.hip -o tools/fence_probe && ./tools/fence_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void fence_cost(double* buf, long long* out, int tile_doubles, int reps) {
  double* p = buf + (size_t)blockIdx.x * tile_doubles;
  long long t_store = 0, t_rel = 0, t_acq = 0;
  for (int r = 0; r < reps; ++r) {
    long long t0 = wall_clock64();
    for (int i = threadIdx.x; i < tile_doubles; i += blockDim.x) p[i] = (double)(i + r);
    __syncthreads();
    long long t1 = wall_clock64();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    long long t2 = wall_clock64();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    long long t3 = wall_clock64();
    t_store += t1 - t0; t_rel += t2 - t1; t_acq += t3 - t2;
  }
  if (threadIdx.x == 0) { out[3 * blockIdx.x] = t_store; out[3 * blockIdx.x + 1] = t_rel; out[3 * blockIdx.x + 2] = t_acq; }
}

// ping-pong between workgroup pairs (2k, 2k+1) through flags + a data tile: latency of one hand-off incl. fences
__global__ void pingpong(int* flags, double* data, long long* out, int tile_doubles, int reps) {
  const int pair = blockIdx.x >> 1, me = blockIdx.x & 1;
  int* f = flags + 2 * pair;
  double* d = data + (size_t)pair * tile_doubles;
  long long t0 = wall_clock64();
  double acc = 0.0;
  for (int r = 1; r <= reps; ++r) {
    if ((r & 1) == me) {                    // my turn to produce
      for (int i = threadIdx.x; i < tile_doubles; i += blockDim.x) d[i] = (double)r;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(f, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (threadIdx.x == 0) {
        long long ts = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r) {
          __builtin_amdgcn_s_sleep(2);
          if (wall_clock64() - ts > 200000000LL) break;      // 2 s at 100 MHz: never hang
        }
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (int i = threadIdx.x; i < tile_doubles; i += blockDim.x) acc += d[i];
    }
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == -1.0) d[0] = acc;
}

int main() {
  const int tile = 16384;                  // 128 KB per workgroup
  for (int blocks : {1, 64, 512}) {
    double* buf; long long* out;
    CHK(hipMalloc(&buf, (size_t)blocks * tile * 8)); CHK(hipMalloc(&out, blocks * 3 * 8));
    const int reps = 50;
    hipLaunchKernelGGL(fence_cost, dim3(blocks), dim3(256), 0, 0, buf, out, tile, reps);
    CHK(hipDeviceSynchronize());
    std::vector<long long> h(blocks * 3);
    CHK(hipMemcpy(h.data(), out, blocks * 3 * 8, hipMemcpyDeviceToHost));
    double s = 0, r = 0, a = 0;
    for (int b = 0; b < blocks; ++b) { s += h[3 * b]; r += h[3 * b + 1]; a += h[3 * b + 2]; }
    printf("fence_cost  %4d workgroups x 128 KB dirty tile: store %.2f us  release fence %.2f us  acquire fence %.2f us  (100 MHz ticks, mean)\n",
           blocks, s / blocks / reps / 100.0, r / blocks / reps / 100.0, a / blocks / reps / 100.0);
    CHK(hipFree(buf)); CHK(hipFree(out));
  }
  for (int pairs : {1, 32, 256}) {
    for (int t : {64, 16384}) {
      int* flags; double* data; long long* out;
      CHK(hipMalloc(&flags, pairs * 2 * 4)); CHK(hipMemset(flags, 0, pairs * 2 * 4));
      CHK(hipMalloc(&data, (size_t)pairs * t * 8)); CHK(hipMalloc(&out, pairs * 2 * 8));
      const int reps = 200;
      hipLaunchKernelGGL(pingpong, dim3(2 * pairs), dim3(256), 0, 0, flags, data, out, t, reps);
      CHK(hipDeviceSynchronize());
      std::vector<long long> h(pairs * 2);
      CHK(hipMemcpy(h.data(), out, pairs * 2 * 8, hipMemcpyDeviceToHost));
      double s = 0; for (auto v : h) s += v;
      printf("pingpong    %4d pairs, %6d-double tile: %.2f us per hand-off (write tile + release + flag + spin + acquire + read tile)\n",
             pairs, t, s / h.size() / reps / 100.0);
      CHK(hipFree(flags)); CHK(hipFree(data)); CHK(hipFree(out));
    }
  }
  return 0;
}
