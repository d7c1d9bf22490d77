// Write-only HBM ceiling on this device (calibration for roofline_gram): vendor memset and three store kernels, 2 and 16 GiB targets.
//   hipcc --offload-arch=gfx950 -O3 tools/store_probe2.hip -o tools/store_probe2 && tools/store_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
// each workgroup owns contiguous CHUNK-byte pieces; a thread writes 4 x 16 B per piece, 4 KB apart (wave = 1 KB contiguous per store)
template <bool NT>
__global__ __launch_bounds__(256) void chunked(f4* p, size_t nchunks, float v) {
  const f4 x = {v, v, v, v};
  for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    f4* q = p + c * 1024 + threadIdx.x;                 // 1024 x 16 B = 16 KB per chunk
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (NT) __builtin_nontemporal_store(x, q + 256 * k); else q[256 * k] = x;
    }
  }
}
// one thread writes 64 contiguous bytes (4 x 16 B): a wave covers 4 KB
__global__ __launch_bounds__(256) void wide(f4* p, size_t n64, float v) {
  const f4 x = {v, v, v, v};
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n64; i += stride) { f4* q = p + 4 * i; q[0] = x; q[1] = x; q[2] = x; q[3] = x; }
}
int main() {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t gib : {2ull, 16ull}) {
    const size_t bytes = gib << 30;
    void* p; if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc %zu GiB failed\n", gib); return 1; }
    auto timeit = [&](const char* name, auto&& fn) {
      fn(); hipDeviceSynchronize();
      const int reps = gib == 2 ? 10 : 3;
      hipEventRecord(e0); for (int r = 0; r < reps; ++r) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%2zu GiB  %-44s %7.0f GB/s\n", gib, name, (double)reps * bytes / ms / 1e6); fflush(stdout);
    };
    timeit("hipMemsetAsync (D8)", [&] { (void)hipMemsetAsync(p, 1, bytes, 0); });
    timeit("hipMemsetD32Async", [&] { (void)hipMemsetD32Async((hipDeviceptr_t)p, 7, bytes / 4, 0); });
    for (int blocks : {1024, 4096, 16384, 65536}) {
      char nm[96];
      snprintf(nm, sizeof nm, "chunked 16 KB per workgroup step, %5d blocks", blocks);
      timeit(nm, [&] { chunked<false><<<blocks, 256>>>((f4*)p, bytes / 16384, 1.0f); });
      snprintf(nm, sizeof nm, "chunked, nontemporal stores,      %5d blocks", blocks);
      timeit(nm, [&] { chunked<true><<<blocks, 256>>>((f4*)p, bytes / 16384, 1.0f); });
      snprintf(nm, sizeof nm, "64 B per thread,                  %5d blocks", blocks);
      timeit(nm, [&] { wide<<<blocks, 256>>>((f4*)p, bytes / 64, 1.0f); });
    }
    hipFree(p);
  }
  return 0;
}
