// Do two HIP streams of this process run kernels concurrently?  16 non-blocking streams are created as lmm_init does (the last 8
// optionally at high priority); a kernel that spins ~2 ms on 64 workgroups is launched on stream a and on stream b; wall time ~2 ms:
// concurrent, ~4 ms: serialised (streams sharing a hardware queue).   hipcc --offload-arch=gfx950 -O2 stream_concurrency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void spin(long long ticks, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
int main(int argc, char** argv) {
  const int prio = argc > 1 ? atoi(argv[1]) : 1;
  hipStream_t st[16];
  int lo = 0, hi = 0;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  printf("priority range: lowest %d highest %d; aux streams %s\n", lo, hi, prio ? "high priority" : "default priority");
  for (int s = 0; s < 16; ++s) {
    if (s >= 8 && prio) hipStreamCreateWithPriority(&st[s], hipStreamNonBlocking, hi);
    else hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
  }
  int* out; hipMalloc(&out, 4096);
  spin<<<64, 64, 0, st[0]>>>(1000, out); hipDeviceSynchronize();
  const int pairs[][2] = {{0, 1}, {0, 4}, {0, 8}, {0, 9}, {1, 9}, {3, 11}, {0, 12}};
  for (auto& p : pairs) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    spin<<<64, 64, 0, st[p[0]]>>>(200000, out);
    spin<<<64, 64, 0, st[p[1]]>>>(200000, out);
    hipDeviceSynchronize();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("streams %2d + %2d : %.2f ms  (%s)\n", p[0], p[1], ms, ms < 3.0 ? "concurrent" : "SERIALISED");
  }
  // event hand-off: a on stream 0, then (event) b on stream 8 while c runs on stream 0
  {
    hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    spin<<<64, 64, 0, st[0]>>>(100000, out);
    hipEventRecord(e, st[0]);
    hipStreamWaitEvent(st[8], e, 0);
    spin<<<64, 64, 0, st[8]>>>(200000, out);
    spin<<<64, 64, 0, st[0]>>>(200000, out);
    hipDeviceSynchronize();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("1 ms on s0, then 2 ms on s8 beside 2 ms on s0: %.2f ms (3 = overlapped, 5 = serialised)\n", ms);
  }
  return 0;
}
