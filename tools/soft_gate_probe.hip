// Can a kernel on stream B be released when a kernel on stream A has STARTED ITS LAST ROUND of workgroups (instead of when it has
// finished: hipStreamWaitEvent)?  Four schedules of two identical kernels, each a whole number of rounds plus a partial one:
//   0 same stream (serial)   1 two streams, no dependency   2 B waits for A's event   3 B waits with hipStreamWaitValue32 on A's
//   finished-workgroup counter (signal memory)   4 B waits behind a one-wave kernel that polls the same counter
// build: hipcc -O3 --offload-arch=gfx950 tools/soft_gate_probe.hip -o tools/soft_gate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void work(int iters, unsigned* progress, float* sink) {
  extern __shared__ char lds[];      // 72 KB: two workgroups per CU like k_spend_bits
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; i++) { a = a * b + 1e-7f; b = b * 0.99999f + 1e-6f; }
  if (a == 12345.f) sink[0] = a + b + lds[threadIdx.x];
  if (threadIdx.x == 0 && progress) __hip_atomic_fetch_add(progress, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void gate(const unsigned* progress, unsigned want) {
  while (__hip_atomic_load(progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) __builtin_amdgcn_s_sleep(64);
}

int main() {
  int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  unsigned* sig = nullptr; CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
  unsigned* dcnt = nullptr; CK(hipMalloc(&dcnt, 8));
  float* sink; CK(hipMalloc(&sink, 4));
  CK(hipFuncSetAttribute((const void*)work, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
  const int resident = 512, rounds = 8, grid = resident * rounds + resident / 3, iters = 300000;
  hipEvent_t t0, t1, ev; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int mode = 0; mode <= 4; mode++) {
    if (mode == 3 && !can) { printf("mode 3 skipped\n"); continue; }
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
      unsigned* cnt = mode == 3 ? sig : dcnt;
      if (mode == 3) *(volatile unsigned*)sig = 0; else CK(hipMemset(dcnt, 0, 8));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, sa));
      CK(hipStreamWaitEvent(sb, t0, 0));
      hipLaunchKernelGGL(work, dim3(grid), dim3(256), 72 * 1024, sa, iters, cnt, sink);
      hipStream_t s2 = mode == 0 ? sa : sb;
      if (mode == 2) { CK(hipEventRecord(ev, sa)); CK(hipStreamWaitEvent(sb, ev, 0)); }
      if (mode == 3) CK(hipStreamWaitValue32(sb, sig, grid - resident, hipStreamWaitValueGte, 0xFFFFFFFFu));
      if (mode == 4) hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, sb, cnt, (unsigned)(grid - resident));
      hipLaunchKernelGGL(work, dim3(grid), dim3(256), 72 * 1024, s2, iters, (unsigned*)nullptr, sink);
      CK(hipEventRecord(ev, sb)); CK(hipStreamWaitEvent(sa, ev, 0));
      CK(hipEventRecord(t1, sa));
      CK(hipEventSynchronize(t1));
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (ms < best) best = ms;
    }
    printf("mode %d: %.3f ms for 2 x (%d + 1/3) rounds\n", mode, best, rounds);
  }
  return 0;
}
