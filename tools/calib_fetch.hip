// calib_fetch.hip — calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for THE ACCESS PATTERN OF THE PIPPENGER BUCKETS (MI355X_MICROARCH.md:
// "FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated: calibrate on a known
// byte count in your own access pattern").  Every lane reads one 144-byte entry (nine global_load_dwordx4) out of nine at a 1 296-byte
// lane stride -- exactly k_spend_bits's bucket_load -- from a 3 GB buffer (no reuse, nothing resident), and writes one back.
// Known bytes: lanes x 144 read, lanes x 144 written.  Run under: rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE) --kernel-trace -- ./calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) k_calib_bucket_rmw(uint32_t* buf, uint32_t lanes, uint32_t salt) {
  uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= lanes) return;
  uint32_t mag = ((gid * 2654435761u) ^ salt) % 9u;
  uint4* p = reinterpret_cast<uint4*>(buf + (size_t)gid * 324 + mag * 36);
  uint4 a[9];
  for (int i = 0; i < 9; i++) a[i] = p[i];
  for (int i = 0; i < 9; i++) { a[i].x += 1; a[i].y ^= a[i].x; }
  for (int i = 0; i < 9; i++) p[i] = a[i];
}
int main() {
  const uint32_t lanes = 1u << 21;                         // 2 M lanes x 1 296 B = 2.7 GB
  uint32_t* d; if (hipMalloc(&d, (size_t)lanes * 1296) != hipSuccess) return 1;
  hipMemset(d, 0, (size_t)lanes * 1296); hipDeviceSynchronize();
  for (int rep = 0; rep < 4; rep++) hipLaunchKernelGGL(k_calib_bucket_rmw, dim3(lanes / 256), dim3(256), 0, 0, d, lanes, 77u * rep);
  hipDeviceSynchronize();
  printf("known bytes per launch: read %llu, written %llu\n", (unsigned long long)lanes * 144, (unsigned long long)lanes * 144);
  return 0;
}
