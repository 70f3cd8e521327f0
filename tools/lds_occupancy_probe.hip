// How many 256-thread workgroups does a CU hold as their dynamic LDS grows?  (development aid, round 4: the step kernel's
// LDS per workgroup decides whether a fourth wave per SIMD is resident; 160 KB per CU on paper)
//   hipcc --offload-arch=gfx950 -O3 tools/lds_occupancy_probe.hip -o tools/lds_occupancy_probe && tools/lds_occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(float *out) {
  extern __shared__ float s[];
  s[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = s[255 - threadIdx.x];
}
int main() {
  (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  printf("{\"blocks_per_cu_by_dynamic_lds_bytes\": {");
  int last = -1;
  bool first = true;
  for (int bytes = 16 * 1024; bytes <= 160 * 1024; bytes += 256) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, bytes) != hipSuccess) n = -1;
    if (n != last) {
      printf("%s\"%d\": %d", first ? "" : ", ", bytes, n);
      first = false;
      last = n;
    }
  }
  printf("}}\n");
  return 0;
}
