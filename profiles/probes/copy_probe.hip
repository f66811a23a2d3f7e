// What does this box's memory system sustain for a float4 copy / fill / read?  (round 5: bench.py's same-run ceiling must be
// a CEILING -- the first two versions of vgx_bench_stream_ceiling reached 4.4-4.7 TB/s where the headline kernel itself moved
// 6.3 TB/s.)   hipcc --offload-arch=gfx950 -O3 -o copy_probe copy_probe.hip ; ./copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT_LD, bool NT_ST, int MODE>  // MODE 0 copy, 1 fill, 2 read
__global__ __launch_bounds__(256) void k(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  f4 acc = {0, 0, 0, 0};
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += UNROLL * stride) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t i = i0 + u * stride;
      v[u] = acc;
      if (MODE != 1 && i < n) v[u] = NT_LD ? __builtin_nontemporal_load(&src[i]) : src[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t i = i0 + u * stride;
      if (MODE == 2) acc += v[u];
      else if (i < n) {
        if (NT_ST) __builtin_nontemporal_store(v[u], &dst[i]); else dst[i] = v[u];
      }
    }
  }
  if (MODE == 2 && acc.x == 1.2345e30f) dst[0] = acc;
}

template <int UNROLL, bool NT_LD, bool NT_ST, int MODE>
void run(const char* name, const f4* s, f4* d, size_t n, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<UNROLL, NT_LD, NT_ST, MODE>), dim3(grid), dim3(256), 0, 0, s, d, n);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<UNROLL, NT_LD, NT_ST, MODE>), dim3(grid), dim3(256), 0, 0, s, d, n);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  ms /= 5;
  const double bytes = (MODE == 0 ? 2.0 : 1.0) * 16.0 * (double)n;
  printf("%-34s grid %6d unroll %d  %8.3f ms  %7.1f GB/s\n", name, grid, UNROLL, ms, bytes / ms / 1e6);
}

int main() {
  const size_t n = (size_t)9240813440ull / 16;  // the row buffers of the headline launch: 16 B x 577.55 M
  f4 *s, *d;
  if (hipMalloc(&s, n * 16) != hipSuccess || hipMalloc(&d, n * 16) != hipSuccess) return 1;
  hipMemset(s, 1, n * 16);
  hipMemset(d, 0, n * 16);
  for (int grid : {2048, 8192, 65536}) {
    run<4, false, false, 0>("copy plain/plain", s, d, n, grid);
    run<4, true, true, 0>("copy nt/nt", s, d, n, grid);
    run<4, false, true, 0>("copy plain/nt", s, d, n, grid);
    run<1, false, false, 0>("copy plain/plain", s, d, n, grid);
    run<8, false, false, 0>("copy plain/plain", s, d, n, grid);
    run<4, false, false, 1>("fill plain", s, d, n, grid);
    run<4, false, true, 1>("fill nt", s, d, n, grid);
    run<4, false, false, 2>("read plain", s, d, n, grid);
    run<4, true, false, 2>("read nt", s, d, n, grid);
  }
  // one thread per float4, no loop (the guide's shape?)
  {
    const int grid = (int)((n + 255) / 256);
    run<1, false, false, 0>("copy one float4 per thread", s, d, n, grid);
    run<1, false, true, 0>("copy one float4 per thread, nt st", s, d, n, grid);
    run<1, false, false, 1>("fill one float4 per thread", s, d, n, grid);
  }
  hipDeviceSynchronize();
  hipFree(s); hipFree(d);
  return 0;
}
