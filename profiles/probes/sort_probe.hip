// Which rocprim sort configuration is cheapest at the sizes of a LiDAR scan's two sorts (merged TSDF
// integrator): 65 536 x (u64 key, u32 value) and 140 000 x (u32 key of 20 bits, u32 value)?
//   hipcc --offload-arch=gfx950 -O3 -o profiles/probes/sort_probe profiles/probes/sort_probe.hip
//   gpurun -- './profiles/probes/sort_probe'
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <random>

template <class Config, class K>
float time_sort(const char* name, size_t n, int end_bit, const std::vector<K>& h) {
  K *k0, *k1;
  uint32_t *v0, *v1;
  hipMalloc(&k0, n * sizeof(K)); hipMalloc(&k1, n * sizeof(K));
  hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
  hipMemcpy(k0, h.data(), n * sizeof(K), hipMemcpyHostToDevice);
  hipMemset(v0, 0, n * 4);
  size_t bytes = 0;
  rocprim::radix_sort_pairs<Config>(nullptr, bytes, k0, k1, v0, v1, n, 0, end_bit, 0);
  void* tmp;
  hipMalloc(&tmp, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) rocprim::radix_sort_pairs<Config>(tmp, bytes, k0, k1, v0, v1, n, 0, end_bit, 0);
  hipEventRecord(a, 0);
  const int R = 50;
  for (int i = 0; i < R; ++i) rocprim::radix_sort_pairs<Config>(tmp, bytes, k0, k1, v0, v1, n, 0, end_bit, 0);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<K> out(n);
  hipMemcpy(out.data(), k1, n * sizeof(K), hipMemcpyDeviceToHost);
  bool sorted = true;
  for (size_t i = 1; i < n; ++i) sorted &= out[i - 1] <= out[i];
  printf("%-44s n=%7zu bits=%2d  %7.1f us  %s\n", name, n, end_bit, ms / R * 1e3, sorted ? "sorted" : "NOT SORTED");
  hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(tmp);
  return ms / R;
}

int main() {
  using namespace rocprim;
  std::mt19937_64 g(1);
  for (size_t n : {65536ul, 20000ul}) {
    std::vector<unsigned long long> h(n);
    for (auto& x : h) x = g() & ((1ull << 63) - 1);
    time_sort<default_config>("u64 default", n, 64, h);
    time_sort<radix_sort_config<default_config, merge_sort_config<512, 256, 8>>>("u64 merge 2048/block", n, 64, h);
    time_sort<radix_sort_config<default_config, merge_sort_config<512, 256, 16>>>("u64 merge 4096/block", n, 64, h);
    time_sort<radix_sort_config<default_config, merge_sort_config<512, 512, 16>>>("u64 merge 8192/block", n, 64, h);
    for (auto& x : h) x &= (1ull << 26) - 1;
    time_sort<radix_sort_config<default_config, default_config, default_config, 0>>("u64 onesweep, 26 key bits", n, 26, h);
    time_sort<default_config>("u64 default, 26 key bits", n, 26, h);
  }
  for (size_t n : {140000ul, 40000ul, 1000000ul}) {
    std::vector<uint32_t> h(n);
    for (auto& x : h) x = (uint32_t)g() & ((1u << 20) - 1);
    time_sort<default_config>("u32 default", n, 20, h);
    time_sort<radix_sort_config<default_config, merge_sort_config<512, 256, 8>>>("u32 merge 2048/block", n, 20, h);
    time_sort<radix_sort_config<default_config, merge_sort_config<512, 256, 16>>>("u32 merge 4096/block", n, 20, h);
    time_sort<radix_sort_config<default_config, merge_sort_config<512, 512, 16>>>("u32 merge 8192/block", n, 20, h);
    time_sort<radix_sort_config<default_config, default_config, default_config, 0>>("u32 onesweep", n, 20, h);
  }
  return 0;
}
