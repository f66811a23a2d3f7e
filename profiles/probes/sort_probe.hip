// Which rocprim stable sort is fastest at the sort-based TSDF paths' sizes?  HIP events over 200 sorts each.
//   hipcc --offload-arch=gfx950 -O2 -o build/sort_probe profiles/probes/sort_probe.hip && gpurun -- build/sort_probe
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)

template <unsigned OE, unsigned SB, unsigned SI>
using Few = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<OE, SB, SI, 128, 128, 4, (1u << 20)>,
                                       rocprim::default_config, (1u << 20)>;
template <unsigned OE, unsigned SB, unsigned SI>
using Msc = rocprim::merge_sort_config<OE, SB, SI, 128, 128, 4, (1u << 20)>;

template <class F>
float time_it(F f, hipStream_t st) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) f();
  (void)hipEventRecord(a, st);
  for (int i = 0; i < 200; ++i) f();
  (void)hipEventRecord(b, st);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0f / 200.0f;
}

template <class K>
int run(size_t n, unsigned bits, const char* what) {
  hipStream_t st = nullptr;
  std::vector<K> h(n);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (K)(bits >= 64 ? s : (s & ((1ull << bits) - 1))); }
  K *k_in, *k_out; uint32_t *v_in, *v_out; void* tmp;
  CK(hipMalloc(&k_in, n * sizeof(K))); CK(hipMalloc(&k_out, n * sizeof(K)));
  CK(hipMalloc(&v_in, n * 4)); CK(hipMalloc(&v_out, n * 4));
  size_t cap = 256u << 20;
  CK(hipMalloc(&tmp, cap));
  CK(hipMemcpy(k_in, h.data(), n * sizeof(K), hipMemcpyHostToDevice));
  CK(hipMemset(v_in, 0, n * 4));
  printf("%-34s n=%8zu bits=%2u:", what, n, bits);
  auto radix_default = [&] { size_t b = cap; (void)rocprim::radix_sort_pairs(tmp, b, k_in, k_out, v_in, v_out, n, 0, bits, st); };
  auto few_512_8 = [&] { size_t b = cap; (void)rocprim::radix_sort_pairs<Few<256, 512, 8>>(tmp, b, k_in, k_out, v_in, v_out, n, 0, bits, st); };
  auto few_256_8 = [&] { size_t b = cap; (void)rocprim::radix_sort_pairs<Few<256, 256, 8>>(tmp, b, k_in, k_out, v_in, v_out, n, 0, bits, st); };
  auto few_256_4 = [&] { size_t b = cap; (void)rocprim::radix_sort_pairs<Few<256, 256, 4>>(tmp, b, k_in, k_out, v_in, v_out, n, 0, bits, st); };
  auto few_1024_4 = [&] { size_t b = cap; (void)rocprim::radix_sort_pairs<Few<256, 1024, 4>>(tmp, b, k_in, k_out, v_in, v_out, n, 0, bits, st); };
  auto merge_default = [&] { size_t b = cap; (void)rocprim::merge_sort(tmp, b, k_in, k_out, v_in, v_out, n, rocprim::less<K>(), st); };
  auto merge_512_8 = [&] { size_t b = cap; (void)rocprim::merge_sort<Msc<256, 512, 8>>(tmp, b, k_in, k_out, v_in, v_out, n, rocprim::less<K>(), st); };
  auto merge_256_8 = [&] { size_t b = cap; (void)rocprim::merge_sort<Msc<256, 256, 8>>(tmp, b, k_in, k_out, v_in, v_out, n, rocprim::less<K>(), st); };
  auto merge_256_4 = [&] { size_t b = cap; (void)rocprim::merge_sort<Msc<256, 256, 4>>(tmp, b, k_in, k_out, v_in, v_out, n, rocprim::less<K>(), st); };
  printf(" radix default %6.1f | radix few(512x8) %6.1f (256x8) %6.1f (256x4) %6.1f (1024x4) %6.1f | merge_sort default %6.1f (512x8) %6.1f (256x8) %6.1f (256x4) %6.1f us\n",
         time_it(radix_default, st), time_it(few_512_8, st), time_it(few_256_8, st), time_it(few_256_4, st), time_it(few_1024_4, st),
         time_it(merge_default, st), time_it(merge_512_8, st), time_it(merge_256_8, st), time_it(merge_256_4, st));
  (void)hipFree(k_in); (void)hipFree(k_out); (void)hipFree(v_in); (void)hipFree(v_out); (void)hipFree(tmp);
  return 0;
}

int main() {
  if (run<unsigned long long>(65536, 64, "merged LiDAR points (64-bit keys)")) return 1;
  if (run<unsigned long long>(307200, 64, "merged depth points (64-bit keys)")) return 1;
  if (run<uint32_t>(65536, 21, "LiDAR points by start slot")) return 1;
  if (run<uint32_t>(138000, 20, "merged LiDAR records")) return 1;
  if (run<uint32_t>(237568, 20, "LiDAR accesses")) return 1;
  if (run<uint32_t>(307200, 21, "depth-image points by start slot")) return 1;
  if (run<uint32_t>(65536, 32, "sorted visiting order (f32 bits)")) return 1;
  return 0;
}
