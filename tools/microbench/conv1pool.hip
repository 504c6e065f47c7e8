// Where does k_conv1_pool (fused conv1 + maxpool1 of the tile scorer) spend its time?  The production kernel with one
// phase removed at a time (EXP bits, see cnn_kernels.hip), 512 tiles.
// Build (from the repo root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Isrcfinder_amd/csrc \
//   tools/microbench/conv1pool.hip -Lsrcfinder_amd -lsrcfinder_amd -Wl,-rpath,$PWD/srcfinder_amd -o tools/microbench/conv1pool
#include "../../srcfinder_amd/csrc/cnn_kernels.hip"
#include <cstdio>
#include <vector>

template <int EXP>
float run(const float *padded, int Wp, int W, int ntiles, const float *w, const float *b, float *out) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv1_pool<EXP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cp_lds_bytes());
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_conv1_pool<EXP>, dim3(64, ntiles), dim3(256), cp_lds_bytes(), 0, padded, Wp, W, 0ll, w, b, out);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r)
    hipLaunchKernelGGL(k_conv1_pool<EXP>, dim3(64, ntiles), dim3(256), cp_lds_bytes(), 0, padded, Wp, W, 0ll, w, b, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("EXP %2d: %8.1f us\n", EXP, ms / 3 * 1e3f);
  return ms / 3;
}

template <int EXP>
float run16(const float *padded, int Wp, int W, int ntiles, const float *w, const float *b, float *out) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv1_pool16<EXP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c2_lds_bytes());
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_conv1_pool16<EXP>, dim3(16, ntiles), dim3(C2_NT), c2_lds_bytes(), 0, padded, Wp, W, 0ll, w, b, out);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r)
    hipLaunchKernelGGL(k_conv1_pool16<EXP>, dim3(16, ntiles), dim3(C2_NT), c2_lds_bytes(), 0, padded, Wp, W, 0ll, w, b, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("16x16 form, EXP %2d: %8.1f us\n", EXP, ms / 3 * 1e3f);
  return ms / 3;
}

int main() {
  const int W = 64, H = 8, ntiles = W * H, Wp = W + 255, Hp = H + 255;
  float *padded, *w, *b, *out;
  (void)hipMalloc(&padded, (size_t)Hp * Wp * 4);
  (void)hipMalloc(&w, 64 * 49 * 4);
  (void)hipMalloc(&b, 64 * 4);
  (void)hipMalloc(&out, (size_t)ntiles * 64 * 64 * 64 * 4);
  std::vector<float> h((size_t)Hp * Wp);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
  (void)hipMemcpy(padded, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(w, h.data(), 64 * 49 * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(b, h.data(), 64 * 4, hipMemcpyHostToDevice);
  run<0>(padded, Wp, W, ntiles, w, b, out);
  run<1>(padded, Wp, W, ntiles, w, b, out);
  run<2>(padded, Wp, W, ntiles, w, b, out);
  run<3>(padded, Wp, W, ntiles, w, b, out);
  run<4>(padded, Wp, W, ntiles, w, b, out);
  run<8>(padded, Wp, W, ntiles, w, b, out);
  run<16>(padded, Wp, W, ntiles, w, b, out);
  run<32>(padded, Wp, W, ntiles, w, b, out);
  run<48>(padded, Wp, W, ntiles, w, b, out);
  run<56>(padded, Wp, W, ntiles, w, b, out);
  run<60>(padded, Wp, W, ntiles, w, b, out);
  run<63>(padded, Wp, W, ntiles, w, b, out);
  run16<0>(padded, Wp, W, ntiles, w, b, out);
  run16<4>(padded, Wp, W, ntiles, w, b, out);
  run16<8>(padded, Wp, W, ntiles, w, b, out);
  run16<32>(padded, Wp, W, ntiles, w, b, out);
  run16<44>(padded, Wp, W, ntiles, w, b, out);
  return 0;
}
