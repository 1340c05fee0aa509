// Probe: sustained dense-MFMA rate of the whole chip (what "peak" means under load on this box).
//   256 x NB workgroups of 256/512 threads, every wave runs a loop of 16 independent v_mfma_f32_16x16x32_f16
//   (or 16x16x4 f32) accumulators; no memory traffic.  Reports TFLOP/s and the shader clock seen by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int F32>
__global__ __launch_bounds__(512) void k(int iters, float* sink, unsigned long long* cyc) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(threadIdx.x * 0.002f - i); }
  const float af = threadIdx.x * 0.5f, bf = threadIdx.x * 0.25f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (F32) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int F32> void run(int threads, int nb, int iters, float* sink, unsigned long long* cyc) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<F32><<<256 * nb, threads>>>(iters / 10, sink, cyc);
  hipEventRecord(e0);
  k<F32><<<256 * nb, threads>>>(iters, sink, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < 256; ++i) c += (double)h[i]; c /= 256;
  const double waves = 256.0 * nb * threads / 64;
  const double flop = waves * iters * 16.0 * (F32 ? 16 * 16 * 4 * 2 : 16 * 16 * 32 * 2);
  printf("%s  %d waves/CU: %8.1f TFLOP/s  (%.3f ms, %.0f ticks per workgroup -> %.2f GHz if ticks are shader clocks; %.1f cycles per MFMA per SIMD)\n",
         F32 ? "f32 16x16x4 " : "f16 16x16x32", nb * threads / 64, flop / (ms * 1e-3) / 1e12, ms, c, c / (ms * 1e-3) / 1e9,
         c / (iters * 16.0 * (nb * threads / 64) / 4.0));
}
int main() {
  float* sink; unsigned long long* cyc;
  hipMalloc(&sink, 64); hipMalloc(&cyc, 256 * 8 * 4);
  for (int nb : {1, 2}) { run<0>(256, nb, 20000, sink, cyc); run<0>(512, nb, 20000, sink, cyc); }
  run<1>(512, 1, 20000, sink, cyc);
  run<0>(512, 1, 200000, sink, cyc);      // 10x longer: does the clock sag?
  return 0;
}
