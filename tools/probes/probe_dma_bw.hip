// Probe: LDS-DMA fill rate per CU as a function of bytes in flight and of where the rows live.
//   grid = 256 workgroups x 512 threads (one per CU), each wave keeps `win` DMA instructions
//   (1 KiB each: 8 rows x 128 B) in flight over its slice of a footprint of `fp_mb` MiB.
//   row stride `stride` bytes between the 8 rows of an instruction (128 = dense, 192/384 = NHWC
//   pixel rows of 96/192 f16 channels), start offset `skew` bytes (16 = rows straddle lines).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_dma16(const v4i rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
template <int WIN>
__global__ __launch_bounds__(512) void k(const char* base, size_t fp_bytes, int iters, int stride, int skew, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  // this workgroup's window of the footprint (wraps)
  const size_t per_wg = fp_bytes / gridDim.x;
  const char* p = base + (size_t)blockIdx.x * per_wg;
  const unsigned long long a = (unsigned long long)p;
  v4i r = {(int)(unsigned)a, (int)((a >> 32) & 0xFFFF), (int)per_wg, 0x00020000};
  const unsigned lane_off = (unsigned)((lane >> 3) * stride + (lane & 7) * 16 + skew);
  const unsigned span = (unsigned)(8 * stride);            // bytes covered by one instruction
  unsigned pos = (unsigned)(wave * span);
  for (int i = 0; i < iters; ++i) {
    unsigned off = pos + lane_off;
    if (off + 16 > per_wg) { pos = (unsigned)(wave * span); off = pos + lane_off; }
    lds_dma16(r, lds_base + (unsigned)((wave * WIN + (i % WIN)) * 1024), off);
    pos += 8 * span;                                         // 8 waves interleave
    if constexpr (WIN == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (WIN == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (WIN == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (WIN == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (WIN == 12) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<float*>(smem + 64);
}
template <int WIN> float run(const char* d, size_t fp, int iters, int stride, int skew, float* sink) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<WIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * WIN * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<WIN><<<256, 512, 8 * WIN * 1024>>>(d, fp, iters / 8, stride, skew, sink);   // warm
  hipEventRecord(e0);
  k<WIN><<<256, 512, 8 * WIN * 1024>>>(d, fp, iters, stride, skew, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  const size_t cap = (size_t)2 << 30;
  char* d; float* sink;
  if (hipMalloc(&d, cap) != hipSuccess) { printf("malloc failed\n"); return 1; }
  hipMalloc(&sink, 4096); hipMemset(d, 1, cap);
  const int iters = 4096;
  printf("%8s %7s %5s %4s | GB/s per CU (chip TB/s) for in-flight KiB per CU = 8,16,32,64,96,128\n", "fp_MiB", "stride", "skew", "");
  const size_t fps[] = {2, 32, 192, 2048};
  const int strides[] = {128, 192, 384};
  for (size_t fpm : fps) for (int st : strides) for (int skew : {0, 16}) {
    if (st == 128 && skew) continue;
    const size_t fp = fpm << 20;
    float ms[6];
    ms[0] = run<1>(d, fp, iters, st, skew, sink); ms[1] = run<2>(d, fp, iters, st, skew, sink);
    ms[2] = run<4>(d, fp, iters, st, skew, sink); ms[3] = run<8>(d, fp, iters, st, skew, sink);
    ms[4] = run<12>(d, fp, iters, st, skew, sink); ms[5] = run<16>(d, fp, iters, st, skew, sink);
    printf("%8zu %7d %5d      |", fpm, st, skew);
    for (int i = 0; i < 6; ++i) {
      const double bytes_cu = (double)iters * 8 * 1024;     // per CU
      const double gbs = bytes_cu / (ms[i] * 1e-3) / 1e9;
      printf(" %6.1f (%4.1f)", gbs, gbs * 256 / 1000);
    }
    printf("\n");
  }
  return 0;
}
