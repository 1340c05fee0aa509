// Probe: how fast can ONE wave issue LDS-DMA instructions, and does re-writing M0 between them matter?
//   256 workgroups x 512 threads; only the first `nw` waves of a workgroup issue DMAs (1 KiB each, dense rows,
//   2 MiB footprint = L2 resident), 8 in flight per wave.
//   mode 0: s_mov m0 before every DMA (what the conv kernels do)
//   mode 1: one s_mov m0 per 4 DMAs, the other three use the instruction offset (offset:1024/2048/3072, which
//           moves BOTH the LDS and the memory address; the memory side is compensated in the VGPR offset)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
//   mode 4: as mode 0 plus a ds_read_b32 + s_waitcnt lgkmcnt(0) after every 4 DMAs
//   mode 2: as mode 0, but every DMA covers 16 rows x 64 B (row stride 192 B) instead of dense bytes;
//   mode 3: 8 rows x 128 B at row stride 192 B (what the K=64 conv stages do for 96-channel f16 rows)
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* base, size_t fp_bytes, int iters, int nw, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const size_t per_wg = fp_bytes / gridDim.x;
  const unsigned long long a = (unsigned long long)(base + (size_t)blockIdx.x * per_wg);
  v4i r = {(int)(unsigned)a, (int)((a >> 32) & 0xFFFF), (int)per_wg, 0x00020000};
  if (wave < nw) {
    const unsigned lane_off = MODE == 2 ? (unsigned)((lane >> 2) * 192 + (lane & 3) * 16)
                            : MODE == 3 ? (unsigned)((lane >> 3) * 192 + (lane & 7) * 16) : (unsigned)(lane * 16);
    unsigned pos = (unsigned)(wave * 4096);
    unsigned long long t0, t1;
    unsigned acc = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; i += 4) {
      if (pos + 4096 > per_wg) pos = (unsigned)(wave * 4096);
      const unsigned off = pos + lane_off;
      const unsigned l0 = lds_base + (unsigned)(wave * 8192 + ((i >> 2) & 1) * 4096);
      if constexpr (MODE != 1) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(l0), "v"(off), "s"(r) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(l0 + 1024), "v"(off + 1024), "s"(r) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(l0 + 2048), "v"(off + 2048), "s"(r) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(l0 + 3072), "v"(off + 3072), "s"(r) : "memory");
      } else {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                     "buffer_load_dwordx4 %1, %2, 0 offen offset:1024 lds\n\t"
                     "buffer_load_dwordx4 %1, %2, 0 offen offset:2048 lds\n\t"
                     "buffer_load_dwordx4 %1, %2, 0 offen offset:3072 lds" ::"s"(l0), "v"(off), "s"(r) : "memory");
      }
      pos += (unsigned)(nw * 4096);
      if constexpr (MODE == 4) {   // an unrelated LDS read + lgkmcnt(0) wait after every 4 DMAs: does it wait for the DMAs?
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_base + 60000u) : "memory");
        acc += v;
      }
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc == 0x12345678u) sink[1] = 1.f;
  }
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<float*>(smem + 64);
}
template <int MODE> void run(const char* d, int nw, unsigned long long* cyc, float* sink) {
  const int iters = 8192;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, 512, 64 * 1024>>>(d, (size_t)2 << 20, iters / 8, nw, cyc, sink);
  hipEventRecord(e0);
  k<MODE><<<256, 512, 64 * 1024>>>(d, (size_t)2 << 20, iters, nw, cyc, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) c += (double)h[b * 8 + w];
  c /= 256.0 * nw;
  printf("mode %d  waves %d | %7.1f s_memtime ticks per DMA per wave | %6.1f GB/s per CU | kernel %.3f ms (%.1f ns per DMA per wave)\n", MODE, nw,
         c / iters, (double)iters * nw * 1024 / (ms * 1e-3) / 1e9, ms, ms * 1e6 / iters);
}
int main() {
  char* d; float* sink; unsigned long long* cyc;
  hipMalloc(&d, (size_t)64 << 20); hipMemset(d, 1, (size_t)64 << 20);
  hipMalloc(&sink, 4096); hipMalloc(&cyc, 2048 * 8);
  for (int nw : {1, 2, 4, 8}) { run<0>(d, nw, cyc, sink); run<1>(d, nw, cyc, sink); run<2>(d, nw, cyc, sink); run<3>(d, nw, cyc, sink); run<4>(d, nw, cyc, sink); }
  return 0;
}
