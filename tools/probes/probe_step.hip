// Probe: the bare K-step skeleton of the ring kernel - is ~2 200 cycles per step inherent to the pattern
//   [wait previous stage + s_barrier] -> [6 LDS-DMAs per wave] -> [14 ds_read_b128 + 24 MFMA per wave]
// with 8 waves per CU, 3 ring slots of 44 KiB, data L2-resident, no address arithmetic, no epilogue?
//   mode 0: as above (DMAs in a burst, then reads + MFMAs)          mode 1: no DMAs (reads + MFMAs + barrier only)
//   mode 2: DMAs + barrier only (no reads / MFMAs)                   mode 3: DMAs spread between the MFMAs
//   mode 4: as 0 without the barrier (waits only)
//   mode 5: mode 0 + the real kernel's per-step K-table ds_read and per-DMA address VALU (add, shift, and-or)
//   mode 6: mode 5 + every 7th DMA of a wave served from a 256 MiB region (HBM/MALL) instead of the window (~14 % misses)
//   mode 8: mode 3 (DMAs between the MFMAs) + mode 5's K-table read and per-DMA address VALU
//   mode 9: mode 8 with the five offsets computed BEFORE the MFMA loop (only s_mov m0 + DMA between the MFMAs)
//   mode 7: mode 6 + an epilogue every 14 steps: 12 pieces of 4 exp + 4 rcp + cvt + one 8-byte store per lane
// window = bytes each workgroup cycles through: 8 KiB (L1), 64 KiB (16 MiB in all: L2), 1 MiB (256 MiB in all: HBM/MALL)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma(const v4i rsrc, unsigned lds_addr, unsigned voff) {
  const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc) : "memory");
}
template <int MODE, int WIN>
__global__ __launch_bounds__(512) void k(const char* base, int steps, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = 352 * 128;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned long long a = (unsigned long long)(base + (size_t)blockIdx.x * (size_t)WIN);
  const v4i r = {(int)(unsigned)a, (int)((a >> 32) & 0xFFFF), WIN, 0x00020000};
  f32x4 acc[3][4];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4, wp = wave >> 1, wc = wave & 1;
  auto lds_off = [&](int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); };
  unsigned pos = 0;
  const unsigned long long a2 = (unsigned long long)(base + ((size_t)64 << 20) + (size_t)blockIdx.x * (size_t)(1 << 20));
  const v4i r2 = {(int)(unsigned)a2, (int)((a2 >> 32) & 0xFFFF), 1 << 20, 0x00020000};
  unsigned* ktab = reinterpret_cast<unsigned*>(smem + 3 * STAGE);
  if (threadIdx.x < 64) ktab[threadIdx.x] = threadIdx.x * 16u;
  __syncthreads();
  unsigned kofs = 0, xinv = lane & 1, tp = 0, dmacount = 0;
  auto issue = [&](int slot, int d) {
    unsigned off = (pos + (unsigned)((wave + 8 * d) * 1024 + lane * 16)) & (unsigned)(WIN - 1);
    if (MODE >= 5 && MODE != 9) off = ((off + kofs) | (((xinv >> tp) & 1u) << 31)) & 0x7FFFFFFFu & (unsigned)(WIN - 1);
    bool far = false;
    if (MODE == 6 || MODE == 7) { far = (dmacount % 7u) == 6u; ++dmacount; }
    if (far) dma(r2, lds_base + (unsigned)(slot * STAGE + (wave + 8 * d) * 1024), (pos * 7u + (unsigned)(lane * 16 + d * 4096)) & ((1u << 20) - 1));
    else dma(r, lds_base + (unsigned)(slot * STAGE + (wave + 8 * d) * 1024), off);
  };
  unsigned long long t0, t1;
  if (MODE != 1) { for (int d = 0; d < 5; ++d) issue(0, d); pos += 45056; for (int d = 0; d < 5; ++d) issue(1, d); pos += 45056; }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  int slot = 0, dslot = 2;
  for (int c = 0; c < steps; ++c) {
    if (MODE == 1) asm volatile("s_barrier" ::: "memory");
    else if (MODE == 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
    if (MODE >= 5) { kofs = ktab[(c & 7) * 8 + (lane & 7)]; tp = (unsigned)(c % 9); }
    if (MODE == 0 || MODE == 2 || MODE == 4 || (MODE >= 5 && MODE <= 7)) { for (int d = 0; d < 5; ++d) issue(dslot, d); }
    unsigned pre[5];
    if (MODE == 9) { for (int d = 0; d < 5; ++d) pre[d] = ((((pos + (unsigned)((wave + 8 * d) * 1024 + lane * 16)) & (unsigned)(WIN - 1)) + kofs) | (((xinv >> tp) & 1u) << 31)) & 0x7FFFFFFFu & (unsigned)(WIN - 1); }
    if (MODE != 2) {
      const unsigned char* xs = smem + slot * STAGE;
      const unsigned char* ws = xs + 256 * 128;
      int dd = 0;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        uint4 af[3], bf[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) af[i] = *reinterpret_cast<const uint4*>(ws + lds_off((wc * 3 + i) * 16 + frow, kk * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const uint4*>(xs + lds_off((wp * 4 + j) * 16 + frow, kk * 4 + fq));
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&af[i]), *reinterpret_cast<const f16x8*>(&bf[j]), acc[i][j], 0, 0, 0);
            if ((MODE == 3 || MODE == 8) && ((kk * 12 + i * 4 + j) % 5 == 0) && dd < 5) { issue(dslot, dd); ++dd; }
            if (MODE == 9 && ((kk * 12 + i * 4 + j) % 5 == 0) && dd < 5) { dma(r, lds_base + (unsigned)(dslot * STAGE + (wave + 8 * dd) * 1024), pre[dd]); ++dd; }
          }
      }
    }
    if (MODE >= 7 && (c % 14) == 13) {
      float* outp = sink + 64 + ((size_t)blockIdx.x * 512 + threadIdx.x) * 2;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { const float x = acc[i][j][q] + 0.1f; v[q] = x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); acc[i][j][q] = 0.f; }
          typedef _Float16 h4 __attribute__((ext_vector_type(4)));
          h4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
          *reinterpret_cast<h4*>(outp) = hv;
        }
    }
    pos += 45056;
    slot = (slot == 2) ? 0 : slot + 1;
    dslot = (dslot == 2) ? 0 : dslot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0];
  if (s == 1234.5f) sink[0] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE, int WIN> void run(const char* d, unsigned long long* cyc, float* sink) {
  const int steps = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, WIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 352 * 128 + 1024);
  k<MODE, WIN><<<256, 512, 3 * 352 * 128 + 1024>>>(d, 200, cyc, sink);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE, WIN><<<256, 512, 3 * 352 * 128 + 1024>>>(d, steps, cyc, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < 2048; ++i) c += (double)h[i]; c /= 2048;
  printf("window %4d KiB per workgroup  mode %d: %7.0f cycles per step (%.3f ms; MFMA-only bound 768, DMA-only bound 640 for 40 KiB)\n", WIN >> 10, MODE, c / steps, ms);
}
int main() {
  char* d; float* sink; unsigned long long* cyc;
  hipMalloc(&d, (size_t)384 << 20); hipMemset(d, 0, (size_t)384 << 20);
  hipMalloc(&sink, 64 * 4 + (size_t)256 * 512 * 8); hipMalloc(&cyc, 2048 * 8);
  run<0, 1 << 16>(d, cyc, sink); run<1, 1 << 16>(d, cyc, sink); run<2, 1 << 16>(d, cyc, sink); run<3, 1 << 16>(d, cyc, sink); run<4, 1 << 16>(d, cyc, sink);
  run<5, 1 << 16>(d, cyc, sink); run<6, 1 << 16>(d, cyc, sink); run<7, 1 << 16>(d, cyc, sink); run<8, 1 << 16>(d, cyc, sink); run<9, 1 << 16>(d, cyc, sink);
  run<0, 1 << 20>(d, cyc, sink); run<2, 1 << 20>(d, cyc, sink);
  run<0, 1 << 13>(d, cyc, sink); run<2, 1 << 13>(d, cyc, sink);
  return 0;
}
