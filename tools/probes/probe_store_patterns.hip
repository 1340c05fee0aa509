// Probe: what the memory system gives to the STORE patterns of the conv epilogues.
//   A map of M pixel rows x ROW bytes (NHWC, f16) is written once by a persistent grid (2 workgroups of 256 threads per CU,
//   each owning a contiguous range of 16-pixel tiles, as the conv kernels do), with one of these instruction shapes:
//     0  coalesced          : every wave instruction writes 1 KiB contiguous bytes (what a fill kernel does)
//     1  halo-slab epilogue : lane (pixel = lane & 15, q = lane >> 4) stores 16 B at pixel row + 64 * j + 16 q, j = 0 .. ROW/64 - 1:
//                             16 rows x 64 B per instruction (conv_h2.h, conv_pw.h)
//     2  ring epilogue      : 8 B at pixel row + 32 * j + 8 q: 16 rows x 32 B per instruction (conv_dmap.h, epilogue_fast)
//     3  as 1, through a transpose: 4 consecutive lanes hold 64 B of one row ... 16 lanes = 256 B of ONE row: 4 rows x 256 B per
//        instruction (what staging a tile through the LDS would allow)
//   `gap` = s_sleep units between the store instructions of a tile (the SiLU of the next 8 channels sits there in the kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(char* base, unsigned bytes, int row, int tiles, int gap) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
  const int per = (tiles + (int)gridDim.x - 1) / (int)gridDim.x;               // 16-pixel tiles per workgroup, contiguous
  const int t0 = (int)blockIdx.x * per, t1 = min(tiles, t0 + per);
  const v4i v = {lane, wave, 3, 4};
  const v2i v2 = {lane, wave};
  for (int t = t0 + wave; t < t1; t += 4) {
    const unsigned tb = (unsigned)t * 16u * (unsigned)row;
    if constexpr (MODE == 0) {
      for (int j = 0; j < 16 * row / 1024; ++j) {
        __builtin_amdgcn_raw_buffer_store_b128(v, r, tb + j * 1024 + lane * 16, 0, 0);
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(1);
      }
    } else if constexpr (MODE == 1) {
      const unsigned lo = (unsigned)((lane & 15) * row + (lane >> 4) * 16);
      for (int j = 0; j < row / 64; ++j) {
        __builtin_amdgcn_raw_buffer_store_b128(v, r, tb + lo + j * 64, 0, 0);
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(1);
      }
    } else if constexpr (MODE == 2) {
      const unsigned lo = (unsigned)((lane & 15) * row + (lane >> 4) * 8);
      for (int j = 0; j < row / 32; ++j) {
        __builtin_amdgcn_raw_buffer_store_b64(v2, r, tb + lo + j * 32, 0, 0);
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(1);
      }
    } else {
      const unsigned lo = (unsigned)((lane >> 4) * row + (lane & 15) * 16);       // 4 rows x 256 B
      for (int q = 0; q < 4; ++q)
        for (int j = 0; j < row / 256; ++j) {                                     // row a multiple of 256 here, else the tail is skipped
          __builtin_amdgcn_raw_buffer_store_b128(v, r, tb + (unsigned)(q * 4 * row) + lo + j * 256, 0, 0);
          for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(1);
        }
    }
  }
}
template <int MODE> float run(char* d, unsigned bytes, int row, int tiles, int gap) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<512, 256>>>(d, bytes, row, tiles, gap);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) k<MODE><<<512, 256>>>(d, bytes, row, tiles, gap);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main() {
  char* d;
  const size_t cap = (size_t)1 << 30;
  if (hipMalloc(&d, cap) != hipSuccess) { printf("malloc failed\n"); return 1; }
  hipMemset(d, 0, cap);
  const int rows[] = {192, 384, 768, 1152};
  printf("%6s %8s %4s | TB/s: coalesced | 16 rows x 64 B | 16 rows x 32 B | 4 rows x 256 B\n", "row_B", "MB", "gap");
  for (int row : rows) {
    for (int gap : {0, 4}) {
      const int M = 64 * 80 * 80 * (row <= 384 ? 2 : 1);
      const unsigned bytes = (unsigned)((size_t)M * row);
      const int tiles = M / 16;
      const float a = run<0>(d, bytes, row, tiles, gap), b = run<1>(d, bytes, row, tiles, gap), c = run<2>(d, bytes, row, tiles, gap);
      const float e = (row % 256 == 0) ? run<3>(d, bytes, row, tiles, gap) : 0.f;
      printf("%6d %8.1f %4d | %6.2f | %6.2f | %6.2f | %6.2f\n", row, bytes / 1e6, gap, bytes / a / 1e9, bytes / b / 1e9, bytes / c / 1e9, e > 0 ? bytes / e / 1e9 : 0.0);
      fflush(stdout);
    }
  }
  hipFree(d);
  return 0;
}
