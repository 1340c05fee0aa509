// Probe: semantics of buffer_load_dwordx4 ... lds (LDS-DMA) on gfx950.
//  (1) lane -> LDS placement for 16-byte loads, (2) what an out-of-range lane writes
//  (zeros or nothing), (3) s_waitcnt vmcnt(0) + barrier makes the data visible to ds_read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void probe(const int* src, unsigned bytes, int* out) {
  __shared__ __attribute__((aligned(16))) int lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = -7;      // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(src), 0, bytes, 0x00020000);
  const int lane = threadIdx.x;
  // lanes 0..63 fetch chunk (63 - lane) of the source (reversed), odd lanes are out of range
  unsigned off = (unsigned)(63 - lane) * 16u;
  if (lane & 1) off = 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
  // second instruction at LDS offset 1024 B with an immediate, in-range for everyone
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, (unsigned)lane * 16u, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<int> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
  int *d, *o;
  (void)hipMalloc(&d, 4096); (void)hipMalloc(&o, 2048);
  (void)hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, 1024, o);
  std::vector<int> r(512);
  if (hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost) != hipSuccess) { printf("memcpy failed\n"); return 1; }
  printf("first instruction (lane L fetched chunk 63-L; odd lanes OOB):\n");
  for (int L = 0; L < 8; ++L) printf(" lane %d -> lds[%d..]= %d %d %d %d\n", L, L * 4, r[L * 4], r[L * 4 + 1], r[L * 4 + 2], r[L * 4 + 3]);
  int zeros = 0, sentinels = 0, ok_even = 0;
  for (int L = 0; L < 64; ++L) {
    if (L & 1) { if (r[L * 4] == 0) ++zeros; else if (r[L * 4] == -7) ++sentinels; }
    else if (r[L * 4] == 1000 + (63 - L) * 4) ++ok_even;
  }
  printf("even lanes placed at base+lane*16 with their own source chunk: %d/32\n", ok_even);
  printf("OOB lanes: wrote zeros %d/32, left sentinel %d/32\n", zeros, sentinels);
  int ok2 = 0;
  for (int L = 0; L < 64; ++L) if (r[256 + L * 4] == 1000 + L * 4) ++ok2;
  printf("second instruction at lds+1024B: %d/64 correct\n", ok2);
  return 0;
}
