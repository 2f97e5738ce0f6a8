// Microbenchmark: how many bytes per clock can ONE CU pull from L2 (data resident: every workgroup re-reads its own small region), by
//   mode 0  global_load_dwordx4 into VGPRs (8 in flight per wave)
//   mode 1  buffer_load_dwordx4 ... lds (LDS-DMA; 16 x 1 KiB pieces per wave in flight, the weight-slab loader's shape)
// as a function of waves per workgroup (one workgroup per CU: 96 KiB of LDS requested) and region size.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/lab/micro/l2_per_cu.out tools/lab/micro/l2_per_cu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ inline u32x4 make_srd(const void* p) {
  const uint64_t a = (uint64_t)p;
  u32x4 r;
  r[0] = (uint32_t)a; r[1] = (uint32_t)(a >> 32) & 0xffff; r[2] = 0xffffffffu; r[3] = 0x00020000u;
  return r;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ src, uint4* __restrict__ out, int region_bytes, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const char* base = (const char*)src + (size_t)blockIdx.x * region_bytes;
  const int pieces = region_bytes / 1024;                    // 1 KiB pieces, round robin over the waves
  if (MODE == 0) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
      for (int p0 = wave * 8; p0 < pieces; p0 += nw * 8) {
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *(const uint4*)(base + (size_t)((p0 + i) % pieces) * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
      }
      asm volatile("" ::: "memory");
    }
    if (acc.x == 0x12345678u) out[threadIdx.x] = acc;
  } else {
    const u32x4 srd = make_srd(base);
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 16384;
    const uint32_t voff = lane * 16;
    for (int it = 0; it < iters; ++it) {
      for (int p0 = wave * 16; p0 < pieces; p0 += nw * 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const uint32_t so = (uint32_t)((p0 + i) % pieces) * 1024;
          asm volatile("s_add_u32 m0, %2, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(srd), "s"(lds), "s"(so), "n"(0) : "memory", "scc");
        }
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // keep two batches in flight
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (smem[threadIdx.x] == 0x7f && iters < 0) out[threadIdx.x] = make_uint4(1, 2, 3, 4);
  }
}

template <int MODE>
void run(const uint4* src, uint4* out, int waves, int region, const char* name) {
  const int blocks = 256, iters = (64 << 20) / region;      // 64 MiB per workgroup
  const size_t lds = 96 * 1024;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, waves * 64, lds>>>(src, out, region, 2);
  hipEventRecord(e0);
  k<MODE><<<blocks, waves * 64, lds>>>(src, out, region, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * iters * region;
  const double tbs = bytes / ms / 1e9;
  printf("%-28s %d waves/CU, %4d KiB region per CU: %6.2f TB/s chip = %5.1f B/clk/CU at 2.1 GHz\n", name, waves, region / 1024, tbs, tbs * 1e12 / 256 / 2.1e9);
}

int main() {
  const size_t n = (size_t)256 * 1024 * 1024;
  uint4 *src, *out;
  hipMalloc(&src, n); hipMalloc(&out, 1 << 20);
  hipMemset(src, 1, n);
  for (int region : {16 * 1024, 64 * 1024, 256 * 1024})
    for (int waves : {1, 2, 4, 8}) {
      run<0>(src, out, waves, region, "global_load_dwordx4 -> VGPR");
      run<1>(src, out, waves, region, "buffer_load_dwordx4 -> LDS");
    }
  return 0;
}
