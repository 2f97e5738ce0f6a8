// Microbenchmark: pure v_mfma_f32_32x32x16_bf16 issue rate on random operands (power/clock-dependent practical peak).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int NACC>
__global__ __launch_bounds__(256) void k(const uint4* in, float* out, int iters) {
  uint4 a[2], b[4];
  for (int i = 0; i < 2; ++i) a[i] = in[threadIdx.x + 256 * i];
  for (int i = 0; i < 4; ++i) b[i] = in[threadIdx.x + 256 * (2 + i)];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 1]), __builtin_bit_cast(bf16x8, b[i & 3]), acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  const int zero = argc > 1 ? atoi(argv[1]) : 0;
  uint4* in; float* out;
  hipMalloc(&in, 256 * 6 * 16); hipMalloc(&out, 4096 * 256 * 4);
  unsigned short* h = (unsigned short*)malloc(256 * 6 * 16);
  for (int i = 0; i < 256 * 6 * 8; ++i) h[i] = zero ? 0 : (unsigned short)((rand() & 0x807f) | 0x3f00 | ((rand() & 3) << 7));
  hipMemcpy(in, h, 256 * 6 * 16, hipMemcpyHostToDevice);
  for (int blocks : {256, 512, 1024}) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<8><<<blocks, 256>>>(in, out, 100);
    hipEventRecord(e0); k<8><<<blocks, 256>>>(in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
    printf("%s data, %d blocks x 4 waves (%d waves/SIMD): %.1f TFLOP/s\n", zero ? "zero" : "random", blocks, blocks / 256, fl / ms / 1e9);
  }
  return 0;
}
