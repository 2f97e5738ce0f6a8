// Microbenchmark: does the MFMA SHAPE change the energy per FLOP?  v_mfma_f32_32x32x16_f16 (1024 accumulators written per 16 384 MACs) against
// v_mfma_f32_16x16x32_f16 (256 per 8 192 MACs), random fp16 operands, every CU busy, each variant looped for ~2.5 s so that rocm-smi can be
// sampled beside it (tools/lab/micro/mfma_energy.sh).  On a power-capped chip the variant with fewer joules per FLOP shows as MORE TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int SHAPE>
__global__ __launch_bounds__(256) void k(const uint4* in, float* out, int iters) {
  uint4 a[2], b[4];
  for (int i = 0; i < 2; ++i) a[i] = in[threadIdx.x + 256 * i];
  for (int i = 0; i < 4; ++i) b[i] = in[threadIdx.x + 256 * (2 + i)];
  float s = 0;
  if constexpr (SHAPE == 32) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i & 1]), __builtin_bit_cast(f16x8, b[i & 3]), acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i & 1]), __builtin_bit_cast(f16x8, b[i & 3]), acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  const int shape = argc > 1 ? atoi(argv[1]) : 32;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;     // 0 random fp16 in [0.5, 2), 1 zeros, 2 random with the low 5 mantissa bits cleared
  const double secs = argc > 3 ? atof(argv[3]) : 2.5;
  uint4* in; float* out;
  hipMalloc(&in, 256 * 6 * 16); hipMalloc(&out, 4096 * 256 * 4);
  unsigned short* h = (unsigned short*)malloc(256 * 6 * 16);
  for (int i = 0; i < 256 * 6 * 8; ++i) {
    unsigned short v = (unsigned short)((rand() & 0x83ff) | 0x3800 | ((rand() & 1) << 10));
    if (mode == 1) v = 0;
    if (mode == 2) v &= 0xffe0;
    h[i] = v;
  }
  hipMemcpy(in, h, 256 * 6 * 16, hipMemcpyHostToDevice);
  const int blocks = 1024, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&](int it) { if (shape == 32) k<32><<<blocks, 256>>>(in, out, it); else k<16><<<blocks, 256>>>(in, out, it); };
  launch(100);
  hipDeviceSynchronize();
  double total_ms = 0; int n = 0;
  while (total_ms < secs * 1e3) {
    hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); total_ms += ms; ++n;
  }
  const double per = shape == 32 ? 8 * 2.0 * 32 * 32 * 16 : 16 * 2.0 * 16 * 16 * 32;
  const double fl = (double)blocks * 4 * iters * per * n;
  printf("mfma %dx%d f16, data mode %d: %.1f TFLOP/s over %.1f s\n", shape, shape, mode, fl / total_ms / 1e9, total_ms / 1e3);
  return 0;
}
