// Microbenchmark (round 6): what does a BYTE cost on this chip, by where it comes from?  Persistent workgroups stream memory for a few seconds while
// rocm-smi is sampled beside them (byte_energy.sh): package power at a known byte rate, against the same grid spinning without memory traffic.
//   mode 0  HBM read:   every workgroup sums a 4 GiB buffer, 1 KiB per wave instruction (the sum is written once at the end)
//   mode 1  HBM write:  every workgroup fills its share of a 4 GiB buffer
//   mode 2  HBM copy:   read + write
//   mode 3  L2 read:    every workgroup re-reads its own 64 KiB (16 MiB in all: L2-resident)
//   mode 4  no memory:  the same grid doing register adds (what the chip draws busy-idle at this occupancy)
// usage: byte_energy.out <mode> <seconds>
// build: hipcc -O3 --offload-arch=gfx950 -o tools/lab/micro/byte_energy.out tools/lab/micro/byte_energy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ x, float4* __restrict__ y, size_t n4, float* __restrict__ out, int reps) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == 3) {
    const float4* mine = x + (size_t)blockIdx.x * 4096;                 // 64 KiB per workgroup
    for (int r = 0; r < reps; ++r)
#pragma unroll 4
      for (int i = threadIdx.x; i < 4096; i += 256) { const float4 v = mine[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  } else if (MODE == 4) {
    for (int r = 0; r < reps * 262144; ++r) { acc.x += 1.0f; acc.y += acc.x; acc.z += acc.y; acc.w += acc.z; }
  } else {
    for (int r = 0; r < reps; ++r)
#pragma unroll 4
      for (size_t i = tid; i < n4; i += nthr) {
        if (MODE == 0) { const float4 v = x[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        if (MODE == 1) y[i] = make_float4((float)r, 1.f, 2.f, 3.f);
        if (MODE == 2) y[i] = x[i];
      }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;   // never true: keeps the loads
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 3.0;
  const size_t bytes = (size_t)4 << 30, n4 = bytes / 16;
  float4 *x, *y; float* out;
  if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  (void)hipMemset(x, 0, bytes); (void)hipMemset(y, 0, bytes);
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount * 8;
  auto launch = [&](int reps) {
    switch (mode) {
      case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, x, y, n4, out, reps); break;
      case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, x, y, n4, out, reps); break;
      case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, x, y, n4, out, reps); break;
      case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, x, y, n4, out, reps * 64); break;
      default: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, x, y, n4, out, reps); break;
    }
  };
  const int reps = 4;
  launch(reps); (void)hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  double el = 0;
  while (el < secs) {
    launch(reps); (void)hipDeviceSynchronize(); ++launches;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  double moved = 0;
  if (mode == 0 || mode == 1) moved = (double)bytes * reps * launches;
  if (mode == 2) moved = 2.0 * bytes * reps * launches;
  if (mode == 3) moved = (double)grid * 65536.0 * 64 * reps * launches;
  printf("byte_energy mode %d: %.2f s, %ld launches, %.1f GB moved, %.0f GB/s\n", mode, el, launches, moved / 1e9, moved / 1e9 / el);
  return 0;
}
