// Microbenchmark: what does the ACCESS PATTERN of the fused bottleneck tail's epilogue (bneck_x3.hpp: a wave owns 32 pixels of an
// 8 x 28 tile of a 56 x 56 x 256-channel f32 map and walks the channels in four 64-channel chunks, 16 bytes per lane per instruction)
// cost against a plain coalesced copy of the same bytes?  Pure loads + stores, no arithmetic; one persistent workgroup per CU.
//   mode 0  coalesced copy: every wave instruction moves 1 KB of consecutive bytes
//   mode 1  the kernel's pattern: lane -> (pixel = lane & 31, 16 B at channel 64c + 32rb + 8q + 4(lane >> 5)), chunk after chunk
//   mode 2  as 1, but all four chunks of a pixel group back to back (1 KB per pixel touched within one burst of 32 loads)
//   mode 3  256-byte segments like mode 1, but a wave instruction covers 4 pixels x 256 B (16 consecutive lanes per pixel)
//   mode 4  as 1 with 128-channel chunks (512-byte segments): two chunks of 16 instructions
//   mode 5  loads as mode 1 (scattered), stores as mode 3 (4 pixels x 256 B per instruction)
//   mode 6  loads as mode 3, stores as mode 1
//   mode 7  128-byte segments: a wave instruction covers 8 pixels x 128 B (8 consecutive lanes per pixel), 32-channel steps
//   mode 8  loads: lane -> (pixel = lane & 31, 16 B at channel 32rb + 16(lane >> 5) + 4q); stores after a 4 x 4 lane transpose:
//           instruction i of a 32-channel row block: lanes 4a .. 4a+3 of lane half h write 64 consecutive bytes of pixel 4a + i
// build: hipcc -O3 --offload-arch=gfx950 -o tools/lab/micro/access_pattern.out tools/lab/micro/access_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int H = 56, W = 56, C = 256, TH = 8, TW = 28, NW = 7;

__device__ inline size_t pixel_of(int frame, int ty, int tx, int p) {
  const int r = p / TW, c = p - r * TW;
  return ((size_t)frame * H + ty * TH + r) * W + tx * TW + c;
}

template <int MODE>
__global__ __launch_bounds__(448) void k(const float4* __restrict__ x, float4* __restrict__ y, int frames) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tiles = frames * (H / TH) * (W / TW);
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int frame = t / 14, rem = t - frame * 14, ty = rem >> 1, tx = rem & 1;
    if (MODE == 0) {
      // 8 rows of 28 KB: wave w copies 4 KB pieces round robin
      for (int r = 0; r < TH; ++r) {
        const size_t base = (((size_t)frame * H + ty * TH + r) * W + tx * TW) * (C / 4);   // float4 units; 28 px * 64 = 1792 float4
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = x[base + (wave * 4 + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) y[base + (wave * 4 + i) * 64 + lane] = v[i];
      }
    } else if (MODE == 1 || MODE == 2 || MODE == 4) {
      const size_t px = pixel_of(frame, ty, tx, wave * 32 + (lane & 31)) * (C / 4);
      const int half = lane >> 5;
      constexpr int NCH = MODE == 1 ? 4 : MODE == 4 ? 2 : 1, PER = 32 / NCH;   // instructions per chunk
      for (int c = 0; c < NCH; ++c) {
        float4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] = x[px + (c * PER + i) * 2 + half];
#pragma unroll
        for (int i = 0; i < PER; ++i) y[px + (c * PER + i) * 2 + half] = v[i];
      }
    } else if (MODE == 5 || MODE == 6) {
      const size_t px = pixel_of(frame, ty, tx, wave * 32 + (lane & 31)) * (C / 4);
      const int half = lane >> 5;
      for (int c = 0; c < 4; ++c) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const size_t scat = px + (c * 8 + i) * 2 + half, coal = pixel_of(frame, ty, tx, wave * 32 + i * 4 + (lane >> 4)) * (C / 4) + c * 16 + (lane & 15);
          v[i] = x[MODE == 5 ? scat : coal];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const size_t scat = px + (c * 8 + i) * 2 + half, coal = pixel_of(frame, ty, tx, wave * 32 + i * 4 + (lane >> 4)) * (C / 4) + c * 16 + (lane & 15);
          y[MODE == 5 ? coal : scat] = v[i];
        }
      }
    } else if (MODE == 7) {
      for (int c = 0; c < 8; ++c) {
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = x[pixel_of(frame, ty, tx, wave * 32 + i * 8 + (lane >> 3)) * (C / 4) + c * 8 + (lane & 7)];
#pragma unroll
        for (int i = 0; i < 4; ++i) y[pixel_of(frame, ty, tx, wave * 32 + i * 8 + (lane >> 3)) * (C / 4) + c * 8 + (lane & 7)] = v[i];
      }
    } else if (MODE == 8) {
      const int half = lane >> 5, a = (lane & 31) >> 2, kk = lane & 3;
      const size_t px = pixel_of(frame, ty, tx, wave * 32 + (lane & 31)) * (C / 4);
      for (int rb = 0; rb < 8; ++rb) {
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = x[px + rb * 8 + half * 4 + q];
#pragma unroll
        for (int i = 0; i < 4; ++i) y[pixel_of(frame, ty, tx, wave * 32 + 4 * a + i) * (C / 4) + rb * 8 + half * 4 + kk] = v[i];
      }
    } else if (MODE == 3) {
      for (int c = 0; c < 4; ++c) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = x[pixel_of(frame, ty, tx, wave * 32 + i * 4 + (lane >> 4)) * (C / 4) + c * 16 + (lane & 15)];
#pragma unroll
        for (int i = 0; i < 8; ++i) y[pixel_of(frame, ty, tx, wave * 32 + i * 4 + (lane >> 4)) * (C / 4) + c * 16 + (lane & 15)] = v[i];
      }
    }
  }
}

template <int MODE>
void run(const float4* x, float4* y, int frames, int blocks, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 448>>>(x, y, frames);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) k<MODE><<<blocks, 448>>>(x, y, frames);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  const double bytes = 2.0 * frames * H * W * C * 4;
  printf("mode %d (%s), %d workgroups: %.3f ms  %.2f TB/s (read + write) = %.1f B/clk/CU\n", MODE, name, blocks, ms, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / (blocks < 256 ? blocks : 256) / 2.1e9);
}

int main(int argc, char** argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 448;
  const size_t n = (size_t)frames * H * W * C * 4;
  float4 *x, *y;
  hipMalloc(&x, n); hipMalloc(&y, n);
  hipMemset(x, 1, n); hipMemset(y, 0, n);
  // fewer workgroups than CUs: what ONE CU can move with a pattern when HBM is not the limit (B/clk/CU at 2.1 GHz in the last column)
  for (int blocks : {16, 64}) {
    run<0>(x, y, frames / 8, blocks, "coalesced copy");
    run<1>(x, y, frames / 8, blocks, "bneck epilogue pattern, 64-channel chunks");
    run<5>(x, y, frames / 8, blocks, "scattered loads, 4-pixel stores");
    run<6>(x, y, frames / 8, blocks, "4-pixel loads, scattered stores");
    run<8>(x, y, frames / 8, blocks, "scattered loads, stores 64 B per 4 lanes");
  }
  for (int blocks : {256}) {
    run<0>(x, y, frames, blocks, "coalesced copy");
    run<1>(x, y, frames, blocks, "bneck epilogue pattern, 64-channel chunks");
    run<4>(x, y, frames, blocks, "bneck pattern, 128-channel chunks");
    run<2>(x, y, frames, blocks, "bneck pattern, whole pixel per burst");
    run<3>(x, y, frames, blocks, "256-B segments, 4 pixels per instruction");
    run<5>(x, y, frames, blocks, "scattered loads, 4-pixel stores");
    run<6>(x, y, frames, blocks, "4-pixel loads, scattered stores");
    run<7>(x, y, frames, blocks, "128-B segments, 8 pixels per instruction");
    run<8>(x, y, frames, blocks, "scattered loads, stores 64 B per 4 lanes (two lane halves complete a line)");
  }
  return 0;
}
