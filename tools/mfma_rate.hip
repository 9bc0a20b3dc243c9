// Sustained MFMA rate on the whole chip with REAL (random) operands: v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16 on the
// same 128 x 64 per-wave tile (8 x 4 resp. 4 x 2 accumulator tiles, 128 accumulator registers), 2 waves per SIMD, no memory
// traffic in the loop.  The matrix clock is power-limited on real data, and the 32x32 form reads half the operand-register
// bytes per flop -- does it sustain a higher rate?      hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void rate_kernel(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = src[(i * 64 + lane + blockIdx.x * 7) & 4095];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = src[((8 + i) * 64 + lane + blockIdx.x * 3) & 4095];
  float sum = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[4][8];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n], a[m], acc[n][m], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 8; ++m) sum += acc[n][m][0] + acc[n][m][1] + acc[n][m][2] + acc[n][m][3];
  } else {
    // 32x32x16: operands 8 bf16 per lane as well (lane = row 0..31, k-group 0..1); 4 x 2 tiles of 32 x 32; TWO k-steps per
    // iteration so that an iteration is the same 128 x 64 x 32 product as the 16x16x32 loop body
    f32x16 acc[2][4];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[n * 2 + ks], a[m * 2 + ks], acc[n][m], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[n][m][r];
  }
  if (sum == 12345.678f) out[0] = sum;  // keep the work alive
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int grid = argc > 2 ? atoi(argv[2]) : 256;
  std::vector<unsigned short> h(4096 * 8);
  unsigned s = 12345u;
  bf16x8* d; float* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int zero = 0; zero < 2; ++zero) {
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = zero ? 0 : (unsigned short)(0x3c00 + ((s >> 9) & 0x3ff) + ((s >> 31) << 15)); }
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int shape = 16; shape <= 32; shape += 16) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (shape == 16) hipLaunchKernelGGL(rate_kernel<16>, dim3(grid), dim3(512), 0, 0, d, o, iters);
        else hipLaunchKernelGGL(rate_kernel<32>, dim3(grid), dim3(512), 0, 0, d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * 128 * 64 * 32 * (double)iters * 8 * grid;
        if (rep) printf("%s operands  mfma %dx%d  %.3f ms  %.0f TF/s\n", zero ? "zero  " : "random", shape, shape, ms, fl / ms / 1e9);
      }
    }
  }
  return 0;
}
