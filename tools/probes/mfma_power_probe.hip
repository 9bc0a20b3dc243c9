// Round 4: what the matrix cores of an MI355X sustain with NOTHING else going on -- no LDS, no global memory: every wave issues
// v_mfma_f32_16x16x32_bf16 back to back on register operands (4 x 4 fragments -> 16 independent accumulators, the register tile of
// a GEMM wave) for ~1 s.  Reports TFLOP/s and the clock the shader ran at (s_memtime ticks / s_memrealtime).  The question behind it:
// gemm256w_tn_grouped_kernel did not get faster when 2 % of its time (the epilogue) was removed, and rocm-smi shows the package at
// its 1400 W limit during the launch: is the dense-bf16 figure of the data sheet (2.5 PFLOP/s at 2.4 GHz) reachable at that limit at all?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_power_probe tools/probes/mfma_power_probe.hip
//   tools/probes/mfma_power_probe [waves per CU = 4] [data: 0 random normal, 1 zeros, 2 constant 1.0] [seconds = 1.0] [shape 16 | 32]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void mfma_loop(const bf16x8* __restrict__ in, float* out, unsigned long long* clk, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(i * 64 + lane)];
    b[i] = in[((4 + i) * 64 + lane)];
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = c1 - c0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

typedef __attribute__((ext_vector_type(16))) float f32x16;
// the same with v_mfma_f32_32x32x16_bf16 (32 768 flops per instruction, half the instructions and operand reads per flop): 2 x 2 fragments
__global__ __launch_bounds__(256) void mfma_loop32(const bf16x8* __restrict__ in, float* out, unsigned long long* clk, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(i * 64 + lane)];
    b[i] = in[((4 + i) * 64 + lane)];
  }
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * (rep & 1)], b[j], acc[i][j], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = c1 - c0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

static unsigned short f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}

int main(int argc, char** argv) {
  const int waves_per_cu = argc > 1 ? atoi(argv[1]) : 4;
  const int data = argc > 2 ? atoi(argv[2]) : 0;
  const double seconds = argc > 3 ? atof(argv[3]) : 1.0;
  const int shape = argc > 4 ? atoi(argv[4]) : 16;  // 16: v_mfma_f32_16x16x32_bf16, 32: v_mfma_f32_32x32x16_bf16
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int wgs = cus * ((waves_per_cu + 3) / 4);
  std::vector<unsigned short> h(8 * 64 * 8);
  srand(1);
  for (auto& v : h) {
    float x = 0.f;
    if (data == 0) {  // sum of 12 uniforms - 6: normal enough
      for (int k = 0; k < 12; ++k) x += (float)rand() / (float)RAND_MAX;
      x = (x - 6.f) * 0.25f;
    } else if (data == 2) x = 1.0f;
    v = f2bf(x);
  }
  bf16x8* din;
  float* dout;
  unsigned long long* dclk;
  hipMalloc(&din, h.size() * 2);
  hipMalloc(&dout, (size_t)wgs * 256 * 4);
  hipMalloc(&dclk, (size_t)wgs * 16);
  hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  int iters = 20000;
  float ms = 0.f;
  for (int pass = 0; pass < 3; ++pass) {  // pass 0 calibrates, pass 1 warms the package up, pass 2 is reported
    hipEventRecord(e0);
    if (shape == 32) hipLaunchKernelGGL(mfma_loop32, dim3(wgs), dim3(256), 0, 0, din, dout, dclk, iters);
    else hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 0, 0, din, dout, dclk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (pass == 0) iters = (int)(iters * (seconds * 1e3 / ms));
  }
  std::vector<unsigned long long> clk(2 * wgs);
  hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost);
  double mhz_sum = 0, mhz_min = 1e9, mhz_max = 0;
  for (int i = 0; i < wgs; ++i) {
    const double mhz = (double)clk[2 * i] / ((double)clk[2 * i + 1] * 0.01);
    mhz_sum += mhz;
    if (mhz < mhz_min) mhz_min = mhz;
    if (mhz > mhz_max) mhz_max = mhz;
  }
  const double flops = (double)wgs * 4 * iters * (shape == 32 ? 32.0 : 64.0) * 2.0 * (shape == 32 ? 32 * 32 * 16 : 16 * 16 * 32);
  const double tf = flops / (ms * 1e-3) / 1e12;
  const double mhz = mhz_sum / wgs;
  // one v_mfma_f32_16x16x32_bf16 = 16384 flops per wave; at 4 SIMDs per CU the data-sheet rate (2.5 PFLOP/s at 2400 MHz over 256 CUs)
  // is 16384 flops per 16 ... cycles: report the fraction of the issue slots at the MEASURED clock instead of assuming it
  const double peak_at_clock = 2500.0 * mhz / 2400.0 * cus / 256.0;
  printf("mfma %s  CUs %d  workgroups %d (%d waves per CU)  data %s  %.1f ms: %.0f TFLOP/s; shader clock mean %.0f MHz (min %.0f max %.0f); "
         "data-sheet rate at that clock %.0f TFLOP/s -> %.3f of the issue slots\n",
         shape == 32 ? "32x32x16" : "16x16x32", cus, wgs, 4 * ((waves_per_cu + 3) / 4), data == 0 ? "normal" : data == 1 ? "zeros" : "ones", ms, tf, mhz, mhz_min, mhz_max, peak_at_clock,
         tf / peak_at_clock);
  return 0;
}
