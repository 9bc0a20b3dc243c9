// Hardware-semantics probes (test infrastructure compiled into the library so they travel to the GPU box).
// They execute raw MFMA / transpose-read / LDS-DMA instructions on host-supplied register images and dump the
// raw results, so the lane<->element maps the real kernels rely on are *measured* (tests/test_probes.py), not
// assumed from documentation.
#include "../common.h"

namespace {

__global__ void probe_mfma16_kernel(const bf16_t* a, const bf16_t* b, float* d, int n) {
  const int l = threadIdx.x;
  for (int r = 0; r < n; ++r) {
    bf16x8 av = *reinterpret_cast<const bf16x8*>(a + ((int64_t)r * 64 + l) * 8);
    bf16x8 bv = *reinterpret_cast<const bf16x8*>(b + ((int64_t)r * 64 + l) * 8);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
    *reinterpret_cast<f32x4*>(d + ((int64_t)r * 64 + l) * 4) = c;
  }
}

__global__ void probe_mfma32_kernel(const bf16_t* a, const bf16_t* b, float* d, int n) {
  const int l = threadIdx.x;
  for (int r = 0; r < n; ++r) {
    bf16x8 av = *reinterpret_cast<const bf16x8*>(a + ((int64_t)r * 64 + l) * 8);
    bf16x8 bv = *reinterpret_cast<const bf16x8*>(b + ((int64_t)r * 64 + l) * 8);
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) d[((int64_t)r * 64 + l) * 16 + i] = c[i];
  }
}

// LDS image of 8192 bf16 (16 KiB); each run: lane reads ds_read_b64_tr_b16 at byte address addr[run][lane]
__global__ void probe_tr16_kernel(const bf16_t* img, const int* addr, bf16_t* out, int n) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[8192];
  const int l = threadIdx.x;
  for (int i = l; i < 8192; i += 64) lds[i] = img[i];
  __syncthreads();
  for (int r = 0; r < n; ++r) {
    const int a = addr[r * 64 + l];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + a));
    *reinterpret_cast<s16x4*>(out + ((int64_t)r * 64 + l) * 4) = v;
  }
}

// One wave issues global_load_lds_dwordx4 from src + src_off[lane] to LDS base `lds_base` (bytes); the 16 KiB LDS
// (pre-filled with 0xFFFF) is dumped.
__global__ void probe_glds_kernel(const char* src, const int* src_off, int lds_base, unsigned short* dump) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  const int l = threadIdx.x;
  for (int i = l; i < 8192; i += 64) lds[i] = 0xFFFF;
  __syncthreads();
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + src_off[l]),
                                   (__attribute__((address_space(3))) void*)((char*)lds + lds_base), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 8192; i += 64) dump[i] = lds[i];
}

// v_mfma_scale_f32_16x16x128_f8f6f4 with both operands fp8 (e4m3): lane l supplies 8 dwords (32 fp8) of A and of B plus one
// dword of E8M0 scales each (byte 0 is used); raw accumulator dump.
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
__global__ void probe_mfma_f8_kernel(const int* a, const int* b, const int* sa, const int* sb, float* d, int n) {
  const int l = threadIdx.x;
  for (int r = 0; r < n; ++r) {
    i32x8_t av, bv;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      av[i] = a[((int64_t)r * 64 + l) * 8 + i];
      bv[i] = b[((int64_t)r * 64 + l) * 8 + i];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa[r * 64 + l], 0, sb[r * 64 + l]);
    *reinterpret_cast<f32x4*>(d + ((int64_t)r * 64 + l) * 4) = c;
  }
}

// What the matrix cores sustain with nothing else going on: every wave issues v_mfma_f32_16x16x32_bf16 back to back on register
// operands (4 x 4 fragments of the caller's image -> 16 independent accumulators, the register tile of a GEMM wave), no LDS, no
// global memory in the loop.  clk[2 * workgroup] = s_memtime ticks (the clock the shader ran at), clk[2 * workgroup + 1] =
// s_memrealtime ticks (100 MHz) over the loop.  The caller times the launch.  (Round 4: the GEMMs of the step run with the package
// at its power limit -- this loop is the ceiling that limit leaves, see DESIGN.md "Power".)
__global__ __launch_bounds__(256) void probe_mfma_rate_kernel(const bf16x8* __restrict__ in, float* out, unsigned long long* clk, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = in[i * 64 + lane];
    b[i] = in[(4 + i) * 64 + lane];
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
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = c1 - c0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

// A stand-in for a collective's kernel on the GPU it shares with backward: `workgroups` x 256 threads that each hold a CU slot (96 KiB of
// LDS: two of them do not fit one CU) and copy their own slab of `buf` back and forth -- light HBM traffic, no matrix work -- until
// `ticks` of the 100 MHz clock have passed.  tools/cu_contention_ab.py runs it beside the NT GEMM launches to measure what CUs held by
// another kernel cost the persistent and the one-tile-per-workgroup forms (distributed.share_cus_with_collectives).
__global__ __launch_bounds__(256) void probe_occupy_kernel(float* buf, long long slab_floats, unsigned long long ticks, unsigned long long* when) {
  extern __shared__ float hold[];  // 96 KiB
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float* mine = buf + (long long)blockIdx.x * slab_floats;
  hold[threadIdx.x] = 0.f;
  unsigned long long now = t0;
  while (now - t0 < ticks) {
    for (long long i = threadIdx.x; i < slab_floats / 2; i += 256) {
      const float v = mine[i];
      mine[slab_floats / 2 + i] = v + hold[(i + threadIdx.x) & 16383];
    }
    now = __builtin_amdgcn_s_memrealtime();
  }
  if (threadIdx.x == 0 && when) {
    when[2 * blockIdx.x] = t0;
    when[2 * blockIdx.x + 1] = now;
  }
}

// Throughput of LDS atomics: 8 waves of one workgroup per CU, each wave `iters` x 16 wave-level atomic adds on its own conflict-free 64-word
// rows (mode 0: ds_add_f32, 1: ds_add_u32, 2: plain ds_write_b32 as the yardstick).  clk[block] = shader cycles of wave 0's loop.
__global__ __launch_bounds__(512) void probe_lds_atomic_kernel(float* out, unsigned long long* clk, int iters, int mode) {
  __shared__ float buf[8 * 16 * 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* mine = buf + w * 16 * 64 + lane;
  for (int i = threadIdx.x; i < 8 * 16 * 64; i += 512) buf[i] = 0.f;
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (mode == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) atomicAdd(mine + j * 64, 1.0f);
    } else if (mode == 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j) atomicAdd(reinterpret_cast<unsigned*>(mine) + j * 64, 1u);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) reinterpret_cast<volatile float*>(mine)[j * 64] = (float)it;
    }
  }
  __syncthreads();
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
  out[(long long)blockIdx.x * 512 + threadIdx.x] = buf[threadIdx.x];
}

}  // namespace

extern "C" {

// out: workgroups x 512 floats; clk: workgroups uint64 (shader cycles of the loop).  Per wave-level instruction: clk / (iters * 16 * 8).
int op_probe_lds_atomic(float* out, void* clk, int workgroups, int iters, int mode, void* stream) {
  OP_CHECK_ARG(out && clk && workgroups > 0 && iters > 0 && mode >= 0 && mode <= 2, "probe_lds_atomic: bad argument");
  hipLaunchKernelGGL(probe_lds_atomic_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, out, (unsigned long long*)clk, iters, mode);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// buf: workgroups x slab_floats floats (each workgroup copies the first half of its slab onto the second); when: nullable, workgroups x 2
// uint64 = start / end of each workgroup in 100 MHz ticks.
int op_probe_occupy(void* buf, long long slab_floats, int workgroups, long long micros, void* when, void* stream) {
  OP_CHECK_ARG(buf && slab_floats >= 512 && workgroups > 0 && micros > 0, "probe_occupy: bad argument");
  OP_ENSURE_LDS(probe_occupy_kernel, 96 * 1024, "probe_occupy");
  hipLaunchKernelGGL(probe_occupy_kernel, dim3(workgroups), dim3(256), 96 * 1024, (hipStream_t)stream, (float*)buf, slab_floats,
                     (unsigned long long)micros * 100ull, (unsigned long long*)when);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// operands: 8 fragments x 64 lanes x 8 bf16 (8 KiB); out: workgroups x 256 floats; clk: workgroups x 2 uint64.  64 MFMAs per wave and
// iteration: flops = workgroups x 4 waves x iters x 64 x 16 384.
int op_probe_mfma_rate(const void* operands, float* out, void* clk, int workgroups, int iters, void* stream) {
  OP_CHECK_ARG(operands && out && clk && workgroups > 0 && iters > 0, "probe_mfma_rate: bad argument");
  hipLaunchKernelGGL(probe_mfma_rate_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)operands, out,
                     (unsigned long long*)clk, iters);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

int op_probe_mfma_f8(const void* a, const void* b, const void* sa, const void* sb, float* d, int n, void* stream) {
  hipLaunchKernelGGL(probe_mfma_f8_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)a, (const int*)b, (const int*)sa,
                     (const int*)sb, d, n);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

int op_probe_mfma16(const void* a, const void* b, float* d, int n, void* stream) {
  hipLaunchKernelGGL(probe_mfma16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, d, n);
  OP_LAUNCH_CHECK();
  return OP_OK;
}
int op_probe_mfma32(const void* a, const void* b, float* d, int n, void* stream) {
  hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, d, n);
  OP_LAUNCH_CHECK();
  return OP_OK;
}
int op_probe_tr16(const void* img, const int* addr, void* out, int n, void* stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)img, addr, (bf16_t*)out, n);
  OP_LAUNCH_CHECK();
  return OP_OK;
}
int op_probe_glds(const void* src, const int* src_off, int lds_base, void* dump, void* stream) {
  hipLaunchKernelGGL(probe_glds_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const char*)src, src_off, lds_base,
                     (unsigned short*)dump);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

}  // extern "C"
