// Shared device/host helpers for the ONE-PEACE gfx950 (CDNA4) kernels.
// Wavefront = 64 lanes everywhere; nothing here is portable to 32-wide hardware on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define OP_OK 0
#define OP_EINVAL (-22)
#define OP_ENOTSUP (-95)

// dtype codes of the C ABI
#define OP_DT_BF16 0
#define OP_DT_F32 1

extern "C" void op_set_error(const char* fmt, ...);

#define OP_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      op_set_error(__VA_ARGS__);           \
      return OP_EINVAL;                    \
    }                                      \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, DEVICE): the attribute is per device, and a process may
// drive more than one (the deployment model is one process per GPU, but nothing here may rely on it).
#define OP_ENSURE_LDS(kernel, bytes, what)                                                                          \
  do {                                                                                                              \
    static unsigned long long done_ = 0;                                                                            \
    int dev_ = 0;                                                                                                   \
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = 0;                                                                \
    if (!((done_ >> (dev_ & 63)) & 1ull)) {                                                                         \
      hipError_t e_ = hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
      if (e_ != hipSuccess) { op_set_error(what ": hipFuncSetAttribute failed: %s", hipGetErrorString(e_)); return (int)e_; } \
      done_ |= 1ull << (dev_ & 63);                                                                                 \
    }                                                                                                               \
  } while (0)

#define OP_LAUNCH_CHECK()                                                  \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      op_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,          \
                   hipGetErrorString(e__));                                \
      return (int)e__;                                                     \
    }                                                                      \
  } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---- per-row e4m3 quantisation (csrc/fp8.hip; fused outputs of the LayerNorm kernels) -------------
// scale = amax / 448 (1 for an all-zero row), q = e4m3(clamp(x / scale)): ONE definition for the stand-alone pass and the fused outputs.
#define OP_FP8_MAX 448.0f
__device__ __forceinline__ float fp8_row_scale(float amax) { return amax > 0.f ? amax * (1.0f / OP_FP8_MAX) : 1.0f; }
__device__ __forceinline__ u32x2 fp8_pack8(const float (&v)[8], float inv) {
  float e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = fminf(fmaxf(v[j] * inv, -OP_FP8_MAX), OP_FP8_MAX);  // (the clamp guards the rounding of amax * inv)
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(e[0], e[1], w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(e[2], e[3], w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(e[4], e[5], w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(e[6], e[7], w1, true);
  return (u32x2){(unsigned)w0, (unsigned)w1};
}

// ---- 8-element vector load/store with fp32 math --------------------------------------------------
template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> {
  typedef bf16x8 raw_t;  // load now, convert at first use (keeps prefetched rows as plain loads in flight)
  // -DOP_NT_STREAM (A/B builds of tools/): non-temporal hint on the streaming loads / stores of the row-wise kernels
#ifdef OP_NT_STREAM
  static __device__ __forceinline__ raw_t ldraw(const bf16_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p)); }
#else
  static __device__ __forceinline__ raw_t ldraw(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
#endif
  // non-temporal forms for the passes that stream hundreds of MB once (training-only kernels: op_ln_geglu_fwd / _bwd; +2.5 ... 3 % on
  // them at the headline shapes and nothing of theirs left in L2 / Infinity Cache in front of the next GEMM's panels).  NOT the default:
  // on inference-size tensors the same hint costs 1 ... 5 % (config 1: the next launch finds its input in the caches).
  static __device__ __forceinline__ raw_t ldraw_nt(const bf16_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p)); }
  static __device__ __forceinline__ void store_nt(bf16_t* p, const float (&v)[8]) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16_t)v[i];
    __builtin_nontemporal_store(r, reinterpret_cast<bf16x8*>(p));
  }
  static __device__ __forceinline__ void cvt(const raw_t& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
  }
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    bf16x8 r = ldraw(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16_t)v[i];
#ifdef OP_NT_STREAM
    __builtin_nontemporal_store(r, reinterpret_cast<bf16x8*>(p));
#else
    *reinterpret_cast<bf16x8*>(p) = r;
#endif
  }
};
template <> struct Vec8<float> {
  struct raw_t { f32x4 a, b; };
  static __device__ __forceinline__ raw_t ldraw(const float* p) {
    raw_t r;
    r.a = *reinterpret_cast<const f32x4*>(p);
    r.b = *reinterpret_cast<const f32x4*>(p + 4);
    return r;
  }
  static __device__ __forceinline__ raw_t ldraw_nt(const float* p) {
    raw_t r;
    r.a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    r.b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
    return r;
  }
  static __device__ __forceinline__ void store_nt(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(p));
    __builtin_nontemporal_store(b, reinterpret_cast<f32x4*>(p + 4));
  }
  static __device__ __forceinline__ void cvt(const raw_t& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = r.a[i]; v[4 + i] = r.b[i]; }
  }
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
};

// streaming access with the cache policy chosen at compile time (NT: see Vec8<bf16_t>::ldraw_nt)
template <typename T, bool NT> __device__ __forceinline__ typename Vec8<T>::raw_t ldraw_sel(const T* p) {
  if constexpr (NT) return Vec8<T>::ldraw_nt(p);
  else return Vec8<T>::ldraw(p);
}
template <typename T, bool NT> __device__ __forceinline__ void store_sel(T* p, const float (&v)[8]) {
  if constexpr (NT) Vec8<T>::store_nt(p, v);
  else Vec8<T>::store(p, v);
}

// Exact-erf GELU pieces: Phi(x) = 0.5 (1 + erf(x / sqrt 2)) and phi(x) = exp(-x^2/2) / sqrt(2 pi) from ONE exponential
// (Abramowitz-Stegun 7.1.26: erfc(u) = t (a1 + t (a2 + ... a5 t)) e^{-u^2}, t = 1 / (1 + p u), |error| <= 1.5e-7 -- four
// orders of magnitude below bf16 resolution; e^{-u^2} with u = |x| / sqrt 2 is exactly the exponential phi needs).  About
// a third of the VALU work of erff() + expf(): the GeGLU epilogue and its backward are VALU-bound on 224 M elements/layer.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  const float u = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * u);
  const float e = __expf(-u * u);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float h = 0.5f * poly * e;  // 0.5 erfc(|u|)
  cdf = x >= 0.f ? 1.0f - h : h;
  pdf = 0.39894228040143267794f * e;
}
__device__ __forceinline__ float gelu_erf(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return x * cdf;
}
// d/dx gelu(x) = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return cdf + x * pdf;
}

// out[n] = (accumulate ? out[n] : 0) + mul[n] * sum_p part[p * pstride + n].  256 threads = 32 columns x 8 part
// groups (independent loads, LDS fold) so the fold is not one long dependent chain per column.
template <typename T>
__global__ __launch_bounds__(256) void partials_reduce_kernel(const float* __restrict__ part, int parts, int64_t pstride,
                                                              int N, const bf16_t* __restrict__ mul, T* __restrict__ out,
                                                              int accumulate) {
  __shared__ float red[8][33];
  const int c = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c;
  float a = 0.f;
  if (n < N) {
#pragma unroll 8
    for (int p = pg; p < parts; p += 8) a += part[(int64_t)p * pstride + n];
  }
  red[pg][c] = a;
  __syncthreads();
  if (pg == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][c];
    if (mul) t *= (float)mul[n];
    out[n] = (T)(t + (accumulate ? (float)out[n] : 0.f));
  }
}

// Up to three such folds in ONE launch (gridDim.y = jobs; a job with a null `out` is skipped): the dw/db pair of a
// LayerNorm backward, dgamma/dbias of a residual branch, the q/k/v bias segments of a fused projection.
template <typename T>
__global__ __launch_bounds__(256) void partials_reduce3_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                               const float* __restrict__ p2, const bf16_t* __restrict__ m0,
                                                               const bf16_t* __restrict__ m1, const bf16_t* __restrict__ m2,
                                                               T* __restrict__ o0, T* __restrict__ o1, T* __restrict__ o2,
                                                               int parts, int64_t pstride, int N, int accumulate,
                                                               float* __restrict__ f2 = nullptr) {  // f2: job 2 in fp32, overwritten
  __shared__ float red[8][33];
  const int job = blockIdx.y;
  const float* part = job == 0 ? p0 : (job == 1 ? p1 : p2);
  const bf16_t* mul = job == 0 ? m0 : (job == 1 ? m1 : m2);
  T* out = job == 0 ? o0 : (job == 1 ? o1 : o2);
  float* outf = job == 2 ? f2 : nullptr;
  if (out == nullptr && outf == nullptr) return;  // uniform
  const int c = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c;
  float a = 0.f;
  if (n < N) {
#pragma unroll 8
    for (int p = pg; p < parts; p += 8) a += part[(int64_t)p * pstride + n];
  }
  red[pg][c] = a;
  __syncthreads();
  if (pg == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][c];
    if (mul) t *= (float)mul[n];
    if (outf) outf[n] = t;
    else out[n] = (T)(t + (accumulate ? (float)out[n] : 0.f));
  }
}

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
