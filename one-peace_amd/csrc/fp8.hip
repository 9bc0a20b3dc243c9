// fp8 (OCP e4m3) variant of the FFN GEMMs for gfx950 -- BASELINE configs[4] ("fp8 MFMA GEMMs"), an explicit opt-in.
//
// Replaces, when enabled, the forward GEMMs of transformer_layer.py:54-67,149-157 (GeGLU up-projection x W0^T / x W1^T and
// the down-projection W2) -- the reference has no fp8 path; parity is stated against the bf16 kernels of gemm.hip
// (tests/test_ops_gpu.py::test_fp8_*: rel-Frobenius <= 5e-2 of the bf16 result).
//
// Numerics: operands are quantised PER ROW (one fp32 scale per activation row / per weight row = output column:
// q = round_e4m3(x * 448 / amax_row)), products accumulate in fp32 on the matrix pipe, and the epilogue multiplies
// acc[m][n] by scale_a[m] * scale_b[n] before the usual bias / GeGLU / residual epilogue.  The MFMA is
// v_mfma_scale_f32_16x16x128_f8f6f4 (the MX instruction: the only fp8 form that runs at twice the bf16 rate on gfx950) with
// every E8M0 block scale = 2^0 -- the per-row fp32 scales carry the dynamic range, the block scales are not used.
// Lane map (tools/probe_f8.py, tests/test_probes_gpu.py): lane (g, t) supplies row t of its operand, 32 fp8 bytes; any
// K permutation used for BOTH operands gives the same sums -- here byte j of lane (g, t) = k-offset g*32 + j of the 128-deep step.
//
// Tiling: the 128 x 128 / 4-wave / two-LDS-buffer structure of gemm_nt_kernel with K-steps of 128 fp8 (the same 128-byte
// LDS rows, XOR swizzle and LDS-DMA staging; half the LDS and L2 bytes per flop of the bf16 kernel).
// Roofline: MFMA, dense fp8 peak 5 PFLOP/s; algorithmic work 2*M*N*K flops per launch.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
constexpr int E8M0_ONE = 0x7F7F7F7F;  // four block scales of 2^0
constexpr float FP8_MAX = 448.0f;     // largest finite e4m3fn value

enum { F8_EPI_BIAS = 0, F8_EPI_GEGLU = 2, F8_EPI_RESID = 3 };

struct Fp8Args {
  const uint8_t* A; int64_t lda; const float* sa;          // activations [M, K] fp8, row scales [M]
  const uint8_t* B[2]; int64_t ldb; const float* sb[2];    // weights [N, K] fp8 (GeGLU: W0, W1), row scales [N]
  const bf16_t* bias;
  bf16_t* C; int64_t ldc;
  bf16_t* H0; bf16_t* H1;
  const bf16_t* resid; int64_t ldr;
  const bf16_t* gamma; const float* rowscale; int rows_per_sample;
  int M, N, K;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ int xcd_remap8(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// LDS row p of the weight tile -> output column (relative to the tile) it feeds: same permutation as gemm_nt_kernel, so
// that a lane ends up with 16 (GeGLU: 8) contiguous output columns of one row.
template <int EPI>
__device__ __forceinline__ int w_row_to_col8(int p) {
  const int i = p & 15;
  if (EPI == F8_EPI_GEGLU) {
    const int pp = p & 63;  // rows 0..63 <- W0, 64..127 <- W1, same column map
    return (pp >> 5) * 32 + (i >> 2) * 8 + ((pp >> 4) & 1) * 4 + (i & 3);
  }
  return (p & ~63) + (i >> 2) * 16 + ((p >> 4) & 3) * 4 + (i & 3);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(const Fp8Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = 128 * 128;  // bytes per operand tile: 128 rows x 128 fp8
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int g = lane >> 4, t = lane & 15;
  constexpr int BN_OUT = (EPI == F8_EPI_GEGLU) ? 64 : 128;

  const int pid = xcd_remap8(blockIdx.x, gridDim.x);
  constexpr int GM = 8;
  const int per_group = GM * p.tiles_n;
  const int first_m = (pid / per_group) * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int in_group = pid % per_group;
  const int pid_m = first_m + in_group % gsz;
  const int pid_n = in_group / gsz;
  const int m0 = pid_m * 128, n0 = pid_n * BN_OUT;

  // staging: slot q = i*256 + tid -> LDS row q>>3, 16-byte slot q&7 (source chunk XOR-swizzled by row&7)
  const uint8_t* srcA[4];
  const uint8_t* srcB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + tid;
    const int row = q >> 3, c = (q & 7) ^ (row & 7);
    const int gm = min(m0 + row, p.M - 1);
    srcA[i] = p.A + (int64_t)gm * p.lda + c * 16;
    const int gn = min(n0 + w_row_to_col8<EPI>(row), p.N - 1);
    const uint8_t* base = (EPI == F8_EPI_GEGLU && row >= 64) ? p.B[1] : p.B[0];
    srcB[i] = base + (int64_t)gn * p.ldb + c * 16;
  }
  auto stage = [&](int buf) {
    char* la = smem + buf * (2 * TILE);
    char* lb = la + TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int wbase = (i * 256 + wid * 64) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcA[i],
                                       (__attribute__((address_space(3))) void*)(la + wbase), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcB[i],
                                       (__attribute__((address_space(3))) void*)(lb + wbase), 16, 0, 0);
      srcA[i] += 128;
      srcB[i] += 128;
    }
  };

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment = 32 bytes of row (.. + t) at k-offset g*32: 16-byte chunks 2g and 2g+1, each XOR (t & 7)
  const int sw0 = ((2 * g) ^ (t & 7)) << 4, sw1 = ((2 * g + 1) ^ (t & 7)) << 4;
  int rdW[4], rdX[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
    rdW[ni] = ((EPI == F8_EPI_GEGLU) ? ((ni >> 1) * 64 + wn * 32 + (ni & 1) * 16 + t) : (wn * 64 + ni * 16 + t)) * 128;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) rdX[mi] = (wm * 64 + mi * 16 + t) * 128;
  auto frag = [&](const char* base) {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(base + sw0);
    const u32x4 hi = *reinterpret_cast<const u32x4*>(base + sw1);
    i32x8 r = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    return r;
  };

  const int nk = p.K / 128;
  stage(0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt landed; everyone is done reading the other buffer
    if (kt + 1 < nk) stage((kt + 1) & 1);
    const char* la = smem + (kt & 1) * (2 * TILE);
    const char* lb = la + TILE;
    i32x8 wf[4], xf[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) wf[ni] = frag(lb + rdW[ni]);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) xf[mi] = frag(la + rdX[mi]);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0, E8M0_ONE, 0, E8M0_ONE);
  }

  // ---- epilogue: dequantise (row scale x column scale), then bias / GeGLU / residual ----
  const int mrow0 = m0 + wm * 64;
  float sa[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) sa[mi] = p.sa[min(mrow0 + mi * 16 + t, p.M - 1)];
  if (EPI == F8_EPI_GEGLU) {
    const int f0 = n0 + wn * 32 + g * 8;  // 8 contiguous f: acc[nl][mi][r] = h0, acc[2+nl][mi][r] = h1, f = f0 + nl*4 + r
    if (f0 >= p.N) return;
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = p.sb[0][f0 + j]; s1[j] = p.sb[1][f0 + j]; }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = mrow0 + mi * 16 + t;
      if (m >= p.M) continue;
      float go[8], h0[8], h1[8];
#pragma unroll
      for (int nl = 0; nl < 2; ++nl)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = acc[nl][mi][r] * (sa[mi] * s0[nl * 4 + r]), b = acc[2 + nl][mi][r] * (sa[mi] * s1[nl * 4 + r]);
          h0[nl * 4 + r] = a;
          h1[nl * 4 + r] = b;
          go[nl * 4 + r] = gelu_erf(a) * b;
        }
      Vec8<bf16_t>::store(p.C + (int64_t)m * p.ldc + f0, go);
      if (p.H0) {
        Vec8<bf16_t>::store(p.H0 + (int64_t)m * p.ldc + f0, h0);
        Vec8<bf16_t>::store(p.H1 + (int64_t)m * p.ldc + f0, h1);
      }
    }
    return;
  }
  const int nc0 = n0 + wn * 64 + g * 16;  // 16 contiguous columns: acc[ni][mi][r] <-> column ni*4 + r
  if (nc0 >= p.N) return;
  float sbv[16], bv[16], gv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int n = min(nc0 + j, p.N - 1);
    sbv[j] = p.sb[0][n];
    bv[j] = p.bias ? (float)p.bias[n] : 0.f;
    gv[j] = (EPI == F8_EPI_RESID && p.gamma) ? (float)p.gamma[n] : 1.f;
  }
  const bool second = nc0 + 8 < p.N;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = mrow0 + mi * 16 + t;
    if (m >= p.M) continue;
    float o[16];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[ni * 4 + r] = acc[ni][mi][r] * (sa[mi] * sbv[ni * 4 + r]) + bv[ni * 4 + r];
    if (EPI == F8_EPI_RESID) {
      const float rs = p.rowscale ? p.rowscale[m / p.rows_per_sample] : 1.f;
      float rv[16], tmp[8];
      Vec8<bf16_t>::load(p.resid + (int64_t)m * p.ldr + nc0, tmp);
#pragma unroll
      for (int j = 0; j < 8; ++j) rv[j] = tmp[j];
      if (second) {
        Vec8<bf16_t>::load(p.resid + (int64_t)m * p.ldr + nc0 + 8, tmp);
#pragma unroll
        for (int j = 0; j < 8; ++j) rv[8 + j] = tmp[j];
      }
      if (p.H0) {  // branch output y (pre layer-scale), needed by the backward pass for d gamma
        float lo[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { lo[j] = o[j]; hi[j] = o[8 + j]; }
        Vec8<bf16_t>::store(p.H0 + (int64_t)m * p.ldc + nc0, lo);
        if (second) Vec8<bf16_t>::store(p.H0 + (int64_t)m * p.ldc + nc0 + 8, hi);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) o[j] = rv[j] + rs * gv[j] * o[j];
    }
    float lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo[j] = o[j]; hi[j] = o[8 + j]; }
    bf16_t* C = p.C + (int64_t)m * p.ldc + nc0;
    Vec8<bf16_t>::store(C, lo);
    if (second) Vec8<bf16_t>::store(C + 8, hi);
  }
}

// Per-row e4m3 quantisation: q[r][c] = fp8(x[r][c] * 448 / amax_r), scale[r] = amax_r / 448 (1 for an all-zero row).
// One wavefront per row, 8 bf16 per lane per pass; rows of up to 8192 columns stay in registers between the two passes.
template <int MAXV>
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q,
                                                             int64_t ldq, float* __restrict__ scale, int64_t rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * ldx;
  float v[MAXV][8];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < cols) {
      Vec8<bf16_t>::load(xr + c, v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[i][j]));
    }
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax * (1.0f / FP8_MAX) : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  uint8_t* qr = q + row * ldq;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < cols) {
      int w0 = 0, w1 = 0;
      // (values are within +-448 by construction; clamp guards the rounding of amax * inv)
      float e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = fminf(fmaxf(v[i][j] * inv, -FP8_MAX), FP8_MAX);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(e[0], e[1], w0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(e[2], e[3], w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(e[4], e[5], w1, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(e[6], e[7], w1, true);
      *reinterpret_cast<u32x2*>(qr + c) = (u32x2){(unsigned)w0, (unsigned)w1};
    }
  }
}

template <int EPI>
int launch_fp8(const Fp8Args& a, hipStream_t s) {
  const size_t sh = 4 * 128 * 128;
  OP_ENSURE_LDS((gemm_fp8_kernel<EPI>), (int)sh, "gemm_fp8");
  hipLaunchKernelGGL((gemm_fp8_kernel<EPI>), dim3(a.tiles_m * a.tiles_n), dim3(256), sh, s, a);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

}  // namespace

extern "C" int op_prof_begin(int family, double work, void* stream);
extern "C" void op_prof_end(int slot, void* stream);

extern "C" {

// x [rows, cols] bf16 (row stride ldx) -> q [rows, cols] fp8 e4m3 (row stride ldq bytes) + scale [rows] fp32 with
// x ~= q * scale[row].  cols % 8 == 0, cols <= 8192.
int op_quant_fp8_rows(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int64_t rows, int64_t cols, void* stream) {
  OP_CHECK_ARG(x && q && scale && rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 8192 && ldx % 8 == 0 && ldq % 8 == 0,
               "quant_fp8_rows: bad arguments (cols %% 8 == 0, cols <= 8192)");
  if (rows == 0) return OP_OK;
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
  if (cols <= 2048)
    hipLaunchKernelGGL(quant_fp8_rows_kernel<4>, grid, dim3(256), 0, s, (const bf16_t*)x, ldx, (uint8_t*)q, ldq, scale, rows, (int)cols);
  else
    hipLaunchKernelGGL(quant_fp8_rows_kernel<16>, grid, dim3(256), 0, s, (const bf16_t*)x, ldx, (uint8_t*)q, ldq, scale, rows, (int)cols);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// C[M,N] (bf16) = epilogue( (A8 . B8^T)[m][n] * sa[m] * sb[n] )  with A8 [M,K] / B8 [N,K] fp8 e4m3 (row strides in bytes) and
// per-row fp32 scales.  epilogue: 0 bias; 2 GeGLU (B0 = wi_0, B1 = wi_1 with scales sb0 / sb1; C = gelu(h0) * h1, optional
// h0 / h1 outputs); 3 residual (C = resid + rowscale[m / rows_per_sample] * gamma[n] * (acc + bias[n]), optional h0 = acc + bias).
// K % 128 == 0, N % 8 == 0, lda / ldb % 16 == 0.
int op_gemm_nt_fp8(const void* A8, int64_t lda, const float* sa, const void* B0, const void* B1, int64_t ldb, const float* sb0,
                   const float* sb1, const void* bias, void* C, int64_t ldc, void* h0, void* h1, const void* resid, int64_t ldr,
                   const void* gamma, const float* rowscale, int64_t rows_per_sample, int64_t M, int64_t N, int64_t K, int epilogue,
                   void* stream) {
  OP_CHECK_ARG(A8 && sa && B0 && sb0 && C, "gemm_nt_fp8: null pointer");
  OP_CHECK_ARG(M >= 0 && N > 0 && K > 0 && K % 128 == 0 && N % 8 == 0 && lda % 16 == 0 && ldb % 16 == 0 && ldc % 8 == 0,
               "gemm_nt_fp8: K %% 128, N %% 8, lda/ldb %% 16, ldc %% 8 required (M=%lld N=%lld K=%lld)", (long long)M, (long long)N,
               (long long)K);
  OP_CHECK_ARG(epilogue == F8_EPI_BIAS || epilogue == F8_EPI_GEGLU || epilogue == F8_EPI_RESID, "gemm_nt_fp8: bad epilogue %d", epilogue);
  if (M == 0) return OP_OK;
  Fp8Args a;
  a.A = (const uint8_t*)A8; a.lda = lda; a.sa = sa;
  a.B[0] = (const uint8_t*)B0; a.B[1] = (const uint8_t*)B1; a.ldb = ldb; a.sb[0] = sb0; a.sb[1] = sb1;
  a.bias = (const bf16_t*)bias; a.C = (bf16_t*)C; a.ldc = ldc; a.H0 = (bf16_t*)h0; a.H1 = (bf16_t*)h1;
  a.resid = (const bf16_t*)resid; a.ldr = ldr; a.gamma = (const bf16_t*)gamma; a.rowscale = rowscale;
  a.rows_per_sample = rows_per_sample > 0 ? (int)rows_per_sample : 1;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = ceil_div(M, 128);
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * (double)M * (double)N * (double)K * (epilogue == F8_EPI_GEGLU ? 2.0 : 1.0);
  if (epilogue == F8_EPI_GEGLU) {  // every argument check comes before the profiler slot is taken: an early return must not leak it
    OP_CHECK_ARG(B1 && sb1, "gemm_nt_fp8: GeGLU needs two weights and two scale vectors");
    OP_CHECK_ARG((h0 == nullptr) == (h1 == nullptr), "gemm_nt_fp8: GeGLU h0/h1 must both be given or both null");
  }
  if (epilogue == F8_EPI_RESID) OP_CHECK_ARG(resid, "gemm_nt_fp8: residual epilogue needs resid");
  const int slot = op_prof_begin(3, flops, stream);
  int rc;
  if (epilogue == F8_EPI_GEGLU) {
    a.tiles_n = ceil_div(N, 64);
    rc = launch_fp8<F8_EPI_GEGLU>(a, s);
  } else {
    a.tiles_n = ceil_div(N, 128);
    if (epilogue == F8_EPI_RESID) {
      rc = launch_fp8<F8_EPI_RESID>(a, s);
    } else {
      rc = launch_fp8<F8_EPI_BIAS>(a, s);
    }
  }
  op_prof_end(slot, stream);
  return rc;
}

}  // extern "C"
