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
#include <string.h>
#include "gemm_epilogue_v.h"  // GemmArgs, epilogue_v<EPI, SCALED> (shared with gemm.hip)

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
  const float sc = fp8_row_scale(amax);
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  uint8_t* qr = q + row * ldq;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < cols) *reinterpret_cast<u32x2*>(qr + c) = fp8_pack8(v[i], inv);
  }
}

// Rows wider than 8192 columns (round 6: the [N, 2F = 12 288] gradient of the GeGLU up-projection, an operand of the fp8 input-gradient
// GEMM): two passes over the row, the second one re-reads it (L2-resident: 24 KiB per wavefront) instead of keeping 192 values per lane.
__global__ __launch_bounds__(256) void quant_fp8_rows_wide_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q,
                                                                  int64_t ldq, float* __restrict__ scale, int64_t rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * ldx;
  float amax = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    float v[8];
    Vec8<bf16_t>::load(xr + c, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
  }
  amax = wave_max(amax);
  const float sc = fp8_row_scale(amax);
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  uint8_t* qr = q + row * ldq;
  for (int c = lane * 8; c < cols; c += 512) {
    float v[8];
    Vec8<bf16_t>::load(xr + c, v);
    *reinterpret_cast<u32x2*>(qr + c) = fp8_pack8(v, inv);
  }
}

// =====================================================================================================================
// gemm256f8_kernel (round 5): the fp8 GEMM on the skeleton of the bf16 production kernel gemm256v_kernel (csrc/gemm.hip) --
// 256 x 256 tile, four waves (one per SIMD) of 128 x 128, accumulators pinned in the 256 AGPRs by inline-asm MFMAs, five 32 KiB LDS
// slots cycled A0 B0 A1 B1 A2 ..., LDS-DMA through buffer descriptors whose base carries the K position (K-tiles past the end are
// fetched through EMPTY descriptors: one uniform loop), at most one memory instruction between two MFMAs, activation fragment as
// FIRST operand so that the epilogue (epilogue_v<EPI, SCALED>) writes 16-byte coalesced rows.  An fp8 K-tile is 128 elements = the
// same 128-byte LDS rows, swizzle and DMA pattern as a 64-deep bf16 K-tile: twice the flops per staged byte.
// What differs is the register blocking: an operand of v_mfma_scale_f32_16x16x128_f8f6f4 is 32 bytes per lane (8 VGPRs) -- the two
// 16-byte chunks (g) and (4 + g) of the lane's row, i.e. the two "halves" the bf16 kernel feeds to two K = 32 MFMAs; the K
// permutation is the same for both operands, so the sums are those of the natural order -- and double-buffering all 16 fragments of
// a wave would take 256 VGPRs.  So per K-tile: the 8 activation fragments are double-buffered (2 x 64 VGPRs), the weight fragments
// travel in two quarter sets of 4 (2 x 32 VGPRs): phase A runs the 32 MFMAs of weight fragments 0-3 while fragments 4-7 of the SAME
// tile are read, phase B those of 4-7 while the next tile's activation fragments and weight fragments 0-3 are read (24 reads + 8
// DMA ops: one per MFMA gap -- an fp8 MFMA holds the pipe for 32 cycles, twice a bf16 one).  K % 256 == 0 (two K-tiles per trip).
// Block scales: all E8M0 = 2^0, the per-row fp32 scales are applied by the epilogue (see the file header).
// =====================================================================================================================
constexpr int F8_SLOT_BYTES = 256 * 128;  // one operand K-tile: 256 rows x 128 fp8
constexpr int F8_SLOTS = 5;
#define F8_WAIT_LGKM(n) __builtin_amdgcn_s_waitcnt(0xC07F | ((n) << 8))
#define F8_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 0xF) | ((((n) >> 4) & 3) << 14))

struct F8Args256 { GemmArgs g; const float* sa; const float* sb; };  // g.A / g.B[0]: fp8 bytes; g.lda / g.ldb / g.K in BYTES = elements

template <int EPI>
__global__ __launch_bounds__(256) void gemm256f8_kernel(const F8Args256 q) {
  static_assert(EPI == EPI_BIAS || EPI == EPI_RESID, "gemm256f8_kernel: plain / bias and residual epilogues");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GemmArgs& p = q.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int g = lane >> 4, t = lane & 15;

  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int GM = p.gm;
  const int per_group = GM * p.tiles_n;
  const int first_m = (pid / per_group) * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int in_group = pid % per_group;
  const int pid_m = first_m + in_group % gsz;
  const int pid_n = in_group / gsz;
  const int m0 = pid_m * 256, n0 = pid_n * 256;
  const int nk = p.K / 128;

  // ---- staging (gemm256v_kernel): op j of an operand tile covers LDS rows j*32 + (tid >> 3), 16-byte slot tid & 7 ----
  const int srow = tid >> 3;
  const int sc = (tid & 7) ^ (srow & 7);
  unsigned offA[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int gm = min(m0 + j * 32 + srow, p.M - 1);
    offA[j] = (unsigned)((int64_t)gm * p.lda + sc * 16);
  }
  const int nrecA = (int)(((int64_t)p.M - 1) * p.lda + p.K);
  const int lanecol = (srow & 15) * 8 + (srow >> 4);  // the column map of epilogue_v (VMAP)
  const unsigned offB = (unsigned)((int64_t)lanecol * p.ldb + sc * 16);
  unsigned soffB[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) soffB[j] = (unsigned)((int64_t)(n0 + (j >> 2) * 128 + ((j >> 1) & 1) * 4 + (j & 1) * 2) * p.ldb);
  const char* ptrA = (const char*)p.A;
  const char* ptrB = (const char*)p.B[0];
  const int nrecB = (int)(((int64_t)p.N - 1) * p.ldb + p.K);
  int ktA = 0, ktB = 0;

  f32x4 acc[2][4][8];  // [64-column block][ni][mi]

  const int fsw[2] = {((0 * 4 + g) ^ (t & 7)) << 4, ((1 * 4 + g) ^ (t & 7)) << 4};
  const int rowX = (wm * 128 + t) * 128;  // + mi * 2048
  auto w_off = [&](int f) { return (wn * 128 + (f >> 2) * 64 + (f & 3) * 16 + t) * 128; };

  int qslot_issue = 0;
  auto rsrc_a = [&]() {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(ptrA + (int64_t)ktA * 128), 0, ktA < nk ? nrecA - ktA * 128 : 0, 0x00020000);
  };
  auto rsrc_b = [&]() {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(ptrB + (int64_t)ktB * 128), 0, ktB < nk ? nrecB - ktB * 128 : 0, 0x00020000);
  };
  auto dma_a = [&](const __amdgpu_buffer_rsrc_t& r, char* dst, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, offA[j], 0, 0, 0);
  };
  auto dma_b = [&](const __amdgpu_buffer_rsrc_t& r, char* dst, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, offB, soffB[j], 0, 0);
  };
  auto advance = [&](bool is_b) {
    if (is_b) ++ktB; else ++ktA;
    qslot_issue = qslot_issue == F8_SLOTS - 1 ? 0 : qslot_issue + 1;
  };
  auto issue_tile = [&](bool is_b) {
    char* dst = smem + qslot_issue * F8_SLOT_BYTES + wid * 1024;
    const __amdgpu_buffer_rsrc_t ra = rsrc_a(), rb = rsrc_b();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (is_b) dma_b(rb, dst, j); else dma_a(ra, dst, j);
    }
    advance(is_b);
  };

  struct Frag { u32x4 h[2]; };  // the two 16-byte chunks (g), (4 + g) of the lane's row = one 32-byte MFMA operand
  auto operand = [](const Frag& f) {
    return (i32x8){(int)f.h[0][0], (int)f.h[0][1], (int)f.h[0][2], (int)f.h[0][3], (int)f.h[1][0], (int)f.h[1][1], (int)f.h[1][2], (int)f.h[1][3]};
  };
  const int one = E8M0_ONE;
  auto mfma = [&](int k, int f, const Frag (&x)[8], const Frag (&w)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc checks asm constraints against x86: a 32-byte "v" operand is an error there,
    // reported nowhere -- the kernel's host stub just goes missing from the object)
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                 : "+a"(acc[f >> 2][f & 3][k])
                 : "v"(operand(x[k])), "v"(operand(w[f & 3])), "v"(one));
#endif
  };
  auto rd_x = [&](Frag (&x)[8], const char* sa_, int r) {  // r = 0..15: fragment r >> 1, half r & 1
    x[r >> 1].h[r & 1] = *reinterpret_cast<const u32x4*>(sa_ + rowX + (r >> 1) * 2048 + fsw[r & 1]);
  };
  auto rd_w = [&](Frag (&w)[4], const char* sb_, int first, int r) {  // r = 0..7: weight fragment first + (r >> 1), half r & 1
    w[r >> 1].h[r & 1] = *reinterpret_cast<const u32x4*>(sb_ + w_off(first + (r >> 1)) + fsw[r & 1]);
  };

  // One K-tile.  Phase A: 32 MFMAs of weight fragments 0-3 (w_lo) on x_cur; reads fragments 4-7 of the same tile into w_hi; DMA of the
  // activation tile two ahead.  Barrier (the next tile has landed; every wave is done with this tile's slots).  Phase B: 32 MFMAs of
  // w_hi; reads the next tile's activation fragments into x_nxt and its weight fragments 0-3 into w_lo; DMA of the weight tile two ahead.
  auto tile_step = [&](const Frag (&x_cur)[8], Frag (&x_nxt)[8], Frag (&w_lo)[4], Frag (&w_hi)[4], const char* sa_, const char* sb_,
                       const char* sa1_, const char* sb1_) {
    {
      char* dst = smem + qslot_issue * F8_SLOT_BYTES + wid * 1024;
      const __amdgpu_buffer_rsrc_t r0 = rsrc_a();
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        mfma(i >> 2, i & 3, x_cur, w_lo);
        if (i & 1) {  // 16 gaps: 8 reads first, then the 8 DMA ops
          __builtin_amdgcn_sched_barrier(0);
          const int o = i >> 1;
          if (o < 8) rd_w(w_hi, sb_, 4, o); else dma_a(r0, dst, o - 8);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      advance(false);
    }
    F8_WAIT_LGKM(0);
    F8_WAIT_VM(8);
    __builtin_amdgcn_s_barrier();
    {
      char* dst = smem + qslot_issue * F8_SLOT_BYTES + wid * 1024;
      const __amdgpu_buffer_rsrc_t r0 = rsrc_b();
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        mfma(i >> 2, 4 + (i & 3), x_cur, w_hi);
        __builtin_amdgcn_sched_barrier(0);
        if ((i & 3) == 3) dma_b(r0, dst, i >> 2);         // 8 DMA ops
        else {
          const int o = (i >> 2) * 3 + (i & 3);           // 24 reads: weight fragments 0-3 of the next tile first, then its activations
          if (o < 8) rd_w(w_lo, sb1_, 0, o); else rd_x(x_nxt, sa1_, o - 8);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      advance(true);
    }
  };

  // ---- prologue: A0 B0 A1 B1; accumulators cleared under the latency of the first loads; fragments of tile 0 ----
  issue_tile(false);
  issue_tile(true);
  issue_tile(false);
  issue_tile(true);
  {
    const bf16x8 zf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, 0" : "=a"(acc[a][b][c]) : "v"(zf));
  }
  F8_WAIT_VM(16);
  __builtin_amdgcn_s_barrier();
  Frag x0[8], x1[8], wlo[4], whi[4];
  int sa = 0, sb = 1;
#pragma unroll
  for (int r = 0; r < 8; ++r) rd_w(wlo, smem + sb * F8_SLOT_BYTES, 0, r);
#pragma unroll
  for (int r = 0; r < 16; ++r) rd_x(x0, smem + sa * F8_SLOT_BYTES, r);
  int nk_last = nk;
  asm volatile("" : "+s"(nk_last));
  auto nxt_slot = [](int s) { return s + 2 >= F8_SLOTS ? s + 2 - F8_SLOTS : s + 2; };
  for (int i = 0; i < nk; i += 2) {  // two K-tiles per trip: the activation fragment sets swap roles every tile
    const int sa1 = nxt_slot(sa), sb1 = nxt_slot(sb), sa2 = nxt_slot(sa1), sb2 = nxt_slot(sb1);
    F8_WAIT_LGKM(0);
    asm volatile("s_nop 0");
    __builtin_amdgcn_sched_barrier(0);
    tile_step(x0, x1, wlo, whi, smem + sa * F8_SLOT_BYTES, smem + sb * F8_SLOT_BYTES, smem + sa1 * F8_SLOT_BYTES, smem + sb1 * F8_SLOT_BYTES);
    F8_WAIT_LGKM(0);
    asm volatile("s_nop 0");
    __builtin_amdgcn_sched_barrier(0);
    tile_step(x1, x0, wlo, whi, smem + sa1 * F8_SLOT_BYTES, smem + sb1 * F8_SLOT_BYTES, smem + sa2 * F8_SLOT_BYTES, smem + sb2 * F8_SLOT_BYTES);
    sa = sa2;
    sb = sb2;
    // the last MFMAs retire INSIDE the loop body on the last trip (see gemm256v_kernel: accumulator copies on the loop-exit edge)
    if (i + 2 >= nk_last) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");
  }
  // the empty fetches and the unused fragment reads behind the last tile must be gone before the epilogue reuses registers
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  epilogue_v<EPI, true>(p, acc, m0 + wm * 128, n0 + wn * 128, g, t, nullptr, false, q.sa, q.sb);
}

template <int EPI>
int launch_fp8_256(const F8Args256& a, hipStream_t s) {
  const size_t sh = (size_t)F8_SLOTS * F8_SLOT_BYTES;
  OP_ENSURE_LDS((gemm256f8_kernel<EPI>), (int)sh, "gemm256f8");
  hipLaunchKernelGGL((gemm256f8_kernel<EPI>), dim3(a.g.tiles_m * a.g.tiles_n), dim3(256), sh, s, a);
  OP_LAUNCH_CHECK();
  return OP_OK;
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
  OP_CHECK_ARG(x && q && scale && rows >= 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldq % 8 == 0,
               "quant_fp8_rows: bad arguments (cols, ldx, ldq %% 8 == 0)");
  if (rows == 0) return OP_OK;
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
  if (cols <= 2048)
    hipLaunchKernelGGL(quant_fp8_rows_kernel<4>, grid, dim3(256), 0, s, (const bf16_t*)x, ldx, (uint8_t*)q, ldq, scale, rows, (int)cols);
  else if (cols <= 8192)
    hipLaunchKernelGGL(quant_fp8_rows_kernel<16>, grid, dim3(256), 0, s, (const bf16_t*)x, ldx, (uint8_t*)q, ldq, scale, rows, (int)cols);
  else
    hipLaunchKernelGGL(quant_fp8_rows_wide_kernel, grid, dim3(256), 0, s, (const bf16_t*)x, ldx, (uint8_t*)q, ldq, scale, rows, (int)cols);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// C[M,N] (bf16) = epilogue( (A8 . B8^T)[m][n] * sa[m] * sb[n] )  with A8 [M,K] / B8 [N,K] fp8 e4m3 (row strides in bytes) and
// per-row fp32 scales.  epilogue: 0 bias; 2 GeGLU (B0 = wi_0, B1 = wi_1 with scales sb0 / sb1; C = gelu(h0) * h1, optional
// h0 / h1 outputs); 3 residual (C = resid + rowscale[m / rows_per_sample] * gamma[n] * (acc + bias[n]), optional h0 = acc + bias).
// K % 128 == 0, N % 8 == 0, lda / ldb % 16 == 0.  Launches that fill the chip with 256 x 256 tiles (N % 256 == 0, K % 256 == 0, plain /
// bias or residual epilogue) run on gemm256f8_kernel, everything else on the 128 x 128 kernel; tune bit 0 forces the latter.
int op_gemm_nt_fp8(const void* A8, int64_t lda, const float* sa, const void* B0, const void* B1, int64_t ldb, const float* sb0,
                   const float* sb1, const void* bias, void* C, int64_t ldc, void* h0, void* h1, const void* resid, int64_t ldr,
                   const void* gamma, const float* rowscale, int64_t rows_per_sample, int64_t M, int64_t N, int64_t K, int epilogue,
                   int64_t tune, void* stream) {
  const bool tune_small = (tune & 1) != 0;  // tune bit 0: keep the 128 x 128 kernel (tests, A/B)
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
  // round 5: launches that fill the chip with 256 x 256 tiles take the four-wave kernel (plain / bias and residual epilogues)
  const bool big = epilogue != F8_EPI_GEGLU && N % 256 == 0 && K % 256 == 0 && ((M + 255) / 256) * (N / 256) >= 256 &&
                   (M - 1) * lda + K < ((int64_t)1 << 31) && (N - 1) * ldb + K < ((int64_t)1 << 31) && M * ldc < ((int64_t)1 << 30) &&
                   (!resid || M * ldr < ((int64_t)1 << 30)) && !(tune_small);
  if (big) {
    F8Args256 b;
    memset(&b, 0, sizeof(b));
    b.g.A = (const bf16_t*)A8; b.g.lda = lda; b.g.B[0] = (const bf16_t*)B0; b.g.ldb = ldb; b.g.n_seg = (int)N;
    b.g.bias[0] = (const bf16_t*)bias; b.g.C = C; b.g.ldc = ldc; b.g.H0 = (bf16_t*)h0; b.g.resid = (const bf16_t*)resid; b.g.ldr = ldr;
    b.g.gamma = (const bf16_t*)gamma; b.g.rowscale = rowscale; b.g.rows_per_sample = a.rows_per_sample;
    b.g.M = (int)M; b.g.N = (int)N; b.g.K = (int)K; b.g.tiles_m = ceil_div(M, 256); b.g.tiles_n = (int)(N / 256);
    b.g.gm = b.g.tiles_n <= 8 ? 1 : 8;
    b.sa = sa; b.sb = sb0;
    rc = epilogue == F8_EPI_RESID ? launch_fp8_256<EPI_RESID>(b, s) : launch_fp8_256<EPI_BIAS>(b, s);
  } else if (epilogue == F8_EPI_GEGLU) {
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
