// First layer of the audio feature extractor, fused: Conv1d(1 -> C, kernel 10, stride 5, optional bias) -> LayerNorm(C) -> GELU,
// forward and backward, straight from the waveform.
//
// Replaces: one_peace/models/adapter/audio.py:254-311 (ConvFeatureExtractionModel, block 0: conv -> dropout(0) -> LayerNorm over channels ->
// GELU) -- rounds 1-4 ran it as an im2col GEMM (K = 10 padded to 64) + op_layernorm_fwd/bwd + a weight-gradient GEMM.
//
// Why a kernel of its own: at the headline batch the layer has 128 x 16 000 = 2.05 M output rows of 512 channels = 2.1 GB per tensor
// and a K of TEN.  The GEMM form writes that tensor, the LayerNorm pass reads and rewrites it, the backward reads it again, writes the
// gradient of the convolution output (2.1 GB) and the weight-gradient GEMM reads that once more: 12.6 GB of traffic around 42 GFLOP.
// Here a row is COMPUTED from its ten waveform samples (20 bytes) wherever it is needed: forward = one 2.1 GB write; backward = one
// 2.1 GB read (the incoming gradient) with the convolution's weight / bias gradients and the LayerNorm's accumulated in registers.
// Roofline: HBM (1 output byte stream each way); the arithmetic is 80 FMAs + LayerNorm + erf-GELU per 8 channels of a row (VALU).
//
// Mapping: one wavefront per row (C <= 512: lane l owns channels 8 l ... 8 l + 7), four rows per 256-thread workgroup, persistent
// stride over the rows.  Numerics follow the unfused path: the convolution output is rounded to bf16 before the statistics (what the
// GEMM stored), statistics in fp32 with a centred second pass, exact-erf GELU (common.h: gelu_parts).
#include "common.h"

namespace {

constexpr int AK = 10;            // kernel width of the first convolution
constexpr int A_MAX_BLOCKS = 1024;

template <bool HAS_BIAS>
__device__ __forceinline__ void conv_row(const bf16_t* __restrict__ wav, int64_t row, int stride, const float (&w)[8][AK], const float (&cb)[8],
                                         float (&v)[8]) {
  float s[AK];
  const bf16_t* p = wav + row * stride;
#pragma unroll
  for (int j = 0; j < AK; ++j) s[j] = (float)p[j];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < AK; ++j) a = __builtin_fmaf(w[i][j], s[j], a);
    if (HAS_BIAS) a += cb[i];
    v[i] = (float)(bf16_t)a;  // the unfused path stored the convolution output as bf16 before the LayerNorm read it
  }
}

template <bool HAS_BIAS>
__global__ __launch_bounds__(256) void conv1_ln_gelu_fwd_kernel(const bf16_t* __restrict__ wav, int stride, const bf16_t* __restrict__ w0,
                                                                const bf16_t* __restrict__ b0, const bf16_t* __restrict__ lnw,
                                                                const bf16_t* __restrict__ lnb, bf16_t* __restrict__ y,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out, int64_t rows,
                                                                int C, float eps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c0 = lane * 8;
  const bool act = c0 < C;
  const float inv = 1.0f / (float)C;
  float w[8][AK], cb[8], g[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    cb[i] = 0.f; g[i] = 1.f; b[i] = 0.f;
#pragma unroll
    for (int j = 0; j < AK; ++j) w[i][j] = act ? (float)w0[(int64_t)(c0 + i) * AK + j] : 0.f;
    if (act) {
      if (HAS_BIAS) cb[i] = (float)b0[c0 + i];
      if (lnw) g[i] = (float)lnw[c0 + i];
      if (lnb) b[i] = (float)lnb[c0 + i];
    }
  }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wid; row < rows; row += (int64_t)gridDim.x * 4) {
    float v[8];
    conv_row<HAS_BIAS>(wav, row, stride, w, cb, v);
    float s = 0.f;
    if (act) {
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i];
    }
    const float mean = wave_sum(s) * inv;
    float ss = 0.f;
    if (act) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; ss += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(ss) * inv + eps);
    if (act) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = gelu_erf((v[i] - mean) * rstd * g[i] + b[i]);
      Vec8<bf16_t>::store_nt(y + row * (int64_t)C + c0, o);
    }
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
  }
}

// Backward: dy [rows, C] -> partial sums per workgroup in ws[gridDim.x][C * (AK + 3)] (fp32):
//   [0, C*AK)        dW0[c][j] = sum_r dx[r][c] * wav[stride r + j]
//   [C*AK, +C)       db0[c]    = sum_r dx[r][c]
//   [.., +C)         dln_w[c]  = sum_r dy gelu'(.) xhat
//   [.., +C)         dln_b[c]  = sum_r dy gelu'(.)
// with dx = rstd * (g w - mean(g w) - xhat * mean(g w xhat)), g = dy * gelu'(xhat w + b) -- op_layernorm_bwd's arithmetic on a row that
// is recomputed from the waveform (the waveform itself needs no gradient).
template <bool HAS_BIAS>
__global__ __launch_bounds__(256) void conv1_ln_gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ wav, int stride,
                                                                const bf16_t* __restrict__ w0, const bf16_t* __restrict__ b0,
                                                                const bf16_t* __restrict__ lnw, const bf16_t* __restrict__ lnb,
                                                                const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                float* __restrict__ ws, int64_t rows, int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4][C] fold buffer
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c0 = lane * 8;
  const bool act = c0 < C;
  const float inv = 1.0f / (float)C;
  // parameters stay PACKED (bf16 pairs: 40 + 12 registers instead of 80 + 24; unpacked where they are used: one VALU op each) -- with
  // the 80 + 24 gradient accumulators the kernel has to fit 256 registers for two waves per SIMD
  bf16x2 wp[8][AK / 2];
  bf16x8 cbp, gp, bp;
  float dw[8][AK], db[8], dlw[8], dlb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    cbp[i] = (bf16_t)0.f; gp[i] = (bf16_t)1.f; bp[i] = (bf16_t)0.f; db[i] = 0.f; dlw[i] = 0.f; dlb[i] = 0.f;
#pragma unroll
    for (int j = 0; j < AK; ++j) { wp[i][j >> 1][j & 1] = act ? w0[(int64_t)(c0 + i) * AK + j] : (bf16_t)0.f; dw[i][j] = 0.f; }
    if (act) {
      if (HAS_BIAS) cbp[i] = b0[c0 + i];
      if (lnw) gp[i] = lnw[c0 + i];
      if (lnb) bp[i] = lnb[c0 + i];
    }
  }
  typename Vec8<bf16_t>::raw_t cur, nxt;
  int64_t row = (int64_t)blockIdx.x * 4 + wid;
  const int64_t rstep = (int64_t)gridDim.x * 4;
  if (row < rows && act) cur = Vec8<bf16_t>::ldraw_nt(dy + row * (int64_t)C + c0);
  for (; row < rows; row += rstep) {
    const int64_t nrow = row + rstep;
    if (nrow < rows && act) nxt = Vec8<bf16_t>::ldraw_nt(dy + nrow * (int64_t)C + c0);
    const float mean = mean_in[row], rstd = rstd_in[row];
    float s[AK];
    {
      const bf16_t* p = wav + row * stride;
#pragma unroll
      for (int j = 0; j < AK; ++j) s[j] = (float)p[j];
    }
    float xh[8], gw[8];
    float s1 = 0.f, s2 = 0.f;
    if (act) {
      float d[8];
      Vec8<bf16_t>::cvt(cur, d);
      asm volatile("" : "+v"(gp), "+v"(bp), "+v"(cbp));  // (keeps the unpacking inside the row loop: hoisted, the parameters are 104 registers again)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("" : "+v"(wp[i][0]), "+v"(wp[i][1]), "+v"(wp[i][2]), "+v"(wp[i][3]), "+v"(wp[i][4]));
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < AK; ++j) a = __builtin_fmaf((float)wp[i][j >> 1][j & 1], s[j], a);
        if (HAS_BIAS) a += (float)cbp[i];
        const float gi_w = (float)gp[i];
        xh[i] = ((float)(bf16_t)a - mean) * rstd;
        const float gi = d[i] * gelu_erf_grad(xh[i] * gi_w + (float)bp[i]);
        dlw[i] += gi * xh[i];
        dlb[i] += gi;
        gw[i] = gi * gi_w;
        s1 += gw[i];
        s2 += gw[i] * xh[i];
      }
    }
    const float m1 = wave_sum(s1) * inv, m2 = wave_sum(s2) * inv;
    if (act) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // (the unfused path rounded dx to bf16 between the LayerNorm backward and the weight-gradient GEMM)
        const float dx = (float)(bf16_t)(rstd * (gw[i] - m1 - xh[i] * m2));
        db[i] += dx;
#pragma unroll
        for (int j = 0; j < AK; ++j) dw[i][j] = __builtin_fmaf(dx, s[j], dw[i][j]);
      }
    }
    cur = nxt;
  }
  // fold the four waves (same channels, different rows) through LDS, one quantity at a time; workgroup partials go to ws
  float* wsb = ws + (int64_t)blockIdx.x * C * (AK + 3);
  for (int q = 0; q < AK + 3; ++q) {
    __syncthreads();
    if (act) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float val;
        if (q < AK) {
          val = 0.f;
#pragma unroll
          for (int j = 0; j < AK; ++j) val = (j == q) ? dw[i][j] : val;  // (compile-time indices: the arrays stay in registers)
        } else {
          val = q == AK ? db[i] : (q == AK + 1 ? dlw[i] : dlb[i]);
        }
        smem[wid * C + c0 + i] = val;
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const float t = smem[c] + smem[C + c] + smem[2 * C + c] + smem[3 * C + c];
      if (q < AK) wsb[(int64_t)c * AK + q] = t;   // dW0 in the weight's own [C][AK] layout
      else wsb[(int64_t)C * AK + (q - AK) * C + c] = t;
    }
  }
}

inline int a_grid(int64_t rows) {
  int64_t blocks = (rows + 3) / 4;
  if (blocks > A_MAX_BLOCKS) blocks = A_MAX_BLOCKS;
  return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" {

// y [rows, C] = GELU(LayerNorm_C(bf16(conv(wav)))) with conv(wav)[r][c] = sum_j w0[c][j] * wav[stride * r + j] (+ b0[c]), j < 10
// (one_peace/models/adapter/audio.py:254-311, block 0 of ConvFeatureExtractionModel); mean / rstd [rows] fp32 are kept for the backward.
// wav: bf16, at least stride * (rows - 1) + 10 elements; C <= 512, C % 8 == 0; b0, lnw, lnb nullable.
int op_audio_conv1_ln_gelu_fwd(const void* wav, int64_t stride, const void* w0, const void* b0, const void* lnw, const void* lnb, void* y,
                               float* mean, float* rstd, int64_t rows, int64_t C, float eps, void* stream) {
  OP_CHECK_ARG(wav && w0 && y && mean && rstd, "audio_conv1_ln_gelu_fwd: null pointer");
  OP_CHECK_ARG(rows >= 0 && C > 0 && C <= 512 && C % 8 == 0 && stride > 0, "audio_conv1_ln_gelu_fwd: C=%lld (<= 512, multiple of 8), stride=%lld",
               (long long)C, (long long)stride);
  if (rows == 0) return OP_OK;
  hipStream_t s = (hipStream_t)stream;
  if (b0)
    hipLaunchKernelGGL((conv1_ln_gelu_fwd_kernel<true>), dim3(a_grid(rows)), dim3(256), 0, s, (const bf16_t*)wav, (int)stride, (const bf16_t*)w0,
                       (const bf16_t*)b0, (const bf16_t*)lnw, (const bf16_t*)lnb, (bf16_t*)y, mean, rstd, rows, (int)C, eps);
  else
    hipLaunchKernelGGL((conv1_ln_gelu_fwd_kernel<false>), dim3(a_grid(rows)), dim3(256), 0, s, (const bf16_t*)wav, (int)stride, (const bf16_t*)w0,
                       (const bf16_t*)b0, (const bf16_t*)lnw, (const bf16_t*)lnb, (bf16_t*)y, mean, rstd, rows, (int)C, eps);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

int64_t op_audio_conv1_ln_gelu_bwd_workspace_bytes(int64_t C) { return (int64_t)A_MAX_BLOCKS * C * (AK + 3) * (int64_t)sizeof(float); }

// Gradients of op_audio_conv1_ln_gelu_fwd's parameters from dy [rows, C] (the waveform gets none): dw0 [C, 10], db0 [C] (nullable),
// dlnw [C], dlnb [C] (nullable), all bf16, overwritten or accumulated.  workspace: op_audio_conv1_ln_gelu_bwd_workspace_bytes(C).
int op_audio_conv1_ln_gelu_bwd(const void* dy, const void* wav, int64_t stride, const void* w0, const void* b0, const void* lnw, const void* lnb,
                               const float* mean, const float* rstd, void* dw0, void* db0, void* dlnw, void* dlnb, void* workspace,
                               int64_t rows, int64_t C, int accumulate, void* stream) {
  OP_CHECK_ARG(dy && wav && w0 && mean && rstd && dw0 && workspace, "audio_conv1_ln_gelu_bwd: null pointer");
  OP_CHECK_ARG(rows > 0 && C > 0 && C <= 512 && C % 8 == 0 && stride > 0, "audio_conv1_ln_gelu_bwd: C=%lld (<= 512, multiple of 8), stride=%lld",
               (long long)C, (long long)stride);
  hipStream_t s = (hipStream_t)stream;
  const int grid = a_grid(rows);
  const size_t sh = (size_t)4 * C * sizeof(float);
  float* ws = (float*)workspace;
  if (b0)
    hipLaunchKernelGGL((conv1_ln_gelu_bwd_kernel<true>), dim3(grid), dim3(256), sh, s, (const bf16_t*)dy, (const bf16_t*)wav, (int)stride,
                       (const bf16_t*)w0, (const bf16_t*)b0, (const bf16_t*)lnw, (const bf16_t*)lnb, mean, rstd, ws, rows, (int)C);
  else
    hipLaunchKernelGGL((conv1_ln_gelu_bwd_kernel<false>), dim3(grid), dim3(256), sh, s, (const bf16_t*)dy, (const bf16_t*)wav, (int)stride,
                       (const bf16_t*)w0, (const bf16_t*)b0, (const bf16_t*)lnw, (const bf16_t*)lnb, mean, rstd, ws, rows, (int)C);
  OP_LAUNCH_CHECK();
  // folds of the workgroup partials: dW0 as one vector of C * AK entries, then the three [C] vectors in one launch
  const int64_t pstride = (int64_t)C * (AK + 3);
  hipLaunchKernelGGL((partials_reduce_kernel<bf16_t>), dim3(ceil_div(C * AK, 32)), dim3(256), 0, s, (const float*)ws, grid, pstride, (int)(C * AK),
                     (const bf16_t*)nullptr, (bf16_t*)dw0, accumulate);
  OP_LAUNCH_CHECK();
  hipLaunchKernelGGL((partials_reduce3_kernel<bf16_t>), dim3(ceil_div(C, 32), 3), dim3(256), 0, s, ws + C * AK, ws + C * AK + C, ws + C * AK + 2 * C,
                     (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (bf16_t*)db0, (bf16_t*)dlnw, (bf16_t*)dlnb, grid,
                     pstride, (int)C, accumulate);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

}  // extern "C"
