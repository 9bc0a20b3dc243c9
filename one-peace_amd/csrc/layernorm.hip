// LayerNorm forward/backward for gfx950, optional fused exact-erf GELU on the output.
//
// Replaces: one_peace/models/components.py:23-26,47-52 (torch.nn.LayerNorm / flash_attn layer_norm seam)
// and, with act=1, the LayerNorm->GELU pairs of adapter/image.py:66-75 and adapter/audio.py:293-301.
//
// Roofline: pure HBM.  Algorithmic bytes (bf16): fwd 4*cols B/row (+8 B stats), bwd 6*cols B/row
// read (x, dy) + write (dx) (+ the two [cols] reductions, negligible).
// Mapping: one wavefront (64 lanes) per row when cols <= 2048, a whole 256-thread workgroup per row
// above that; every lane moves 16-byte (bf16x8) or 32-byte (f32x8) chunks, statistics in fp32 with a
// centred second pass (same numerics as torch: biased variance, eps inside the rsqrt).
#include "common.h"

namespace {

constexpr int LN_MAX_BLOCKS = 4096;       // workspace bound for the dw/db partials
constexpr int g_ln_blocks_fwd = 512, g_ln_blocks_bwd = 512;  // persistent-grid caps (round-1 sweep: tools/ln_sweep.py, profiles/)
// The fused GeGLU + LayerNorm(F) passes have their own caps (tools/ln_geglu_sweep.py, 32 896 x 6144, round 3): forward 256 / 512 /
// 1024 / 2048 / 4096 workgroups -> 0.389 / 0.265 / 0.240 / 0.237 / 0.236 ms; backward (dw / db partials per workgroup) 0.530 /
// 0.403 / 0.417 / 0.448 / 0.507 ms.
#ifndef OP_LN_GEGLU_BLOCKS_FWD
#define OP_LN_GEGLU_BLOCKS_FWD 2048
#endif
#ifndef OP_LN_GEGLU_BLOCKS_BWD
#define OP_LN_GEGLU_BLOCKS_BWD 512
#endif

template <int NW>
__device__ __forceinline__ float group_sum(float v, float* red) {
  v = wave_sum(v);
  if (NW == 1) return v;
  const int wid = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from the previous use
  if ((threadIdx.x & 63) == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}

// Two sums with ONE barrier pair (the backward kernels need mean(dy w) and mean(dy w xhat) of the same row); red: >= 2 * NW floats.
template <int NW>
__device__ __forceinline__ void group_sum2(float& a, float& b, float* red) {
  a = wave_sum(a);
  b = wave_sum(b);
  if (NW == 1) return;
  const int wid = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from the previous use
  if ((threadIdx.x & 63) == 0) { red[wid] = a; red[NW + wid] = b; }
  __syncthreads();
  float ta = 0.f, tb = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) { ta += red[i]; tb += red[NW + i]; }
  a = ta;
  b = tb;
}

template <int NW>
__device__ __forceinline__ float group_max(float v, float* red) {
  v = wave_max(v);
  if (NW == 1) return v;
  const int wid = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wid] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
  return t;
}

// Q8 (round 5; bf16, no GELU): the row is ALSO written as fp8 e4m3 with a per-row scale -- exactly what op_quant_fp8_rows makes of the
// bf16 output (amax over the bf16-rounded values), without its pass over the output: the operand of the fp8 FFN GEMMs (csrc/fp8.hip).
template <typename T, int CH, int NW, bool GELU, bool NT, bool Q8 = false>  // NT: non-temporal row accesses (tensors far beyond the caches: training)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                     const T* __restrict__ b, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int64_t rows, int cols, float eps, uint8_t* __restrict__ q8 = nullptr,
                                                     float* __restrict__ q8_scale = nullptr, const int* __restrict__ xmap = nullptr) {
  // xmap (nullable, ABI 9): row r of the input is row xmap[r] of a LARGER matrix x (< 0: no row -- normalised as a row of zeros, which is
  // what op_rows_gather's packed copy holds there); y / mean / rstd are indexed by r.  The packed rows of the samples a residual
  // branch keeps, read straight from the full activation matrix.
  __shared__ float red[4];
  constexpr int G = 64 * NW;
  const int tig = (NW == 1) ? (threadIdx.x & 63) : threadIdx.x;
  const int64_t row0 = (NW == 1) ? ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) : blockIdx.x;
  const int64_t rstep = (NW == 1) ? (int64_t)gridDim.x * 4 : gridDim.x;
  const float inv = 1.0f / (float)cols;

  float wv[CH][8], bv[CH][8];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = (tig + G * i) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { wv[i][j] = 1.f; bv[i][j] = 0.f; }
    if (c < cols) {
      if (w) Vec8<T>::load(w + c, wv[i]);
      if (b) Vec8<T>::load(b + c, bv[i]);
    }
  }
  // NW == 4: every thread of the block walks the same rows (uniform trip count -> barriers are safe).
  // The next row is requested before the current one is reduced: twice the bytes in flight per wave.
  typedef typename Vec8<T>::raw_t raw_t;
  raw_t cur[CH], nxt[CH];
  // (the table entry of a row is requested one row BEFORE its data: behind each other in one trip, every row paid an L2 latency)
  auto fetch = [&](int64_t src, raw_t (&dst)[CH]) {  // (uniform per row group: the entry is the same for every thread of the row)
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
        if (src >= 0) dst[i] = ldraw_sel<T, NT>(x + src * (int64_t)cols + c);
        else dst[i] = raw_t{};
      }
    }
  };
  int64_t nsrc = row0 + rstep < rows ? (xmap ? (int64_t)xmap[row0 + rstep] : row0 + rstep) : -1;
  if (row0 < rows) fetch(xmap ? (int64_t)xmap[row0] : row0, cur);
  for (int64_t row = row0; row < rows; row += rstep) {
    const int64_t nrow = row + rstep, nnrow = nrow + rstep;
    if (nrow < rows) fetch(nsrc, nxt);
    if (nnrow < rows) nsrc = xmap ? (int64_t)xmap[nnrow] : nnrow;
    float v[CH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
        Vec8<T>::cvt(cur[i], v[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[i][j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
      }
    }
    const float mean = group_sum<NW>(s, red) * inv;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; ss += d * d; }
      }
    }
    const float rstd = rsqrtf(group_sum<NW>(ss, red) * inv + eps);
    T* yr = y + row * (int64_t)cols;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = (v[i][j] - mean) * rstd * wv[i][j] + bv[i][j];
          o[j] = GELU ? gelu_erf(t) : t;
        }
        store_sel<T, NT>(yr + c, o);
        if constexpr (Q8) {  // the output as the next reader sees it (rounded to T); v is free now
#pragma unroll
          for (int j = 0; j < 8; ++j) { v[i][j] = (float)(T)o[j]; amax = fmaxf(amax, fabsf(v[i][j])); }
        }
      }
    }
    if constexpr (Q8) {
      const float sc = fp8_row_scale(group_max<NW>(amax, red));
      const float qinv = 1.0f / sc;
      uint8_t* qr = q8 + row * (int64_t)cols;
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = (tig + G * i) * 8;
        if (c < cols) *reinterpret_cast<u32x2*>(qr + c) = fp8_pack8(v[i], qinv);
      }
      if (tig == 0) q8_scale[row] = sc;
    }
    if (tig == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) cur[i] = nxt[i];
  }
}

// y = LayerNorm_F(bf16(gelu(h0) * h1)) straight from the two pre-activations of the GeGLU up-projection (row stride ldh: they are
// the halves of one [rows, 2 * cols] matrix written by a PLAIN two-segment GEMM).  Takes the exact-erf GELU (16 VALU instructions
// per element, 13 % of the fused-epilogue GEMM: one wave per SIMD cannot hide it) out of the GEMM's epilogue and into a kernel
// that is HBM-bound with VALU to spare; the product is rounded to bf16 before the statistics, exactly as the backward
// (ln_geglu_bwd_kernel) re-creates it.  Algorithmic bytes: 2 reads + 1 write of [rows, cols] bf16.
template <int CH, int NW, bool Q8 = false>  // Q8: see ln_fwd_kernel
__global__ __launch_bounds__(NW == 1 ? 256 : 64 * NW) void ln_geglu_fwd_kernel(const bf16_t* __restrict__ h0, const bf16_t* __restrict__ h1,
                                                           int64_t ldh, const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                           bf16_t* __restrict__ y, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, int64_t rows, int cols, float eps,
                                                           uint8_t* __restrict__ q8 = nullptr, float* __restrict__ q8_scale = nullptr) {
  __shared__ float red[4];
  constexpr int G = 64 * NW;
  const int tig = (NW == 1) ? (threadIdx.x & 63) : threadIdx.x;
  const int64_t row0 = (NW == 1) ? ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) : blockIdx.x;
  const int64_t rstep = (NW == 1) ? (int64_t)gridDim.x * 4 : gridDim.x;
  const float inv = 1.0f / (float)cols;
  float wv[CH][8], bv[CH][8];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = (tig + G * i) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { wv[i][j] = 1.f; bv[i][j] = 0.f; }
    if (c < cols) {
      if (w) Vec8<bf16_t>::load(w + c, wv[i]);
      if (b) Vec8<bf16_t>::load(b + c, bv[i]);
    }
  }
  bf16x8 c0[CH], c1[CH], n0[CH], n1[CH];
  if (row0 < rows) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
        c0[i] = Vec8<bf16_t>::ldraw_nt(h0 + row0 * ldh + c);
        c1[i] = Vec8<bf16_t>::ldraw_nt(h1 + row0 * ldh + c);
      }
    }
  }
  for (int64_t row = row0; row < rows; row += rstep) {
    const int64_t nrow = row + rstep;
    if (nrow < rows) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = (tig + G * i) * 8;
        if (c < cols) {
          n0[i] = Vec8<bf16_t>::ldraw_nt(h0 + nrow * ldh + c);
          n1[i] = Vec8<bf16_t>::ldraw_nt(h1 + nrow * ldh + c);
        }
      }
    }
    float v[CH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
        float a[8], bb[8];
        Vec8<bf16_t>::cvt(c0[i], a);
        Vec8<bf16_t>::cvt(c1[i], bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[i][j] = (float)(bf16_t)(gelu_erf(a[j]) * bb[j]);
          s += v[i][j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
      }
    }
    const float mean = group_sum<NW>(s, red) * inv;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; ss += d * d; }
      }
    }
    const float rstd = rsqrtf(group_sum<NW>(ss, red) * inv + eps);
    bf16_t* yr = y + row * (int64_t)cols;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * wv[i][j] + bv[i][j];
        Vec8<bf16_t>::store_nt(yr + c, o);
        if constexpr (Q8) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { v[i][j] = (float)(bf16_t)o[j]; amax = fmaxf(amax, fabsf(v[i][j])); }
        }
      }
    }
    if constexpr (Q8) {
      const float sc = fp8_row_scale(group_max<NW>(amax, red));
      const float qinv = 1.0f / sc;
      uint8_t* qr = q8 + row * (int64_t)cols;
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = (tig + G * i) * 8;
        if (c < cols) *reinterpret_cast<u32x2*>(qr + c) = fp8_pack8(v[i], qinv);
      }
      if (tig == 0) q8_scale[row] = sc;
    }
    if (tig == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) { c0[i] = n0[i]; c1[i] = n1[i]; }
  }
}

// dx = rstd * (dy*w - mean(dy*w) - xhat * mean(dy*w*xhat));  dw = sum_rows dy*xhat;  db = sum_rows dy.
// Partial dw/db of each workgroup go to ws[gridDim.x][2][cols] (fp32); ln_bwd_reduce_kernel folds them.
template <typename T, int CH, int NW, bool GELU, bool NT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const T* __restrict__ w, const T* __restrict__ b,
                                                     const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const T* __restrict__ add,
                                                     T* __restrict__ dx, float* __restrict__ ws, int64_t rows, int cols,
                                                     const int* __restrict__ xmap = nullptr) {
  // xmap (nullable, ABI 9; see ln_fwd_kernel): x, add and dx are rows xmap[r] of LARGER matrices (dy / mean / rstd are indexed by r);
  // a row without a source (< 0) is a row of zeros for x, has no `add` and is not stored.  dx may be `add` (in place: only the mapped
  // rows change -- the rows of dropped samples keep the gradient of the skip connection).
  extern __shared__ __attribute__((aligned(16))) float smem[];  // NW==1: [4][cols] ; NW==4: [4]
  float* red = smem;
  constexpr int G = 64 * NW;
  const int tig = (NW == 1) ? (threadIdx.x & 63) : threadIdx.x;
  const int wid = threadIdx.x >> 6;
  const int64_t row0 = (NW == 1) ? ((int64_t)blockIdx.x * 4 + wid) : blockIdx.x;
  const int64_t rstep = (NW == 1) ? (int64_t)gridDim.x * 4 : gridDim.x;
  const float inv = 1.0f / (float)cols;

  float wv[CH][8], bv[CH][8], dwa[CH][8], dba[CH][8];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = (tig + G * i) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { wv[i][j] = 1.f; bv[i][j] = 0.f; dwa[i][j] = 0.f; dba[i][j] = 0.f; }
    if (c < cols) {
      if (w) Vec8<T>::load(w + c, wv[i]);
      if (GELU && b) Vec8<T>::load(b + c, bv[i]);
    }
  }
  typedef typename Vec8<T>::raw_t raw_t;
  raw_t curx[CH], curg[CH], nxtx[CH], nxtg[CH], addv[CH];
  // (uniform per row group; the entry of row r + 2 steps is requested while row r is computed: see ln_fwd_kernel)
  int64_t src = row0 < rows ? (xmap ? (int64_t)xmap[row0] : row0) : -1;
  int64_t nsrc = row0 + rstep < rows ? (xmap ? (int64_t)xmap[row0 + rstep] : row0 + rstep) : -1, nnsrc = -1;
  if (row0 < rows) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
        if (src >= 0) curx[i] = ldraw_sel<T, NT>(x + src * (int64_t)cols + c);
        else curx[i] = raw_t{};
        curg[i] = ldraw_sel<T, NT>(dy + row0 * (int64_t)cols + c);
      }
    }
  }
  for (int64_t row = row0; row < rows; row += rstep) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    const int64_t nrow = row + rstep, nnrow = nrow + rstep;
    if (nnrow < rows) nnsrc = xmap ? (int64_t)xmap[nnrow] : nnrow;
#pragma unroll
    for (int i = 0; i < CH; ++i) {  // residual-path gradient of this row + both operands of the next row
      const int c = (tig + G * i) * 8;
      if (c < cols) {
        if (add && src >= 0) addv[i] = ldraw_sel<T, NT>(add + src * (int64_t)cols + c);
        if (nrow < rows) {
          if (nsrc >= 0) nxtx[i] = ldraw_sel<T, NT>(x + nsrc * (int64_t)cols + c);
          else nxtx[i] = raw_t{};
          nxtg[i] = ldraw_sel<T, NT>(dy + nrow * (int64_t)cols + c);
        }
      }
    }
    float xh[CH][8], g[CH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
        Vec8<T>::cvt(curx[i], xh[i]);
        Vec8<T>::cvt(curg[i], g[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[i][j] = (xh[i][j] - mean) * rstd;
          if (GELU) g[i][j] *= gelu_erf_grad(xh[i][j] * wv[i][j] + bv[i][j]);
          dwa[i][j] += g[i][j] * xh[i][j];
          dba[i][j] += g[i][j];
          const float gw = g[i][j] * wv[i][j];
          g[i][j] = gw;
          s1 += gw;
          s2 += gw * xh[i][j];
        }
      }
    }
    group_sum2<NW>(s1, s2, red);
    const float c1 = s1 * inv, c2 = s2 * inv;
    T* dr = dx + (src >= 0 ? src : 0) * (int64_t)cols;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols && src >= 0) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (g[i][j] - c1 - xh[i][j] * c2);
        if (add) {
          float av[8];
          Vec8<T>::cvt(addv[i], av);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += av[j];
        }
        store_sel<T, NT>(dr + c, o);
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) { curx[i] = nxtx[i]; curg[i] = nxtg[i]; }
    src = nsrc;
    nsrc = nnsrc;
  }
  if (ws == nullptr) return;  // uniform
  float* wsb = ws + (int64_t)blockIdx.x * 2 * cols;
  if (NW == 1) {
    // fold the 4 waves (same column mapping, different rows) through LDS, dw then db
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = (tig + G * i) * 8;
        if (c < cols) {
#pragma unroll
          for (int j = 0; j < 8; ++j) smem[wid * cols + c + j] = pass == 0 ? dwa[i][j] : dba[i][j];
        }
      }
      __syncthreads();
      for (int c = threadIdx.x; c < cols; c += 256)
        wsb[pass * cols + c] = smem[c] + smem[cols + c] + smem[2 * cols + c] + smem[3 * cols + c];
    }
  } else {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { wsb[c + j] = dwa[i][j]; wsb[cols + c + j] = dba[i][j]; }
      }
    }
  }
}

// LayerNorm(F) backward fused with the GeGLU backward (transformer_layer.py:64-67,111-118): the FFN's inner
// sub-LayerNorm takes g = gelu(h0) * h1; g is RECOMPUTED from the saved h0/h1 (never stored for backward) and
//   dg  = rstd * (dy*w - mean(dy*w) - xhat * mean(dy*w*xhat)),   xhat = (g - mean) * rstd
//   dh0 = dg * h1 * gelu'(h0),   dh1 = dg * gelu(h0)
// Algorithmic bytes: 3 reads + 2 writes of [rows, cols] bf16 (the unfused pair moved 8 passes).
template <int CH, int NW>
__global__ __launch_bounds__(NW == 1 ? 256 : 64 * NW) void ln_geglu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ h0,
                                                           const bf16_t* __restrict__ h1, const bf16_t* __restrict__ w,
                                                           const float* __restrict__ mean_in,
                                                           const float* __restrict__ rstd_in, bf16_t* __restrict__ dh0,
                                                           bf16_t* __restrict__ dh1, int64_t ldd, int64_t ldh, float* __restrict__ ws,
                                                           int64_t rows, int cols) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // NW==1: [4][cols] ; NW==4: [4]
  float* red = smem;
  constexpr int G = 64 * NW;
  const int tig = (NW == 1) ? (threadIdx.x & 63) : threadIdx.x;
  const int wid = threadIdx.x >> 6;
  const int64_t row0 = (NW == 1) ? ((int64_t)blockIdx.x * 4 + wid) : blockIdx.x;
  const int64_t rstep = (NW == 1) ? (int64_t)gridDim.x * 4 : gridDim.x;
  const float inv = 1.0f / (float)cols;
  float dwa[CH][8], dba[CH][8];
  bf16x8 wraw[CH];  // the LayerNorm weight stays packed (4 registers per 8 columns; unpacked where it is used: one VALU op per element)
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = (tig + G * i) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { dwa[i][j] = 0.f; dba[i][j] = 0.f; }
#pragma unroll
    for (int j = 0; j < 8; ++j) wraw[i][j] = (bf16_t)1.0f;
    if (c < cols && w) wraw[i] = Vec8<bf16_t>::ldraw(w + c);
  }
  // Registers (<= 256 for two waves per SIMD): the row is kept as raw bf16 (h0, h1) + fp32 dy*w, Phi(h0) and gelu'(h0) across the two
  // block reductions, so that the NEXT row's three operands can already be in flight.
  bf16x8 r0[CH], r1[CH], rg[CH], n0[CH], n1[CH], ng[CH];
  if (row0 < rows) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
        r0[i] = Vec8<bf16_t>::ldraw_nt(h0 + row0 * ldh + c);
        r1[i] = Vec8<bf16_t>::ldraw_nt(h1 + row0 * ldh + c);
        rg[i] = Vec8<bf16_t>::ldraw_nt(dy + row0 * (int64_t)cols + c);
      }
    }
  }
  for (int64_t row = row0; row < rows; row += rstep) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    const int64_t base = row * (int64_t)cols;
    const int64_t nrow = row + rstep;
    if (nrow < rows) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = (tig + G * i) * 8;
        if (c < cols) {
          n0[i] = Vec8<bf16_t>::ldraw_nt(h0 + nrow * ldh + c);
          n1[i] = Vec8<bf16_t>::ldraw_nt(h1 + nrow * ldh + c);
          ng[i] = Vec8<bf16_t>::ldraw_nt(dy + nrow * (int64_t)cols + c);
        }
      }
    }
    float gw[CH][8], cdfs[CH][8], gps[CH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
        float a[8], b[8];
        Vec8<bf16_t>::cvt(r0[i], a);
        Vec8<bf16_t>::cvt(r1[i], b);
        Vec8<bf16_t>::cvt(rg[i], gw[i]);
        float wv[8];
        asm volatile("" : "+v"(wraw[i]));  // (keeps the unpacking inside the row loop: hoisted, it is 24 registers again)
        Vec8<bf16_t>::cvt(wraw[i], wv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float cdf, pdf;
          gelu_parts(a[j], cdf, pdf);
          cdfs[i][j] = cdf;
          gps[i][j] = cdf + a[j] * pdf;  // gelu'(h0): kept for the output pass (round 5: that pass took a second exponential per element)
          // the forward rounded g to bf16 before the LayerNorm statistics were taken
          const float gval = (float)(bf16_t)(a[j] * cdf * b[j]);
          const float xh = (gval - mean) * rstd;
          const float d = gw[i][j];
          dwa[i][j] += d * xh;
          dba[i][j] += d;
          const float t = d * wv[j];
          gw[i][j] = t;
          s1 += t;
          s2 += t * xh;
        }
      }
    }
    group_sum2<NW>(s1, s2, red);
    const float c1 = s1 * inv, c2 = s2 * inv;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
        float a[8], b[8], o0[8], o1[8];
        Vec8<bf16_t>::cvt(r0[i], a);
        Vec8<bf16_t>::cvt(r1[i], b);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float ge = a[j] * cdfs[i][j];  // Phi(h0) and gelu'(h0) are kept from the first pass: no transcendental here
          const float xh = ((float)(bf16_t)(ge * b[j]) - mean) * rstd;
          // the unfused path rounded dg to bf16 between the two kernels; keep fp32 here
          const float dg = rstd * (gw[i][j] - c1 - xh * c2);
          o0[j] = dg * b[j] * gps[i][j];
          o1[j] = dg * ge;
        }
        Vec8<bf16_t>::store_nt(dh0 + row * ldd + c, o0);
        Vec8<bf16_t>::store_nt(dh1 + row * ldd + c, o1);
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) { r0[i] = n0[i]; r1[i] = n1[i]; rg[i] = ng[i]; }
  }
  if (ws == nullptr) return;  // uniform
  float* wsb = ws + (int64_t)blockIdx.x * 2 * cols;
  if (NW == 1) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = (tig + G * i) * 8;
        if (c < cols) {
#pragma unroll
          for (int j = 0; j < 8; ++j) smem[wid * cols + c + j] = pass == 0 ? dwa[i][j] : dba[i][j];
        }
      }
      __syncthreads();
      for (int c = threadIdx.x; c < cols; c += 256)
        wsb[pass * cols + c] = smem[c] + smem[cols + c] + smem[2 * cols + c] + smem[3 * cols + c];
    }
  } else {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (tig + G * i) * 8;
      if (c < cols) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { wsb[c + j] = dwa[i][j]; wsb[cols + c + j] = dba[i][j]; }
      }
    }
  }
}

inline int ln_grid(int64_t rows, int nw, int cap) {
  int64_t blocks = nw == 1 ? (rows + 3) / 4 : rows;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

template <typename T, bool GELU>
int ln_fwd_dispatch(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, int64_t rows,
                    int cols, float eps, hipStream_t s, const int* xmap = nullptr) {
  const T* X = (const T*)x; const T* W = (const T*)w; const T* B = (const T*)b; T* Y = (T*)y;
  // cache policy: non-temporal row accesses for a TRAINING pass (the caller wants the statistics: a backward follows) over a matrix of
  // >= 64 MiB -- it streams once, and what it left in L2 / Infinity Cache only displaced the next GEMM's panels (-2 % on the headline
  // step with the hint on every row-wise kernel); inference keeps the default policy: there the consumer finds the row matrix in the
  // caches (+1 ... 5 % with the hint at the image tower's sizes; profiles/r4_nontemporal_ab.txt)
  const bool nt = mean != nullptr && (int64_t)rows * cols * (int64_t)sizeof(T) >= ((int64_t)64 << 20);
#define LN_F(CH, NW)                                                                                                                      \
  do {                                                                                                                                    \
    if (nt) hipLaunchKernelGGL((ln_fwd_kernel<T, CH, NW, GELU, true>), dim3(ln_grid(rows, NW, g_ln_blocks_fwd)), dim3(256), 0, s, X, W, B, Y, mean, rstd, rows, cols, eps, (uint8_t*)nullptr, (float*)nullptr, xmap); \
    else hipLaunchKernelGGL((ln_fwd_kernel<T, CH, NW, GELU, false>), dim3(ln_grid(rows, NW, g_ln_blocks_fwd)), dim3(256), 0, s, X, W, B, Y, mean, rstd, rows, cols, eps, (uint8_t*)nullptr, (float*)nullptr, xmap); \
  } while (0)
  if (cols <= 512) LN_F(1, 1);
  else if (cols <= 1024) LN_F(2, 1);
  else if (cols <= 1536) LN_F(3, 1);
  else if (cols <= 2048) LN_F(4, 1);
  else if (cols <= 4096) LN_F(2, 4);
  else if (cols <= 6144) LN_F(3, 4);
  else if (cols <= 8192) LN_F(4, 4);
  else { op_set_error("layernorm: cols %d > 8192 unsupported", cols); return OP_ENOTSUP; }
#undef LN_F
  OP_LAUNCH_CHECK();
  return OP_OK;
}

template <typename T, bool GELU>
int ln_bwd_dispatch(const void* dy, const void* x, const void* w, const void* b, const float* mean, const float* rstd,
                    const void* add, void* dx, void* dw, void* db, float* ws, int64_t rows, int cols, int accumulate,
                    hipStream_t s, const int* xmap = nullptr) {
  const T* DY = (const T*)dy; const T* X = (const T*)x; const T* W = (const T*)w; const T* B = (const T*)b; T* DX = (T*)dx;
  const T* ADD = (const T*)add;
  int grid = 0;
  float* wsk = (dw || db) ? ws : nullptr;
  const bool nt = (int64_t)rows * cols * (int64_t)sizeof(T) >= ((int64_t)64 << 20);  // (see ln_fwd_dispatch)
#define LN_B(CH, NW)                                                                                        \
  do {                                                                                                      \
    grid = ln_grid(rows, NW, g_ln_blocks_bwd);                                                                               \
    size_t sh = (NW == 1) ? (size_t)4 * cols * sizeof(float) : 64;                                          \
    if (nt) hipLaunchKernelGGL((ln_bwd_kernel<T, CH, NW, GELU, true>), dim3(grid), dim3(256), sh, s, DY, X, W, B, mean,   \
                       rstd, ADD, DX, wsk, rows, cols, xmap);                                                    \
    else hipLaunchKernelGGL((ln_bwd_kernel<T, CH, NW, GELU, false>), dim3(grid), dim3(256), sh, s, DY, X, W, B, mean,   \
                       rstd, ADD, DX, wsk, rows, cols, xmap);                                                    \
  } while (0)
  if (cols <= 512) LN_B(1, 1);
  else if (cols <= 1024) LN_B(2, 1);
  else if (cols <= 1536) LN_B(3, 1);
  else if (cols <= 2048) LN_B(4, 1);
  else if (cols <= 4096) LN_B(2, 4);
  else if (cols <= 6144) LN_B(3, 4);
  else if (cols <= 8192) LN_B(4, 4);
  else { op_set_error("layernorm: cols %d > 8192 unsupported", cols); return OP_ENOTSUP; }
#undef LN_B
  OP_LAUNCH_CHECK();
  if (wsk) {
    hipLaunchKernelGGL((partials_reduce3_kernel<T>), dim3(ceil_div(cols, 32), 2), dim3(256), 0, s, ws, ws + cols,
                       (const float*)nullptr, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (T*)dw,
                       (T*)db, (T*)nullptr, grid, (int64_t)2 * cols, cols, accumulate);
    OP_LAUNCH_CHECK();
  }
  return OP_OK;
}

}  // namespace

extern "C" {


// Bytes of fp32 workspace op_layernorm_bwd needs for dw/db partials.
int64_t op_layernorm_bwd_workspace_bytes(int64_t rows, int64_t cols) {
  (void)rows;
  return (int64_t)LN_MAX_BLOCKS * 2 * cols * (int64_t)sizeof(float);
}

int op_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, int64_t rows,
                     int64_t cols, float eps, int act_gelu, int dtype, const int* x_rows, void* stream) {
  OP_CHECK_ARG(x && y, "layernorm_fwd: null x/y");
  OP_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0, "layernorm_fwd: cols=%lld must be a positive multiple of 8",
               (long long)cols);
  if (rows == 0) return OP_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OP_DT_BF16)
    return act_gelu ? ln_fwd_dispatch<bf16_t, true>(x, w, b, y, mean, rstd, rows, (int)cols, eps, s, x_rows)
                    : ln_fwd_dispatch<bf16_t, false>(x, w, b, y, mean, rstd, rows, (int)cols, eps, s, x_rows);
  if (dtype == OP_DT_F32)
    return act_gelu ? ln_fwd_dispatch<float, true>(x, w, b, y, mean, rstd, rows, (int)cols, eps, s, x_rows)
                    : ln_fwd_dispatch<float, false>(x, w, b, y, mean, rstd, rows, (int)cols, eps, s, x_rows);
  op_set_error("layernorm_fwd: bad dtype %d", dtype);
  return OP_EINVAL;
}

// op_layernorm_fwd for bf16 without GELU that ALSO writes the output row-quantised to fp8 e4m3: q8 [rows, cols] bytes and
// q8_scale [rows] fp32, bit-identical to op_quant_fp8_rows(y) -- the activation operand of op_gemm_nt_fp8 without a pass of its own
// (the sub-LayerNorm in front of the FFN, transformer_layer.py:196-199, when the opt-in fp8 forward is on).
int op_layernorm_fwd_q8(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, void* q8, float* q8_scale,
                        int64_t rows, int64_t cols, float eps, void* stream) {
  OP_CHECK_ARG(x && y && q8 && q8_scale, "layernorm_fwd_q8: null pointer");
  OP_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 8192, "layernorm_fwd_q8: cols=%lld unsupported", (long long)cols);
  if (rows == 0) return OP_OK;
  hipStream_t s = (hipStream_t)stream;
#define LN_Q(CH, NW)                                                                                                                         \
  hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, CH, NW, false, false, true>), dim3(ln_grid(rows, NW, g_ln_blocks_fwd)), dim3(256), 0, s,        \
                     (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, mean, rstd, rows, (int)cols, eps, (uint8_t*)q8, q8_scale)
  if (cols <= 512) LN_Q(1, 1);
  else if (cols <= 1024) LN_Q(2, 1);
  else if (cols <= 1536) LN_Q(3, 1);
  else if (cols <= 2048) LN_Q(4, 1);
  else if (cols <= 4096) LN_Q(2, 4);
  else if (cols <= 6144) LN_Q(3, 4);
  else LN_Q(4, 4);
#undef LN_Q
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// dx = LN backward (+ add, the gradient arriving through the residual path, optional and may alias dx)
int op_layernorm_bwd(const void* dy, const void* x, const void* w, const void* b, const float* mean, const float* rstd,
                     const void* add, void* dx, void* dw, void* db, void* workspace, int64_t rows, int64_t cols,
                     int act_gelu, int accumulate, int dtype, const int* x_rows, void* stream) {
  OP_CHECK_ARG(dy && x && dx && mean && rstd, "layernorm_bwd: null pointer");
  OP_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0, "layernorm_bwd: cols=%lld must be a positive multiple of 8",
               (long long)cols);
  OP_CHECK_ARG(!(dw || db) || workspace, "layernorm_bwd: dw/db requested without workspace");
  if (rows == 0) return OP_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OP_DT_BF16)
    return act_gelu ? ln_bwd_dispatch<bf16_t, true>(dy, x, w, b, mean, rstd, add, dx, dw, db, (float*)workspace, rows,
                                                    (int)cols, accumulate, s, x_rows)
                    : ln_bwd_dispatch<bf16_t, false>(dy, x, w, b, mean, rstd, add, dx, dw, db, (float*)workspace, rows,
                                                     (int)cols, accumulate, s, x_rows);
  if (dtype == OP_DT_F32)
    return act_gelu ? ln_bwd_dispatch<float, true>(dy, x, w, b, mean, rstd, add, dx, dw, db, (float*)workspace, rows,
                                                   (int)cols, accumulate, s, x_rows)
                    : ln_bwd_dispatch<float, false>(dy, x, w, b, mean, rstd, add, dx, dw, db, (float*)workspace, rows,
                                                    (int)cols, accumulate, s, x_rows);
  op_set_error("layernorm_bwd: bad dtype %d", dtype);
  return OP_EINVAL;
}

// Backward of  y = LayerNorm_F(gelu(h0) * h1) * w + b  w.r.t. h0, h1, w, b in one pass (bf16 only); mean/rstd are the
// forward statistics of g = gelu(h0) * h1.  Replaces the LayerNorm backward + GeGLU backward pair of the FFN
// (one_peace/models/transformer/transformer_layer.py:64-67,111-118); g itself is not needed.
// y [rows, cols] = LayerNorm(bf16(gelu(h0) * h1)) (+ mean / rstd, nullable); h0, h1: row stride ldh (0 = cols).
int op_ln_geglu_fwd(const void* h0, const void* h1, int64_t ldh, const void* w, const void* b, void* y, float* mean, float* rstd,
                    int64_t rows, int64_t cols, float eps, void* stream) {
  OP_CHECK_ARG(h0 && h1 && y, "ln_geglu_fwd: null pointer");
  if (ldh <= 0) ldh = cols;
  OP_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 8192 && ldh >= cols && ldh % 8 == 0, "ln_geglu_fwd: cols=%lld ldh=%lld unsupported",
               (long long)cols, (long long)ldh);
  if (rows == 0) return OP_OK;
  hipStream_t s = (hipStream_t)stream;
#define LNG_F(CH, NW)                                                                                                       \
  hipLaunchKernelGGL((ln_geglu_fwd_kernel<CH, NW>), dim3(ln_grid(rows, NW, OP_LN_GEGLU_BLOCKS_FWD)), dim3(NW == 1 ? 256 : 64 * NW), 0, s, \
                     (const bf16_t*)h0, (const bf16_t*)h1, ldh, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, mean, rstd, rows, \
                     (int)cols, eps)
  if (cols <= 512) LNG_F(1, 1);
  else if (cols <= 1024) LNG_F(2, 1);
  else if (cols <= 1536) LNG_F(3, 1);
  else if (cols <= 2048) LNG_F(4, 1);
  else if (cols <= 4096) LNG_F(2, 4);
  else if (cols <= 6144) LNG_F(3, 4);
  else LNG_F(4, 4);
#undef LNG_F
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// op_ln_geglu_fwd that also writes the row-quantised fp8 copy of its output (see op_layernorm_fwd_q8): the operand of the fp8 down-projection.
int op_ln_geglu_fwd_q8(const void* h0, const void* h1, int64_t ldh, const void* w, const void* b, void* y, float* mean, float* rstd, void* q8,
                       float* q8_scale, int64_t rows, int64_t cols, float eps, void* stream) {
  OP_CHECK_ARG(h0 && h1 && y && q8 && q8_scale, "ln_geglu_fwd_q8: null pointer");
  if (ldh <= 0) ldh = cols;
  OP_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 8192 && ldh >= cols && ldh % 8 == 0, "ln_geglu_fwd_q8: cols=%lld ldh=%lld unsupported",
               (long long)cols, (long long)ldh);
  if (rows == 0) return OP_OK;
  hipStream_t s = (hipStream_t)stream;
#define LNG_Q(CH, NW)                                                                                                       \
  hipLaunchKernelGGL((ln_geglu_fwd_kernel<CH, NW, true>), dim3(ln_grid(rows, NW, OP_LN_GEGLU_BLOCKS_FWD)), dim3(NW == 1 ? 256 : 64 * NW), 0, s, \
                     (const bf16_t*)h0, (const bf16_t*)h1, ldh, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, mean, rstd, rows, \
                     (int)cols, eps, (uint8_t*)q8, q8_scale)
  if (cols <= 512) LNG_Q(1, 1);
  else if (cols <= 1024) LNG_Q(2, 1);
  else if (cols <= 1536) LNG_Q(3, 1);
  else if (cols <= 2048) LNG_Q(4, 1);
  else if (cols <= 4096) LNG_Q(2, 4);
  else if (cols <= 6144) LNG_Q(3, 4);
  else LNG_Q(4, 4);
#undef LNG_Q
  OP_LAUNCH_CHECK();
  return OP_OK;
}

int op_ln_geglu_bwd(const void* dy, const void* h0, const void* h1, const void* w, const float* mean, const float* rstd,
                    void* dh0, void* dh1, int64_t ldd, int64_t ldh, void* dw, void* db, void* workspace, int64_t rows, int64_t cols,
                    int accumulate, void* stream) {
  OP_CHECK_ARG(dy && h0 && h1 && dh0 && dh1 && mean && rstd, "ln_geglu_bwd: null pointer");
  if (ldd <= 0) ldd = cols;
  if (ldh <= 0) ldh = cols;
  OP_CHECK_ARG(ldd >= cols && ldd % 8 == 0 && ldh >= cols && ldh % 8 == 0, "ln_geglu_bwd: row strides %lld / %lld", (long long)ldd, (long long)ldh);
  OP_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 8192, "ln_geglu_bwd: cols=%lld unsupported", (long long)cols);
  OP_CHECK_ARG(!(dw || db) || workspace, "ln_geglu_bwd: dw/db requested without workspace");
  if (rows == 0) return OP_OK;
  hipStream_t s = (hipStream_t)stream;
  float* wsk = (dw || db) ? (float*)workspace : nullptr;
  int grid = 0;
#define LNG_B(CH, NW)                                                                                              \
  do {                                                                                                             \
    grid = ln_grid(rows, NW, OP_LN_GEGLU_BLOCKS_BWD);                                                                     \
    size_t sh = (NW == 1) ? (size_t)4 * cols * sizeof(float) : 64;                                                 \
    hipLaunchKernelGGL((ln_geglu_bwd_kernel<CH, NW>), dim3(grid), dim3(NW == 1 ? 256 : 64 * NW), sh, s,           \
                       (const bf16_t*)dy, (const bf16_t*)h0, (const bf16_t*)h1, (const bf16_t*)w, mean, rstd,      \
                       (bf16_t*)dh0, (bf16_t*)dh1, ldd, ldh, wsk, rows, (int)cols);                                \
  } while (0)
  if (cols <= 512) LNG_B(1, 1);
  else if (cols <= 1024) LNG_B(2, 1);
  else if (cols <= 1536) LNG_B(3, 1);
  else if (cols <= 2048) LNG_B(4, 1);
  else if (cols <= 4096) LNG_B(2, 4);
  else if (cols <= 6144) LNG_B(3, 4);
  else LNG_B(4, 4);
#undef LNG_B
  OP_LAUNCH_CHECK();
  if (wsk) {
    hipLaunchKernelGGL((partials_reduce3_kernel<bf16_t>), dim3(ceil_div(cols, 32), 2), dim3(256), 0, s, wsk, wsk + cols,
                       (const float*)nullptr, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const bf16_t*)nullptr,
                       (bf16_t*)dw, (bf16_t*)db, (bf16_t*)nullptr, grid, (int64_t)2 * cols, (int)cols, accumulate);
    OP_LAUNCH_CHECK();
  }
  return OP_OK;
}

}  // extern "C"
