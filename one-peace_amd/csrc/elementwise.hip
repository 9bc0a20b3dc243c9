// HBM-bound helper kernels of the ONE-PEACE hot path (gfx950): transposes, column reductions, GeGLU backward,
// layer-scale/drop-path residual backward, L2-normalise, InfoNCE rows, AdamW, relative-position bias tables.
// All move 16-byte vectors per lane, accumulate in fp32 and are deterministic (two-stage reductions, no
// floating-point atomics except the relative-position table scatter which is documented below).
#include "common.h"
#include <string.h>

namespace {

constexpr int CS_MAX_PARTS = 256;

// ------------------------------------------------------------------------------------------------------------
// bf16 2-D transpose  out[c][r] = in[r][c]   (64x64 tiles through LDS, 8-byte global accesses on both sides)
// ------------------------------------------------------------------------------------------------------------
// scale (nullable): out[c][r] = bf16(scale[r] * in[r][c]) -- the dgrad copy of a residual branch's last Linear with the layer scale
// folded in (ops._transposed(..., scale=gamma): dx = (rowscale * dout) . (gamma o W), transformer_layer.py:70-88)
__device__ __forceinline__ void transpose_tile(int bx, int by, const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                        int rows, int cols, int64_t ld_in, int64_t ld_out,
                                                        const bf16_t* __restrict__ scale = nullptr) {
  __shared__ bf16_t tile[64][64 + 2];
  const int r0 = by * 64, c0 = bx * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, 4 elements each along the fast axis
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + k * 16, c = c0 + tx * 4;
    if (r < rows) {
      const float sc = scale ? (float)scale[r] : 1.f;
      if (c + 3 < cols && (ld_in & 3) == 0) {
        bf16x4 v = *reinterpret_cast<const bf16x4*>(in + (int64_t)r * ld_in + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[ty + k * 16][tx * 4 + j] = scale ? (bf16_t)(sc * (float)v[j]) : v[j];
      } else {
        for (int j = 0; j < 4; ++j)
          if (c + j < cols) tile[ty + k * 16][tx * 4 + j] = scale ? (bf16_t)(sc * (float)in[(int64_t)r * ld_in + c + j]) : in[(int64_t)r * ld_in + c + j];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + k * 16, r = r0 + tx * 4;  // output row = input column
    if (c < cols) {
      if (r + 3 < rows && (ld_out & 3) == 0) {
        bf16x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = tile[tx * 4 + j][ty + k * 16];
        *reinterpret_cast<bf16x4*>(out + (int64_t)c * ld_out + r) = v;
      } else {
        for (int j = 0; j < 4; ++j)
          if (r + j < rows) out[(int64_t)c * ld_out + r + j] = tile[tx * 4 + j][ty + k * 16];
      }
    }
  }
}


__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int rows, int cols,
                                                        int64_t ld_in, int64_t ld_out, const bf16_t* __restrict__ scale) {
  transpose_tile(blockIdx.x, blockIdx.y, in, out, rows, cols, ld_in, ld_out, scale);
}

// Many transposes in ONE launch (the dgrad copies of all weights after an optimiser step: 520 matrices at 4B).  Descriptor i
// covers tiles [tile0[i], tile0[i+1]) of the launch; a workgroup finds its matrix by binary search.
struct TransposeDesc { const bf16_t* in; bf16_t* out; int rows, cols; int64_t ld_in, ld_out; int tile0, tiles_x; const bf16_t* scale; };
__global__ __launch_bounds__(256) void transpose_batched_kernel(const TransposeDesc* __restrict__ table, int n) {
  const int b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].tile0 <= b) lo = mid; else hi = mid - 1;
  }
  const TransposeDesc d = table[lo];
  const int local = b - d.tile0;
  transpose_tile(local % d.tiles_x, local / d.tiles_x, d.in, d.out, d.rows, d.cols, d.ld_in, d.ld_out, d.scale);
}

// ------------------------------------------------------------------------------------------------------------
// column sums:  part[p][n] = sum over this block's rows of  rs[m] * x[m][n] * (y ? y[m][n] : 1)
// grid = (ceil(N / 512), parts); block 256 = 4 waves x 64 lanes x 8 columns
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y,
                                                             const float* __restrict__ rowscale, int rps,
                                                             float* __restrict__ part, int64_t M, int N, int64_t ld) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + lane * 8;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < N) {
    const int64_t m0 = (int64_t)blockIdx.y * 4 + wid, step = (int64_t)gridDim.y * 4;
    if (!y && !rowscale) {
      // plain column sums (the q / v bias gradients: 2 x 225 MB per layer): FOUR rows in flight per wave -- one load -> add per
      // trip left the kernel at 4.0 TB/s (round 3); the rows are added in the same order as before
      int64_t m = m0;
      for (; m + 3 * step < M; m += 4 * step) {
        typename Vec8<bf16_t>::raw_t r0 = Vec8<bf16_t>::ldraw(x + m * ld + c), r1 = Vec8<bf16_t>::ldraw(x + (m + step) * ld + c),
                                     r2 = Vec8<bf16_t>::ldraw(x + (m + 2 * step) * ld + c), r3 = Vec8<bf16_t>::ldraw(x + (m + 3 * step) * ld + c);
        float v0[8], v1[8], v2[8], v3[8];
        Vec8<bf16_t>::cvt(r0, v0);
        Vec8<bf16_t>::cvt(r1, v1);
        Vec8<bf16_t>::cvt(r2, v2);
        Vec8<bf16_t>::cvt(r3, v3);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (((a[j] + v0[j]) + v1[j]) + v2[j]) + v3[j];
      }
      for (; m < M; m += step) {
        float xv[8];
        Vec8<bf16_t>::load(x + m * ld + c, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += xv[j];
      }
    } else {
      for (int64_t m = m0; m < M; m += step) {
        float xv[8];
        Vec8<bf16_t>::load(x + m * ld + c, xv);
        float s = rowscale ? rowscale[m / rps] : 1.f;
        if (y) {
          float yv[8];
          Vec8<bf16_t>::load(y + m * ld + c, yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] += s * xv[j] * yv[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] += s * xv[j];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[wid][lane * 8 + j] = a[j];
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int cc = blockIdx.x * 512 + i;
    if (cc < N) part[(int64_t)blockIdx.y * N + cc] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  }
}

// Layer-scale gradient from the weight gradient's side product (op_gemm_tn_grouped: rowdot) instead of from the branch output.
// grid = ceil(N / 64) blocks of 4 waves: wave w folds slots w, w + 4, ... of its 64 columns (four loads in flight per lane: one thread
// per column walking up to 144 slots one dependent load at a time took 36 us per launch), the four partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void gamma_grad_finish_kernel(const float* __restrict__ rowdot, int slots,
                                                                const bf16_t* __restrict__ b0, const float* __restrict__ g00,
                                                                const bf16_t* __restrict__ b1, const float* __restrict__ g01,
                                                                const bf16_t* __restrict__ b2, const float* __restrict__ g02,
                                                                bf16_t* __restrict__ dgamma, int N, int accumulate) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (n < N) {
    int s = w;
    for (; s + 12 < slots; s += 16) {
      a0 += rowdot[(int64_t)s * N + n];
      a1 += rowdot[(int64_t)(s + 4) * N + n];
      a2 += rowdot[(int64_t)(s + 8) * N + n];
      a3 += rowdot[(int64_t)(s + 12) * N + n];
    }
    for (; s < slots; s += 4) a0 += rowdot[(int64_t)s * N + n];
  }
  part[w][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (w != 0 || n >= N) return;
  float t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);  // the UNSCALED gradient's side product: exact for any gamma, also 0
  if (g00) t += (b0 ? (float)b0[n] : 0.f) * g00[n];
  if (g01) t += (b1 ? (float)b1[n] : 0.f) * g01[n];
  if (g02) t += (b2 ? (float)b2[n] : 0.f) * g02[n];
  dgamma[n] = (bf16_t)(t + (accumulate ? (float)dgamma[n] : 0.f));
}

// ------------------------------------------------------------------------------------------------------------
// residual-branch backward in one pass (transformer_layer.py:70-88):  out = resid + rs[m] * gamma[n] * y[m][n]
//   dbranch[m][n] = rs[m] * gamma[n] * dout[m][n]
//   part[0][p][n] = sum_rows rs * dout * y        (-> dgamma)
//   part[1][p][n] = sum_rows rs * dout            (-> dbias of the branch's last Linear, times gamma in the fold)
// grid = (ceil(N / 512), parts); block 256 = 4 waves x 64 lanes x 8 columns; two rows in flight per wave.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resid_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ y,
                                                        const bf16_t* __restrict__ gamma, const float* __restrict__ rowscale,
                                                        int rps, bf16_t* __restrict__ dbranch, float* __restrict__ part,
                                                        int64_t M, int N, const int* __restrict__ rows) {
  // rows (nullable, ABI 9): row m of dout is row rows[m] of a LARGER matrix (< 0: no row -- reads as zeros); dbranch / y stay packed
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + lane * 8;
  float ag[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ab[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float gv[8] = {1, 1, 1, 1, 1, 1, 1, 1};
  if (c < N) {
    if (gamma) Vec8<bf16_t>::load(gamma + c, gv);
    const int64_t step = (int64_t)gridDim.y * 4;
    const int64_t mfirst = (int64_t)blockIdx.y * 4 + wid;
    // (table entries of the NEXT trip are requested before this trip's rows are used: in one trip, entry -> row is an L2 latency per trip)
    int64_t e0 = (rows && mfirst < M) ? rows[mfirst] : mfirst, e1 = (rows && mfirst + step < M) ? rows[mfirst + step] : mfirst + step;
    for (int64_t m = mfirst; m < M; m += 2 * step) {
      const int64_t m2 = m + step;
      const bool two = m2 < M;
      const int64_t s0 = e0, s1 = two ? e1 : -1;
      const int64_t mn = m + 2 * step;
      if (rows) {
        if (mn < M) e0 = rows[mn];
        if (mn + step < M) e1 = rows[mn + step];
      } else {
        e0 = mn;
        e1 = mn + step;
      }
      const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
      bf16x8 d0 = s0 >= 0 ? Vec8<bf16_t>::ldraw(dout + s0 * N + c) : zero8, d1, y0, y1;
      if (y) y0 = Vec8<bf16_t>::ldraw(y + m * N + c);
      if (two) {
        d1 = s1 >= 0 ? Vec8<bf16_t>::ldraw(dout + s1 * N + c) : zero8;
        if (y) y1 = Vec8<bf16_t>::ldraw(y + m2 * N + c);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1 && !two) break;
        const int64_t mm = h == 0 ? m : m2;
        float d[8], o[8];
        Vec8<bf16_t>::cvt(h == 0 ? d0 : d1, d);
        const float sc = rowscale ? rowscale[mm / rps] : 1.f;
        if (y) {
          float yv[8];
          Vec8<bf16_t>::cvt(h == 0 ? y0 : y1, yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) ag[j] += sc * d[j] * yv[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = sc * d[j];
          ab[j] += t;
          o[j] = t * gv[j];
        }
        Vec8<bf16_t>::store(dbranch + mm * N + c, o);
      }
    }
  }
  if (part == nullptr) return;  // uniform
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 0 && y == nullptr) continue;  // uniform
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wid][lane * 8 + j] = pass == 0 ? ag[j] : ab[j];
    __syncthreads();
    float* dst = part + ((int64_t)pass * gridDim.y + blockIdx.y) * N;
    for (int i = threadIdx.x; i < 512; i += 256) {
      const int cc = blockIdx.x * 512 + i;
      if (cc < N) dst[cc] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// GeGLU backward (transformer_layer.py:64-67):  g = gelu(h0) * h1
//   dh0 = dg * h1 * gelu'(h0),  dh1 = dg * gelu(h0)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* __restrict__ dg, const bf16_t* __restrict__ h0,
                                                        const bf16_t* __restrict__ h1, bf16_t* __restrict__ dh0,
                                                        bf16_t* __restrict__ dh1, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float g[8], a[8], b[8], o0[8], o1[8];
    Vec8<bf16_t>::load(dg + i * 8, g);
    Vec8<bf16_t>::load(h0 + i * 8, a);
    Vec8<bf16_t>::load(h1 + i * 8, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float cdf, pdf;
      gelu_parts(a[j], cdf, pdf);
      o0[j] = g[j] * b[j] * (cdf + a[j] * pdf);
      o1[j] = g[j] * a[j] * cdf;
    }
    Vec8<bf16_t>::store(dh0 + i * 8, o0);
    Vec8<bf16_t>::store(dh1 + i * 8, o1);
  }
}

// ------------------------------------------------------------------------------------------------------------
// residual/layer-scale backward (transformer_layer.py:70-88):  out = resid + rs[m] * gamma[n] * y[m][n]
//   dy[m][n] = rs[m] * gamma[n] * dout[m][n]   (written to dbranch)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_rows_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ gamma,
                                                         const float* __restrict__ rowscale, int rps,
                                                         bf16_t* __restrict__ dbranch, int64_t M, int N) {
  const int n8 = N / 8;
  const int64_t total = M * n8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / n8;
    const int c = (int)(i - m * n8) * 8;
    float d[8], gv[8];
    Vec8<bf16_t>::load(dout + m * N + c, d);
    const float s = rowscale ? rowscale[m / rps] : 1.f;
    if (gamma) {
      Vec8<bf16_t>::load(gamma + c, gv);
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] *= s * gv[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] *= s;
    }
    Vec8<bf16_t>::store(dbranch + m * N + c, d);
  }
}

// ------------------------------------------------------------------------------------------------------------
// F.normalize(x, dim=1) (one_peace_retrieval.py:112): y = x / max(||x||, eps); one wave per row
// ------------------------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const bf16_t* __restrict__ x, TO* __restrict__ y,
                                                         float* __restrict__ inv_norm, int rows, int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float ss = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    float v[8];
    Vec8<bf16_t>::load(x + (int64_t)row * cols + c, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
  }
  ss = wave_sum(ss);
  const float inv = 1.f / fmaxf(sqrtf(ss), eps);
  for (int c = lane * 8; c < cols; c += 512) {
    float v[8];
    Vec8<bf16_t>::load(x + (int64_t)row * cols + c, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= inv;
    Vec8<TO>::store(y + (int64_t)row * cols + c, v);
  }
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
}

// dx = inv * (dy - y * <y, dy>)   (valid while ||x|| > eps, which always holds for projected CLS features)
template <typename TY>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const TY* __restrict__ dy, const TY* __restrict__ y,
                                                         const float* __restrict__ inv_norm, bf16_t* __restrict__ dx,
                                                         int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float dot = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    float a[8], b[8];
    Vec8<TY>::load(dy + (int64_t)row * cols + c, a);
    Vec8<TY>::load(y + (int64_t)row * cols + c, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) dot += a[j] * b[j];
  }
  dot = wave_sum(dot);
  const float inv = inv_norm[row];
  for (int c = lane * 8; c < cols; c += 512) {
    float a[8], b[8];
    Vec8<TY>::load(dy + (int64_t)row * cols + c, a);
    Vec8<TY>::load(y + (int64_t)row * cols + c, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = inv * (a[j] - b[j] * dot);
    Vec8<bf16_t>::store(dx + (int64_t)row * cols + c, a);
  }
}

// ------------------------------------------------------------------------------------------------------------
// InfoNCE rows (image_text_pretrain_loss.py:164-185 + adjust_label_smoothed_nll_loss :17-27).
// One workgroup per row of sim [rows][n] (fp32).  Writes the row loss, the argmax hit, <dsim, sim> (for the
// logit-scale gradient) and overwrites sim with d(loss_row)/d(sim) * gscale.
//   loss_row = -(1-eps-e)*lp[t] - e*sum_j lp[j],  e = eps/(n-1),  lp = sim - lse
//   d/ds_k   = p_k*((1-eps-e) + n*e) - (1-eps-e)*[k==t] - e
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void infonce_rows_kernel(float* __restrict__ sim, int n, int64_t ld, int target0,
                                                           float eps_ls, float gscale, float* __restrict__ row_loss,
                                                           float* __restrict__ row_hit, float* __restrict__ row_dot,
                                                           int write_grad) {
  __shared__ float red[4];
  __shared__ float redv[4];
  __shared__ int redi[4];
  const int row = blockIdx.x;
  float* s = sim + (int64_t)row * ld;
  const int tgt = target0 + row;
  float mx = -INFINITY;
  int amax = 0x7fffffff;
  float sum = 0.f;
  for (int k = threadIdx.x; k < n; k += 256) {
    const float v = s[k];
    sum += v;
    if (v > mx) { mx = v; amax = k; }  // first maximum within this thread's increasing k
  }
  // block arg-max (ties -> smallest index, as torch.argmax on CPU returns the first maximal element)
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o);
    const int oi = __shfl_xor(amax, o);
    if (om > mx || (om == mx && oi < amax)) { mx = om; amax = oi; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { redv[threadIdx.x >> 6] = mx; redi[threadIdx.x >> 6] = amax; }
  __syncthreads();
  mx = redv[0]; amax = redi[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (redv[w] > mx || (redv[w] == mx && redi[w] < amax)) { mx = redv[w]; amax = redi[w]; }
  const float ssum = block_sum_256(sum, red);
  float ex = 0.f;
  for (int k = threadIdx.x; k < n; k += 256) ex += __expf(s[k] - mx);
  const float lse = mx + logf(block_sum_256(ex, red));
  const float e = (eps_ls != 0.f) ? eps_ls / (float)(n - 1) : 0.f;
  const float wt = 1.f - eps_ls - e;
  const float st = s[tgt];
  const float loss = -wt * (st - lse) - e * (ssum - (float)n * lse);
  float dot = 0.f;
  if (write_grad) {
    const float pk_coef = wt + (float)n * e;
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 256) {
      const float v = s[k];
      float d = __expf(v - lse) * pk_coef - e - (k == tgt ? wt : 0.f);
      d *= gscale;
      dot += d * v;
      s[k] = d;
    }
    dot = block_sum_256(dot, red);
  }
  if (threadIdx.x == 0) {
    row_loss[row] = loss;
    row_hit[row] = (amax == tgt) ? 1.f : 0.f;
    row_dot[row] = dot;
  }
}

// ------------------------------------------------------------------------------------------------------------
// AdamW as the reference does it (one_peace/optim/adam.py:186-253): fp32 math on bf16 params, decoupled
// decay applied to the parameter before the Adam update, eps added to sqrt(v).
// Algorithmic bytes: 22 B/param (p r+w 4, g r 2, m and v r+w 16).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(bf16_t* __restrict__ p, const bf16_t* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n8,
                                                    float beta1, float beta2, float eps, float step_size,
                                                    float decay_mul, float grad_scale, const float* __restrict__ sqnorm,
                                                    float clip_norm) {
  if (sqnorm != nullptr && clip_norm > 0.f) {  // fairseq/utils.py:393-397: clip_coef = clamp(max_norm / (norm + 1e-6), max=1)
    const float norm = fabsf(grad_scale) * sqrtf(*sqnorm);
    grad_scale *= fminf(1.0f, clip_norm / (norm + 1e-6f));
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float pv[8], gv[8], mv[8], vv[8];
    Vec8<bf16_t>::load(p + i * 8, pv);
    Vec8<bf16_t>::load(g + i * 8, gv);
    Vec8<float>::load(m + i * 8, mv);
    Vec8<float>::load(v + i * 8, vv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gr = gv[j] * grad_scale;
      mv[j] = mv[j] * beta1 + (1.f - beta1) * gr;
      vv[j] = vv[j] * beta2 + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(vv[j]) + eps;
      pv[j] = pv[j] * decay_mul - step_size * (mv[j] / denom);
    }
    Vec8<bf16_t>::store(p + i * 8, pv);
    Vec8<float>::store(m + i * 8, mv);
    Vec8<float>::store(v + i * 8, vv);
  }
}

// The same update over a flat buffer that is partitioned into parameter groups (trainer.py:265-278, utils/layer_decay.py:34-77:
// "layer_<id>_<decay|no_decay>" groups with their own lr_scale and weight decay; optim/base_optimizer.py:8-14 sets
// lr_g = lr * lr_scale_g): group g owns the 8-element vectors [end8[g-1], end8[g]).  ONE launch for all groups: the group of a
// vector is found by binary search in a table staged in LDS (<= 256 groups), and re-used while consecutive vectors stay in it.
constexpr int ADAM_MAX_GROUPS = 256;
__global__ __launch_bounds__(256) void adamw_groups_kernel(bf16_t* __restrict__ p, const bf16_t* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, int64_t n8,
                                                           const int64_t* __restrict__ end8, const float* __restrict__ lr_scale,
                                                           const float* __restrict__ wd, int n_groups, float lr, float beta1,
                                                           float beta2, float eps, float bias_corr, float grad_scale,
                                                           const float* __restrict__ sqnorm, float clip_norm) {
  __shared__ int64_t s_end[ADAM_MAX_GROUPS];
  __shared__ float s_step[ADAM_MAX_GROUPS], s_decay[ADAM_MAX_GROUPS];
  for (int i = threadIdx.x; i < n_groups; i += 256) {
    const float lr_g = lr * lr_scale[i];
    s_end[i] = end8[i];
    s_step[i] = lr_g * bias_corr;          // lr_g * sqrt(1 - beta2^t) / (1 - beta1^t)
    s_decay[i] = 1.f - wd[i] * lr_g;       // p <- p - wd * lr_g * p, before the Adam update (adam.py:243-246)
  }
  __syncthreads();
  if (sqnorm != nullptr && clip_norm > 0.f) {
    const float norm = fabsf(grad_scale) * sqrtf(*sqnorm);
    grad_scale *= fminf(1.0f, clip_norm / (norm + 1e-6f));
  }
  int grp = 0;
  int64_t lo = 0, hi = s_end[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    if (i < lo || i >= hi) {  // first group whose end is > i
      int a = 0, b = n_groups - 1;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (s_end[mid] > i) b = mid; else a = mid + 1;
      }
      grp = a;
      lo = grp ? s_end[grp - 1] : 0;
      hi = s_end[grp];
    }
    const float step_size = s_step[grp], decay_mul = s_decay[grp];
    float pv[8], gv[8], mv[8], vv[8];
    Vec8<bf16_t>::load(p + i * 8, pv);
    Vec8<bf16_t>::load(g + i * 8, gv);
    Vec8<float>::load(m + i * 8, mv);
    Vec8<float>::load(v + i * 8, vv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gr = gv[j] * grad_scale;
      mv[j] = mv[j] * beta1 + (1.f - beta1) * gr;
      vv[j] = vv[j] * beta2 + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(vv[j]) + eps;
      pv[j] = pv[j] * decay_mul - step_size * (mv[j] / denom);
    }
    Vec8<bf16_t>::store(p + i * 8, pv);
    Vec8<float>::store(m + i * 8, mv);
    Vec8<float>::store(v + i * 8, vv);
  }
}

// sum of squares, stage 1: one partial per workgroup (grid-stride over 8-element vectors); stage 2 folds the partials
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const bf16_t* __restrict__ x, int64_t n8, float* __restrict__ part) {
  __shared__ float red[4];
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float v[8];
    Vec8<bf16_t>::load(x + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) a += v[j] * v[j];
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += part[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------------------------
// Relative-position bias (adapter/image.py:164-171, text.py:76-83, audio.py:117-124):
// bias[h][i][j] = table[bucket[i][j]][h].  The reference expands this to [B, heads, S, S] per forward; here it
// is built ONCE per table as [heads][S][Spad] bf16 and shared by every sample (attention reads it from L2).
// Columns j >= S are zero-filled (the attention kernels mask them by index).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relpos_build_kernel(const bf16_t* __restrict__ table, const int* __restrict__ bucket,
                                                           int64_t bucket_ld, bf16_t* __restrict__ out, int heads, int S,
                                                           int Spad, int transposed) {
  const int64_t total = (int64_t)S * Spad;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / Spad), j = (int)(idx - (int64_t)i * Spad);
    if (j < S) {
      const int b = transposed ? bucket[(int64_t)j * bucket_ld + i] : bucket[(int64_t)i * bucket_ld + j];
      for (int h = 0; h < heads; ++h) out[((int64_t)h * S + i) * Spad + j] = table[(int64_t)b * heads + h];
    } else {
      for (int h = 0; h < heads; ++h) out[((int64_t)h * S + i) * Spad + j] = (bf16_t)0.f;
    }
  }
}

// dtable[bucket[i][j]][h] += dbias[h][i][j]  (fp32 atomics: <= heads*S*S adds onto num_rel*heads addresses once
// per backward; summation order is not fixed, error is fp32 round-off of a few-thousand-term sum)
__global__ __launch_bounds__(256) void relpos_bwd_kernel(const float* __restrict__ dbias, const int* __restrict__ bucket,
                                                         int64_t bucket_ld, float* __restrict__ dtable, int heads, int S,
                                                         int Spad) {
  const int64_t total = (int64_t)S * S;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / S), j = (int)(idx - (int64_t)i * S);
    const int b = bucket[(int64_t)i * bucket_ld + j];
    for (int h = 0; h < heads; ++h) atomicAdd(&dtable[(int64_t)b * heads + h], dbias[((int64_t)h * S + i) * Spad + j]);
  }
}

// Per-sample bias images of the masked-pretraining passes (adapter/image.py:188-204,229-246: the adapters gather a different
// token subset per sample, and with it rows AND columns of the dense [B, heads, S, S] bias): here straight from the table,
//   out[b][h][i][j] = table[bucket[ids[b][i]][ids[b][j]]][h]   (ids = position ids of the K kept tokens of sample b)
// so neither the dense bias nor its gathers are ever materialised.  transposed: out[b][h][j][i] holds that value.
__global__ __launch_bounds__(256) void relpos_build_ids_kernel(const bf16_t* __restrict__ table, const int* __restrict__ bucket,
                                                               int64_t bucket_ld, const int* __restrict__ ids, bf16_t* __restrict__ out,
                                                               int B, int heads, int K, int Kpad, int transposed) {
  const int64_t total = (int64_t)B * K * Kpad;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int j = (int)(idx % Kpad);
    const int64_t bi = idx / Kpad;
    const int i = (int)(bi % K), b = (int)(bi / K);
    bf16_t* o = out + (((int64_t)b * heads) * K + i) * Kpad + j;
    if (j < K) {
      const int pi = ids[(int64_t)b * K + i], pj = ids[(int64_t)b * K + j];
      const int bk = transposed ? bucket[(int64_t)pj * bucket_ld + pi] : bucket[(int64_t)pi * bucket_ld + pj];
      for (int h = 0; h < heads; ++h) o[(int64_t)h * K * Kpad] = table[(int64_t)bk * heads + h];
    } else {
      for (int h = 0; h < heads; ++h) o[(int64_t)h * K * Kpad] = (bf16_t)0.f;
    }
  }
}

// Stage 1 of the table gradient of per-sample images: dense[h][ids[b][i]][ids[b][j]] += dbias[b][h][i][j] -- the per-sample
// slabs are folded into ONE full-sequence image (heads x Sfull x Sfull addresses: little contention), which the ordinary
// relpos_bwd_kernel then scatters onto the table.  (Adding the slabs straight onto the table was 29 ms per call: ten million
// fp32 atomics on a few thousand addresses.)
__global__ __launch_bounds__(256) void relpos_fold_ids_kernel(const float* __restrict__ dbias, const int* __restrict__ ids,
                                                              float* __restrict__ dense, int B, int heads, int K, int Kpad, int Sfull) {
  const int64_t total = (int64_t)B * heads * K * K;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int j = (int)(idx % K);
    int64_t r = idx / K;
    const int i = (int)(r % K);
    r /= K;
    const int h = (int)(r % heads), b = (int)(r / heads);
    const float v = dbias[((((int64_t)b * heads) + h) * K + i) * Kpad + j];
    atomicAdd(&dense[((int64_t)h * Sfull + ids[(int64_t)b * K + i]) * Sfull + ids[(int64_t)b * K + j]], v);
  }
}

inline int ew_grid(int64_t work_items) {
  int64_t b = (work_items + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------------------------------------------------
// Stochastic depth without the multiplications by zero (transformer_layer.py:78-88: a dropped sample's branch output is
// multiplied by 0 before it is added to the residual).  The rows of the samples a residual branch KEEPS are packed into a
// smaller matrix, the branch runs on that, and its result is merged back into the full activation matrix:
//   gather:  dst[seg.dst_row0 + j*S + t] = src[seg.src_row0 + kept[j]*S + t]   j < n_kept;  the rest of the segment's dst rows = 0
//            (each segment is padded to a multiple of 64 rows: the weight-gradient kernels want K % 64 == 0, and 0-rows add 0)
//   merge:   out[r] = upd[seg.dst_row0 + inv[sample]*S + t]  if the sample of row r is kept (inv >= 0), else base[r]
// One wave per row, 16-byte vectors; `list` holds the int32 kept[] lists (gather) or inv[] lists (merge) of the segments.
// ------------------------------------------------------------------------------------------------------------
struct RowsSeg { int64_t src_row0, dst_row0; int S, n_kept, dst_rows, n_samples, list_off, pad_; };
struct RowsSegs { int nseg, pad_; RowsSeg seg[4]; };

__global__ __launch_bounds__(256) void rows_gather_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                          const int* __restrict__ list, const RowsSegs d, int64_t dst_total, int cols) {
  const int lane = threadIdx.x & 63;
  const int n8 = cols / 8;
  for (int64_t rc = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); rc < dst_total; rc += (int64_t)gridDim.x * 4) {
    int64_t from = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < d.nseg && rc >= d.seg[i].dst_row0 && rc < d.seg[i].dst_row0 + d.seg[i].dst_rows) {
        const int local = (int)(rc - d.seg[i].dst_row0);
        const int j = local / d.seg[i].S;
        if (j < d.seg[i].n_kept) from = d.seg[i].src_row0 + (int64_t)list[d.seg[i].list_off + j] * d.seg[i].S + (local - j * d.seg[i].S);
      }
    }
    bf16x8* o = reinterpret_cast<bf16x8*>(dst + rc * cols);
    if (from >= 0) {
      const bf16x8* in = reinterpret_cast<const bf16x8*>(src + from * cols);
      for (int c = lane; c < n8; c += 64) o[c] = in[c];
    } else {
      bf16x8 z;
#pragma unroll
      for (int k = 0; k < 8; ++k) z[k] = (bf16_t)0.f;
      for (int c = lane; c < n8; c += 64) o[c] = z;
    }
  }
}

__global__ __launch_bounds__(256) void rows_merge_kernel(const bf16_t* __restrict__ base, const bf16_t* __restrict__ upd,
                                                         bf16_t* __restrict__ out, const int* __restrict__ list, const RowsSegs d,
                                                         int64_t total, int cols) {
  const int lane = threadIdx.x & 63;
  const int n8 = cols / 8;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < total; r += (int64_t)gridDim.x * 4) {
    int64_t from = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < d.nseg && r >= d.seg[i].src_row0 && r < d.seg[i].src_row0 + (int64_t)d.seg[i].n_samples * d.seg[i].S) {
        const int local = (int)(r - d.seg[i].src_row0);
        const int smp = local / d.seg[i].S;
        const int j = list[d.seg[i].list_off + smp];
        if (j >= 0) from = d.seg[i].dst_row0 + (int64_t)j * d.seg[i].S + (local - smp * d.seg[i].S);
      }
    }
    if (from < 0 && out == base) continue;  // in place: rows of dropped samples stay what they are
    if (from >= 0 && upd == nullptr) continue;  // (ABI 9) no packed matrix: only the rows of dropped samples are copied
    const bf16x8* in = reinterpret_cast<const bf16x8*>(from >= 0 ? upd + from * cols : base + r * cols);
    bf16x8* o = reinterpret_cast<bf16x8*>(out + r * cols);
    for (int c = lane; c < n8; c += 64) o[c] = in[c];
  }
}

// packed row -> row of the full matrix (-1: a surplus row of a rounded-up segment): what op_rows_gather reads, as a table for the kernels
// that read / write THROUGH it (ABI 9: op_layernorm_*, op_resid_bwd, the residual epilogue of op_gemm_nt)
__global__ __launch_bounds__(256) void rows_map_kernel(int* __restrict__ map, const int* __restrict__ list, const RowsSegs d, int64_t dst_total) {
  const int64_t rc = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (rc >= dst_total) return;
  int64_t from = -1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < d.nseg && rc >= d.seg[i].dst_row0 && rc < d.seg[i].dst_row0 + d.seg[i].dst_rows) {
      const int local = (int)(rc - d.seg[i].dst_row0);
      const int j = local / d.seg[i].S;
      if (j < d.seg[i].n_kept) from = d.seg[i].src_row0 + (int64_t)list[d.seg[i].list_off + j] * d.seg[i].S + (local - j * d.seg[i].S);
    }
  }
  map[rc] = (int)from;
}

}  // namespace

extern "C" {

int op_transpose_scaled(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, const void* scale,
                        void* stream) {
  OP_CHECK_ARG(in && out && rows >= 0 && cols >= 0, "transpose: bad args");
  if (rows == 0 || cols == 0) return OP_OK;
  dim3 grid(ceil_div(cols, 64), ceil_div(rows, 64));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, (int)rows,
                     (int)cols, ld_in, ld_out, (const bf16_t*)scale);
  OP_LAUNCH_CHECK();
  return OP_OK;
}
int op_transpose(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, void* stream) {
  return op_transpose_scaled(in, out, rows, cols, ld_in, ld_out, nullptr, stream);
}

// Batched transposes: `table` is a DEVICE array of n descriptors {in, out, rows, cols, ld_in, ld_out, tile0, tiles_x}
// (struct TransposeDesc, 56 bytes: two pointers, two int32, two int64, two int32, one pointer; tile0 = running sum of
// ceil(cols/64)*ceil(rows/64), tiles_x = ceil(cols/64), scale = nullable bf16 [rows] row scales of `in`); total_tiles = the sum over all descriptors.
int op_transpose_batched(const void* table, int64_t n, int64_t total_tiles, void* stream) {
  OP_CHECK_ARG(table && n > 0 && total_tiles > 0, "transpose_batched: bad args");
  hipLaunchKernelGGL(transpose_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const TransposeDesc*)table, (int)n);
  OP_LAUNCH_CHECK();
  return OP_OK;
}
int64_t op_transpose_desc_bytes(void) { return (int64_t)sizeof(TransposeDesc); }

int64_t op_colsum_workspace_bytes(int64_t N) { return (int64_t)CS_MAX_PARTS * N * (int64_t)sizeof(float); }

// out[n] = (accumulate ? out[n] : 0) + mul[n] * sum_m rowscale[m/rps] * x[m][n] * (y ? y[m][n] : 1)
// out_dtype: 0 bf16, 1 f32.  y, rowscale, mul nullable.
int op_colsum(const void* x, const void* y, const float* rowscale, int64_t rows_per_sample, const void* mul, void* out,
              void* workspace, int64_t M, int64_t N, int accumulate, int out_dtype, void* stream) {
  OP_CHECK_ARG(x && out && workspace, "colsum: null pointer");
  OP_CHECK_ARG(N > 0 && N % 8 == 0, "colsum: N must be a multiple of 8");
  hipStream_t s = (hipStream_t)stream;
  int parts = (int)((M + 63) / 64);
  if (parts > CS_MAX_PARTS) parts = CS_MAX_PARTS;
  if (parts < 1) parts = 1;
  dim3 grid(ceil_div(N, 512), parts);
  hipLaunchKernelGGL(colsum_partial_kernel, grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)y, rowscale,
                     (int)(rows_per_sample > 0 ? rows_per_sample : 1), (float*)workspace, M, (int)N, N);
  OP_LAUNCH_CHECK();
  if (out_dtype == OP_DT_BF16)
    hipLaunchKernelGGL((partials_reduce_kernel<bf16_t>), dim3(ceil_div(N, 32)), dim3(256), 0, s, (const float*)workspace, parts,
                       N, (int)N, (const bf16_t*)mul, (bf16_t*)out, accumulate);
  else
    hipLaunchKernelGGL((partials_reduce_kernel<float>), dim3(ceil_div(N, 32)), dim3(256), 0, s, (const float*)workspace, parts,
                       N, (int)N, (const bf16_t*)mul, (float*)out, accumulate);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

int64_t op_resid_bwd_workspace_bytes(int64_t N) { return 2 * (int64_t)CS_MAX_PARTS * N * (int64_t)sizeof(float); }

// One pass over the gradient of  out = resid + rowscale[m/rps] * gamma[n] * y[m][n]  (transformer_layer.py:70-88,
// y = the branch's last Linear output incl. its bias):
//   dbranch = rowscale * gamma * dout;  dgamma (+)= sum_m rowscale * dout * y;  dbias (+)= sum_m dbranch
// gamma, rowscale, y/dgamma, dbias nullable; dgamma/dbias are bf16 [N]; accumulate: add into them instead of overwriting.
int op_resid_bwd(const void* dout, const void* y, const void* gamma, const float* rowscale, int64_t rows_per_sample,
                 void* dbranch, void* dgamma, void* dbias, float* g0, void* workspace, int64_t M, int64_t N, int accumulate,
                 const int* dout_rows, void* stream) {
  OP_CHECK_ARG(dout && dbranch, "resid_bwd: null pointer");
  OP_CHECK_ARG(N > 0 && N % 8 == 0, "resid_bwd: N must be a multiple of 8");
  OP_CHECK_ARG(!dgamma || y, "resid_bwd: dgamma needs y");
  OP_CHECK_ARG(!(dgamma && g0), "resid_bwd: g0 (dbranch without gamma) and dgamma (from y) are alternatives");
  OP_CHECK_ARG(!(dgamma || dbias || g0) || workspace, "resid_bwd: dgamma/dbias/g0 requested without workspace");
  if (M == 0) {
    if (g0) (void)hipMemsetAsync(g0, 0, (size_t)N * sizeof(float), (hipStream_t)stream);
    return OP_OK;
  }
  hipStream_t s = (hipStream_t)stream;
  int parts = (int)((M + 63) / 64);
  if (parts > CS_MAX_PARTS) parts = CS_MAX_PARTS;
  if (parts < 1) parts = 1;
  float* ws = (dgamma || dbias || g0) ? (float*)workspace : nullptr;
  // (ABI 8) g0 wanted = the layer-scale gradient comes from the weight gradient: dbranch stays WITHOUT gamma (the weight-gradient and
  // input-gradient GEMMs carry it: op_gemm_tn_grouped's rscale, op_transpose_scaled), dbias still gets it in the fold
  hipLaunchKernelGGL(resid_bwd_kernel, dim3(ceil_div(N, 512), parts), dim3(256), 0, s, (const bf16_t*)dout,
                     (const bf16_t*)(dgamma ? y : nullptr), (const bf16_t*)(g0 ? nullptr : gamma), rowscale,
                     (int)(rows_per_sample > 0 ? rows_per_sample : 1), (bf16_t*)dbranch, ws, M, (int)N, dout_rows);
  OP_LAUNCH_CHECK();
  if (ws) {
    // job 0: dgamma from sum rs*dout*y; job 1: dbias = gamma * sum rs*dout; job 2 (g0): the same partials without gamma, fp32
    hipLaunchKernelGGL((partials_reduce3_kernel<bf16_t>), dim3(ceil_div(N, 32), g0 ? 3 : 2), dim3(256), 0, s, ws,
                       ws + (int64_t)parts * N, ws + (int64_t)parts * N, (const bf16_t*)nullptr, (const bf16_t*)gamma,
                       (const bf16_t*)nullptr, (bf16_t*)dgamma, (bf16_t*)dbias, (bf16_t*)nullptr, parts, N, (int)N,
                       accumulate, g0);
    OP_LAUNCH_CHECK();
  }
  return OP_OK;
}

// dgamma[n] (+)= sum_s rowdot[s][n] + sum_i b_i[n] * g0_i[n]  (see include/onepeace_hip.h)
int op_gamma_grad_finish(const float* rowdot, int64_t slots, const void* b0, const float* g00, const void* b1, const float* g01,
                         const void* b2, const float* g02, void* dgamma, int64_t N, int accumulate, void* stream) {
  OP_CHECK_ARG(rowdot && dgamma && N > 0 && slots > 0, "gamma_grad_finish: null pointer");
  hipLaunchKernelGGL(gamma_grad_finish_kernel, dim3(ceil_div(N, 64)), dim3(256), 0, (hipStream_t)stream, rowdot, (int)slots,
                     (const bf16_t*)b0, g00, (const bf16_t*)b1, g01, (const bf16_t*)b2, g02, (bf16_t*)dgamma, (int)N, accumulate);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// Column sums of x [M, n_seg * seg_cols] (row stride = n_seg * seg_cols) delivered per segment: out_i (bf16 [seg_cols],
// nullable = skip) (+)= sum_m x[m][i * seg_cols + n].  The q/k/v bias gradients of the fused projection
// (multihead_attention.py:57-62: k_proj has no bias -> its slot is null).  n_seg <= 3.
int op_colsum_segments(const void* x, void* out0, void* out1, void* out2, void* workspace, int64_t M, int64_t n_seg,
                       int64_t seg_cols, int accumulate, void* stream) {
  OP_CHECK_ARG(x && workspace, "colsum_segments: null pointer");
  OP_CHECK_ARG(n_seg >= 1 && n_seg <= 3 && seg_cols > 0 && seg_cols % 8 == 0, "colsum_segments: bad segments");
  hipStream_t s = (hipStream_t)stream;
  const int64_t N = n_seg * seg_cols;
  int parts = (int)((M + 63) / 64);
  if (parts > CS_MAX_PARTS) parts = CS_MAX_PARTS;
  if (parts < 1) parts = 1;
  // only the segments that have a consumer are read (k_proj has no bias: a third of the q|k|v gradient stays untouched)
  void* outs[3] = {out0, out1, out2};
  const float* ws = (const float*)workspace;
  for (int i = 0; i < (int)n_seg; ++i) {
    if (outs[i] == nullptr) continue;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(ceil_div(seg_cols, 512), parts), dim3(256), 0, s,
                       (const bf16_t*)x + i * seg_cols, (const bf16_t*)nullptr, (const float*)nullptr, 1,
                       (float*)workspace + (int64_t)i * parts * seg_cols, M, (int)seg_cols, N);
    OP_LAUNCH_CHECK();
  }
  const int64_t st = (int64_t)parts * seg_cols;
  hipLaunchKernelGGL((partials_reduce3_kernel<bf16_t>), dim3(ceil_div(seg_cols, 32), (int)n_seg), dim3(256), 0, s, ws, ws + st,
                     ws + 2 * st, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (bf16_t*)out0,
                     (bf16_t*)out1, (bf16_t*)out2, parts, seg_cols, (int)seg_cols, accumulate);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

int op_geglu_bwd(const void* dg, const void* h0, const void* h1, void* dh0, void* dh1, int64_t numel, void* stream) {
  OP_CHECK_ARG(dg && h0 && h1 && dh0 && dh1, "geglu_bwd: null pointer");
  OP_CHECK_ARG(numel % 8 == 0, "geglu_bwd: numel must be a multiple of 8");
  if (numel == 0) return OP_OK;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(ew_grid(numel / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dg,
                     (const bf16_t*)h0, (const bf16_t*)h1, (bf16_t*)dh0, (bf16_t*)dh1, numel / 8);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// dbranch[m][n] = rowscale[m/rps] * gamma[n] * dout[m][n]   (gamma, rowscale nullable)
int op_scale_rows(const void* dout, const void* gamma, const float* rowscale, int64_t rows_per_sample, void* dbranch,
                  int64_t M, int64_t N, void* stream) {
  OP_CHECK_ARG(dout && dbranch, "scale_rows: null pointer");
  OP_CHECK_ARG(N % 8 == 0, "scale_rows: N must be a multiple of 8");
  if (M == 0) return OP_OK;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(ew_grid(M * (N / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout,
                     (const bf16_t*)gamma, rowscale, (int)(rows_per_sample > 0 ? rows_per_sample : 1), (bf16_t*)dbranch, M,
                     (int)N);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// y (out_dtype 0 bf16 / 1 f32) = x / max(||x||_2, eps) per row; inv_norm[rows] saved for the backward
int op_l2norm_fwd(const void* x, void* y, float* inv_norm, int64_t rows, int64_t cols, float eps, int out_dtype, void* stream) {
  OP_CHECK_ARG(x && y && cols % 8 == 0, "l2norm_fwd: bad args");
  if (rows == 0) return OP_OK;
  if (out_dtype == OP_DT_BF16)
    hipLaunchKernelGGL((l2norm_fwd_kernel<bf16_t>), dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)y, inv_norm, (int)rows, (int)cols, eps);
  else
    hipLaunchKernelGGL((l2norm_fwd_kernel<float>), dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (float*)y, inv_norm, (int)rows, (int)cols, eps);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

int op_l2norm_bwd(const void* dy, const void* y, const float* inv_norm, void* dx, int64_t rows, int64_t cols, int y_dtype,
                  void* stream) {
  OP_CHECK_ARG(dy && y && inv_norm && dx && cols % 8 == 0, "l2norm_bwd: bad args");
  if (rows == 0) return OP_OK;
  if (y_dtype == OP_DT_BF16)
    hipLaunchKernelGGL((l2norm_bwd_kernel<bf16_t>), dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dy, (const bf16_t*)y, inv_norm, (bf16_t*)dx, (int)rows, (int)cols);
  else
    hipLaunchKernelGGL((l2norm_bwd_kernel<float>), dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)dy, (const float*)y, inv_norm, (bf16_t*)dx, (int)rows, (int)cols);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// sim [rows][n] fp32 (ld) is overwritten by gscale * d loss_row / d sim when write_grad != 0.
int op_infonce_rows(float* sim, int64_t rows, int64_t n, int64_t ld, int64_t target0, float label_smoothing, float gscale,
                    float* row_loss, float* row_hit, float* row_dot, int write_grad, void* stream) {
  OP_CHECK_ARG(sim && row_loss && row_hit && row_dot, "infonce_rows: null pointer");
  OP_CHECK_ARG(n >= 2 && target0 >= 0 && target0 + rows <= n, "infonce_rows: targets out of range");
  if (rows == 0) return OP_OK;
  hipLaunchKernelGGL(infonce_rows_kernel, dim3((int)rows), dim3(256), 0, (hipStream_t)stream, sim, (int)n, ld, (int)target0,
                     label_smoothing, gscale, row_loss, row_hit, row_dot, write_grad);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// out[0] = sum of squares of a bf16 vector (fp32, deterministic two-stage fold): the global gradient norm of
// fairseq/utils.py:349-391 over the flat gradient buffer.  workspace: 1024 floats.  numel % 8 == 0.
int op_sqnorm(const void* x, int64_t numel, float* workspace, float* out, void* stream) {
  OP_CHECK_ARG(x && workspace && out && numel % 8 == 0, "sqnorm: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(1024), dim3(256), 0, s, (const bf16_t*)x, numel / 8, workspace);
  OP_LAUNCH_CHECK();
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, 1024, out);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// One AdamW update over a flat bf16 parameter range (numel % 8 == 0).  step >= 1.  The gradient is multiplied by
// grad_scale (1/world after a SUM all-reduce, trainer.py:917-923) and, when grad_sqnorm (device scalar = sum of squares of
// the UNSCALED gradients of ALL ranges, op_sqnorm) and clip_norm > 0 are given, by the reference's clip coefficient
// (trainer.py:929, fairseq/utils.py:393-397) -- computed on the device, no host synchronisation.
int op_adamw_step(void* p, const void* g, float* m, float* v, int64_t numel, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int64_t step, float grad_scale, const float* grad_sqnorm, float clip_norm,
                  void* stream) {
  OP_CHECK_ARG(p && g && m && v, "adamw: null pointer");
  OP_CHECK_ARG(numel % 8 == 0 && step >= 1, "adamw: numel must be a multiple of 8 and step >= 1");
  if (numel == 0) return OP_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
  const float decay_mul = 1.f - weight_decay * lr;
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(numel / 8)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)p, (const bf16_t*)g,
                     m, v, numel / 8, beta1, beta2, eps, step_size, decay_mul, grad_scale, grad_sqnorm, clip_norm);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// AdamW over a flat buffer partitioned into n_groups (<= 256) contiguous parameter groups in ONE launch: group g covers
// elements [8 * group_end8[g-1], 8 * group_end8[g]) (device table, ascending, last entry = numel / 8) and uses
// lr * group_lr_scale[g] and group_weight_decay[g] (device tables; they are static, the scheduled lr is the scalar argument).
// Same update rule, clip handling and byte traffic as op_adamw_step.
int op_adamw_step_groups(void* p, const void* g, float* m, float* v, int64_t numel, const int64_t* group_end8,
                         const float* group_lr_scale, const float* group_weight_decay, int64_t n_groups, float lr, float beta1,
                         float beta2, float eps, int64_t step, float grad_scale, const float* grad_sqnorm, float clip_norm,
                         void* stream) {
  OP_CHECK_ARG(p && g && m && v && group_end8 && group_lr_scale && group_weight_decay, "adamw_groups: null pointer");
  OP_CHECK_ARG(numel % 8 == 0 && step >= 1 && n_groups >= 1 && n_groups <= ADAM_MAX_GROUPS,
               "adamw_groups: numel %% 8 == 0, step >= 1, 1 <= n_groups <= %d required", ADAM_MAX_GROUPS);
  if (numel == 0) return OP_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_groups_kernel, dim3(ew_grid(numel / 8)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)p,
                     (const bf16_t*)g, m, v, numel / 8, group_end8, group_lr_scale, group_weight_decay, (int)n_groups, lr, beta1,
                     beta2, eps, (float)(sqrt(bc2) / bc1), grad_scale, grad_sqnorm, clip_norm);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// transposed != 0 writes out[h][key][query] (the image the dK/dV kernel reads) instead of out[h][query][key]
int op_relpos_bias_build(const void* table, const int* bucket, int64_t bucket_ld, void* out, int64_t heads, int64_t S,
                         int64_t Spad, int transposed, void* stream) {
  OP_CHECK_ARG(table && bucket && out && Spad >= S && Spad % 8 == 0, "relpos_bias_build: bad args");
  hipLaunchKernelGGL(relpos_build_kernel, dim3(ew_grid(S * Spad)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)table,
                     bucket, bucket_ld, (bf16_t*)out, (int)heads, (int)S, (int)Spad, transposed);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// Per-sample images from position ids: out [B][heads][K][Kpad] bf16 (pad columns zero), ids [B][K] int32 (valid positions;
// the caller maps padding to a valid id and masks it through key_pad, as the reference does: adapter/image.py:241-246).
int op_relpos_bias_build_ids(const void* table, const int* bucket, int64_t bucket_ld, const int* ids, void* out, int64_t B,
                             int64_t heads, int64_t K, int64_t Kpad, int transposed, void* stream) {
  OP_CHECK_ARG(table && bucket && ids && out && B > 0 && K > 0 && Kpad >= K && Kpad % 8 == 0, "relpos_bias_build_ids: bad args");
  hipLaunchKernelGGL(relpos_build_ids_kernel, dim3(ew_grid(B * K * Kpad)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)table,
                     bucket, bucket_ld, ids, (bf16_t*)out, (int)B, (int)heads, (int)K, (int)Kpad, transposed);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// Table gradient from the per-sample bias gradients dbias [B][heads][K][Kpad] fp32 (one slab per sample from op_attn_bwd);
// dtable [num_rel][heads] fp32 is accumulated into (pre-zero it).  dense_ws: fp32 [heads][Sfull][Sfull] scratch, ZEROED by the
// caller, Sfull = extent of the bucket table the ids index (the slabs are first folded into it, then scattered onto the table).
int op_relpos_bias_bwd_ids(const float* dbias, const int* bucket, int64_t bucket_ld, const int* ids, float* dense_ws, int64_t Sfull,
                           float* dtable, int64_t B, int64_t heads, int64_t K, int64_t Kpad, void* stream) {
  OP_CHECK_ARG(dbias && bucket && ids && dense_ws && dtable && B > 0 && K > 0 && Sfull > 0, "relpos_bias_bwd_ids: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(relpos_fold_ids_kernel, dim3(ew_grid(B * heads * K * K)), dim3(256), 0, s, dbias, ids, dense_ws, (int)B,
                     (int)heads, (int)K, (int)Kpad, (int)Sfull);
  hipLaunchKernelGGL(relpos_bwd_kernel, dim3(ew_grid(Sfull * Sfull)), dim3(256), 0, s, (const float*)dense_ws, bucket, bucket_ld,
                     dtable, (int)heads, (int)Sfull, (int)Sfull);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

int op_relpos_bias_bwd(const float* dbias, const int* bucket, int64_t bucket_ld, float* dtable, int64_t heads, int64_t S,
                       int64_t Spad, void* stream) {
  OP_CHECK_ARG(dbias && bucket && dtable, "relpos_bias_bwd: null pointer");
  hipLaunchKernelGGL(relpos_bwd_kernel, dim3(ew_grid(S * S)), dim3(256), 0, (hipStream_t)stream, dbias, bucket, bucket_ld,
                     dtable, (int)heads, (int)S, (int)Spad);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// Row packing of the samples a stochastic-depth branch keeps (see rows_gather_kernel).  Per segment i < nseg (<= 4), host arrays:
// src_row0 (first row of the segment in the full matrix), dst_row0 (first row in the packed matrix), S (rows per sample),
// n_kept, dst_rows (n_kept * S rounded up by the caller; the surplus rows are written as zeros), n_samples, list_off (where the
// segment's list starts in `list`).  `list` (device int32): gather: the kept sample numbers in ascending order; merge: for every
// sample of the segment its position among the kept ones or -1.  cols % 8 == 0, rows 16-byte aligned.
static int rows_segs(RowsSegs& d, int64_t nseg, const int64_t* src_row0, const int64_t* dst_row0, const int64_t* S, const int64_t* n_kept,
                     const int64_t* dst_rows, const int64_t* n_samples, const int64_t* list_off) {
  if (nseg < 1 || nseg > 4 || !src_row0 || !dst_row0 || !S || !n_kept || !dst_rows || !n_samples || !list_off) return 0;
  memset(&d, 0, sizeof(d));
  d.nseg = (int)nseg;
  for (int i = 0; i < (int)nseg; ++i) {
    if (S[i] < 1 || n_kept[i] < 0 || n_kept[i] > n_samples[i] || dst_rows[i] < n_kept[i] * S[i]) return 0;
    d.seg[i].src_row0 = src_row0[i]; d.seg[i].dst_row0 = dst_row0[i]; d.seg[i].S = (int)S[i]; d.seg[i].n_kept = (int)n_kept[i];
    d.seg[i].dst_rows = (int)dst_rows[i]; d.seg[i].n_samples = (int)n_samples[i]; d.seg[i].list_off = (int)list_off[i];
  }
  return 1;
}

int op_rows_gather(const void* src, void* dst, const int* list, int64_t nseg, const int64_t* src_row0, const int64_t* dst_row0,
                   const int64_t* S, const int64_t* n_kept, const int64_t* dst_rows, const int64_t* n_samples, const int64_t* list_off,
                   int64_t dst_total, int64_t cols, void* stream) {
  RowsSegs d;
  OP_CHECK_ARG(src && dst && list && cols > 0 && cols % 8 == 0, "rows_gather: bad argument");
  OP_CHECK_ARG(rows_segs(d, nseg, src_row0, dst_row0, S, n_kept, dst_rows, n_samples, list_off), "rows_gather: bad segment table");
  if (dst_total == 0) return OP_OK;
  int64_t nb = (dst_total + 3) / 4;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(rows_gather_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, list, d,
                     dst_total, (int)cols);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// out may be `base` itself (then only the rows of kept samples are written).
int op_rows_merge(const void* base, const void* upd, void* out, const int* list, int64_t nseg, const int64_t* src_row0,
                  const int64_t* dst_row0, const int64_t* S, const int64_t* n_kept, const int64_t* dst_rows, const int64_t* n_samples,
                  const int64_t* list_off, int64_t total, int64_t cols, void* stream) {
  RowsSegs d;
  OP_CHECK_ARG(base && out && list && cols > 0 && cols % 8 == 0, "rows_merge: bad argument");
  OP_CHECK_ARG(upd || out != base, "rows_merge: without upd (copy of the dropped samples' rows) out must not be base");
  OP_CHECK_ARG(rows_segs(d, nseg, src_row0, dst_row0, S, n_kept, dst_rows, n_samples, list_off), "rows_merge: bad segment table");
  if (total == 0) return OP_OK;
  int64_t nb = (total + 3) / 4;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(rows_merge_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)base, (const bf16_t*)upd,
                     (bf16_t*)out, list, d, total, (int)cols);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// (ABI 9) map[r] = the row of the full matrix packed row r stands for (op_rows_gather's source row), -1 for the surplus rows.
int op_rows_map(int* map, const int* list, int64_t nseg, const int64_t* src_row0, const int64_t* dst_row0, const int64_t* S,
                const int64_t* n_kept, const int64_t* dst_rows, const int64_t* n_samples, const int64_t* list_off, int64_t dst_total,
                void* stream) {
  RowsSegs d;
  OP_CHECK_ARG(map && list, "rows_map: null pointer");
  OP_CHECK_ARG(rows_segs(d, nseg, src_row0, dst_row0, S, n_kept, dst_rows, n_samples, list_off), "rows_map: bad segment table");
  if (dst_total == 0) return OP_OK;
  hipLaunchKernelGGL(rows_map_kernel, dim3((int)((dst_total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, map, list, d, dst_total);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

}  // extern "C"
