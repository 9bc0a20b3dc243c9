// Scaled-dot-product attention with relative-position bias and key-padding mask for gfx950 (head_dim = 64).
//
// Replaces multihead_attention.py:102-115 (bmm QK^T, += bias, fp32 softmax, bmm PV) and the xformers
// memory_efficient_attention seam at :79-101.  The reference materialises scores [B*heads, S, S] (bf16 + an
// fp32 softmax copy) and a dense bias [B, heads, S, S]; here scores never leave registers and the bias is the
// per-table [heads][S][Spad] image built once by op_relpos_bias_build (shared by every sample, L2 resident).
//
// Formulation ("swapped", everything stays in registers):
//   S^T[key][q]  = K . Q^T        first MFMA operand = K rows (ds_read_b128 from a swizzled LDS tile),
//                                 second = Q rows held in registers.  Result layout: lane (g,t) holds query
//                                 column q = t and keys g*4+r of every 16-key block -> the softmax reduction
//                                 over keys is 16 in-lane values + 2 cross-lane steps (xor 16, 32).
//   O^T[d][q]   += V^T . P^T      P^T is already in second-operand layout (contraction index = keys = the rows
//                                 of the S^T accumulator); V^T fragments come from the row-major V tile through
//                                 ds_read_b64_tr_b16 (hardware transpose read), key slots permuted to match.
// One workgroup = 4 waves x 32 queries (BQ = 128) of one (sample, head); K/V tiles of 64 keys are staged
// global -> registers -> LDS with the next tile's loads in flight during the current tile's MFMAs.
//
// Roofline: MFMA.  Algorithmic flops per launch = 4 * B * heads * S * S * 64 (QK^T + PV, 2 flops/MAC).
#include "common.h"

namespace {

constexpr int HD = 64;
constexpr int BQ = 128;
constexpr int BKV = 64;
constexpr int VSTRIDE = 160;  // bytes per V row in LDS (128 + 32 pad: 8 consecutive rows tile the 64 banks)

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; int64_t ld;  // row stride (elements) of q/k/v rows
  const bf16_t* bias;      // [heads][S][Spad] or null
  const uint8_t* key_pad;  // [B][Spad] (1 = padded key) or null
  bf16_t* out; int64_t ldo;
  float* lse;              // [B][heads][S]
  int B, S, Spad, heads;
  float scale;
};

__device__ __forceinline__ bf16x8 pack8(const float* a, const float* b) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = (bf16_t)a[i]; r[4 + i] = (bf16_t)b[i]; }
  return r;
}

__device__ __forceinline__ s16x4 tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

__device__ __forceinline__ bf16x8 join_tr(s16x4 a, s16x4 b) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[BKV * 128 + BKV * VSTRIDE];
  char* ldsK = smem;
  char* ldsV = smem + BKV * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, t = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * BQ + wid * 32;
  const bool wave_active = q0 < p.S;
  const int64_t row_base = (int64_t)b * p.S;

  // ---- Q fragments (second MFMA operand): lane (g,t) <- Q[q0 + qb*16 + t][kk*32 + g*8 .. +7] ----
  bf16x8 qf[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qi = min(q0 + qb * 16 + t, p.S - 1);
    const bf16_t* qp = p.q + (row_base + qi) * p.ld + h * HD;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[qb][kk] = *reinterpret_cast<const bf16x8*>(qp + kk * 32 + g * 8);
  }

  f32x4 ot[2][4];
  float m_run[2], l_run[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
#pragma unroll
    for (int db = 0; db < 4; ++db) ot[qb][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging map: thread -> two 16-byte chunks of the K tile and of the V tile ----
  u32x4 rk[2], rv[2];
  int st_row[2], st_c[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c2 = tid + 256 * i;
    st_row[i] = c2 >> 3;
    st_c[i] = c2 & 7;
  }
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kr = min(k0 + st_row[i], p.S - 1);
      const int64_t off = (row_base + kr) * p.ld + h * HD + st_c[i] * 8;
      rk[i] = *reinterpret_cast<const u32x4*>(p.k + off);
      rv[i] = *reinterpret_cast<const u32x4*>(p.v + off);
    }
  };
  auto write_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(ldsK + st_row[i] * 128 + ((st_c[i] ^ (st_row[i] & 7)) << 4)) = rk[i];
      *reinterpret_cast<u32x4*>(ldsV + st_row[i] * VSTRIDE + st_c[i] * 16) = rv[i];
    }
  };

  const int ntiles = (p.S + BKV - 1) / BKV;
  load_tile(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * BKV;
    __syncthreads();
    write_tile();
    __syncthreads();
    if (kt + 1 < ntiles) load_tile(k0 + BKV);
    if (!wave_active) continue;  // wave-uniform; barriers above are still reached every iteration

    // ---- S^T = K . Q^T ----
    f32x4 st[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) st[qb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ldsK + (kb * 16 + t) * 128 + (((kk * 4 + g) ^ (t & 7)) << 4));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          st[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][kk], st[qb][kb], 0, 0, 0);
      }
    }

    // ---- scale + bias + masks; online softmax per query column ----
    unsigned padw[4] = {0u, 0u, 0u, 0u};
    if (p.key_pad) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        padw[kb] = *reinterpret_cast<const unsigned*>(p.key_pad + (int64_t)b * p.Spad + k0 + kb * 16 + g * 4);
    }
    bf16x8 pf[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qi = min(q0 + qb * 16 + t, p.S - 1);
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int key = k0 + kb * 16 + g * 4;
        float bb[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
          const bf16x4 bv = *reinterpret_cast<const bf16x4*>(p.bias + ((int64_t)h * p.S + qi) * p.Spad + key);
#pragma unroll
          for (int r = 0; r < 4; ++r) bb[r] = (float)bv[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = st[qb][kb][r] * p.scale + bb[r];
          const bool masked = (key + r >= p.S) || ((padw[kb] >> (8 * r)) & 0xffu);
          s = masked ? -INFINITY : s;
          st[qb][kb][r] = s;
          mx = fmaxf(mx, s);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[qb], mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __expf(m_run[qb] - m_use);  // m_run = -inf -> 0
      m_run[qb] = m_new;
      float psum = 0.f;
      float pv[4][4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __expf(st[qb][kb][r] - m_use);
          pv[kb][r] = e;
          psum += e;
        }
      l_run[qb] = l_run[qb] * alpha + psum;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[qb][db][r] *= alpha;
      pf[qb][0] = pack8(pv[0], pv[1]);  // key slots e: kb = 2m + (e >> 2), r = e & 3
      pf[qb][1] = pack8(pv[2], pv[3]);
    }

    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const char* base = ldsV + (t >> 2) * VSTRIDE + (db * 16 + (t & 3) * 4) * 2;
        const s16x4 v0 = tr_read(base + ((2 * m) * 16 + g * 4) * VSTRIDE);
        const s16x4 v1 = tr_read(base + ((2 * m + 1) * 16 + g * 4) * VSTRIDE);
        const bf16x8 vf = join_tr(v0, v1);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          ot[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][m], ot[qb][db], 0, 0, 0);
      }
    }
  }

  if (!wave_active) return;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l = l_run[qb];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const int qi = q0 + qb * 16 + t;
    if (qi >= p.S) continue;
    const float inv = 1.f / l;
    bf16_t* op = p.out + (row_base + qi) * p.ldo + h * HD;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(ot[qb][db][r] * inv);
      *reinterpret_cast<bf16x4*>(op + db * 16 + g * 4) = o;
    }
    if (g == 0 && p.lse) p.lse[((int64_t)b * p.heads + h) * p.S + qi] = m_run[qb] + logf(l);
  }
}

}  // namespace

extern "C" int op_prof_begin(int family, double work, void* stream);
extern "C" void op_prof_end(int slot, void* stream);

extern "C" {

// q, k, v: bf16 rows of `ld` elements (row = b*S + s), head h occupies columns [h*64, h*64+64) of each pointer
// (so one packed [B*S, 3H] projection output serves all three with pointer offsets 0, H, 2H).
// bias: bf16 [heads][S][Spad] or null.  key_pad: uint8 [B][Spad], non-zero = masked key, or null.
// out: bf16 [B*S][ldo] (head h at columns h*64..).  lse: fp32 [B][heads][S] (natural log) or null.
int op_attn_fwd(const void* q, const void* k, const void* v, int64_t ld, const void* bias, const void* key_pad, void* out,
                int64_t ldo, float* lse, int64_t B, int64_t S, int64_t Spad, int64_t heads, int64_t head_dim, float scale,
                void* stream) {
  OP_CHECK_ARG(q && k && v && out, "attn_fwd: null pointer");
  OP_CHECK_ARG(head_dim == HD, "attn_fwd: head_dim %lld unsupported (only 64)", (long long)head_dim);
  OP_CHECK_ARG(B > 0 && S > 0 && heads > 0 && ld % 8 == 0 && ldo % 4 == 0, "attn_fwd: bad sizes");
  OP_CHECK_ARG((!bias && !key_pad) || (Spad >= ((S + 63) / 64) * 64 && Spad % 8 == 0),
               "attn_fwd: Spad must be >= S rounded up to 64");
  AttnArgs a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.ld = ld;
  a.bias = (const bf16_t*)bias; a.key_pad = (const uint8_t*)key_pad;
  a.out = (bf16_t*)out; a.ldo = ldo; a.lse = lse;
  a.B = (int)B; a.S = (int)S; a.Spad = (int)Spad; a.heads = (int)heads; a.scale = scale;
  dim3 grid(ceil_div(S, BQ), (unsigned)heads, (unsigned)B);
  const int slot = op_prof_begin(1, 4.0 * (double)B * (double)heads * (double)S * (double)S * HD, stream);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  op_prof_end(slot, stream);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

}  // extern "C"
