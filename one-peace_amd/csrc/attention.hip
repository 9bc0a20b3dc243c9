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
#include <type_traits>

#include "common.h"

namespace {

constexpr int HD = 64;
constexpr int BQ = 128;
constexpr int BKV = 64;
constexpr int VSTRIDE = 160;  // bytes per V row in LDS (128 + 32 pad: 8 consecutive rows tile the 64 banks)

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; int64_t ld;  // row stride (elements) of q/k/v rows
  const bf16_t* bias;      // [heads][S][Spad] or null
  int64_t bias_bs;         // elements between the bias images of consecutive samples (0: one image shared by all samples)
  const uint8_t* key_pad;  // [B][Spad] (1 = padded key) or null
  bf16_t* out; int64_t ldo;
  float* lse;              // [B][heads][lse_ld]
  int64_t lse_ld;
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

// byte offset of the 8-byte piece a transpose-read lane supplies inside a [64 rows][64 cols] bf16 tile stored with
// 128-byte rows and 16-byte slots XOR-swizzled by (row & 7):  row = rowblk*16 + g*4 + (t>>2), cols db*16 + (t&3)*4 ..+3
__device__ __forceinline__ int tr_off_swz(int rowblk, int db, int g, int t) {
  const int row = rowblk * 16 + g * 4 + (t >> 2);
  const int c = db * 2 + ((t >> 1) & 1);
  return row * 128 + ((c ^ (row & 7)) << 4) + (t & 1) * 8;
}

// XCD-aware work mapping.  Workgroups are dispatched round-robin over the 8 XCDs by linear id, so the 2-5 workgroups that
// share the K/V (or Q/dO) rows of one (sample, head) would land on different XCDs and each L2 would fetch those rows from
// HBM again.  Re-deal the linear id so that consecutive work items run on ONE XCD: (bx, by, bz) replace blockIdx.
__device__ __forceinline__ void xcd_work_item(int& bx, int& by, int& bz) {
  const int nx = gridDim.x, ny = gridDim.y;
  const int total = nx * ny * (int)gridDim.z;
  int lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
  if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
  bx = lin % nx;
  by = (lin / nx) % ny;
  bz = lin / (nx * ny);
}

// HAS_BIAS / HAS_PAD are template parameters so that the bias and key-pad loads of a tile are plain straight-line loads,
// issued together BEFORE the tile's QK^T MFMAs (as run-time `if (p.bias)` branches the compiler emitted eight serialised
// load -> s_waitcnt pairs per tile after them: one exposed L2 round trip per fragment).
template <bool HAS_BIAS, bool HAS_PAD>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[BKV * 128 + BKV * VSTRIDE];
  char* ldsK = smem;
  char* ldsV = smem + BKV * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, t = lane & 15;
  int bx, h, b;
  xcd_work_item(bx, h, b);
  const int q0 = bx * BQ + wid * 32;
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

  // per-lane bias / key-pad row bases (clamped query row, key offset g*4 folded in); every tile of 64 keys lies inside Spad
  const bf16_t* brow[2] = {nullptr, nullptr};
  if constexpr (HAS_BIAS) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qi = min(q0 + qb * 16 + t, p.S - 1);
      brow[qb] = p.bias + (int64_t)b * p.bias_bs + ((int64_t)h * p.S + qi) * p.Spad + g * 4;
    }
  }
  const uint8_t* padrow = HAS_PAD ? p.key_pad + (int64_t)b * p.Spad + g * 4 : nullptr;

  // One 64-key tile.  FULL: all 64 keys are inside the sequence -> no bounds masks, no partial-block control flow.
  auto tile = [&](const int k0, auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    const int nkb = FULL ? 4 : ((p.S - k0 + 15) >> 4);  // valid 16-key blocks (1..4)
    bf16x4 bv[2][4];
    unsigned padw[4] = {0u, 0u, 0u, 0u};
    if constexpr (HAS_BIAS) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) bv[qb][kb] = *reinterpret_cast<const bf16x4*>(brow[qb] + k0 + kb * 16);
    }
    if constexpr (HAS_PAD) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) padw[kb] = *reinterpret_cast<const unsigned*>(padrow + k0 + kb * 16);
    }

    // ---- S^T = K . Q^T ----
    f32x4 st[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) st[qb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (!FULL && kb >= nkb) break;  // uniform: key blocks past the sequence end are masked anyway
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ldsK + (kb * 16 + t) * 128 + (((kk * 4 + g) ^ (t & 7)) << 4));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          st[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][kk], st[qb][kb], 0, 0, 0);
      }
    }

    // ---- scale + bias + masks; online softmax per query column ----
    bf16x8 pf[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int key = k0 + kb * 16 + g * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = st[qb][kb][r] * p.scale;
          if constexpr (HAS_BIAS) s += (float)bv[qb][kb][r];
          if constexpr (!FULL || HAS_PAD) {
            bool masked = false;
            if constexpr (!FULL) masked = key + r >= p.S;
            if constexpr (HAS_PAD) masked = masked || ((padw[kb] >> (8 * r)) & 0xffu);
            s = masked ? -INFINITY : s;
          }
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
      if (!FULL && 2 * m >= nkb) break;  // P is exactly zero there
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
  };

  const int ntiles = (p.S + BKV - 1) / BKV;
  load_tile(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * BKV;
    __syncthreads();
    write_tile();
    __syncthreads();
    // (measured: issuing this prefetch AFTER the tile's bias loads -- so that the bias wait does not include it -- is 10 %
    // slower here, unlike in the backward kernels)
    if (kt + 1 < ntiles) load_tile(k0 + BKV);
    if (!wave_active) continue;  // wave-uniform; the barriers above are still reached every iteration
    if (k0 + BKV <= p.S) tile(k0, std::true_type{});
    else tile(k0, std::false_type{});
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
    if (g == 0 && p.lse) p.lse[((int64_t)b * p.heads + h) * p.lse_ld + qi] = m_run[qb] + logf(l);
  }
}


// =====================================================================================================================
// "Resident" forward for sequences of up to RES_MAX_S keys -- the model's 64 / 250 / 257-token streams (text, 5 s audio,
// 256^2 image).  What round 2's ablations of the streaming kernel above showed at these lengths (tools/attn_abl.py,
// profiles/r2_attention_notes.md): the MFMAs are 25 % of a wave's issue time, the softmax VALU stream (8.6 instructions
// per score: scale, bf16 -> f32 bias conversion + add, max, subtract, exp argument scaling, exp, sum, pack) is the bound,
// and the bias fragments -- 8-byte pieces of 16 different lines per load instruction -- cost another 40 % in the texture
// addresser.  So this kernel
//   * adds the bias with the MATRIX pipe: the bias image is stored "fragment-major" (op_attn_bias_pack: one contiguous
//     1 KiB block per [16 queries x 32 keys], laid out exactly as the first-operand fragment of two 16-key blocks), one
//     fully coalesced 16-byte load per lane fetches two blocks, and  S^T += Bias^T-fragment . Selector  (selector = 1/scale
//     on the diagonal, exact in bf16 for head_dim 64) adds it inside the accumulator: no conversion, no add, no 8-byte
//     loads;
//   * works in the exp2 domain: p = exp2(acc * (scale * log2 e) - m2) is ONE fma + ONE v_exp_f32 per score (3.5 VALU
//     instructions per score all told);
//   * keeps the WHOLE K and V of its (sample, head) resident in LDS (S x 128 B each, <= 80 KiB together, staged once by
//     LDS-DMA; ONE barrier), after which every wave runs its own 16-query block over all key tiles with no barriers and
//     no lockstep -- a tail tile only costs its valid 16-key blocks -- at 5 waves per SIMD (<= 96 VGPRs);
//   * deals the 16-query blocks of a (sample, head) evenly to ceil(nqb / 16) ... workgroups of nw = blocks-per-workgroup
//     waves (S = 257: 17 blocks -> 2 workgroups of 9 and 8 waves instead of 3 x 128 query rows; S = 64: 4 waves).
// LDS image: rows of 128 B, 16-byte slot index XOR (row & 7), for K (ds_read_b128 fragments) AND V (ds_read_b64_tr_b16
// fragments, the image the backward kernels already read K^T from: bank-conflict free for both).
// =====================================================================================================================
constexpr int RES_MAX_S = 320;   // 2 x 320 rows x 128 B = 80 KiB -> two workgroups per CU
constexpr int RES_MAX_NW = 10;   // waves per workgroup (20 query blocks at S = 320 -> 2 workgroups)
constexpr float LOG2E = 1.4426950408889634f;

// elements of one fragment-major bias block: [16 queries x 32 keys] = 64 lanes x 8 bf16
constexpr int FRAG_BLOCK = 512;

template <bool HAS_BIAS, bool HAS_PAD>
__global__ __launch_bounds__(RES_MAX_NW * 64, 5) void attn_fwd_res_kernel(AttnArgs p, const bf16_t* __restrict__ bias_frag,
                                                                            int rows_pad, int qb_per_wg, int abl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsK = smem;
  char* ldsV = smem + rows_pad * 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int g = lane >> 4, t = lane & 15;
  int bx, h, b;
  xcd_work_item(bx, h, b);
  const int64_t row_base = (int64_t)b * p.S;

  // ---- stage K and V: one wave-level LDS-DMA instruction = 8 rows x 128 B (1 KiB, lane-linear in LDS; the swizzle is
  // applied to the per-lane SOURCE chunk).  Rows past the sequence end re-read row S-1: finite values, P is 0 there. ----
  if (!(abl & 1)) {  // (abl: timing ablations, tools only -- 1 = no K/V staging, 2 = no compute)
    const int ngrp = rows_pad >> 3;
    const int r_in = lane >> 3, slot = lane & 7;
    for (int grp = wid; grp < 2 * ngrp; grp += nw) {
      const bool isv = grp >= ngrp;
      const int gi = isv ? grp - ngrp : grp;
      const int r = gi * 8 + r_in;
      const int kr = min(r, p.S - 1);
      const bf16_t* src = (isv ? p.v : p.k) + (row_base + kr) * p.ld + h * HD + ((slot ^ (r & 7)) << 3);
      char* dst = (isv ? ldsV : ldsK) + gi * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  }

  // ---- this wave's 16-query blocks: wid, wid + nw, ... of the workgroup's qb_per_wg (normally one: nw = qb_per_wg; the launch may
  // give a workgroup fewer waves than blocks -- S = 257: 17 blocks as 9 + 8 on workgroups of EIGHT waves, which spread evenly over
  // the four SIMDs, the ninth block being a second trip of wave 0) ----
  const int nqb = (p.S + 15) >> 4, nkp = (p.S + 31) >> 5;
  for (int lblk = wid, trip = 0; trip == 0 || lblk < qb_per_wg; lblk += nw, ++trip) {
  const int qblk = bx * qb_per_wg + lblk;
  const bool active = lblk < qb_per_wg && qblk < nqb;
  const int q0 = min(qblk, nqb - 1) * 16;
  const int qi = min(q0 + t, p.S - 1);

  bf16x8 qf[2];
  {
    const bf16_t* qp = p.q + (row_base + qi) * p.ld + h * HD;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qp + kk * 32 + g * 8);
  }
  // bias fragments of this query block: block (kp) at fbase + kp * FRAG_BLOCK, 16 bytes per lane
  const bf16_t* fbase = nullptr;
  if constexpr (HAS_BIAS) {
    const int64_t per_head = (int64_t)nqb * nkp * FRAG_BLOCK;
    fbase = bias_frag + ((p.bias_bs != 0 ? (int64_t)b * p.heads : 0) + h) * per_head + (int64_t)(q0 >> 4) * nkp * FRAG_BLOCK + lane * 8;
  }
  const uint8_t* padrow = HAS_PAD ? p.key_pad + (int64_t)b * p.Spad + g * 4 : nullptr;
  // selectors: second operand that copies first-operand column j = t (key block 0 of a pair) / j = 16 + t (block 1) of
  // the bias fragment into output column t, times 1/scale
  bf16x8 sel_lo, sel_hi;
  {
    const bf16_t inv = (bf16_t)(1.0f / p.scale), zero = (bf16_t)0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sel_lo[i] = (g * 8 + i == t) ? inv : zero;
      sel_hi[i] = (g * 8 + i == 16 + t) ? inv : zero;
    }
  }

  if (trip == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the only barrier: K and V are resident
  }
  if (!active || (abl & 2)) continue;

  f32x4 ot[4];
  float m2_run = -INFINITY, l_run = 0.f;  // running max in the exp2 domain (scores * scale * log2 e), running sum
#pragma unroll
  for (int db = 0; db < 4; ++db) ot[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int trsw[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) trsw[db] = tr_off_swz(0, db, g, t);
  const int kswz[2] = {((0 * 4 + g) ^ (t & 7)) << 4, ((1 * 4 + g) ^ (t & 7)) << 4};
  const float c1 = p.scale * LOG2E;

  auto tile = [&](const int k0, auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    const int nkb = FULL ? 4 : ((p.S - k0 + 15) >> 4);  // valid 16-key blocks (1..4)
    bf16x8 bf[2];
    unsigned padw[4] = {0u, 0u, 0u, 0u};
    if constexpr (HAS_BIAS) {  // (a one-tile-ahead register prefetch spills at 96 VGPRs and measured 5-15 % slower)
#pragma unroll
      for (int m = 0; m < 2; ++m)
        if (FULL || 2 * m < nkb) bf[m] = *reinterpret_cast<const bf16x8*>(fbase + ((k0 >> 5) + m) * FRAG_BLOCK);
    }
    if constexpr (HAS_PAD) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) padw[kb] = *reinterpret_cast<const unsigned*>(padrow + k0 + kb * 16);
    }
    const char* kt = ldsK + k0 * 128;
    const char* vt = ldsV + k0 * 128;

    // ---- S^T = K . Q^T  (+ Bias^T / scale through the selector) ----
    f32x4 st[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) st[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (!FULL && kb >= nkb) break;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt + (kb * 16 + t) * 128 + kswz[kk]);
        st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], st[kb], 0, 0, 0);
      }
      if constexpr (HAS_BIAS)
        st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[kb >> 1], (kb & 1) ? sel_hi : sel_lo, st[kb], 0, 0, 0);
    }

    // ---- masks; online softmax per query column in the exp2 domain (only over the valid key blocks of a tail tile) ----
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (!FULL && kb >= nkb) break;
      const int key = k0 + kb * 16 + g * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (!FULL || HAS_PAD) {
          bool masked = false;
          if constexpr (!FULL) masked = key + r >= p.S;
          if constexpr (HAS_PAD) masked = masked || ((padw[kb] >> (8 * r)) & 0xffu);
          st[kb][r] = masked ? -INFINITY : st[kb][r];
        }
        mx = fmaxf(mx, st[kb][r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m2_new = fmaxf(m2_run, mx * c1);
    const float m2_use = (m2_new == -INFINITY) ? 0.f : m2_new;
    const float alpha = __builtin_amdgcn_exp2f(m2_run - m2_use);  // m2_run = -inf -> 0
    m2_run = m2_new;
    float psum = 0.f;
    float pv[4][4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (FULL || kb < nkb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][r], c1, -m2_use));
          pv[kb][r] = e;
          psum += e;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[kb][r] = 0.f;
      }
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r) ot[db][r] *= alpha;
    bf16x8 pf[2];
    pf[0] = pack8(pv[0], pv[1]);  // key slots e: kb = 2m + (e >> 2), r = e & 3
    pf[1] = pack8(pv[2], pv[3]);

    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (!FULL && 2 * m >= nkb) break;  // P is exactly zero there
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const bf16x8 vf = join_tr(tr_read(vt + trsw[db] + (2 * m) * 2048), tr_read(vt + trsw[db] + (2 * m + 1) * 2048));
        ot[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[m], ot[db], 0, 0, 0);
      }
    }
  };

  const int ntiles = (p.S + BKV - 1) / BKV;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * BKV;
    if (k0 + BKV <= p.S) tile(k0, std::true_type{});
    else tile(k0, std::false_type{});
  }

  float l = l_run;
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (q0 + t >= p.S) continue;
  const float inv = 1.f / l;
  bf16_t* op = p.out + (row_base + q0 + t) * p.ldo + h * HD;
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(ot[db][r] * inv);
    *reinterpret_cast<bf16x4*>(op + db * 16 + g * 4) = o;
  }
  if (g == 0 && p.lse) p.lse[((int64_t)b * p.heads + h) * p.lse_ld + q0 + t] = m2_run * (1.0f / LOG2E) + logf(l);
  }
}

// =====================================================================================================================
// Persistent forward for the 193 ... 257-token streams (5 s audio: 250, 256^2 image: 257) -- round 4.
// What the resident kernel above costs at these lengths (tools/attn_bench.py + PMC, B = 128): a workgroup lives ~12 us, of which
// its waves COMPUTE ~2.5 us -- every workgroup first stages 66 ... 74 KiB of K / V (a prologue that runs at the ~11 B/cycle/CU of
// a cold burst: 2.5 ... 3 us with every CU doing the same) and sets up selectors / addresses for ONE 16-query block per wave;
// two workgroups per CU overlap only partly (SQ_WAIT_ANY 57 % of the wave cycles), K / V are staged twice per (sample, head), and
// at S = 257 the seventeenth query block makes one wave of the first workgroup run twice as long as the other fifteen.  Here:
//   * ONE workgroup of 16 waves per CU walks the (sample, head) items  blockIdx, blockIdx + gridDim, ...  -- the order one-item
//     workgroups are dispatched in, so the 24 heads of a sample (the 128-byte slices of one 9 KiB qkv row) are still read by the
//     chip at the same time;
//   * K / V of item i + 1 are fetched by LDS-DMA into the OTHER half of a double buffer (2 x 72 KiB at 288 rows) while item i is
//     computed: one barrier per item, no staging phase; the next item's Q fragments travel in registers, the bias fragments of the
//     next key tile are requested one tile ahead (there is room for that at 4 waves per SIMD = 128 registers);
//   * wave w owns query block w of EVERY item (selectors, swizzle offsets, fragment addresses are set up once per launch);
//   * S = 257: the lone query of block 16 is spread over the workgroup BY KEYS -- wave w runs it against key block w (wave 0 also
//     against block 16: one QK^T fragment, one softmax step, one half-empty PV fragment: +8 % per wave instead of +100 % for one),
//     the sixteen partial (max, sum, O) triples meet in LDS and are merged by one wave after the NEXT item's barrier.
// Same swapped formulation, bias on the matrix pipe, exp2-domain softmax and LDS image as attn_fwd_res_kernel; the per-tile code is
// that kernel's, so the 16 regular query blocks give the same bits; the lone query's softmax is merged in a different order
// (fp32 online-softmax merge: same value up to fp32 rounding).
// =====================================================================================================================
constexpr int PERS_NW = 8;        // waves per workgroup: two 16-query blocks each (256 registers per wave: everything prefetched, no spills;
                                  // sixteen waves of one block each -- 128 registers -- spilled 35 of them, and a spill reload next to
                                  // LDS-DMA is an s_waitcnt vmcnt(0): the K / V fetch of the next item stopped overlapping)
constexpr int PERS_MAX_ROWS = 288;
constexpr int PERS_SCR = 68;      // floats per (item parity, wave) of the merge scratch: O[64], m2, l, pad
constexpr int PERS_PPT = 2;       // LDS-DMA pieces of the next item a wave issues per key tile (<= 9 pieces over >= 4 tiles; 5 tiles at 9)

// transpose read as inline asm: through the builtin the compiler orders it behind every pending LDS-DMA write (s_waitcnt vmcnt(0) in
// front of the first one after each DMA issue: the next item's K / V fetch would stop overlapping with this item's tiles).  The
// caller waits (lgkmcnt) and fences the scheduler before the first use.
__device__ __forceinline__ s16x4 tr_read_a(const char* p) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"((unsigned)(uintptr_t)((__attribute__((address_space(3))) const char*)p)));
  return r;
}
#define ATTN_WAIT_LGKM0()                         \
  do {                                            \
    __builtin_amdgcn_s_waitcnt(0xC07F);           \
    __builtin_amdgcn_sched_barrier(0);            \
  } while (0)

// The bias / pad loads of the item loop as inline asm + hand-placed waits.  vmcnt counts in ISSUE order, so a load issued behind
// pieces of the next item's fetch cannot be waited for without those pieces; and the compiler, which does its own bookkeeping for
// the loads it can see, answers anything it cannot count (a loop-carried prefetch next to LDS-DMA) with vmcnt(0) = the whole fetch.
// Protocol of a key tile: wait until only the PERS_PPT youngest operations are outstanding (= the previous tile's fetch pieces; this
// tile's small loads, issued before them, have landed), request the NEXT tile's small loads, then this tile's fetch pieces.
__device__ __forceinline__ void gload16_asm(bf16x8& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p)); }
__device__ __forceinline__ void gload4_asm(unsigned& d, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p)); }
// the same with a wave-uniform base in SGPRs and a 32-bit per-lane byte offset (ONE address register for all loads of a row instead
// of a 64-bit pair per load: the kernels at the register limit); OFF: immediate byte offset (< 4096)
// (s_nop 4: a VMEM instruction needs five wait states behind a VALU / SALU write of an SGPR it reads -- the compiler restores spilled
// SGPRs with v_readlane right in front of the statement and does not know what is inside it: the first version of the round-4
// backward kernel read its bases one instruction after they were written and faulted; tools/check_mfma_hazards.py has the rule)
template <int OFF>
__device__ __forceinline__ void gload16_s(bf16x8& d, const void* sbase, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void gload4_s(unsigned& d, const void* sbase, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF));
}

template <bool HAS_BIAS, bool HAS_PAD>
__global__ __launch_bounds__(PERS_NW * 64, 2) void attn_fwd_pers_kernel(AttnArgs p, const bf16_t* __restrict__ bias_frag, int rows_pad,
                                                                         int nitems) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, t = lane & 15;
  const int KVB = 2 * rows_pad * 128;                       // bytes of one K | V buffer
  float* scratch = reinterpret_cast<float*>(smem + 2 * KVB);  // [2][PERS_NW][PERS_SCR]
  const int nqb = (p.S + 15) >> 4, nkp = (p.S + 31) >> 5;
  const bool has_left = nqb > 2 * PERS_NW;   // S = 257 (launch condition: then the 17th block holds exactly one query)
  const int q0 = wid * 32;                   // this wave's query blocks: rows q0 .. q0 + 15 and q0 + 16 .. q0 + 31
  const int nact = min(2, max(0, nqb - 2 * wid));  // how many of them exist (uniform)
  const int64_t per_head = (int64_t)nqb * nkp * FRAG_BLOCK;
  const int ntiles = (p.S + BKV - 1) / BKV;
  const float c1 = p.scale * LOG2E;

  bf16x8 sel_lo, sel_hi;
  {
    const bf16_t inv = (bf16_t)(1.0f / p.scale), zero = (bf16_t)0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sel_lo[i] = (g * 8 + i == t) ? inv : zero;
      sel_hi[i] = (g * 8 + i == 16 + t) ? inv : zero;
    }
  }
  int trsw[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) trsw[db] = tr_off_swz(0, db, g, t);
  const int kswz[2] = {((0 * 4 + g) ^ (t & 7)) << 4, ((1 * 4 + g) ^ (t & 7)) << 4};

  // ---- LDS-DMA of one item's K and V: 8 rows x 128 B per wave-level instruction (see attn_fwd_res_kernel) ----
  const int ngrp = rows_pad >> 3;
  const int r_in = lane >> 3, slot = lane & 7;
  const int npiece = (2 * ngrp - wid + PERS_NW - 1) / PERS_NW;  // pieces of this wave: groups wid, wid + PERS_NW, ...
  auto piece = [&](int it, int buf, int j) {
    const int b = it / p.heads, h = it - b * p.heads;
    const int grp = wid + j * PERS_NW;
    const bool isv = grp >= ngrp;
    const int gi = isv ? grp - ngrp : grp;
    const int r = gi * 8 + r_in;
    const int kr = min(r, p.S - 1);
    const bf16_t* src = (isv ? p.v : p.k) + ((int64_t)b * p.S + kr) * p.ld + h * HD + ((slot ^ (r & 7)) << 3);
    char* dst = smem + buf * KVB + (isv ? rows_pad * 128 : 0) + gi * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto load_q = [&](int it, int qrow, bf16x8 (&q)[2]) {
    const int b = it / p.heads, h = it - b * p.heads;
    const bf16_t* qp = p.q + ((int64_t)b * p.S + qrow) * p.ld + h * HD;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) q[kk] = *reinterpret_cast<const bf16x8*>(qp + kk * 32 + g * 8);
  };
  const int qi_main[2] = {min(q0 + t, p.S - 1), min(q0 + 16 + t, p.S - 1)};

  // small loads of (item, key tile): bias fragments of this wave's two query blocks (a block that does not exist re-reads the last
  // one: never stored) and the key-pad words; tile index == ntiles means "tile 0 of the next item" for the caller
  bf16x8 bn[2][2];
  unsigned padn[4] = {0u, 0u, 0u, 0u};
  auto small_loads = [&](int it, int kt) {
    const int b = it / p.heads, h = it - b * p.heads;
    if constexpr (HAS_BIAS) {
      const bf16_t* fhead = bias_frag + ((p.bias_bs != 0 ? (int64_t)b * p.heads : 0) + h) * per_head + lane * 8;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const bf16_t* fb = fhead + (int64_t)min(2 * wid + qb, nqb - 1) * nkp * FRAG_BLOCK;
#pragma unroll
        for (int m = 0; m < 2; ++m) gload16_asm(bn[qb][m], fb + min(2 * kt + m, nkp - 1) * FRAG_BLOCK);
      }
    }
    if constexpr (HAS_PAD) {
      const uint8_t* pr = p.key_pad + (int64_t)b * p.Spad + g * 4 + kt * BKV;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) gload4_asm(padn[kb], pr + kb * 16);
    }
  };

  // late loads, issued in the LAST key tile of an item (registers are free there: S = 257's fifth tile is one 16-key block): the Q
  // fragments of the next item and, with a lone query, its Q fragments, bias fragments and pad words for this wave's key pair(s)
  constexpr int NSMALL = (HAS_BIAS ? 4 : 0) + (HAS_PAD ? 4 : 0);  // operations of small_loads
  bf16x8 qn[2][2], ql[2], bfl[2];
  unsigned padl[2][2] = {{0u, 0u}, {0u, 0u}};
  auto late_loads = [&](int it_next, int it) {
    {
      const int b = it_next / p.heads, h = it_next - b * p.heads;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const bf16_t* qp = p.q + ((int64_t)b * p.S + qi_main[qb]) * p.ld + h * HD;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) gload16_asm(qn[qb][kk], qp + kk * 32 + g * 8);
      }
    }
    if (has_left) {
      const int b = it / p.heads, h = it - b * p.heads;
      const bf16_t* qp = p.q + ((int64_t)b * p.S + p.S - 1) * p.ld + h * HD;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) gload16_asm(ql[kk], qp + kk * 32 + g * 8);
      if constexpr (HAS_BIAS) {
        const bf16_t* fleft = bias_frag + ((p.bias_bs != 0 ? (int64_t)b * p.heads : 0) + h) * per_head + lane * 8 + (int64_t)(nqb - 1) * nkp * FRAG_BLOCK;
        gload16_asm(bfl[0], fleft + wid * FRAG_BLOCK);
        gload16_asm(bfl[1], fleft + min(PERS_NW, nkp - 1) * FRAG_BLOCK);
      }
      if constexpr (HAS_PAD) {
        const uint8_t* pr = p.key_pad + (int64_t)b * p.Spad + g * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          gload4_asm(padl[0][j], pr + (2 * wid + j) * 16);
          gload4_asm(padl[1][j], pr + min((2 * PERS_NW + j) * 16, p.Spad - 16));
        }
      }
    }
  };

  int item = blockIdx.x;
  const int step = gridDim.x;
  if (item < nitems) {
    for (int j = 0; j < npiece; ++j) piece(item, 0, j);
    load_q(item, qi_main[0], qn[0]);
    load_q(item, qi_main[1], qn[1]);
    small_loads(item, 0);
  }
  int buf = 0, prev_item = -1;
  // everything the item loop finds in flight at its top has landed at the END of the previous trip (or here): see the tile
  auto wait_all = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qn[0][0]), "+v"(qn[0][1]), "+v"(qn[1][0]), "+v"(qn[1][1]), "+v"(bn[0][0]), "+v"(bn[0][1]), "+v"(bn[1][0]), "+v"(bn[1][1]),
                 "+v"(padn[0]), "+v"(padn[1]), "+v"(padn[2]), "+v"(padn[3]) : : "memory");
  };
  wait_all();

  // merge of the lone query's partials of item `it` (scratch half `sb`) by the calling wave: lane = head dimension
  auto merge_left = [&](int it, int sb) {
    const float* sc = scratch + sb * (PERS_NW * PERS_SCR);
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < PERS_NW; ++w) M = fmaxf(M, sc[w * PERS_SCR + 64]);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < PERS_NW; ++w) {
      const float f = __builtin_amdgcn_exp2f(sc[w * PERS_SCR + 64] - M);  // (-inf - M = -inf -> 0: a wave whose keys were all masked)
      L = __builtin_fmaf(sc[w * PERS_SCR + 65], f, L);
      o = __builtin_fmaf(sc[w * PERS_SCR + lane], f, o);
    }
    const int b = it / p.heads, h = it - b * p.heads;
    const int qrow = p.S - 1;
    p.out[((int64_t)b * p.S + qrow) * p.ldo + h * HD + lane] = (bf16_t)(o / L);
    if (lane == 0 && p.lse) p.lse[((int64_t)b * p.heads + h) * p.lse_ld + qrow] = M * (1.0f / LOG2E) + logf(L);
  };

  for (; item < nitems; item += step, buf ^= 1) {
    const int b = item / p.heads, h = item - b * p.heads;
    const int64_t row_base = (int64_t)b * p.S;
    __syncthreads();  // this item's K / V have landed (every wave waited for its own pieces at the end of the previous trip); every
                      // wave is done with the other buffer and with the scratch half it re-uses
    bf16x8 qf[2][2] = {{qn[0][0], qn[0][1]}, {qn[1][0], qn[1][1]}};
    const bool more = item + step < nitems;
    const int nxt = more ? item + step : item;  // (no next item: the current one is fetched again into the idle buffer -- nothing branches)
    if (has_left && prev_item >= 0 && wid == ((prev_item / step) & (PERS_NW - 1))) merge_left(prev_item, buf ^ 1);
    prev_item = item;

    const char* ldsK = smem + buf * KVB;
    const char* ldsV = ldsK + rows_pad * 128;

    f32x4 ot[2][4];
    float m2_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int db = 0; db < 4; ++db) ot[qb][db] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto tile = [&](const int kt, auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
      const int k0 = kt * BKV;
      const int nkb = FULL ? 4 : ((p.S - k0 + 15) >> 4);
      // ---- the pipeline of small loads and fetch pieces (see above).  The wait for a tile's small loads stands at the END of the
      // previous tile's code (below), not here: a register an inline-asm load is still writing must not be live across a loop back
      // edge -- the register allocator is free to put copies at a loop header, and it did (round 4: copies of the in-flight Q
      // fragments in front of the wait of a sample loop = stale operands, NaN; tools/check_mfma_hazards.py looks for exactly that) ----
      bf16x8 bf[2][2];
      unsigned padw[4] = {0u, 0u, 0u, 0u};
      if constexpr (HAS_BIAS) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int m = 0; m < 2; ++m) bf[qb][m] = bn[qb][m];
      }
      if constexpr (HAS_PAD) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) padw[kb] = padn[kb];
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (FULL) {  // (FULL = not the last tile of the item: the caller peels that one)
        small_loads(item, kt + 1);
      } else {
        late_loads(nxt, item);
        small_loads(nxt, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 0; jj < PERS_PPT; ++jj) piece(nxt, buf ^ 1, min(kt * PERS_PPT + jj, npiece - 1));  // (past the last piece: that piece again)
      __builtin_amdgcn_sched_barrier(0);
      if (nact == 0) return;  // (uniform) a wave without query blocks (short sequences) only fetches its share

      const char* kt_ = ldsK + k0 * 128;
      const char* vt = ldsV + k0 * 128;
      f32x4 st[2][4];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) st[qb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        if (!FULL && kb >= nkb) break;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt_ + (kb * 16 + t) * 128 + kswz[kk]);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) st[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][kk], st[qb][kb], 0, 0, 0);
        }
        if constexpr (HAS_BIAS) {
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
            st[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[qb][kb >> 1], (kb & 1) ? sel_hi : sel_lo, st[qb][kb], 0, 0, 0);
        }
      }
      // V^T fragments of the tile's first 32 keys: requested BEFORE the softmax arithmetic, which covers their LDS latency (the second
      // half is requested in front of the first half's MFMAs); waiting right behind the request left ~150 cycles exposed twice per
      // tile with two waves per SIMD
      s16x4 va0[4], va1[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        va0[db] = tr_read_a(vt + trsw[db]);
        va1[db] = tr_read_a(vt + trsw[db] + 2048);
      }
      bf16x8 pf[2][2];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          if (!FULL && kb >= nkb) break;
          const int key = k0 + kb * 16 + g * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (!FULL || HAS_PAD) {
              bool masked = false;
              if constexpr (!FULL) masked = key + r >= p.S;
              if constexpr (HAS_PAD) masked = masked || ((padw[kb] >> (8 * r)) & 0xffu);
              st[qb][kb][r] = masked ? -INFINITY : st[qb][kb][r];
            }
            mx = fmaxf(mx, st[qb][kb][r]);
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m2_new = fmaxf(m2_run[qb], mx * c1);
        const float m2_use = (m2_new == -INFINITY) ? 0.f : m2_new;
        const float alpha = __builtin_amdgcn_exp2f(m2_run[qb] - m2_use);
        m2_run[qb] = m2_new;
        float psum = 0.f;
        float pv[4][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          if (FULL || kb < nkb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qb][kb][r], c1, -m2_use));
              pv[kb][r] = e;
              psum += e;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[kb][r] = 0.f;
          }
        }
        l_run[qb] = l_run[qb] * alpha + psum;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int r = 0; r < 4; ++r) ot[qb][db][r] *= alpha;
        pf[qb][0] = pack8(pv[0], pv[1]);
        pf[qb][1] = pack8(pv[2], pv[3]);
      }
      const bool second = FULL || nkb > 2;  // (uniform) keys 32 .. 63 of the tile hold keys
      ATTN_WAIT_LGKM0();
      s16x4 vb0[4], vb1[4];
      if (second) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          vb0[db] = tr_read_a(vt + trsw[db] + 2 * 2048);
          vb1[db] = tr_read_a(vt + trsw[db] + 3 * 2048);
        }
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const bf16x8 vf = join_tr(va0[db], va1[db]);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) ot[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][0], ot[qb][db], 0, 0, 0);
      }
      if (second) {
        ATTN_WAIT_LGKM0();
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 vf = join_tr(vb0[db], vb1[db]);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) ot[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][1], ot[qb][db], 0, 0, 0);
        }
      }
    };
    // the next tile's small loads have landed when only this tile's PERS_PPT fetch pieces are outstanding (see the tile)
    auto wait_small = [&]() {
      if constexpr (HAS_BIAS && HAS_PAD)
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(bn[0][0]), "+v"(bn[0][1]), "+v"(bn[1][0]), "+v"(bn[1][1]), "+v"(padn[0]), "+v"(padn[1]), "+v"(padn[2]), "+v"(padn[3]) : "n"(PERS_PPT));
      else if constexpr (HAS_BIAS)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(bn[0][0]), "+v"(bn[0][1]), "+v"(bn[1][0]), "+v"(bn[1][1]) : "n"(PERS_PPT));
      else if constexpr (HAS_PAD)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(padn[0]), "+v"(padn[1]), "+v"(padn[2]), "+v"(padn[3]) : "n"(PERS_PPT));
    };
#pragma unroll 1
    for (int kt = 0; kt < ntiles - 1; ++kt) {
      tile(kt, std::true_type{});
      wait_small();
    }
    tile(ntiles - 1, std::false_type{});  // the last tile: partial (or full at S = 256: its masks are then all false); issues the late loads
    if (has_left) {  // (BEFORE the stores below: the hand-counted wait in here must see exactly the operations of the last tile behind it)
      // ---- the lone query (row S - 1 = 256) against this wave's key blocks 2w, 2w + 1 (wave 0: also block 16): every query column of
      // the fragment is that row ----
      f32x4 ol[4];
      float m2 = -INFINITY, lsum = 0.f;
#pragma unroll
      for (int db = 0; db < 4; ++db) ol[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // the late loads of the last tile: behind them only the next item's first small loads and that tile's fetch pieces were issued
      asm volatile("s_waitcnt vmcnt(%8)" : "+v"(ql[0]), "+v"(ql[1]), "+v"(bfl[0]), "+v"(bfl[1]), "+v"(padl[0][0]), "+v"(padl[0][1]), "+v"(padl[1][0]), "+v"(padl[1][1]) : "n"(NSMALL + PERS_PPT));
      const int npc = wid == 0 ? 2 : 1;
      for (int pc = 0; pc < npc; ++pc) {
        const int kp = pc == 0 ? wid : PERS_NW;  // absolute 32-key pair block
        f32x4 s4[2];
        const unsigned padw[2] = {padl[pc][0], padl[pc][1]};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          s4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ldsK + ((2 * kp + j) * 16 + t) * 128 + kswz[kk]);
            s4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, ql[kk], s4[j], 0, 0, 0);
          }
          if constexpr (HAS_BIAS) s4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfl[pc], j ? sel_hi : sel_lo, s4[j], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            bool masked = (2 * kp + j) * 16 + g * 4 + r >= p.S;
            if constexpr (HAS_PAD) masked = masked || ((padw[j] >> (8 * r)) & 0xffu);
            s4[j][r] = masked ? -INFINITY : s4[j][r];
            mx = fmaxf(mx, s4[j][r]);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m2_new = fmaxf(m2, mx * c1);
        const float m2_use = (m2_new == -INFINITY) ? 0.f : m2_new;
        const float alpha = __builtin_amdgcn_exp2f(m2 - m2_use);
        m2 = m2_new;
        float pv[2][4];
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pv[j][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s4[j][r], c1, -m2_use));
            psum += pv[j][r];
          }
        lsum = lsum * alpha + psum;
        const bf16x8 pf = pack8(pv[0], pv[1]);
        s16x4 v0[4], v1[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          v0[db] = tr_read_a(ldsV + trsw[db] + (2 * kp) * 2048);
          v1[db] = tr_read_a(ldsV + trsw[db] + (2 * kp + 1) * 2048);
        }
        ATTN_WAIT_LGKM0();
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ol[db][r] *= alpha;
          ol[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_tr(v0[db], v1[db]), pf, ol[db], 0, 0, 0);
        }
      }
      lsum += __shfl_xor(lsum, 16);
      lsum += __shfl_xor(lsum, 32);
      float* sc = scratch + (buf * PERS_NW + wid) * PERS_SCR;
      if (t == 0) {
#pragma unroll
        for (int db = 0; db < 4; ++db) *reinterpret_cast<f32x4*>(sc + db * 16 + g * 4) = ol[db];
        if (g == 0) { sc[64] = m2; sc[65] = lsum; }
      }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float l = l_run[qb];
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      const int qrow = q0 + qb * 16 + t;
      if (qb < nact && qrow < p.S) {
        const float inv = 1.f / l;
        bf16_t* op = p.out + (row_base + qrow) * p.ldo + h * HD;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(ot[qb][db][r] * inv);
          *reinterpret_cast<bf16x4*>(op + db * 16 + g * 4) = o;
        }
        if (g == 0 && p.lse) p.lse[((int64_t)b * p.heads + h) * p.lse_ld + qrow] = m2_run[qb] * (1.0f / LOG2E) + logf(l);
      }
    }
    wait_all();  // the next item's K / V pieces of this wave, its Q fragments and first small loads (and this item's stores)
  }
  if (has_left && prev_item >= 0) {
    __syncthreads();
    if (wid == 0) merge_left(prev_item, buf ^ 1);
  }
}

// Fragment-major bias image (what attn_fwd_res_kernel's bias MFMA reads): for every 16-query block qb and 32-key pair-block
// kp one contiguous block of 64 lanes x 8 bf16; lane (g,t) holds bias[q = qb*16 + (g&1)*8 + i][key = kp*32 + (g>>1)*16 + t],
// i = 0..7, zero outside the sequence.  src: row-major image [n_img][S][Spad] (n_img = heads, or B * heads per-sample).
__global__ __launch_bounds__(256) void bias_pack_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int S, int Spad,
                                                        int nqb, int nkp, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (image, qb, kp, lane)
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  int64_t blk = idx >> 6;
  const int kp = (int)(blk % nkp);
  blk /= nkp;
  const int qb = (int)(blk % nqb);
  const int64_t img = blk / nqb;
  const int g = lane >> 4, t = lane & 15;
  const int key = kp * 32 + (g >> 1) * 16 + t;
  const int qbase = qb * 16 + (g & 1) * 8;
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = qbase + i;
    v[i] = (q < S && key < S) ? src[(img * S + q) * Spad + key] : (bf16_t)0.f;
  }
  *reinterpret_cast<bf16x8*>(dst + idx * 8) = v;
}

// =====================================================================================================================
// Backward.  With P = softmax(scale*QK^T + bias), D[q] = sum_d dO[q][d]*O[q][d]:
//   dP = dO V^T,  dS = P o (dP - D),  dV = P^T dO,  dK = scale * dS^T Q,  dQ = scale * dS K,  dBias = sum_b dS.
// Three kernels, each recomputing P from the saved log-sum-exp (no S x S tensor is ever stored):
//   attn_bwd_dkdv : workgroup owns 128 keys (4 waves x 32), loops over 64-query tiles; un-swapped S[q][key] so that
//                   P / dS are already in second-operand layout for the contractions over queries
//                   (dV^T = dO^T P, dK^T = Q^T dS; dO^T / Q^T fragments by ds_read_b64_tr_b16 from the same LDS tiles).
//   attn_bwd_dq   : workgroup owns 128 queries, loops over 64-key tiles; swapped S^T[key][q] as in the forward;
//                   dQ^T = K^T dS^T with K^T fragments by transpose-read.
//   attn_bwd_dbias: workgroup owns a (128-query, 64-key) tile of one head and loops over the samples of its batch
//                   chunk, accumulating dS in registers (the reference gets this from autograd's sum over the
//                   expanded batch dim, adapter/image.py:164-171); chunks are combined with fp32 atomics.
// =====================================================================================================================
struct AttnBwdArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; int64_t ld;
  const bf16_t* dout; int64_t ldo;   // [B*S][ldo]
  const bf16_t* bias;                // [heads][S][Spad]  (rows = query)
  const bf16_t* biasT;               // [heads][S][Spad]  (rows = key), same values transposed; columns >= S must be FINITE
  const bf16_t* bias_frag;           // fragment-major image (op_attn_bias_pack) or null
  int64_t bias_bs;                   // elements between the images of consecutive samples (0: shared by all samples)
  const uint8_t* key_pad;            // [B][Spad]
  const float* lse;                  // [B][heads][Spad]
  const float* delta;                // [B][heads][Spad]
  const bf16_t* out;                 // forward output rows [B*S][ldo], or null.  Given: the dQ kernels compute delta = rowsum(dO o O) per
  float* delta_w;                    // head themselves and store it here (= delta) for the dK/dV kernel, which then runs after them
  bf16_t* dq; bf16_t* dk; bf16_t* dv; int64_t ldg;  // gradient rows (same packing as q/k/v)
  float* dbias;                      // [heads][S][Spad] fp32, pre-zeroed
  int B, S, Spad, heads, bchunk;
  float scale;
  int lone_keys;                     // dK/dV: a trailing key block of <= 16 keys is split over the waves by queries (tune bit 11: off)
};

// D[b][h][q] = sum_d dO * O ; one thread per (row, head, 8 dims), 8-lane groups reduce
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out,
                                                         int64_t ldo, float* __restrict__ delta, int B, int S, int Spad,
                                                         int heads) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = heads * 8;
  const int64_t row = gid / per_row;
  const int rem = (int)(gid - row * per_row);
  float s = 0.f;
  if (row < (int64_t)B * S) {
    float a[8], b[8];
    Vec8<bf16_t>::load(dout + row * ldo + rem * 8, a);
    Vec8<bf16_t>::load(out + row * ldo + rem * 8, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j] * b[j];
  }
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 4);
  if (row < (int64_t)B * S && (rem & 7) == 0) {
    const int h = rem >> 3;
    const int64_t bb = row / S;
    const int qi = (int)(row - bb * S);
    delta[(bb * heads + h) * Spad + qi] = s;
  }
}

// delta[q] = sum_d dO[q][d] * O[q][d] of one head from the first-operand fragments of a 16-query block (lane (g, t): row t, dims
// kk*32 + g*8 .. +7): 16 products per lane, then the four g-groups.  Every lane of a row ends up with the row's sum.
__device__ __forceinline__ float delta_from_frags(const bf16x8 (&dO)[2], const bf16x8 (&O)[2]) {
  float s = 0.f;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int i = 0; i < 8; ++i) s = __builtin_fmaf((float)dO[kk][i], (float)O[kk][i], s);
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  return s;
}

// Softmax part rewritten in round 2 like the resident forward kernel (profiles/r2_experiments.md: these kernels are bound by
// their VALU stream and by 8-byte bias loads, not by the matrix pipe): the bias is added by the matrix pipe -- the transposed
// image's rows are already the second-operand fragment [32 queries x 16 keys] of  S += Selector_j(1/scale) . BiasT-fragment  --
// P = exp2(s * scale*log2e - lse*log2e) is one fma + one v_exp_f32, dS = P * (dP - delta) two more: 5 VALU instructions per
// score instead of 11.  Keys that do not exist or are padded need NO masking here: a key is a COLUMN of every product of this
// kernel, so whatever its P / dS columns hold only reaches its own dK / dV rows, which are written as zeros (padded keys) or
// not at all (keys >= S).  Query rows >= S of the last tile get lse = +inf (P = 0) and delta = 0.
// LONE [r4]: the workgroup's keys fit ONE 16-key block (the 257th token of the image stream: one key).  Until round 4 that
// workgroup ran the whole query loop on one wave with two key blocks (three idle waves, a dead block): S = 256 -> 257 cost +40 % on
// this kernel.  Now its four waves split every 64-query tile by 16-query blocks (wave w: half m = w >> 1, block j = w & 1; the other
// block of the half enters the dV / dK products as zeros) and their partial dK / dV are summed through LDS at the end.
template <bool HAS_BIAS, int KBW, bool LONE>  // KBW: 16-key blocks per wave (2: 128 keys per workgroup; 1: 64 -- see op_attn_bwd)
__device__ __forceinline__ void attn_bwd_dkdv_body(const AttnBwdArgs& p, char* smem, int bx, int h, int b) {
  constexpr int KB = LONE ? 1 : KBW;
  char* ldsQ = smem;
  char* ldsO = smem + 64 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, t = lane & 15;
  const int kbase = LONE ? bx * (64 * KBW) : bx * (64 * KBW) + wid * (16 * KBW);
  const bool wave_active = LONE || kbase < p.S;
  const int my_m = wid >> 1, my_j = wid & 1;  // (LONE) this wave's 16-query block of every tile
  const int64_t row_base = (int64_t)b * p.S;

  // K, V fragments (second operand): lane (g,t) <- X[kbase + kb*16 + t][kk*32 + g*8 ..]
  bf16x8 kf[KB][2], vf[KB][2];
  bool kdead[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int key = kbase + kb * 16 + t;
    const int kc = min(key, p.S - 1);
    const int64_t off = (row_base + kc) * p.ld + h * HD;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      kf[kb][kk] = *reinterpret_cast<const bf16x8*>(p.k + off + kk * 32 + g * 8);
      vf[kb][kk] = *reinterpret_cast<const bf16x8*>(p.v + off + kk * 32 + g * 8);
    }
    kdead[kb] = p.key_pad && key < p.S && p.key_pad[(int64_t)b * p.Spad + key];  // padded key: its dK / dV rows are zero
  }
  f32x4 dvT[KB][4], dkT[KB][4];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int db = 0; db < 4; ++db) { dvT[kb][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dkT[kb][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  u32x4 rq[2], ro[2];
  int st_row[2], st_c[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int c2 = tid + 256 * i; st_row[i] = c2 >> 3; st_c[i] = c2 & 7; }
  auto load_tile = [&](int q0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qr = min(q0 + st_row[i], p.S - 1);
      rq[i] = *reinterpret_cast<const u32x4*>(p.q + (row_base + qr) * p.ld + h * HD + st_c[i] * 8);
      ro[i] = *reinterpret_cast<const u32x4*>(p.dout + (row_base + qr) * p.ldo + h * HD + st_c[i] * 8);
    }
  };
  auto write_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = st_row[i] * 128 + ((st_c[i] ^ (st_row[i] & 7)) << 4);
      *reinterpret_cast<u32x4*>(ldsQ + off) = rq[i];
      *reinterpret_cast<u32x4*>(ldsO + off) = ro[i];
    }
  };
  int trsw[4];  // transpose-read piece offset for rowblk 0; rowblk adds 16 rows = 2048 bytes (row & 7 unchanged)
#pragma unroll
  for (int db = 0; db < 4; ++db) trsw[db] = tr_off_swz(0, db, g, t);

  const int ntiles = (p.S + 63) / 64;
  const float* lse_b = p.lse + ((int64_t)b * p.heads + h) * p.Spad;
  const float* del_b = p.delta + ((int64_t)b * p.heads + h) * p.Spad;
  // transposed bias image rows of this lane's two keys: 8 consecutive queries at g*8 = one second-operand fragment
  const bf16_t* brow[KB] = {};
  bf16x8 sel_lo, sel_hi;  // first operand that copies second-operand row j = t (j = 16 + t) into output row t, times 1/scale
  if constexpr (HAS_BIAS) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int key = min(kbase + kb * 16 + t, p.S - 1);
      brow[kb] = p.biasT + (int64_t)b * p.bias_bs + ((int64_t)h * p.S + key) * p.Spad + g * 8;
    }
    const bf16_t inv = (bf16_t)(1.0f / p.scale), zero = (bf16_t)0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sel_lo[i] = (g * 8 + i == t) ? inv : zero;
      sel_hi[i] = (g * 8 + i == 16 + t) ? inv : zero;
    }
  }
  const float c1 = p.scale * LOG2E;
  load_tile(0);
  for (int qt = 0; qt < ntiles; ++qt) {
    const int q0 = qt * 64;
    __syncthreads();
    write_tile();
    __syncthreads();
    if (!wave_active) {
      load_tile(q0 + 64);
      continue;
    }
    // lse / delta / bias fragments of both 32-query halves first, THEN the next tile's Q / dO prefetch: s_waitcnt vmcnt
    // counts in issue order, so small loads issued behind the prefetch would make their wait a wait for the prefetch.
    f32x4 l4[2][2], d4[2][2];
    bf16x8 bfr[2][KB];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int qrow = q0 + (2 * m + j) * 16 + g * 4;  // + r; < Spad
        l4[m][j] = *reinterpret_cast<const f32x4*>(lse_b + qrow);
        d4[m][j] = *reinterpret_cast<const f32x4*>(del_b + qrow);
      }
      if constexpr (HAS_BIAS) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) bfr[m][kb] = *reinterpret_cast<const bf16x8*>(brow[kb] + q0 + m * 32);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    load_tile(q0 + 64);  // unconditional (rows clamped to S-1): a branch here would force s_waitcnt vmcnt(0) below
    __builtin_amdgcn_sched_barrier(0);

    // per 32-query half m:  S[q][key] = Q K^T (+ bias / scale), dP[q][key] = dO V^T  (lane (g,t): q = qb*16 + g*4 + r, key = t),
    // then P, dS in place, then dV^T[d][key] += dO^T[d][q] P[q][key] and dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (q0 + m * 32 >= p.S) break;  // uniform: no valid query in this half of the tile
      if (LONE && (m != my_m || q0 + (2 * m + my_j) * 16 >= p.S)) continue;  // (per wave) not this wave's block / no valid query in it
      f32x4 s[2][KB], dp[2][KB];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) { s[j][kb] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j][kb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (LONE && j != my_j) continue;  // the other block stays zero: it adds nothing to dV / dK below
        if (q0 + (2 * m + j) * 16 >= p.S) continue;  // (uniform) no valid query in this block (the lone 257th query: j = 0 only)
        const int qb = 2 * m + j;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int off = (qb * 16 + t) * 128 + (((kk * 4 + g) ^ (t & 7)) << 4);
          const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(ldsQ + off);
          const bf16x8 ofr = *reinterpret_cast<const bf16x8*>(ldsO + off);
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            s[j][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[kb][kk], s[j][kb], 0, 0, 0);
            dp[j][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr, vf[kb][kk], dp[j][kb], 0, 0, 0);
          }
        }
        if constexpr (HAS_BIAS) {
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
            s[j][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(j ? sel_hi : sel_lo, bfr[m][kb], s[j][kb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if ((LONE && j != my_j) || q0 + (2 * m + j) * 16 >= p.S) continue;
        const int qrow = q0 + (2 * m + j) * 16 + g * 4;  // + r
        float l2[4], dl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // rows >= S: lse / delta are unspecified (possibly NaN / inf) -> P = 0, delta = 0
          const bool live = qrow + r < p.S;
          l2[r] = live ? l4[m][j][r] * LOG2E : INFINITY;
          dl[r] = live ? d4[m][j][r] : 0.f;
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][kb][r], c1, -l2[r]));
            s[j][kb][r] = pr;
            dp[j][kb][r] = pr * (dp[j][kb][r] - dl[r]);
          }
        }
      }
      bf16x8 pfr[KB], dsf[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a0[r] = s[0][kb][r]; a1[r] = s[1][kb][r];
          b0[r] = dp[0][kb][r]; b1[r] = dp[1][kb][r];
        }
        pfr[kb] = pack8(a0, a1);
        dsf[kb] = pack8(b0, b1);
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const bf16x8 oT = join_tr(tr_read(ldsO + trsw[db] + (2 * m) * 2048), tr_read(ldsO + trsw[db] + (2 * m + 1) * 2048));
        const bf16x8 qT = join_tr(tr_read(ldsQ + trsw[db] + (2 * m) * 2048), tr_read(ldsQ + trsw[db] + (2 * m + 1) * 2048));
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          dvT[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oT, pfr[kb], dvT[kb][db], 0, 0, 0);
          dkT[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT, dsf[kb], dkT[kb][db], 0, 0, 0);
        }
      }
    }
  }
  if (!wave_active) return;
  if constexpr (LONE) {  // sum the four waves' partial dV^T / dK^T (4 KiB per wave and matrix) into wave 0
    f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      __syncthreads();  // (first round: every wave is out of the tile loop; second: wave 0 has read the first round)
      if (wid > 0) {
#pragma unroll
        for (int db = 0; db < 4; ++db) red[((wid - 1) * 4 + db) * 64 + lane] = which ? dkT[0][db] : dvT[0][db];
      }
      __syncthreads();
      if (wid == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            if (which) dkT[0][db] += red[(w * 4 + db) * 64 + lane];
            else dvT[0][db] += red[(w * 4 + db) * 64 + lane];
          }
      }
    }
    if (wid > 0) return;
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int key = kbase + kb * 16 + t;
    if (key >= p.S) continue;
    bf16_t* kp = p.dk + (row_base + key) * p.ldg + h * HD;
    bf16_t* vp = p.dv + (row_base + key) * p.ldg + h * HD;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      bf16x4 a, c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[r] = kdead[kb] ? (bf16_t)0.f : (bf16_t)(dkT[kb][db][r] * p.scale);
        c[r] = kdead[kb] ? (bf16_t)0.f : (bf16_t)dvT[kb][db][r];
      }
      *reinterpret_cast<bf16x4*>(kp + db * 16 + g * 4) = a;
      *reinterpret_cast<bf16x4*>(vp + db * 16 + g * 4) = c;
    }
  }
}

template <bool HAS_BIAS, int KB>
__global__ __launch_bounds__(256, KB == 1 ? 3 : 2) void attn_bwd_dkdv_kernel(AttnBwdArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 128];
  int bx, h, b;
  xcd_work_item(bx, h, b);
  if (KB == 2 && bx > 0 && p.S - bx * 128 <= 16 && p.lone_keys) attn_bwd_dkdv_body<HAS_BIAS, KB, true>(p, smem, bx, h, b);  // (uniform)
  else attn_bwd_dkdv_body<HAS_BIAS, KB, false>(p, smem, bx, h, b);
}

// Shared by the dQ and dBias kernels: for one 64-key tile, lane (g,t) computes dS^T[key = kb*16 + g*4 + r][q = t]
// for its QB query blocks.  KF/VF: functors returning the first-operand fragment (K or V rows) for (kb, kk).
// HAS_BIAS is a template parameter and, with HOIST, the QB x 4 bias fragments and the key-pad words are loaded BEFORE the
// MFMAs (a run-time `if (p.bias)` per fragment compiled to serialised load -> s_waitcnt pairs after them); masked entries
// get -inf on the exponent's INPUT, so the exponentials are straight-line code (`dead ? 0 : exp(x)` became one exec-mask
// branch per element).
// `prefetch()` issues the caller's next-tile global loads; it runs AFTER the hoisted bias / pad loads (s_waitcnt vmcnt counts
// in issue order: a bias wait behind the prefetch would wait for the prefetch too).
template <int QB, bool HAS_BIAS, bool HOIST, typename KF, typename VF, typename BF, typename PF>
__device__ __forceinline__ void ds_tile(const AttnBwdArgs& p, int b, int h, int k0, int q0w, int g, int t,
                                        const bf16x8 (&qf)[QB][2], const bf16x8 (&of)[QB][2], const float (&lse)[QB],
                                        const float (&del)[QB], KF kfrag, VF vfrag, BF biasfrag, PF prefetch,
                                        f32x4 (&ds)[QB][4]) {
  unsigned padw[4] = {0u, 0u, 0u, 0u};
  auto load_pad = [&]() {
    if (p.key_pad) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        padw[kb] = *reinterpret_cast<const unsigned*>(p.key_pad + (int64_t)b * p.Spad + k0 + kb * 16 + g * 4);
    }
  };
  if constexpr (HOIST) load_pad();
  bf16x4 bv[QB][4];
  if constexpr (HAS_BIAS && HOIST) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int qi = min(q0w + qb * 16 + t, p.S - 1);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) bv[qb][kb] = biasfrag(qb, kb, qi, k0 + kb * 16 + g * 4);
    }
  }
  if constexpr (HOIST) __builtin_amdgcn_sched_barrier(0);  // keep the small loads ahead of the prefetch
  prefetch();
  if constexpr (HOIST) __builtin_amdgcn_sched_barrier(0);
  f32x4 st[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) { st[qb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f}; ds[qb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    if (k0 + kb * 16 >= p.S) break;  // uniform; those dS entries are forced to zero below
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 kfr = kfrag(kb, kk);
      const bf16x8 vfr = vfrag(kb, kk);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        st[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[qb][kk], st[qb][kb], 0, 0, 0);
        ds[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, of[qb][kk], ds[qb][kb], 0, 0, 0);
      }
    }
  }
  if constexpr (!HOIST) load_pad();  // (the kernels at the register limit keep the short live ranges)
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qraw = q0w + qb * 16 + t;
    const int qi = min(qraw, p.S - 1);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int key = k0 + kb * 16 + g * 4;
      if constexpr (HAS_BIAS && !HOIST) bv[qb][kb] = biasfrag(qb, kb, qi, key);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool dead = (key + r >= p.S) || ((padw[kb] >> (8 * r)) & 0xffu) || (qraw >= p.S);
        float x = st[qb][kb][r] * p.scale;
        if constexpr (HAS_BIAS) x += (float)bv[qb][kb][r];
        x -= lse[qb];
        float pr;
        if constexpr (HOIST) pr = __expf(dead ? -INFINITY : x);
        else pr = dead ? 0.f : __expf(x);
        const float dsv = pr * (ds[qb][kb][r] - del[qb]);
        ds[qb][kb][r] = dead ? 0.f : dsv;
      }
    }
  }
}

template <bool HAS_BIAS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnBwdArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 128];
  char* ldsK = smem;
  char* ldsV = smem + 64 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, t = lane & 15;
  int bx, h, b;
  xcd_work_item(bx, h, b);
  const int q0w = bx * BQ + wid * 32;
  const bool wave_active = q0w < p.S;
  const int64_t row_base = (int64_t)b * p.S;

  bf16x8 qf[2][2], of[2][2];
  float lse[2], del[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qi = min(q0w + qb * 16 + t, p.S - 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qf[qb][kk] = *reinterpret_cast<const bf16x8*>(p.q + (row_base + qi) * p.ld + h * HD + kk * 32 + g * 8);
      of[qb][kk] = *reinterpret_cast<const bf16x8*>(p.dout + (row_base + qi) * p.ldo + h * HD + kk * 32 + g * 8);
    }
    lse[qb] = p.lse[((int64_t)b * p.heads + h) * p.Spad + qi];
    if (p.out) {  // (uniform) delta from dO and O, stored for the dK/dV kernel
      bf16x8 oo[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) oo[kk] = *reinterpret_cast<const bf16x8*>(p.out + (row_base + qi) * p.ldo + h * HD + kk * 32 + g * 8);
      del[qb] = delta_from_frags(of[qb], oo);
      if (g == 0 && q0w + qb * 16 + t < p.S) p.delta_w[((int64_t)b * p.heads + h) * p.Spad + qi] = del[qb];
    } else {
      del[qb] = p.delta[((int64_t)b * p.heads + h) * p.Spad + qi];
    }
  }
  f32x4 dqT[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 4; ++db) dqT[qb][db] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 rk[2], rv[2];
  int st_row[2], st_c[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int c2 = tid + 256 * i; st_row[i] = c2 >> 3; st_c[i] = c2 & 7; }
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
      const int off = st_row[i] * 128 + ((st_c[i] ^ (st_row[i] & 7)) << 4);
      *reinterpret_cast<u32x4*>(ldsK + off) = rk[i];
      *reinterpret_cast<u32x4*>(ldsV + off) = rv[i];
    }
  };
  int trsw[4];  // transpose-read piece offset for rowblk 0; rowblk adds 16 rows = 2048 bytes (row & 7 unchanged)
#pragma unroll
  for (int db = 0; db < 4; ++db) trsw[db] = tr_off_swz(0, db, g, t);

  const int ntiles = (p.S + BKV - 1) / BKV;
  load_tile(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * BKV;
    __syncthreads();
    write_tile();
    __syncthreads();
    if (!wave_active) {
      load_tile(k0 + BKV);
      continue;
    }
    auto prefetch = [&]() { load_tile(k0 + BKV); };  // unconditional: rows are clamped to S-1 past the last tile
    f32x4 ds[2][4];
    auto kfrag = [&](int kb, int kk) {
      return *reinterpret_cast<const bf16x8*>(ldsK + (kb * 16 + t) * 128 + (((kk * 4 + g) ^ (t & 7)) << 4));
    };
    auto vfrag = [&](int kb, int kk) {
      return *reinterpret_cast<const bf16x8*>(ldsV + (kb * 16 + t) * 128 + (((kk * 4 + g) ^ (t & 7)) << 4));
    };
    auto biasfrag = [&](int, int, int qi, int key) {
      return *reinterpret_cast<const bf16x4*>(p.bias + (int64_t)b * p.bias_bs + ((int64_t)h * p.S + qi) * p.Spad + key);
    };
    ds_tile<2, HAS_BIAS, true>(p, b, h, k0, q0w, g, t, qf, of, lse, del, kfrag, vfrag, biasfrag, prefetch, ds);
    // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (k0 + m * 32 >= p.S) break;
      bf16x8 dsf[2];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float a0[4], a1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { a0[r] = ds[qb][2 * m][r]; a1[r] = ds[qb][2 * m + 1][r]; }
        dsf[qb] = pack8(a0, a1);
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const bf16x8 kT = join_tr(tr_read(ldsK + trsw[db] + (2 * m) * 2048), tr_read(ldsK + trsw[db] + (2 * m + 1) * 2048));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          dqT[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT, dsf[qb], dqT[qb][db], 0, 0, 0);
      }
    }
  }
  if (!wave_active) return;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qi = q0w + qb * 16 + t;
    if (qi >= p.S) continue;
    bf16_t* qp = p.dq + (row_base + qi) * p.ldg + h * HD;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      bf16x4 a;
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = (bf16_t)(dqT[qb][db][r] * p.scale);
      *reinterpret_cast<bf16x4*>(qp + db * 16 + g * 4) = a;
    }
  }
}

// dQ and dBias in ONE pass (sequences of up to NT * 64 keys, NT <= 6): grid (q tiles of 64 = 4 waves x 16 queries, heads,
// batch chunks).  A workgroup walks the samples of its chunk; for each sample it streams the K/V tiles exactly like
// attn_bwd_dq_kernel, and the dS tiles it computes anyway for dQ are ALSO summed over the samples in registers
// (NT x 4 accumulators per lane) and added to dbias once at the end -- the separate dBias kernel recomputed S and dP
// (two of the five backward matmuls) just for that sum.  One 16-query block per wave keeps the NT x 64 dS accumulators
// within the register budget; 64-row query tiles also waste less on S = 257 (5 tiles = 320 rows instead of 3 x 128).
template <int NT, bool FRAG, int LASTB>  // LASTB: 16-key blocks of the last key tile that can hold keys (1 or 4): S = 257 is 4 tiles + ONE
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_dbias_kernel(AttnBwdArgs p) {  // block, whose 3 absent neighbours cost no accumulator
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 128];
  char* ldsK = smem;
  char* ldsV = smem + 64 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, t = lane & 15;
  int bx, h, chunk;
  xcd_work_item(bx, h, chunk);
  const int q0w = bx * 64 + wid * 16;
  const bool wave_active = q0w < p.S;
  const int qi = min(q0w + t, p.S - 1);

  f32x4 acc[NT][4];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) acc[kt][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 rk[2], rv[2];
  int st_row[2], st_c[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int c2 = tid + 256 * i; st_row[i] = c2 >> 3; st_c[i] = c2 & 7; }
  auto load_tile = [&](int b, int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kr = min(k0 + st_row[i], p.S - 1);
      const int64_t off = ((int64_t)b * p.S + kr) * p.ld + h * HD + st_c[i] * 8;
      rk[i] = *reinterpret_cast<const u32x4*>(p.k + off);
      rv[i] = *reinterpret_cast<const u32x4*>(p.v + off);
    }
  };
  auto write_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = st_row[i] * 128 + ((st_c[i] ^ (st_row[i] & 7)) << 4);
      *reinterpret_cast<u32x4*>(ldsK + off) = rk[i];
      *reinterpret_cast<u32x4*>(ldsV + off) = rv[i];
    }
  };
  bf16x8 qn[1][2], on[1][2], oon[2];
  float lsen[1], deln[1];
  const bool own_delta = p.out != nullptr;  // delta from dO and O here, stored for the dK/dV kernel (which then runs after this one)
  auto load_q = [&](int b) {
    const int64_t row = (int64_t)b * p.S + qi;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qn[0][kk] = *reinterpret_cast<const bf16x8*>(p.q + row * p.ld + h * HD + kk * 32 + g * 8);
      on[0][kk] = *reinterpret_cast<const bf16x8*>(p.dout + row * p.ldo + h * HD + kk * 32 + g * 8);
      if (own_delta) oon[kk] = *reinterpret_cast<const bf16x8*>(p.out + row * p.ldo + h * HD + kk * 32 + g * 8);
    }
    lsen[0] = p.lse[((int64_t)b * p.heads + h) * p.Spad + qi];
    if (!own_delta) deln[0] = p.delta[((int64_t)b * p.heads + h) * p.Spad + qi];
  };
  int trsw[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) trsw[db] = tr_off_swz(0, db, g, t);

  // fragment-major bias image of this wave's query block, selectors of the bias MFMA (see attn_fwd_res_kernel)
  const int nqb = (p.S + 15) >> 4, nkp = (p.S + 31) >> 5;
  const int64_t per_head = (int64_t)nqb * nkp * FRAG_BLOCK;
  const bf16_t* fbase = nullptr;
  bf16x8 sel_lo, sel_hi;
  const float c1 = p.scale * LOG2E;
  if constexpr (FRAG) {
    fbase = p.bias_frag + (int64_t)h * per_head + (int64_t)(min(q0w, p.S - 1) >> 4) * nkp * FRAG_BLOCK + lane * 8;
    const bf16_t inv = (bf16_t)(1.0f / p.scale), zero = (bf16_t)0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sel_lo[i] = (g * 8 + i == t) ? inv : zero;
      sel_hi[i] = (g * 8 + i == 16 + t) ? inv : zero;
    }
  }
  const int b_begin = chunk * p.bchunk, b_end = min(p.B, (chunk + 1) * p.bchunk);
  if (b_begin >= b_end) return;
  load_tile(b_begin, 0);
  load_q(b_begin);
  for (int b = b_begin; b < b_end; ++b) {
    bf16x8 qf[1][2], of[1][2];
    float lse[1], del[1];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) { qf[0][kk] = qn[0][kk]; of[0][kk] = on[0][kk]; }
    lse[0] = lsen[0];
    if (own_delta) {
      del[0] = delta_from_frags(of[0], oon);
      if (g == 0 && q0w + t < p.S) p.delta_w[((int64_t)b * p.heads + h) * p.Spad + qi] = del[0];
    } else {
      del[0] = deln[0];
    }
    if (b + 1 < b_end) load_q(b + 1);
    // The bias fragments do not depend on the sample; left alone the compiler hoists all NT x 4 of them out of this loop
    // (8 VGPRs per key tile) and spills accumulators instead.  They are L2-resident: reload per sample.
    int bias_resample = 0;
    asm volatile("" : "+s"(bias_resample));
    f32x4 dqT[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dqT[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // NOT unrolled: an unrolled body costs ~35 VGPRs per key tile on top of the 16 accumulators; the accumulator set is
    // picked by a uniform compare chain instead.
#pragma unroll 1
    for (int kt = 0; kt < NT; ++kt) {
      const int k0 = kt * BKV;
      __syncthreads();
      write_tile();
      __syncthreads();
      // next tile of this sample, or the first tile of the next one; unconditional (clamped) so that the compiler can count
      // the outstanding loads instead of waiting for vmcnt(0)
      const bool last_kt = kt + 1 >= NT;
      const int nb = last_kt ? min(b + 1, b_end - 1) : b, nk0 = last_kt ? 0 : k0 + BKV;
      auto prefetch = [&]() { load_tile(nb, nk0); };
      if (!wave_active) {
        prefetch();
        continue;
      }
      f32x4 ds[1][4];
      auto kfrag = [&](int kb, int kk) {
        return *reinterpret_cast<const bf16x8*>(ldsK + (kb * 16 + t) * 128 + (((kk * 4 + g) ^ (t & 7)) << 4));
      };
      auto vfrag = [&](int kb, int kk) {
        return *reinterpret_cast<const bf16x8*>(ldsV + (kb * 16 + t) * 128 + (((kk * 4 + g) ^ (t & 7)) << 4));
      };
      auto biasfrag = [&](int, int, int qrow, int key) {
        return *reinterpret_cast<const bf16x4*>(p.bias + bias_resample + (int64_t)b * p.bias_bs + ((int64_t)h * p.S + qrow) * p.Spad + key);
      };
      if constexpr (FRAG) {
        // Round-2 softmax part (see attn_fwd_res_kernel): bias through the matrix pipe from the fragment-major image (ONE
        // coalesced 16-byte load per lane per 32 keys), P = exp2(s * scale*log2e - lse*log2e) as one fma + one v_exp_f32,
        // dS = P * (dP - delta); only keys are masked (they are the contraction index of dQ), and only in tiles that can
        // hold a dead key (the last tile, or any tile of a padded sample).  A dead QUERY is a column of everything here:
        // it needs no masking, its dQ row / dBias row is simply not written.
        bf16x8 bfr[2];
        unsigned padw[4] = {0u, 0u, 0u, 0u};
        const bf16_t* fb = fbase + (p.bias_bs != 0 ? (int64_t)b * p.heads * per_head : 0) + bias_resample;
#pragma unroll
        for (int m = 0; m < 2; ++m) bfr[m] = *reinterpret_cast<const bf16x8*>(fb + (int64_t)min((k0 >> 5) + m, nkp - 1) * FRAG_BLOCK);
        if (p.key_pad) {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
            padw[kb] = *reinterpret_cast<const unsigned*>(p.key_pad + (int64_t)b * p.Spad + k0 + kb * 16 + g * 4);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the small loads ahead of the prefetch (vmcnt counts in issue order)
        prefetch();
        __builtin_amdgcn_sched_barrier(0);
        f32x4 st[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { st[kb] = (f32x4){0.f, 0.f, 0.f, 0.f}; ds[0][kb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          if (k0 + kb * 16 >= p.S) break;  // uniform; those dS entries stay exactly zero
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag(kb, kk), qf[0][kk], st[kb], 0, 0, 0);
            ds[0][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag(kb, kk), of[0][kk], ds[0][kb], 0, 0, 0);
          }
          st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kb >> 1], (kb & 1) ? sel_hi : sel_lo, st[kb], 0, 0, 0);
        }
        const float nl2 = -lse[0] * LOG2E, dl = del[0];
        const bool may_mask = (k0 + BKV > p.S) || (p.key_pad != nullptr);  // uniform
        if (may_mask) {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            const int key = k0 + kb * 16 + g * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool dead = (key + r >= p.S) || ((padw[kb] >> (8 * r)) & 0xffu);
              const float pr = __builtin_amdgcn_exp2f(dead ? -INFINITY : __builtin_fmaf(st[kb][r], c1, nl2));
              ds[0][kb][r] = pr * (ds[0][kb][r] - dl);  // dP and delta of a live query are finite: 0 * x = 0
            }
          }
        } else {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              ds[0][kb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][r], c1, nl2)) * (ds[0][kb][r] - dl);
        }
      } else {
        ds_tile<1, true, (NT < 6)>(p, b, h, k0, q0w, g, t, qf, of, lse, del, kfrag, vfrag, biasfrag, prefetch, ds);
      }
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        if (kt == c) {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
            if (c < NT - 1 || kb < LASTB) acc[c][kb] += ds[0][kb];
        }
      }
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (k0 + m * 32 >= p.S) break;
        float a0[4], a1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { a0[r] = ds[0][2 * m][r]; a1[r] = ds[0][2 * m + 1][r]; }
        const bf16x8 dsf = pack8(a0, a1);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 kT = join_tr(tr_read(ldsK + trsw[db] + (2 * m) * 2048), tr_read(ldsK + trsw[db] + (2 * m + 1) * 2048));
          dqT[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT, dsf, dqT[db], 0, 0, 0);
        }
      }
    }
    if (wave_active && q0w + t < p.S) {
      bf16_t* qp = p.dq + ((int64_t)b * p.S + q0w + t) * p.ldg + h * HD;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        bf16x4 a;
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = (bf16_t)(dqT[db][r] * p.scale);
        *reinterpret_cast<bf16x4*>(qp + db * 16 + g * 4) = a;
      }
    }
  }
  if (!wave_active || q0w + t >= p.S) return;
  // slab `chunk` of dbias belongs to this batch chunk alone: plain read-modify-write (fp32 atomics from several chunks on
  // one slab cost more than the whole rest of this kernel).  Four loads in flight per key tile, then four stores.
  float* drow = p.dbias + (((int64_t)chunk * p.heads + h) * p.S + q0w + t) * p.Spad + g * 4;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    f32x4 cur[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
      if ((kt < NT - 1 || kb < LASTB) && kt * BKV + kb * 16 < p.Spad) cur[kb] = *reinterpret_cast<const f32x4*>(drow + kt * BKV + kb * 16);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
      if ((kt < NT - 1 || kb < LASTB) && kt * BKV + kb * 16 < p.Spad)
        *reinterpret_cast<f32x4*>(drow + kt * BKV + kb * 16) = cur[kb] + acc[kt][kb];
  }
}

// =====================================================================================================================
// Persistent dQ + dBias kernel for the 193 ... 257-token streams with a shared bias image (round 4; the forward's skeleton).
// attn_bwd_dq_dbias_kernel above streams every K / V tile through registers into LDS with two barriers per tile, 64 queries per
// workgroup (K / V of a (sample, head) are staged five times at S = 257), one 16-query block per wave at two waves per SIMD: its
// waves wait 54 % of their cycles.  Here a workgroup of 8 waves owns ONE (head, half of the query blocks) for a chunk of the batch:
// wave w keeps the dBias accumulators of query block half * 8 + w over all keys (68 registers at S = 257) while it walks the
// samples; K / V of sample b + 1 arrive by LDS-DMA in the other half of a double buffer while sample b is computed (one barrier
// per sample), K^T fragments of dQ^T += K^T dS^T come from the same resident K by transpose reads; Q / dO / O fragments, lse of
// the next sample and the bias / pad words of the next tile travel as inline-asm loads with hand-counted waits (see
// attn_fwd_pers_kernel).  delta = rowsum(dO o O) is computed here and stored for the dK/dV kernel.  S = 257: the lone query is
// spread over the LAST half's workgroup by keys -- wave w: key pair w (wave 0 also pair 8) -- its dQ partials are summed through
// LDS after the next sample's barrier, its dBias row lives in two more accumulators per wave.
// =====================================================================================================================
template <bool HAS_BIAS, bool HAS_PAD, int NT, int LASTB, bool LEFT>  // LEFT: S = 257, the workgroups of the last query half
__device__ __forceinline__ void attn_bwd_dq_pers_body(const AttnBwdArgs& p, char* smem, int rows_pad, int nqh, int qhalf0) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, t = lane & 15;
  const int KVB = 2 * rows_pad * 128;
  float* scratch = reinterpret_cast<float*>(smem + 2 * KVB);  // [2][PERS_NW][64]: dQ partials of the lone query
  const int nqb = (p.S + 15) >> 4, nkp = (p.S + 31) >> 5;
  // grid: (chunk, head, query half)
  const int qhalf = qhalf0, h = (blockIdx.x / nqh) % p.heads, chunk = blockIdx.x / (nqh * p.heads);
  const int b_begin = chunk * p.bchunk, b_end = min(p.B, b_begin + p.bchunk);
  if (b_begin >= b_end) return;
  const int qblk = qhalf * PERS_NW + wid;
  const bool active = qblk < min(nqb, 2 * PERS_NW);
  constexpr bool has_left = LEFT;  // S = 257: this workgroup also owns the lone query (row 256)
  const int q0w = min(qblk, nqb - 1) * 16;
  const int qi = min(q0w + t, p.S - 1);
  const float c1 = p.scale * LOG2E;
  const int64_t per_head = (int64_t)nqb * nkp * FRAG_BLOCK;

  bf16x8 sel_lo, sel_hi;
  {
    const bf16_t inv = (bf16_t)(1.0f / p.scale), zero = (bf16_t)0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sel_lo[i] = (g * 8 + i == t) ? inv : zero;
      sel_hi[i] = (g * 8 + i == 16 + t) ? inv : zero;
    }
  }
  int trsw[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) trsw[db] = tr_off_swz(0, db, g, t);
  const int kswz[2] = {((0 * 4 + g) ^ (t & 7)) << 4, ((1 * 4 + g) ^ (t & 7)) << 4};

  const int ngrp = rows_pad >> 3;
  const int r_in = lane >> 3, slot = lane & 7;
  const int npiece = (2 * ngrp - wid + PERS_NW - 1) / PERS_NW;
  auto piece = [&](int b, int buf, int j) {
    const int grp = wid + j * PERS_NW;
    const bool isv = grp >= ngrp;
    const int gi = isv ? grp - ngrp : grp;
    // The lane's source offset is recomputed per piece from a laundered input (six VALU instructions): as a loop invariant the
    // compiler keeps one 64-bit address per piece alive over the whole kernel -- 18 registers this kernel does not have (they were
    // spilled, and every reload next to the fetch is an s_waitcnt vmcnt(0)).  Uniform base + 32-bit lane offset: the saddr form.
    int rl = r_in;
    asm volatile("" : "+v"(rl));
    const int r = gi * 8 + rl;
    const int kr = min(r, p.S - 1);
    const unsigned voff = (unsigned)((kr * (int)p.ld + ((slot ^ (r & 7)) << 3)) * 2);
    const char* sbase = (const char*)((isv ? p.v : p.k) + (int64_t)b * p.S * p.ld + h * HD);
    char* dst = smem + buf * KVB + (isv ? rows_pad * 128 : 0) + gi * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + voff),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  // ---- in-flight (inline-asm) loads: next tile's bias fragments / pad words; next sample's query-side operands ----
  constexpr int NSMALL = (HAS_BIAS ? 2 : 0) + (HAS_PAD ? 4 : 0);
  // fetch pieces a wave issues in key tile kt: all of the next sample's <= 9 in the first three tiles -- they have landed long before
  // the barrier at the top of the next sample (two per tile up to the last one left the last two in flight there: exposed latency)
  constexpr int BPPT = 3;
  auto ppt_of = [](int kt) { return kt < 3 ? BPPT : 0; };
  // wave-uniform bases in SGPRs + one 32-bit lane offset per kind of row (see gload16_s)
  const bf16_t* fwave = HAS_BIAS ? p.bias_frag + (int64_t)h * per_head + (int64_t)(q0w >> 4) * nkp * FRAG_BLOCK : nullptr;  // this wave's query block
  const bf16_t* fleft = HAS_BIAS ? p.bias_frag + (int64_t)h * per_head + (int64_t)(nqb - 1) * nkp * FRAG_BLOCK : nullptr;  // the lone query's block
  const unsigned vo_frag = lane * 16, vo_pad = g * 4;
  const unsigned vo_q = (unsigned)(((int64_t)qi * p.ld + h * HD + g * 8) * 2), vo_o = (unsigned)(((int64_t)qi * p.ldo + h * HD + g * 8) * 2);
  const unsigned vo_ql = (unsigned)(((int64_t)(p.S - 1) * p.ld + h * HD + g * 8) * 2), vo_ol = (unsigned)(((int64_t)(p.S - 1) * p.ldo + h * HD + g * 8) * 2);
  const unsigned vo_lse = qi * 4, vo_lsel = (p.S - 1) * 4;
  bf16x8 bn[2];
  unsigned padn[4] = {0u, 0u, 0u, 0u};
  auto small_loads = [&](int b, int kt) {
    if constexpr (HAS_BIAS) {
      gload16_s<0>(bn[0], fwave + min(2 * kt, nkp - 1) * FRAG_BLOCK, vo_frag);
      gload16_s<0>(bn[1], fwave + min(2 * kt + 1, nkp - 1) * FRAG_BLOCK, vo_frag);
    }
    if constexpr (HAS_PAD) {
      const uint8_t* pr = p.key_pad + (int64_t)b * p.Spad + kt * BKV;
      gload4_s<0>(padn[0], pr, vo_pad);
      gload4_s<16>(padn[1], pr, vo_pad);
      gload4_s<32>(padn[2], pr, vo_pad);
      gload4_s<48>(padn[3], pr, vo_pad);
    }
  };
  bf16x8 qn[2], don[2], oon[2], ql[2], dol[2], ool[2], bfl[2];
  float lsen, lsel;
  unsigned padl[2][2] = {{0u, 0u}, {0u, 0u}};
  auto query_loads = [&](int b, unsigned voq, unsigned voo, unsigned vol, bf16x8 (&q)[2], bf16x8 (&d)[2], bf16x8 (&o)[2], float& l) {
    const bf16_t* sq = p.q + (int64_t)b * p.S * p.ld;
    const bf16_t* sd = p.dout + (int64_t)b * p.S * p.ldo;
    const bf16_t* so = p.out + (int64_t)b * p.S * p.ldo;
    gload16_s<0>(q[0], sq, voq);
    gload16_s<64>(q[1], sq, voq);
    gload16_s<0>(d[0], sd, voo);
    gload16_s<64>(d[1], sd, voo);
    gload16_s<0>(o[0], so, voo);
    gload16_s<64>(o[1], so, voo);
    unsigned lw;
    gload4_s<0>(lw, p.lse + ((int64_t)b * p.heads + h) * p.Spad, vol);
    l = __builtin_bit_cast(float, lw);
  };
  // LEFT: the next sample's operands are requested later (behind the stores of this sample: registers) -- see the sample loop
  auto late_loads = [&](int b_next, int b) {
    (void)b_next;
    if (has_left) {
      query_loads(b, vo_ql, vo_ol, vo_lsel, ql, dol, ool, lsel);
      if constexpr (HAS_BIAS) {
        gload16_s<0>(bfl[0], fleft + wid * FRAG_BLOCK, vo_frag);
        gload16_s<0>(bfl[1], fleft + min(PERS_NW, nkp - 1) * FRAG_BLOCK, vo_frag);
      }
      if constexpr (HAS_PAD) {
        const uint8_t* pr = p.key_pad + (int64_t)b * p.Spad;
        gload4_s<0>(padl[0][0], pr + 2 * wid * 16, vo_pad);
        gload4_s<16>(padl[0][1], pr + 2 * wid * 16, vo_pad);
        gload4_s<0>(padl[1][0], pr + min(2 * PERS_NW * 16, p.Spad - 32), vo_pad);
        gload4_s<16>(padl[1][1], pr + min(2 * PERS_NW * 16, p.Spad - 32), vo_pad);
      }
    }
  };

  // dBias accumulators of this wave's query block over all keys, and of the lone query over this wave's key pair(s)
  f32x4 acc[NT][4], accl[2][2];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) acc[kt][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) accl[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int j = 0; j < npiece; ++j) piece(b_begin, 0, j);
  query_loads(b_begin, vo_q, vo_o, vo_lse, qn, don, oon, lsen);
  small_loads(b_begin, 0);
  int buf = 0, prev_b = -1;
  auto merge_left = [&](int b, int sb) {  // sum of the eight dQ partials of the lone query; lane = head dimension
    const float* sc = scratch + sb * (PERS_NW * 64);
    float o = 0.f;
#pragma unroll
    for (int w = 0; w < PERS_NW; ++w) o += sc[w * 64 + lane];
    p.dq[((int64_t)b * p.S + p.S - 1) * p.ldg + h * HD + lane] = (bf16_t)(o * p.scale);
  };

  // Everything a sample finds in flight at its top has landed at the END of the previous trip (or here): a register an inline-asm
  // load is still writing must not be live across the loop's back edge -- the register allocator puts copies at a loop header when it
  // likes, and it did (copies of the in-flight Q / dO / O fragments in front of the wait: stale operands, NaN).  Every such register
  // is an operand of the wait statement, so nothing that reads one can move above it either.
  auto wait_all = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qn[0]), "+v"(qn[1]), "+v"(don[0]), "+v"(don[1]), "+v"(oon[0]), "+v"(oon[1]), "+v"(lsen), "+v"(bn[0]), "+v"(bn[1]),
                 "+v"(padn[0]), "+v"(padn[1]), "+v"(padn[2]), "+v"(padn[3]) : : "memory");
  };
  wait_all();
  for (int b = b_begin; b < b_end; ++b, buf ^= 1) {
    __syncthreads();
    const bf16x8 qf[2] = {qn[0], qn[1]}, dof[2] = {don[0], don[1]};
    const float dl = delta_from_frags(dof, oon);
    const float nl2 = -lsen * LOG2E;
    if (active && g == 0 && q0w + t < p.S) p.delta_w[((int64_t)b * p.heads + h) * p.Spad + q0w + t] = dl;
    const int nxt = b + 1 < b_end ? b + 1 : b;  // (no next sample: this one is fetched again into the idle buffer -- nothing branches)
    if (has_left && prev_b >= 0 && wid == (prev_b & (PERS_NW - 1))) merge_left(prev_b, buf ^ 1);
    prev_b = b;
    const char* ldsK = smem + buf * KVB;
    const char* ldsV = ldsK + rows_pad * 128;
    f32x4 dqT[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dqT[db] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // One 64-key tile, two 16-key blocks at a time (S^T, dP^T -> P, dS -> dBias accumulators -> dQ^T; 16 score registers instead of
    // 32: with the 68 dBias accumulators this kernel lives at the register limit).  The in-flight small loads of the NEXT tile are
    // requested after this tile's last use of the current ones (bias: the S^T MFMAs; pad words: the dS step) into the SAME registers,
    // followed by this tile's share of the next sample's fetch; the hand-counted wait at the top of a tile therefore allows exactly
    // those PERS_PPT pieces to be outstanding.
    auto tile = [&](const int kt, auto last_c) {
      constexpr bool LAST = decltype(last_c)::value;
      const int k0 = kt * BKV;
      // ---- in-flight loads: wait for this tile's small loads (behind them only the previous tile's fetch pieces were issued), move
      // them out of the landing registers, request the next tile's (LEFT: the same registers are re-used later instead, see below) ----
      if (kt > 0) {
        if (ppt_of(kt - 1) > 0) {
          if constexpr (HAS_BIAS && HAS_PAD)
            asm volatile("s_waitcnt vmcnt(%6)" : "+v"(bn[0]), "+v"(bn[1]), "+v"(padn[0]), "+v"(padn[1]), "+v"(padn[2]), "+v"(padn[3]) : "n"(BPPT));
          else if constexpr (HAS_BIAS)
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bn[0]), "+v"(bn[1]) : "n"(BPPT));
          else if constexpr (HAS_PAD)
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(padn[0]), "+v"(padn[1]), "+v"(padn[2]), "+v"(padn[3]) : "n"(BPPT));
        } else {
          if constexpr (HAS_BIAS && HAS_PAD)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(bn[0]), "+v"(bn[1]), "+v"(padn[0]), "+v"(padn[1]), "+v"(padn[2]), "+v"(padn[3]));
          else if constexpr (HAS_BIAS)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(bn[0]), "+v"(bn[1]));
          else if constexpr (HAS_PAD)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(padn[0]), "+v"(padn[1]), "+v"(padn[2]), "+v"(padn[3]));
        }
      }
      bf16x8 bf[2];
      unsigned padw[4] = {0u, 0u, 0u, 0u};
      if constexpr (HAS_BIAS) { bf[0] = bn[0]; bf[1] = bn[1]; }
      if constexpr (HAS_PAD) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) padw[kb] = padn[kb];
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!LAST) {
        if constexpr (!LEFT) {  // the next sample's operands a whole tile (+ the short last one) ahead; BEFORE the small loads, whose
          if (kt == NT - 2) query_loads(nxt, vo_q, vo_o, vo_lse, qn, don, oon, lsen);  // wait then does not include them
        }
        small_loads(b, kt + 1);
      } else {
        late_loads(nxt, b);
        small_loads(nxt, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 0; jj < BPPT; ++jj)
        if (ppt_of(kt) > 0) piece(nxt, buf ^ 1, min(kt * BPPT + jj, npiece - 1));
      __builtin_amdgcn_sched_barrier(0);
      const char* kt_ = ldsK + k0 * 128;
      const char* vt = ldsV + k0 * 128;
      constexpr int NKB = LAST ? LASTB : 4;  // 16-key blocks of this tile that can hold keys
      constexpr int NM = (NKB + 1) / 2;
      if (!active) return;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        f32x4 st[2], ds[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int kb = 2 * m + j;
          st[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          ds[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (kb >= NKB) continue;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt_ + (kb * 16 + t) * 128 + kswz[kk]);
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vt + (kb * 16 + t) * 128 + kswz[kk]);
            st[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], st[j], 0, 0, 0);
            ds[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[kk], ds[j], 0, 0, 0);
          }
          if constexpr (HAS_BIAS) st[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[m], j ? sel_hi : sel_lo, st[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int kb = 2 * m + j;
          if (kb >= NKB) continue;
          const int key = k0 + kb * 16 + g * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = __builtin_fmaf(st[j][r], c1, nl2);
            if constexpr (LAST || HAS_PAD) {
              bool dead = false;
              if constexpr (LAST) dead = key + r >= p.S;
              if constexpr (HAS_PAD) dead = dead || ((padw[kb] >> (8 * r)) & 0xffu);
              x = dead ? -INFINITY : x;
            }
            ds[j][r] = __builtin_amdgcn_exp2f(x) * (ds[j][r] - dl);  // dP and delta of a live query are finite: 0 * x = 0
          }
        }
        if constexpr (HAS_BIAS) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (2 * m + j < NKB) acc[kt][2 * m + j] += ds[j];
        }
        float a0[4], a1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { a0[r] = ds[0][r]; a1[r] = ds[1][r]; }
        const bf16x8 dsf = pack8(a0, a1);
        s16x4 k0r[4], k1r[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          k0r[db] = tr_read_a(kt_ + trsw[db] + (2 * m) * 2048);
          k1r[db] = tr_read_a(kt_ + trsw[db] + (2 * m + 1) * 2048);
        }
        ATTN_WAIT_LGKM0();
#pragma unroll
        for (int db = 0; db < 4; ++db) dqT[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_tr(k0r[db], k1r[db]), dsf, dqT[db], 0, 0, 0);
      }
    };
    // (unrolled: with a run-time tile index the compiler turns the compare chain that picks the accumulator set into an indexed
    // access and moves the accumulators to scratch memory; every load in the tile is asm or LDS, fenced: nothing gets hoisted)
#pragma unroll
    for (int kt = 0; kt < NT - 1; ++kt) tile(kt, std::false_type{});
    tile(NT - 1, std::true_type{});

    if (active && q0w + t < p.S) {
      bf16_t* qp = p.dq + ((int64_t)b * p.S + q0w + t) * p.ldg + h * HD;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        bf16x4 a;
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = (bf16_t)(dqT[db][r] * p.scale);
        *reinterpret_cast<bf16x4*>(qp + db * 16 + g * 4) = a;
      }
    }
    if (has_left) {
      // The next sample's query-side operands: in this instantiation requested HERE, where the accumulators of dQ and this sample's
      // fragments are dead (in the last tile, next to the lone query's operands, they did not fit: 33 spilled registers); the lone
      // query's phase below and the barrier cover part of their latency.
      query_loads(nxt, vo_q, vo_o, vo_lse, qn, don, oon, lsen);
      // the late loads of the last tile must have landed: behind them came that tile's small loads (no fetch pieces: NT >= 4), the
      // seven requests above (and this wave's stores, if it had any: with them the wait only asks for more than it needs -- it never
      // under-waits)
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSMALL + 7) : "memory");
      asm volatile("" : "+v"(ql[0]), "+v"(ql[1]), "+v"(dol[0]), "+v"(dol[1]), "+v"(ool[0]), "+v"(ool[1]), "+v"(bfl[0]), "+v"(bfl[1]), "+v"(lsel),
                   "+v"(padl[0][0]), "+v"(padl[0][1]), "+v"(padl[1][0]), "+v"(padl[1][1]));
      const float dll = delta_from_frags(dol, ool);
      const float nl2l = -lsel * LOG2E;
      if (wid == 0 && lane == 0) p.delta_w[((int64_t)b * p.heads + h) * p.Spad + p.S - 1] = dll;
      f32x4 dql[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) dql[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {  // (unrolled: pc indexes register arrays)
        if (pc == 1 && wid != 0) break;
        const int kp = pc == 0 ? wid : PERS_NW;  // absolute 32-key pair block
        f32x4 s4[2], d4[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          s4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          d4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ldsK + ((2 * kp + j) * 16 + t) * 128 + kswz[kk]);
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(ldsV + ((2 * kp + j) * 16 + t) * 128 + kswz[kk]);
            s4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, ql[kk], s4[j], 0, 0, 0);
            d4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dol[kk], d4[j], 0, 0, 0);
          }
          if constexpr (HAS_BIAS) s4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfl[pc], j ? sel_hi : sel_lo, s4[j], 0, 0, 0);
        }
        float a0[4], a1[4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            bool dead = (2 * kp + j) * 16 + g * 4 + r >= p.S;
            if constexpr (HAS_PAD) dead = dead || ((padl[pc][j] >> (8 * r)) & 0xffu);
            const float x = dead ? -INFINITY : __builtin_fmaf(s4[j][r], c1, nl2l);
            const float v = __builtin_amdgcn_exp2f(x) * (d4[j][r] - dll);
            d4[j][r] = v;
            (j ? a1 : a0)[r] = v;
          }
        if constexpr (HAS_BIAS) { accl[pc][0] += d4[0]; accl[pc][1] += d4[1]; }
        const bf16x8 dsf = pack8(a0, a1);
        s16x4 k0r[4], k1r[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          k0r[db] = tr_read_a(ldsK + trsw[db] + (2 * kp) * 2048);
          k1r[db] = tr_read_a(ldsK + trsw[db] + (2 * kp + 1) * 2048);
        }
        ATTN_WAIT_LGKM0();
#pragma unroll
        for (int db = 0; db < 4; ++db) dql[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_tr(k0r[db], k1r[db]), dsf, dql[db], 0, 0, 0);
      }
      float* sc = scratch + (buf * PERS_NW + wid) * 64;
      if (t == 0) {
#pragma unroll
        for (int db = 0; db < 4; ++db) *reinterpret_cast<f32x4*>(sc + db * 16 + g * 4) = dql[db];
      }
    }

    wait_all();
  }
  if (has_left && prev_b >= 0) {
    __syncthreads();
    if (wid == 0) merge_left(prev_b, buf ^ 1);
  }
  if constexpr (HAS_BIAS) {
    if (p.dbias == nullptr) return;
    // slab `chunk` of dbias belongs to this batch chunk alone: plain read-modify-write (see attn_bwd_dq_dbias_kernel)
    if (active && q0w + t < p.S) {
      float* drow = p.dbias + (((int64_t)chunk * p.heads + h) * p.S + q0w + t) * p.Spad + g * 4;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        f32x4 cur[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          if ((kt < NT - 1 || kb < LASTB) && kt * BKV + kb * 16 < p.Spad) cur[kb] = *reinterpret_cast<const f32x4*>(drow + kt * BKV + kb * 16);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          if ((kt < NT - 1 || kb < LASTB) && kt * BKV + kb * 16 < p.Spad)
            *reinterpret_cast<f32x4*>(drow + kt * BKV + kb * 16) = cur[kb] + acc[kt][kb];
      }
    }
    if (has_left && t == 0) {  // the lone query's row: this wave's key pair(s); every query column of the fragment held that row
      float* drow = p.dbias + (((int64_t)chunk * p.heads + h) * p.S + p.S - 1) * p.Spad + g * 4;
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        if (pc == 1 && wid != 0) break;
        const int kp = pc == 0 ? wid : PERS_NW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int key0 = (2 * kp + j) * 16;
          if (key0 < p.Spad) {
            float* d = drow + key0;
            *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(d) + accl[pc][j];
          }
        }
      }
    }
  }
}

template <bool HAS_BIAS, bool HAS_PAD, int NT, int LASTB>
__global__ __launch_bounds__(PERS_NW * 64, 2) void attn_bwd_dq_pers_kernel(AttnBwdArgs p, int rows_pad, int nqh) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int qhalf = blockIdx.x % nqh;
  const bool lone = ((p.S + 15) >> 4) > 2 * PERS_NW && qhalf == nqh - 1;  // (uniform) S = 257: the last half's workgroups own the lone query
  if constexpr (NT == 5) {
    if (lone) {
      attn_bwd_dq_pers_body<HAS_BIAS, HAS_PAD, NT, LASTB, true>(p, smem, rows_pad, nqh, qhalf);
      return;
    }
  }
  attn_bwd_dq_pers_body<HAS_BIAS, HAS_PAD, NT, LASTB, false>(p, smem, rows_pad, nqh, qhalf);
}

// =====================================================================================================================
// Persistent dK / dV kernel for the 193 ... 257-token streams with a shared bias image (round 4; the counterpart of
// attn_bwd_dq_pers_kernel).  attn_bwd_dkdv_kernel stages every 64-query tile of Q / dO through registers into LDS with two barriers per
// tile and does so in each of the 2 - 3 key workgroups of a (sample, head); its waves wait 35 % of their cycles.  Here ONE workgroup
// of 8 waves per CU walks the items (sample, head) = blockIdx, blockIdx + gridDim, ...: wave w owns keys 32 w ... 32 w + 31 (K / V
// fragments and the dK^T / dV^T accumulators in registers); ALL of Q and dO of an item, its lse and delta rows (and, S = 257, the K / V
// row of key 256) sit in LDS -- fetched by LDS-DMA into the other half of a double buffer while the previous item is computed: one
// barrier per item.  K / V fragments, key-padding bytes and the first bias fragments of the next item and the transposed-bias
// fragments of the next query tile travel as inline-asm loads with hand-counted waits (protocol and hazards: attn_fwd_pers_kernel).
// S = 257: the lone KEY is not a ninth wave -- wave w runs it against the 32-query half w (wave 0 also against the lone query), the
// eight partial dK / dV rows meet in LDS and are summed by one wave after the next item's barrier.  The per-tile arithmetic is
// attn_bwd_dkdv_kernel's: same bits for keys 0 ... 255, the lone key's sums are taken in another order (fp32).
// =====================================================================================================================
constexpr int PERS_SCRL = 136;  // floats per (item parity, wave) of the lone key's scratch: dV[64], dK[64], dead flag, pad

template <int OFF>
__device__ __forceinline__ void gload1_s(unsigned& d, const void* sbase, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_ubyte %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF));
}

// 8-byte buffer store (raw descriptor, per-lane byte offset; a lane whose offset lies behind the descriptor's size is DROPPED by the
// hardware, but the instruction is issued and counted by vmcnt all the same: the persistent kernels need an exact count of the stores
// behind their last loads).  Wait states as in gemm.hip: store_b128_padded.
__device__ __forceinline__ u32x4 attn_raw_rsrc(const void* base, int nbytes) {
  const uint64_t a = (uint64_t)base;
  return (u32x4){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) & 0xffffu,
                 (unsigned)__builtin_amdgcn_readfirstlane(nbytes), 0x00020000u};
}
__device__ __forceinline__ void bstore8(const bf16x4& data, const u32x4& rsrc, unsigned voff) {
  asm volatile("s_nop 4\n\tbuffer_store_dwordx2 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(data), "v"(voff), "s"(rsrc) : "memory");
}

template <bool HAS_BIAS, bool HAS_PAD, int NT, bool LONE>  // NT: 64-query tiles (S = 257: 5, the fifth holds one query); LONE: S = 257
__global__ __launch_bounds__(PERS_NW * 64, 2) void attn_bwd_dkdv_pers_kernel(AttnBwdArgs p, int rows_pad, int nitems) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, t = lane & 15;
  const int QBY = rows_pad * 128;         // bytes of the Q (or dO) rows of one buffer
  const int BUFB = 2 * QBY + 6144;        // Q | dO | lse (2 KiB) | delta (2 KiB) | lone K row x 8 (1 KiB) | lone V row x 8 (1 KiB)
  float* scratch = reinterpret_cast<float*>(smem + 2 * BUFB);  // [2][PERS_NW][PERS_SCRL]
  const int kbase = wid * 32;
  const bool wave_active = kbase < min(p.S, 256);
  const float c1 = p.scale * LOG2E;

  bf16x8 sel_lo, sel_hi;
  {
    const bf16_t inv = (bf16_t)(1.0f / p.scale), zero = (bf16_t)0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sel_lo[i] = (g * 8 + i == t) ? inv : zero;
      sel_hi[i] = (g * 8 + i == 16 + t) ? inv : zero;
    }
  }
  int trsw[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) trsw[db] = tr_off_swz(0, db, g, t);
  const int kswz[2] = {((0 * 4 + g) ^ (t & 7)) << 4, ((1 * 4 + g) ^ (t & 7)) << 4};

  // ---- LDS-DMA of one item: groups of 1 KiB (8 rows x 128 B of Q, then of dO; two chunks of the lse row, two of the delta row, the
  // lone key's K row and V row eight times each) dealt to the waves ----
  const int ngrp = rows_pad >> 3;
  const int NG = 2 * ngrp + (LONE ? 6 : 4);
  const int r_in = lane >> 3, slot = lane & 7;
  const int npiece = (NG - wid + PERS_NW - 1) / PERS_NW;
  auto piece = [&](int it, int buf, int j) {
    const int b = it / p.heads, h = it - b * p.heads;
    const int grp = wid + j * PERS_NW;
    int rl = r_in;
    asm volatile("" : "+v"(rl));  // (laundered: see attn_bwd_dq_pers_body -- no per-piece address kept alive over the kernel)
    const char* sbase;
    unsigned voff;
    int dst_off;
    if (grp < 2 * ngrp) {
      const bool isd = grp >= ngrp;
      const int gi = isd ? grp - ngrp : grp;
      const int r = gi * 8 + rl;
      const int qr = min(r, p.S - 1);
      const int ldx = isd ? (int)p.ldo : (int)p.ld;
      voff = (unsigned)((qr * ldx + ((slot ^ (r & 7)) << 3)) * 2);
      sbase = isd ? (const char*)(p.dout + (int64_t)b * p.S * p.ldo + h * HD) : (const char*)(p.q + (int64_t)b * p.S * p.ld + h * HD);
      dst_off = (isd ? QBY : 0) + gi * 1024;
    } else {
      const int e = grp - 2 * ngrp;  // 0, 1: lse chunks; 2, 3: delta chunks; 4: lone K row; 5: lone V row
      if (e < 4) {
        voff = (unsigned)min((e & 1) * 1024 + (rl * 8 + slot) * 16, p.Spad * 4 - 16);
        sbase = (const char*)((e < 2 ? p.lse : p.delta) + ((int64_t)b * p.heads + h) * p.Spad);
      } else {
        voff = (unsigned)(((p.S - 1) * (int)p.ld + slot * 8) * 2);
        sbase = (const char*)((e == 4 ? p.k : p.v) + (int64_t)b * p.S * p.ld + h * HD);
      }
      dst_off = 2 * QBY + e * 1024;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + voff),
                                     (__attribute__((address_space(3))) void*)(smem + buf * BUFB + dst_off), 16, 0, 0);
  };

  // ---- in-flight (inline-asm) loads ----
  // K / V fragments of this wave's keys (second operand: lane (g, t) <- X[kbase + kb * 16 + t][kk * 32 + g * 8 ..]) and their pad bytes
  const int key_kb[2] = {min(kbase + t, p.S - 1), min(kbase + 16 + t, p.S - 1)};
  const unsigned vo_k[2] = {(unsigned)((key_kb[0] * (int)p.ld + g * 8) * 2), (unsigned)((key_kb[1] * (int)p.ld + g * 8) * 2)};
  // LONE: requested straight into the fragment registers once the last tile is done with them (the lone key's phase and the stores cover
  // the latency: a second set of 32 landing registers does not fit next to that phase); otherwise into landing registers in the last tile
  bf16x8 kf[2][2], vf[2][2], kn[LONE ? 1 : 2][2], vn[LONE ? 1 : 2][2];
  unsigned kdn[2] = {0u, 0u}, kdl = 0u;
  auto kv_loads = [&](int it, bf16x8 (&kd)[2][2], bf16x8 (&vd)[2][2]) {
    const int b = it / p.heads, h = it - b * p.heads;
    const bf16_t* sk = p.k + (int64_t)b * p.S * p.ld + h * HD;
    const bf16_t* sv = p.v + (int64_t)b * p.S * p.ld + h * HD;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      gload16_s<0>(kd[kb][0], sk, vo_k[kb]);
      gload16_s<64>(kd[kb][1], sk, vo_k[kb]);
      gload16_s<0>(vd[kb][0], sv, vo_k[kb]);
      gload16_s<64>(vd[kb][1], sv, vo_k[kb]);
    }
    if constexpr (HAS_PAD) {
      const uint8_t* pr = p.key_pad + (int64_t)b * p.Spad;
      gload1_s<0>(kdn[0], pr, (unsigned)key_kb[0]);
      gload1_s<0>(kdn[1], pr, (unsigned)key_kb[1]);
      if constexpr (LONE) gload1_s<0>(kdl, pr, (unsigned)(p.S - 1));
    }
  };
  constexpr int NKV = 8 + (HAS_PAD ? (LONE ? 3 : 2) : 0);
  // transposed bias image rows of this lane's two keys: 8 consecutive queries at g * 8 = one second-operand fragment per 32 queries
  const unsigned vo_b[2] = {(unsigned)((key_kb[0] * p.Spad + g * 8) * 2), (unsigned)((key_kb[1] * p.Spad + g * 8) * 2)};
  const unsigned vo_bl = (unsigned)(((p.S - 1) * p.Spad + g * 8) * 2);  // the lone key's row
  bf16x8 bn[2][2], bfl[2];
  auto bias_loads = [&](int it, auto qt_c, auto m_c) {  // the fragments of half m of query tile QT (two key blocks: two loads)
    constexpr int QT = decltype(qt_c)::value, M = decltype(m_c)::value;
    if constexpr (HAS_BIAS) {
      const int h = it % p.heads;
      const bf16_t* sb = p.biasT + (int64_t)h * p.S * p.Spad;
      gload16_s<(QT * 64 + M * 32) * 2>(bn[M][0], sb, vo_b[0]);
      gload16_s<(QT * 64 + M * 32) * 2>(bn[M][1], sb, vo_b[1]);
    }
  };
  constexpr int NBH = HAS_BIAS ? 2 : 0;  // operations of one bias_loads
  auto lone_bias_loads = [&](int it) {  // the lone key against this wave's 32-query half (and, wave 0, against the lone query's block)
    if constexpr (HAS_BIAS && LONE) {
      const int h = it % p.heads;
      const bf16_t* sb = p.biasT + (int64_t)h * p.S * p.Spad;
      gload16_s<0>(bfl[0], sb + wid * 32, vo_bl);
      gload16_s<512>(bfl[1], sb, vo_bl);
    }
  };
  constexpr int NLB = (HAS_BIAS && LONE) ? 2 : 0;
  constexpr int BPPT = NT == 5 ? 3 : 4;  // fetch pieces a wave issues in each query tile but the last (<= 10 pieces)

  int item = blockIdx.x;
  const int step = gridDim.x;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if (item < nitems) {
    for (int j = 0; j < npiece; ++j) piece(item, 0, j);
    if constexpr (LONE) kv_loads(item, kf, vf);
    else kv_loads(item, kn, vn);
    bias_loads(item, I0{}, I0{});
    bias_loads(item, I0{}, I1{});
  }
  // everything an item finds in flight at its top has landed at the END of the previous trip (or here): all loads, i.e. everything but
  // the NST stores issued behind them (vmcnt completes in issue order)
#define DKDV_WAIT_LOADS(NST)                                                                                                                  \
  do {                                                                                                                                     \
    if constexpr (LONE)                                                                                                                    \
      asm volatile("s_waitcnt vmcnt(%15)" : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[1][0]), "+v"(kf[1][1]), "+v"(vf[0][0]), "+v"(vf[0][1]),   \
                   "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(bn[0][0]), "+v"(bn[0][1]), "+v"(bn[1][0]), "+v"(bn[1][1]), "+v"(kdn[0]), "+v"(kdn[1]), \
                   "+v"(kdl) : "n"(NST) : "memory");                                                                                      \
    else                                                                                                                                   \
      asm volatile("s_waitcnt vmcnt(%15)" : "+v"(kn[0][0]), "+v"(kn[0][1]), "+v"(kn[1][0]), "+v"(kn[1][1]), "+v"(vn[0][0]), "+v"(vn[0][1]),   \
                   "+v"(vn[1][0]), "+v"(vn[1][1]), "+v"(bn[0][0]), "+v"(bn[0][1]), "+v"(bn[1][0]), "+v"(bn[1][1]), "+v"(kdn[0]), "+v"(kdn[1]), \
                   "+v"(kdl) : "n"(NST) : "memory");                                                                                      \
  } while (0)
  DKDV_WAIT_LOADS(0);
  int buf = 0, prev_item = -1;
  auto merge_lone = [&](int it, int sb) {  // dK / dV rows of key S - 1: the sum of the eight partials; lane = head dimension
    const float* sc = scratch + sb * (PERS_NW * PERS_SCRL);
    float dv = 0.f, dk = 0.f;
#pragma unroll
    for (int w = 0; w < PERS_NW; ++w) { dv += sc[w * PERS_SCRL + lane]; dk += sc[w * PERS_SCRL + 64 + lane]; }
    const bool dead = sc[128] != 0.f;
    const int b = it / p.heads, h = it - b * p.heads;
    const int64_t row = ((int64_t)b * p.S + p.S - 1) * p.ldg + h * HD + lane;
    p.dv[row] = dead ? (bf16_t)0.f : (bf16_t)dv;
    p.dk[row] = dead ? (bf16_t)0.f : (bf16_t)(dk * p.scale);
  };

  for (; item < nitems; item += step, buf ^= 1) {
    const int b = item / p.heads, h = item - b * p.heads;
    // this item's rows have landed (every wave waited for its own pieces at the end of the previous trip); every wave is done with the
    // other buffer and with the scratch half it re-uses.  A raw barrier: __syncthreads() would wait for vmcnt(0), i.e. for the dK / dV
    // stores of the previous item, which are left to drain under this item's first tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (!LONE) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { kf[kb][kk] = kn[kb][kk]; vf[kb][kk] = vn[kb][kk]; }
    }
    const bool kdead[2] = {HAS_PAD && kdn[0] != 0u, HAS_PAD && kdn[1] != 0u};
    const bool ldead = HAS_PAD && LONE && kdl != 0u;
    const int nxt = item + step < nitems ? item + step : item;  // (no next item: this one is fetched again into the idle buffer)
    if (LONE && prev_item >= 0 && wid == ((prev_item / step) & (PERS_NW - 1))) merge_lone(prev_item, buf ^ 1);
    prev_item = item;
    const char* ldsQ = smem + buf * BUFB;
    const char* ldsO = ldsQ + QBY;
    const float* ldsL = reinterpret_cast<const float*>(ldsQ + 2 * QBY);
    const float* ldsD = ldsL + 512;

    f32x4 dvT[2][4], dkT[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int db = 0; db < 4; ++db) { dvT[kb][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dkT[kb][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // one 32-query half against NKB 16-key blocks whose K / V fragments are kx / vx: S, dP -> P, dS -> dV^T, dK^T (attn_bwd_dkdv_kernel)
    auto half = [&](const int q0h, auto nkb_c, const bf16x8 (&kx)[2][2], const bf16x8 (&vx)[2][2], const bf16x8 (&bx)[2],
                    f32x4 (&dvx)[2][4], f32x4 (&dkx)[2][4]) {
      constexpr int NKB = decltype(nkb_c)::value;
      f32x4 s[2][NKB], dp[2][NKB];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) { s[j][kb] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j][kb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      const bool second = q0h + 16 < p.S;  // (uniform) the half's second 16-query block holds a query
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j == 1 && !second) continue;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int off = (q0h + j * 16 + t) * 128 + kswz[kk];
          const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(ldsQ + off);
          const bf16x8 ofr = *reinterpret_cast<const bf16x8*>(ldsO + off);
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb) {
            s[j][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kx[kb][kk], s[j][kb], 0, 0, 0);
            dp[j][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr, vx[kb][kk], dp[j][kb], 0, 0, 0);
          }
        }
        if constexpr (HAS_BIAS) {
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb) s[j][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(j ? sel_hi : sel_lo, bx[kb], s[j][kb], 0, 0, 0);
        }
      }
      // the transposed dO / Q fragments of the first two head-dimension blocks: requested BEFORE the softmax arithmetic, which covers
      // their LDS latency (as in attn_fwd_pers_kernel)
      const s16x4 zero4 = {0, 0, 0, 0};
      s16x4 o0[2], o1[2], q0r[2], q1r[2];
      auto tr_request = [&](int dh) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int db = 2 * dh + i;
          o0[i] = tr_read_a(ldsO + q0h * 128 + trsw[db]);
          q0r[i] = tr_read_a(ldsQ + q0h * 128 + trsw[db]);
          o1[i] = zero4;
          q1r[i] = zero4;
          if (second) {
            o1[i] = tr_read_a(ldsO + q0h * 128 + trsw[db] + 2048);
            q1r[i] = tr_read_a(ldsQ + q0h * 128 + trsw[db] + 2048);
          }
        }
      };
      float l2[2][4], dl[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j == 1 && !second) continue;
        const int qrow = q0h + j * 16 + g * 4;  // + r
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(ldsL + qrow);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(ldsD + qrow);
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // rows >= S: lse / delta are unspecified (possibly NaN / inf) -> P = 0, delta = 0
          const bool live = qrow + r < p.S;
          l2[j][r] = live ? l4[r] * LOG2E : INFINITY;
          dl[j][r] = live ? d4[r] : 0.f;
        }
      }
      // (behind the compiler's own LDS reads above: its lgkmcnt bookkeeping does not see the asm reads, and a wait it places for one of
      // its loads behind them would wait for them too)
      __builtin_amdgcn_sched_barrier(0);
      tr_request(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j == 1 && !second) continue;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][kb][r], c1, -l2[j][r]));
            s[j][kb][r] = pr;
            dp[j][kb][r] = pr * (dp[j][kb][r] - dl[j][r]);
          }
        }
      }
      bf16x8 pfr[NKB], dsf[NKB];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a0[r] = s[0][kb][r]; a1[r] = s[1][kb][r];
          b0[r] = dp[0][kb][r]; b1[r] = dp[1][kb][r];
        }
        pfr[kb] = pack8(a0, a1);
        dsf[kb] = pack8(b0, b1);
      }
#pragma unroll
      for (int dh = 0; dh < 2; ++dh) {  // two head-dimension blocks at a time: 8 transpose reads in flight
        if (dh == 1) tr_request(1);
        ATTN_WAIT_LGKM0();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int db = 2 * dh + i;
          const bf16x8 oT = join_tr(o0[i], o1[i]), qT = join_tr(q0r[i], q1r[i]);
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb) {
            dvx[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oT, pfr[kb], dvx[kb][db], 0, 0, 0);
            dkx[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT, dsf[kb], dkx[kb][db], 0, 0, 0);
          }
        }
      }
    };

    // Query tile QT.  Transposed-bias fragments travel one HALF ahead in the registers the MFMAs read them from (no copies: the
    // kernel lives at the register limit): half 0 of tile QT + 1 is requested when half 0 of tile QT is done, half 1 when half 1 is
    // done; the counted waits allow exactly what was requested behind the awaited pair (fetch pieces, the other half's pair).
    auto tile = [&](auto qt_c) {
      constexpr int QT = decltype(qt_c)::value;
      constexpr bool LAST = QT == NT - 1;
      using QN = std::integral_constant<int, LAST ? 0 : QT + 1>;
      if constexpr (!LAST) {
#pragma unroll
        for (int jj = 0; jj < BPPT; ++jj) piece(nxt, buf ^ 1, min(QT * BPPT + jj, npiece - 1));  // (past the last piece: that piece again)
      }
      __builtin_amdgcn_sched_barrier(0);
      if (wave_active) half(QT * 64, std::integral_constant<int, 2>{}, kf, vf, bn[0], dvT, dkT);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!LAST) bias_loads(item, QN{}, I0{});
      // half 1's fragments of THIS tile (requested at the end of the previous tile; tile 0: landed with the item)
      if constexpr (HAS_BIAS && QT > 0) {
        if constexpr (!LAST) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bn[1][0]), "+v"(bn[1][1]) : "n"(BPPT + NBH));
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(bn[1][0]), "+v"(bn[1][1]));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (wave_active && QT * 64 + 32 < p.S) half(QT * 64 + 32, std::integral_constant<int, 2>{}, kf, vf, bn[1], dvT, dkT);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!LAST) {
        bias_loads(item, QN{}, I1{});
        if constexpr (HAS_BIAS) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bn[0][0]), "+v"(bn[0][1]) : "n"(NBH));  // half 0 of the next tile
      } else {  // late loads: the lone key's bias fragments first, then the next item's first bias fragments (and, !LONE, its K / V)
        lone_bias_loads(item);
        bias_loads(nxt, I0{}, I0{});
        bias_loads(nxt, I0{}, I1{});
        if constexpr (!LONE) kv_loads(nxt, kn, vn);
      }
    };
    // (unrolled by hand: the tile index is a compile-time constant -- immediate offsets of the bias loads, counted waits)
    tile(std::integral_constant<int, 0>{});
    tile(std::integral_constant<int, 1>{});
    tile(std::integral_constant<int, 2>{});
    tile(std::integral_constant<int, 3>{});
    if constexpr (NT == 5) tile(std::integral_constant<int, 4>{});

    if constexpr (LONE) {  // key S - 1 = 256: this wave's 32-query half (wave 0 also the lone query's block); all 16 key columns of the
                           // fragments are that key (its row is replicated in LDS), column t = 0 is the one that is kept
      // the next item's K / V fragments: the tiles are done with the registers; the lone key's bias fragments were requested first in the
      // last tile: behind them the next item's first bias fragments and these
      kv_loads(nxt, kf, vf);
      if constexpr (NLB > 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bfl[0]), "+v"(bfl[1]) : "n"(NKV + 2 * NBH));
      const char* ldsLK = ldsQ + 2 * QBY + 4096;
      bf16x8 kl[2][2], vl[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        kl[0][kk] = *reinterpret_cast<const bf16x8*>(ldsLK + (kk * 4 + g) * 16);
        vl[0][kk] = *reinterpret_cast<const bf16x8*>(ldsLK + 1024 + (kk * 4 + g) * 16);
        kl[1][kk] = kl[0][kk];
        vl[1][kk] = vl[0][kk];
      }
      f32x4 dvl[2][4], dkl[2][4];
#pragma unroll
      for (int db = 0; db < 4; ++db) { dvl[0][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dkl[0][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      const bf16x8 bl0[2] = {bfl[0], bfl[0]}, bl1[2] = {bfl[1], bfl[1]};
      half(wid * 32, std::integral_constant<int, 1>{}, kl, vl, bl0, dvl, dkl);
      if (wid == 0) half(256, std::integral_constant<int, 1>{}, kl, vl, bl1, dvl, dkl);
      float* sc = scratch + (buf * PERS_NW + wid) * PERS_SCRL;
      if (t == 0) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          *reinterpret_cast<f32x4*>(sc + db * 16 + g * 4) = dvl[0][db];
          *reinterpret_cast<f32x4*>(sc + 64 + db * 16 + g * 4) = dkl[0][db];
        }
      }
      if (wid == 0 && lane == 0) sc[128] = ldead ? 1.f : 0.f;
    }

    {  // 16 buffer stores per wave, none of them behind a branch: lanes whose key does not exist -- all lanes of a wave without keys --
       // are dropped by the descriptor, and ONE wait statement on one path follows (in a two-armed version the compiler put the next
       // item's copies of the landing registers in front of one arm's wait: tools/check_mfma_hazards.py)
      const int nrec = ((min(p.S, 256) - 1) * (int)p.ldg + HD) * 2;
      const u32x4 rk = attn_raw_rsrc(p.dk + (int64_t)b * p.S * p.ldg + h * HD, nrec);
      const u32x4 rv = attn_raw_rsrc(p.dv + (int64_t)b * p.S * p.ldg + h * HD, nrec);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const unsigned voff = (unsigned)(((kbase + kb * 16 + t) * (int)p.ldg + g * 4) * 2);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          bf16x4 a, c;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            a[r] = kdead[kb] ? (bf16_t)0.f : (bf16_t)(dkT[kb][db][r] * p.scale);
            c[r] = kdead[kb] ? (bf16_t)0.f : (bf16_t)dvT[kb][db][r];
          }
          bstore8(a, rk, voff + db * 32);
          bstore8(c, rv, voff + db * 32);
        }
      }
      DKDV_WAIT_LOADS(16);
    }
  }
  if (LONE && prev_item >= 0) {
    __syncthreads();
    if (wid == 0) merge_lone(prev_item, buf ^ 1);
  }
}

#undef DKDV_WAIT_LOADS

// grid (q tiles of 128, key tiles of 64, heads * batch chunks).  The bias fragment of the (head, q tile, key tile) is the
// same for every sample of the chunk (per-sample bias images: chunks of ONE sample, gradient slab b for sample b -- the path of
// masked pretraining with more kept tokens than the merged kernel's 384) and is loaded once; each sample's K/V tile goes through LDS once for all four waves (next
// sample's tile in flight during the MFMAs); dS is accumulated in registers over the chunk, chunks combine by fp32 atomics.
__global__ __launch_bounds__(256, 2) void attn_bwd_dbias_kernel(AttnBwdArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 128];
  char* ldsK = smem;
  char* ldsV = smem + 64 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, t = lane & 15;
  const int h = blockIdx.z % p.heads, chunk = blockIdx.z / p.heads;
  const int q0w = blockIdx.x * BQ + wid * 32;
  const int k0 = blockIdx.y * BKV;
  const bool wave_active = q0w < p.S;
  f32x4 acc[2][4];
  bf16x4 breg[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qi = min(q0w + qb * 16 + t, p.S - 1);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      acc[qb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.bias)  // (per-sample images: this workgroup's chunk is ONE sample, launch condition)
        breg[qb][kb] = *reinterpret_cast<const bf16x4*>(p.bias + (int64_t)(chunk * p.bchunk) * p.bias_bs + ((int64_t)h * p.S + qi) * p.Spad +
                                                        k0 + kb * 16 + g * 4);
    }
  }
  u32x4 rk[2], rv[2];
  int st_row[2], st_c[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int c2 = tid + 256 * i; st_row[i] = c2 >> 3; st_c[i] = c2 & 7; }
  auto load_tile = [&](int b) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kr = min(k0 + st_row[i], p.S - 1);
      const int64_t off = ((int64_t)b * p.S + kr) * p.ld + h * HD + st_c[i] * 8;
      rk[i] = *reinterpret_cast<const u32x4*>(p.k + off);
      rv[i] = *reinterpret_cast<const u32x4*>(p.v + off);
    }
  };
  auto write_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = st_row[i] * 128 + ((st_c[i] ^ (st_row[i] & 7)) << 4);
      *reinterpret_cast<u32x4*>(ldsK + off) = rk[i];
      *reinterpret_cast<u32x4*>(ldsV + off) = rv[i];
    }
  };
  const int b_begin = chunk * p.bchunk, b_end = min(p.B, (chunk + 1) * p.bchunk);
  // per-sample query-side operands (Q and dO fragments, lse, delta) are prefetched one sample ahead, like the K/V tile
  bf16x8 qn[2][2], on[2][2];
  float lsen[2], deln[2];
  auto load_q = [&](int b) {
    const int64_t row_base = (int64_t)b * p.S;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qi = min(q0w + qb * 16 + t, p.S - 1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        qn[qb][kk] = *reinterpret_cast<const bf16x8*>(p.q + (row_base + qi) * p.ld + h * HD + kk * 32 + g * 8);
        on[qb][kk] = *reinterpret_cast<const bf16x8*>(p.dout + (row_base + qi) * p.ldo + h * HD + kk * 32 + g * 8);
      }
      lsen[qb] = p.lse[((int64_t)b * p.heads + h) * p.Spad + qi];
      deln[qb] = p.delta[((int64_t)b * p.heads + h) * p.Spad + qi];
    }
  };
  if (b_begin < b_end) {
    load_tile(b_begin);
    if (wave_active) load_q(b_begin);
  }
  for (int b = b_begin; b < b_end; ++b) {
    __syncthreads();
    write_tile();
    __syncthreads();
    if (b + 1 < b_end) load_tile(b + 1);
    if (!wave_active) continue;
    bf16x8 qf[2][2], of[2][2];
    float lse[2], del[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) { qf[qb][kk] = qn[qb][kk]; of[qb][kk] = on[qb][kk]; }
      lse[qb] = lsen[qb];
      del[qb] = deln[qb];
    }
    if (b + 1 < b_end) load_q(b + 1);
    auto kfrag = [&](int kb, int kk) {
      return *reinterpret_cast<const bf16x8*>(ldsK + (kb * 16 + t) * 128 + (((kk * 4 + g) ^ (t & 7)) << 4));
    };
    auto vfrag = [&](int kb, int kk) {
      return *reinterpret_cast<const bf16x8*>(ldsV + (kb * 16 + t) * 128 + (((kk * 4 + g) ^ (t & 7)) << 4));
    };
    auto biasfrag = [&](int qb, int kb, int, int) { return breg[qb][kb]; };
    f32x4 ds[2][4];
    ds_tile<2, true, false>(p, b, h, k0, q0w, g, t, qf, of, lse, del, kfrag, vfrag, biasfrag, [] {}, ds);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[qb][kb][r] += ds[qb][kb][r];
  }
  if (!wave_active) return;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qi = q0w + qb * 16 + t;
    if (qi >= p.S) continue;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      // shared image: every chunk adds into the one slab; per-sample images: slab b of sample b, written by this workgroup alone
      float* dst = p.dbias + ((p.bias_bs != 0 ? (int64_t)chunk * p.heads : 0) + h) * ((int64_t)p.S * p.Spad) + (int64_t)qi * p.Spad + k0 +
                   kb * 16 + g * 4;
      if (k0 + kb * 16 >= p.Spad) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(dst + r, acc[qb][kb][r]);
    }
  }
}

// Per-call tuning words (last argument before the stream; 0 = what production uses; the library keeps no tuning state):
//   op_attn_fwd:  bit 0 = always the streaming kernel (tests / A-B timing); bits 1-2 = timing ablations of the resident kernel
//                 (tools only: 1 no K/V staging, 2 no compute); bits 3-6 = waves per workgroup of the resident kernel; bit 7 = no
//                 persistent kernel (193 ... 257 tokens then run the resident one: tests / A-B timing)
//   op_attn_bwd / op_attn_bwd_dbias_slabs:  bit 0 = separate dQ and dBias kernels instead of the merged one (tests);
//                 bit 1 = round 2's batch-chunk rule of the merged kernel (A/B timing); bits 2-3 = 2: dK/dV kernel with 64 keys per workgroup;
//                 bits 4-9 = forced number of batch chunks of the merged kernel (sweep of tools/attn_chunks_ab.py: the rule's choice is
//                 within 1 % of the best of {2 ... 16} at S = 257 / 250 / 65); bit 10 = no persistent dQ (+ dBias) kernel (193 ... 257
//                 tokens then run the kernels of rounds 1-3: tests / A-B timing)

template <bool HAS_BIAS, bool HAS_PAD>
int launch_fwd_res(const AttnArgs& a, const bf16_t* frag, dim3 grid, int nw, size_t sh, int rows_pad, int qb_per_wg, int abl,
                   hipStream_t s) {
  OP_ENSURE_LDS((attn_fwd_res_kernel<HAS_BIAS, HAS_PAD>), 2 * RES_MAX_S * 128, "attn_fwd");
  hipLaunchKernelGGL((attn_fwd_res_kernel<HAS_BIAS, HAS_PAD>), grid, dim3(nw * 64), sh, s, a, frag, rows_pad, qb_per_wg, abl);
  return OP_OK;
}

template <bool HAS_BIAS, bool HAS_PAD>
int launch_fwd_pers(const AttnArgs& a, const bf16_t* frag, int nwg, size_t sh, int rows_pad, int nitems, hipStream_t s) {
  OP_ENSURE_LDS((attn_fwd_pers_kernel<HAS_BIAS, HAS_PAD>), 4 * PERS_MAX_ROWS * 128 + 2 * PERS_NW * PERS_SCR * 4, "attn_fwd");
  hipLaunchKernelGGL((attn_fwd_pers_kernel<HAS_BIAS, HAS_PAD>), dim3(nwg), dim3(PERS_NW * 64), sh, s, a, frag, rows_pad, nitems);
  return OP_OK;
}

// number of batch chunks (= dbias slabs) of the merged dQ + dBias kernel; 1 when the separate kernels run.
// A workgroup of that kernel walks the samples of its chunk, two workgroups fit a CU (256 VGPRs), and all workgroups of a launch
// take the same time: the launch costs  rounds x samples-per-chunk  with rounds = ceil(workgroups / (2 x CUs)).  Round 2 aimed at
// ">= 768 workgroups", which at 257 tokens (5 query tiles x 24 heads) gave 7 chunks of 19 samples = 840 workgroups = 1.64 -> 2
// rounds x 19 = 38 sample-times; 4 chunks of 32 (480 workgroups, one round) or 8 of 16 cost 32.  Chosen here: the chunk count with
// the smallest cost (+ a little per chunk for the slab read-modify-write at the end of every workgroup; fewer slabs on a tie).
// CUs of the CURRENT device, cached per device (a process may drive several)
inline int attn_num_cus() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

inline int dbias_chunks(int64_t B, int64_t S, int64_t heads, bool merge, bool round2_rule = false, int forced = 0) {
  if (!merge || ceil_div(S, BKV) > 6) return 1;
  if (forced > 0) return ceil_div(B, ceil_div(B, min((int64_t)forced, B)));  // (tune bits 4-9: A/B timing)
  if (round2_rule) {  // (tune bit 1: A/B timing)
    const int64_t base2 = (int64_t)ceil_div(S, 64) * heads;
    int chunks = (int)((768 + base2 - 1) / base2);
    if (chunks < 1) chunks = 1;
    if (chunks > B) chunks = (int)B;
    return ceil_div(B, ceil_div(B, chunks));
  }
  const int cus = attn_num_cus();
  const int64_t base = (int64_t)ceil_div(S, 64) * heads, slots = 2 * (int64_t)cus;
  int best = 1;
  double best_cost = 1e30;
  for (int c = 1; c <= 32 && c <= B; ++c) {
    const int bchunk = ceil_div(B, c), n = ceil_div(B, bchunk);
    if (n != c) continue;  // the same split as a smaller c
    const double cost = (double)ceil_div(base * n, slots) * bchunk + 0.05 * n;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = c; }
  }
  return best;
}

// batch chunks (= dbias slabs) of the persistent dQ + dBias kernel: one workgroup per (chunk, head, half of the query blocks), at most
// one workgroup per CU
inline int pers_bwd_chunks(int64_t B, int64_t S, int64_t heads) {
  const int nqh = ceil_div(min((int64_t)ceil_div(S, 16), (int64_t)2 * PERS_NW), PERS_NW);
  int c = (int)(attn_num_cus() / (heads * nqh));
  if (c < 1) c = 1;
  if (c > B) c = (int)B;
  return ceil_div(B, ceil_div(B, c));
}
inline bool pers_bwd_shape(int64_t S) { return S > 192 && S <= 257; }

template <bool HAS_BIAS, bool HAS_PAD>
int launch_bwd_dq_pers(const AttnBwdArgs& a, int nwg, size_t sh, int rows_pad, int nqh, hipStream_t s) {
  if (a.S > 256) {
    OP_ENSURE_LDS((attn_bwd_dq_pers_kernel<HAS_BIAS, HAS_PAD, 5, 1>), 4 * PERS_MAX_ROWS * 128 + 2 * PERS_NW * 64 * 4, "attn_bwd");
    hipLaunchKernelGGL((attn_bwd_dq_pers_kernel<HAS_BIAS, HAS_PAD, 5, 1>), dim3(nwg), dim3(PERS_NW * 64), sh, s, a, rows_pad, nqh);
  } else {
    OP_ENSURE_LDS((attn_bwd_dq_pers_kernel<HAS_BIAS, HAS_PAD, 4, 4>), 4 * PERS_MAX_ROWS * 128 + 2 * PERS_NW * 64 * 4, "attn_bwd");
    hipLaunchKernelGGL((attn_bwd_dq_pers_kernel<HAS_BIAS, HAS_PAD, 4, 4>), dim3(nwg), dim3(PERS_NW * 64), sh, s, a, rows_pad, nqh);
  }
  return OP_OK;
}

// persistent dK / dV kernel: one workgroup per CU walks the (sample, head) items
inline int dkdv_pers_rows(int64_t S) { return (int)((S + 15) / 16) * 16; }
inline size_t dkdv_pers_lds(int64_t S) { return (size_t)2 * (2 * dkdv_pers_rows(S) * 128 + 6144) + 2 * PERS_NW * PERS_SCRL * 4; }
template <bool HAS_BIAS, bool HAS_PAD>
int launch_bwd_dkdv_pers(const AttnBwdArgs& a, hipStream_t s) {
  const int rows_pad = dkdv_pers_rows(a.S);
  const size_t sh = dkdv_pers_lds(a.S);
  const int nitems = a.B * a.heads;
  const int nwg = min(nitems, attn_num_cus());
  if (a.S > 256) {
    OP_ENSURE_LDS((attn_bwd_dkdv_pers_kernel<HAS_BIAS, HAS_PAD, 5, true>), (int)dkdv_pers_lds(257), "attn_bwd");
    hipLaunchKernelGGL((attn_bwd_dkdv_pers_kernel<HAS_BIAS, HAS_PAD, 5, true>), dim3(nwg), dim3(PERS_NW * 64), sh, s, a, rows_pad, nitems);
  } else {
    OP_ENSURE_LDS((attn_bwd_dkdv_pers_kernel<HAS_BIAS, HAS_PAD, 4, false>), (int)dkdv_pers_lds(256), "attn_bwd");
    hipLaunchKernelGGL((attn_bwd_dkdv_pers_kernel<HAS_BIAS, HAS_PAD, 4, false>), dim3(nwg), dim3(PERS_NW * 64), sh, s, a, rows_pad, nitems);
  }
  return OP_OK;
}

}  // namespace

extern "C" int op_prof_begin(int family, double work, void* stream);
extern "C" void op_prof_end(int slot, void* stream);

extern "C" {

// dbias of op_attn_bwd is fp32 [slabs][heads][S][Spad], pre-zeroed, slabs = this value (the batch chunks of the merged
// dQ + dBias kernel add into their own slab without atomics; sum the slabs afterwards).  `tune` as for op_attn_bwd.
// (the persistent dQ + dBias kernel of the 193 ... 257-token streams has its own chunk rule; which of the two kernels a call takes
// also depends on arguments this query does not see -- the larger count is returned, slabs a kernel does not write stay zero)
int64_t op_attn_bwd_dbias_slabs(int64_t B, int64_t S, int64_t heads, int64_t tune) {
  const int64_t n = dbias_chunks(B, S, heads, !(tune & 1), (tune & 2) != 0, (int)((tune >> 4) & 63));
  if (!(tune & 1) && !(tune & 1024) && pers_bwd_shape(S)) return max(n, (int64_t)pers_bwd_chunks(B, S, heads));
  return n;
}

// q, k, v: bf16 rows of `ld` elements (row = b*S + s), head h occupies columns [h*64, h*64+64) of each pointer
// (so one packed [B*S, 3H] projection output serves all three with pointer offsets 0, H, 2H).
// bias: bf16 [heads][S][Spad] or null.  key_pad: uint8 [B][Spad], non-zero = masked key, or null.
// out: bf16 [B*S][ldo] (head h at columns h*64..).  lse: fp32 [B][heads][lse_ld] (natural log) or null.
int op_attn_fwd(const void* q, const void* k, const void* v, int64_t ld, const void* bias, int64_t bias_batch_stride,
                const void* bias_frag, const void* key_pad, void* out, int64_t ldo, float* lse, int64_t lse_ld, int64_t B,
                int64_t S, int64_t Spad, int64_t heads, int64_t head_dim, float scale, int64_t tune, void* stream) {
  OP_CHECK_ARG(q && k && v && out, "attn_fwd: null pointer");
  OP_CHECK_ARG(head_dim == HD, "attn_fwd: head_dim %lld unsupported (only 64)", (long long)head_dim);
  OP_CHECK_ARG(B > 0 && S > 0 && heads > 0 && ld % 8 == 0 && ldo % 4 == 0, "attn_fwd: bad sizes");
  OP_CHECK_ARG((!bias && !key_pad) || (Spad >= ((S + 63) / 64) * 64 && Spad % 8 == 0),
               "attn_fwd: Spad must be >= S rounded up to 64");
  AttnArgs a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.ld = ld;
  a.bias = (const bf16_t*)bias; a.bias_bs = bias ? bias_batch_stride : 0; a.key_pad = (const uint8_t*)key_pad;
  a.out = (bf16_t*)out; a.ldo = ldo; a.lse = lse; a.lse_ld = lse_ld > 0 ? lse_ld : S;
  a.B = (int)B; a.S = (int)S; a.Spad = (int)Spad; a.heads = (int)heads; a.scale = scale;
  dim3 grid(ceil_div(S, BQ), (unsigned)heads, (unsigned)B);
  const int slot = op_prof_begin(1, 4.0 * (double)B * (double)heads * (double)S * (double)S * HD, stream);
  hipStream_t s = (hipStream_t)stream;
  // resident-K/V kernel: needs the fragment-major bias image when a bias is used, and 1/scale exact in bf16 (head_dim 64)
  const bool inv_exact = (float)(bf16_t)(1.0f / scale) * scale == 1.0f;
  // persistent kernel (round 4): 193 ... 256 tokens (<= 16 query blocks = one per wave) or exactly 257 (256 + a lone query); tune bit 7: off
  if (!(tune & 1) && !(tune & 128) && S > 192 && S <= 257 && (!bias || (bias_frag && inv_exact))) {
    const int rows_pad = ceil_div(S, 32) * 32;
    const size_t sh = (size_t)4 * rows_pad * 128 + 2 * PERS_NW * PERS_SCR * 4;
    const int nitems = (int)(B * heads);
    const int nwg = min(nitems, attn_num_cus());
    const bf16_t* fr = (const bf16_t*)bias_frag;
    int rc;
    if (bias && key_pad) rc = launch_fwd_pers<true, true>(a, fr, nwg, sh, rows_pad, nitems, s);
    else if (bias) rc = launch_fwd_pers<true, false>(a, fr, nwg, sh, rows_pad, nitems, s);
    else if (key_pad) rc = launch_fwd_pers<false, true>(a, fr, nwg, sh, rows_pad, nitems, s);
    else rc = launch_fwd_pers<false, false>(a, fr, nwg, sh, rows_pad, nitems, s);
    op_prof_end(slot, stream);
    if (rc != OP_OK) return rc;
    OP_LAUNCH_CHECK();
    return OP_OK;
  }
  if (!(tune & 1) && S <= RES_MAX_S && (!bias || (bias_frag && inv_exact))) {
    const int nqb = ceil_div(S, 16);
    const int nwg = ceil_div(nqb, RES_MAX_NW);
    const int qb_per_wg = ceil_div(nqb, nwg);              // 16-query blocks per workgroup
    int nwaves = qb_per_wg;                                  // waves per workgroup (a wave takes blocks wid, wid + nwaves, ...)
    if ((tune >> 3) & 15) nwaves = min(qb_per_wg, (int)((tune >> 3) & 15));   // (tests / A-B timing)
    else if (qb_per_wg == 9) nwaves = 8;                     // S = 257 ... 272: see the kernel
    const int rows_pad = ceil_div(S, 32) * 32;
    const size_t sh = (size_t)2 * rows_pad * 128;
    const dim3 rgrid(nwg, (unsigned)heads, (unsigned)B);
    const int abl = (int)((tune >> 1) & 3);
    const bf16_t* fr = (const bf16_t*)bias_frag;
    int rc;
    if (bias && key_pad) rc = launch_fwd_res<true, true>(a, fr, rgrid, nwaves, sh, rows_pad, qb_per_wg, abl, s);
    else if (bias) rc = launch_fwd_res<true, false>(a, fr, rgrid, nwaves, sh, rows_pad, qb_per_wg, abl, s);
    else if (key_pad) rc = launch_fwd_res<false, true>(a, fr, rgrid, nwaves, sh, rows_pad, qb_per_wg, abl, s);
    else rc = launch_fwd_res<false, false>(a, fr, rgrid, nwaves, sh, rows_pad, qb_per_wg, abl, s);
    op_prof_end(slot, stream);
    if (rc != OP_OK) return rc;
    OP_LAUNCH_CHECK();
    return OP_OK;
  }
  if (bias && key_pad) hipLaunchKernelGGL((attn_fwd_kernel<true, true>), grid, dim3(256), 0, s, a);
  else if (bias) hipLaunchKernelGGL((attn_fwd_kernel<true, false>), grid, dim3(256), 0, s, a);
  else if (key_pad) hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_fwd_kernel<false, false>), grid, dim3(256), 0, s, a);
  op_prof_end(slot, stream);
  OP_LAUNCH_CHECK();
  return OP_OK;
}


// Elements of the fragment-major image of n_img bias images of an S-token sequence (see op_attn_bias_pack).
int64_t op_attn_bias_frag_elems(int64_t n_img, int64_t S) {
  return n_img * (int64_t)ceil_div(S, 16) * ceil_div(S, 32) * FRAG_BLOCK;
}

// Repack row-major bias images  src [n_img][S][Spad] (bf16; n_img = heads for a shared image, B * heads for per-sample
// images)  into the fragment-major layout the resident forward kernel adds with the matrix pipe:
// dst [n_img][ceil(S/16)][ceil(S/32)][64][8] bf16, zero outside the sequence (op_attn_bias_frag_elems elements).
int op_attn_bias_pack(const void* src, void* dst, int64_t n_img, int64_t S, int64_t Spad, void* stream) {
  OP_CHECK_ARG(src && dst && n_img > 0 && S > 0 && Spad >= S, "attn_bias_pack: bad args");
  const int nqb = ceil_div(S, 16), nkp = ceil_div(S, 32);
  const int64_t total = n_img * nqb * nkp * 64;
  hipLaunchKernelGGL(bias_pack_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                     (bf16_t*)dst, (int)S, (int)Spad, nqb, nkp, total);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// delta[b][h][q] = sum_d dout * out  (fp32, row stride Spad); dout/out: [B*S][ldo]
int op_attn_bwd_delta(const void* dout, const void* out, int64_t ldo, float* delta, int64_t B, int64_t S, int64_t Spad,
                      int64_t heads, void* stream) {
  OP_CHECK_ARG(dout && out && delta && ldo % 8 == 0, "attn_bwd_delta: bad args");
  const int64_t threads = B * S * heads * 8;
  hipLaunchKernelGGL(attn_delta_kernel, dim3(ceil_div(threads, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout,
                     (const bf16_t*)out, ldo, delta, (int)B, (int)S, (int)Spad, (int)heads);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// Gradients of op_attn_fwd.  bias [heads][S][Spad] (rows = query) and biasT (same values, rows = key) are both needed
// when a bias was used; lse/delta are fp32 [B][heads][Spad]; dq/dk/dv rows have stride ldg (packed like q/k/v);
// dbias (fp32 [op_attn_bwd_dbias_slabs()][heads][S][Spad], pre-zeroed by the caller, accumulated into) is optional.
// bias_batch_stride != 0: bias / biasT hold one image per sample (that many elements apart; the masked-pretraining
// branch gathers a different token subset per sample, adapter/image.py:229-246); dbias then has B slabs, one per sample.
// out (nullable): the forward output rows (stride ldo).  Given, `delta` is a WORKSPACE: the dQ kernels compute it from dout and
// out and the dK/dV kernel (launched after them) reads it -- op_attn_bwd_delta's pass over both matrices is not needed.
int op_attn_bwd(const void* q, const void* k, const void* v, int64_t ld, const void* dout, const void* out, int64_t ldo, const void* bias,
                const void* biasT, const void* bias_frag, int64_t bias_batch_stride, const void* key_pad, const float* lse,
                float* delta, void* dq,
                void* dk, void* dv, int64_t ldg, float* dbias, int64_t B, int64_t S, int64_t Spad, int64_t heads,
                int64_t head_dim, float scale, int64_t tune, void* stream) {
  const bool merge_dbias = !(tune & 1);
  OP_CHECK_ARG(q && k && v && dout && lse && delta && dq && dk && dv, "attn_bwd: null pointer");
  OP_CHECK_ARG(head_dim == HD, "attn_bwd: head_dim %lld unsupported (only 64)", (long long)head_dim);
  OP_CHECK_ARG(Spad >= ((S + 127) / 128) * 128 && Spad % 8 == 0, "attn_bwd: Spad must be >= S rounded up to 128");
  OP_CHECK_ARG((bias == nullptr) == (biasT == nullptr), "attn_bwd: bias and biasT must be given together");
  OP_CHECK_ARG(!dbias || bias, "attn_bwd: dbias without a bias");
  OP_CHECK_ARG(ld % 8 == 0 && ldo % 8 == 0 && ldg % 4 == 0, "attn_bwd: bad leading dims");
  AttnBwdArgs a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.ld = ld;
  a.dout = (const bf16_t*)dout; a.ldo = ldo; a.bias = (const bf16_t*)bias; a.biasT = (const bf16_t*)biasT;
  a.bias_frag = bias ? (const bf16_t*)bias_frag : nullptr;
  a.bias_bs = bias ? bias_batch_stride : 0;
  a.key_pad = (const uint8_t*)key_pad; a.lse = lse; a.delta = delta; a.out = (const bf16_t*)out; a.delta_w = delta;
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.ldg = ldg; a.dbias = dbias;
  a.B = (int)B; a.S = (int)S; a.Spad = (int)Spad; a.heads = (int)heads; a.scale = scale;
  a.lone_keys = !(tune & 2048);
  {  // batch chunk per workgroup: enough workgroups to fill the chip, as few atomic rounds as possible
    const int64_t base = (int64_t)ceil_div(S, BQ) * ceil_div(S, BKV) * heads;
    int chunks = (int)((1024 + base - 1) / base);
    if (chunks < 1) chunks = 1;
    if (chunks > B) chunks = (int)B;
    a.bchunk = ceil_div(B, chunks);
  }
  hipStream_t s = (hipStream_t)stream;
  const double fl = 4.0 * (double)B * (double)heads * (double)S * (double)S * HD;
  int slot;
  const int bchunk_dkdv = a.bchunk;
  auto launch_dkdv = [&]() {
    AttnBwdArgs d = a;
    d.bchunk = bchunk_dkdv;
    const int sl = op_prof_begin(2, 2.0 * fl, stream);
    // keys per workgroup: 128 (two 16-key blocks per wave, 243 VGPRs, 2 waves/SIMD) or 64 (one block, 164 VGPRs, 3 waves/SIMD, a
    // third less padding at S = 257).  Measured (tools/attn_dkdv_ab.py, backward + dBias, B = 128): S = 257 0.6708 vs 0.6733 ms,
    // S = 250 0.5179 vs 0.5467, S = 65 0.1233 vs 0.1445, S = 785 1.436 vs 1.470 -- the narrower wave reads every Q / dO fragment
    // for half as many MFMAs; 128 stays.  tune bits 2-3: 2 forces 64 (A/B timing).
    const bool kb1 = ((tune >> 2) & 3) == 2;
    if (kb1) {
      if (bias) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<true, 1>), dim3(ceil_div(S, 64), (unsigned)heads, (unsigned)B), dim3(256), 0, s, d);
      else hipLaunchKernelGGL((attn_bwd_dkdv_kernel<false, 1>), dim3(ceil_div(S, 64), (unsigned)heads, (unsigned)B), dim3(256), 0, s, d);
    } else {
      if (bias) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<true, 2>), dim3(ceil_div(S, 128), (unsigned)heads, (unsigned)B), dim3(256), 0, s, d);
      else hipLaunchKernelGGL((attn_bwd_dkdv_kernel<false, 2>), dim3(ceil_div(S, 128), (unsigned)heads, (unsigned)B), dim3(256), 0, s, d);
    }
    op_prof_end(sl, stream);
  };
  if (!out) {  // delta precomputed (op_attn_bwd_delta): the kernels are independent
    launch_dkdv();
    OP_LAUNCH_CHECK();
  }
  const int nt = ceil_div(S, BKV);
  // persistent dQ (+ dBias) kernel (round 4): 193 ... 257 tokens, shared bias image in fragment-major form (or no bias), delta computed
  // in the kernel (out given); tune bit 10: off
  if (merge_dbias && !(tune & 1024) && out && pers_bwd_shape(S) && a.bias_bs == 0 &&
      (!bias || (a.bias_frag != nullptr && (float)(bf16_t)(1.0f / scale) * scale == 1.0f))) {
    const int nqb = ceil_div(S, 16);
    const int nqh = ceil_div(min(nqb, 2 * PERS_NW), PERS_NW);
    const int chunks = pers_bwd_chunks(B, S, heads);
    a.bchunk = ceil_div(B, chunks);
    const int rows_pad = S > 256 ? PERS_MAX_ROWS : 256;
    const size_t sh = (size_t)4 * rows_pad * 128 + 2 * PERS_NW * 64 * 4;
    const int nwg = chunks * (int)heads * nqh;
    AttnBwdArgs d = a;
    if (!bias) d.dbias = nullptr;
    slot = op_prof_begin(2, 1.5 * fl, stream);
    int rc;
    if (bias && key_pad) rc = launch_bwd_dq_pers<true, true>(d, nwg, sh, rows_pad, nqh, s);
    else if (bias) rc = launch_bwd_dq_pers<true, false>(d, nwg, sh, rows_pad, nqh, s);
    else if (key_pad) rc = launch_bwd_dq_pers<false, true>(d, nwg, sh, rows_pad, nqh, s);
    else rc = launch_bwd_dq_pers<false, false>(d, nwg, sh, rows_pad, nqh, s);
    op_prof_end(slot, stream);
    if (rc != OP_OK) return rc;
    OP_LAUNCH_CHECK();
    a.bchunk = bchunk_dkdv;
    // persistent dK / dV kernel (round 4; tune bit 12: the rounds 1-3 kernel).  Measured at B = 128 (tools/attn_pers_ab.py): S = 257
    // -12 % on the backward pair, S = 250 -2 %, S = 197 -1 %, S = 256 +5 % (two full key workgroups: nothing to gain) -- bit 13 forces it
    if (!(tune & 4096) && (!bias || biasT) && (S != 256 || (tune & 8192))) {
      const int sl = op_prof_begin(2, 2.0 * fl, stream);
      if (bias && key_pad) rc = launch_bwd_dkdv_pers<true, true>(a, s);
      else if (bias) rc = launch_bwd_dkdv_pers<true, false>(a, s);
      else if (key_pad) rc = launch_bwd_dkdv_pers<false, true>(a, s);
      else rc = launch_bwd_dkdv_pers<false, false>(a, s);
      op_prof_end(sl, stream);
      if (rc != OP_OK) return rc;
    } else {
      launch_dkdv();
    }
    OP_LAUNCH_CHECK();
    return OP_OK;
  }
  if (dbias && nt <= 6 && merge_dbias) {  // dQ and dBias together: dS summed over the batch chunk in registers
    // per-sample bias: every sample is its own chunk, slab b of dbias is the gradient of sample b's bias image
    const int chunks = a.bias_bs != 0 ? (int)B : dbias_chunks(B, S, heads, true, (tune & 2) != 0, (int)((tune >> 4) & 63));
    a.bchunk = ceil_div(B, chunks);
    const dim3 grid(ceil_div(S, 64), (unsigned)heads, (unsigned)chunks);
    const bool one_block = ceil_div(S - (nt - 1) * BKV, 16) == 1;  // last key tile = a single 16-key block (S = 257: 256 + CLS)
    // (the fragment path of 5 full tiles / 6 tiles would spill its dS accumulators: those lengths keep the round-1 softmax code)
    const bool frag = a.bias_frag != nullptr && (float)(bf16_t)(1.0f / scale) * scale == 1.0f && (nt <= 4 || (nt == 5 && one_block));
#define DQDB(N)                                                                                                           \
  do {                                                                                                                    \
    if (frag && one_block) hipLaunchKernelGGL((attn_bwd_dq_dbias_kernel<N, true, 1>), grid, dim3(256), 0, s, a);           \
    else if (frag) hipLaunchKernelGGL((attn_bwd_dq_dbias_kernel<N, true, 4>), grid, dim3(256), 0, s, a);                   \
    else if (one_block) hipLaunchKernelGGL((attn_bwd_dq_dbias_kernel<N, false, 1>), grid, dim3(256), 0, s, a);             \
    else hipLaunchKernelGGL((attn_bwd_dq_dbias_kernel<N, false, 4>), grid, dim3(256), 0, s, a);                            \
  } while (0)
    slot = op_prof_begin(2, 1.5 * fl, stream);
    switch (nt) {
      case 1: DQDB(1); break;
      case 2: DQDB(2); break;
      case 3: DQDB(3); break;
      case 4: DQDB(4); break;
      case 5: DQDB(5); break;
      default: DQDB(6); break;
    }
#undef DQDB
    op_prof_end(slot, stream);
    OP_LAUNCH_CHECK();
    if (out) {
      launch_dkdv();
      OP_LAUNCH_CHECK();
    }
    return OP_OK;
  }
  slot = op_prof_begin(2, 1.5 * fl, stream);
  if (bias) hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, dim3(ceil_div(S, BQ), (unsigned)heads, (unsigned)B), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, dim3(ceil_div(S, BQ), (unsigned)heads, (unsigned)B), dim3(256), 0, s, a);
  op_prof_end(slot, stream);
  OP_LAUNCH_CHECK();
  if (out) {
    launch_dkdv();
    OP_LAUNCH_CHECK();
  }
  if (dbias) {
    if (a.bias_bs != 0) a.bchunk = 1;  // per-sample bias images: one sample per workgroup, slab b of dbias for sample b
    const int chunks = ceil_div(B, a.bchunk);
    hipLaunchKernelGGL(attn_bwd_dbias_kernel, dim3(ceil_div(S, BQ), ceil_div(S, BKV), (unsigned)(heads * chunks)), dim3(256), 0,
                       s, a);
    OP_LAUNCH_CHECK();
  }
  return OP_OK;
}

}  // extern "C"
