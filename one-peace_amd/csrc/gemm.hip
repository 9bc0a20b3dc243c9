// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] . B[N,K]^T  (both operands K-contiguous = nn.Linear layout)
// with the epilogues the ONE-PEACE encoder layer needs fused in.
//
// Replaces (reference file:line):
//   * q/k/v/out projections, multihead_attention.py:63-66,103-105,124 (three weights = three "segments"
//     of one launch; k_proj has no bias)
//   * GeGLU  gelu(x W0^T) * (x W1^T), transformer_layer.py:54-67      -> EPI_GEGLU (two weights, one launch)
//   * fused_dropout_res  residual + droppath(gamma * y), transformer_layer.py:70-88 -> EPI_RESID
//   * scale * local @ all^T of the contrastive head, image_text_pretrain_loss.py:171-172 -> EPI_F32
//
// Kernels (chosen per launch by plan_gemm / launch256) and the launches each one serves:
//   gemm256v_kernel / gemm256p_kernel   256x256, four waves of 128x128 (one wave per SIMD); p = persistent + grouped form.  Every
//                                       bias / residual / plain launch that fills the chip: 15 of the 16 NT launches of a training layer
//   gemm256b_kernel                     256x256 BK = 64, eight waves of 128x64: GeGLU-epilogue launches that fill the chip (inference,
//                                       training passes whose FFN is not split into a plain launch + op_ln_geglu_fwd)
//   gemm256_kernel                      256x256 BK = 32 four-stage: 256x256 plans that do NOT fill the chip, split-K slabs, fp32 outputs,
//                                       N or K outside the four-wave kernels' rules
//   gemm_nt_kernel                      128x128 (small launches, tail rows; described next)
//   gemm256w_tn_kernel / gemm256_tn_kernel   weight gradients C = A^T B outside the grouped launch: four waves for <= 108 output tiles at
//                                       K >= 16384, eight waves otherwise;  gemm256w_tn_grouped_kernel: all weight gradients of a layer
// (Round 5 removed round 2's four-wave kernel gemm256w_kernel, the second instruction schedule of gemm256v_kernel and the six timing
// ablations of gemm256_kernel: tools-only flavours.)
// Tiling of the 128x128 kernel: one output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA
// 16x16x32 accumulators), BK = 64, two LDS buffers (2 x 32 KiB), one barrier per K-tile.  Operand tiles
// are [128 rows][64 k] bf16 (128-byte rows) with the 16-byte slot index XOR-ed by (row & 7): the
// ds_read_b128 fragment reads are then bank-conflict free (each 16-lane group covers all 64 banks once).
// Staging is either LDS-DMA (global_load_lds_dwordx4: the swizzle is applied to the per-lane SOURCE address,
// the LDS image stays lane-linear) or a register-staged fallback with identical LDS image.
//
// The MFMA is issued "swapped" (first operand = weight rows, second = activation rows) so that a lane ends
// up with 16 *contiguous output columns* of one output row: weight row i of a 16-row fragment lands in lanes
// (i>>2) as register (i&3), and the loader permutes which weight row sits in LDS row p so that lane group g
// owns columns g*16 .. g*16+15 of the wave's 64.  Every C row is therefore written as 128 contiguous bytes
// by 4 lanes with 16-byte stores.
//
// Roofline: MFMA (dense bf16, 2.5 PFLOP/s).  Algorithmic work = 2*M*N*K flops per launch.
#include "common.h"
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <vector>

#include "gemm_epilogue_v.h"  // EPI_*, GemmArgs, xcd_remap, resid_out, epilogue_v (shared with fp8.hip)

namespace {

constexpr int BM = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

// LDS row p of the weight tile -> output column (relative to the tile) it feeds.
template <int EPI>
__device__ __forceinline__ int w_row_to_col(int p) {
  const int i = p & 15;
  if (EPI == EPI_GEGLU) {
    const int pp = p & 63;  // rows 0..63 <- W0, 64..127 <- W1, same column map
    return (pp >> 5) * 32 + (i >> 2) * 8 + ((pp >> 4) & 1) * 4 + (i & 3);
  }
  return (p & ~63) + (i >> 2) * 16 + ((p >> 4) & 3) * 4 + (i & 3);
}

// Shared epilogue.  Lane (g, t) holds, for mi = 0..MI-1, output row m = mrow0 + mi*16 + t and 16 contiguous columns
// nbase + g*16 .. +15 (acc[ni][mi][r] <-> column ni*4 + r); GeGLU: 8 columns fbase + g*8 .. +7 with
// acc[nl][mi][r] = h0, acc[2+nl][mi][r] = h1 of column nl*4 + r.
template <int EPI, int MI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, void* Cout, f32x4 (&acc)[4][MI], int mrow0, int nbase,
                                              int fbase, int g, int t, int64_t bias_off = 0) {  // bias_off: batched launches (element offset of the batch's bias vector)
  if (EPI == EPI_GEGLU) {
    const int f0 = fbase + g * 8;  // 8 contiguous f:  acc[nl][mi][r] = h0, acc[2+nl][mi][r] = h1, f = f0 + nl*4 + r
    if (f0 >= p.N) return;
    bf16_t* G = (bf16_t*)Cout;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = mrow0 + mi * 16 + t;
      if (m >= p.M) continue;
      float go[8], h0[8], h1[8];
#pragma unroll
      for (int nl = 0; nl < 2; ++nl)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = acc[nl][mi][r], b = acc[2 + nl][mi][r];
          h0[nl * 4 + r] = a;
          h1[nl * 4 + r] = b;
          go[nl * 4 + r] = gelu_erf(a) * b;
        }
      Vec8<bf16_t>::store(G + (int64_t)m * p.ldc + f0, go);
      if (p.H0) {
        Vec8<bf16_t>::store(p.H0 + (int64_t)m * p.ldc + f0, h0);
        Vec8<bf16_t>::store(p.H1 + (int64_t)m * p.ldc + f0, h1);
      }
    }
    return;
  }

  const int nc0 = nbase + g * 16;
  if (nc0 >= p.N) return;
  float bv[16];
  {
    const int seg = nc0 / p.n_seg;
    const bf16_t* bp = p.bias[seg];
    if (bp) {
      bp += bias_off;
      float tmp[8];
      Vec8<bf16_t>::load(bp + (nc0 - seg * p.n_seg), tmp);
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[j] = tmp[j];
      if (nc0 + 8 < p.N) {
        Vec8<bf16_t>::load(bp + (nc0 - seg * p.n_seg) + 8, tmp);
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[8 + j] = tmp[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[8 + j] = 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) bv[j] = 0.f;
    }
  }
  const bool second = nc0 + 8 < p.N;
  float alpha = 1.f;
  if (EPI == EPI_F32 && p.alpha) alpha = *p.alpha;
  constexpr bool RES = epi_is_resid(EPI), ROWS = EPI == EPI_RESID_ROWS;
  float gv[16];
  if (RES) {
    if (p.gamma) {
      float tmp[8];
      Vec8<bf16_t>::load(p.gamma + nc0, tmp);
#pragma unroll
      for (int j = 0; j < 8; ++j) gv[j] = tmp[j];
      if (second) {
        Vec8<bf16_t>::load(p.gamma + nc0 + 8, tmp);
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[8 + j] = tmp[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) gv[j] = 1.f;
    }
  }
  // residual epilogue: ALL residual rows (and row scales) of the MI fragments are requested first -- one load -> wait -> store
  // chain per fragment left eight HBM latencies in a row per wave, which the four waves of the one-wave-per-SIMD kernels cannot hide
  typename Vec8<bf16_t>::raw_t rraw[RES ? MI : 1][2];
  float rsv[RES ? MI : 1];
  if (RES) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int mc = min(mrow0 + mi * 16 + t, p.M - 1);
      // EPI_RESID_ROWS: the row of resid / C behind the lane's row (see gemm_epilogue_v.h); read again at the store (a cached load
      // instead of MI live registers: the eight-wave kernels have none to spare)
      const bf16_t* rp = p.resid + (int64_t)(ROWS ? max(p.rows[mc], 0) : mc) * p.ldr + nc0;
      rraw[mi][0] = Vec8<bf16_t>::ldraw(rp);
      rraw[mi][1] = Vec8<bf16_t>::ldraw(second ? rp + 8 : rp);
      rsv[mi] = p.rowscale ? p.rowscale[(mc + p.m_off) / p.rows_per_sample] : 1.f;
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mrow0 + mi * 16 + t;
    if (m >= p.M) continue;
    float o[16];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[ni * 4 + r] = acc[ni][mi][r];
    if (EPI == EPI_BIAS) {
#pragma unroll
      for (int j = 0; j < 16; ++j) o[j] += bv[j];
    } else if (EPI == EPI_F32) {
#pragma unroll
      for (int j = 0; j < 16; ++j) o[j] = alpha * o[j] + bv[j];
    } else if (RES) {
      const float rs = rsv[mi];
      float rv[16];
      float tmp[8];
      Vec8<bf16_t>::cvt(rraw[mi][0], tmp);
#pragma unroll
      for (int j = 0; j < 8; ++j) rv[j] = tmp[j];
      Vec8<bf16_t>::cvt(rraw[mi][1], tmp);
#pragma unroll
      for (int j = 0; j < 8; ++j) rv[8 + j] = tmp[j];
#pragma unroll
      for (int j = 0; j < 16; ++j) o[j] += bv[j];
      if (p.H0) {  // branch output y (pre layer-scale), needed by the backward pass for d gamma
        float lo[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { lo[j] = o[j]; hi[j] = o[8 + j]; }
        Vec8<bf16_t>::store(p.H0 + (int64_t)m * p.ldc + nc0, lo);
        if (second) Vec8<bf16_t>::store(p.H0 + (int64_t)m * p.ldc + nc0 + 8, hi);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) o[j] = resid_out(rv[j], rs, gv[j], o[j]);
    }
    if (EPI == EPI_F32) {
      float* C = (float*)Cout + (int64_t)m * p.ldc + nc0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q >= 2 && !second) break;
        *reinterpret_cast<f32x4*>(C + q * 4) = (f32x4){o[q * 4], o[q * 4 + 1], o[q * 4 + 2], o[q * 4 + 3]};
      }
    } else {
      const int mo = ROWS ? p.rows[m] : m;
      if (ROWS && mo < 0) continue;  // a row without a place in the full matrix
      bf16_t* C = (bf16_t*)Cout + (int64_t)mo * p.ldc + nc0;
      float lo[8], hi[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { lo[j] = o[j]; hi[j] = o[8 + j]; }
      Vec8<bf16_t>::store(C, lo);
      if (second) Vec8<bf16_t>::store(C + 8, hi);
    }
  }
}

// Batched launches (round 5; op_gemm_nt_batched): blockIdx.z = problem of a batch of equally shaped products whose operands lie at constant
// element strides (the 16 groups of the audio positional convolution, adapter/audio.py:57-84: one launch of 16 x 268 tiles instead of 16
// launches that fill half the chip each); all strides 0 = the plain launch.
struct NtBatch { int64_t a, b, c, bias; };

template <int EPI, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmArgs p, const NtBatch bt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t zb = blockIdx.z;
  const int wm = wid >> 1, wn = wid & 1;
  const int g = lane >> 4, t = lane & 15;
  constexpr int BN_OUT = (EPI == EPI_GEGLU) ? 64 : 128;

  // ---- workgroup -> tile (XCD-contiguous chunks, then grouped along M for L2 reuse of the W panels) ----
  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  constexpr int GM = 8;
  const int per_group = GM * p.tiles_n;
  const int first_m = (pid / per_group) * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int in_group = pid % per_group;
  const int pid_m = first_m + in_group % gsz;
  const int pid_n = in_group / gsz;
  const int m0 = pid_m * BM, n0 = pid_n * BN_OUT;

  // ---- per-thread staging addresses: 4 x 16-byte slots of A and of B per K-tile ----
  const bf16_t* srcA[4];
  const bf16_t* srcB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + tid;
    const int row = q >> 3, pc = q & 7;
    const int c = pc ^ (row & 7);
    const int gm = min(m0 + row, p.M - 1);
    srcA[i] = p.A + zb * bt.a + (int64_t)gm * p.lda + c * 8;
    int gn = min(n0 + w_row_to_col<EPI>(row), p.N - 1);
    const bf16_t* base;
    if (EPI == EPI_GEGLU) {
      base = (row < 64) ? p.B[0] : p.B[1];
    } else {
      const int seg = gn / p.n_seg;
      base = p.B[seg];
      gn -= seg * p.n_seg;
    }
    srcB[i] = base + zb * bt.b + (int64_t)gn * p.ldb + c * 8;
  }

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) inside a tile: row*128 + ((kk*4+g) ^ (row&7))*16, row&7 == t&7
  int rdW[4], rdX[4], swz[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) swz[kk] = ((kk * 4 + g) ^ (t & 7)) << 4;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int row = (EPI == EPI_GEGLU) ? ((ni >> 1) * 64 + wn * 32 + (ni & 1) * 16 + t) : (wn * 64 + ni * 16 + t);
    rdW[ni] = row * 128;
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) rdX[mi] = (wm * 64 + mi * 16 + t) * 128;

  int nk = p.K / BK;
  void* Cout = (EPI == EPI_F32) ? (void*)((float*)p.C + zb * bt.c) : (void*)((bf16_t*)p.C + zb * bt.c);
  if (p.kt_per_split > 0) {  // split-K: this workgroup owns K-tiles [z*kps, min(nk, (z+1)*kps)) and its own output slab
    const int kt0 = blockIdx.y * p.kt_per_split;
    nk = min(nk - kt0, p.kt_per_split);
#pragma unroll
    for (int i = 0; i < 4; ++i) { srcA[i] += (int64_t)kt0 * BK; srcB[i] += (int64_t)kt0 * BK; }
    Cout = (float*)p.C + (int64_t)blockIdx.y * p.slab;
  }
  u32x4 ra[4], rb[4];

  auto stage_glds = [&](int buf) {
    char* la = smem + buf * (2 * TILE_BYTES);
    char* lb = la + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int wbase = (i * 256 + wid * 64) * 16;  // wave-uniform; hardware adds lane*16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcA[i],
                                       (__attribute__((address_space(3))) void*)(la + wbase), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcB[i],
                                       (__attribute__((address_space(3))) void*)(lb + wbase), 16, 0, 0);
    }
  };
  auto load_regs = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const u32x4*>(srcA[i]);
      rb[i] = *reinterpret_cast<const u32x4*>(srcB[i]);
    }
  };
  auto write_regs = [&](int buf) {
    char* la = smem + buf * (2 * TILE_BYTES);
    char* lb = la + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = (i * 256 + tid) * 16;
      *reinterpret_cast<u32x4*>(la + off) = ra[i];
      *reinterpret_cast<u32x4*>(lb + off) = rb[i];
    }
  };
  auto advance = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) { srcA[i] += BK; srcB[i] += BK; }
  };
  auto compute = [&](int buf) {
    const char* la = smem + buf * (2 * TILE_BYTES);
    const char* lb = la + TILE_BYTES;
    // kk = 0 fragments, then the kk = 1 fragment reads are interleaved (one per two MFMAs) with the kk = 0 MFMAs
    bf16x8 wf0[4], xf0[4], wf1[4], xf1[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) wf0[ni] = *reinterpret_cast<const bf16x8*>(lb + rdW[ni] + swz[0]);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) xf0[mi] = *reinterpret_cast<const bf16x8*>(la + rdX[mi] + swz[0]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int idx = 2 * j + h, mi = idx >> 2, ni = idx & 3;
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[ni], xf0[mi], acc[ni][mi], 0, 0, 0);
      }
      if (j < 4) wf1[j] = *reinterpret_cast<const bf16x8*>(lb + rdW[j] + swz[1]);
      else xf1[j - 4] = *reinterpret_cast<const bf16x8*>(la + rdX[j - 4] + swz[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[ni], xf1[mi], acc[ni][mi], 0, 0, 0);
  };

  if (GLDS) {
    stage_glds(0);
    advance();
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // tile kt landed; everyone is done reading the other buffer
      if (kt + 1 < nk) { stage_glds((kt + 1) & 1); advance(); }
      compute(kt & 1);
    }
  } else {
    load_regs();
    advance();
    write_regs(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) { load_regs(); advance(); }
      compute(kt & 1);
      if (kt + 1 < nk) write_regs((kt + 1) & 1);
      __syncthreads();
    }
  }

  gemm_epilogue<EPI, 4>(p, Cout, acc, m0 + wm * 64, n0 + wn * 64, n0 + wn * 32, g, t, zb * bt.bias);
}

// =====================================================================================================================
// 256 x 256 tile, 8 waves (2 x 4, 128 x 64 per wave = 4 x 8 accumulators), BK = 32, FOUR LDS stages of 32 KiB
// (A 256x32 + B 256x32), LDS-DMA staging three stages ahead with a COUNTED vmcnt and a raw s_barrier, so loads stay
// in flight across barriers (one barrier per 32 MFMAs per wave).  Twice the flops per byte pulled from L2 of the 128^2
// kernel.  Operand rows are 64 bytes: the 16-byte slot index is XOR-ed with (row&1) | ((row>>2)&1)<<1, which makes the
// ds_read_b128 fragment reads bank-conflict free under the gfx950 bank model (MI355X_MICROARCH.md, LDS section).
// =====================================================================================================================
constexpr int BM2 = 256, BK2 = 32, STAGES2 = 4;
constexpr int OPER2_BYTES = BM2 * BK2 * 2;      // 16 KiB per operand per stage
constexpr int STAGE2_BYTES = 2 * OPER2_BYTES;   // 32 KiB

__device__ __forceinline__ int swz64(int row) { return (row & 1) | (((row >> 2) & 1) << 1); }

// ds_read_b64_tr_b16 through inline asm, NOT the builtin: the compiler cannot tell what a transpose read touches, so next to
// LDS-DMA it puts an s_waitcnt vmcnt(0) in front of the first one after every barrier -- which drained the whole four-stage
// LDS-DMA pipeline once per K-step (round 1's TN kernel ran that way).  As asm the read is invisible to its wait insertion; the
// kernels wait for the results themselves (lgkmcnt(0) before the barrier of the step that consumes them).  `hi`: + 2048 bytes.
__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const char*)p);
}
__device__ __forceinline__ s16x4 tr_read16(unsigned addr, bool hi) {
  s16x4 r;
  if (hi) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(r) : "v"(addr));
  else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
__device__ __forceinline__ bf16x8 join16(s16x4 a, s16x4 b) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

template <int EPI>
__device__ __forceinline__ int w_row_to_col256(int p) {
  const int i = p & 15;
  if (EPI == EPI_GEGLU) {
    const int pp = p & 127;  // rows 0..127 <- W0, 128..255 <- W1
    return (pp >> 5) * 32 + (i >> 2) * 8 + ((pp >> 4) & 1) * 4 + (i & 3);
  }
  return (p & ~63) + (i >> 2) * 16 + ((p >> 4) & 3) * 4 + (i & 3);
}

// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]); the builtin form keeps
// the compiler's own scoreboard in sync, so it does not add conservative waits of its own around ours.
#define WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 0xF) | ((((n) >> 4) & 3) << 14))

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int g = lane >> 4, t = lane & 15;
  constexpr int BN_OUT = (EPI == EPI_GEGLU) ? 128 : 256;

  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int GM = p.gm;
  const int per_group = GM * p.tiles_n;
  const int first_m = (pid / per_group) * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int in_group = pid % per_group;
  const int pid_m = first_m + in_group % gsz;
  const int pid_n = in_group / gsz;
  const int m0 = pid_m * BM2, n0 = pid_n * BN_OUT;

  int nk = p.K / BK2;
  int kt0 = 0;
  void* Cout = p.C;
  if (p.kt_per_split > 0) {  // split-K: K-stages [z*kps, min(nk, (z+1)*kps)), own fp32 output slab
    kt0 = blockIdx.y * p.kt_per_split;
    nk = min(nk - kt0, p.kt_per_split);
    Cout = (float*)p.C + (int64_t)blockIdx.y * p.slab;
  }

  // ---- staging: wave-uniform base pointers + 32-bit per-lane byte offsets (2 x 16 bytes of A and of B per stage) ----
  // slot q = i*512 + tid covers LDS row q>>2, 16-byte slot q&3; i = 0 -> rows 0..127, i = 1 -> rows 128..255.
  const char* baseA = (const char*)(p.A + (int64_t)kt0 * BK2);
  const char* baseB[2];
  unsigned offA[2], offB[2];
  {
    const int seg = (EPI == EPI_GEGLU) ? 0 : n0 / p.n_seg;  // a 256-wide tile never straddles segments (n_seg % 256 == 0)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = i * 512 + tid;
      const int row = q >> 2, pc = q & 3;
      const int c = pc ^ swz64(row);
      const int gm = min(m0 + row, p.M - 1);
      offA[i] = (unsigned)(((int64_t)gm * p.lda + c * 8) * 2);
      int gn = min(n0 + w_row_to_col256<EPI>(row), p.N - 1);
      const bf16_t* wb;
      if (EPI == EPI_GEGLU) {
        wb = i == 0 ? p.B[0] : p.B[1];
      } else {
        wb = p.B[seg];
        gn -= seg * p.n_seg;
      }
      baseB[i] = (const char*)(wb + (int64_t)kt0 * BK2);
      offB[i] = (unsigned)(((int64_t)gn * p.ldb + c * 8) * 2);
    }
  }

  f32x4 acc[4][8];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment reads: every fragment row is (multiple of 16) + t, so the swizzle term only depends on the lane
  const int fsw = (g ^ swz64(t)) << 4;
  const int baseX = (wm * 128 + t) * 64 + fsw;                                              // + mi * 1024
  const int baseW = OPER2_BYTES + ((EPI == EPI_GEGLU) ? (wn * 32 + t) : (wn * 64 + t)) * 64 + fsw;  // + per-ni constant

  auto issue = [&](int slot) {
    char* la = smem + slot * STAGE2_BYTES;
    char* lb = la + OPER2_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int wbase = (i * 512 + wid * 64) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseA + offA[i]),
                                       (__attribute__((address_space(3))) void*)(la + wbase), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseB[i] + offB[i]),
                                       (__attribute__((address_space(3))) void*)(lb + wbase), 16, 0, 0);
    }
    baseA += BK2 * 2;
    baseB[0] += BK2 * 2;
    baseB[1] += BK2 * 2;
  };
  auto wait_landed = [&](int younger) {  // `younger` = stages issued after the one we need (4 LDS-DMA ops each)
    if (younger >= 3) WAIT_VM(12);
    else if (younger == 2) WAIT_VM(8);
    else if (younger == 1) WAIT_VM(4);
    else WAIT_VM(0);
  };
  auto read_frags = [&](int kt, bf16x8 (&wf)[4], bf16x8 (&xf)[8]) {
    const char* st = smem + (kt & (STAGES2 - 1)) * STAGE2_BYTES;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int o = (EPI == EPI_GEGLU) ? ((ni >> 1) * 128 + (ni & 1) * 16) * 64 : ni * 16 * 64;
      wf[ni] = *reinterpret_cast<const bf16x8*>(st + baseW + o);
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) xf[mi] = *reinterpret_cast<const bf16x8*>(st + baseX + mi * 1024);
  };
  auto mma = [&](const bf16x8 (&wf)[4], const bf16x8 (&xf)[8]) {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
  };
  // One pipeline step.  Four LDS stages are in flight; the fragments of stage kt+1 are read into the OTHER register set
  // while the 32 MFMAs of stage kt run.  Order: own LDS reads retired -> stage kt+1 landed (counted vmcnt) -> ONE
  // barrier (stage kt+1 visible to all waves AND every wave holds stage kt in registers, so slot kt&3 is free) ->
  // {refill of that slot with stage kt+4, fragment reads of kt+1, MFMAs of kt} as ONE interleaved instruction stream:
  // both waves of a SIMD leave the barrier together, so any load-issue phase ahead of the MFMAs would idle the matrix
  // pipe; sched_group_barrier spreads the 16 memory instructions between the 32 MFMAs (1 per 2) instead.
  auto step_steady = [&](int kt, const bf16x8 (&cur_w)[4], const bf16x8 (&cur_x)[8], bf16x8 (&nxt_w)[4], bf16x8 (&nxt_x)[8]) {
    WAIT_LGKM0();
    WAIT_VM(8);
    __builtin_amdgcn_s_barrier();
    const char* st = smem + ((kt + 1) & (STAGES2 - 1)) * STAGE2_BYTES;  // fragments of the next stage
    char* la = smem + (kt & (STAGES2 - 1)) * STAGE2_BYTES;              // slot being refilled with stage kt+4
    char* lb = la + OPER2_BYTES;
#pragma unroll
    for (int j = 0; j < 16; ++j) {  // 16 groups of {2 MFMA, 1 memory instruction}, order pinned
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int idx = 2 * j + h, mi = idx >> 2, ni = idx & 3;
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur_w[ni], cur_x[mi], acc[ni][mi], 0, 0, 0);
      }
      if (j < 4) {
        const int o = (EPI == EPI_GEGLU) ? ((j >> 1) * 128 + (j & 1) * 16) * 64 : j * 16 * 64;
        nxt_w[j] = *reinterpret_cast<const bf16x8*>(st + baseW + o);
      } else if (j < 12) {
        nxt_x[j - 4] = *reinterpret_cast<const bf16x8*>(st + baseX + (j - 4) * 1024);
      } else {
        const int i = (j - 12) >> 1;
        const int wbase = (i * 512 + wid * 64) * 16;
        if ((j & 1) == 0)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseA + offA[i]),
                                           (__attribute__((address_space(3))) void*)(la + wbase), 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseB[i] + offB[i]),
                                           (__attribute__((address_space(3))) void*)(lb + wbase), 16, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    baseA += BK2 * 2;
    baseB[0] += BK2 * 2;
    baseB[1] += BK2 * 2;
  };
  auto step_tail = [&](int kt, const bf16x8 (&cur_w)[4], const bf16x8 (&cur_x)[8], bf16x8 (&nxt_w)[4], bf16x8 (&nxt_x)[8]) {
    const bool has_next = kt + 1 < nk;
    WAIT_LGKM0();
    if (has_next) wait_landed(min(nk - 2 - kt, STAGES2 - 2));
    __builtin_amdgcn_s_barrier();
    if (has_next) read_frags(kt + 1, nxt_w, nxt_x);
    mma(cur_w, cur_x);
  };

#pragma unroll
  for (int s0 = 0; s0 < STAGES2; ++s0)
    if (s0 < nk) issue(s0);
  bf16x8 wfA[4], xfA[8], wfB[4], xfB[8];
  wait_landed(min(nk - 1, STAGES2 - 1));
  __builtin_amdgcn_s_barrier();
  read_frags(0, wfA, xfA);
  int kt = 0;
  for (; kt + STAGES2 + 1 < nk; kt += 2) {  // steady state: stages kt+4 and kt+5 still to be issued (nk is even)
    step_steady(kt, wfA, xfA, wfB, xfB);
    step_steady(kt + 1, wfB, xfB, wfA, xfA);
  }
  for (; kt < nk; kt += 2) {  // last (up to) four stages: nothing left to prefetch
    step_tail(kt, wfA, xfA, wfB, xfB);
    step_tail(kt + 1, wfB, xfB, wfA, xfA);
  }
  gemm_epilogue<EPI, 8>(p, Cout, acc, m0 + wm * 128, n0 + wn * 64, n0 + wn * 32, g, t);
}

// =====================================================================================================================
// BK = 64 flavour of the 256 x 256 NT kernel ("full-line staging").  Same wave layout, MFMA order, fragment double
// buffering and {2 MFMA, 1 memory instruction} interleave as gemm256_kernel, but every operand row in LDS is a full
// 128-byte line: one wave-level global_load_lds moves 8 rows x 128 B instead of 16 rows x 64 B, which halves the number
// of L2 requests / texture-addresser lines per byte (the 64-byte segments of the BK = 32 kernel are what limits its main
// loop, profiles/r1_gemm_experiments.md).  LDS: FIVE 32 KiB slots (all 160 KiB), each one operand K-tile
// [256 rows][64 k], 16-byte slot index XOR-ed with (row & 7) (conflict-free ds_read_b128, same image as the 128^2
// kernel).  Operand tiles go through the slots in the order A0 B0 A1 B1 A2 ...; a K-tile is two half-steps of 32 MFMAs
// per wave; half-step (i,0) issues A(i+2), half-step (i,1) issues B(i+2); ONE barrier per K-tile, at the start of (i,1):
//   RAW: before it every wave retires its share of A(i+1), B(i+1) (counted vmcnt: only A(i+2) may stay in flight) -- the
//        fragments of (i+1,0) are read after it;
//   WAR: before it every wave has retired its fragment reads of tile i (lgkmcnt(0)), so the slots of A(i), B(i) may be
//        refilled after it (B(i+2) -> slot of A(i) in (i,1), A(i+3) -> slot of B(i) in (i+1,0)).
// =====================================================================================================================
constexpr int SLOT3_BYTES = 256 * 128;  // one operand K-tile
constexpr int SLOTS3 = 5;

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256b_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int g = lane >> 4, t = lane & 15;
  constexpr int BN_OUT = (EPI == EPI_GEGLU) ? 128 : 256;

  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int GM = p.gm;
  const int per_group = GM * p.tiles_n;
  const int first_m = (pid / per_group) * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int in_group = pid % per_group;
  const int pid_m = first_m + in_group % gsz;
  const int pid_n = in_group / gsz;
  const int m0 = pid_m * BM2, n0 = pid_n * BN_OUT;

  int nk = p.K / 64;
  int kt0 = 0;
  void* Cout = p.C;
  if (p.kt_per_split > 0) {  // split-K (kt_per_split counts 32-deep stages and is even)
    kt0 = blockIdx.y * (p.kt_per_split >> 1);
    nk = min(nk - kt0, p.kt_per_split >> 1);
    Cout = (float*)p.C + (int64_t)blockIdx.y * p.slab;
  }

  // ---- staging: op j of an operand tile covers LDS rows j*64 + (tid >> 3), 16-byte slot tid & 7 ----
  const int srow = tid >> 3;                       // row within the 64-row group
  const int sc = (tid & 7) ^ (srow & 7);           // source k-chunk of this lane's LDS slot
  const char* baseA = (const char*)(p.A + (int64_t)kt0 * 64);
  unsigned offA[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int gm = min(m0 + j * 64 + srow, p.M - 1);
    offA[j] = (unsigned)(((int64_t)gm * p.lda + sc * 8) * 2);
  }
  // weight rows: LDS row j*64 + q  <-  output column (j-dependent uniform term) + f(q); no N clamp (N % BN_OUT == 0)
  const char* baseB[4];
  unsigned offB;
  {
    const int seg = (EPI == EPI_GEGLU) ? 0 : n0 / p.n_seg;
    const int colq = w_row_to_col256<EPI>(srow);  // column of LDS row `srow` of group 0
    offB = (unsigned)(((int64_t)colq * p.ldb + sc * 8) * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16_t* wb;
      int col0;
      if (EPI == EPI_GEGLU) {
        wb = p.B[j >> 1];
        col0 = n0 + (j & 1) * 64;
      } else {
        wb = p.B[seg];
        col0 = n0 - seg * p.n_seg + j * 64;
      }
      baseB[j] = (const char*)(wb + (int64_t)col0 * p.ldb + (int64_t)kt0 * 64);
    }
  }

  f32x4 acc[4][8];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment reads: row = (multiple of 16) + t  ->  swizzle term (t & 7); half h of the K-tile = 16-byte chunks h*4 + g
  const int fsw[2] = {((0 * 4 + g) ^ (t & 7)) << 4, ((1 * 4 + g) ^ (t & 7)) << 4};
  const int rowX = (wm * 128 + t) * 128;                                                   // + mi * 2048
  const int rowW = ((EPI == EPI_GEGLU) ? (wn * 32 + t) : (wn * 64 + t)) * 128;             // + per-ni constant
  auto w_off = [&](int ni) { return (EPI == EPI_GEGLU) ? ((ni >> 1) * 128 + (ni & 1) * 16) * 128 : ni * 16 * 128; };

  // operand-tile sequence q = 2*i (A_i), 2*i + 1 (B_i); slot = q % 5
  int qslot_issue = 0;  // slot of the next operand tile to be issued
  auto issue_a = [&]() {
    char* dst = smem + qslot_issue * SLOT3_BYTES + wid * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseA + offA[j]),
                                       (__attribute__((address_space(3))) void*)(dst + j * 8192), 16, 0, 0);
    baseA += 128;
    qslot_issue = qslot_issue == SLOTS3 - 1 ? 0 : qslot_issue + 1;
  };
  auto issue_b = [&]() {
    char* dst = smem + qslot_issue * SLOT3_BYTES + wid * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseB[j] + offB),
                                       (__attribute__((address_space(3))) void*)(dst + j * 8192), 16, 0, 0);
      baseB[j] += 128;
    }
    qslot_issue = qslot_issue == SLOTS3 - 1 ? 0 : qslot_issue + 1;
  };
  auto read_frags = [&](const char* sa, const char* sb, int h, bf16x8 (&wf)[4], bf16x8 (&xf)[8]) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) wf[ni] = *reinterpret_cast<const bf16x8*>(sb + rowW + w_off(ni) + fsw[h]);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) xf[mi] = *reinterpret_cast<const bf16x8*>(sa + rowX + mi * 2048 + fsw[h]);
  };
  auto mma = [&](const bf16x8 (&wf)[4], const bf16x8 (&xf)[8]) {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
  };
  // One half-step: 32 MFMAs on `cur`, the 12 fragment reads of the NEXT half-step (from sa/sb, half h) and the 4 LDS-DMA
  // ops of one operand tile, as 16 pinned groups of {2 MFMA, 1 memory instruction}.  WHAT: 0 = issue an A tile, 1 = a B
  // tile, 2 = nothing left to issue.
  auto half_step = [&](const bf16x8 (&cur_w)[4], const bf16x8 (&cur_x)[8], bf16x8 (&nxt_w)[4], bf16x8 (&nxt_x)[8],
                       const char* sa, const char* sb, int h, bool do_read, int what) {
    char* dst = smem + qslot_issue * SLOT3_BYTES + wid * 1024;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int idx = 2 * j + hh, mi = idx >> 2, ni = idx & 3;
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur_w[ni], cur_x[mi], acc[ni][mi], 0, 0, 0);
      }
      if (j < 4) {
        if (do_read) nxt_w[j] = *reinterpret_cast<const bf16x8*>(sb + rowW + w_off(j) + fsw[h]);
      } else if (j < 12) {
        if (do_read) nxt_x[j - 4] = *reinterpret_cast<const bf16x8*>(sa + rowX + (j - 4) * 2048 + fsw[h]);
      } else {
        const int jj = j - 12;
        if (what == 0)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseA + offA[jj]),
                                           (__attribute__((address_space(3))) void*)(dst + jj * 8192), 16, 0, 0);
        else if (what == 1)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseB[jj] + offB),
                                           (__attribute__((address_space(3))) void*)(dst + jj * 8192), 16, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (what == 0) {
      baseA += 128;
    } else if (what == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) baseB[j] += 128;
    }
    if (what != 2) qslot_issue = qslot_issue == SLOTS3 - 1 ? 0 : qslot_issue + 1;
  };

  // ---- prologue: A0 B0 [A1 B1]; fragments of (0,0) ----
  issue_a();
  issue_b();
  if (nk > 1) { issue_a(); issue_b(); WAIT_VM(8); } else { WAIT_VM(0); }
  __builtin_amdgcn_s_barrier();
  bf16x8 wfA[4], xfA[8], wfB[4], xfB[8];
  int sa = 0, sb = 1;  // slots of A_i, B_i
  read_frags(smem + sa * SLOT3_BYTES, smem + sb * SLOT3_BYTES, 0, wfA, xfA);
  for (int i = 0; i < nk; ++i) {
    const char* pa = smem + sa * SLOT3_BYTES;
    const char* pb = smem + sb * SLOT3_BYTES;
    const int sa1 = sa + 2 >= SLOTS3 ? sa + 2 - SLOTS3 : sa + 2, sb1 = sb + 2 >= SLOTS3 ? sb + 2 - SLOTS3 : sb + 2;
    const bool more2 = i + 2 < nk, more1 = i + 1 < nk;
    // (i,0): MFMAs of half 0, fragments of half 1 (same tile), A(i+2)
    half_step(wfA, xfA, wfB, xfB, pa, pb, 1, true, more2 ? 0 : 2);
    WAIT_LGKM0();
    if (more1) { if (more2) WAIT_VM(4); else WAIT_VM(0); }
    __builtin_amdgcn_s_barrier();
    // (i,1): MFMAs of half 1, fragments of (i+1,0), B(i+2)
    half_step(wfB, xfB, wfA, xfA, smem + sa1 * SLOT3_BYTES, smem + sb1 * SLOT3_BYTES, 0, more1, more2 ? 1 : 2);
    sa = sa1;
    sb = sb1;
  }
  gemm_epilogue<EPI, 8>(p, Cout, acc, m0 + wm * 128, n0 + wn * 64, n0 + wn * 32, g, t);
}

#define WAIT_LGKM(n) __builtin_amdgcn_s_waitcnt(0xC07F | ((n) << 8))

// =====================================================================================================================
// gemm256v_kernel: the four-wave 256 x 256 NT kernel (production for every launch that fills the chip; gemm256p_kernel below is its
// persistent / grouped form).  ONE wave per SIMD, 128 x 128 per wave (2 x 2 waves, 8 x 8 accumulator tiles = 256 AGPRs): 64 KiB
// instead of 96 KiB of fragment reads per 32-deep half-step and CU against the eight-wave kernels -- the LDS pipe (DMA writes +
// fragment reads) is what bounds their main loop (profiles/r2_experiments.md section 5).  Same LDS image (BK = 64, 128-byte rows,
// five 32 KiB slots cycled A0 B0 A1 B1 ...), barrier protocol and per-accumulator MFMA order as gemm256b_kernel: bit-identical
// results.  With a single wave per SIMD nothing hides a stall of that wave, so: every wait is explicit (the compiler's own wait
// insertion degrades to vmcnt(0) / lgkmcnt(0) next to LDS-DMA), the accumulators are pinned in the AGPRs by issuing the MFMAs as
// inline asm ("+a"), the steady-state loop is free of branches, and
// (a) LDS-DMA goes through buffer descriptors -- `buffer_load_dwordx4 v, s[rsrc], s_off offen lds`: the K position travels in the
//     descriptor's base address (scalar adds), no per-load 64-bit VALU address add --
// (b) every gap between two MFMAs carries AT MOST ONE memory instruction, placed by a compile-time table (vs_read / vs_dma): a
//     16x16x32 MFMA occupies the matrix pipe for 16 cycles, and whatever the wave has to issue between two MFMAs beyond ~12 cycles
//     idles it.  (Round 2's four-wave kernel gemm256w -- bursts of {address add, M0, 2 ds_read_b128, LDS-DMA} after every 8th MFMA:
//     matrix pipe 56 % busy, profiles/pmc/r2_gemm256w_nt_qkv_b128.txt -- and the four other placements measured in round 3,
//     profiles/r3_gemm_sched_ab.txt, all within 1 % of each other, left the source in round 5.)
// Placement (SCHED 3): per 8 MFMAs a fragment read after the 1st and after the 4th, the LDS-DMA op after the 6th.
// =====================================================================================================================
__host__ __device__ constexpr int vs_read(int S, int i, int which) {  // fragment read `which` (0/1) issued after MFMA i, or -1
  return which != 0 ? -1 : (i & 7) == 0 ? 2 * (i >> 3) : (i & 7) == 3 ? 2 * (i >> 3) + 1 : -1;
}
__host__ __device__ constexpr int vs_dma(int S, int i) {  // LDS-DMA op issued after MFMA i, or -1
  return (i & 7) == 5 ? i >> 3 : -1;
}

#ifdef OP_GEMM_TIMELINE  // tools/gemm_timeline.py only: per-workgroup time stamps (100 MHz s_memrealtime) of gemm256v_kernel
__device__ unsigned long long* g_timeline = nullptr;
#define TL_MARK(k)                                                                                       \
  do {                                                                                                   \
    if (g_timeline && threadIdx.x == 0) g_timeline[(int64_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define TL_MARK(k)
#endif

template <int EPI, int SCHED>
__global__ __launch_bounds__(256) void gemm256v_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TL_MARK(1);
#ifdef OP_GEMM_TIMELINE
  if (g_timeline && threadIdx.x == 0)
    g_timeline[(int64_t)blockIdx.x * 8] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int g = lane >> 4, t = lane & 15;
  constexpr int BN_OUT = (EPI == EPI_GEGLU) ? 128 : 256;
  constexpr bool VMAP = EPI == EPI_BIAS || epi_is_resid(EPI);  // ends in epilogue_v: its operand order and column map

  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int GM = p.gm;
  const int per_group = GM * p.tiles_n;
  const int first_m = (pid / per_group) * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int in_group = pid % per_group;
  const int pid_m = first_m + in_group % gsz;
  const int pid_n = in_group / gsz;
  const int m0 = pid_m * BM2, n0 = pid_n * BN_OUT;
  const int nk = p.K / 64;

  // ---- staging: op j (0..7) of an operand tile covers LDS rows j*32 + (tid >> 3), 16-byte slot tid & 7 ----
  // Both operands go through buffer descriptors whose BASE is the operand's first element of the K-tile being fetched and whose
  // size is what is left of the matrix from there (0 for K-tiles past the end: the loop below issues tiles i+2 unconditionally,
  // those fetch nothing).  Activations: per-lane byte offset of the (clamped) row in a VGPR per op; weights: per-lane offset of
  // the lane's row inside a 32-row group in ONE VGPR + the group's first row as an SGPR offset per op.
  const int srow = tid >> 3;                       // 0..31
  const int sc = (tid & 7) ^ (srow & 7);
  unsigned offA[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int gm = min(m0 + j * 32 + srow, p.M - 1);
    offA[j] = (unsigned)(((int64_t)gm * p.lda + sc * 8) * 2);
  }
  const int nrecA = (int)((((int64_t)p.M - 1) * p.lda + p.K) * 2);
  unsigned offB;
  unsigned soffB[8];
  const char* ptrB[2];
  {
    const int seg = (EPI == EPI_GEGLU) ? 0 : n0 / p.n_seg;
    // (VMAP, the column map of epilogue_v: LDS row j*32 + srow = wave (j >> 2), block (j >> 1) & 1, ni (j & 1)*2 + (srow >> 4),
    // t = srow & 15  ->  column (j >> 2)*128 + t*8 + block*4 + ni)
    const int lanecol = VMAP ? (srow & 15) * 8 + (srow >> 4)
                             : (EPI == EPI_GEGLU) ? (((srow & 15) >> 2) * 8 + (srow >> 4) * 4 + (srow & 3))
                                                  : (((srow & 15) >> 2) * 16 + (srow >> 4) * 4 + (srow & 3));
    offB = (unsigned)(((int64_t)lanecol * p.ldb + sc * 8) * 2);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col0 = (EPI == EPI_GEGLU) ? n0 + (j & 3) * 32
                       : n0 - seg * p.n_seg + (VMAP ? (j >> 2) * 128 + ((j >> 1) & 1) * 4 + (j & 1) * 2 : (j >> 1) * 64 + (j & 1) * 8);
      soffB[j] = (unsigned)((int64_t)col0 * p.ldb * 2);
    }
    ptrB[0] = (const char*)((EPI == EPI_GEGLU) ? p.B[0] : p.B[seg]);
    ptrB[1] = (const char*)((EPI == EPI_GEGLU) ? p.B[1] : p.B[seg]);
  }
  const int rowsB = (EPI == EPI_GEGLU) ? p.N : min(p.n_seg, p.N);
  const int nrecB = (int)((((int64_t)rowsB - 1) * p.ldb + p.K) * 2);
  int ktA = 0, ktB = 0;  // next K-tile of each operand to be fetched

  f32x4 acc[2][4][8];  // [64-column block][ni][mi]; cleared while the first operand tiles are in flight (below)

  const int fsw[2] = {((0 * 4 + g) ^ (t & 7)) << 4, ((1 * 4 + g) ^ (t & 7)) << 4};
  const int rowX = (wm * 128 + t) * 128;  // + mi * 2048
  auto w_off = [&](int f) {  // byte offset of weight fragment f = blk * 4 + ni inside the operand tile
    const int blk = f >> 2, ni = f & 3;
    if (EPI == EPI_GEGLU) return ((ni >> 1) * 128 + (wn * 2 + blk) * 32 + (ni & 1) * 16 + t) * 128;
    return (wn * 128 + blk * 64 + ni * 16 + t) * 128;
  };

  int qslot_issue = 0;
  struct Rsrc { __amdgpu_buffer_rsrc_t a, b0, b1; };
  auto rsrc_a = [&]() {  // descriptor of activation K-tile ktA (empty past the end)
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + (int64_t)ktA * 128), 0, ktA < nk ? nrecA - ktA * 128 : 0, 0x00020000);
  };
  auto rsrc_b = [&](int which) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(ptrB[which] + (int64_t)ktB * 128), 0, ktB < nk ? nrecB - ktB * 128 : 0, 0x00020000);
  };
  auto dma_a = [&](const __amdgpu_buffer_rsrc_t& r, char* dst, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, offA[j], 0, 0, 0);
  };
  auto dma_b = [&](const __amdgpu_buffer_rsrc_t& r0, const __amdgpu_buffer_rsrc_t& r1, char* dst, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds((EPI == EPI_GEGLU && j >= 4) ? r1 : r0, (__attribute__((address_space(3))) void*)(dst + j * 4096), 16,
                                             offB, soffB[j], 0, 0);
  };
  auto advance = [&](bool is_b) {
    if (is_b) ++ktB; else ++ktA;
    qslot_issue = qslot_issue == SLOTS3 - 1 ? 0 : qslot_issue + 1;
  };
  auto issue_tile = [&](bool is_b) {
    char* dst = smem + qslot_issue * SLOT3_BYTES + wid * 1024;
    const __amdgpu_buffer_rsrc_t ra = rsrc_a(), rb0 = rsrc_b(0), rb1 = rsrc_b(1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (is_b) dma_b(rb0, rb1, dst, j); else dma_a(ra, dst, j);
    }
    advance(is_b);
  };

  // One half-step: 64 MFMAs on `cur` (complete), the 16 fragment reads of the next half-step into `nxt` and the 8 LDS-DMA ops of
  // one operand tile (IS_B: a weight tile, else an activation tile), each between the two MFMAs the SCHED table names.
  auto half_step = [&](auto b_tag, const bf16x8 (&cur_w)[8], const bf16x8 (&cur_x)[8], bf16x8 (&nxt_w)[8], bf16x8 (&nxt_x)[8],
                       const char* sa, const char* sb, int h) {
    constexpr bool IS_B = decltype(b_tag)::value;
    char* dst = smem + qslot_issue * SLOT3_BYTES + wid * 1024;
    const __amdgpu_buffer_rsrc_t r0 = IS_B ? rsrc_b(0) : rsrc_a(), r1 = (IS_B && EPI == EPI_GEGLU) ? rsrc_b(1) : r0;
    auto rd = [&](int r) {  // r = 0..15: weight fragments first (the first MFMAs of the next half-step need all eight of them)
      if (r < 8) nxt_w[r] = *reinterpret_cast<const bf16x8*>(sb + w_off(r) + fsw[h]);
      else nxt_x[r - 8] = *reinterpret_cast<const bf16x8*>(sa + rowX + (r - 8) * 2048 + fsw[h]);
    };
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const int k = i >> 3, f = i & 7;
      if constexpr (VMAP) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[f >> 2][f & 3][k]) : "v"(cur_x[k]), "v"(cur_w[f]));
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[f >> 2][f & 3][k]) : "v"(cur_w[f]), "v"(cur_x[k]));
      const int rd0 = vs_read(SCHED, i, 0), rd1 = vs_read(SCHED, i, 1), d = vs_dma(SCHED, i);
      if (rd0 >= 0 || rd1 >= 0 || d >= 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (rd0 >= 0) rd(rd0);
        if (rd1 >= 0) rd(rd1);
        if (d >= 0) { if (IS_B) dma_b(r0, r1, dst, d); else dma_a(r0, dst, d); }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    advance(IS_B);
  };
  using IsA = std::integral_constant<bool, false>;
  using IsB = std::integral_constant<bool, true>;

  // ---- prologue: A0 B0 A1 B1 (nk >= 2); fragments of (0,0) ----
  // (tools/gemm_timeline.py: 2.6 ... 4.1 us from kernel entry to the first fragments, 0.6 of it index math; the rest is the latency
  // of the first operand tiles when all 256 CUs start a round together -- issuing the second tile pair behind the accumulator
  // clear instead of in front of it changed nothing, round 3)
  TL_MARK(7);
  issue_tile(false);
  issue_tile(true);
  issue_tile(false);
  issue_tile(true);
  {  // clear the accumulators UNDER the latency of the loads above: 64 MFMAs 0 x 0 + 0 (as compiler-generated v_accvgpr_write
     // the 256 writes are rematerialised constants the scheduler places wherever it likes -- it put them behind the wait)
    const bf16x8 zf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, 0" : "=a"(acc[a][b][c]) : "v"(zf));
  }
  TL_MARK(2);
  WAIT_VM(16);
  __builtin_amdgcn_s_barrier();
  TL_MARK(3);
  bf16x8 wfA[8], xfA[8], wfB[8], xfB[8];
  int sa = 0, sb = 1;
#pragma unroll
  for (int r = 0; r < 8; ++r) wfA[r] = *reinterpret_cast<const bf16x8*>(smem + sb * SLOT3_BYTES + w_off(r) + fsw[0]);
#pragma unroll
  for (int r = 0; r < 8; ++r) xfA[r] = *reinterpret_cast<const bf16x8*>(smem + sa * SLOT3_BYTES + rowX + r * 2048 + fsw[0]);
  // ONE loop over all K-tiles, no peeled tail: tile i issues tiles i+2 also when they do not exist (empty descriptors: nothing is
  // fetched) and its second half-step reads the "next tile's" fragments also when there is none (never used).  A peeled tail made
  // the register allocator shuffle accumulators (v_accvgpr_read / _mov) on the loop-exit edge, i.e. right behind inline-asm MFMAs
  // it cannot see: those copies read accumulators whose MFMAs were still in flight (tools/check_mfma_hazards.py looks for that).
  int nk_last = nk;
  asm volatile("" : "+s"(nk_last));
  for (int i = 0; i < nk; ++i) {
    const int sa1 = sa + 2 >= SLOTS3 ? sa + 2 - SLOTS3 : sa + 2, sb1 = sb + 2 >= SLOTS3 ? sb + 2 - SLOTS3 : sb + 2;
    WAIT_LGKM(0);
    asm volatile("s_nop 0");
    __builtin_amdgcn_sched_barrier(0);
    half_step(IsA{}, wfA, xfA, wfB, xfB, smem + sa * SLOT3_BYTES, smem + sb * SLOT3_BYTES, 1);   // (i,0): A(i+2)
    WAIT_LGKM(0);
    WAIT_VM(8);
    __builtin_amdgcn_s_barrier();
    half_step(IsB{}, wfB, xfB, wfA, xfA, smem + sa1 * SLOT3_BYTES, smem + sb1 * SLOT3_BYTES, 0);  // (i,1): B(i+2)
    sa = sa1;
    sb = sb1;
    // The accumulators are written by inline-asm MFMAs the hazard recogniser does not see, and on the loop-exit edge the register
    // allocator reads / moves accumulators (v_accvgpr_read / _mov; neither a "memory" clobber nor operand dependences of a later
    // statement keep those copies away from the edge): the last MFMAs retire INSIDE the loop body, on the last trip only.  The
    // trip test goes through a value the optimiser cannot equate with the exit condition, or it would sink the statement
    // behind the edge again.
    if (i + 1 >= nk_last) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
  }
  // the empty fetches and the unused fragment reads of the last tile must be gone before the registers / the LDS get their next owner
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  TL_MARK(4);
  if constexpr (EPI == EPI_BIAS || epi_is_resid(EPI)) {
    epilogue_v<EPI>(p, acc, m0 + wm * 128, n0 + wn * 128, g, t);
  } else {
    gemm_epilogue<EPI, 8>(p, p.C, acc[0], m0 + wm * 128, n0 + wn * 128, n0 + (wn * 2) * 32, g, t);
    gemm_epilogue<EPI, 8>(p, p.C, acc[1], m0 + wm * 128, n0 + wn * 128 + 64, n0 + (wn * 2 + 1) * 32, g, t);
  }
  TL_MARK(5);
#ifdef OP_GEMM_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TL_MARK(6);
#endif
}

// =====================================================================================================================
// gemm256p_kernel: PERSISTENT, GROUPED form of gemm256v_kernel (schedule 3).  One workgroup per CU walks the tile list
// b = blockIdx, blockIdx + gridDim, ... (the same tile -> CU assignment the hardware dispatcher produces for one-tile
// workgroups, so the panels shared through L2 and the K-lockstep of a round are unchanged), and the K-tile stream never stops
// at a tile boundary: the loop that issues K-tile i+2 of the current tile issues K-tiles 0 and 1 of the NEXT tile during its
// last two trips, and the last half-step reads the next tile's first fragments.  What a one-tile workgroup pays per tile --
// dispatch, kernel-argument loads, address set-up, the latency of the first operand tiles, and an epilogue whose stores have
// to drain before the CU gets its next workgroup (K-scan of round 2's one-tile four-wave kernel: 26 us per three-round launch, 20 % of a
// K = 1536 launch) -- is paid once per launch or overlaps with the next tile's main loop.
// Grouped: the tile list may span up to three PROBLEMS that share N, K and the epilogue but have their own activation
// matrix, row count, weights, bias, layer-scale vector, residual and outputs -- the three modality FFNs of an encoder layer as
// ONE launch (a text pass alone is 192 tiles on 256 CUs), transformer_layer.py:203-226.  M-tiles of the problems are
// concatenated; rows past a problem's end are fetched as zeros (buffer descriptor sized to the problem) and never stored.
// =====================================================================================================================
struct GroupArgs {
  int nprob;
  const bf16_t* A[3]; int M[3]; int mt_end[3];  // mt_end: running sum of 256-row tiles (entries >= nprob - 1 hold tiles_m)
  const bf16_t* B[3][3];                         // [problem][weight segment along N]; GeGLU: 0 = wi_0, 1 = wi_1
  const bf16_t* bias[3][3];
  void* C[3]; bf16_t* H0[3]; bf16_t* H1[3];
  const bf16_t* resid[3]; const bf16_t* gamma[3]; const float* rowscale[3]; int rows_per_sample[3];
  const int* rows[3];  // EPI_RESID_ROWS (see gemm_epilogue_v.h)
  int64_t lda, ldb, ldc, ldr;
  const float* alpha;
  int n_seg, N, K, tiles_m, tiles_n, gm;
};

template <class T>
__device__ __forceinline__ T sel3(const T (&a)[3], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : a[2]; }

template <int EPI>
__global__ __launch_bounds__(256) void gemm256p_kernel(const GroupArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SCHED = 3;
  static_assert(EPI == EPI_BIAS || epi_is_resid(EPI), "gemm256p_kernel: plain / bias and residual epilogues (epilogue_v)");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int g = lane >> 4, t = lane & 15;
  constexpr int BN_OUT = 256;
  const int nk = p.K / 64;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int stride_b = gridDim.x;

  // ---- tile b -> problem, first row, first column; descriptors of its operand panels (all wave-uniform) ----
  struct Tile { int prob, m0, n0; const char* a; int na; const char* b0; int nb; };
  auto locate = [&](int b) {
    Tile T;
    const bool valid = b < ntiles;
    const int pid = xcd_remap(valid ? b : 0, ntiles);
    const int per_group = p.gm * p.tiles_n;
    const int first_m = (pid / per_group) * p.gm;
    const int gsz = min(p.tiles_m - first_m, p.gm);
    const int in_group = pid % per_group;
    const int pid_m = first_m + in_group % gsz, pid_n = in_group / gsz;
    T.prob = (pid_m >= p.mt_end[0] ? 1 : 0) + (pid_m >= p.mt_end[1] ? 1 : 0);
    T.m0 = (pid_m - (T.prob == 0 ? 0 : T.prob == 1 ? p.mt_end[0] : p.mt_end[1])) * BM2;
    T.n0 = pid_n * BN_OUT;
    const int Mp = sel3(p.M, T.prob);
    T.a = (const char*)(sel3(p.A, T.prob) + (int64_t)T.m0 * p.lda);
    T.na = valid ? (int)((((int64_t)Mp - 1 - T.m0) * p.lda + p.K) * 2) : 0;  // bytes from the tile's first row to the matrix end
    const int seg = T.n0 / p.n_seg, c0 = T.n0 - seg * p.n_seg;
    const bf16_t* w = T.prob == 0 ? (seg == 0 ? p.B[0][0] : seg == 1 ? p.B[0][1] : p.B[0][2])
                    : T.prob == 1 ? (seg == 0 ? p.B[1][0] : seg == 1 ? p.B[1][1] : p.B[1][2])
                                  : (seg == 0 ? p.B[2][0] : seg == 1 ? p.B[2][1] : p.B[2][2]);
    T.b0 = (const char*)(w + (int64_t)c0 * p.ldb);
    T.nb = valid ? (int)((((int64_t)min(p.n_seg, p.N) - 1 - c0) * p.ldb + p.K) * 2) : 0;
    return T;
  };

  // ---- staging (see gemm256v_kernel): op j covers LDS rows j*32 + (tid >> 3); nothing here depends on the tile ----
  const int srow = tid >> 3;
  const int sc = (tid & 7) ^ (srow & 7);
  // op j of the A panel: ONE per-lane offset + a scalar offset per op (eight per-lane offsets cost seven VGPRs the epilogue of the
  // finished tile -- which runs with the next tile's first fragments live -- had to spill around)
  const unsigned offA = (unsigned)(((int64_t)srow * p.lda + sc * 8) * 2);
  unsigned soffA[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) soffA[j] = (unsigned)((int64_t)(j * 32) * p.lda * 2);
  const int lanecol = (srow & 15) * 8 + (srow >> 4);  // ends in epilogue_v: its operand order and column map
  const unsigned offB = (unsigned)(((int64_t)lanecol * p.ldb + sc * 8) * 2);
  unsigned soffB[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    soffB[j] = (unsigned)((int64_t)((j >> 2) * 128 + ((j >> 1) & 1) * 4 + (j & 1) * 2) * p.ldb * 2);

  f32x4 acc[2][4][8];  // [64-column block][ni][mi]
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  const int fsw[2] = {((0 * 4 + g) ^ (t & 7)) << 4, ((1 * 4 + g) ^ (t & 7)) << 4};
  const int rowX = (wm * 128 + t) * 128;  // + mi * 2048
  auto w_off = [&](int f) {
    const int blk = f >> 2, ni = f & 3;
    return (wn * 128 + blk * 64 + ni * 16 + t) * 128;
  };

  int qslot_issue = 0;
  auto rsrc = [&](const char* base, int nrec, int kt) {  // K-tile kt of a panel: base + kt*128 bytes, what is left of it (or nothing)
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (int64_t)kt * 128), 0, nrec > 0 ? nrec - kt * 128 : 0, 0x00020000);
  };
  auto dma_a = [&](const __amdgpu_buffer_rsrc_t& r, char* dst, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, offA, soffA[j], 0, 0);
  };
  auto dma_b = [&](const __amdgpu_buffer_rsrc_t& r0, char* dst, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, offB, soffB[j], 0, 0);
  };
  auto next_slot = [&]() { qslot_issue = qslot_issue == SLOTS3 - 1 ? 0 : qslot_issue + 1; };
  auto issue_tile = [&](bool is_b, const Tile& T, int kt) {
    char* dst = smem + qslot_issue * SLOT3_BYTES + wid * 1024;
    const __amdgpu_buffer_rsrc_t ra = rsrc(T.a, T.na, kt), rb0 = rsrc(T.b0, T.nb, kt);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (is_b) dma_b(rb0, dst, j); else dma_a(ra, dst, j);
    }
    next_slot();
  };
  auto half_step = [&](auto b_tag, const Tile& T, int kt, const bf16x8 (&cur_w)[8], const bf16x8 (&cur_x)[8], bf16x8 (&nxt_w)[8],
                       bf16x8 (&nxt_x)[8], const char* sa, const char* sb, int h) {
    constexpr bool IS_B = decltype(b_tag)::value;
    char* dst = smem + qslot_issue * SLOT3_BYTES + wid * 1024;
    const __amdgpu_buffer_rsrc_t r0 = IS_B ? rsrc(T.b0, T.nb, kt) : rsrc(T.a, T.na, kt);
    auto rd = [&](int r) {
      if (r < 8) nxt_w[r] = *reinterpret_cast<const bf16x8*>(sb + w_off(r) + fsw[h]);
      else nxt_x[r - 8] = *reinterpret_cast<const bf16x8*>(sa + rowX + (r - 8) * 2048 + fsw[h]);
    };
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const int k = i >> 3, f = i & 7;
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[f >> 2][f & 3][k]) : "v"(cur_x[k]), "v"(cur_w[f]));
      const int rd0 = vs_read(SCHED, i, 0), rd1 = vs_read(SCHED, i, 1), d = vs_dma(SCHED, i);
      if (rd0 >= 0 || rd1 >= 0 || d >= 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (rd0 >= 0) rd(rd0);
        if (rd1 >= 0) rd(rd1);
        if (d >= 0) { if (IS_B) dma_b(r0, dst, d); else dma_a(r0, dst, d); }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    next_slot();
  };
  using IsA = std::integral_constant<bool, false>;
  using IsB = std::integral_constant<bool, true>;

  int b = blockIdx.x;
  Tile cur = locate(b);
  // ---- prologue of the FIRST tile only: A0 B0 A1 B1, fragments of (0,0) ----
  issue_tile(false, cur, 0);
  issue_tile(true, cur, 0);
  issue_tile(false, cur, 1);
  issue_tile(true, cur, 1);
  WAIT_VM(16);
  __builtin_amdgcn_s_barrier();
  bf16x8 wfA[8], xfA[8], wfB[8], xfB[8];
  int sa = 0, sb = 1;
#pragma unroll
  for (int r = 0; r < 8; ++r) wfA[r] = *reinterpret_cast<const bf16x8*>(smem + sb * SLOT3_BYTES + w_off(r) + fsw[0]);
#pragma unroll
  for (int r = 0; r < 8; ++r) xfA[r] = *reinterpret_cast<const bf16x8*>(smem + sa * SLOT3_BYTES + rowX + r * 2048 + fsw[0]);
  int nk_last = nk;
  asm volatile("" : "+s"(nk_last));

  for (; b < ntiles; b += stride_b) {
    const Tile nxt = locate(b + stride_b);  // past the list: empty descriptors, nothing is fetched
    for (int i = 0; i < nk; ++i) {
      const bool wrap = i + 2 >= nk;       // K-tile i+2 of this tile, or K-tile i+2-nk of the next one
      const int kt = wrap ? i + 2 - nk : i + 2;
      const Tile& src = wrap ? nxt : cur;
      const int sa1 = sa + 2 >= SLOTS3 ? sa + 2 - SLOTS3 : sa + 2, sb1 = sb + 2 >= SLOTS3 ? sb + 2 - SLOTS3 : sb + 2;
      WAIT_LGKM(0);
      asm volatile("s_nop 0");
      __builtin_amdgcn_sched_barrier(0);
      half_step(IsA{}, src, kt, wfA, xfA, wfB, xfB, smem + sa * SLOT3_BYTES, smem + sb * SLOT3_BYTES, 1);
      WAIT_LGKM(0);
      WAIT_VM(8);
      __builtin_amdgcn_s_barrier();
      half_step(IsB{}, src, kt, wfB, xfB, wfA, xfA, smem + sa1 * SLOT3_BYTES, smem + sb1 * SLOT3_BYTES, 0);
      sa = sa1;
      sb = sb1;
      // last trip: the inline-asm MFMAs retire before the loop exit, where the register allocator may read / move accumulators
      // (see gemm256v_kernel)
      if (i + 1 >= nk_last) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    }
    {  // epilogue of the finished tile; its stores drain while the next tile's main loop runs
      GemmArgs q;
      q.M = sel3(p.M, cur.prob); q.N = p.N; q.K = p.K; q.n_seg = p.n_seg; q.ldc = p.ldc; q.ldr = p.ldr; q.m_off = 0;
      q.C = sel3(p.C, cur.prob); q.H0 = sel3(p.H0, cur.prob); q.H1 = sel3(p.H1, cur.prob);
      q.resid = sel3(p.resid, cur.prob); q.gamma = sel3(p.gamma, cur.prob); q.rowscale = sel3(p.rowscale, cur.prob);
      q.rows_per_sample = sel3(p.rows_per_sample, cur.prob); q.alpha = p.alpha; q.rows = sel3(p.rows, cur.prob);
      q.bias[0] = q.bias[1] = q.bias[2] = nullptr;  // (unused: the tile's bias vector travels as an argument, see gemm_epilogue)
      // the tile lies in ONE weight segment (n_seg is a multiple of the tile width, checked at launch): kernel-argument table look-up
      const int segt = cur.n0 / p.n_seg;
      const bf16_t* bt = p.bias[cur.prob][segt];
      epilogue_v<EPI>(q, acc, cur.m0 + wm * 128, cur.n0 + wn * 128, g, t, bt, true);
    }
    zero_acc();
    cur = nxt;
  }
  // the empty fetches and the unused fragment reads behind the last tile must be gone before the LDS / registers get their next owner
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// =====================================================================================================================
// TN variant of the 256x256 kernel:  C[M,N] = sum_k A[k][m] * B[k][n]  with BOTH operands stored K-major ([K, M] and
// [K, N] row-major) -- the weight-gradient GEMM dW = dy^T x straight from the activation matrices, no transposed copies.
// Same four-stage LDS-DMA pipeline and epilogue; operand tiles are [32 k][256] (512-byte rows) and the MFMA fragments
// (8 consecutive k for one m / n) come from ds_read_b64_tr_b16: a 16-lane group reads a [4 k][16 cols] block and each lane
// receives one column.  The column a lane receives is chosen through the addresses the provider lanes supply, which gives
// the same "16 contiguous output columns per lane" accumulator layout as the NT kernel for free.
// 16-byte slots are XOR-swizzled per k-row (different functions for the two operands: the m-operand reads are conflict
// free, the n-operand reads are 2-way by construction).
// =====================================================================================================================
__device__ __forceinline__ int swzA_tn(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 1; }

// Epilogue of the TN kernels.  Both operands' fragments hold 16 CONSECUTIVE columns (the n-operand reads of the NT-style
// "16 contiguous output columns per lane" permutation use half of every 16-byte LDS chunk and are 2-way bank conflicts by
// construction: a quarter of the kernel's LDS cycles, profiles/pmc/r2_gemm256_tn_wgrad_b128.txt), so lane (g, t) holds
// acc[f][mi][r] = C[mrow0 + mi*16 + t][nbase + f*16 + g*4 + r]: four consecutive columns per fragment.  fp32 slab (split-K),
// plain bf16 or accumulation into the existing bf16 gradient (C += acc; `resid` = C).
template <int EPI, int NF>
__device__ __forceinline__ void tn_epilogue(const GemmArgs& p, void* Cout, f32x4 (&acc)[NF][8], int mrow0, int nbase, int g, int t) {
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = mrow0 + mi * 16 + t;
    if (m >= p.M) continue;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int n = nbase + f * 16 + g * 4;
      if (n >= p.N) continue;  // N % 8 == 0: a group of four never straddles the edge
      const f32x4 a = acc[f][mi];
      if (EPI == EPI_F32) {
        // (plain stores on purpose: the fold kernel re-reads the slabs right away -- non-temporal stores measured +3 % on the launch)
        *reinterpret_cast<f32x4*>((float*)Cout + (int64_t)m * p.ldc + n) = a;
      } else {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;
        bf16_t* c = (bf16_t*)Cout + (int64_t)m * p.ldc + n;
        float o[4] = {a[0], a[1], a[2], a[3]};
        if (EPI == EPI_RESID) {
          const bf16x4v r = *reinterpret_cast<const bf16x4v*>(p.resid + (int64_t)m * p.ldr + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (float)r[j] + o[j];
        }
        bf16x4v w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (bf16_t)o[j];
        *reinterpret_cast<bf16x4v*>(c) = w;
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_tn_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int g = lane >> 4, t = lane & 15;

  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int GM = p.gm;
  const int per_group = GM * p.tiles_n;
  const int first_m = (pid / per_group) * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int in_group = pid % per_group;
  const int pid_m = first_m + in_group % gsz;
  const int pid_n = in_group / gsz;
  const int m0 = pid_m * BM2, n0 = pid_n * 256;

  int nk = p.K / BK2;
  int kt0 = 0;
  void* Cout = p.C;
  if (p.kt_per_split > 0) {
    kt0 = blockIdx.y * p.kt_per_split;
    nk = min(nk - kt0, p.kt_per_split);
    Cout = (float*)p.C + (int64_t)blockIdx.y * p.slab;
  }

  // staging: slot q = i*512 + tid -> k-row q>>5 (i = 0: rows 0..15, i = 1: 16..31), 16-byte slot q&31
  const char* baseA = (const char*)(p.A + (int64_t)kt0 * BK2 * p.lda);
  const char* baseB = (const char*)(p.B[0] + (int64_t)kt0 * BK2 * p.ldb);
  unsigned offA[2], offB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = i * 512 + tid;
    const int kr = q >> 5, pc = q & 31;
    const int ca = min(m0 + (pc ^ swzA_tn(kr)) * 8, p.M - 8);
    const int cb = min(n0 + (pc ^ swzA_tn(kr)) * 8, p.N - 8);
    offA[i] = (unsigned)(((int64_t)kr * p.lda + ca) * 2);
    offB[i] = (unsigned)(((int64_t)kr * p.ldb + cb) * 2);
  }
  const int64_t stepA = (int64_t)BK2 * p.lda * 2, stepB = (int64_t)BK2 * p.ldb * 2;

  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transpose-read provider addresses (bytes inside a stage); k-row of this lane = g*8 + h*4 + (t>>2), h = 0,1 (+2048 B)
  const int krow = g * 8 + (t >> 2);
  const int rowoff = krow * 512;
  int rdX[8], rdW[4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int chunk = wm * 16 + mi * 2 + ((t & 3) >> 1);
    rdX[mi] = rowoff + ((chunk ^ swzA_tn(krow)) << 4) + (t & 1) * 8;
  }
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int chunk = wn * 8 + ni * 2 + ((t & 3) >> 1);  // 16 consecutive n per fragment, like the m-operand: conflict free
    rdW[ni] = OPER2_BYTES + rowoff + ((chunk ^ swzA_tn(krow)) << 4) + (t & 1) * 8;
  }

  auto issue = [&](int slot) {
    char* la = smem + slot * STAGE2_BYTES;
    char* lb = la + OPER2_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int wbase = (i * 512 + wid * 64) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseA + offA[i]),
                                       (__attribute__((address_space(3))) void*)(la + wbase), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseB + offB[i]),
                                       (__attribute__((address_space(3))) void*)(lb + wbase), 16, 0, 0);
    }
    baseA += stepA;
    baseB += stepB;
  };
  auto wait_landed = [&](int younger) {
    if (younger >= 3) WAIT_VM(12);
    else if (younger == 2) WAIT_VM(8);
    else if (younger == 1) WAIT_VM(4);
    else WAIT_VM(0);
  };
  struct Frags { s16x4 w[4][2]; s16x4 x[8][2]; };
  auto read_frags = [&](int kt, Frags& f) {
    const char* st = smem + (kt & (STAGES2 - 1)) * STAGE2_BYTES;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      f.w[ni][0] = tr_read16(lds_addr(st) + rdW[ni], false);
      f.w[ni][1] = tr_read16(lds_addr(st) + rdW[ni], true);
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      f.x[mi][0] = tr_read16(lds_addr(st) + rdX[mi], false);
      f.x[mi][1] = tr_read16(lds_addr(st) + rdX[mi], true);
    }
  };
  auto mma = [&](const Frags& f) {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join16(f.w[ni][0], f.w[ni][1]), join16(f.x[mi][0], f.x[mi][1]),
                                                              acc[ni][mi], 0, 0, 0);
  };
  auto step_steady = [&](int kt, const Frags& cur, Frags& nxt) {
    WAIT_LGKM0();
    WAIT_VM(8);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);  // the MFMAs below consume asm-read fragments: nothing may move above the waits
    const char* st = smem + ((kt + 1) & (STAGES2 - 1)) * STAGE2_BYTES;
    char* la = smem + (kt & (STAGES2 - 1)) * STAGE2_BYTES;
    char* lb = la + OPER2_BYTES;
#pragma unroll
    for (int j = 0; j < 32; ++j) {  // 32 groups of {1 MFMA, <= 1 memory instruction}, order pinned
      const int mi = j >> 2, ni = j & 3;
      acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join16(cur.w[ni][0], cur.w[ni][1]), join16(cur.x[mi][0], cur.x[mi][1]),
                                                            acc[ni][mi], 0, 0, 0);
      if (j < 8) {
        nxt.w[j >> 1][j & 1] = tr_read16(lds_addr(st) + rdW[j >> 1], (j & 1) != 0);
      } else if (j < 24) {
        const int e = j - 8;
        nxt.x[e >> 1][e & 1] = tr_read16(lds_addr(st) + rdX[e >> 1], (e & 1) != 0);
      } else if (j < 28) {
        const int i = (j - 24) >> 1;
        const int wbase = (i * 512 + wid * 64) * 16;
        if ((j & 1) == 0)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseA + offA[i]),
                                           (__attribute__((address_space(3))) void*)(la + wbase), 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseB + offB[i]),
                                           (__attribute__((address_space(3))) void*)(lb + wbase), 16, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    baseA += stepA;
    baseB += stepB;
  };
  auto step_tail = [&](int kt, const Frags& cur, Frags& nxt) {
    const bool has_next = kt + 1 < nk;
    WAIT_LGKM0();
    if (has_next) wait_landed(min(nk - 2 - kt, STAGES2 - 2));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) read_frags(kt + 1, nxt);
    mma(cur);
    __builtin_amdgcn_sched_barrier(0);
  };

#pragma unroll
  for (int s0 = 0; s0 < STAGES2; ++s0)
    if (s0 < nk) issue(s0);
  Frags fA, fB;
  wait_landed(min(nk - 1, STAGES2 - 1));
  __builtin_amdgcn_s_barrier();
  read_frags(0, fA);
  int kt = 0;
  for (; kt + STAGES2 + 1 < nk; kt += 2) {
    step_steady(kt, fA, fB);
    step_steady(kt + 1, fB, fA);
  }
  for (; kt < nk; kt += 2) {
    step_tail(kt, fA, fB);
    step_tail(kt + 1, fB, fA);
  }
  tn_epilogue<EPI, 4>(p, Cout, acc, m0 + wm * 128, n0 + wn * 64, g, t);
}

// Four-wave flavour of the TN kernel (one wave per SIMD, 128 x 128 per wave; see gemm256v_kernel): the same four-stage
// [32 k][256] LDS image and transpose-read fragments, 32 ds_read_b64_tr_b16 per 64 MFMAs and wave instead of 24 per 32, i.e. a
// third fewer LDS bytes per MFMA; accumulators pinned in AGPRs (inline-asm MFMA), steady state unrolled over both fragment sets.
template <int EPI>
__global__ __launch_bounds__(256) void gemm256w_tn_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
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
  const int m0 = pid_m * BM2, n0 = pid_n * 256;

  int nk = p.K / BK2;
  int kt0 = 0;
  void* Cout = p.C;
  if (p.kt_per_split > 0) {
    kt0 = blockIdx.y * p.kt_per_split;
    nk = min(nk - kt0, p.kt_per_split);
    Cout = (float*)p.C + (int64_t)blockIdx.y * p.slab;
  }

  // staging: slot q = i*256 + tid (i = 0..3) -> k-row q>>5, 16-byte slot q&31
  const char* baseA = (const char*)(p.A + (int64_t)kt0 * BK2 * p.lda);
  const char* baseB = (const char*)(p.B[0] + (int64_t)kt0 * BK2 * p.ldb);
  unsigned offA[4], offB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + tid;
    const int kr = q >> 5, pc = q & 31;
    const int ca = min(m0 + (pc ^ swzA_tn(kr)) * 8, p.M - 8);
    const int cb = min(n0 + (pc ^ swzA_tn(kr)) * 8, p.N - 8);
    offA[i] = (unsigned)(((int64_t)kr * p.lda + ca) * 2);
    offB[i] = (unsigned)(((int64_t)kr * p.ldb + cb) * 2);
  }
  const int64_t stepA = (int64_t)BK2 * p.lda * 2, stepB = (int64_t)BK2 * p.ldb * 2;

  f32x4 acc[2][4][8];  // [64-column block][ni][mi]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int krow = g * 8 + (t >> 2);
  const int rowoff = krow * 512;
  int rdX[8], rdW[8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int chunk = wm * 16 + mi * 2 + ((t & 3) >> 1);
    rdX[mi] = rowoff + ((chunk ^ swzA_tn(krow)) << 4) + (t & 1) * 8;
  }
#pragma unroll
  for (int f = 0; f < 8; ++f) {  // f = blk * 4 + ni
    const int chunk = wn * 16 + f * 2 + ((t & 3) >> 1);  // 16 consecutive n per fragment, like the m-operand: conflict free
    rdW[f] = OPER2_BYTES + rowoff + ((chunk ^ swzA_tn(krow)) << 4) + (t & 1) * 8;
  }

  auto dma = [&](char* la, int j) {  // op j = 0..7 of a stage: A ops 0..3, B ops 4..7
    const int i = j & 3;
    const int wbase = (i * 256 + wid * 64) * 16;
    if (j < 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseA + offA[i]),
                                       (__attribute__((address_space(3))) void*)(la + wbase), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseB + offB[i]),
                                       (__attribute__((address_space(3))) void*)(la + OPER2_BYTES + wbase), 16, 0, 0);
  };
  auto issue = [&](int slot) {
#pragma unroll
    for (int j = 0; j < 8; ++j) dma(smem + slot * STAGE2_BYTES, j);
    baseA += stepA;
    baseB += stepB;
  };
  auto wait_landed = [&](int younger) {  // 8 LDS-DMA ops per stage and wave
    if (younger >= 3) WAIT_VM(24);
    else if (younger == 2) WAIT_VM(16);
    else if (younger == 1) WAIT_VM(8);
    else WAIT_VM(0);
  };
  struct Frags { s16x4 w[8][2]; s16x4 x[8][2]; };
  auto rd = [&](const char* st, Frags& f, int r) {  // r = 0..31: the 16 weight-side halves first, then the activation side
    if (r < 16) f.w[r >> 1][r & 1] = tr_read16(lds_addr(st) + rdW[r >> 1], (r & 1) != 0);
    else f.x[(r - 16) >> 1][r & 1] = tr_read16(lds_addr(st) + rdX[(r - 16) >> 1], (r & 1) != 0);
  };
  auto mfma1 = [&](const Frags& f, int j) {  // j = 0..63: activation fragment j>>3, weight fragment j&7
    const int mi = j >> 3, ff = j & 7;
    const bf16x8 wv = join16(f.w[ff][0], f.w[ff][1]), xv = join16(f.x[mi][0], f.x[mi][1]);
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[ff >> 2][ff & 3][mi]) : "v"(wv), "v"(xv));
  };
  // steady step: 64 MFMAs on `cur`, the 32 transpose reads of stage kt+1 into `nxt`, the refill of this stage's slot with stage
  // kt+4 -- one memory instruction per MFMA in the first 40 of the 64 groups (reads first, they are waited for next)
  auto step_steady = [&](int kt, const Frags& cur, Frags& nxt) {
    WAIT_LGKM(0);
    WAIT_VM(16);
    __builtin_amdgcn_s_barrier();
    const char* st = smem + ((kt + 1) & (STAGES2 - 1)) * STAGE2_BYTES;
    char* la = smem + (kt & (STAGES2 - 1)) * STAGE2_BYTES;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      mfma1(cur, j);
      if (j < 32) rd(st, nxt, j);
      else if (j < 40) dma(la, j - 32);
      if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    baseA += stepA;
    baseB += stepB;
  };
  auto step_tail = [&](int kt, const Frags& cur, Frags& nxt) {
    const bool has_next = kt + 1 < nk;
    WAIT_LGKM(0);
    if (has_next) wait_landed(min(nk - 2 - kt, STAGES2 - 2));
    __builtin_amdgcn_s_barrier();
    if (has_next) {
      const char* st = smem + ((kt + 1) & (STAGES2 - 1)) * STAGE2_BYTES;
#pragma unroll
      for (int r = 0; r < 32; ++r) rd(st, nxt, r);
    }
#pragma unroll
    for (int j = 0; j < 64; ++j) mfma1(cur, j);
  };

#pragma unroll
  for (int s0 = 0; s0 < STAGES2; ++s0)
    if (s0 < nk) issue(s0);
  Frags fA, fB;
  wait_landed(min(nk - 1, STAGES2 - 1));
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int r = 0; r < 32; ++r) rd(smem, fA, r);
  int kt = 0;
  for (; kt + STAGES2 + 1 < nk; kt += 2) {
    step_steady(kt, fA, fB);
    step_steady(kt + 1, fB, fA);
  }
  for (; kt < nk; kt += 2) {
    step_tail(kt, fA, fB);
    step_tail(kt + 1, fB, fA);
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // the inline-asm MFMAs are invisible to the hazard recogniser
  tn_epilogue<EPI, 4>(p, Cout, acc[0], m0 + wm * 128, n0 + wn * 128, g, t);
  tn_epilogue<EPI, 4>(p, Cout, acc[1], m0 + wm * 128, n0 + wn * 128 + 64, g, t);
}

template <int EPI>
int launch256_tn(const GemmArgs& a, hipStream_t s, int splits, bool four_waves) {
  const dim3 grid(a.tiles_m * a.tiles_n, splits);
  const size_t sh = STAGES2 * STAGE2_BYTES;
  OP_ENSURE_LDS((gemm256_tn_kernel<EPI>), (int)sh, "gemm_tn");
  if (four_waves) {
    OP_ENSURE_LDS((gemm256w_tn_kernel<EPI>), (int)sh, "gemm_tn");
    hipLaunchKernelGGL((gemm256w_tn_kernel<EPI>), grid, dim3(256), sh, s, a);
  } else {
    hipLaunchKernelGGL((gemm256_tn_kernel<EPI>), grid, dim3(512), sh, s, a);
  }
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// =====================================================================================================================
// gemm256w_tn_grouped_kernel: ALL weight gradients of an encoder layer as ONE persistent launch, no split-K.
// A weight-gradient GEMM has few output tiles (36 ... 288 of 256 x 256) and a very long K (the token count: 8 192 ... 73 216),
// so a launch per weight needs split-K to fill 256 CUs: fp32 slabs (the merged wi_0|wi_1 gradient wrote 7 slabs = 528 MB to
// produce a 37.7 MB gradient) and a fold kernel that reads them back (20 ms of the 4B step).  Here up to TN_MAX_PROB problems
// (own operands, sizes, K and output) form ONE tile list: a layer of the lock-step pass is q|k|v, out-proj and wi_0|wi_1, wo of
// three modalities = 1440 tiles, 5.6 per CU, walked by one workgroup per CU -- every tile runs its WHOLE K and accumulates
// straight into the bf16 gradient.
// Scheduling (round 5; host: tn_build_schedule): EIGHT queues, one per XCD; a workgroup draws from the queue of its XCD (blockIdx & 7;
// speed only, any mapping is correct) with one returning atomic per tile -- issued inside the epilogue of the tile it follows -- and
// steals from the other queues when its own is empty.  What shares an operand panel through an XCD's L2 is what that XCD's
// workgroups run AT THE SAME TIME in K-lockstep, so the unit dealt to a queue is a WAVE: as many consecutive slots of a problem's
// slot order (tn_geom: a rectangle of its tile grid, e.g. 5.3 rows x 6 columns) as the XCD has workgroups.  32 tiles of a wave read 6
// panels of the narrow operand and 6 of the wide one instead of 64: 0.375 panels per tile.  Waves never mix K: tiles of one K
// finish together and the workgroups take the next wave together (problems of equal K are cut as one list); waves are dealt
// longest K first to the queue with the least work (LPT), which ends within a few per cent of the ideal makespan.
// Round 4 dealt six-tile GROUPS round-robin over the queues: an XCD then held groups of different problems side by side -- nothing
// to share between them -- and fetched 16.9 GB per layer for 4.5 GB of operands (profiles/r4_gemm_hbm_traffic.json); its two odd
// workgroups per XCD (32 = 5 groups + 2) drew single tiles from the BACK of the queue ("solo", still available: tune bit 10).
// Main loop, LDS image, fragment order = gemm256w_tn_kernel (bit-identical per-tile results to its unsplit launch).
// =====================================================================================================================
constexpr int TN_MAX_PROB = 16;
constexpr int TN_MAX_RUNS = 160;
constexpr int TN_CTR_STRIDE = 8;   // 64-bit counters 64 bytes apart: queues 0..7 (claims from the front | from the back << 32), then the exit counter
struct TnProb {
  const bf16_t* A; const bf16_t* B; bf16_t* C;  // C[M,N] (+)= A[K,M]^T B[K,N]
  int64_t lda, ldb, ldc;
  int M, N, K, tiles_m, tiles_n, accumulate;
  // optional (round 5): rowdot[s][m] = sum over the 128 columns n of slot s of W[m][n] * (this launch's fp32 product)[m][n], s < N / 128 --
  // the layer-scale gradient of a residual branch taken from the weight gradient of its last Linear (transformer_layer.py:70-88; see
  // op_gemm_tn_grouped)
  // (round 6) rscale != nullptr (only with rowdot): C[m][:] += rscale[m] * product[m][:], while rowdot sums W * the UNSCALED product -- with A the
  // un-gamma-scaled branch gradient, rscale = gamma and W the last Linear's weight, rowdot IS the layer-scale gradient (no division)
  const bf16_t* W; int64_t ldw; float* rowdot; const bf16_t* rscale;
};
// A queue is a list of RUNS: `n` consecutive slots of one problem's slot order (below), from slot0 on.
struct TnRun { int prob_n; int slot0; };  // prob_n = problem << 24 | n
struct TnGroupArgs {
  int nprob; int solo_from; unsigned long long* ctr;  // solo_from: see the kernel
  int qlen[8];                                        // slots in queue x
  short run_begin[10];                                // runs of queue x: [run_begin[x], run_begin[x + 1])
  TnRun runs[TN_MAX_RUNS];
  TnProb pr[TN_MAX_PROB];
};

// Slot order of a problem: its tiles group by group -- a group = all tiles along the SHORT dimension of the tile grid (chunks of <= 8
// when that is longer; a partial last chunk has empty slots), groups in order along the long dimension.  Consecutive slots therefore
// walk a [rows x short-dimension] rectangle of the tile grid row by row: W consecutive slots touch ceil(W / gs) + 1 panels of the
// wide operand and all gs panels of the narrow one.
struct TnGeom { int along_n, gs, nch, csz, ng; };
__host__ __device__ __forceinline__ TnGeom tn_geom(int tiles_m, int tiles_n) {
  TnGeom G;
  G.along_n = tiles_n <= tiles_m;
  G.gs = G.along_n ? tiles_n : tiles_m;
  G.nch = (G.gs + 7) >> 3;
  G.csz = (G.gs + G.nch - 1) / G.nch;
  G.ng = (G.along_n ? tiles_m : tiles_n) * G.nch;
  return G;
}
__host__ __device__ __forceinline__ int tn_queue_len(const TnGroupArgs& p, int x) { return p.qlen[x]; }
// slot q of queue x -> problem and tile; false for an empty slot
__host__ __device__ __forceinline__ bool tn_decode(const TnGroupArgs& p, int x, int q, int& prob, int& tm, int& tn) {
  for (int r = p.run_begin[x]; r < p.run_begin[x + 1]; ++r) {
    const int n = p.runs[r].prob_n & 0xffffff;
    if (q < n) {
      prob = p.runs[r].prob_n >> 24;
      const TnGeom G = tn_geom(p.pr[prob].tiles_m, p.pr[prob].tiles_n);
      const int sl = p.runs[r].slot0 + q;
      const int gi = sl / G.csz, s = sl - gi * G.csz;
      const int li = gi / G.nch, c = gi - li * G.nch;
      const int si = c * G.csz + s;
      tm = G.along_n ? li : si;
      tn = G.along_n ? si : li;
      return si < G.gs;
    }
    q -= n;
  }
  prob = 0; tm = tn = 0;
  return false;
}

// The epilogue of the grouped weight-gradient kernel, through LDS (the operand stages are free once every wave has left the main
// loop): the wave parks each [128 rows x 64 columns] fp32 half of its block in its own 32 KiB (256-byte rows, 16-byte chunks
// XOR-swizzled by the row: the fragment-layout writes -- 8 lanes = 8 rows of one chunk column -- and the row-layout reads are both
// conflict free) and reads it back with a lane per 8 consecutive columns: the read-modify-write of the bf16 gradient is then 16
// bytes per lane, 8 lanes = one 128-byte line, 16 loads + 16 stores per half.  (Round 4's first epilogue wrote straight from the
// MFMA fragment layout: 32 + 32 eight-byte accesses per half with adjacent lanes on different ROWS, which the texture addresser
// took one lane at a time -- 20 us of a tile.)  The old values travel in two batches of eight while the block is parked / while the
// first batch is folded; `between` (the next ticket's draw) runs after the first half so that its returning atomic is not in front
// of loads the wave waits for.  GUARD: tiles on the matrix edge (M, N are multiples of 8: a lane's 8 columns are in or out together).
// rowdot != nullptr (full tiles, accumulate): the fp32 block is also multiplied with the same block of W and summed along the rows -- a lane folds its 8
// columns of 16 rows over both 64-column halves, the 8 lanes of a row are folded by three shuffles, one store per row and wave into the
// wave's own slot of the [N / 128][M] partial matrix.
// (ROWDOT is a run-time, wave-uniform switch of the <ACCUM, !GUARD> instantiation: a fifth inlined copy of the epilogue made the register
// allocator hoist accumulator reads over the branches and spill 243 dwords per lane.)
template <bool ACCUM, bool GUARD, typename F>
__device__ __forceinline__ void tn_epilogue_lds(bf16_t* C, int64_t ldc, f32x4 (&acc)[2][4][8], int mrow0, int nbase, int M, int N, int lane, char* wlds,
                                                F between, const bf16_t* Wm = nullptr, int64_t ldw = 0, float* rowdot = nullptr,
                                                const bf16_t* rscale = nullptr) {
  const bool ROWDOT = ACCUM && !GUARD && rowdot != nullptr;
  const bool RSCALE = ROWDOT && rscale != nullptr;
  const int g = lane >> 4, t = lane & 15;
  const int rrow = lane >> 3, cp = lane & 7;
  float rd[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) rd[i] = 0.f;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    bf16_t* cbase = C + (int64_t)(mrow0 + rrow) * ldc + nbase + blk * 64 + cp * 8;
    const bf16_t* wbase = ROWDOT ? Wm + (int64_t)(mrow0 + rrow) * ldw + nbase + blk * 64 + cp * 8 : nullptr;
    const bool col_ok = !GUARD || nbase + blk * 64 + cp * 8 < N;
    auto ok = [&](int i) { return !GUARD || (col_ok && mrow0 + rrow + i * 8 < M); };
    bf16x8 old[8];
    if (ACCUM) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (ok(i)) old[i] = *reinterpret_cast<const bf16x8*>(cbase + (int64_t)i * 8 * ldc);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int row = mi * 16 + t;
        *reinterpret_cast<f32x4*>(wlds + row * 256 + (((f * 4 + g) ^ t) << 4)) = acc[blk][f][mi];
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      bf16x8 nxt[8], wrow[4];
      float gam[4] = {1.f, 1.f, 1.f, 1.f};  // row scales of the four rows in flight (1: fma(1, p, old) rounds like old + p)
      if (ACCUM && hb == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (ok(8 + i)) nxt[i] = *reinterpret_cast<const bf16x8*>(cbase + (int64_t)(8 + i) * 8 * ldc);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (ROWDOT && (i & 3) == 0) {  // W rows four at a time (the kernel sits at 254 of 256 VGPRs: eight in flight spilled)
#pragma unroll
          for (int j = 0; j < 4; ++j) wrow[j] = *reinterpret_cast<const bf16x8*>(wbase + (int64_t)(hb * 8 + i + j) * 8 * ldw);
          if (RSCALE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) gam[j] = (float)rscale[mrow0 + rrow + (hb * 8 + i + j) * 8];
          }
        }
        const int row = (hb * 8 + i) * 8 + rrow;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(wlds + row * 256 + (((2 * cp) ^ (row & 15)) << 4));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(wlds + row * 256 + (((2 * cp + 1) ^ (row & 15)) << 4));
        bf16x8 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          w[r] = (bf16_t)(ACCUM ? __builtin_fmaf(gam[i & 3], lo[r], (float)old[i][r]) : lo[r]);
          w[4 + r] = (bf16_t)(ACCUM ? __builtin_fmaf(gam[i & 3], hi[r], (float)old[i][4 + r]) : hi[r]);
        }
        if (ok(hb * 8 + i)) *reinterpret_cast<bf16x8*>(cbase + (int64_t)(hb * 8 + i) * 8 * ldc) = w;
        if (ROWDOT) {
          float sdot = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) sdot += lo[r] * (float)wrow[i & 3][r] + hi[r] * (float)wrow[i & 3][4 + r];
          rd[hb * 8 + i] += sdot;
        }
      }
      if (ACCUM && hb == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) old[i] = nxt[i];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (blk == 0) {
      between();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (ROWDOT) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = rd[i];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      // (round 6) slot nbase / 128 of the [2 * tiles_n][M] partial matrix: every (slot, row) is written exactly once per launch -- no
      // atomics, no zeroing, the same bits every run; op_gamma_grad_finish folds the slots in a fixed order
      if (cp == 0) rowdot[(int64_t)(nbase >> 7) * M + mrow0 + rrow + i * 8] = v;
    }
  }
}

__global__ __launch_bounds__(256) void gemm256w_tn_grouped_kernel(const TnGroupArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int sh_next;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int g = lane >> 4, t = lane & 15;

  // ---- queue state of the fetching thread (thread 0): home queue first, then the others in turn ----
  const int home = blockIdx.x & 7;
  const bool solo = (int)(blockIdx.x >> 3) >= p.solo_from;  // draws from the back of the queues
  int tried = 0;                  // queues found empty so far
  int xn = home;                  // queue of the ticket in flight
  unsigned long long qn = 0;      // the ticket: both claim counts of the queue before this claim
  auto draw = [&]() {             // one returning atomic on the current queue's counter
    xn = (home + tried) & 7;
    qn = __hip_atomic_fetch_add(p.ctr + xn * TN_CTR_STRIDE, solo ? (1ull << 32) : 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto resolve = [&]() {  // ticket -> code (queue << 24 | slot), stealing while queues turn out empty; -1: nothing left anywhere
    for (;;) {
      const int len = tn_queue_len(p, xn);
      const int front = (int)(unsigned)qn, back = (int)(qn >> 32);
      if (front + back < len) return (xn << 24) | (solo ? len - 1 - back : front);
      if (++tried >= 8) return -1;
      draw();
    }
  };
  if (tid == 0) {
    draw();
    sh_next = resolve();
  }
  __syncthreads();
  int code = __builtin_amdgcn_readfirstlane(sh_next);

  f32x4 acc[2][4][8];  // [64-column block][ni][mi], pinned in AGPRs by the inline-asm MFMAs

#ifdef OP_GEMM_TIMELINE
  int tl_slot = 0;  // per workgroup 32 records of four words: ticket code, tile start, main loop end, tile end (s_memrealtime)
  unsigned long long tl_cyc = 0;  // s_memtime at the tile's start: the shader clock over the main loop goes into bits 36+ of word 0
#define TLG_CYC0() tl_cyc = __builtin_readcyclecounter()
#define TLG_CYC1()                                                                                                      \
  do {                                                                                                                  \
    if (g_timeline && tid == 0 && tl_slot < 32)                                                                         \
      g_timeline[(int64_t)blockIdx.x * 128 + tl_slot * 4] |= (unsigned long long)(__builtin_readcyclecounter() - tl_cyc) << 36; \
  } while (0)
#define TLG(k, v)                                                                                                       \
  do {                                                                                                                  \
    if (g_timeline && tid == 0 && tl_slot < 32) g_timeline[(int64_t)blockIdx.x * 128 + tl_slot * 4 + (k)] = (v);          \
  } while (0)
#else
#define TLG(k, v)
#define TLG_CYC0()
#define TLG_CYC1()
#endif
  while (code >= 0) {
    int prob, pid_m, pid_n;
    const bool valid = tn_decode(p, code >> 24, code & 0xffffff, prob, pid_m, pid_n);
    TLG(0, (unsigned long long)(unsigned)code | ((unsigned long long)(valid ? prob + 1 : 0) << 32));
    TLG(1, __builtin_amdgcn_s_memrealtime());
    TLG_CYC0();
    if (valid) {
      // ---- transpose-read provider addresses (see gemm256w_tn_kernel).  Tile-independent, but derived PER TILE from an opaque copy of the
      // lane number: 16 registers that are only live in the main loop -- kept across the epilogue (which since round 5 may carry the
      // row-dot side product) they pushed seven kernel-lifetime values into scratch ----
      int lane_t = lane;
      asm volatile("" : "+v"(lane_t));
      const int g_ = lane_t >> 4, t_ = lane_t & 15;
      const int krow = g_ * 8 + (t_ >> 2);
      const int rowoff = krow * 512;
      int rdX[8], rdW[8];
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const int chunk = wm * 16 + mi * 2 + ((t_ & 3) >> 1);
        rdX[mi] = rowoff + ((chunk ^ swzA_tn(krow)) << 4) + (t_ & 1) * 8;
      }
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const int chunk = wn * 16 + f * 2 + ((t_ & 3) >> 1);
        rdW[f] = OPER2_BYTES + rowoff + ((chunk ^ swzA_tn(krow)) << 4) + (t_ & 1) * 8;
      }
      const TnProb& q = p.pr[prob];
      const int M = q.M, N = q.N;
      const int64_t lda = q.lda, ldb = q.ldb;
      const int m0 = pid_m * BM2, n0 = pid_n * 256;
      const int nk = q.K / BK2;
      const char* baseA = (const char*)q.A;
      const char* baseB = (const char*)q.B;
      unsigned offA[4], offB[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = i * 256 + tid;
        const int kr = s >> 5, pc = s & 31;
        const int ca = min(m0 + (pc ^ swzA_tn(kr)) * 8, M - 8);
        const int cb = min(n0 + (pc ^ swzA_tn(kr)) * 8, N - 8);
        offA[i] = (unsigned)(((int64_t)kr * lda + ca) * 2);
        offB[i] = (unsigned)(((int64_t)kr * ldb + cb) * 2);
      }
      const int64_t stepA = (int64_t)BK2 * lda * 2, stepB = (int64_t)BK2 * ldb * 2;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

      auto dma = [&](char* la, int j) {  // op j = 0..7 of a stage: A ops 0..3, B ops 4..7
        const int i = j & 3;
        const int wbase = (i * 256 + wid * 64) * 16;
        if (j < 4)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseA + offA[i]),
                                           (__attribute__((address_space(3))) void*)(la + wbase), 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseB + offB[i]),
                                           (__attribute__((address_space(3))) void*)(la + OPER2_BYTES + wbase), 16, 0, 0);
      };
      auto issue = [&](int slot) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dma(smem + slot * STAGE2_BYTES, j);
        baseA += stepA;
        baseB += stepB;
      };
      auto wait_landed = [&](int younger) {  // 8 LDS-DMA ops per stage and wave
        if (younger >= 3) WAIT_VM(24);
        else if (younger == 2) WAIT_VM(16);
        else if (younger == 1) WAIT_VM(8);
        else WAIT_VM(0);
      };
      struct Frags { s16x4 w[8][2]; s16x4 x[8][2]; };
      auto rd = [&](const char* st, Frags& f, int r) {
        if (r < 16) f.w[r >> 1][r & 1] = tr_read16(lds_addr(st) + rdW[r >> 1], (r & 1) != 0);
        else f.x[(r - 16) >> 1][r & 1] = tr_read16(lds_addr(st) + rdX[(r - 16) >> 1], (r & 1) != 0);
      };
      auto mfma1 = [&](const Frags& f, int j) {
        const int mi = j >> 3, ff = j & 7;
        const bf16x8 wv = join16(f.w[ff][0], f.w[ff][1]), xv = join16(f.x[mi][0], f.x[mi][1]);
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[ff >> 2][ff & 3][mi]) : "v"(wv), "v"(xv));
      };
      auto step_steady = [&](int kt, const Frags& cur, Frags& nxt) {
        WAIT_LGKM(0);
        WAIT_VM(16);
        __builtin_amdgcn_s_barrier();
        const char* st = smem + ((kt + 1) & (STAGES2 - 1)) * STAGE2_BYTES;
        char* la = smem + (kt & (STAGES2 - 1)) * STAGE2_BYTES;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
          mfma1(cur, j);
          if (j < 32) rd(st, nxt, j);
          else if (j < 40) dma(la, j - 32);
          if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        baseA += stepA;
        baseB += stepB;
      };
      auto step_tail = [&](int kt, const Frags& cur, Frags& nxt) {
        const bool has_next = kt + 1 < nk;
        WAIT_LGKM(0);
        if (has_next) wait_landed(min(nk - 2 - kt, STAGES2 - 2));
        __builtin_amdgcn_s_barrier();
        if (has_next) {
          const char* st = smem + ((kt + 1) & (STAGES2 - 1)) * STAGE2_BYTES;
#pragma unroll
          for (int r = 0; r < 32; ++r) rd(st, nxt, r);
        }
#pragma unroll
        for (int j = 0; j < 64; ++j) mfma1(cur, j);
      };

#pragma unroll
      for (int s0 = 0; s0 < STAGES2; ++s0)
        if (s0 < nk) issue(s0);
      Frags fA, fB;
      wait_landed(min(nk - 1, STAGES2 - 1));
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int r = 0; r < 32; ++r) rd(smem, fA, r);
      int kt = 0;
      for (; kt + STAGES2 + 1 < nk; kt += 2) {
        step_steady(kt, fA, fB);
        step_steady(kt + 1, fB, fA);
      }
      for (; kt < nk; kt += 2) {
        step_tail(kt, fA, fB);
        step_tail(kt + 1, fB, fA);
      }
      // The inline-asm MFMAs are invisible to the hazard recogniser: 24 wait states before anything reads an accumulator.  The
      // accumulators are OPERANDS of the padding statements, or the scheduler hoists v_accvgpr_read above them (it did: caught by
      // tools/check_mfma_hazards.py); volatile asm statements keep their order, so the later groups sit behind the s_nops too.
#define TN_ACC8(b, n) "+a"(acc[b][n][0]), "+a"(acc[b][n][1]), "+a"(acc[b][n][2]), "+a"(acc[b][n][3]), "+a"(acc[b][n][4]), "+a"(acc[b][n][5]), "+a"(acc[b][n][6]), "+a"(acc[b][n][7])
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : TN_ACC8(0, 0), TN_ACC8(0, 1), TN_ACC8(0, 2));
      asm volatile("" : TN_ACC8(0, 3), TN_ACC8(1, 0), TN_ACC8(1, 1));
      asm volatile("" : TN_ACC8(1, 2), TN_ACC8(1, 3));
#undef TN_ACC8
      TLG(2, __builtin_amdgcn_s_memrealtime());
      TLG_CYC1();
      // the NEXT tile's ticket: drawn inside the epilogue, in flight during the rest of it.  (Drawn at the start of the tile -- round 4's first
      // version -- the last tickets of a launch sat for up to a whole tile with workgroups that were busy, while workgroups that
      // became free found the queues empty and left: finish times spread over 700 us of a 4.5 ms launch.)
      auto draw_next = [&]() { if (tid == 0) draw(); };
      __syncthreads();  // every wave is out of the main loop: the operand stages are free for the epilogue
      {
        char* wlds = smem + wid * 32768;
        const int mr = m0 + wm * 128, nb = n0 + wn * 128;
        if (m0 + 256 <= M && n0 + 256 <= N) {  // (uniform) tile inside the matrix
          if (q.accumulate) tn_epilogue_lds<true, false>(q.C, q.ldc, acc, mr, nb, M, N, lane, wlds, draw_next, q.W, q.ldw, q.rowdot, q.rscale);  // (rowdot: host checked accumulate, M, N % 256 == 0)
          else tn_epilogue_lds<false, false>(q.C, q.ldc, acc, mr, nb, M, N, lane, wlds, draw_next);
        } else {
          if (q.accumulate) tn_epilogue_lds<true, true>(q.C, q.ldc, acc, mr, nb, M, N, lane, wlds, draw_next);
          else tn_epilogue_lds<false, true>(q.C, q.ldc, acc, mr, nb, M, N, lane, wlds, draw_next);
        }
      }
    }
    TLG(3, __builtin_amdgcn_s_memrealtime());
#ifdef OP_GEMM_TIMELINE
    ++tl_slot;
#endif
    if (tid == 0) {
      if (!valid) draw();  // (an empty slot has no main loop behind which the draw was issued)
      sh_next = resolve();
    }
    __syncthreads();  // the ticket is published, and every wave is done with the LDS stages the next tile overwrites
    code = __builtin_amdgcn_readfirstlane(sh_next);
    __syncthreads();  // ... and has read it before thread 0 publishes the one after
  }
  if (tid == 0) {  // the last workgroup to leave re-arms the counters for the next launch on this stream
    const unsigned long long gone = __hip_atomic_fetch_add(p.ctr + 8 * TN_CTR_STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gone == gridDim.x - 1) {
#pragma unroll 1
      for (int x = 0; x <= 8; ++x) __hip_atomic_store(p.ctr + x * TN_CTR_STRIDE, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Per-call tuning word of the GEMM entry points (last argument before the stream; 0 = the defaults production uses).  The
// library keeps NO tuning state: tests and tools that want a specific kernel flavour pass it with the call.
//   bits 0-1   tile: 0 auto, 1 force 128x128, 2 force 256x256
//   bits 2-3   flavour of the 256x256 NT kernel: 0 auto (launches that fill every CU: BK = 64 full-line staging, four waves of
//              128 x 128 when K >= 3072, eight waves of 128 x 64 otherwise), 1 BK = 32 four-stage, 2 eight-wave full-line, 3 four-wave
//   bits 4-6   tail-rows split: 0 default (when it saves a round and K >= 1024), 1 off, 3 whenever it saves a round, 4 always
//   bits 7-11  M-tiles per L2 group of the 256x256 kernels (0 = auto)
//   bits 12-14 unused (rounds 1-4: timing ablations of the BK = 32 kernel)
//   bits 15-18 forced K-split count of small problems (tools only)
//   bit  19    register-staged operand path instead of LDS-DMA (128x128 kernel; tests)
//   bits 20-22 four-wave NT launches: 0 auto (persistent gemm256p_kernel for K <= 2048, else gemm256v_kernel), 3 gemm256v_kernel,
//              6 gemm256p_kernel; op_gemm_nt_grouped: 7 = one tile per workgroup instead of the persistent walk (tests, A/B)
struct GemmTune { int tile_mode, fullline, tail_rows, gm, force_splits, glds, sched; };
static GemmTune decode_tune(int64_t t) {
  GemmTune T;
  T.tile_mode = (int)(t & 3);
  const int fl = (int)((t >> 2) & 3);
  T.fullline = fl == 0 ? 2 : fl == 3 ? 3 : fl - 1;  // internal: 0 BK = 32, 1 eight-wave full-line, 2 auto, 3 four-wave
  const int tr = (int)((t >> 4) & 7);
  T.tail_rows = tr == 0 ? 1 : tr - 1;         // internal: 0 off, 1 default, 2 whenever it saves a round, 3 always
  T.gm = (int)((t >> 7) & 31);
  T.force_splits = (int)((t >> 15) & 15);
  T.glds = ((t >> 19) & 1) ? 0 : 1;
  T.sched = (int)((t >> 20) & 7);
  return T;
}

constexpr int V_SCHED_DEFAULT = 3;  // four-wave NT launches with K > 2048 when the tune word does not name a kernel: gemm256v_kernel

template <int EPI, int SCHED>
int launch256v(const GemmArgs& a, hipStream_t s, dim3 grid, size_t sh) {
  OP_ENSURE_LDS((gemm256v_kernel<EPI, SCHED>), (int)sh, "gemm256v");
  hipLaunchKernelGGL((gemm256v_kernel<EPI, SCHED>), grid, dim3(256), sh, s, a);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

// persistent grouped kernel: one workgroup per CU (or per tile when there are fewer tiles)
static int num_cus() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n - n % 8;  // a multiple of the 8 XCDs: workgroup w stays on XCD w % 8 over its whole tile list
    if (cached[dev] <= 0) cached[dev] = 8;
  }
  return cached[dev];
}

template <int EPI>
int launch256p(const GroupArgs& ga, hipStream_t s, bool persistent = true) {
  const size_t sh = (size_t)SLOTS3 * SLOT3_BYTES;
  OP_ENSURE_LDS((gemm256p_kernel<EPI>), sh, "gemm256p");
  const int ntiles = ga.tiles_m * ga.tiles_n;
  hipLaunchKernelGGL((gemm256p_kernel<EPI>), dim3((!persistent || ntiles < num_cus()) ? ntiles : num_cus()), dim3(256), sh, s, ga);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

static int launch256p_any(const GroupArgs& ga, int epi, hipStream_t s, bool persistent) {
  if (epi == EPI_BIAS) return launch256p<EPI_BIAS>(ga, s, persistent);
  if (epi == EPI_RESID) return launch256p<EPI_RESID>(ga, s, persistent);
  if (epi == EPI_RESID_ROWS) return launch256p<EPI_RESID_ROWS>(ga, s, persistent);
  return OP_ENOTSUP;
}

// single-problem GroupArgs of a plain launch
static GroupArgs group_of(const GemmArgs& a, int epi) {
  GroupArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.nprob = 1;
  ga.A[0] = a.A; ga.M[0] = a.M;
  ga.tiles_m = ceil_div(a.M, 256);
  ga.tiles_n = ceil_div(a.N, epi == EPI_GEGLU ? 128 : 256);
  ga.mt_end[0] = ga.mt_end[1] = ga.mt_end[2] = ga.tiles_m;
  for (int i = 0; i < 3; ++i) { ga.B[0][i] = a.B[i]; ga.bias[0][i] = a.bias[i]; }
  ga.C[0] = a.C; ga.H0[0] = a.H0; ga.H1[0] = a.H1; ga.resid[0] = a.resid; ga.gamma[0] = a.gamma; ga.rowscale[0] = a.rowscale;
  ga.rows_per_sample[0] = a.rows_per_sample; ga.rows[0] = a.rows;
  ga.lda = a.lda; ga.ldb = a.ldb; ga.ldc = a.ldc; ga.ldr = a.ldr; ga.alpha = a.alpha;
  ga.n_seg = a.n_seg; ga.N = a.N; ga.K = a.K; ga.gm = a.gm;
  return ga;
}

template <int EPI>
int launch256(const GemmArgs& a, hipStream_t s, const GemmTune& T, int splits = 1) {
  const dim3 grid(a.tiles_m * a.tiles_n, splits);
  const size_t sh = STAGES2 * STAGE2_BYTES;
  if constexpr (EPI != EPI_RESID_ROWS) OP_ENSURE_LDS((gemm256_kernel<EPI>), (int)sh, "gemm256");
  const bool fills = (int64_t)a.tiles_m * a.tiles_n >= 256 && splits == 1;
  // four-wave flavour (one wave per SIMD, 128 x 128 per wave): +9 ... +14 % at K = 6144, +1 ... +5 % at K = 1536 (bias and
  // residual epilogues); the GeGLU launch (VALU-heavy epilogue on half as many waves) is 2 % slower and stays on eight waves;
  // bit-identical results either way
  if ((T.fullline == 3 || (T.fullline == 2 && fills && EPI != EPI_GEGLU)) && splits == 1 &&
      a.N % ((EPI == EPI_GEGLU) ? 128 : 256) == 0 && a.K % 64 == 0 && a.K >= 128) {
    const size_t sh5 = (size_t)SLOTS3 * SLOT3_BYTES;
    // Round 5: single problems with a SHORT K walk their tiles on the persistent kernel as well.  Round 3 measured it 1.5 ... 3 % behind
    // one tile per workgroup there -- with 16-28 scratch operations (vmcnt(0) waits next to the DMA stream) at every tile boundary; with
    // the tile walk at ScratchSize 0 it is ahead where tiles are short (K = 1536: q|k|v -1.7 %, FFN up-projection -1 ... -2 %, input
    // gradients of out-proj / down-projection -1.7 ... -3.6 %), level or behind at K >= 4608 (profiles/r5_blas_compare_sched6_ab.txt);
    // whole step 700.3 -> 695.2 ms with every four-wave launch persistent (profiles/r5_bench_sched6_samebox_*.json).
    const bool short_k = (EPI == EPI_BIAS || epi_is_resid(EPI)) && a.m_off == 0 && a.K <= 2048;
    const int sched = T.sched == 0 ? (short_k ? 6 : V_SCHED_DEFAULT) : T.sched;
    if constexpr (EPI == EPI_BIAS || epi_is_resid(EPI)) {
      if (sched == 6 && a.m_off == 0) return launch256p<EPI>(group_of(a, EPI), s, true);
    }
    return launch256v<EPI, 3>(a, s, grid, sh5);
  }
  if constexpr (EPI == EPI_RESID_ROWS) {  // (gemm_nt_impl sends these launches to the 128 x 128 kernel: see four_wave_256)
    op_set_error("gemm_nt: the row-table residual epilogue has no eight-wave 256 x 256 kernel");
    return OP_ENOTSUP;
  } else {
  // T.fullline: 0 BK = 32, 1 full-line always, 2 (default) full-line when the launch fills every CU at least once
  if ((T.fullline == 1 || (T.fullline == 2 && fills)) &&
      a.N % ((EPI == EPI_GEGLU) ? 128 : 256) == 0 && a.K % 64 == 0) {
    const size_t sh5 = (size_t)SLOTS3 * SLOT3_BYTES;
    OP_ENSURE_LDS((gemm256b_kernel<EPI>), (int)sh5, "gemm256b");
    hipLaunchKernelGGL((gemm256b_kernel<EPI>), grid, dim3(512), sh5, s, a);
  } else {
    hipLaunchKernelGGL((gemm256_kernel<EPI>), grid, dim3(512), sh, s, a);
  }
  OP_LAUNCH_CHECK();
  return OP_OK;
  }
}

// does launch256 take a four-wave kernel (gemm256v / gemm256p) for this launch?  (the row-table residual epilogue exists in those and in
// the 128 x 128 kernel only: the eight-wave 256 x 256 kernels have no registers left for it)
static bool four_wave_256(int64_t M, int64_t N, int64_t K, int splits, const GemmTune& T) {
  const bool fills = (int64_t)ceil_div(M, 256) * ceil_div(N, 256) >= 256 && splits == 1;
  return (T.fullline == 3 || (T.fullline == 2 && fills)) && splits == 1 && N % 256 == 0 && K % 64 == 0 && K >= 128;
}

template <int EPI>
int launch(const GemmArgs& a, int glds, hipStream_t s, int splits = 1, int batch = 1, NtBatch bt = NtBatch{0, 0, 0, 0}) {
  const dim3 grid(a.tiles_m * a.tiles_n, splits, batch);
  const size_t sh = 4 * TILE_BYTES;
  if (glds) OP_ENSURE_LDS((gemm_nt_kernel<EPI, true>), sh, "gemm");
  else OP_ENSURE_LDS((gemm_nt_kernel<EPI, false>), sh, "gemm");
  if (glds)
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, true>), grid, dim3(256), sh, s, a, bt);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, false>), grid, dim3(256), sh, s, a, bt);
  OP_LAUNCH_CHECK();
  return OP_OK;
}


// out[m][n] (bf16) = sum_z slab_z[m][n] (fp32); 8 elements per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int64_t slab,
                                                            bf16_t* __restrict__ out, int64_t ldc, int M, int N,
                                                            int accumulate) {
  const int n8 = N / 8;
  const int64_t total = (int64_t)M * n8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / n8;
    const int c = (int)(i - m * n8) * 8;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (accumulate) Vec8<bf16_t>::load(out + m * ldc + c, a);
    // up to eight slabs requested before the first is added (a load -> add chain per slab left the fold at 4.2 TB/s, round 3);
    // the slabs are added in the same order as before
    for (int z0 = 0; z0 < splits; z0 += 8) {
      typename Vec8<float>::raw_t raw[8];
#pragma unroll
      for (int z = 0; z < 8; ++z)
        if (z0 + z < splits) raw[z] = Vec8<float>::ldraw(ws + (z0 + z) * slab + m * N + c);
#pragma unroll
      for (int z = 0; z < 8; ++z)
        if (z0 + z < splits) {
          float v[8];
          Vec8<float>::cvt(raw[z], v);
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] += v[j];
        }
    }
    Vec8<bf16_t>::store(out + m * ldc + c, a);
  }
}

// Split-K fold WITH the epilogue (bias / residual + layer-scale + drop-path, optional branch output): lets the small
// latency-bound launches that carry an epilogue (the <= 128 leftover rows of a tail-rows split) split K as well.
struct FoldArgs {
  const float* ws; int splits; int64_t slab;
  bf16_t* out; int64_t ldc; int M, N;
  const bf16_t* bias[3]; int n_seg;
  int resid_epi;  // 0: out = acc + bias; 1: out = resid + rowscale * gamma * (acc + bias), h0 (optional) = acc + bias
  const bf16_t* resid; int64_t ldr; const bf16_t* gamma; const float* rowscale; int rows_per_sample, m_off;
  bf16_t* h0;
  const int* rows;  // nullable (resid_epi only): resid / out are rows rows[m] of larger matrices, < 0 = no place (see EPI_RESID_ROWS)
};
__global__ __launch_bounds__(256) void splitk_fold_epilogue_kernel(const FoldArgs p) {
  const int n8 = p.N / 8;
  const int64_t total = (int64_t)p.M * n8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / n8;
    const int c = (int)(i - m * n8) * 8;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int z = 0; z < p.splits; ++z) {
      float v[8];
      Vec8<float>::load(p.ws + z * p.slab + m * p.N + c, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
    const int seg = c / p.n_seg;
    const bf16_t* bp = p.bias[seg];
    if (bp) {
      float b[8];
      Vec8<bf16_t>::load(bp + (c - seg * p.n_seg), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += b[j];
    }
    int64_t mo = m;
    if (p.resid_epi) {
      if (p.h0) Vec8<bf16_t>::store(p.h0 + m * p.ldc + c, a);
      if (p.rows) {
        mo = p.rows[m];
        if (mo < 0) continue;
      }
      float r[8], gv[8];
      Vec8<bf16_t>::load(p.resid + mo * p.ldr + c, r);
      const float rs = p.rowscale ? p.rowscale[(m + p.m_off) / p.rows_per_sample] : 1.f;
      if (p.gamma) {
        Vec8<bf16_t>::load(p.gamma + c, gv);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = resid_out(r[j], rs, gv[j], a[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = r[j] + rs * a[j];
      }
    }
    Vec8<bf16_t>::store(p.out + mo * p.ldc + c, a);
  }
}

// Launch plan: tile size and split-K factor from a wave-quantisation model.  One "round" fills every workgroup slot
// once (128x128: 2 per CU = 512, 256x256: 1 per CU = 256); relative slot-round costs are calibrated on MI355X
// micro-benchmarks (128x128 ~ 900 TF/s, 256x256 ~ 1040 TF/s sustained).
struct GemmPlan { int tile; int splits; int kt_per_split; };

GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K, int epilogue, bool allow_256, bool allow_split, int64_t ws_bytes, const GemmTune& T) {
  const double c128 = 2.0 * 128 * 128 / 920.0, c256 = 256.0 * 256 / 1080.0;  // time of one slot-round per unit K
  const int64_t t128 = (int64_t)ceil_div(M, 128) * ceil_div(N, epilogue == EPI_GEGLU ? 64 : 128);
  const int64_t t256 = (int64_t)ceil_div(M, 256) * ceil_div(N, epilogue == EPI_GEGLU ? 128 : 256);
  GemmPlan best = {128, 1, 0};
  double best_t = 1e300;
  if (t128 <= 256 && epilogue != EPI_GEGLU) {
    // Small problems (batch-1 feature extraction, M = 257; the leftover rows of a tail-rows split): fewer 128x128 tiles than
    // CUs.  Measured under hipGraph replay with cold weights (tools/gemm_small_m.py, profiles/r1_gemm_small_m.txt): a launch
    // costs 4.3 us + 0.62 us per 64-deep K-step of the slowest workgroup while every workgroup has a CU to itself, ~1.2 us
    // per step with two per CU; the 256x256 kernel is 8 us + 0.5 us per 32-deep step and never wins here.  Splitting K
    // spreads the K-steps over the idle CUs at the price of the fold launch (4.5 us) and the fp32 slab round trip.
    const int nk = (int)(K / BK);
    for (int s = 1; s <= 8; ++s) {
      if (s > 1 && (!allow_split || (int64_t)s * M * N * 4 > ws_bytes || nk / s < 2)) break;
      const int kps = ceil_div(nk, s), eff_s = ceil_div(nk, kps);
      const int64_t blocks = t128 * eff_s;
      const double per_cu = blocks <= 256 ? 1.0 : 0.94 * (double)((blocks + 255) / 256);
      double t = 4.3 + 0.62 * per_cu * kps;
      if (eff_s > 1) t += 4.5 + (double)eff_s * M * N * 8.0 / 4.0e6;  // slab bytes at 4 TB/s, in us
      if (T.force_splits > 0 ? s == T.force_splits : t < best_t) { best_t = t; best = {128, eff_s, eff_s > 1 ? kps : 0}; }
    }
    return best;
  }
  for (int tile = 128; tile <= 256; tile += 128) {
    if (tile == 256 && !allow_256) continue;
    const int bk = tile == 128 ? BK : BK2;
    const int nk = (int)(K / bk);
    for (int s = 1; s <= 8; ++s) {
      if (s > 1 && (!allow_split || (int64_t)s * M * N * 4 > ws_bytes || nk / s < 8)) break;
      int kps = ceil_div(nk, s);
      kps += kps & 1;  // the 256x256 kernel consumes K-stages in pairs
      const int eff_s = ceil_div(nk, kps);
      const int64_t blocks = (tile == 128 ? t128 : t256) * eff_s;
      const int64_t rounds = tile == 128 ? (blocks + 511) / 512 : (blocks + 255) / 256;
      double t = (double)rounds * (tile == 128 ? c128 : c256) * kps * bk;
      if (eff_s > 1) t += (double)eff_s * M * N * 8.0 / 4.0e3 + 1.0e4;  // slab write + read at ~4 TB/s + one extra launch, in the same units (0.512 ns)
      if (t < best_t) { best_t = t; best = {tile, eff_s, eff_s > 1 ? kps : 0}; }
    }
  }
  return best;
}


// The whole launch decision of op_gemm_nt in one place (also served to the host by op_gemm_plan, so that it can be tested
// without a GPU): tile, K-splits, whether the epilogue moves to the fold kernel, and whether the <= 128 leftover rows of
// M % 256 become a second, small launch.
struct NtDecision { GemmPlan plan; bool fold_epi; bool tail_split; int64_t m_main; };

NtDecision decide_nt(int64_t M, int64_t N, int64_t K, int epilogue, bool has_bias0, bool seg_ok, bool off32_ok, bool fold_layout_ok,
                     bool have_ws, int64_t ws_bytes, bool allow_tail_split, const GemmTune& T) {
  NtDecision d;
  const bool allow_256 = T.tile_mode != 1 && T.glds && seg_ok && off32_ok;
  // split-K: bias-free plain launches (weight gradients, dgrads), and -- for launches of a few M-tiles (the leftover rows
  // of a tail-rows split, batch-1 feature extraction: M = 257), which are latency-bound on K with most CUs idle -- also
  // bias / residual epilogues, applied by the fold kernel.  plan_gemm's cost model decides whether a split pays.
  d.fold_epi = (epilogue == EPI_RESID || (epilogue == EPI_BIAS && has_bias0)) && M <= 1024 && fold_layout_ok;
  const bool allow_split = ((epilogue == EPI_BIAS && !has_bias0) || d.fold_epi) && have_ws && N % 8 == 0;
  // Tail rows first.  When M is not a multiple of 256, the N-tiles of the partial last M-tile can cost a whole extra round of
  // every CU (M = 128 x 257: 774 tiles = 3.02 rounds for N = 1536).  If dropping them saves a round, the full M-tiles run as
  // one launch and the <= 128 leftover rows as a second, small one (128 x 128 tiles).  This is decided on the UNSPLIT plan: a
  // K-split of the whole problem would also hide the fourth round (7 half-length rounds instead of 4) but pays fp32 slabs of the
  // whole output for it (the K = 12 288 FFN input gradient at 32 896 tokens: 1.34 ms split against 0.87 ms tail-split, round 3).
  d.tail_split = false;
  d.m_main = M;
  const GemmPlan unsplit = plan_gemm(M, N, K, epilogue, allow_256, false, 0, T);
  if (allow_tail_split && T.tail_rows && (unsplit.tile == 256 || (T.tile_mode == 2 && allow_256)) && M > 256) {
    const int64_t m_rem = M % 256;
    const int64_t tn = ceil_div(N, epilogue == EPI_GEGLU ? 128 : 256);
    const int64_t r_full = ceil_div(ceil_div(M, 256) * tn, 256), r_main = ceil_div(((M - m_rem) / 256) * tn, 256);
    if (m_rem > 0 && m_rem <= 128 && (T.tail_rows == 3 || (r_main < r_full && (T.tail_rows == 2 || K >= 1024)))) {
      d.tail_split = true;
      d.m_main = M - m_rem;
      d.plan = unsplit;
      if (T.tile_mode == 2 && allow_256 && d.plan.tile != 256) d.plan = {256, 1, 0};
      return d;
    }
  }
  d.plan = plan_gemm(M, N, K, epilogue, allow_256, allow_split, ws_bytes, T);
  if (T.tile_mode == 2 && allow_256 && d.plan.tile != 256) d.plan = {256, 1, 0};
  return d;
}

}  // namespace

extern "C" int op_prof_begin(int family, double work, void* stream);
extern "C" void op_prof_end(int slot, void* stream);

extern "C" {

// Generic entry.  epilogue: 0 bias->bf16, 1 alpha*acc+bias -> f32, 2 GeGLU, 3 residual.
//   A [M,K] bf16 (lda), B0/B1/B2: weight segments [n_seg, K] each (ldb); for GeGLU B0 = wi_0, B1 = wi_1 [N, K].
//   bias0..2 nullable [n_seg] bf16.  C [M,N] (ldc) bf16 (f32 for epilogue 1).
//   GeGLU: C = gelu(h0)*h1, optional h0/h1 [M,N] bf16 (same ldc) for the backward pass.
//   residual: C = resid + rowscale[m / rows_per_sample] * gamma[n] * (acc + bias[n]); gamma, rowscale nullable;
//             resid may alias C (in-place accumulate); h0 (optional) receives y = acc + bias.
//   workspace (optional, fp32 scratch of workspace_bytes): enables split-K for bias-free epilogue-0 launches with few
//   output tiles and a long K (the weight-gradient GEMMs).
static int gemm_nt_impl(const void* A, int64_t lda, const void* B0, const void* B1, const void* B2, int64_t ldb, int64_t n_seg,
                        const void* bias0, const void* bias1, const void* bias2, void* C, int64_t ldc, void* h0, void* h1,
                        const void* resid, int64_t ldr, const void* gamma, const float* rowscale, int64_t rows_per_sample,
                        const float* alpha, int64_t M, int64_t N, int64_t K, int epilogue, void* workspace,
                        int64_t workspace_bytes, void* stream, int64_t m_off, bool allow_tail_split, const GemmTune& T,
                        const int* rows = nullptr, int64_t rows_total = 0) {
  OP_CHECK_ARG(A && B0 && C, "gemm_nt: null A/B/C");
  if (rows) {  // (ABI 9) residual epilogue through a row table: resid / C are the bases of matrices of rows_total rows
    OP_CHECK_ARG(epilogue == EPI_RESID, "gemm_nt: resid_rows goes with the residual epilogue");
    OP_CHECK_ARG(rows_total > 0 && N % 8 == 0 && ldc % 8 == 0 && ldr % 8 == 0 &&
                 (rows_total * (ldc > ldr ? ldc : ldr) + N) * 2 < (int64_t)0xfff00000ll,
                 "gemm_nt: resid_rows: the full matrices must lie below 4 GiB (rows_total=%lld)", (long long)rows_total);
  }
  OP_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm_nt: bad sizes M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  OP_CHECK_ARG(K % BK == 0, "gemm_nt: K=%lld must be a multiple of %d (pad on the host)", (long long)K, BK);  // => even number of 32-deep stages
  OP_CHECK_ARG(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "gemm_nt: N, lda, ldb must be multiples of 8");
  OP_CHECK_ARG(epilogue >= 0 && epilogue <= 3, "gemm_nt: bad epilogue %d", epilogue);
  if (M == 0) return OP_OK;
  GemmArgs a;
  a.A = (const bf16_t*)A; a.lda = lda;
  a.B[0] = (const bf16_t*)B0; a.B[1] = (const bf16_t*)B1; a.B[2] = (const bf16_t*)B2; a.ldb = ldb;
  a.bias[0] = (const bf16_t*)bias0; a.bias[1] = (const bf16_t*)bias1; a.bias[2] = (const bf16_t*)bias2;
  a.C = C; a.ldc = ldc; a.H0 = (bf16_t*)h0; a.H1 = (bf16_t*)h1;
  a.resid = (const bf16_t*)resid; a.ldr = ldr; a.gamma = (const bf16_t*)gamma; a.rowscale = rowscale;
  a.rows_per_sample = rows_per_sample > 0 ? (int)rows_per_sample : 1;
  a.alpha = alpha;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.gm = 4; a.m_off = (int)m_off; a.rows = rows;
  a.n_seg = (int)(n_seg > 0 ? n_seg : N);
  a.tiles_m = ceil_div(M, BM);
  if (epilogue == EPI_GEGLU) {
    OP_CHECK_ARG(B1, "gemm_nt: GeGLU needs two weights");
    OP_CHECK_ARG((h0 == nullptr) == (h1 == nullptr), "gemm_nt: GeGLU h0/h1 must both be given or both null");
    a.n_seg = (int)N;
    a.tiles_n = ceil_div(N, 64);
  } else {
    const int nsegs = ceil_div(N, a.n_seg);
    OP_CHECK_ARG(nsegs <= 3, "gemm_nt: at most 3 weight segments");
    OP_CHECK_ARG(nsegs == 1 || a.n_seg % 128 == 0, "gemm_nt: multi-segment launch needs n_seg %% 128 == 0");
    OP_CHECK_ARG(nsegs < 2 || B1, "gemm_nt: missing B1");
    OP_CHECK_ARG(nsegs < 3 || B2, "gemm_nt: missing B2");
    a.tiles_n = ceil_div(N, 128);
  }
  if (epilogue == EPI_RESID) OP_CHECK_ARG(resid, "gemm_nt: residual epilogue needs resid");
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * (double)M * (double)N * (double)K * (epilogue == EPI_GEGLU ? 2.0 : 1.0);
  const bool seg_ok = epilogue == EPI_GEGLU || a.n_seg >= (int)N || a.n_seg % 256 == 0;
  const bool off32_ok = (M * lda < ((int64_t)1 << 30)) && (N * ldb < ((int64_t)1 << 30));
  const NtDecision dec = decide_nt(M, N, K, epilogue, bias0 != nullptr, seg_ok, off32_ok, a.n_seg % 8 == 0 && ldc % 8 == 0,
                                   workspace != nullptr, workspace_bytes, allow_tail_split, T);
  const GemmPlan plan = dec.plan;
  const bool fold_epi = dec.fold_epi;
  if (dec.tail_split) {
    const int64_t m_main = dec.m_main, m_rem = M - dec.m_main;
    const int64_t esz = epilogue == EPI_F32 ? 4 : 2;
    int rc = gemm_nt_impl(A, lda, B0, B1, B2, ldb, n_seg, bias0, bias1, bias2, C, ldc, h0, h1, resid, ldr, gamma, rowscale,
                          rows_per_sample, alpha, m_main, N, K, epilogue, workspace, workspace_bytes, stream, m_off, false, T, rows,
                          rows_total);
    if (rc != OP_OK) return rc;
    // (with a row table C / resid stay the bases of the full matrices: the table moves on instead)
    return gemm_nt_impl((const bf16_t*)A + m_main * lda, lda, B0, B1, B2, ldb, n_seg, bias0, bias1, bias2,
                        rows ? C : (void*)((char*)C + m_main * ldc * esz), ldc, h0 ? (bf16_t*)h0 + m_main * ldc : nullptr,
                        h1 ? (bf16_t*)h1 + m_main * ldc : nullptr,
                        (resid && !rows) ? (const void*)((const bf16_t*)resid + m_main * ldr) : resid, ldr,
                        gamma, rowscale, rows_per_sample, alpha, m_rem, N, K, epilogue, workspace, workspace_bytes, stream,
                        m_off + m_main, false, T, rows ? rows + m_main : nullptr,
                        rows_total);  // (the small launch may split K: 12 tiles alone are latency-bound)
  }
  a.kt_per_split = plan.kt_per_split;
  a.slab = (int64_t)M * N;
  int epi = (epilogue == EPI_RESID && rows) ? (int)EPI_RESID_ROWS : epilogue;
  void* c_final = C;
  const int64_t ldc_final = ldc;
  FoldArgs fold;
  if (plan.splits > 1) {  // partial sums go to fp32 slabs, folded by splitk_reduce_kernel / splitk_fold_epilogue_kernel
    fold.ws = (const float*)workspace; fold.splits = plan.splits; fold.slab = a.slab;
    fold.out = (bf16_t*)C; fold.ldc = ldc; fold.M = (int)M; fold.N = (int)N;
    fold.bias[0] = a.bias[0]; fold.bias[1] = a.bias[1]; fold.bias[2] = a.bias[2]; fold.n_seg = a.n_seg;
    fold.resid_epi = epilogue == EPI_RESID; fold.resid = a.resid; fold.ldr = a.ldr; fold.gamma = a.gamma;
    fold.rowscale = a.rowscale; fold.rows_per_sample = a.rows_per_sample; fold.m_off = a.m_off; fold.h0 = a.H0; fold.rows = rows;
    a.C = workspace;
    a.ldc = N;
    a.alpha = nullptr;
    a.bias[0] = a.bias[1] = a.bias[2] = nullptr;  // applied once, by the fold
    a.H0 = a.H1 = nullptr;
    epi = EPI_F32;
  }
  const int slot = op_prof_begin(0, flops, stream);
  int rc;
  if (plan.tile == 256 && !(epi == EPI_RESID_ROWS && !four_wave_256(M, N, K, plan.splits, T))) {
    a.tiles_m = ceil_div(M, 256);
    a.tiles_n = ceil_div(N, epilogue == EPI_GEGLU ? 128 : 256);
    // L2 tile-group depth (tools/gemm_gm.py): few column tiles (N = 1536) -> walk all N-tiles of ONE M-tile together
    // (+3-5 %); wide outputs -> 8 M-tiles per group (+1-2 % over 4)
    a.gm = T.gm > 0 ? T.gm : (a.tiles_n <= 8 ? 1 : 8);
    switch (epi) {
      case EPI_BIAS: rc = launch256<EPI_BIAS>(a, s, T, plan.splits); break;
      case EPI_F32: rc = launch256<EPI_F32>(a, s, T, plan.splits); break;
      case EPI_GEGLU: rc = launch256<EPI_GEGLU>(a, s, T, plan.splits); break;
      case EPI_RESID_ROWS: rc = launch256<EPI_RESID_ROWS>(a, s, T, plan.splits); break;
      default: rc = launch256<EPI_RESID>(a, s, T, plan.splits); break;
    }
  } else {
    switch (epi) {
      case EPI_BIAS: rc = launch<EPI_BIAS>(a, T.glds, s, plan.splits); break;
      case EPI_F32: rc = launch<EPI_F32>(a, T.glds, s, plan.splits); break;
      case EPI_GEGLU: rc = launch<EPI_GEGLU>(a, T.glds, s, plan.splits); break;
      case EPI_RESID_ROWS: rc = launch<EPI_RESID_ROWS>(a, T.glds, s, plan.splits); break;
      default: rc = launch<EPI_RESID>(a, T.glds, s, plan.splits); break;
    }
  }
  if (rc == OP_OK && plan.splits > 1) {
    int64_t nb = ((int64_t)M * (N / 8) + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (fold_epi)
      hipLaunchKernelGGL(splitk_fold_epilogue_kernel, dim3((int)nb), dim3(256), 0, s, fold);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)nb), dim3(256), 0, s, (const float*)workspace, plan.splits, a.slab,
                         (bf16_t*)c_final, ldc_final, (int)M, (int)N, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { op_set_error("gemm_nt: split-K reduce launch failed: %s", hipGetErrorString(e)); rc = (int)e; }
  }
  op_prof_end(slot, stream);
  return rc;
}

// `batch` equally shaped products  C_z[M,N] = A_z[M,K] W_z[N,K]^T (+ bias_z[N])  whose operands lie at constant element strides
// (stride_* between consecutive problems; bias / stride_bias optional): ONE launch of the 128 x 128 kernel with blockIdx.z = z, no
// split-K.  The per-group GEMMs of a grouped Conv1d over strided patch views (adapter/audio.py:57-84, GroupedConv1dSameFn): each of them
// alone fills half the chip.  Rules as op_gemm_nt's plain epilogue (K % 64 == 0, N % 8 == 0, lda / ldb % 8 == 0, ldc % 4 == 0).
int op_gemm_nt_batched(const void* A, int64_t lda, int64_t stride_a, const void* W, int64_t ldb, int64_t stride_b, const void* bias,
                       int64_t stride_bias, void* C, int64_t ldc, int64_t stride_c, int64_t M, int64_t N, int64_t K, int64_t batch,
                       void* stream) {
  OP_CHECK_ARG(A && W && C, "gemm_nt_batched: null A/W/C");
  OP_CHECK_ARG(M >= 0 && N > 0 && K > 0 && batch >= 1 && batch <= 65535, "gemm_nt_batched: bad sizes M=%lld N=%lld K=%lld batch=%lld", (long long)M,
               (long long)N, (long long)K, (long long)batch);
  OP_CHECK_ARG(K % BK == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "gemm_nt_batched: K %% 64, N %% 8, lda / ldb %% 8, ldc %% 4");
  OP_CHECK_ARG(stride_a % 8 == 0 && stride_b % 8 == 0 && stride_c % 4 == 0 && stride_bias % 8 == 0, "gemm_nt_batched: strides must keep 16-byte alignment");
  if (M == 0) return OP_OK;
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = (const bf16_t*)A; a.lda = lda;
  a.B[0] = (const bf16_t*)W; a.ldb = ldb; a.n_seg = (int)N;
  a.bias[0] = (const bf16_t*)bias;
  a.C = C; a.ldc = ldc;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.gm = 4; a.rows_per_sample = 1;
  a.tiles_m = ceil_div(M, BM);
  a.tiles_n = ceil_div(N, 128);
  const int slot = op_prof_begin(0, 2.0 * (double)batch * (double)M * (double)N * (double)K, stream);
  const int rc = launch<EPI_BIAS>(a, 1, (hipStream_t)stream, 1, (int)batch, NtBatch{stride_a, stride_b, stride_c, stride_bias});
  op_prof_end(slot, stream);
  return rc;
}

// Host-only query (no GPU needed): the launch decision op_gemm_nt takes for a dense [M,K] x [N,K]^T problem with a single
// weight segment, contiguous operands and -- when workspace_bytes > 0 -- a split-K scratch of that size.
// plan[0] = tile (128 | 256), plan[1] = K-splits of the (main) launch, plan[2] = 1 if the epilogue runs in the fold kernel,
// plan[3] = rows split off into a second small launch (0 = none).
int op_gemm_plan(int64_t M, int64_t N, int64_t K, int epilogue, int has_bias, int64_t workspace_bytes, int64_t tune, int* plan) {
  const GemmTune T = decode_tune(tune);
  OP_CHECK_ARG(plan && M > 0 && N > 0 && K > 0 && K % BK == 0 && epilogue >= 0 && epilogue <= 3, "gemm_plan: bad arguments");
  const bool off32_ok = (M * K < ((int64_t)1 << 30)) && (N * K < ((int64_t)1 << 30));
  const NtDecision d = decide_nt(M, N, K, epilogue, has_bias != 0, true, off32_ok, N % 8 == 0, workspace_bytes > 0,
                                 workspace_bytes, true, T);
  NtDecision m = d;
  if (d.tail_split)  // the plan of the main launch is taken again for its own row count, as op_gemm_nt does
    m = decide_nt(d.m_main, N, K, epilogue, has_bias != 0, true, off32_ok, N % 8 == 0, workspace_bytes > 0, workspace_bytes, false, T);
  plan[0] = m.plan.tile;
  plan[1] = m.plan.splits;
  plan[2] = (m.plan.splits > 1 && m.fold_epi) ? 1 : 0;
  plan[3] = d.tail_split ? (int)(M - d.m_main) : 0;
  return OP_OK;
}

int op_gemm_nt(const void* A, int64_t lda, const void* B0, const void* B1, const void* B2, int64_t ldb, int64_t n_seg,
               const void* bias0, const void* bias1, const void* bias2, void* C, int64_t ldc, void* h0, void* h1,
               const void* resid, int64_t ldr, const void* gamma, const float* rowscale, int64_t rows_per_sample,
               const float* alpha, int64_t M, int64_t N, int64_t K, int epilogue, void* workspace, int64_t workspace_bytes,
               int64_t tune, const int* resid_rows, int64_t resid_rows_total, void* stream) {
  return gemm_nt_impl(A, lda, B0, B1, B2, ldb, n_seg, bias0, bias1, bias2, C, ldc, h0, h1, resid, ldr, gamma, rowscale,
                      rows_per_sample, alpha, M, N, K, epilogue, workspace, workspace_bytes, stream, 0, true, decode_tune(tune), resid_rows,
                      resid_rows_total);
}


// Grouped NT GEMM: up to three problems C_p[M_p, N] = epilogue(A_p[M_p, K] . W_p[N, K]^T) that share N, K, the leading dimensions
// and the epilogue, in ONE launch of the persistent kernel -- the text / image / audio FFN of an encoder layer
// (transformer_layer.py:203-226; each modality's rows go through its own weights).  All array arguments are HOST arrays with
// nprob entries (B / bias: nprob x 2, [p*2 + 0] = the weight (GeGLU: wi_0), [p*2 + 1] = wi_1 for GeGLU, else unused); h0, h1, resid,
// gamma, rowscale, bias entries may be null, a whole array pointer may be null.  Epilogues 0 (bias), 3 (residual); 2 (GeGLU) returns
// OP_ENOTSUP since round 5 (h1 and the second weight of a problem are unused).  Requirements: N % 256 == 0, K % 64 == 0, K >= 128, lda / ldb / ldc % 8 == 0, every operand below 2 GiB.
// Returns OP_ENOTSUP when the shape does not qualify (the caller then launches the problems one by one).
int op_gemm_nt_grouped(int64_t nprob, const void* const* A, const int64_t* M, int64_t lda, const void* const* B, int64_t ldb,
                       const void* const* bias, void* const* C, int64_t ldc, void* const* h0, void* const* h1,
                       const void* const* resid, int64_t ldr, const void* const* gamma, const float* const* rowscale,
                       const int64_t* rows_per_sample, int64_t N, int64_t K, int epilogue, int64_t tune, const int* const* resid_rows,
                       int64_t resid_rows_total, void* stream) {
  const GemmTune T = decode_tune(tune);
  if (resid_rows) {  // (ABI 9) see op_gemm_nt: all problems or none, one row count for all the full matrices
    OP_CHECK_ARG(epilogue == EPI_RESID && resid_rows_total > 0 && ldr % 8 == 0 &&
                 (resid_rows_total * (ldc > ldr ? ldc : ldr) + N) * 2 < (int64_t)0xfff00000ll,
                 "gemm_nt_grouped: resid_rows: residual epilogue, full matrices below 4 GiB");
    for (int i = 0; i < nprob && i < 3; ++i) OP_CHECK_ARG(resid_rows[i], "gemm_nt_grouped: resid_rows[%d] is null", i);
  }
  OP_CHECK_ARG(nprob >= 1 && nprob <= 3 && A && M && B && C, "gemm_nt_grouped: 1..3 problems, non-null A / M / B / C arrays");
  OP_CHECK_ARG(epilogue == EPI_BIAS || epilogue == EPI_GEGLU || epilogue == EPI_RESID, "gemm_nt_grouped: epilogue %d", epilogue);
  if (epilogue == EPI_GEGLU) {  // round 5: the persistent kernel serves the two epilogues the step groups (plain / bias, residual); a
    op_set_error("gemm_nt_grouped: the GeGLU epilogue has no grouped form (launch op_gemm_nt per problem)");  // GeGLU launch per problem fills the chip
    return OP_ENOTSUP;
  }
  const int bn = 256;
  if (N <= 0 || K < 128 || N % bn != 0 || K % 64 != 0 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0 ||
      N * ldb >= ((int64_t)1 << 30)) {
    op_set_error("gemm_nt_grouped: shape N=%lld K=%lld not supported by the persistent kernel", (long long)N, (long long)K);
    return OP_ENOTSUP;
  }
  GroupArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.nprob = (int)nprob;
  int tiles = 0;
  double rows = 0;
  for (int i = 0; i < 3; ++i) {
    if (i < nprob) {
      OP_CHECK_ARG(M[i] > 0 && A[i] && C[i] && B[i * 2], "gemm_nt_grouped: problem %d: empty or null operand", i);
      if (M[i] * lda >= ((int64_t)1 << 30) || M[i] * ldc >= ((int64_t)1 << 30)) {
        op_set_error("gemm_nt_grouped: problem %d too large for 32-bit offsets", i);
        return OP_ENOTSUP;
      }
      ga.A[i] = (const bf16_t*)A[i]; ga.M[i] = (int)M[i];
      ga.B[i][0] = (const bf16_t*)B[i * 2]; ga.B[i][1] = (const bf16_t*)B[i * 2 + 1];
      ga.bias[i][0] = bias ? (const bf16_t*)bias[i * 2] : nullptr;
      ga.C[i] = C[i]; ga.H0[i] = h0 ? (bf16_t*)h0[i] : nullptr; ga.H1[i] = h1 ? (bf16_t*)h1[i] : nullptr;
      ga.resid[i] = resid ? (const bf16_t*)resid[i] : nullptr;
      if (epilogue == EPI_RESID) OP_CHECK_ARG(ga.resid[i], "gemm_nt_grouped: residual epilogue needs resid");
      ga.gamma[i] = gamma ? (const bf16_t*)gamma[i] : nullptr;
      ga.rowscale[i] = rowscale ? rowscale[i] : nullptr;
      ga.rows_per_sample[i] = rows_per_sample && rows_per_sample[i] > 0 ? (int)rows_per_sample[i] : 1;
      ga.rows[i] = resid_rows ? resid_rows[i] : nullptr;
      tiles += ceil_div(M[i], 256);
      rows += (double)M[i];
    } else {
      ga.rows_per_sample[i] = 1;
    }
    ga.mt_end[i] = tiles;
  }
  ga.tiles_m = tiles;
  ga.tiles_n = (int)(N / bn);
  ga.lda = lda; ga.ldb = ldb; ga.ldc = ldc; ga.ldr = ldr; ga.alpha = nullptr;
  ga.n_seg = (int)N; ga.N = (int)N; ga.K = (int)K;
  ga.gm = ga.tiles_n <= 8 ? 1 : 8;
  const int slot = op_prof_begin(0, 2.0 * rows * (double)N * (double)K, stream);
  // persistent (one workgroup per CU walks the tile list, K-tile stream continuous across tile boundaries) unless the caller asks
  // for one tile per workgroup (tune sched = 7; tests, A/B): with the round-3 epilogue (coalesced non-temporal stores: a short
  // drain in front of the next tile's loads) the persistent form is 2.2 % faster on both grouped launches of the step
  // (tools/gemm_grouped_bench.py: down-projection + residual 1.0856 -> 1.0619 ms, K = 6144 dgrad 1.0055 -> 0.9825 ms)
  const int rc = launch256p_any(ga, resid_rows ? (int)EPI_RESID_ROWS : epilogue, (hipStream_t)stream, T.sched != 7);
  op_prof_end(slot, stream);
  return rc;
}

// C[M,N] (bf16, ldc) = A^T B with A [K, M] (lda) and B [K, N] (ldb) both row-major bf16: the weight-gradient GEMM
// dW[out,in] = dy[tokens,out]^T x[tokens,in] (autograd of nn.Linear) without transposed operand copies.
// Requirements: K % 64 == 0, M % 8 == 0, N % 8 == 0, lda/ldb % 8 == 0.  Returns OP_ENOTSUP (-95) when the shape does not
// qualify (the caller then uses op_transpose + op_gemm_nt).  accumulate != 0: C += A^T B (gradient accumulation into a
// pre-existing buffer).  workspace: optional fp32 scratch enabling split-K.
int op_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
               int accumulate, void* workspace, int64_t workspace_bytes, int64_t tune, void* stream) {
  const GemmTune T = decode_tune(tune);
  OP_CHECK_ARG(A && B && C, "gemm_tn: null pointer");
  if (K % 64 != 0 || M % 8 != 0 || N % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0 || M < 8 || N < 8 ||
      31 * lda + M >= ((int64_t)1 << 30) || 31 * ldb + N >= ((int64_t)1 << 30)) {
    op_set_error("gemm_tn: shape M=%lld N=%lld K=%lld not supported by the transpose-read kernel", (long long)M, (long long)N,
                 (long long)K);
    return OP_ENOTSUP;
  }
  if (M == 0 || N == 0) return OP_OK;
  GemmArgs a;
  a.A = (const bf16_t*)A; a.lda = lda;
  a.B[0] = (const bf16_t*)B; a.B[1] = a.B[2] = nullptr; a.ldb = ldb; a.n_seg = (int)N;
  a.bias[0] = a.bias[1] = a.bias[2] = nullptr;
  a.C = C; a.ldc = ldc; a.H0 = a.H1 = nullptr; a.resid = nullptr; a.ldr = 0; a.gamma = nullptr; a.rowscale = nullptr;
  a.rows_per_sample = 1; a.alpha = nullptr; a.rows = nullptr;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.gm = 4; a.m_off = 0;
  a.tiles_m = ceil_div(M, 256);
  a.tiles_n = ceil_div(N, 256);
  a.gm = T.gm > 0 ? T.gm : (a.tiles_n <= 8 ? 1 : 8);
  // split-K: fill the chip (256 slots per round) while keeping chunks long and even
  const int nk = (int)(K / BK2);
  const int64_t tiles = (int64_t)a.tiles_m * a.tiles_n;
  int best_s = 1;
  double best_t = 1e300;
  for (int s = 1; s <= 16; ++s) {
    if (s > 1 && (!workspace || (int64_t)s * M * N * 4 > workspace_bytes || nk / s < 16)) break;
    int kps = ceil_div(nk, s);
    kps += kps & 1;
    const int eff = ceil_div(nk, kps);
    const double rounds = (double)((tiles * eff + 255) / 256);
    double t = rounds * kps * (256.0 * 256 / 1040.0) * BK2;
    if (eff > 1) t += (double)eff * M * N * 8.0 / 4.0e3 + 1.0e4;
    if (t < best_t) { best_t = t; best_s = eff; a.kt_per_split = eff > 1 ? kps : 0; }
  }
  a.slab = (int64_t)M * N;
  hipStream_t s = (hipStream_t)stream;
  // kernel flavour (tune bits 2-3): 0 auto, 2 eight waves of 128 x 64, 3 four waves of 128 x 128 (bit-identical results)
  // measured at K = 32 896 tokens: four waves +14 % on 36 output tiles (1536 x 1536), +6 % on 108, -1 % on 144; -2 % at K = 8 192
  const bool four_waves = T.fullline == 3 || (T.fullline == 2 && tiles <= 108 && K >= 16384);
  const int slot = op_prof_begin(0, 2.0 * (double)M * (double)N * (double)K, stream);
  int rc;
  if (best_s > 1) {
    a.C = workspace;
    a.ldc = N;
    rc = launch256_tn<EPI_F32>(a, s, best_s, four_waves);
    if (rc == OP_OK) {
      int64_t nb = ((int64_t)M * (N / 8) + 255) / 256;
      if (nb > 2048) nb = 2048;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)nb), dim3(256), 0, s, (const float*)workspace, best_s, a.slab,
                         (bf16_t*)C, ldc, (int)M, (int)N, accumulate);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { op_set_error("gemm_tn: split-K reduce launch failed: %s", hipGetErrorString(e)); rc = (int)e; }
    }
  } else if (accumulate) {  // C += A^T B in place through the residual epilogue
    a.kt_per_split = 0;
    a.resid = (const bf16_t*)C;
    a.ldr = ldc;
    rc = launch256_tn<EPI_RESID>(a, s, 1, four_waves);
  } else {
    a.kt_per_split = 0;
    rc = launch256_tn<EPI_BIAS>(a, s, 1, four_waves);
  }
  op_prof_end(slot, stream);
  return rc;
}

// Host: the run tables of gemm256w_tn_grouped_kernel (see its header).  ga.pr[] is filled, problems sorted by K (longest first).
// per_xcd = workgroups per queue; solo: round 4's form (waves of a multiple of the group size, the other workgroups draw from the back).
static int tn_build_schedule(TnGroupArgs& ga, int nwg, bool solo) {
  const int per_xcd = nwg / 8 > 0 ? nwg / 8 : 1;
  const TnGeom G0 = tn_geom(ga.pr[0].tiles_m, ga.pr[0].tiles_n);
  int wave = per_xcd;
  ga.solo_from = 1 << 30;
  if (solo && G0.csz >= 2 && per_xcd >= G0.csz) ga.solo_from = wave = (per_xcd / G0.csz) * G0.csz;
  for (;; wave *= 2) {
    std::vector<TnRun> q[8];
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    while (i < ga.nprob) {
      int j = i;  // problems i .. j-1: one K class, cut into waves as ONE slot list
      while (j < ga.nprob && ga.pr[j].K == ga.pr[i].K) ++j;
      int pi = i, s0 = 0;
      auto slots_of = [&](int k) { const TnGeom G = tn_geom(ga.pr[k].tiles_m, ga.pr[k].tiles_n); return G.ng * G.csz; };
      while (pi < j) {
        int x = 0;
        for (int y = 1; y < 8; ++y)
          if (load[y] < load[x]) x = y;
        int left = wave;
        while (left > 0 && pi < j) {
          const int take = std::min(left, slots_of(pi) - s0);
          if (!q[x].empty() && (q[x].back().prob_n >> 24) == pi && q[x].back().slot0 + (q[x].back().prob_n & 0xffffff) == s0)
            q[x].back().prob_n += take;
          else
            q[x].push_back(TnRun{(pi << 24) | take, s0});
          load[x] += (double)take * ((double)ga.pr[pi].K + 400.0);
          left -= take;
          s0 += take;
          if (s0 == slots_of(pi)) { ++pi; s0 = 0; }
        }
      }
      i = j;
    }
    size_t total = 0;
    for (int x = 0; x < 8; ++x) total += q[x].size();
    if (total > (size_t)TN_MAX_RUNS) continue;  // (tiny forced workgroup counts: coarser waves)
    int r = 0;
    for (int x = 0; x < 8; ++x) {
      ga.run_begin[x] = (short)r;
      ga.qlen[x] = 0;
      for (const TnRun& t : q[x]) { ga.runs[r++] = t; ga.qlen[x] += t.prob_n & 0xffffff; }
    }
    ga.run_begin[8] = ga.run_begin[9] = (short)r;
    return wave;
  }
}

// Bytes of the counter block op_gemm_tn_grouped needs: device memory the caller zeroes ONCE; every launch leaves it zeroed.
// One block per stream that may run the op (launches on one stream are ordered, the block is re-armed by the launch itself).
int64_t op_gemm_tn_grouped_counter_bytes(void) { return (int64_t)(9 * TN_CTR_STRIDE) * 8; }

// Host-only query (works without a GPU): the tile queues op_gemm_tn_grouped builds for these problem sizes and `workgroups` (0: one
// per CU, 256 on a host without a device; bit 10 of `tune` as in op_gemm_tn_grouped).  Writes, queue by queue (0..7) and in draw
// order, one record of four int32 per tile: queue, problem (the caller's index), tile row, tile column; returns the number of
// records (every output tile of every problem appears exactly once), or -22 when `cap` records do not suffice.
int64_t op_gemm_tn_grouped_plan(int64_t nprob, const int64_t* M, const int64_t* N, const int64_t* K, int64_t workgroups, int64_t tune,
                                int32_t* out, int64_t cap) {
  if (nprob < 1 || nprob > TN_MAX_PROB || !M || !N || !K || !out) return OP_EINVAL;
  int order[TN_MAX_PROB];
  for (int i = 0; i < (int)nprob; ++i) order[i] = i;
  for (int i = 1; i < (int)nprob; ++i)
    for (int j = i; j > 0 && K[order[j]] > K[order[j - 1]]; --j) { const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp; }
  TnGroupArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.nprob = (int)nprob;
  int64_t tiles = 0;
  for (int i = 0; i < (int)nprob; ++i) {
    ga.pr[i].tiles_m = ceil_div(M[order[i]], 256);
    ga.pr[i].tiles_n = ceil_div(N[order[i]], 256);
    ga.pr[i].K = (int)K[order[i]];
    tiles += (int64_t)ga.pr[i].tiles_m * ga.pr[i].tiles_n;
  }
  int nwg = workgroups > 0 ? (int)workgroups : num_cus();
  if (nwg > tiles) nwg = (int)tiles;
  tn_build_schedule(ga, nwg, ((tune >> 10) & 1) != 0);
  int64_t n = 0;
  for (int x = 0; x < 8; ++x) {
    const int len = tn_queue_len(ga, x);
    for (int q = 0; q < len; ++q) {
      int prob, tm, tn;
      if (!tn_decode(ga, x, q, prob, tm, tn)) continue;
      if (n >= cap) return OP_EINVAL;
      out[4 * n] = x; out[4 * n + 1] = order[prob]; out[4 * n + 2] = tm; out[4 * n + 3] = tn;
      ++n;
    }
  }
  return n;
}

// Up to 16 weight-gradient GEMMs  C_i[M_i,N_i] (bf16, ldc_i) (+)= A_i^T B_i  (A_i [K_i, M_i], B_i [K_i, N_i] row-major bf16: dy and
// x of nn.Linear, autograd's dW = dy^T x) as ONE persistent launch without split-K (gemm256w_tn_grouped_kernel): every output
// tile runs its whole K and is written / accumulated once.  Shape rules per problem as op_gemm_tn; returns OP_ENOTSUP (nothing
// launched) when a problem does not qualify -- the caller then uses op_gemm_tn per problem.  Problems may come in any order.
// The gradient is read-modify-written in 16-byte pieces: ldc_i % 8 == 0 and C_i 16-byte aligned as well.
// tune: bits 0-9 = forced number of workgroups (0: one per CU, at most one per tile); bit 10 = round 4's solo workgroups (see the kernel).
int op_gemm_tn_grouped(int64_t nprob, const void* const* A, const int64_t* lda, const void* const* B, const int64_t* ldb, void* const* C,
                       const int64_t* ldc, const int64_t* M, const int64_t* N, const int64_t* K, const int32_t* accumulate,
                       const void* const* W, const int64_t* ldw, float* const* rowdot, const void* const* rscale, void* counters, int64_t tune,
                       void* stream) {
  OP_CHECK_ARG(nprob >= 1 && nprob <= TN_MAX_PROB, "gemm_tn_grouped: %lld problems (1 ... %d)", (long long)nprob, TN_MAX_PROB);
  OP_CHECK_ARG(A && lda && B && ldb && C && ldc && M && N && K && accumulate && counters, "gemm_tn_grouped: null pointer");
  int order[TN_MAX_PROB];
  for (int i = 0; i < (int)nprob; ++i) {
    OP_CHECK_ARG(A[i] && B[i] && C[i], "gemm_tn_grouped: null operand of problem %d", i);
    if (K[i] % 64 != 0 || K[i] < 64 || M[i] % 8 != 0 || N[i] % 8 != 0 || lda[i] % 8 != 0 || ldb[i] % 8 != 0 || M[i] < 8 || N[i] < 8 ||
        31 * lda[i] + M[i] >= ((int64_t)1 << 30) || 31 * ldb[i] + N[i] >= ((int64_t)1 << 30) || ldc[i] % 8 != 0 || ((uintptr_t)C[i] & 15) != 0) {
      op_set_error("gemm_tn_grouped: problem %d (M=%lld N=%lld K=%lld) not supported by the transpose-read kernel", i, (long long)M[i],
                   (long long)N[i], (long long)K[i]);
      return OP_ENOTSUP;
    }
    order[i] = i;
  }
  for (int i = 1; i < (int)nprob; ++i)  // longest K first (stable insertion sort): greedy list scheduling wants the long tiles early
    for (int j = i; j > 0 && K[order[j]] > K[order[j - 1]]; --j) { const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp; }
  TnGroupArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.nprob = (int)nprob;
  ga.ctr = (unsigned long long*)counters;
  double work = 0.0;
  int64_t tiles = 0;
  for (int i = 0; i < (int)nprob; ++i) {
    const int s = order[i];
    TnProb& q = ga.pr[i];
    q.A = (const bf16_t*)A[s]; q.B = (const bf16_t*)B[s]; q.C = (bf16_t*)C[s];
    q.lda = lda[s]; q.ldb = ldb[s]; q.ldc = ldc[s];
    q.M = (int)M[s]; q.N = (int)N[s]; q.K = (int)K[s];
    q.tiles_m = ceil_div(M[s], 256); q.tiles_n = ceil_div(N[s], 256);
    q.accumulate = accumulate[s] != 0;
    if (rowdot != nullptr && rowdot[s] != nullptr) {  // (the side product rides on the full-tile accumulate epilogue)
      OP_CHECK_ARG(W != nullptr && ldw != nullptr && W[s] != nullptr && q.accumulate && M[s] % 256 == 0 && N[s] % 256 == 0 && ldw[s] % 8 == 0 &&
                       ((uintptr_t)W[s] & 15) == 0,
                   "gemm_tn_grouped: problem %d: rowdot needs W (16-byte aligned, ldw %% 8 == 0), accumulate and M, N multiples of 256", s);
      q.W = (const bf16_t*)W[s]; q.ldw = ldw[s]; q.rowdot = rowdot[s];
      q.rscale = rscale != nullptr ? (const bf16_t*)rscale[s] : nullptr;
    } else {
      OP_CHECK_ARG(rscale == nullptr || rscale[s] == nullptr, "gemm_tn_grouped: problem %d: rscale rides on the rowdot epilogue (rowdot is null)", s);
    }
    tiles += (int64_t)q.tiles_m * q.tiles_n;
    work += 2.0 * (double)M[s] * (double)N[s] * (double)K[s];
  }
  int nwg = (int)(tune & 1023);
  if (nwg <= 0) nwg = num_cus();
  if (nwg > tiles) nwg = (int)tiles;
  tn_build_schedule(ga, nwg, ((tune >> 10) & 1) != 0);
  const size_t sh = STAGES2 * STAGE2_BYTES;
  OP_ENSURE_LDS(gemm256w_tn_grouped_kernel, (int)sh, "gemm_tn_grouped");
  const int slot = op_prof_begin(0, work, stream);
  hipLaunchKernelGGL(gemm256w_tn_grouped_kernel, dim3(nwg), dim3(256), sh, (hipStream_t)stream, ga);
  op_prof_end(slot, stream);
  OP_LAUNCH_CHECK();
  return OP_OK;
}

}  // extern "C"

#ifdef OP_GEMM_TIMELINE
extern "C" int op_debug_gemm_timeline(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &buf, sizeof(buf));
}
#endif
