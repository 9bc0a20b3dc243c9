// Shared by csrc/gemm.hip and csrc/fp8.hip: the argument block of the NT GEMM kernels and the epilogue of the four-wave 256 x 256
// kernels (gemm256v_kernel / gemm256p_kernel in bf16, gemm256f8_kernel in fp8).  Everything lives in an anonymous namespace: each
// translation unit gets its own copy.
#pragma once
#include "common.h"
#include <type_traits>

namespace {

// EPI_RESID_ROWS (internal: op_gemm_nt's residual epilogue with resid_rows, ABI 9): the residual is read from, and the output written
// to, row rows[m] of LARGER matrices (resid / C are their bases; rows[m] < 0: the row is computed and dropped) -- the packed rows of
// the samples a residual branch keeps under stochastic depth go straight back to their places in the full activation matrix.
enum { EPI_BIAS = 0, EPI_F32 = 1, EPI_GEGLU = 2, EPI_RESID = 3, EPI_RESID_ROWS = 4 };
constexpr bool epi_is_resid(int e) { return e == EPI_RESID || e == EPI_RESID_ROWS; }

struct GemmArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* B[3]; int64_t ldb; int n_seg;
  const bf16_t* bias[3];
  void* C; int64_t ldc;
  bf16_t* H0; bf16_t* H1;
  const bf16_t* resid; int64_t ldr;
  const bf16_t* gamma; const float* rowscale; int rows_per_sample;
  const float* alpha;
  int M, N, K;
  int tiles_m, tiles_n;
  int kt_per_split;   // K-tiles handled by one workgroup (split-K along blockIdx.y); 0 = all
  int64_t slab;       // elements between split-K output slabs
  int gm;             // 256x256 kernels: M-tiles per L2 group (tile order: gm M-tiles x all N-tiles, M fastest)
  int m_off;          // row index of A's first row in the caller's matrix (rowscale lookup of a tail-rows launch)
  const int* rows;    // EPI_RESID_ROWS: the row of resid / C that stands behind row m of the launch (-1: none); else unused
};

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// resid + rowscale * gamma * y with the roundings pinned (product of the two scales, then one fused multiply-add): every epilogue
// that applies the residual form -- in-kernel or in the split-K fold -- gives the same bits whatever the optimiser would contract.
__device__ __forceinline__ float resid_out(float r, float rs, float gv, float y) { return __builtin_fmaf(__fmul_rn(rs, gv), y, r); }

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue of the FOUR-WAVE kernels (gemm256v / gemm256p; plain / bias and residual forms): same arithmetic, operation for
// operation, as gemm_epilogue above (bit-identical output), with the accumulators in the OTHER orientation.
//
// What the time stamps inside gemm256v_kernel said (tools/gemm_timeline.py, profiles/r3_gemm_timeline.txt): the shared epilogue
// costs 4.3 us per tile in its plain form and 13 us in the residual form even with 48 workgroups on the chip, 5-6 / 16-18 us
// with 256 -- 11 % of a K = 1536 launch.  Ablations of the same build: the arithmetic alone (accumulator reads, bias, bf16
// conversion; stores replaced by a register sink) takes 0.9 us, the 32 stores alone (of a constant) take 4.2 us.  Re-writing the
// arithmetic (1425 -> 630 instructions) and making the four lanes of a row cover 64 contiguous bytes changed nothing: the
// texture addresser coalesces ADJACENT lanes only, and in the "weights as first operand" orientation adjacent lanes (t, t + 1)
// are different ROWS -- every lane's 16 bytes went to the L2 as a request of their own, 64 requests per instruction, ~69 cycles
// per store instruction and CU.
//
// So the kernels that end here issue their MFMAs with the ACTIVATION fragment as first operand (same fragments, same LDS
// reads, operands swapped): accumulator acc[blk][ni][mi][r] of lane (g, t) is then row mi*16 + g*4 + r of the wave's 128 rows
// and the weight-tile row blk*64 + ni*16 + t, and the weight rows are staged so that this is COLUMN t*8 + blk*4 + ni of the
// wave's 128 columns (VMAP).  Per (mi, r) a lane holds 8 contiguous columns = one 16-byte store; the 16 lanes of a g-group
// write one whole 256-byte row segment, a store instruction writes four of them: fully coalesced, 32 stores per wave.
// C, the residual and the branch output go through buffer descriptors based at the wave's first row / column whose size ends
// at the last valid row (rows >= M are dropped / read as zero by the hardware: no guards); the per-lane offset is ONE VGPR,
// the row of a (mi, r) pair an SGPR offset.  N % 256 == 0 and n_seg % 256 == 0 (launch conditions).
//
// NOTE on the stores: `buffer_store_dwordx4 ... s_off offen` reads its data VGPRs late, and hipcc (ROCm 7.2) does not keep a
// following VALU write of those VGPRs away from it -- neither across a block boundary nor inside a block (its hazard table has
// the rule for an immediate offset only).  Seen as dword 1 of lanes 12-15 of every 16-lane row of a store going out overwritten
// (tools/gemm_epi_debug.py), first behind a per-store branch, then in straight-line code with a separate `s_nop` statement that
// the scheduler had moved behind the overwriting instruction.  Every store here is therefore ONE asm statement that contains
// its own wait states (store_b128_padded) -- behind it for the data VGPRs, and in front of it for its SGPR operands, which an
// opaque asm statement does not get from the compiler either -- and tools/check_mfma_hazards.py looks for both patterns in the
// compiled ISA.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void unpack_bf16x8(const u32x4& raw, float (&f)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f[2 * q] = __builtin_bit_cast(float, raw[q] << 16);
    f[2 * q + 1] = __builtin_bit_cast(float, raw[q] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack_bf16x8(const float (&f)[8]) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (bf16_t)f[j];
  return __builtin_bit_cast(u32x4, v);
}

// 16-byte buffer store + the wait states that keep the next writer of its data VGPRs away (one asm statement: a separate
// s_nop was scheduled BEHIND such writers; see the note above).  Descriptor: raw buffer, base / size in bytes.
__device__ __forceinline__ u32x4 raw_rsrc(const void* base, int nbytes) {
  const uint64_t a = (uint64_t)base;
  return (u32x4){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) & 0xffffu,
                 (unsigned)__builtin_amdgcn_readfirstlane(nbytes), 0x00020000u};
}
__device__ __forceinline__ void store_b128_padded(const u32x4& data, const u32x4& rsrc, int voff, int soff) {
  // (s_nop 4 in front: the SGPR operands may have been written by the SALU / v_readfirstlane just before -- 5 wait states that the
  // hazard recogniser cannot insert for an instruction it does not see; without them a store went out with the PREVIOUS row offset)
  // `nt`: the output is written once and not read again by this launch -- without the hint every round leaves 128 KiB of dirty
  // lines per CU (the whole 4 MiB of an XCD's L2) in front of the operand panels (tools/gemm_lib_ab.py: -5 % on the N = 12288
  // up-projection, -10 % on the K = 1536 residual launch, +-1 % elsewhere; sc1 / sc0 sc1 nt measured no better)
#ifndef OP_EXP_STORE_BITS
#define OP_EXP_STORE_BITS " nt"
#endif
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen" OP_EXP_STORE_BITS "\n\ts_nop 1" ::"v"(data), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// SCALED (fp8 operands, csrc/fp8.hip): the accumulator is dequantised by the row scale of the activation row and the row scale of the
// weight row (= output column) before anything else:  y = acc * (sa[m] * sb[n]) (+ bias ...).  sa is indexed by the launch's row, sb by
// the column inside the tile's weight segment (like the bias vector).
template <int EPI, bool SCALED = false>
__device__ __forceinline__ void epilogue_v(const GemmArgs& p, f32x4 (&acc)[2][4][8], int mrow0, int ncol0, int g, int t,
                                           const bf16_t* bias_of_tile = nullptr, const bool tile_bias = false, const float* sa = nullptr,
                                           const float* sb = nullptr) {
  // tile_bias (a compile-time constant at every call site): the caller hands over the bias vector of the weight segment its tile
  // lies in (n_seg % 256 == 0) instead of the table p.bias[] -- gemm256p_kernel builds its GemmArgs per tile in registers, and ONE
  // dynamically indexed member put the whole struct into scratch (round 4: 232-288 bytes per lane, 16-28 scratch operations per tile)
  static_assert(EPI == EPI_BIAS || epi_is_resid(EPI), "epilogue_v: plain / bias and residual epilogues only");
  constexpr bool ROWS = EPI == EPI_RESID_ROWS;
  const int rows_left = min(p.M - mrow0, 128);  // wave-uniform
  if (rows_left <= 0) return;
  const int ldc = (int)p.ldc;
  const int nrec = ((rows_left - 1) * ldc + 128) * 2;
  // ROWS: the descriptor starts at the matrix base (+ the tile's first column) and spans "everything" (the host checked that the whole
  // matrix lies below ROWS_SPAN bytes); the row goes into the per-lane offset, a row without a place gets an offset behind the span:
  // its store is dropped / its load returns zeros -- the same mechanism that guards rows >= M in the plain form
  constexpr unsigned ROWS_SPAN = 0xfffff000u, ROWS_NONE = 0xfffffff0u;
  const u32x4 rc = ROWS ? raw_rsrc((bf16_t*)p.C + ncol0, (int)ROWS_SPAN) : raw_rsrc((bf16_t*)p.C + (int64_t)mrow0 * p.ldc + ncol0, nrec);
  const int voff = (g * 4 * ldc + t * 8) * 2;  // + ((mi*16 + r) * ldc) * 2 as the scalar offset
  const int seg = ncol0 / p.n_seg;
  const bf16_t* bp = tile_bias ? bias_of_tile : p.bias[seg];
  // the lane's 8 columns: acc[j >> 2][j & 3][mi][r] <-> column t*8 + j
  // (accumulator reads as volatile asm: they keep their place between the -- volatile -- stores.  Plain reads are hoisted ahead of
  // the whole epilogue by the register allocator's live-range splitting and the surplus spilled: the last 12-24 bytes of scratch)
  float sbv[8];
  u32x4 sav[8];  // row scales of the lane's 32 rows: rows mi*16 + g*4 .. + 3 are four consecutive floats (one 16-byte load per mi)
  if constexpr (SCALED) {
    const float* sp = sb + (ncol0 - seg * p.n_seg) + t * 8;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { sbv[j] = s0[j]; sbv[4 + j] = s1[j]; }
    // all of them up front through a descriptor that ends at the last valid row (rows past M read 0: never stored).  One load per
    // row between the stores -- which carry a "memory" clobber -- would expose a full memory latency 32 times per wave.
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(sa + mrow0), 0, rows_left * 4, 0x00020000);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) sav[mi] = __builtin_amdgcn_raw_buffer_load_b128(rsa, g * 16, mi * 64, 0);
  }
  auto row_of = [&](int mi, int r, float (&o)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(o[j]) : "a"(acc[j >> 2][j & 3][mi][r]));
    if constexpr (SCALED) {
      const unsigned sbits = sav[mi][r];  // (through a scalar: __builtin_bit_cast of a vector-element lvalue reads element 0 -- hipcc 7.2)
      const float sr = __builtin_bit_cast(float, sbits);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] *= sr * sbv[j];
    }
  };

  if constexpr (EPI == EPI_BIAS) {
    auto stores = [&](auto bias_tag) {
      constexpr bool BIAS = decltype(bias_tag)::value;
      float bv[8];
      if constexpr (BIAS) unpack_bf16x8(*reinterpret_cast<const u32x4*>(bp + (ncol0 - seg * p.n_seg) + t * 8), bv);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        // (one fragment row block at a time: left alone the scheduler hoists accumulator reads of later blocks over the stores and
        // spills what does not fit -- the four-wave kernels own all 256 + 256 registers)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float o[8];
          row_of(mi, r, o);
          if constexpr (BIAS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += bv[j];
          }
#if defined(OP_EXP_EPI) && OP_EXP_EPI == 1  // (tools/gemm_timeline.py ablations: 1 = arithmetic without the stores, 2 = stores without arithmetic)
          { const u32x4 pk = pack_bf16x8(o); asm volatile("" ::"v"(pk)); }
#elif defined(OP_EXP_EPI) && OP_EXP_EPI == 2
          store_b128_padded((u32x4){0u, 0u, 0u, 0u}, rc, voff, (mi * 16 + r) * ldc * 2);
#else
          store_b128_padded(pack_bf16x8(o), rc, voff, (mi * 16 + r) * ldc * 2);
#endif
        }
      }
    };
    if (bp) stores(std::true_type{});
    else stores(std::false_type{});
  } else {
    const int ldr = (int)p.ldr;
    const __amdgpu_buffer_rsrc_t rr =
        ROWS ? __builtin_amdgcn_make_buffer_rsrc((void*)(p.resid + ncol0), 0, (int)ROWS_SPAN, 0x00020000)
             : __builtin_amdgcn_make_buffer_rsrc((void*)(p.resid + (int64_t)mrow0 * p.ldr + ncol0), 0, ((rows_left - 1) * ldr + 128) * 2, 0x00020000);
    const int voff_r = (g * 4 * ldr + t * 8) * 2;
    // ROWS: the lane's 32 rows are mi*16 + g*4 + r: four consecutive entries of the row table per mi (one 16-byte load each); the
    // entries of a chunk are requested TWO chunks ahead (its residual loads, one chunk ahead, depend on them) and live until its
    // stores: 24 registers instead of 32 (all up front: 20 bytes of scratch per lane in the persistent kernel).  Entries behind the
    // last row of the launch read 0 and are masked by `rows_left`.
    u32x4 rowv[ROWS ? 8 : 1];
    const __amdgpu_buffer_rsrc_t rm =
        __builtin_amdgcn_make_buffer_rsrc((void*)(ROWS ? p.rows + mrow0 : nullptr), 0, ROWS ? rows_left * 4 : 0, 0x00020000);
    auto load_rows = [&](int c) {
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) rowv[ROWS ? c * 2 + m2 : 0] = __builtin_amdgcn_raw_buffer_load_b128(rm, g * 16, (c * 2 + m2) * 64, 0);
    };
    if constexpr (ROWS) {
      load_rows(0);
      load_rows(1);
    }
    auto row_off = [&](int mi, int r, int ld) -> int {  // ROWS: byte offset of the lane's 16 bytes in row rows[...] (or behind the span)
      const unsigned sv = rowv[ROWS ? mi : 0][r];  // (through a scalar: see row_of)
      const int src = (int)sv;
      const bool ok = src >= 0 && mi * 16 + g * 4 + r < rows_left;
      return ok ? (int)((unsigned)src * (unsigned)ld * 2u + (unsigned)t * 16u) : (int)ROWS_NONE;
    };
    // rowscale index of a row by multiply-high: exact while (rows + m_off) * rows_per_sample < 2^32 (else a true division)
    const unsigned rps = (unsigned)p.rows_per_sample;
    const bool exact = p.rowscale && (uint64_t)((unsigned)(p.M + p.m_off)) * rps < (1ull << 32) && rps > 1;
    const unsigned magic = exact ? 0xffffffffu / rps + 1u : 0u;
    // residual rows (and their row scales) in chunks of two 16-row fragments (8 rows per lane: 8 loads, 32 + 8 VGPRs), one chunk
    // ahead of the arithmetic: the first chunk's latency is exposed (together with the bias / gamma loads), the later ones hide
    u32x4 rraw[2][2][4];
    float rsv[2][2][4];
    auto load_chunk = [&](int c, u32x4 (&dst)[2][4], float (&rs)[2][4]) {  // chunk c: fragments mi = 2c, 2c + 1
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (c * 2 + m2) * 16 + r;  // + g * 4 in the lane offset
          // (plain loads: a non-temporal hint made the K = 1536 residual launch 7 % slower, tools/gemm_lib_ab.py)
          if constexpr (ROWS) dst[m2][r] = __builtin_amdgcn_raw_buffer_load_b128(rr, row_off(c * 2 + m2, r, ldr), 0, 0);
          else dst[m2][r] = __builtin_amdgcn_raw_buffer_load_b128(rr, voff_r, row * ldr * 2, 0);
          if (p.rowscale) {
            const unsigned mc = (unsigned)(min(mrow0 + row + g * 4, p.M - 1) + p.m_off);
            rs[m2][r] = p.rowscale[exact ? __umulhi(mc, magic) : mc / rps];
          } else {
            rs[m2][r] = 1.f;
          }
        }
    };
    load_chunk(0, rraw[0], rsv[0]);
    float bv[8], gv[8];
    {
      u32x4 braw = (u32x4){0u, 0u, 0u, 0u}, graw = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
      if (bp) braw = *reinterpret_cast<const u32x4*>(bp + (ncol0 - seg * p.n_seg) + t * 8);
      if (p.gamma) graw = *reinterpret_cast<const u32x4*>(p.gamma + ncol0 + t * 8);
      unpack_bf16x8(braw, bv);
      unpack_bf16x8(graw, gv);
    }
    const bool has_y = p.H0 != nullptr;
    const u32x4 ry = raw_rsrc(p.H0 + (int64_t)mrow0 * p.ldc + ncol0, has_y ? nrec : 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if constexpr (ROWS) {
        if (c + 2 < 4) load_rows(c + 2);
      }
      if (c + 1 < 4) load_chunk(c + 1, rraw[(c + 1) & 1], rsv[(c + 1) & 1]);
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) {
        __builtin_amdgcn_sched_barrier(0);  // (as above)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mi = c * 2 + m2;
          const int soff = (mi * 16 + r) * ldc * 2;
          const float rs = rsv[c & 1][m2][r];
          float o[8], rv[8];
          row_of(mi, r, o);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += bv[j];
          if (has_y) store_b128_padded(pack_bf16x8(o), ry, voff, soff);  // branch output y (pre layer-scale)
          unpack_bf16x8(rraw[c & 1][m2][r], rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = resid_out(rv[j], rs, gv[j], o[j]);
          if constexpr (ROWS) store_b128_padded(pack_bf16x8(o), rc, row_off(mi, r, ldc), 0);
          else store_b128_padded(pack_bf16x8(o), rc, voff, soff);
        }
      }
    }
  }
}

}  // namespace
