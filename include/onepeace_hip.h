/* onepeace_hip.h -- C ABI of libonepeace_hip.so: the ONE-PEACE hot path as hand-written HIP for MI355X (gfx950).
 *
 * The reference (OFA-Sys/ONE-PEACE @ 2024-10-08) has no FFI of its own: its "operator API" is three import-time seams
 * in Python.  This header is the boundary a binding for those seams talks to; every entry point
 * names the reference code it replaces (file:line relative to the reference tree).  The Python binding that ships
 * here is one-peace_amd/hip.py (ctypes); INTEGRATION.md shows the reference-side stubs.
 *
 * Contract (all functions):
 *   - plain pointers + sizes, no framework types.  Pointers are DEVICE pointers unless stated otherwise.
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library never allocates, frees or keeps a
 *     pointer past the call.
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); no internal synchronisation.
 *     Stateless and re-entrant: no tuning knobs or caches live in the library (kernel flavours are selected by the per-call
 *     `tune` word of the GEMM / attention entry points, 0 = production defaults); safe from the Python main thread,
 *     autograd's backward thread and recompute.  The only process-wide state is the opt-in launch profiler (op_prof_*:
 *     mutex-protected event log, off by default) and hipFuncSetAttribute bookkeeping (idempotent).
 *   - return 0 on success, a negative errno-style code (-22 EINVAL, -95 ENOTSUP) or a positive hipError_t otherwise;
 *     never throws, never aborts.  op_last_error() gives the thread-local message.
 *   - dtype codes: 0 = bf16, 1 = f32.  Matrices are row-major; `ld*` are row strides in ELEMENTS.
 */
#ifndef ONEPEACE_HIP_H
#define ONEPEACE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ------------------------------------------------------------------------------------------------ */
int op_abi_version(void); /* 2: per-call `tune` words replaced the process-wide knob setters; 3: op_gemm_nt_grouped, op_ln_geglu_fwd, ldd / ldh of op_ln_geglu_bwd, `out` of op_attn_bwd; 4: op_gemm_tn_grouped; 5: op_probe_mfma_rate, op_rows_gather / op_rows_merge, 16-byte rule of op_gemm_tn_grouped's C; 6: the op_probe_* entry points left for libonepeace_probe.so (include/onepeace_probe.h), op_gemm_nt_grouped answers OP_ENOTSUP for the GeGLU epilogue, op_gemm_tn_grouped_plan takes the workgroup count and tune word; 7: W / ldw / rowdot of op_gemm_tn_grouped, g0 of op_resid_bwd, op_gamma_grad_finish, op_gemm_nt_batched, op_audio_conv1_ln_gelu_fwd / _bwd; 8: the layer-scale gradient without a division -- rscale of op_gemm_tn_grouped, op_transpose_scaled + the scale member of the op_transpose_batched descriptor, op_resid_bwd leaves gamma out of dbranch when g0 is asked for, op_gamma_grad_finish lost its gamma argument; rowdot is a [N / 128][M] matrix of partial sums written once each (no atomics); 9: row tables -- a residual branch that runs on the samples stochastic depth keeps reads and writes the FULL activation matrix through op_rows_map's table instead of through packed copies: x_rows of op_layernorm_fwd / op_layernorm_bwd, dout_rows of op_resid_bwd, resid_rows / resid_rows_total of op_gemm_nt and op_gemm_nt_grouped, op_rows_merge without `upd` copies the dropped samples' rows only; op_prof_reserve */
const char* op_last_error(void);

/* Live per-kernel-family timing with HIP events recorded on the launch stream (used by bench.py's `roofline`).
 * family 0 = GEMM (work = flops), 1 = attention forward, 2 = attention backward.  HOST pointers. */
int op_prof_enable(int on);
int op_prof_reserve(int events); /* pre-creates events (two per profiled launch; created without the system-scope fence) */
int op_prof_collect(double* ms, int64_t* count, double* work, int n_families);

/* ---- LayerNorm (+ optional fused exact-erf GELU) ---------------------------------------------------------------
 * Replaces one_peace/models/components.py:23-26,47-52 (torch.nn.LayerNorm / flash_attn layer_norm seam) for
 * self_attn_layer_norm, self_attn.ln (sub-LN), final_layer_norm, the FFN LN(F) (transformer_layer.py:154), the
 * per-modality final norms (transformer_encoder.py:201-220), and with act_gelu=1 the LayerNorm->GELU pairs of the
 * stems (adapter/image.py:66-75, adapter/audio.py:293-301).  x,y [rows, cols] contiguous; w,b [cols] or NULL;
 * mean,rstd fp32 [rows] (NULL = not wanted).  cols % 8 == 0, cols <= 8192. */
/* (ABI 9) x_rows: nullable DEVICE int32 [rows] (op_rows_map): row r of the input is row x_rows[r] of a LARGER matrix x; an entry < 0
 * stands for a row of zeros (the surplus rows of a rounded-up segment).  y / mean / rstd are indexed by r. */
int op_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, int64_t rows,
                     int64_t cols, float eps, int act_gelu, int dtype, const int* x_rows, void* stream);
int64_t op_layernorm_bwd_workspace_bytes(int64_t rows, int64_t cols);
/* dx = LN backward (+ `add`: gradient arriving through the residual path, may be NULL, may alias dx);
 * dw, db [cols] optional (need `workspace`); accumulate != 0 adds into dw/db. */
int op_layernorm_bwd(const void* dy, const void* x, const void* w, const void* b, const float* mean, const float* rstd,
                     const void* add, void* dx, void* dw, void* db, void* workspace, int64_t rows, int64_t cols,
                     int act_gelu, int accumulate, int dtype, const int* x_rows, void* stream);
/* (ABI 9) x_rows as in op_layernorm_fwd: x, add AND dx are then rows x_rows[r] of larger matrices (dy, mean, rstd by r); rows with an
 * entry < 0 are not stored.  With dx == add (in place) the rows no entry names keep `add`: the gradient of the skip connection. */

/* ---- bf16 MFMA GEMM  C[M,N] = A[M,K] . W[N,K]^T  with fused epilogues ---------------------------------------------
 * Replaces: q/k/v/out projections (multihead_attention.py:63-66,103-105,124; up to three weight segments of n_seg rows
 * per launch, k_proj has no bias), GeGLU (transformer_layer.py:54-67), Linear(F->H) (:156), fused_dropout_res
 * (transformer_layer.py:70-88), the similarity matmuls of the contrastive head (image_text_pretrain_loss.py:171-172),
 * text/image/audio_proj (one_peace_retrieval.py:110-121) and the hMLP patch convolutions (adapter/image.py:66-75).
 * epilogue 0: C(bf16) = acc + bias
 *          1: C(f32)  = alpha[0] * acc + bias             (alpha: device scalar or NULL)
 *          2: GeGLU:  C(bf16)[M,N] = gelu(A W0^T) * (A W1^T); B0 = wi_0, B1 = wi_1 ([N,K] each); h0/h1 (optional, both
 *             or none) receive the two pre-activations for the backward pass
 *          3: C(bf16) = resid + rowscale[m / rows_per_sample] * gamma[n] * (acc + bias[n]); gamma/rowscale NULL = 1;
 *             resid may alias C; h0 (optional) receives acc + bias.
 * K % 64 == 0, N % 8 == 0, lda/ldb % 8 == 0 (host pads otherwise: one-peace_amd/ops.py gemm_any).
 * workspace (optional fp32 scratch): lets the launch planner split K over several workgroups for bias-free epilogue-0
 * launches with few output tiles and a long K (weight gradients); tile size (128^2 / 256^2) and the split are chosen
 * from a wave-quantisation model. */
int op_gemm_nt(const void* A, int64_t lda, const void* B0, const void* B1, const void* B2, int64_t ldb, int64_t n_seg,
               const void* bias0, const void* bias1, const void* bias2, void* C, int64_t ldc, void* h0, void* h1,
               const void* resid, int64_t ldr, const void* gamma, const float* rowscale, int64_t rows_per_sample,
               const float* alpha, int64_t M, int64_t N, int64_t K, int epilogue, void* workspace, int64_t workspace_bytes,
               int64_t tune, const int* resid_rows, int64_t resid_rows_total, void* stream);
/* (ABI 9) resid_rows: nullable DEVICE int32 [M] (op_rows_map), residual epilogue only: `resid` and `C` are the BASES of matrices of
 * resid_rows_total rows (each below 4 GiB), the residual is read from and the result written to row resid_rows[m]; rows with an entry
 * < 0 are computed and dropped.  h0 (the branch output) and rowscale stay indexed by m.  The packed rows of the samples a residual
 * branch keeps go straight back to their places (transformer_layer.py:78-88 without the products by zero). */
/* C[M,N] (bf16) = A^T B with A [K,M] (lda), B [K,N] (ldb) row-major bf16: the weight-gradient GEMM dW = dy^T x of
 * nn.Linear (autograd of components.py:29-34 users) straight from the activation matrices (transpose-read fragments,
 * no transposed copies).  K % 64 == 0, M/N/lda/ldb % 8 == 0, else returns -95 (use op_transpose + op_gemm_nt).
 * accumulate != 0: C += A^T B.  workspace: optional fp32 scratch enabling split-K. */
int op_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
               int accumulate, void* workspace, int64_t workspace_bytes, int64_t tune, void* stream);
/* Grouped form of op_gemm_nt: up to three problems C_p[M_p,N] = epilogue(A_p[M_p,K] . W_p[N,K]^T) that share N, K, the leading
 * dimensions and the epilogue (0 bias, 2 GeGLU, 3 residual) in ONE launch of the persistent 256x256 kernel -- replaces the three
 * per-modality FFN launches of an encoder layer (transformer_layer.py:203-226: text / image / audio rows go through their own
 * text_ffn / image_ffn / audio_ffn).  Every array argument is a HOST array: A, M, C, h0, h1, resid, gamma, rowscale,
 * rows_per_sample have nprob entries; B and bias have 2 * nprob ([2p] = the weight / its bias, [2p + 1] = wi_1 of a GeGLU problem).
 * h0 / h1 / resid / gamma / rowscale / bias (arrays or entries) may be NULL.  Results are bit-identical to nprob op_gemm_nt calls.
 * Needs N % 256 == 0 (GeGLU: % 128), K % 64 == 0, K >= 128, lda / ldb / ldc % 8 == 0, operands < 2 GiB; else returns -95. */
int op_gemm_nt_grouped(int64_t nprob, const void* const* A, const int64_t* M, int64_t lda, const void* const* B, int64_t ldb,
                       const void* const* bias, void* const* C, int64_t ldc, void* const* h0, void* const* h1,
                       const void* const* resid, int64_t ldr, const void* const* gamma, const float* const* rowscale,
                       const int64_t* rows_per_sample, int64_t N, int64_t K, int epilogue, int64_t tune, const int* const* resid_rows,
                       int64_t resid_rows_total, void* stream); /* resid_rows: nullable HOST array of nprob DEVICE tables, see op_gemm_nt */
/* (ABI 7) `batch` equally shaped products  C_z[M,N] = A_z[M,K] W_z[N,K]^T (+ bias_z[N])  with operands at constant element strides as ONE
 * launch (blockIdx.z = z; 128 x 128 tiles, no split-K): the per-group GEMMs of the audio adapter's grouped positional Conv1d over
 * strided patch views (one_peace/models/adapter/audio.py:57-84; each group alone fills half the chip).  bias nullable. */
int op_gemm_nt_batched(const void* A, int64_t lda, int64_t stride_a, const void* W, int64_t ldb, int64_t stride_b, const void* bias,
                       int64_t stride_bias, void* C, int64_t ldc, int64_t stride_c, int64_t M, int64_t N, int64_t K, int64_t batch,
                       void* stream);
/* (ABI 7) First block of the audio feature extractor, fused and straight from the waveform (one_peace/models/adapter/audio.py:254-311,
 * ConvFeatureExtractionModel block 0: Conv1d(1 -> C, kernel 10, stride `stride`, optional bias) -> LayerNorm over channels -> GELU):
 *   y[r][c] = GELU(LN_C(bf16(sum_j w0[c][j] * wav[stride * r + j] (+ b0[c]))))        rows x C bf16, C <= 512, C % 8 == 0
 * mean / rstd [rows] fp32 are kept for the backward, which RECOMPUTES the row from its ten samples and returns the parameter gradients
 * (dw0 [C, 10], db0 [C], dlnw [C], dlnb [C]; bf16; nullable except dw0; overwritten or accumulated) -- the 2.1 GB convolution output
 * of the headline batch is never written, re-read or kept.  workspace: op_audio_conv1_ln_gelu_bwd_workspace_bytes(C). */
int op_audio_conv1_ln_gelu_fwd(const void* wav, int64_t stride, const void* w0, const void* b0, const void* lnw, const void* lnb, void* y,
                               float* mean, float* rstd, int64_t rows, int64_t C, float eps, void* stream);
int64_t op_audio_conv1_ln_gelu_bwd_workspace_bytes(int64_t C);
int op_audio_conv1_ln_gelu_bwd(const void* dy, const void* wav, int64_t stride, const void* w0, const void* b0, const void* lnw, const void* lnb,
                               const float* mean, const float* rstd, void* dw0, void* db0, void* dlnw, void* dlnb, void* workspace,
                               int64_t rows, int64_t C, int accumulate, void* stream);
/* Grouped form of op_gemm_tn: up to 16 weight-gradient GEMMs  C_i[M_i,N_i] (bf16, ldc_i) (+)= A_i^T B_i  with their own operands,
 * sizes, K_i and outputs as ONE persistent launch WITHOUT split-K: the tile list of all problems is walked by one workgroup per
 * CU (per-XCD queues of WAVES -- as many consecutive tiles of a problem's tile rectangle as the XCD has workgroups, all of one K, so
 * that what an XCD runs at the same time shares its operand panels through the L2; longest K first, work stealing), every output tile runs its whole K
 * and is written / accumulated exactly once -- no fp32 slabs, no fold kernel.  Replaces the weight-gradient launches autograd
 * makes per nn.Linear of an encoder layer (transformer_layer.py:165-228 backward: q|k|v, out_proj, wi_0|wi_1 and wo of every
 * modality FFN); per-tile results are bit-identical to an unsplit op_gemm_tn.  Every array argument is a HOST array of nprob
 * entries.  counters: op_gemm_tn_grouped_counter_bytes() bytes of device memory zeroed ONCE by the caller (the launch re-arms
 * it; one block per stream).  Shape rules per problem as op_gemm_tn (+ ldc % 8 == 0 and C_i 16-byte aligned: the gradient is read-modify-written in 16-byte pieces); returns -95 and launches nothing when a
 * problem does not qualify.  tune: bits 0-9 forced number of workgroups (0 = one per CU); bit 10: round 4's form of the queues
 * (waves of a multiple of six tiles, the odd workgroups of an XCD draw single tiles from the back; A/B timing). */
int64_t op_gemm_tn_grouped_counter_bytes(void);
/* Host-only (no GPU needed): the tile queues op_gemm_tn_grouped builds for these sizes, `workgroups` (0: one per CU; 256 without a
 * device) and `tune` -- records of four int32 (queue 0..7, problem index of the caller, tile row, tile column) queue by queue in
 * draw order; returns the record count (= the number of 256 x 256 output tiles, each exactly once) or -22 when `cap` records do
 * not suffice. */
int64_t op_gemm_tn_grouped_plan(int64_t nprob, const int64_t* M, const int64_t* N, const int64_t* K, int64_t workgroups, int64_t tune,
                                int32_t* out, int64_t cap);
/* (ABI 7) W / ldw / rowdot: nullable HOST arrays; for a problem with rowdot[i] != NULL (requires accumulate[i], M_i and N_i multiples
 * of 256, W_i [M_i, N_i] bf16 16-byte aligned, ldw_i % 8 == 0) the launch also WRITES the partial row dots
 *   rowdot_i[s][m] = sum_{n in [128 s, 128 s + 128)} W_i[m][n] * P_i[m][n],   s < N_i / 128,   rowdot_i: fp32 [N_i / 128][M_i]
 * (ABI 8: every entry stored exactly once -- no atomics, no zeroing, run-to-run identical; ABI 7 added into one [M_i] vector with fp32
 * atomics; P_i = THIS launch's fp32 product A_i^T B_i before it is rounded into C_i).
 * (ABI 8) rscale: nullable HOST array; rscale[i] != NULL (bf16 [M_i], only with rowdot[i]) makes the accumulation
 * C_i[m][:] += rscale_i[m] * P_i[m][:] while rowdot still sums W_i * the UNSCALED P_i.  With A = rowscale * dout, the output gradient
 * of a residual branch WITHOUT the layer scale (op_resid_bwd with g0), B = input of the branch's last Linear, W = that Linear's
 * weight and rscale = gamma:  C += the weight gradient gamma[n] * (A^T B)[n][:], and sum_s rowdot[s][n] = sum_k W[n][k] * (A^T B)[n][k] IS the
 * layer-scale gradient sum_rows rowscale * dout * (x W^T) (+ the bias term of op_gamma_grad_finish) -- for any gamma, including 0 --
 * and the branch output y never has to be written or kept (one_peace/models/transformer/transformer_layer.py:70-88). */
int op_gemm_tn_grouped(int64_t nprob, const void* const* A, const int64_t* lda, const void* const* B, const int64_t* ldb, void* const* C,
                       const int64_t* ldc, const int64_t* M, const int64_t* N, const int64_t* K, const int32_t* accumulate,
                       const void* const* W, const int64_t* ldw, float* const* rowdot, const void* const* rscale, void* counters,
                       int64_t tune, void* stream);
/* `tune` (op_gemm_nt, op_gemm_tn, op_gemm_plan): per-call tuning word, 0 = what production uses.  The library keeps NO tuning
 * state, so every entry point is a pure function of its arguments; tests and tools select a kernel flavour with the call:
 * bits 0-1 tile (0 auto: a cost model picks 128x128 or 256x256 tiles, K-splits and the tail-rows split; 1 force 128x128;
 * 2 force 256x256); bits 2-3 flavour of the 256x256 NT kernel (0 auto, 1 BK = 32, 2 eight-wave full-line, 3 four-wave full-line); bits 4-6 tail-rows split (0 default, 1 off,
 * 3 whenever it saves a round, 4 always); bits 7-11 M-tiles per L2 group (0 auto); bits 12-14 timing ablations of the 256x256
 * kernel (tools; wrong results); bits 15-18 forced K-split count of small problems (tools); bit 19 register-staged operands
 * instead of LDS-DMA (global_load_lds); bits 20-22 kernel of the four-wave NT launches (0 auto, 1 / 3 one-tile workgroups with
 * that instruction schedule, 6 persistent workgroups, 7 the round-2 kernel; op_gemm_nt_grouped: persistent workgroups unless 7). */
/* Host-only query (no GPU needed): the launch decision op_gemm_nt takes for a dense, single-segment problem.
 * plan[0] = tile (128 | 256), plan[1] = K-splits, plan[2] = 1 if the epilogue runs in the split-K fold kernel,
 * plan[3] = leftover rows (M % 256) split off into a second, small launch (0 = none). */
int op_gemm_plan(int64_t M, int64_t N, int64_t K, int epilogue, int has_bias, int64_t workspace_bytes, int64_t tune, int* plan);

/* ---- fp8 (OCP e4m3) variant of the FFN GEMMs: BASELINE configs[4], explicit opt-in (csrc/fp8.hip) -------------------------
 * No reference counterpart (the reference trains in bf16/fp16, trainer.py:86-88); replaces, when the caller opts in, the
 * forward GEMMs of transformer_layer.py:54-67,149-157 and (round 6) the two input-gradient GEMMs of their backward (the weight
 * gradients stay bf16).  op_quant_fp8_rows: any cols % 8 == 0 (rows wider than 8192 columns are read twice).  Per-row quantisation (x ~= q * scale[row], q = e4m3 of x * 448 / amax_row),
 * fp32 accumulation on v_mfma_scale_f32_16x16x128_f8f6f4, dequantisation by scale_a[m] * scale_b[n] in the epilogue. */
int op_quant_fp8_rows(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int64_t rows, int64_t cols, void* stream);
/* Round 5 (ABI 6): op_layernorm_fwd (bf16, no GELU) / op_ln_geglu_fwd that ALSO write their output row-quantised to fp8 e4m3 -- q8
 * [rows, cols] bytes, q8_scale [rows] fp32, bit-identical to op_quant_fp8_rows(y) -- so that the fp8 FFN forward needs no
 * quantisation pass of its own (transformer_layer.py:196-199: the sub-LayerNorm in front of the FFN; :149-157: LayerNorm(F)). */
int op_layernorm_fwd_q8(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, void* q8, float* q8_scale,
                        int64_t rows, int64_t cols, float eps, void* stream);
int op_ln_geglu_fwd_q8(const void* h0, const void* h1, int64_t ldh, const void* w, const void* b, void* y, float* mean, float* rstd, void* q8,
                       float* q8_scale, int64_t rows, int64_t cols, float eps, void* stream);
/* epilogue 0: + bias; 2: GeGLU (B0 = wi_0, B1 = wi_1, scales sb0 / sb1, optional h0 / h1); 3: residual epilogue of op_gemm_nt.
 * A8 [M,K], B8 [N,K] fp8 bytes (row strides lda / ldb in bytes, multiples of 16); K % 128 == 0; C / h0 / h1 / resid bf16. */
int op_gemm_nt_fp8(const void* A8, int64_t lda, const float* sa, const void* B0, const void* B1, int64_t ldb, const float* sb0,
                   const float* sb1, const void* bias, void* C, int64_t ldc, void* h0, void* h1, const void* resid, int64_t ldr,
                   const void* gamma, const float* rowscale, int64_t rows_per_sample, int64_t M, int64_t N, int64_t K, int epilogue,
                   int64_t tune, void* stream);

/* ---- attention ---------------------------------------------------------------------------------------------------
 * Replaces multihead_attention.py:102-115 (bmm QK^T, += attn_mask, fp32 softmax, bmm PV) and the xformers seam
 * :79-101, plus the dense-bias assembly of transformer_encoder.py:144-162: bias is the per-table image
 * [heads][S][Spad] from op_relpos_bias_build, key padding a byte mask [B][Spad] (non-zero = masked key).
 * q,k,v: bf16 rows of `ld` elements, row = b*S + s, head h at columns [h*64, h*64+64) (a packed [B*S, 3H] projection
 * output serves all three).  out [B*S][ldo]; lse fp32 [B][heads][lse_ld] (natural log) or NULL.  head_dim == 64.
 * Spad >= S rounded up to 128 when bias/key_pad/backward are used.
 * bias_batch_stride: 0 = one bias image shared by all samples; otherwise elements between per-sample images
 * [B][heads][S][Spad] (masked pretraining gathers a different token subset per sample, adapter/image.py:229-246); the
 * backward then returns one dbias slab per sample.
 * bias_frag (optional): the same image(s) in the fragment-major layout of op_attn_bias_pack; with it (or without any
 * bias) sequences of up to 320 keys run the resident-K/V kernel, which adds the bias with the matrix pipe; a biased call
 * without bias_frag always runs the streaming kernel. */
int op_attn_fwd(const void* q, const void* k, const void* v, int64_t ld, const void* bias, int64_t bias_batch_stride,
                const void* bias_frag, const void* key_pad, void* out, int64_t ldo, float* lse, int64_t lse_ld, int64_t B,
                int64_t S, int64_t Spad, int64_t heads, int64_t head_dim, float scale, int64_t tune, void* stream);
/* Fragment-major repack of n_img row-major bias images [n_img][S][Spad] (n_img = heads, or B * heads for per-sample
 * images): dst [n_img][ceil(S/16)][ceil(S/32)][64 lanes][8] bf16 = op_attn_bias_frag_elems(n_img, S) elements, zero
 * outside the sequence.  (No reference counterpart: the reference adds a dense [B, heads, S, S] tensor,
 * multihead_attention.py:107-108.) */
int64_t op_attn_bias_frag_elems(int64_t n_img, int64_t S);
int op_attn_bias_pack(const void* src, void* dst, int64_t n_img, int64_t S, int64_t Spad, void* stream);
/* delta[b][h][q] = sum_d dout*out (fp32, row stride Spad) */
int op_attn_bwd_delta(const void* dout, const void* out, int64_t ldo, float* delta, int64_t B, int64_t S, int64_t Spad,
                      int64_t heads, void* stream);
/* autograd of the above (reference: torch autograd through the bmm/softmax ops).  biasT = bias with rows = key; its pad
 * columns [S, Spad) must hold FINITE values (op_relpos_bias_build writes zeros): the dK/dV kernel adds the bias with the matrix
 * pipe, where 0 * NaN would leak into live rows.  lse / delta entries and the q-major image's columns at [S, Spad) stay
 * unspecified.  bias_frag (optional): op_attn_bias_pack of `bias`; with it the merged dQ + dBias kernel adds the bias with
 * the matrix pipe too.  dq/dk/dv rows have stride ldg; dbias fp32 [slabs][heads][S][Spad] (optional, accumulated into:
 * pre-zero it; the gradient is the sum over slabs, slabs = op_attn_bwd_dbias_slabs(B, S, heads, tune)). */
int64_t op_attn_bwd_dbias_slabs(int64_t B, int64_t S, int64_t heads, int64_t tune);
/* out (nullable): the forward output rows (same stride ldo as dout).  Given, `delta` is a WORKSPACE of the call: the dQ kernels
 * compute delta = rowsum(dout o out) per head from fragments they load anyway and the dK/dV kernel, launched behind them, reads
 * it -- op_attn_bwd_delta's pass over both matrices is not needed.  NULL: `delta` must hold op_attn_bwd_delta's result. */
int op_attn_bwd(const void* q, const void* k, const void* v, int64_t ld, const void* dout, const void* out, int64_t ldo,
                const void* bias, const void* biasT, const void* bias_frag, int64_t bias_batch_stride, const void* key_pad,
                const float* lse, float* delta, void* dq,
                void* dk, void* dv, int64_t ldg, float* dbias, int64_t B, int64_t S, int64_t Spad, int64_t heads,
                int64_t head_dim, float scale, int64_t tune, void* stream);

/* ---- relative-position bias tables -------------------------------------------------------------------------------
 * Replaces get_rel_pos_bias of adapter/image.py:164-171, adapter/text.py:76-83, adapter/audio.py:117-124:
 * out[h][i][j] = table[bucket[i][j]][h] (bf16 [heads][S][Spad], columns >= S zero); transposed != 0 -> out[h][j][i]. */
int op_relpos_bias_build(const void* table, const int* bucket, int64_t bucket_ld, void* out, int64_t heads, int64_t S,
                         int64_t Spad, int transposed, void* stream);
/* dtable[bucket[i][j]][h] += dbias[h][i][j]  (fp32; dtable pre-zeroed by the caller) */
int op_relpos_bias_bwd(const float* dbias, const int* bucket, int64_t bucket_ld, float* dtable, int64_t heads, int64_t S,
                       int64_t Spad, void* stream);
/* Per-sample images for the masked-pretraining passes (adapter/image.py:188-204,229-246, adapter/text.py, adapter/audio.py:
 * gather_features selects rows and columns preserve_ids[b] of the dense bias): out[b][h][i][j] = table[bucket[ids[b][i]][ids[b][j]]][h]
 * straight from the table -- out [B][heads][K][Kpad] bf16, pad columns zero; transposed != 0: rows = keys.  ids [B][K] int32
 * position ids (padding already mapped to a valid id, adapter/image.py:241-243; it is masked through key_pad).  The backward
 * folds the per-sample gradient slabs dbias [B][heads][K][Kpad] of op_attn_bwd into dense_ws (fp32 [heads][Sfull][Sfull], zeroed
 * by the caller; Sfull = extent of the bucket table) and scatters that onto dtable [num_rel][heads] (pre-zeroed). */
int op_relpos_bias_build_ids(const void* table, const int* bucket, int64_t bucket_ld, const int* ids, void* out, int64_t B,
                             int64_t heads, int64_t K, int64_t Kpad, int transposed, void* stream);
int op_relpos_bias_bwd_ids(const float* dbias, const int* bucket, int64_t bucket_ld, const int* ids, float* dense_ws, int64_t Sfull,
                           float* dtable, int64_t B, int64_t heads, int64_t K, int64_t Kpad, void* stream);

/* ---- HBM-bound helpers of the layer backward -----------------------------------------------------------------------
 * (the reference gets these from autograd over transformer_layer.py:54-88,149-157) */
int op_transpose(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, void* stream);
/* (ABI 8) out[c][r] = bf16(scale[r] * in[r][c]); scale: bf16 [rows] or NULL (= op_transpose).  The input-gradient copy of the last
 * Linear of a residual branch with the layer scale folded in: dx = (rowscale * dout) . (gamma o W)
 * (one_peace/models/transformer/transformer_layer.py:70-88; the operand op_resid_bwd writes with g0 carries no gamma). */
int op_transpose_scaled(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, const void* scale,
                        void* stream);
/* Many transposes in one launch (the dgrad copies of all weights after an optimiser step).  table: DEVICE array of n
 * descriptors, op_transpose_desc_bytes() bytes each, laid out as { const void* in; void* out; int32 rows, cols;
 * int64 ld_in, ld_out; int32 tile0, tiles_x; const void* scale; } with tile0 = running sum of ceil(cols/64)*ceil(rows/64),
 * tiles_x = ceil(cols/64) and scale as in op_transpose_scaled (NULL = none); total_tiles = that sum over all descriptors. */
int op_transpose_batched(const void* table, int64_t n, int64_t total_tiles, void* stream);
int64_t op_transpose_desc_bytes(void);
int64_t op_colsum_workspace_bytes(int64_t N);
/* Per-segment column sums of x [M, n_seg*seg_cols]: the q/k/v bias gradients of the fused projection
 * (one_peace/models/transformer/multihead_attention.py:57-62; a null out_i skips that segment).  workspace: op_colsum_workspace_bytes(n_seg*seg_cols). */
int op_colsum_segments(const void* x, void* out0, void* out1, void* out2, void* workspace, int64_t M, int64_t n_seg,
                       int64_t seg_cols, int accumulate, void* stream);
/* Backward of out = resid + rowscale[m/rps] * gamma[n] * y[m][n] (layer-scale + drop-path residual,
 * one_peace/models/transformer/transformer_layer.py:70-88,190-196,224-226) in one pass:
 * dbranch = rowscale*gamma*dout; dgamma (+)= sum_m rowscale*dout*y; dbias (+)= sum_m rowscale*gamma*dout.  Nullable: y+dgamma, gamma,
 * rowscale, dbias, g0.  (ABI 7) g0 (fp32 [N], overwritten): sum_m rowscale*dout, i.e. dbias WITHOUT the gamma factor -- what
 * op_gamma_grad_finish multiplies with the last Linear's bias when dgamma is taken from the weight gradient instead of from y.
 * (ABI 8) With g0 != NULL (an alternative to y + dgamma) dbranch = rowscale*dout, WITHOUT gamma: the weight-gradient launch applies
 * gamma to its output rows (op_gemm_tn_grouped: rscale) and the input-gradient GEMM reads a gamma-scaled weight copy
 * (op_transpose_scaled), so that the layer-scale gradient needs no division by gamma. */
int64_t op_resid_bwd_workspace_bytes(int64_t N);
int op_resid_bwd(const void* dout, const void* y, const void* gamma, const float* rowscale, int64_t rows_per_sample,
                 void* dbranch, void* dgamma, void* dbias, float* g0, void* workspace, int64_t M, int64_t N, int accumulate,
                 const int* dout_rows, void* stream); /* (ABI 9) dout_rows: nullable DEVICE int32 [M]: row m of dout is row dout_rows[m] of a larger matrix (< 0: zeros); y / dbranch by m */
/* (ABI 7; 8: no gamma argument, no division, partial slots instead of atomics) Layer-scale gradient of a residual branch
 * out = resid + rowscale * gamma * (x W^T + b)  WITHOUT the branch output:
 *   dgamma[n] (+)= sum_{s < slots} rowdot[s][n] + sum_i b_i[n] * g0_i[n]
 * rowdot: fp32 [slots][N], op_gemm_tn_grouped's partial row dots over the UNSCALED gradient of one or more weight sets laid out slot
 * after slot (sum_s = sum_k W[n][k] * ((rowscale*dout)^T x)[n][k]); folded in slot order: deterministic;
 * (b_i, g0_i): bias of the last Linear and op_resid_bwd's g0 for up to three weight sets that share gamma (the three modality FFNs of
 * a lock-step layer, transformer_layer.py:203-226; b_i NULL = no bias).  Exact for every gamma, 0 included (the reference:
 * transformer_layer.py:78-88).  dgamma: bf16 [N]. */
int op_gamma_grad_finish(const float* rowdot, int64_t slots, const void* b0, const float* g00, const void* b1, const float* g01,
                         const void* b2, const float* g02, void* dgamma, int64_t N, int accumulate, void* stream);
/* Backward of LayerNorm_F(gelu(h0) * h1) w.r.t. h0, h1 and the LayerNorm affine in one pass (the FFN's GeGLU + inner
 * sub-LayerNorm, transformer_layer.py:64-67,111-118); mean/rstd: forward statistics.  workspace: op_layernorm_bwd_workspace_bytes.
 * ldh: row stride of h0 / h1 (0 = cols).  ldd: row stride (elements) of dh0 / dh1, 0 = cols -- the two gradients may be the halves of one [rows, 2 * cols] matrix, which
 * is then ONE operand for the merged wi_0 | wi_1 weight-gradient and input-gradient GEMMs. */
int op_ln_geglu_bwd(const void* dy, const void* h0, const void* h1, const void* w, const float* mean, const float* rstd,
                    void* dh0, void* dh1, int64_t ldd, int64_t ldh, void* dw, void* db, void* workspace, int64_t rows,
                    int64_t cols, int accumulate, void* stream);
/* Forward of the same pair: y [rows, cols] = LayerNorm_F(bf16(gelu(h0) * h1)) with exact-erf GELU, + mean / rstd (nullable).
 * h0 / h1: row stride ldh (0 = cols) -- the halves of the [rows, 2 * cols] output of ONE plain two-segment up-projection GEMM
 * (wi_0 | wi_1).  Used by the training path instead of the GEMM's EPI_GEGLU epilogue: the erf costs the GEMM more than this
 * HBM-bound pass costs in total. */
int op_ln_geglu_fwd(const void* h0, const void* h1, int64_t ldh, const void* w, const void* b, void* y, float* mean, float* rstd,
                    int64_t rows, int64_t cols, float eps, void* stream);
/* out[n] = (accumulate ? out[n] : 0) + mul[n] * sum_m rowscale[m/rps] * x[m][n] * (y ? y[m][n] : 1); y/rowscale/mul NULL ok */
int op_colsum(const void* x, const void* y, const float* rowscale, int64_t rows_per_sample, const void* mul, void* out,
              void* workspace, int64_t M, int64_t N, int accumulate, int out_dtype, void* stream);
/* GeGLU backward: dh0 = dg*h1*gelu'(h0), dh1 = dg*gelu(h0) */
int op_geglu_bwd(const void* dg, const void* h0, const void* h1, void* dh0, void* dh1, int64_t numel, void* stream);
/* dbranch[m][n] = rowscale[m/rps] * gamma[n] * dout[m][n]  (backward of fused_dropout_res wrt the branch) */
int op_scale_rows(const void* dout, const void* gamma, const float* rowscale, int64_t rows_per_sample, void* dbranch,
                  int64_t M, int64_t N, void* stream);

/* ---- contrastive head ----------------------------------------------------------------------------------------------
 * F.normalize(x, dim=1) of one_peace_retrieval.py:112,116,120 / one_peace_pretrain.py:165-173. */
int op_l2norm_fwd(const void* x, void* y, float* inv_norm, int64_t rows, int64_t cols, float eps, int out_dtype, void* stream);
int op_l2norm_bwd(const void* dy, const void* y, const float* inv_norm, void* dx, int64_t rows, int64_t cols, int y_dtype,
                  void* stream);
/* compute_itc_loss / compute_atc_loss + adjust_label_smoothed_nll_loss (image_text_pretrain_loss.py:17-27,164-185;
 * audio_text_pretrain_loss.py:160-181): per row of sim [rows][n] (fp32): fp32 log-softmax, NLL at target0+row with the
 * reference's eps/(n-1) smoothing, argmax hit, <dsim, sim>; sim is overwritten by gscale * dloss_row/dsim. */
int op_infonce_rows(float* sim, int64_t rows, int64_t n, int64_t ld, int64_t target0, float label_smoothing, float gscale,
                    float* row_loss, float* row_hit, float* row_dot, int write_grad, void* stream);

/* ---- optimiser -------------------------------------------------------------------------------------------------------
 * One AdamW update over a flat bf16 parameter range, the rule of one_peace/optim/adam.py:186-253 (what the reference
 * runs without apex): fp32 moments, decoupled decay before the update, eps added to sqrt(v).  g is multiplied by
 * grad_scale inside the kernel (1/world after a SUM all-reduce).  numel % 8 == 0, step >= 1. */
int op_adamw_step(void* p, const void* g, float* m, float* v, int64_t numel, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int64_t step, float grad_scale, const float* grad_sqnorm, float clip_norm,
                  void* stream);
/* The same update over a flat buffer partitioned into n_groups <= 256 contiguous parameter groups, ONE launch: the param groups
 * of trainer.py:265-278 / utils/layer_decay.py:34-77 ("layer_<id>_<decay|no_decay>" with lr_scale and weight_decay; lr_g = lr *
 * lr_scale_g as optim/base_optimizer.py:8-14 sets it).  group_end8[g] (device, ascending, int64) = end of group g in 8-element
 * vectors; group_lr_scale / group_weight_decay: device fp32 tables. */
int op_adamw_step_groups(void* p, const void* g, float* m, float* v, int64_t numel, const int64_t* group_end8,
                         const float* group_lr_scale, const float* group_weight_decay, int64_t n_groups, float lr, float beta1,
                         float beta2, float eps, int64_t step, float grad_scale, const float* grad_sqnorm, float clip_norm,
                         void* stream);
/* out[0] = sum of squares of a bf16 vector in fp32 (the global gradient norm of fairseq/fairseq/utils.py:349-391 over the
 * flat gradient buffer; feeds op_adamw_step's device-side clip coefficient, trainer.py:929).  workspace: 1024 floats. */
int op_sqnorm(const void* x, int64_t numel, float* workspace, float* out, void* stream);

/* ---- stochastic depth without the multiplications by zero ------------------------------------------------------------
 * The reference multiplies the branch output of a dropped sample by 0 (transformer_layer.py:78-88: fused_dropout_res).  These two
 * entries pack the rows of the samples a residual branch KEEPS into a smaller matrix (the branch then runs on that alone) and merge
 * its result back.  Per segment i < nseg (<= 4; a segment = the B samples x S tokens of one modality inside the packed activation
 * matrix), host arrays: src_row0 (first row in the full matrix), dst_row0 (first row in the packed matrix), S, n_kept, dst_rows
 * (>= n_kept * S, rounded up by the caller; the surplus rows are written as ZEROS so that they add nothing to a weight gradient),
 * n_samples, list_off (start of the segment's list inside `list`, a DEVICE int32 array).
 *   op_rows_gather: list = the kept sample numbers, ascending:  dst[dst_row0 + j*S + t] = src[src_row0 + list[j]*S + t].
 *   op_rows_merge:  list = per sample its position among the kept ones or -1:  out[r] = upd[dst_row0 + list[sample]*S + t] for
 *                   kept samples, base[r] otherwise; out may be base (then only the kept rows are written).
 * bf16 rows of `cols` elements (cols % 8 == 0), densely packed. */
int op_rows_gather(const void* src, void* dst, const int* list, int64_t nseg, const int64_t* src_row0, const int64_t* dst_row0,
                   const int64_t* S, const int64_t* n_kept, const int64_t* dst_rows, const int64_t* n_samples, const int64_t* list_off,
                   int64_t dst_total, int64_t cols, void* stream);
int op_rows_merge(const void* base, const void* upd, void* out, const int* list, int64_t nseg, const int64_t* src_row0,
                  const int64_t* dst_row0, const int64_t* S, const int64_t* n_kept, const int64_t* dst_rows, const int64_t* n_samples,
                  const int64_t* list_off, int64_t total, int64_t cols, void* stream);
/* (ABI 9) op_rows_merge with upd == NULL (out != base): only the rows of the DROPPED samples are copied base -> out (the kept rows were
 * written through a row table).  op_rows_map: map[r] (DEVICE int32 [dst_total]) = the row of the full matrix that packed row r stands
 * for -- what op_rows_gather would read -- or -1 for the surplus rows; `list` = the kept lists as for op_rows_gather. */
int op_rows_map(int* map, const int* list, int64_t nseg, const int64_t* src_row0, const int64_t* dst_row0, const int64_t* S,
                const int64_t* n_kept, const int64_t* dst_rows, const int64_t* n_samples, const int64_t* list_off, int64_t dst_total,
                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ONEPEACE_HIP_H */
