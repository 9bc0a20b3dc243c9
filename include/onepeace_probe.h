/* libonepeace_probe.so -- hardware-semantics and power probes for gfx950.  TEST AND MEASUREMENT INFRASTRUCTURE, not part of the
 * product library (libonepeace_hip.so, include/onepeace_hip.h) since ABI version 6: tests/test_probes_gpu.py measures the lane maps
 * the kernels rely on with it, bench.py the MFMA rate the package sustains at its power limit (roofline.power_limited_peak).
 * Same conventions as onepeace_hip.h: plain pointers, caller-owned memory, work enqueued on `stream`, 0 or an error code
 * (message: this library's own op_last_error). */
#ifndef ONEPEACE_PROBE_H
#define ONEPEACE_PROBE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* op_last_error(void);

int op_probe_mfma16(const void* a, const void* b, float* d, int n, void* stream);
int op_probe_mfma32(const void* a, const void* b, float* d, int n, void* stream);
int op_probe_tr16(const void* img, const int* addr, void* out, int n, void* stream);
int op_probe_glds(const void* src, const int* src_off, int lds_base, void* dump, void* stream);
/* raw v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3): a, b = [n][64 lanes][8 dwords], sa, sb = [n][64] E8M0 scale dwords */
int op_probe_mfma_f8(const void* a, const void* b, const void* sa, const void* sb, float* d, int n, void* stream);
/* Register-only MFMA loop (no LDS, no global memory inside): `workgroups` x 4 waves each issue iters x 64 v_mfma_f32_16x16x32_bf16 on
 * the 8 operand fragments of `operands` (8 x 64 lanes x 8 bf16).  out: workgroups x 256 floats; clk: workgroups x 2 uint64 = shader
 * clock ticks and 100 MHz ticks over the loop.  The caller times the launch: flops = workgroups x 4 x iters x 64 x 16384.  bench.py
 * uses it to report the MFMA rate the package sustains at its power limit beside the data-sheet peak. */
int op_probe_mfma_rate(const void* operands, float* out, void* clk, int workgroups, int iters, void* stream);
/* A stand-in for a collective's kernel sharing the GPU with backward: `workgroups` x 256 threads, 96 KiB of LDS each (one per CU), copying their own
 * slab (slab_floats floats of buf per workgroup) for `micros` microseconds.  when: NULL or workgroups x 2 uint64 (start, end in 100 MHz
 * ticks).  tools/cu_contention_ab.py: what CUs held by another kernel cost the persistent / one-tile NT GEMM launches. */
/* Throughput of LDS atomics: `workgroups` x 8 waves, each wave iters x 16 conflict-free wave-level operations on LDS (mode 0 ds_add_f32,
 * 1 ds_add_u32, 2 ds_write_b32).  out: workgroups x 512 floats; clk: workgroups uint64 = shader cycles of the loop; cycles per
 * wave-level instruction of the CU = clk / (iters * 16 * 8).  (Round 6: what decided the fused attention backward.) */
int op_probe_lds_atomic(float* out, void* clk, int workgroups, int iters, int mode, void* stream);
int op_probe_occupy(void* buf, long long slab_floats, int workgroups, long long micros, void* when, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ONEPEACE_PROBE_H */
