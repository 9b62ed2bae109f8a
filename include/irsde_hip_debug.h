/* irsde_hip_debug.h — kernel-level TEST and TUNING hooks of libirsde_hip.so.
 *
 * Not part of the drop-in boundary (include/irsde_hip.h): nothing here replaces a reference interface.  The parity
 * tests use irsde_debug_conv to exercise one convolution code path at a time; tools/ uses irsde_bench_conv for the
 * A/B measurements logged under profiles/.  Neither changes the launch plan of an engine: the `naive` / `variant`
 * selector is scoped to the one call (reset to the production dispatch before returning).
 */
#ifndef IRSDE_HIP_DEBUG_H
#define IRSDE_HIP_DEBUG_H

#include "irsde_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel-level test hook: one implicit-GEMM convolution (csrc/conv_igemm.hip) on NHWC device tensors.
 * in0/in1: [B][Hin][Win][C0|C1] (channel concat, in1 may be NULL); w_oihw/bias: HOST, reference layout;
 * film: device [rows][2*Cout] or NULL; res/out: device [B][Ho][Wo][Cout].  `naive` selects the code path under test:
 *   0 production dispatch      1 VALU cross-check kernel
 *   2 / 3 Winograd F(2x2,3x3) / F(4x4,3x3) (3x3 s1 p1 only); 12 / 13 and 22 / 23: the same with the component GEMMs forced
 *         onto the tile-loop kernel (all / 2 components per block)
 *   46 / 47 the direct implicit GEMM on the PAIR kernels (conv_igemm.hip: fp32 storage, activations split into 16-bit hi + lo pieces while
 *           staged, three cross products on the 16-bit MFMA): fp16 / bf16 pieces
 *   44 / 45 three-launch Winograd F(4x4,3x3) with the engine's pair GEMM (pair-interleaved operands, LDS-DMA): fp16 / bf16 hi + lo pieces
 *   42 / 43 three-launch Winograd F(4x4,3x3) with split-operand component GEMMs on the bf16 MFMA pipe (csrc/gemm_split.hip): 2 / 3 bf16 planes
 *   34 the 64-cout fused Winograd kernel (r03: Cout and C0 + C1 multiples of 64); 36: with its cout-block-by-XCD block mapping forced wherever legal
 *   35 / 37 the same kernel's fp16-pair twin (IRSDE_FLAG_SPLIT_F16X2; all four hi / lo cross products on v_mfma_f32_16x16x32_f16) / with the mapping forced
 *   33 the fused Winograd F(4x4,3x3) kernel (csrc/wino_fused.hip; 3x3 s1 p1, H and W multiples of 4, C0, C1 and Cout multiples of 32)
 *   4 bf16-MFMA mode (halo kernel for eligible 3x3 layers); 160 / 161 its generic 256 / 128 tile
 *   5 fp16-MFMA mode (IRSDE_FLAG_FP16; halo kernel for eligible 3x3 layers); 165 its generic 128 tile
 *   204 / 260 / 261 the same three with bf16 activation storage (inputs / residual are rounded, the result widened back)
 *   162 / 262 / 166 (r05) modes 4 / 204 / 5 with the 512-pixel x 128-channel halo kernel forced (conv3x3_halo2_kernel; layers with >= 128 output channels),
 *   163 / 263 / 167 with the 256-pixel halo kernel forced
 *   100 + v: tile variant v of the fp32 kernel (3 = 256x128, 50 = 256x256, 73 = tile-loop kernel for 1x1 layers)
 * splits > 1 forces split-K.  Synchronises `stream`. */
int irsde_debug_conv(const float* in0, int C0, const float* in1, int C1, int B, int Hin, int Win, int in_shift,
                     const float* w_oihw, int Cout, int KH, int KW, int stride, int pad, const float* bias,
                     const float* film, int film_bstride, int silu, const float* res, float* out, int naive,
                     int splits, void* stream);

/* Kernel-level test hook of csrc/gemm_split.hip: C_z[M][N] = A_z[M][K] . B_z[N][K]^T for z < ncomp (device f32 tensors, z-major),
 * operands split into `nplanes` (2 or 3) bf16 pieces on the device (plane-major prototype kernel; nplanes = 42 / 44: the engine's
 * pair-interleaved two-piece kernel with bf16 / fp16 pieces), products on v_mfma_f32_32x32x16_bf16, f32 accumulate.  K a multiple of 32.  Synchronises `stream`. */
int irsde_debug_split_gemm(const float* A, const float* B, float* C, int M, int N, int K, int ncomp, int nplanes, void* stream);

/* Kernel tuning hook: average ms of one KxK convolution (pad K/2, or 4x4 s2 p1) on random NHWC data.
 * variant: 0 production dispatch, 3 / 50 fp32 256x128 / 256x256 tiles, 5 one block per CU, 6 generic pointer staging instead of
 * buffer descriptors, 7 LDS-transposed instead of direct epilogue, 60 / 61 / 62 bf16 mode (256 tile / 128 tile / automatic
 * incl. the halo kernel), 63 = 62 with bf16 activation storage, 64 / 65 = 62 and 66 / 67 = 63 with the 512- / 256-pixel halo kernel forced (r05), 80 / 81 Winograd F(4x4,3x3) fused kernel / three-launch path (3x3 s1 only), 82 the fused kernel once with its
 * phase timeline printed to stdout, 400 the 64-cout fused Winograd kernel (r03; 401 / 402: its weight fragments / patch loads read zeros
 * without memory traffic, 403: 12 instead of 18 weight units in flight, 404 / 405: its fp16-pair twin with 12 / 18 units in flight, 406 / 407 / 408: 18 weight units in flight with the non-temporal hint on the epilogue traffic / also on the patch loads / nowhere, 410: 12 units + the epilogue hint (= 400, production)), 412 / 413 the three-launch Winograd layer with split-operand GEMMs (2 / 3
 * bf16 planes on the 128 x 128 plane-major prototype kernel), 421 / 422 / 423 the component GEMMs alone: native f32 / 2 planes / 3 planes, 480 / 481 / 482 a direct layer on the PAIR kernels (fp16 / bf16 pieces / without the 256 x 256 tile), 472 the
 * engine's pair-interleaved two-plane GEMM alone (473 / 475 / 476: without its global loads / MFMAs / output stores); epi: 0 none, 1 FiLM+SiLU, 2 SiLU+residual. */
int irsde_bench_conv(int variant, int B, int H, int W, int Cin, int Cout, int K, int stride, int up, int epi, int iters,
                     double* ms_out);
/* Times naf_chain_kernel (csrc/naf_chain.hip) alone on synthetic data: `nblocks` consecutive 512-channel NAFBlocks on B images of 8 x 8 pixels,
 * `iters` launches; variant 1 residual stream in registers + ring of 8 weight fragments (= 0, production), 2 residual stream in L2 + ring of 16,
 * 22 / 24 the kernel on 2 / 4 work-groups per image (r06; 25, PROBES build: 24 + its per-phase cycle stamps; 26, PROBES build: 24 with one group per image missing — the call must fail with the spin timeout).
 * *ms_out = milliseconds per launch.  (tools/naf_chain_bench.py) */
int irsde_bench_naf_chain(int variant, int nblocks, int B, int iters, double* ms_out);
/* Test / measurement hook (process-wide): the number of concurrent sub-batches irsde_sample splits a ConditionalNAFNet batch into — n >= 1 forces it
 * (1 = never split; clipped to 4 and to a divisor of the batch), 0 returns to the heuristic (2 parts from 8 images on (r06; r05: 64), and only where a level
 * runs as a NAFBlock chain).  Plans are cached per split, so changing it never invalidates anything. */
int irsde_debug_force_subbatches(int n);
/* Test / measurement hook (process-wide): work-groups per image of the NAFBlock chain kernel in plans built from now on — 1 the one-group kernel, 2 / 4
 * forced (where 8 ceil(B / 8) g groups fit the compute units next to the call's other sub-batches, else fewer), 0 returns to the rule (as many as fit).
 * Plans already built keep their choice: use a fresh engine (or another batch shape) per setting. */
int irsde_debug_force_chain_groups(int g);

#ifdef __cplusplus
}
#endif
#endif /* IRSDE_HIP_DEBUG_H */
