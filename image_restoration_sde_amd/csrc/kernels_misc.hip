// Memory-bound kernels of the IR-SDE score network and sampler for gfx950 (wave64).
//   channel LayerNorm            module_util.py:70-79
//   LinearAttention core         module_util.py:163-176   (K·V^T and ctx^T·Q on the fp32 MFMA pipe)
//   input prep                   DenoisingUNet_arch.py:90-94 (xt-cond, cat, reflect pad)
//   time embedding / FiLM rows   module_util.py:29-41, DenoisingUNet_arch.py:42-47, module_util.py:127-141
//   reverse-step update + RNG    sde_utils.py:44-48,175-223
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace irsde {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx8 __attribute__((ext_vector_type(8)));

namespace {

// 16-bit operand type of the fused attention kernels' projections: IEEE fp16 (the hi + lo pairs of IRSDE_FLAG_SPLIT_F16X2, IRSDE_FLAG_FP16) or
// bf16 (IRSDE_FLAG_BF16): the same 32x32x16 MFMA shape and fragment layout, fp32 accumulation
template <bool F16OP>
struct Mf16 {
    using x8 = bf16x8;
    using x4 = bf16x4;
    static __device__ __forceinline__ floatx16 mfma(x8 a, x8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct Mf16<true> {
    using x8 = f16x8;
    using x4 = f16x4;
    static __device__ __forceinline__ floatx16 mfma(x8 a, x8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// Activation storage type T (fp32, or bf16 under IRSDE_FLAG_BF16_ACT): arithmetic is fp32 either way.
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)v; }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    const floatx4 f = {v.x, v.y, v.z, v.w};
    *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(f, bf16x4);  // RNE
}

__device__ __forceinline__ float wave_xor_sum(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Channel LayerNorm: y = (x - mean) * (var + eps)^-1/2 * g (+ res), per pixel over C (NHWC: C contiguous).
// L lanes cooperate on one pixel (L = largest power of two <= min(64, C/4)); 64/L pixels per wave.
// Two-pass (mean, then centred variance) in registers, like torch.var/torch.mean.
// ---------------------------------------------------------------------------------------------
constexpr int kLnMaxVec = 8;  // float4 vectors per lane -> C <= 2048

// KV = float4 vectors per lane (1, 2, 4, 8), U = independent pixel groups per iteration: all U x KV loads are issued before the
// first reduction and the U cross-lane reductions interleave — with one pixel group per iteration a wave had a single load
// in flight and the kernel sat at ~2.5 TB/s, latency-bound.
template <typename T, int KV, int U>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ g,
                                                        const T* __restrict__ res, T* __restrict__ out,
                                                        const long long M, const int C, const int L, const float eps,
                                                        const float* __restrict__ fscale = nullptr,
                                                        const float* __restrict__ fshift = nullptr,
                                                        const int film_bstride = 0, const long long ppi = 1) {
    const int lane = threadIdx.x & 63;
    const int ppw = 64 / L;
    const int li = lane % L;
    const int sub = lane / L;
    const int nvec = C >> 2;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const float invC = 1.0f / (float)C;
    for (long long base = wave_id * ppw * U; base < M; base += nwaves * ppw * U) {
        float4 v[U][KV];
        float s[U], q[U];
        long long pix[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pix[u] = base + (long long)u * ppw + sub;
            const T* xp = x + (pix[u] < M ? pix[u] : 0) * C;
            s[u] = 0.f;
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                const int vi = li + k * L;
                const float4 t = ld4(xp + 4 * (vi < nvec ? vi : 0));  // unconditional load from a clamped index, then select
                v[u][k] = vi < nvec ? t : make_float4(0.f, 0.f, 0.f, 0.f);
                s[u] += (v[u][k].x + v[u][k].y) + (v[u][k].z + v[u][k].w);
            }
        }
        for (int o = L >> 1; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < U; ++u) s[u] += __shfl_xor(s[u], o, 64);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float mean = s[u] * invC;
            q[u] = 0.f;
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                const int vi = li + k * L;
                if (vi < nvec) {
                    v[u][k].x -= mean; v[u][k].y -= mean; v[u][k].z -= mean; v[u][k].w -= mean;
                    q[u] += (v[u][k].x * v[u][k].x + v[u][k].y * v[u][k].y) + (v[u][k].z * v[u][k].z + v[u][k].w * v[u][k].w);
                }
            }
        }
        for (int o = L >> 1; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < U; ++u) q[u] += __shfl_xor(q[u], o, 64);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pix[u] >= M) continue;
            const float rstd = 1.0f / sqrtf(q[u] * invC + eps);
            T* op = out + pix[u] * C;
            const float4* gp = reinterpret_cast<const float4*>(g);
            const T* rp = res ? res + pix[u] * C : nullptr;
            const size_t frow = fscale ? (size_t)(film_bstride ? pix[u] / ppi : 0) * film_bstride : 0;
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                const int vi = li + k * L;
                if (vi < nvec) {
                    const float4 gg = gp[vi];
                    float4 o;
                    o.x = v[u][k].x * rstd * gg.x;
                    o.y = v[u][k].y * rstd * gg.y;
                    o.z = v[u][k].z * rstd * gg.z;
                    o.w = v[u][k].w * rstd * gg.w;
                    if (fscale) {  // NAFNet: x * (scale + 1) + shift  (DenoisingNAFNet_arch.py:64,75)
                        const float4 s4 = reinterpret_cast<const float4*>(fscale + frow)[vi];
                        const float4 h4 = reinterpret_cast<const float4*>(fshift + frow)[vi];
                        o.x = o.x * (s4.x + 1.0f) + h4.x; o.y = o.y * (s4.y + 1.0f) + h4.y;
                        o.z = o.z * (s4.z + 1.0f) + h4.z; o.w = o.w * (s4.w + 1.0f) + h4.w;
                    }
                    if (rp) {
                        const float4 r = ld4(rp + 4 * vi);
                        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                    }
                    st4(op + 4 * vi, o);
                }
            }
        }
    }
}

template <typename T>
static void launch_ln_t(const T* x, const float* g, const T* res, T* out, int64_t M, int C, float eps, const float* fscale,
                        const float* fshift, int film_bstride, int64_t ppi, hipStream_t s) {
    if (C % 4 || C > 4 * 64 * kLnMaxVec) throw HipError("layernorm: unsupported channel count " + std::to_string(C));
    int L = 1;
    while (L * 2 <= 64 && L * 2 <= C / 4) L *= 2;
    const int need = (C / 4 + L - 1) / L;
    if (need > kLnMaxVec) throw HipError("layernorm: channel count too large");
    const int ppw = 64 / L;
    const int U = need <= 2 ? 4 : (need <= 4 ? 2 : 1);
    const int64_t waves = (M + (int64_t)ppw * U - 1) / ((int64_t)ppw * U);
    int64_t blocks = (waves + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), blk(256);
#define IRSDE_LN(KV, UU) hipLaunchKernelGGL((layernorm_kernel<T, KV, UU>), grid, blk, 0, s, x, g, res, out, (long long)M, C, L, eps, fscale, fshift, film_bstride, (long long)ppi)
    if (need <= 1) IRSDE_LN(1, 4);
    else if (need <= 2) IRSDE_LN(2, 4);
    else if (need <= 4) IRSDE_LN(4, 2);
    else IRSDE_LN(8, 1);
#undef IRSDE_LN
    IRSDE_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// LinearAttention.  qkv: [B][N][384] = q(4 heads x 32) | k | v.
//   q <- softmax over d (32, per pixel) * 32^-0.5 ; k <- softmax over N (per channel) ; v <- v / N
//   ctx[d][e] = sum_n k[d][n] v[e][n]          (32 x N) x (N x 32)   -> one 32x32 MFMA tile per head
//   out[e][n] = sum_d ctx[d][e] q[d][n]        (N x 32) x (32 x 32)
// Pass 1: per N-chunk, sum_n exp(k-m) and sum_n exp(k-m) v^T on v_mfma_f32_32x32x2_f32 with a running maximum m
//         (A[i=d][k=n] and B[k=n][j=e] are both 128-B coalesced rows straight from HBM; k and v are read once).
// Pass 2: combine chunks (exp(m_chunk - m) factors), fold 1/(S_d N) and the q scale into ctx.
// Pass 3: per 32-pixel tile: softmax(q) in registers (16 d per lane half + one permute), 16 MFMAs.
// ---------------------------------------------------------------------------------------------
constexpr int kHeads = 4;
constexpr int kDh = 32;
constexpr int kHid = kHeads * kDh;  // 128
constexpr int kQkv = 3 * kHid;      // 384

// Pass 1+2 fused: one read of k and v.  Every wave keeps a running maximum of its pixels' k (per channel d) and rescales
// its context accumulator when the maximum grows (online softmax over N); the four waves and later the N-chunks are
// merged with exp(m_part - m_total) factors — the same sums as the two-pass form, without the separate max pass over k.
template <typename T>
__global__ __launch_bounds__(256) void attn_ctx_partial_kernel(const T* __restrict__ qkv, float* __restrict__ pmax,
                                                               float* __restrict__ pctx, float* __restrict__ psum,
                                                               const int N, const int chunk_len, const int nch) {
    __shared__ float red[4][1024 + 64];
    const int ch = blockIdx.x;
    const int bh = blockIdx.y;  // b*4 + head
    const int b = bh >> 2, head = bh & 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;

    const int n0 = ch * chunk_len;
    const int n1 = min(N, n0 + chunk_len);
    const T* base = qkv + (size_t)b * N * kQkv + head * kDh + l31;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float ssum = 0.f;
    float mrun = -INFINITY;  // running max of k[d = l31] over this wave's pixels (same value in both lane halves)
    // wave w takes pixel pairs w, w+4, ...; lane half h takes pixel 2*pair + h
    constexpr int U = 8;  // 16 scalar loads (k, v of 8 pixel pairs) in flight per wave
    for (int pr = wave; 2 * pr < n1 - n0; pr += 4 * U) {
        float kx[U], vv[U];
        float mit = -INFINITY;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = n0 + 2 * (pr + 4 * u) + h;
            const bool ok = n < n1;
            const T* pp = base + (size_t)(ok ? n : n0) * kQkv;
            const float kl = ld1(pp + kHid), vl = ld1(pp + 2 * kHid);  // unconditional loads (clamped address), then select
            kx[u] = ok ? kl : -INFINITY;
            vv[u] = ok ? vl : 0.f;
            mit = fmaxf(mit, kx[u]);
        }
        mit = fmaxf(mit, __shfl_xor(mit, 32, 64));
        if (__any(mit > mrun)) {  // wave-uniform: after the first few iterations the maxima rarely move
            const float mnew = fmaxf(mrun, mit);  // finite: the first pixel pair of every wave-iteration exists
            const float alpha = expf(mrun - mnew);  // 0 on the first iteration (mrun = -inf)
            mrun = mnew;
            // the accumulator tile is ctx[d][e]: row d = (r&3) + 8(r>>2) + 4h needs the factor of channel d, which lives
            // in lane d of either half => fetch it with a lane read
            ssum *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = (r & 3) + 8 * (r >> 2) + 4 * h;
                acc[r] *= __shfl(alpha, d, 64);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float e = expf(kx[u] - mrun);  // exp(-inf) = 0 for pixels past the chunk
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e, vv[u], acc, 0, 0, 0);
            ssum += e;
        }
    }
    ssum += __shfl_xor(ssum, 32, 64);
    // merge the four waves: M = max_w m_w, scale wave w by exp(m_w - M)
    if (h == 0) red[wave][1024 + 32 + l31] = mrun;
    __syncthreads();
    float mblk = fmaxf(fmaxf(red[0][1024 + 32 + l31], red[1][1024 + 32 + l31]),
                       fmaxf(red[2][1024 + 32 + l31], red[3][1024 + 32 + l31]));
    const float wscale = mrun == -INFINITY ? 0.f : expf(mrun - mblk);  // a wave without pixels contributes nothing
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * h;
        red[wave][d * 32 + l31] = acc[r] * __shfl(wscale, d, 64);
    }
    if (h == 0) red[wave][1024 + l31] = ssum * wscale;
    __syncthreads();
    float* oc = pctx + ((size_t)bh * nch + ch) * 1024;
    for (int i = threadIdx.x; i < 1024; i += 256) oc[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    if (threadIdx.x < 32) {
        const int i = 1024 + threadIdx.x;
        psum[((size_t)bh * nch + ch) * 32 + threadIdx.x] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        pmax[((size_t)b * nch + ch) * kHid + head * kDh + threadIdx.x] = mblk;
    }
}

// LayerNorm of a row whose C/4 16-byte pieces sit in C/4 consecutive (aligned) lanes: two-pass mean / centred variance like
// layernorm_kernel (and torch.var / torch.mean), then * g.  C4 = C/4 is a power of two <= 64.
// Sum over an aligned group of W consecutive lanes (W a power of two <= 64), every lane of the group ends with the total.  r05: on the
// vector pipe only — DPP quad permutes / row mirrors inside a 16-lane row, v_permlane16_swap / v_permlane32_swap (gfx950) across rows —
// instead of __shfl_xor's ds_bpermute round trips (one LDS latency per butterfly step: 10 dependent steps per staged 16-byte piece put
// ~4.8k cycles of LDS latency into every 128-pixel tile of the fused attention kernels).  The mirror steps are valid because after the
// quad steps all lanes of a quad hold the same partial sum (and so on upwards); both operands of every add are the same two numbers in
// every lane of the group, so all lanes end with bit-identical totals.
// v_permlane16_swap / v_permlane32_swap (gfx950): rows 1, 3 (lanes 32 - 63) of the first operand change places with rows 0, 2 (lanes 0 - 31) of the second.
// With x in both operands the two results are [A A C C] / [B B D D] (resp. [lo lo] / [hi hi]): their sum (max) is the xor-16 (xor-32) butterfly step.
// Inline assembly on purpose (hipcc, ROCm 7.2; found by the parity tests, r05): through __builtin_amdgcn_permlane*_swap the optimiser folds the case
// "both operands hold the same value" into "both results = result 0" (also when the second operand is an identity DPP copy of the first), and
// `auto r = builtin(...); r[0], r[1]` reads element 0 twice even for different operands — either way the "sum" silently becomes 2 x one half.
// s_nop 1 = the two wait states the swap needs behind a VALU write of its operands (LLVM gfx950 hazard rule); the asm is opaque to the hazard recogniser.
struct SwapPair { float a, b; };
__device__ __forceinline__ SwapPair swap16(const float v) {
    SwapPair r{v, v};
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(r.a), "+v"(r.b));
    return r;
}
__device__ __forceinline__ SwapPair swap32(const float v) {
    SwapPair r{v, v};
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r.a), "+v"(r.b));
    return r;
}
template <int W>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(W >= 1 && W <= 64 && (W & (W - 1)) == 0, "group width must be a power of two <= 64");
    if constexpr (W >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    if constexpr (W >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    if constexpr (W >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    if constexpr (W >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); // row_mirror
    if constexpr (W >= 32) {
        const SwapPair r = swap16(v);
        v = r.a + r.b;
    }
    if constexpr (W >= 64) {
        const SwapPair r = swap32(v);
        v = r.a + r.b;
    }
    return v;
}

// lane l and lane l ^ 32 combined (the two halves of a v_mfma_f32_32x32 accumulator column): v_permlane32_swap, no LDS round trip
__device__ __forceinline__ float half_sum(const float v) {
    const SwapPair r = swap32(v);
    return r.a + r.b;
}
__device__ __forceinline__ float half_max(const float v) {
    const SwapPair r = swap32(v);
    return fmaxf(r.a, r.b);
}

template <int C4>
__device__ __forceinline__ floatx4 ln_piece(floatx4 v, const floatx4 g, const float eps) {
    const float s = group_sum<C4>((v.x + v.y) + (v.z + v.w));
    const float mean = s * (1.0f / (float)(4 * C4));
    v -= mean;
    const float q = group_sum<C4>((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
    // v_rsq_f32 (1 ulp) instead of 1 / sqrtf (v_sqrt + a full-precision division: ~15 vector instructions per piece on the pipe the f32 MFMAs share)
    const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / (float)(4 * C4)) + eps);
    return v * rstd * g;
}

// the same for rows stored as bf16 (IRSDE_FLAG_BF16_ACT): a 16-byte piece holds 8 channels, a row = C8 = C/8 consecutive lanes (a power of two <= 32)
template <int C8>
__device__ __forceinline__ floatx8 ln_piece8(floatx8 v, const float* __restrict__ g8, const float eps) {
    const float s = group_sum<C8>(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
    const float mean = s * (1.0f / (float)(8 * C8));
    v -= mean;
    const float q = group_sum<C8>(((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7])));
    const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / (float)(8 * C8)) + eps);
    const floatx4 ga = *reinterpret_cast<const floatx4*>(g8), gb = *reinterpret_cast<const floatx4*>(g8 + 4);
    const floatx8 g = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
    return v * rstd * g;
}

// k, v projection + context in one kernel (fp32): the k and v thirds of to_qkv never reach HBM.
//   per 128-pixel tile of one image:  [k | v](128 px x 64) = xn(128 x C) . W_{k,v}^T   for each head (wave = head)
//   then straight from the accumulator registers: online softmax of k over the pixels and ctx[d][e] += exp(k - m)[px][d] v[px][e].
// The accumulator layout of v_mfma_f32_32x32x2_f32 (lane = column, registers = rows) is exactly the operand layout of the
// second product (A[i = d][k = pixel], B[k = pixel][j = e]: both indexed by the lane's column and a pixel pair picked by
// the register index and the lane half), so nothing is transposed or staged between the two GEMMs.
// xn tile in LDS ([128][C + 4]: odd number of 16-byte slots per row => conflict-free ds_read_b128), weight fragments
// straight from L2 (a wave's 64 weight rows x 8 k = two 16-byte loads per lane, one K step ahead).
// Output = the (max, sum, context) partials of attn_ctx_partial_kernel, merged by attn_ctx_finalize_kernel.
constexpr int kKvTile = 128;
// PAIR (r03, IRSDE_FLAG_SPLIT_F16X2): the k / v projection on v_mfma_f32_32x32x16_f16 with every operand as an fp16 hi + lo pair (three
// cross products per f32 product: fp32-equivalent, 3/16 of the f32-MFMA cycles): the staged tile is split while it is written to LDS
// (two planes of [128][C] fp16, rows of 2 C + 16 bytes), the weights come as two pre-split planes (wkv_pair: [2][256][C], scaled by
// the power of two that w_inv_scale undoes on the accumulators).  Everything after the projection (softmax, context) is unchanged f32.
// MODE (r05): 0 = f32, 1 = PAIR, 2 / 3 = the 16-bit operand modes (IRSDE_FLAG_BF16 / IRSDE_FLAG_FP16; the reference is fp32 only): ONE bf16 / fp16 plane
// of the staged tile and of the weights (wkv_pair: [256][C], unscaled), one product per f32 product — with the projection at 1/16 of its f32
// cycles the kernel is bound by the tile it streams.  ABF (IRSDE_FLAG_BF16_ACT, MODE 2): xn is a bf16 tensor, pieces of 8 channels.
template <int C, int MODE = 0, bool ABF = false>
__global__ __launch_bounds__(256, 2) void attn_kv_ctx_kernel(const float* __restrict__ xn, const float* __restrict__ wkv,
                                                             float* __restrict__ pmax, float* __restrict__ pctx,
                                                             float* __restrict__ psum, const int N, const int chunk_len,
                                                             const int nch, const float* __restrict__ ln_g, const float ln_eps,
                                                             const unsigned short* __restrict__ wkv_pair = nullptr, const float w_inv_scale = 1.f) {
    extern __shared__ __attribute__((aligned(16))) float kv_smem[];
    constexpr bool PAIR = MODE == 1, P16 = MODE != 0;
    using M16 = Mf16<MODE == 1 || MODE == 3>;
    static_assert(!ABF || MODE == 2, "bf16 activation storage goes with the bf16 operand mode");
    constexpr int RB = 2 * C + 16;   // 16-bit modes: bytes per LDS row (an odd number of 16-byte slots)
    const int ch = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, head = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    constexpr int LDA = C + 4;
    const int n0 = ch * chunk_len;
    const int n1 = min(N, n0 + chunk_len);
    const float* wk = wkv + (size_t)(head * kDh + l31) * C + 4 * h;          // k rows of this head (to_qkv rows 128..255)
    const float* wv = wkv + (size_t)(kHid + head * kDh + l31) * C + 4 * h;   // v rows (256..383)
    constexpr int c4n = C >> 2;
    floatx16 ctx;
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
    float ssum = 0.f, mrun = -INFINITY;
    for (int t0 = n0; t0 < n1; t0 += kKvTile) {
        // stage the tile (rows past the chunk repeat its last pixel; they are masked below)
        auto stage_tile = [&](auto full_tag) {
            // r05: a full tile is one contiguous run of 128 * C floats (NHWC): element 4 i of it is piece i — no per-piece row / column / clamp arithmetic
            constexpr bool FULL = decltype(full_tag)::value;
            const float* tile = xn + ((size_t)b * N + t0) * C;
            // r05: 8 loads in flight before the first LDS write of a pass, passes unrolled (r02: a rolled loop of 4 — four exposed HBM round trips per
            // tile; the 144 accumulator registers it was fitted next to are not live while the tile is staged: they are zeroed behind the barrier)
            constexpr int NP = kKvTile * c4n / 256;
            constexpr int NB = NP % 8 == 0 ? 8 : 4;
#pragma unroll
            for (int j0 = 0; j0 < NP; j0 += NB) {
                floatx4 st[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    int i = tid + 256 * (j0 + j), row, c4;
                    if constexpr (256 % c4n == 0) { row = tid / c4n + (256 / c4n) * (j0 + j); c4 = tid % c4n; }   // (no carry: tid < 256 — constants the compiler can fold into offsets)
                    else { row = i / c4n; c4 = i - row * c4n; }
                    if constexpr (FULL) st[j] = *reinterpret_cast<const floatx4*>(tile + 4 * i);
                    else st[j] = *reinterpret_cast<const floatx4*>(xn + ((size_t)b * N + min(t0 + row, n1 - 1)) * C + 4 * c4);
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    int i = tid + 256 * (j0 + j), row, c4;
                    if constexpr (256 % c4n == 0) { row = tid / c4n + (256 / c4n) * (j0 + j); c4 = tid % c4n; }   // (no carry: tid < 256 — constants the compiler can fold into offsets)
                    else { row = i / c4n; c4 = i - row * c4n; }
                    // ln_g != nullptr: the input is the block's x and PreNorm's LayerNorm (module_util.py:82-90) runs here,
                    // on the staged pieces (power-of-two C only: a row = C/4 consecutive lanes)
                    if constexpr ((c4n & (c4n - 1)) == 0)
                        if (ln_g) st[j] = ln_piece<c4n>(st[j], *reinterpret_cast<const floatx4*>(ln_g + 4 * c4), ln_eps);
                    if constexpr (PAIR) {
                        const f16x4 hi = __builtin_convertvector(st[j], f16x4);
                        const f16x4 lo = __builtin_convertvector(st[j] - __builtin_convertvector(hi, floatx4), f16x4);
                        char* dst = reinterpret_cast<char*>(kv_smem) + row * RB + 8 * c4;
                        *reinterpret_cast<f16x4*>(dst) = hi;
                        *reinterpret_cast<f16x4*>(dst + kKvTile * RB) = lo;
                    } else if constexpr (P16) {   // one plane, RNE
                        *reinterpret_cast<typename M16::x4*>(reinterpret_cast<char*>(kv_smem) + row * RB + 8 * c4) = __builtin_convertvector(st[j], typename M16::x4);
                    } else {
                        *reinterpret_cast<floatx4*>(kv_smem + row * LDA + 4 * c4) = st[j];
                    }
                }
            }
        };
        // ABF: the tile is 128 * C bf16 (one contiguous run), 16-byte pieces of 8 channels, a row = C / 8 consecutive lanes
        auto stage_tile_bf = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            constexpr int c8n = C >> 3;
            const char* xb = reinterpret_cast<const char*>(xn);
            const char* tile = xb + ((size_t)b * N + t0) * C * 2;
            constexpr int NP = kKvTile * c8n / 256;
            constexpr int NB = NP % 8 == 0 ? 8 : 4;
            static_assert(256 % c8n == 0 && NP % NB == 0, "piece / thread mapping");
#pragma unroll
            for (int j0 = 0; j0 < NP; j0 += NB) {
                bf16x8 st[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int i = tid + 256 * (j0 + j), row = tid / c8n + (256 / c8n) * (j0 + j), c8 = tid % c8n;
                    if constexpr (FULL) st[j] = *reinterpret_cast<const bf16x8*>(tile + 16 * (size_t)i);
                    else st[j] = *reinterpret_cast<const bf16x8*>(xb + (((size_t)b * N + min(t0 + row, n1 - 1)) * C + 8 * c8) * 2);
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int row = tid / c8n + (256 / c8n) * (j0 + j), c8 = tid % c8n;
                    if (ln_g) st[j] = __builtin_convertvector(ln_piece8<c8n>(__builtin_convertvector(st[j], floatx8), ln_g + 8 * c8, ln_eps), bf16x8);
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(kv_smem) + row * RB + 16 * c8) = st[j];
                }
            }
        };
        if constexpr (ABF) {
            if (t0 + kKvTile <= n1) stage_tile_bf(std::true_type{}); else stage_tile_bf(std::false_type{});
        } else {
            if (t0 + kKvTile <= n1) stage_tile(std::true_type{}); else stage_tile(std::false_type{});
        }
        __syncthreads();
        floatx16 ak[4], av[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { ak[rt][r] = 0.f; av[rt][r] = 0.f; }
        if constexpr (PAIR) {
            // K steps of 16: lane (row l31, half h) holds k = 16 ks + 8 h .. + 7 of the A rows (pixels) and of its weight row
            const char* arow = reinterpret_cast<const char*>(kv_smem) + l31 * RB + 16 * h;
            const char* wkp = reinterpret_cast<const char*>(wkv_pair) + ((size_t)(head * kDh + l31) * C + 8 * h) * 2;
            const char* wvp = reinterpret_cast<const char*>(wkv_pair) + ((size_t)(kHid + head * kDh + l31) * C + 8 * h) * 2;
            constexpr size_t WPL = (size_t)2 * kHid * C * 2;   // bytes between the hi and the lo weight plane
            constexpr int NK16 = C / 16;
            f16x8 wk2[2][2], wv2[2][2];   // [slot][plane], one K step ahead
            wk2[0][0] = *reinterpret_cast<const f16x8*>(wkp);
            wk2[0][1] = *reinterpret_cast<const f16x8*>(wkp + WPL);
            wv2[0][0] = *reinterpret_cast<const f16x8*>(wvp);
            wv2[0][1] = *reinterpret_cast<const f16x8*>(wvp + WPL);
            auto kstep = [&](const int ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                const int kp = (ks + 1 < NK16 ? ks + 1 : ks) * 32;
                wk2[nxt][0] = *reinterpret_cast<const f16x8*>(wkp + kp);
                wk2[nxt][1] = *reinterpret_cast<const f16x8*>(wkp + WPL + kp);
                wv2[nxt][0] = *reinterpret_cast<const f16x8*>(wvp + kp);
                wv2[nxt][1] = *reinterpret_cast<const f16x8*>(wvp + WPL + kp);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(arow + rt * 32 * RB + 32 * ks);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(arow + kKvTile * RB + rt * 32 * RB + 32 * ks);
                    ak[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wk2[cur][1], ak[rt], 0, 0, 0);
                    av[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wv2[cur][1], av[rt], 0, 0, 0);
                    ak[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wk2[cur][0], ak[rt], 0, 0, 0);
                    av[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wv2[cur][0], av[rt], 0, 0, 0);
                    ak[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wk2[cur][0], ak[rt], 0, 0, 0);
                    av[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wv2[cur][0], av[rt], 0, 0, 0);
                }
            };
            // C >= 128: unrolled by two (the register ping-pong needs an even factor) — the full unroll of 16 (8) steps spilled 236 (72) VGPRs;
            // the 4-step loop of C = 64 stays fully unrolled (measured 9 % slower at 2)
            if constexpr (C > 64) {
#pragma unroll 2
                for (int ks = 0; ks < NK16; ++ks) kstep(ks);
            } else {
#pragma unroll
                for (int ks = 0; ks < NK16; ++ks) kstep(ks);
            }
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { ak[rt][r] *= w_inv_scale; av[rt][r] *= w_inv_scale; }
        } else if constexpr (P16) {
            // one 16-bit plane: K steps of 16, lane (row l31, half h) holds k = 16 ks + 8 h .. + 7 of its pixel rows and of its weight rows
            using x8 = typename M16::x8;
            const char* arow = reinterpret_cast<const char*>(kv_smem) + l31 * RB + 16 * h;
            const char* wkp = reinterpret_cast<const char*>(wkv_pair) + ((size_t)(head * kDh + l31) * C + 8 * h) * 2;
            const char* wvp = reinterpret_cast<const char*>(wkv_pair) + ((size_t)(kHid + head * kDh + l31) * C + 8 * h) * 2;
            constexpr int NK16 = C / 16;
            x8 wk1[2], wv1[2];   // one K step ahead
            wk1[0] = *reinterpret_cast<const x8*>(wkp);
            wv1[0] = *reinterpret_cast<const x8*>(wvp);
#pragma unroll
            for (int ks = 0; ks < NK16; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                const int kp = (ks + 1 < NK16 ? ks + 1 : ks) * 32;
                wk1[nxt] = *reinterpret_cast<const x8*>(wkp + kp);
                wv1[nxt] = *reinterpret_cast<const x8*>(wvp + kp);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const x8 a = *reinterpret_cast<const x8*>(arow + rt * 32 * RB + 32 * ks);
                    ak[rt] = M16::mfma(a, wk1[cur], ak[rt]);
                    av[rt] = M16::mfma(a, wv1[cur], av[rt]);
                }
            }
        } else {
            const float* arow = kv_smem + l31 * LDA + 4 * h;
            // weight fragments (from L2) one K step = 32 MFMAs (2048 cycles) ahead of their use: two register slots, K loop
            // unrolled by two only (a full unroll of up to 32 steps x 32 MFMAs spills)
            constexpr int NK = C / 8;
            static_assert(NK % 2 == 0, "C must be a multiple of 16");
            floatx4 wkr[2], wvr[2];
            wkr[0] = *reinterpret_cast<const floatx4*>(wk);
            wvr[0] = *reinterpret_cast<const floatx4*>(wv);
#pragma unroll 1
            for (int ks0 = 0; ks0 < NK; ks0 += 2)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ks = ks0 + u;
                const int k8 = 8 * ks;
                {
                    const int kp = ks + 1 < NK ? k8 + 8 : k8;  // past the end: a harmless reload
                    wkr[u ^ 1] = *reinterpret_cast<const floatx4*>(wk + kp);
                    wvr[u ^ 1] = *reinterpret_cast<const floatx4*>(wv + kp);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch in front of this step's MFMAs
                const floatx4 bk = wkr[u], bv = wvr[u];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const floatx4 a = *reinterpret_cast<const floatx4*>(arow + rt * 32 * LDA + k8);
                    ak[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bk.x, ak[rt], 0, 0, 0);
                    av[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bv.x, av[rt], 0, 0, 0);
                    ak[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bk.y, ak[rt], 0, 0, 0);
                    av[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bv.y, av[rt], 0, 0, 0);
                    ak[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bk.z, ak[rt], 0, 0, 0);
                    av[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bv.z, av[rt], 0, 0, 0);
                    ak[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bk.w, ak[rt], 0, 0, 0);
                    av[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bv.w, av[rt], 0, 0, 0);
                }
            }
        }
        // accumulator register r of row tile rt: pixel t0 + 32 rt + (r&3) + 8 (r>>2) + 4h, column d (k) / e (v) = l31
        float mit = -INFINITY;
        if (t0 + kKvTile > n1) {   // (uniform) only the last tile of a chunk has pixels to mask: 128 compares + 256 selects per lane otherwise
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = t0 + 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (n >= n1) { ak[rt][r] = -INFINITY; av[rt][r] = 0.f; }
                }
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mit = fmaxf(mit, ak[rt][r]);
        mit = half_max(mit);
        if (__any(mit > mrun)) {  // wave-uniform; the tile's first pixel exists, so mnew is finite
            const float mnew = fmaxf(mrun, mit);
            const float alpha = expf(mrun - mnew);  // 0 on the first tile (mrun = -inf)
            mrun = mnew;
            ssum *= alpha;
            // context row d = (r&3) + 8(r>>2) + 4h needs the factor of channel d, which lives in lane d of either half
#pragma unroll
            for (int r = 0; r < 16; ++r) ctx[r] *= __shfl(alpha, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
        }
        // exp(k - m) for the whole tile FIRST (one fma + v_exp_f32 per element, in place), then the 64 MFMAs back to back:
        // a vector instruction between two f32 MFMAs costs its own time plus ~16 cycles (tools/probe/mfma_valu_samewave.hip),
        // and expf() in front of every MFMA (~17 instructions) made each of them cost ~150 cycles instead of 64.  The common
        // factor 2^(-m log2e) carries one rounding of m log2e; it is the same for every pixel of channel d and cancels
        // against ssum.  exp2(-inf) = 0 for the masked pixels.
        {
            const float ml = mrun * 1.44269504088896341f;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ak[rt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(ak[rt][r], 1.44269504088896341f, -ml));
                    ssum += ak[rt][r];
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE >= 2) {
            // 16-bit operand modes: the context product on the 16-bit MFMA too — exp(k - m) in [0, 1] and v rounded once (RNE), fp32 accumulation.  A K step
            // of 16 pixels takes the lane's registers 8 s .. 8 s + 7 of a row tile for A (its column d) and B (its column e) alike: the same pixel
            // order in both operands, 8 MFMAs of 8 passes instead of 64 of 16.  (ssum stays the fp32 sum of the unrounded exponentials.)
            using x8 = typename M16::x8;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    const floatx8 ka = {ak[rt][8 * sh], ak[rt][8 * sh + 1], ak[rt][8 * sh + 2], ak[rt][8 * sh + 3],
                                        ak[rt][8 * sh + 4], ak[rt][8 * sh + 5], ak[rt][8 * sh + 6], ak[rt][8 * sh + 7]};
                    const floatx8 va = {av[rt][8 * sh], av[rt][8 * sh + 1], av[rt][8 * sh + 2], av[rt][8 * sh + 3],
                                        av[rt][8 * sh + 4], av[rt][8 * sh + 5], av[rt][8 * sh + 6], av[rt][8 * sh + 7]};
                    ctx = M16::mfma(__builtin_convertvector(ka, x8), __builtin_convertvector(va, x8), ctx);
                }
        } else {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) ctx = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[rt][r], av[rt][r], ctx, 0, 0, 0);
        }
        __syncthreads();  // every wave is done with the tile in LDS
    }
    ssum = half_sum(ssum);
    const int bh = b * kHeads + head;
    float* oc = pctx + ((size_t)bh * nch + ch) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) oc[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = ctx[r];
    if (h == 0) {
        psum[((size_t)bh * nch + ch) * 32 + l31] = ssum;
        pmax[((size_t)b * nch + ch) * kHid + head * kDh + l31] = mrun;
    }
}

// Merge of the chunk partials: ctx[d][e] = sum_c pctx_c[d][e] e^{m_c[d] - m[d]} / sum_c psum_c[d] e^{m_c[d] - m[d]} / N * scale.
// grid (B * heads, 4): block = 8 context rows d of one (image, head); 1024 threads = 256 elements x 4 chunk slices (r05: one block of 1024
// threads per (image, head) walked all chunks serially — at the small batches of the strong-scaling shards (B = 2: 256 chunks per image, 8 blocks
// on the chip) that took longer than the k / v kernel it follows).  The slices are combined in fixed order: deterministic.
__global__ __launch_bounds__(1024) void attn_ctx_finalize_kernel(const float* __restrict__ pctx,
                                                                 const float* __restrict__ psum,
                                                                 const float* __restrict__ pmax,
                                                                 float* __restrict__ ctx, const int nch,
                                                                 const float inv_n, const float scale) {
    __shared__ float sm[4][8];
    __shared__ float ss[4][256];
    __shared__ float sz[4][8];
    const int bh = blockIdx.x, dq = blockIdx.y;
    const int b = bh >> 2, head = bh & 3;
    const int el = threadIdx.x & 255, sl = threadIdx.x >> 8;
    const int dl = el >> 5, e = el & 31, d = 8 * dq + dl;
    const float* pm = pmax + (size_t)b * nch * kHid + head * kDh + d;
    float mx = -INFINITY;
    for (int c = sl; c < nch; c += 4) mx = fmaxf(mx, pm[(size_t)c * kHid]);
    if (e == 0) sm[sl][dl] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sm[0][dl], sm[1][dl]), fmaxf(sm[2][dl], sm[3][dl]));
    const float* pc = pctx + (size_t)bh * nch * 1024 + d * 32 + e;
    const float* ps = psum + (size_t)bh * nch * 32 + d;
    float s = 0.f, z = 0.f;
#pragma unroll 4
    for (int c = sl; c < nch; c += 4) {
        const float w = expf(pm[(size_t)c * kHid] - mx);
        s += pc[(size_t)c * 1024] * w;
        z += ps[(size_t)c * 32] * w;
    }
    ss[sl][el] = s;
    if (e == 0) sz[sl][dl] = z;
    __syncthreads();
    if (sl == 0) {
        const float st = (ss[0][el] + ss[1][el]) + (ss[2][el] + ss[3][el]);
        const float zt = (sz[0][dl] + sz[1][dl]) + (sz[2][dl] + sz[3][dl]);
        ctx[(size_t)bh * 1024 + d * 32 + e] = st / zt * inv_n * scale;
    }
}

constexpr int kOutTilesPerBlock = 8;  // 256 pixels per block

template <typename T>
__global__ __launch_bounds__(256) void attn_out_kernel(const T* __restrict__ qkv, const float* __restrict__ ctx,
                                                       T* __restrict__ out, const int N, const int qstride) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, head = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    // B operand: ctx[d = 16h + s][e = l31]
    float cb[16];
    const float* cp = ctx + ((size_t)(b * kHeads + head)) * 1024 + (16 * h) * 32 + l31;
#pragma unroll
    for (int s = 0; s < 16; ++s) cb[s] = cp[s * 32];

    // the q rows of tile t+1 are loaded before the softmax / MFMAs of tile t (one tile of loads always in flight)
    auto load_q = [&](int t, float* dst) {
        const int nb = (blockIdx.x * kOutTilesPerBlock + t) * 32;
        const int nn = nb + l31;
        const T* qp = qkv + ((size_t)b * N + (nn < N ? nn : (nb < N ? nb : 0))) * qstride + head * kDh + 16 * h;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 t4 = ld4(qp + 4 * v);
            dst[4 * v + 0] = t4.x; dst[4 * v + 1] = t4.y; dst[4 * v + 2] = t4.z; dst[4 * v + 3] = t4.w;
        }
    };
    float qn[16];
    load_q(0, qn);
    for (int t = 0; t < kOutTilesPerBlock; ++t) {
        const int nbase = (blockIdx.x * kOutTilesPerBlock + t) * 32;
        if (nbase >= N) break;
        float q[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) q[s] = qn[s];
        if (t + 1 < kOutTilesPerBlock) load_q(t + 1, qn);
        float m = q[0];
#pragma unroll
        for (int s = 1; s < 16; ++s) m = fmaxf(m, q[s]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float z = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            q[s] = expf(q[s] - m);
            z += q[s];
        }
        z += __shfl_xor(z, 32, 64);
        const float iz = 1.0f / z;
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[s] * iz, cb[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int nn = nbase + row;
            if (nn < N) st1(out + ((size_t)b * N + nn) * kHid + head * kDh + l31, acc[r]);
        }
    }
}

// q projection + softmax over d + context product + to_out + bias + LayerNorm + residual in one kernel (fp32, C = 64 / 128):
// neither q nor the attention output nor the to_out result reach HBM — the second half of the fused LinearAttention block
// (module_util.py:163-178: q.softmax(dim=-2) * scale, einsum with the context, to_out = Conv2d + LayerNorm, Residual).
// One block = 128 pixels of one image, 4 waves:
//   A  xn tile -> LDS Xs[128][C+4]
//   B  wave = head: Q^T[d][px] = Wq_h . xn^T   (A operand = weight rows straight from L2, B operand = Xs rows): 4 pixel tiles
//   C  softmax over d in registers: the lane's 16 rows + the other lane half (the 1/sqrt(32) scale is folded into ctx)
//   D  out^T[e][px] = ctx^T . softmax(q): A = 16 preloaded context registers, B = the lane's own q registers (the accumulator
//      layout of B is the operand layout of D: no transposition)
//   E  out tile -> LDS Os[128][128+4] (row = pixel, column = head*32 + e)
//   F  wave = 32-pixel tile: y^T[c][px] = Wout . out^T  (A = to_out rows from L2, B = Os rows), + bias
//   G  LayerNorm over c in registers (C/32 x 16 values + the other lane half), * g, -> LDS (the wave's own rows), then
//      coalesced 16-byte passes: + x (residual), store y.
// PAIR (r03, IRSDE_FLAG_SPLIT_F16X2): the two projections (B: q, F: to_out) on v_mfma_f32_32x32x16_f16 with fp16 hi + lo operand
// pairs (three cross products per f32 product): the xn tile and the attention-output tile are written to LDS as two fp16 planes
// (rows of 2 C + 16 resp. 272 bytes), the weights come as pre-split planes (wq_pair [2][128][C], wout_pair [2][C][128], scaled by powers
// of two that wq_inv / wout_inv undo on the accumulators).  Softmax, the context product, LayerNorm and the residual stay f32.
// MODE 2 / 3 (r05; IRSDE_FLAG_BF16 / IRSDE_FLAG_FP16): one bf16 / fp16 plane per operand (wq_pair [128][C], wout_pair [C][128], unscaled), one product.
// ABF (IRSDE_FLAG_BF16_ACT, MODE 2): xn / xres / y are bf16 tensors (pieces of 8 channels); the arithmetic between them is the same fp32.
template <int C, int MODE = 0, bool ABF = false>
__global__ __launch_bounds__(256, 2) void attn_q_out_fused_kernel(const float* __restrict__ xn, const float* __restrict__ xres,
                                                               const float* __restrict__ wq, const float* __restrict__ ctx,
                                                               const float* __restrict__ wout, const float* __restrict__ bias,
                                                               const float* __restrict__ g2, float* __restrict__ y, const int N,
                                                               const float eps, const float* __restrict__ ln_g,
                                                               const unsigned short* __restrict__ wq_pair = nullptr,
                                                               const unsigned short* __restrict__ wout_pair = nullptr, const float wq_inv = 1.f,
                                                               const float wout_inv = 1.f) {
    constexpr int TP = 128, LDA = C + 4, LDO = kHid + 4, RT = C / 32;
    constexpr bool PAIR = MODE == 1, P16 = MODE != 0;
    using M16 = Mf16<MODE == 1 || MODE == 3>;
    static_assert(!ABF || MODE == 2, "bf16 activation storage goes with the bf16 operand mode");
    constexpr int RBX = 2 * C + 16, RBO = 2 * kHid + 16;   // 16-bit modes: bytes per row of the xn / attention-output planes
    constexpr int LDY = C > kHid ? C + 4 : LDO;  // row stride of the normalised rows (phase G): they must not overlap the next wave's rows
    extern __shared__ __attribute__((aligned(16))) float qo_smem[];
    // one LDS region (67.6 KB -> two blocks per CU, whose phases overlap): the xn tile (phases A, B), then — behind a barrier —
    // the attention output tile (E, F), then, wave-locally, the normalised rows (G)
    float* Xs = qo_smem;  // [128][C+4]
    float* Os = qo_smem;  // [128][132]
    const int b = blockIdx.y, t0 = blockIdx.x * TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const size_t img = (size_t)b * N;

    // ---- A: stage the xn tile (rows past N repeat the last pixel; their results are never stored)
    constexpr int C4 = C / 4;
    const bool full_tile = t0 + TP <= N;   // uniform; a full tile is one contiguous run of 128 * C floats (r05: no per-piece index arithmetic)
    // r05: wave w stages ITS OWN 32-row slab of the tile (pieces i = lane + 64 j of the slab: the same pieces it adds back and stores in phase G), so when
    // the residual IS the staged tensor (xres == xn: the PreNorm-fused plan) the raw pieces stay in registers across the block (C <= 128: 32 / 64
    // registers of the 256 a wave has) and phase G neither re-reads the tile nor waits for it
    constexpr int NQ = 32 * C4 / 64;
    constexpr bool KEEP = !PAIR && !ABF && C <= 128;
    floatx4 xkeep[KEEP ? NQ : 1];
    // ABF: the wave's slab is 32 * C bf16; pieces of 8 channels, a row = C8 consecutive lanes; the raw pieces stay in registers (C / 4 of them) for C <= 128
    constexpr int C8 = C / 8, NQ8 = 32 * C8 / 64;
    constexpr bool KEEP8 = ABF && C <= 128;
    const bool keep_res = (KEEP || KEEP8) && full_tile && xres == xn;
    bf16x8 xkeep8[KEEP8 ? NQ8 : 1];
    auto stage_tile = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const float* slab_in = xn + (img + t0 + wave * 32) * C;
        // 8 loads in flight before the first LDS write of a pass (a rolled loop waits for every load in turn)
#pragma unroll
        for (int j0 = 0; j0 < NQ; j0 += 8) {
            floatx4 st[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = lane + 64 * (j0 + j), row = wave * 32 + lane / C4 + (64 / C4) * (j0 + j), c4 = lane % C4;   // (C4 divides 64)
                if constexpr (FULL) st[j] = *reinterpret_cast<const floatx4*>(slab_in + 4 * i);
                else st[j] = *reinterpret_cast<const floatx4*>(xn + (img + min(t0 + row, N - 1)) * C + 4 * c4);
                if constexpr (FULL && KEEP) xkeep[j0 + j] = st[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = wave * 32 + lane / C4 + (64 / C4) * (j0 + j), c4 = lane % C4;
                // ln_g != nullptr: xn == the block's x and PreNorm's LayerNorm runs here on the staged pieces
                if (ln_g) st[j] = ln_piece<C4>(st[j], *reinterpret_cast<const floatx4*>(ln_g + 4 * c4), eps);
                if constexpr (PAIR) {
                    const f16x4 hi = __builtin_convertvector(st[j], f16x4);
                    const f16x4 lo = __builtin_convertvector(st[j] - __builtin_convertvector(hi, floatx4), f16x4);
                    char* dst = reinterpret_cast<char*>(qo_smem) + row * RBX + 8 * c4;
                    *reinterpret_cast<f16x4*>(dst) = hi;
                    *reinterpret_cast<f16x4*>(dst + TP * RBX) = lo;
                } else if constexpr (P16) {   // one plane, RNE
                    *reinterpret_cast<typename M16::x4*>(reinterpret_cast<char*>(qo_smem) + row * RBX + 8 * c4) = __builtin_convertvector(st[j], typename M16::x4);
                } else {
                    *reinterpret_cast<floatx4*>(Xs + row * LDA + 4 * c4) = st[j];
                }
            }
        }
    };
    auto stage_tile_bf = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const char* xb = reinterpret_cast<const char*>(xn);
        const char* slab_in = xb + (img + t0 + wave * 32) * C * 2;
        constexpr int NB = NQ8 < 8 ? NQ8 : 8;
#pragma unroll
        for (int j0 = 0; j0 < NQ8; j0 += NB) {
            bf16x8 st[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int i = lane + 64 * (j0 + j), row = wave * 32 + lane / C8 + (64 / C8) * (j0 + j), c8 = lane % C8;   // (C8 divides 64)
                if constexpr (FULL) st[j] = *reinterpret_cast<const bf16x8*>(slab_in + 16 * (size_t)i);
                else st[j] = *reinterpret_cast<const bf16x8*>(xb + ((img + min(t0 + row, N - 1)) * C + 8 * c8) * 2);
                if constexpr (FULL && KEEP8) xkeep8[j0 + j] = st[j];
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int row = wave * 32 + lane / C8 + (64 / C8) * (j0 + j), c8 = lane % C8;
                if (ln_g) st[j] = __builtin_convertvector(ln_piece8<C8>(__builtin_convertvector(st[j], floatx8), ln_g + 8 * c8, eps), bf16x8);
                *reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(qo_smem) + row * RBX + 16 * c8) = st[j];
            }
        }
    };
    if constexpr (ABF) {
        if (full_tile) stage_tile_bf(std::true_type{}); else stage_tile_bf(std::false_type{});
    } else {
        if (full_tile) stage_tile(std::true_type{}); else stage_tile(std::false_type{});
    }
    // context operand of phase D: ctx[d = (s&3) + 8(s>>2) + 4h][e = l31] of head = wave
    float cb[16];
    {
        const float* cp = ctx + ((size_t)(b * kHeads + wave)) * 1024 + l31;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) cb[s2] = cp[((s2 & 3) + 8 * (s2 >> 2) + 4 * h) * 32];
    }
    // fp16 operands (MODE 3): the context carries v / N (module_util.py:168) — O(1e-6) on a 256 x 256 map, in fp16's subnormal range — so it and with it the
    // attention output go through the 16-bit MFMAs scaled by ctx_up = 2^ceil(log2 N) (exact), undone on the to_out accumulators in front of the bias
    float ctx_up = 1.f;
    if constexpr (MODE == 3) {
        ctx_up = exp2f(ceilf(log2f((float)N)));
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) cb[s2] *= ctx_up;
    }
    __syncthreads();

    // ---- B: Q^T of head = wave
    floatx16 q[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) q[ct][r] = 0.f;
    if constexpr (PAIR) {
        const char* wrow = reinterpret_cast<const char*>(wq_pair) + ((size_t)(wave * kDh + l31) * C + 8 * h) * 2;
        constexpr size_t WPL = (size_t)kHid * C * 2;   // bytes between the hi and the lo plane of Wq
        const char* brow = reinterpret_cast<const char*>(qo_smem) + l31 * RBX + 16 * h;
        constexpr int NK16 = C / 16;
        f16x8 wa2[2][2];
        wa2[0][0] = *reinterpret_cast<const f16x8*>(wrow);
        wa2[0][1] = *reinterpret_cast<const f16x8*>(wrow + WPL);
#pragma unroll
        for (int ks = 0; ks < NK16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            const int kp = (ks + 1 < NK16 ? ks + 1 : ks) * 32;
            wa2[nxt][0] = *reinterpret_cast<const f16x8*>(wrow + kp);
            wa2[nxt][1] = *reinterpret_cast<const f16x8*>(wrow + WPL + kp);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(brow + ct * 32 * RBX + 32 * ks);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(brow + TP * RBX + ct * 32 * RBX + 32 * ks);
                q[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa2[cur][0], bl, q[ct], 0, 0, 0);
                q[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa2[cur][1], bh, q[ct], 0, 0, 0);
                q[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa2[cur][0], bh, q[ct], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) q[ct][r] *= wq_inv;
    } else if constexpr (P16) {
        using x8 = typename M16::x8;
        const char* wrow = reinterpret_cast<const char*>(wq_pair) + ((size_t)(wave * kDh + l31) * C + 8 * h) * 2;
        const char* brow = reinterpret_cast<const char*>(qo_smem) + l31 * RBX + 16 * h;
        constexpr int NK16 = C / 16;
        x8 wa1[2];
        wa1[0] = *reinterpret_cast<const x8*>(wrow);
#pragma unroll
        for (int ks = 0; ks < NK16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            wa1[nxt] = *reinterpret_cast<const x8*>(wrow + (ks + 1 < NK16 ? ks + 1 : ks) * 32);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
                q[ct] = M16::mfma(wa1[cur], *reinterpret_cast<const x8*>(brow + ct * 32 * RBX + 32 * ks), q[ct]);
        }
    } else
    {
        const float* wrow = wq + (size_t)(wave * kDh + l31) * C + 4 * h;
        const float* brow = Xs + l31 * LDA + 4 * h;
        // weight fragments two K steps (32 MFMAs) ahead of their use: they come from L2
        constexpr int NK = C / 8;
        floatx4 wa[3];
        wa[0] = *reinterpret_cast<const floatx4*>(wrow);
        wa[1] = *reinterpret_cast<const floatx4*>(wrow + 8);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            if (ks + 2 < NK) wa[(ks + 2) % 3] = *reinterpret_cast<const floatx4*>(wrow + 8 * (ks + 2));
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch in front of this step's MFMAs (the scheduler sinks it to its use)
            const floatx4 a = wa[ks % 3];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const floatx4 bb = *reinterpret_cast<const floatx4*>(brow + ct * 32 * LDA + 8 * ks);
                q[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bb.x, q[ct], 0, 0, 0);
                q[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bb.y, q[ct], 0, 0, 0);
                q[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bb.z, q[ct], 0, 0, 0);
                q[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bb.w, q[ct], 0, 0, 0);
            }
        }
    }
    __syncthreads();  // every wave is done reading Xs: the region becomes Os
    // ---- C + D + E per pixel tile: q[ct][r] = Q[d = (r&3) + 8(r>>2) + 4h][pixel 32 ct + l31]
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        float m = q[ct][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, q[ct][r]);
        m = half_max(m);
        float z = 0.f;
        const float ml = m * 1.44269504088896341f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            q[ct][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(q[ct][r], 1.44269504088896341f, -ml));  // (the factor of m's rounding cancels in / z)
            z += q[ct][r];
        }
        z = half_sum(z);
        const float iz = __builtin_amdgcn_rcpf(z);   // v_rcp_f32 (1 ulp): softmax normalisation
        floatx16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        if constexpr (MODE >= 2) {
            // 16-bit operand modes: the context operand (rounded once per block) and the softmax probabilities (in [0, 1]) on the 16-bit MFMA, fp32
            // accumulation.  K step sh = the d values of the lane's registers 8 sh .. 8 sh + 7, the same order in cb[] and q[]: 2 MFMAs instead of 16.
            using x8 = typename M16::x8;
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                // (MODE 3: cb carries the power of two `ctx_up`, see below)
                const floatx8 ca = {cb[8 * sh], cb[8 * sh + 1], cb[8 * sh + 2], cb[8 * sh + 3], cb[8 * sh + 4], cb[8 * sh + 5], cb[8 * sh + 6], cb[8 * sh + 7]};
                const floatx8 qa = {q[ct][8 * sh], q[ct][8 * sh + 1], q[ct][8 * sh + 2], q[ct][8 * sh + 3],
                                    q[ct][8 * sh + 4], q[ct][8 * sh + 5], q[ct][8 * sh + 6], q[ct][8 * sh + 7]};
                o = M16::mfma(__builtin_convertvector(ca, x8), __builtin_convertvector(qa * iz, x8), o);
            }
        } else {
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) o = __builtin_amdgcn_mfma_f32_32x32x2f32(cb[s2], q[ct][s2] * iz, o, 0, 0, 0);
        }
        // o[r] = out[pixel 32 ct + l31][e = (r&3) + 8(r>>2) + 4h]: four 16-byte groups per lane
        if constexpr (PAIR) {
            char* orow = reinterpret_cast<char*>(qo_smem) + (ct * 32 + l31) * RBO + (wave * kDh + 4 * h) * 2;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const floatx4 v = floatx4{o[4 * gq], o[4 * gq + 1], o[4 * gq + 2], o[4 * gq + 3]};
                const f16x4 hi = __builtin_convertvector(v, f16x4);
                const f16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, floatx4), f16x4);
                *reinterpret_cast<f16x4*>(orow + 16 * gq) = hi;
                *reinterpret_cast<f16x4*>(orow + TP * RBO + 16 * gq) = lo;
            }
        } else if constexpr (P16) {
            char* orow = reinterpret_cast<char*>(qo_smem) + (ct * 32 + l31) * RBO + (wave * kDh + 4 * h) * 2;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<typename M16::x4*>(orow + 16 * gq) =
                    __builtin_convertvector(floatx4{o[4 * gq], o[4 * gq + 1], o[4 * gq + 2], o[4 * gq + 3]}, typename M16::x4);
        } else {
            float* orow = Os + (ct * 32 + l31) * LDO + wave * kDh + 4 * h;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<floatx4*>(orow + 8 * gq) = floatx4{o[4 * gq], o[4 * gq + 1], o[4 * gq + 2], o[4 * gq + 3]};
        }
    }
    __syncthreads();  // Os complete

    // ---- F: y^T[c][px] of the 32-pixel tile = wave
    floatx16 yv[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[rt][r] = 0.f;
    if constexpr (PAIR) {
        const char* wrow = reinterpret_cast<const char*>(wout_pair) + ((size_t)l31 * kHid + 8 * h) * 2;
        constexpr size_t WPL = (size_t)C * kHid * 2;   // bytes between the hi and the lo plane of Wout
        const char* brow = reinterpret_cast<const char*>(qo_smem) + (wave * 32 + l31) * RBO + 16 * h;
        constexpr int NK16 = kHid / 16;
        // r06: C = 256 (RT = 8 row tiles) walks its row tiles in two halves: 128 accumulators + a double-buffered fragment pair per row tile of ALL eight
        // was 256 + registers (177 spilled to scratch); per accumulator the products arrive in the same order (bit-identical)
        constexpr int RH = RT > 4 ? RT / 4 : RT;
#pragma unroll
        for (int r0 = 0; r0 < RT; r0 += RH) {
            f16x8 wa2[2][2][RH];
#pragma unroll
            for (int rt = 0; rt < RH; ++rt) {
                wa2[0][0][rt] = *reinterpret_cast<const f16x8*>(wrow + (size_t)(r0 + rt) * 32 * kHid * 2);
                wa2[0][1][rt] = *reinterpret_cast<const f16x8*>(wrow + WPL + (size_t)(r0 + rt) * 32 * kHid * 2);
            }
#pragma unroll
            for (int ks = 0; ks < NK16; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                const int kp = (ks + 1 < NK16 ? ks + 1 : ks) * 32;
#pragma unroll
                for (int rt = 0; rt < RH; ++rt) {
                    wa2[nxt][0][rt] = *reinterpret_cast<const f16x8*>(wrow + (size_t)(r0 + rt) * 32 * kHid * 2 + kp);
                    wa2[nxt][1][rt] = *reinterpret_cast<const f16x8*>(wrow + WPL + (size_t)(r0 + rt) * 32 * kHid * 2 + kp);
                }
                const f16x8 bh = *reinterpret_cast<const f16x8*>(brow + 32 * ks);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(brow + TP * RBO + 32 * ks);
#pragma unroll
                for (int rt = 0; rt < RH; ++rt) yv[r0 + rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa2[cur][0][rt], bl, yv[r0 + rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RH; ++rt) yv[r0 + rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa2[cur][1][rt], bh, yv[r0 + rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RH; ++rt) yv[r0 + rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa2[cur][0][rt], bh, yv[r0 + rt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[rt][r] *= wout_inv;
    } else if constexpr (P16) {
        using x8 = typename M16::x8;
        const char* wrow = reinterpret_cast<const char*>(wout_pair) + ((size_t)l31 * kHid + 8 * h) * 2;
        const char* brow = reinterpret_cast<const char*>(qo_smem) + (wave * 32 + l31) * RBO + 16 * h;
        constexpr int NK16 = kHid / 16;
        x8 wa1[2][RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) wa1[0][rt] = *reinterpret_cast<const x8*>(wrow + (size_t)rt * 32 * kHid * 2);
#pragma unroll
        for (int ks = 0; ks < NK16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            const int kp = (ks + 1 < NK16 ? ks + 1 : ks) * 32;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) wa1[nxt][rt] = *reinterpret_cast<const x8*>(wrow + (size_t)rt * 32 * kHid * 2 + kp);
            const x8 bb = *reinterpret_cast<const x8*>(brow + 32 * ks);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) yv[rt] = M16::mfma(wa1[cur][rt], bb, yv[rt]);
        }
        if constexpr (MODE == 3) {
            const float dn = 1.0f / ctx_up;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[rt][r] *= dn;
        }
    } else
    {
        const float* wrow = wout + (size_t)l31 * kHid + 4 * h;
        const float* brow = Os + (wave * 32 + l31) * LDO + 4 * h;
        // to_out weight fragments (RT per K step, from L2) PF K steps ahead of their use (C = 256: one step = 32 MFMAs already,
        // and eight fragments per step leave no registers for a deeper ring)
        constexpr int NK = kHid / 8;
        constexpr int PF = RT > 4 ? 1 : 2, RING = PF + 1;
        floatx4 wa[RING][RT];
#pragma unroll
        for (int pf = 0; pf < PF; ++pf)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) wa[pf][rt] = *reinterpret_cast<const floatx4*>(wrow + (size_t)rt * 32 * kHid + 8 * pf);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            if (ks + PF < NK) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    wa[(ks + PF) % RING][rt] = *reinterpret_cast<const floatx4*>(wrow + (size_t)rt * 32 * kHid + 8 * (ks + PF));
            }
            __builtin_amdgcn_sched_barrier(0);
            const floatx4 bb = *reinterpret_cast<const floatx4*>(brow + 8 * ks);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const floatx4 a = wa[ks % RING][rt];
                yv[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bb.x, yv[rt], 0, 0, 0);
                yv[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bb.y, yv[rt], 0, 0, 0);
                yv[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bb.z, yv[rt], 0, 0, 0);
                yv[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bb.w, yv[rt], 0, 0, 0);
            }
        }
    }
    // ---- G: bias, LayerNorm over the C channels of pixel 32 wave + l31 (this lane holds c = 32 rt + (r&3) + 8(r>>2) + 4h)
    float sum = 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            yv[rt][r] += bias[32 * rt + (r & 3) + 8 * (r >> 2) + 4 * h];
            sum += yv[rt][r];
        }
    sum = half_sum(sum);
    const float mean = sum * (1.0f / (float)C);
    float sq = 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            yv[rt][r] -= mean;
            sq += yv[rt][r] * yv[rt][r];
        }
    sq = half_sum(sq);
    const float rstd = __builtin_amdgcn_rsqf(sq * (1.0f / (float)C) + eps);
    // Ys = this wave's own 32 rows of the region.  C <= 128: row stride LDO, exactly the Os rows only this wave read in phase F.
    // C = 256: the rows are wider than an Os row, so they overlap other waves' Os rows -> wait until every wave has left phase F.
    // PAIR: the fp16 planes put other waves' Os rows inside this wave's Ys rows for every C -> always wait.
    if constexpr (C > kHid || P16) __syncthreads();
    float* yrow = Os + (wave * 32 + l31) * LDY + 4 * h;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const floatx4 g4 = *reinterpret_cast<const floatx4*>(g2 + 32 * rt + 8 * gq + 4 * h);
            *reinterpret_cast<floatx4*>(yrow + 32 * rt + 8 * gq) =
                floatx4{yv[rt][4 * gq] * rstd * g4.x, yv[rt][4 * gq + 1] * rstd * g4.y, yv[rt][4 * gq + 2] * rstd * g4.z,
                        yv[rt][4 * gq + 3] * rstd * g4.w};
        }
    // the wave re-reads only its own rows: LDS operations of one wave complete in order, no block barrier needed
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    auto residual_store = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const size_t slab = (img + t0 + wave * 32) * C;   // the wave's 32 rows are contiguous: element 4 i of the slab is piece i
        floatx4 xr[NQ];
        if (FULL && KEEP && keep_res) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) xr[j] = xkeep[KEEP ? j : 0];   // the pieces staged in phase A
        } else {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {  // residual loads first, all in flight
                const int i = lane + 64 * j, row = lane / C4 + (64 / C4) * j, c4 = lane % C4;   // (C4 divides 64)
                if constexpr (FULL) xr[j] = *reinterpret_cast<const floatx4*>(xres + slab + 4 * i);
                else xr[j] = *reinterpret_cast<const floatx4*>(xres + (img + min(t0 + wave * 32 + row, N - 1)) * C + 4 * c4);
            }
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int i = lane + 64 * j, row = lane / C4 + (64 / C4) * j, c4 = lane % C4;   // (C4 divides 64)
            const int n = t0 + wave * 32 + row;
            const floatx4 v = *reinterpret_cast<const floatx4*>(Os + (wave * 32 + row) * LDY + 4 * c4);
            if constexpr (FULL) *reinterpret_cast<floatx4*>(y + slab + 4 * i) = v + xr[j];
            else if (n < N) *reinterpret_cast<floatx4*>(y + (img + n) * C + 4 * c4) = v + xr[j];
        }
    };
    // ABF: residual and output are bf16 rows; the sum is formed in fp32 and rounded once (RNE)
    auto residual_store_bf = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const char* rb = reinterpret_cast<const char*>(xres);
        char* yb = reinterpret_cast<char*>(y);
        const size_t slab = (img + t0 + wave * 32) * C * 2;   // byte offset of the wave's 32 contiguous rows
        bf16x8 xr[NQ8];
        if (FULL && KEEP8 && keep_res) {
#pragma unroll
            for (int j = 0; j < NQ8; ++j) xr[j] = xkeep8[KEEP8 ? j : 0];
        } else {
#pragma unroll
            for (int j = 0; j < NQ8; ++j) {
                const int i = lane + 64 * j, row = lane / C8 + (64 / C8) * j, c8 = lane % C8;
                if constexpr (FULL) xr[j] = *reinterpret_cast<const bf16x8*>(rb + slab + 16 * (size_t)i);
                else xr[j] = *reinterpret_cast<const bf16x8*>(rb + ((img + min(t0 + wave * 32 + row, N - 1)) * C + 8 * c8) * 2);
            }
        }
#pragma unroll
        for (int j = 0; j < NQ8; ++j) {
            const int i = lane + 64 * j, row = lane / C8 + (64 / C8) * j, c8 = lane % C8;
            const int n = t0 + wave * 32 + row;
            const float* yr = Os + (wave * 32 + row) * LDY + 8 * c8;
            const floatx4 va = *reinterpret_cast<const floatx4*>(yr), vb = *reinterpret_cast<const floatx4*>(yr + 4);
            const floatx8 v = floatx8{va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w} + __builtin_convertvector(xr[j], floatx8);
            const bf16x8 o = __builtin_convertvector(v, bf16x8);
            if constexpr (FULL) *reinterpret_cast<bf16x8*>(yb + slab + 16 * (size_t)i) = o;
            else if (n < N) *reinterpret_cast<bf16x8*>(yb + ((img + n) * C + 8 * c8) * 2) = o;
        }
    };
    if constexpr (ABF) {
        if (full_tile) residual_store_bf(std::true_type{}); else residual_store_bf(std::false_type{});
    } else {
        if (full_tile) residual_store(std::true_type{}); else residual_store(std::false_type{});
    }
}

// ---------------------------------------------------------------------------------------------
// Full softmax attention (denoising-sde bottleneck) — module_util.py:182-204.  qkv: [B][N][384], out: [B][N][128].
// One wave = one 32-query tile of one (batch, head); flash-style loop over 32-key tiles on the fp32 MFMA pipe:
//   S^T = K Q^T   (A = K rows, B = Q^T; 16 x v_mfma_f32_32x32x2_f32)   -> lane (q = lane&31, half h) holds 16 of the 32 scores
//   online softmax over the keys of this query: 16 registers + one cross-half shuffle
//   O^T += V^T P^T (A = V rows straight from HBM, B = the lane's own P registers — the swapped product needs no transposition)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void full_attn_kernel(const float* __restrict__ qkv, float* __restrict__ out, const int N,
                                                        const float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int bh = blockIdx.y, b = bh >> 2, head = bh & 3;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= N) return;  // wave-uniform
    const float* base = qkv + (size_t)b * N * kQkv + head * kDh;
    const int q = q0 + l31;
    float qreg[16];
    {
        const float4* qp = reinterpret_cast<const float4*>(base + (size_t)(q < N ? q : q0) * kQkv + 16 * h);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 t4 = qp[v];
            qreg[4 * v + 0] = t4.x * scale; qreg[4 * v + 1] = t4.y * scale;
            qreg[4 * v + 2] = t4.z * scale; qreg[4 * v + 3] = t4.w * scale;
        }
    }
    float m = -INFINITY, l = 0.f;
    floatx16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    for (int j0 = 0; j0 < N; j0 += 32) {
        const int j = j0 + l31;
        float kreg[16];
        {
            const float4* kp = reinterpret_cast<const float4*>(base + (size_t)(j < N ? j : 0) * kQkv + kHid + 16 * h);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 t4 = kp[v];
                kreg[4 * v + 0] = t4.x; kreg[4 * v + 1] = t4.y; kreg[4 * v + 2] = t4.z; kreg[4 * v + 3] = t4.w;
            }
        }
        float vreg[16];  // V[j0 + jj_h(s)][d = l31], jj_h(s) = (s&3) + 8(s>>2) + 4h  (the C-layout row of register s)
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const int jj = j0 + (s2 & 3) + 8 * (s2 >> 2) + 4 * h;
            vreg[s2] = jj < N ? base[(size_t)jj * kQkv + 2 * kHid + l31] : 0.f;
        }
        floatx16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[s2], qreg[s2], st, 0, 0, 0);
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (jj >= N) st[r] = -INFINITY;
            tmax = fmaxf(tmax, st[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(m, tmax);
        const float alpha = expf(m - mnew);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = expf(st[r] - mnew);
            psum += st[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        l = l * alpha + psum;
        m = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[s2], st[s2], o, 0, 0, 0);
    }
    if (q < N) {
        const float il = 1.0f / l;
        float* op = out + ((size_t)b * N + q) * kHid + head * kDh;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)  // registers 4g..4g+3 are d = 8g + 4h + {0..3}
            *reinterpret_cast<float4*>(op + 8 * g4 + 4 * h) =
                make_float4(o[4 * g4] * il, o[4 * g4 + 1] * il, o[4 * g4 + 2] * il, o[4 * g4 + 3] * il);
    }
}

// ---------------------------------------------------------------------------------------------
// Input prep: x0[b][y][x][0..P) over the physically zero-bordered (3 px) image, NCHW -> NHWC.
// ---------------------------------------------------------------------------------------------
__global__ void prep_input_kernel(const float* __restrict__ xt, const float* __restrict__ cond, float* __restrict__ x0,
                                  const int B, const int in_nc, const int P, const int H, const int W, const int Hp,
                                  const int Wp, const int reflect) {
    const int Hb = Hp + 6, Wb = Wp + 6;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * Hb * Wb;
    if (idx >= total) return;
    const int xb = (int)(idx % Wb);
    const size_t t1 = idx / Wb;
    const int yb = (int)(t1 % Hb);
    const int b = (int)(t1 / Hb);
    float v[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // in_nc <= 8
    const int y = yb - 3, x = xb - 3;
    // UNet: F.pad(..., 'reflect') to a multiple of 2^depth; NAFNet: zero pad (DenoisingNAFNet_arch.py:189-194)
    if (y >= 0 && y < Hp && x >= 0 && x < Wp && (reflect || (y < H && x < W))) {
        const int sy = y < H ? y : 2 * (H - 1) - y;  // F.pad(..., 'reflect') on the right/bottom
        const int sx = x < W ? x : 2 * (W - 1) - x;
        for (int c = 0; c < in_nc; ++c) {
            const size_t o = (((size_t)b * in_nc + c) * H + sy) * W + sx;
            if (cond) {
                const float cv = cond[o];
                v[c] = xt[o] - cv;
                v[in_nc + c] = cv;
            } else {
                v[c] = xt[o];  // denoising-sde variant: no condition input
            }
        }
    }
    float* op = x0 + idx * P;
    for (int c = 0; c < P; ++c) op[c] = v[c];
}

// ---------------------------------------------------------------------------------------------
// Time embedding path (rows = timesteps of the table, or batch items for tensor-valued t).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

__global__ void sinusoid_kernel(const float* __restrict__ tvals, const float* __restrict__ freqs,
                                float* __restrict__ out, const int rows, const int half) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * half) return;
    const int r = idx / half, i = idx % half;
    const float a = tvals[r] * freqs[i];
    out[(size_t)r * 2 * half + i] = sinf(a);
    out[(size_t)r * 2 * half + half + i] = cosf(a);
}

// one wave per output feature; lanes stride the input dimension (coalesced weight row)
__global__ __launch_bounds__(256) void row_linear_kernel(const float* __restrict__ in, const int in_stride,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         float* __restrict__ out, const int out_stride, const int rows,
                                                         const int in_dim, const int out_dim, const int act_in,
                                                         const int act_out) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= out_dim) return;
    const float* w = W + (size_t)o * in_dim;
    const float bo = bias ? bias[o] : 0.f;
    for (int r = 0; r < rows; ++r) {
        const float* x = in + (size_t)r * in_stride;
        float s = 0.f;
        for (int i = lane; i < in_dim; i += 64) s = fmaf(act_apply(x[i], act_in), w[i], s);
        s = wave_xor_sum(s, 64);
        if (lane == 0) out[(size_t)r * out_stride + o] = act_apply(s + bo, act_out);
    }
}

// One block (the step counter is read and advanced here: more blocks would race on it), 1024 threads, 16-byte pieces, 8 loads in flight per thread:
// the NAFNet FiLM row is 67 072 floats, and the 256-thread scalar loop this replaces took 104 us per step (profiles/r04_final_latent_bench_kernel_trace_stats.txt)
__global__ __launch_bounds__(1024) void step_begin_kernel(StepState* st, const float* __restrict__ film_table, const int film_row,
                                                          float* __restrict__ film_cur, const float* __restrict__ coef_table) {
    const int t = st->t_next;
    __syncthreads();
    const float* src = film_table + (size_t)t * film_row;
    if ((film_row & 3) == 0 && ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(film_cur)) & 15) == 0) {
        const int n4 = film_row >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(film_cur);
        for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * (int)blockDim.x) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * (int)blockDim.x;
                v[j] = s4[i < n4 ? i : i0];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * (int)blockDim.x;
                if (i < n4) d4[i] = v[j];
            }
        }
    } else {
        for (int i = threadIdx.x; i < film_row; i += blockDim.x) film_cur[i] = src[i];
    }
    if (threadIdx.x < 12) st->coef[threadIdx.x] = coef_table[(size_t)t * 12 + threadIdx.x];
    if (threadIdx.x == 0) {
        st->t = t;
        st->t_next = t - 1;
    }
}

__global__ void set_step_kernel(StepState* st, const int t_next) {
    st->t = t_next;
    st->t_next = t_next;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) + Box-Muller.  counter = (quad, t, image, 0x1D5DE), key = seed.
// Keyed by the GLOBAL image index so results do not depend on how the batch is sharded over GPUs.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void philox_normal4(uint32_t quad, uint32_t t, uint32_t image, uint64_t seed, float z[4]) {
    uint32_t r[4];
    philox4x32_10(quad, t, image, 0x1D5DEu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float s = 1.0f / 16777216.0f;
    const float u0 = ((float)(r[0] >> 8) + 0.5f) * s, u1 = ((float)(r[1] >> 8) + 0.5f) * s;
    const float u2 = ((float)(r[2] >> 8) + 0.5f) * s, u3 = ((float)(r[3] >> 8) + 0.5f) * s;
    const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    float sa, ca, sb, cb;
    sincosf(6.28318530717958647692f * u1, &sa, &ca);
    sincosf(6.28318530717958647692f * u3, &sb, &cb);
    z[0] = ra * ca; z[1] = ra * sa; z[2] = rb * cb; z[3] = rb * sb;
}

// coef row: [0] theta_t [1] sigma_t [2] sigma_bar_t [3] dt [4] sqrt(dt) [5] x0_gain=e^{Theta_t dt}
//           [6] posterior term1 [7] term2 [8] posterior std [9..11] unused
__global__ __launch_bounds__(256) void sde_update_kernel(const UpdateParams p) {
    const int b = blockIdx.y;
    const int CHW = p.C * p.H * p.W;
    const int HW = p.H * p.W;
    const int quad = blockIdx.x * blockDim.x + threadIdx.x;
    if (quad * 4 >= CHW) return;
    const int t = p.st ? p.st->t : p.t_imm;
    const float* cf = p.st ? p.st->coef : p.coef_imm;
    const float theta = cf[0], sigma = cf[1], sbar = cf[2], dt = cf[3];
    const float sqdt = cf[4], gain = cf[5], t1 = cf[6], t2 = cf[7];
    const float pstd = cf[8];
    int mode = p.mode;
    const float* noise = p.noise;
    long long noise_tstride = p.noise_tstride;
    unsigned long long seed = p.seed, image_offset = p.image_offset;
    if (p.ctl) {
        mode = p.ctl->mode; noise = p.ctl->noise; noise_tstride = p.ctl->noise_tstride;
        seed = p.ctl->seed; image_offset = p.ctl->image_offset;
    }
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    const bool need_z = mode != 1 && mode != 4;
    const int bg = b + p.batch0;   // index of this image in the call's batch (sub-batch plans: batch0 > 0)
    if (need_z && !noise) philox_normal4((uint32_t)quad, (uint32_t)t, (uint32_t)(image_offset + bg), seed, z);
    for (int k = 0; k < 4; ++k) {
        const int e = quad * 4 + k;
        if (e >= CHW) break;
        const size_t si = (size_t)b * CHW + e;
        const int c = e / HW, rem = e - c * HW, y = rem / p.W, xx = rem - y * p.W;
        const float eps_hat = p.pred[(size_t)b * p.sb + (size_t)c * p.sc + (size_t)y * p.sy + (size_t)xx * p.sx];
        const float x = p.x[si], mu = p.mu[si];
        if (need_z && noise) z[k] = noise[(size_t)t * noise_tstride + (size_t)bg * CHW + e];
        float xn;
        if (mode >= 3) {
            // DenoisingSDE (sde_utils.py:448-457, 44-48): mode 3 reverse_sde, mode 4 reverse_ode; cf[9] = exp(-2 Theta_t dt)
            const float score = -eps_hat / sbar;
            if (mode == 3) {
                const float drift = -0.5f * (sigma * sigma) * (1.0f + cf[9]) * score * dt;
                const float disp = sigma * (z[k] * sqdt);
                xn = x - drift - disp;
            } else {
                const float drift = -0.5f * (sigma * sigma) * cf[9] * score * dt;
                xn = x - drift;
            }
        } else if (mode == 2) {
            // sde_utils.py:237-239, 197-205, 219-223
            const float x0 = (x - mu - sbar * eps_hat) * gain + mu;
            const float mean = t1 * (x - mu) + t2 * (x0 - mu) + mu;
            xn = mean + pstd * z[k];
        } else {
            const float score = -eps_hat / sbar;  // sde_utils.py:184-185
            if (mode == 0) {
                const float drift = (theta * (mu - x) - sigma * sigma * score) * dt;  // :175-176
                const float disp = sigma * (z[k] * sqdt);                               // :181-182
                xn = x - drift - disp;                                                   // :44-45
            } else {
                const float drift = (theta * (mu - x) - 0.5f * (sigma * sigma) * score) * dt;  // :178-179
                xn = x - drift;                                                                  // :47-48
            }
        }
        p.x[si] = xn;
    }
}

__global__ void set_ctl_kernel(SampleCtl* ctl, const int mode, const float* noise, const long long noise_tstride,
                               const unsigned long long seed, const unsigned long long image_offset) {
    ctl->mode = mode; ctl->pad = 0; ctl->noise = noise; ctl->noise_tstride = noise_tstride;
    ctl->seed = seed; ctl->image_offset = image_offset;
}

__global__ void unpack_pred_kernel(const float* __restrict__ pred, float* __restrict__ out, const int B, const int C,
                                   const int H, const int W, const int Hp, const int Wp, const int stride) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * C * H * W;
    if (idx >= total) return;
    const int x = (int)(idx % W);
    size_t r = idx / W;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    out[idx] = pred[(((size_t)b * Hp + y) * Wp + x) * stride + c];
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, const int B, const int C,
                                    const int H, const int W) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * C * H * W;
    if (idx >= total) return;
    const int x = (int)(idx % W);
    size_t r = idx / W;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    out[idx] = ld1(in + (((size_t)b * H + y) * W + x) * C + c);
}

// ---------------------------------------------------------------------------------------------
// NAFNet (Refusion) — DenoisingNAFNet_arch.py:15-82
// depthwise 3x3 (+bias) -> SimpleGate, with per-tile channel sums for the SCA global average pool.
// Tile = kDwRows image rows x (pixel lanes x kDwRun) columns of one image; thread = (pixel lane, 4-channel group): the
// 2 x 9 depthwise weights of its channels live in registers, and the lane walks kDwRun columns to the right keeping a
// (kDwRows + 2) x 3 register window per gate half: one new window column = 6 rows x 2 halves of 16-byte loads yields
// kDwRows = 4 outputs, so every input row passes through the CU's L1 1.5 times (a one-row walk re-read it 3 times, and the
// kernel ran at the ~13 B/clk a CU can pull from L2, not at HBM speed).  Consecutive lanes = consecutive channel groups:
// each load instruction covers whole 128-byte lines.
// ---------------------------------------------------------------------------------------------
constexpr int kDwRows = 4;
constexpr int kDwRun = 16;

struct DwGeom {
    int gpp, PP, run, tiles_x, tiles_y;
};
static inline DwGeom dw_geom(int H, int W, int c) {
    DwGeom g;
    const int G = c >> 2;
    g.gpp = G < 256 ? G : 256;
    g.PP = 256 / g.gpp;
    // columns a lane walks: kDwRun, but no more than it takes the block's pixel lanes to cover the image width (r04: on the latent feature maps,
    // 64 .. 16 columns wide, a run of 16 left three of four lanes without work); at least 4, so that the two window columns a walk starts with stay < 50 % extra
    g.run = (W + g.PP - 1) / g.PP;
    g.run = g.run < 4 ? 4 : g.run > kDwRun ? kDwRun : g.run;
    g.tiles_x = (W + g.PP * g.run - 1) / (g.PP * g.run);
    g.tiles_y = (H + kDwRows - 1) / kDwRows;
    return g;
}

__global__ __launch_bounds__(256) void dwconv_gate_kernel(const float* __restrict__ u, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          float* __restrict__ partial, const int H, const int W,
                                                          const int c, const int tiles_x, const int ntiles, const int run) {
    __shared__ float4 red[256];
    constexpr int R = kDwRows;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int HW = H * W;
    const int G = c >> 2;                        // float4 groups of gated channels
    const int gpp = G < 256 ? G : 256;           // groups handled per pass
    const int PP = 256 / gpp;                    // pixel lanes per pass
    const int C2 = 2 * c;
    for (int gc = 0; gc < G; gc += gpp) {
        const int g = gc + (int)(threadIdx.x % gpp);
        const int pl = threadIdx.x / gpp;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        const int y0 = ty * R, x0 = (tx * PP + pl) * run;
        if (g < G && pl < PP && x0 < W) {
            const int ch = g * 4;
            const float4 b1 = *reinterpret_cast<const float4*>(bias + ch);
            const float4 b2 = *reinterpret_cast<const float4*>(bias + c + ch);
            float4 w1[9], w2[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                w1[k] = *reinterpret_cast<const float4*>(w + k * C2 + ch);
                w2[k] = *reinterpret_cast<const float4*>(w + k * C2 + c + ch);
            }
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* ub = u + (size_t)b * HW * C2 + ch;
            auto ld = [&](int iy, int ix, int half) -> float4 {
                if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return zero;
                return *reinterpret_cast<const float4*>(ub + ((size_t)iy * W + ix) * C2 + half * c);
            };
            float4 win1[R + 2][3], win2[R + 2][3];  // [row y0 - 1 + r][column slot] of the two gate halves
            // one step: the new column x + 1 goes to slot CN (the oldest), the outputs of column x use slots (CL, CM, CN)
            auto step = [&](const int x, auto cl, auto cm, auto cn) {
                constexpr int CL = decltype(cl)::value, CM = decltype(cm)::value, CN = decltype(cn)::value;
#pragma unroll
                for (int r = 0; r < R + 2; ++r) {
                    win1[r][CN] = ld(y0 - 1 + r, x + 1, 0);
                    win2[r][CN] = ld(y0 - 1 + r, x + 1, 1);
                }
#pragma unroll
                for (int ro = 0; ro < R; ++ro) {
                    const int y = y0 + ro;
                    if (y >= H) break;
                    float4 a1 = b1, a2 = b2;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int slot[3] = {CL, CM, CN};
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float4 v1 = win1[ro + ky][slot[kx]], v2 = win2[ro + ky][slot[kx]];
                            const float4 q1 = w1[ky * 3 + kx], q2 = w2[ky * 3 + kx];
                            a1.x = fmaf(v1.x, q1.x, a1.x); a1.y = fmaf(v1.y, q1.y, a1.y);
                            a1.z = fmaf(v1.z, q1.z, a1.z); a1.w = fmaf(v1.w, q1.w, a1.w);
                            a2.x = fmaf(v2.x, q2.x, a2.x); a2.y = fmaf(v2.y, q2.y, a2.y);
                            a2.z = fmaf(v2.z, q2.z, a2.z); a2.w = fmaf(v2.w, q2.w, a2.w);
                        }
                    }
                    const float4 o = make_float4(a1.x * a2.x, a1.y * a2.y, a1.z * a2.z, a1.w * a2.w);
                    *reinterpret_cast<float4*>(out + ((size_t)b * HW + (size_t)y * W + x) * c + ch) = o;
                    sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
                }
            };
#pragma unroll
            for (int r = 0; r < R + 2; ++r) {  // columns x0 - 1 and x0 into slots 0 and 1: the first step loads slot 2
                win1[r][0] = ld(y0 - 1 + r, x0 - 1, 0); win2[r][0] = ld(y0 - 1 + r, x0 - 1, 1);
                win1[r][1] = ld(y0 - 1 + r, x0, 0);     win2[r][1] = ld(y0 - 1 + r, x0, 1);
            }
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            const int xe = x0 + run < W ? x0 + run : W;
            for (int x = x0; x < xe; x += 3) {  // three steps per iteration: the slot roles rotate at compile time
                step(x, I0{}, I1{}, I2{});
                if (x + 1 < xe) step(x + 1, I1{}, I2{}, I0{});
                if (x + 2 < xe) step(x + 2, I2{}, I0{}, I1{});
            }
        }
        red[threadIdx.x] = sum;
        __syncthreads();
        if (pl == 0 && g < G) {  // fixed-order reduction over the pixel lanes: deterministic
            float4 t = red[threadIdx.x];
            for (int q = 1; q < PP; ++q) {
                const float4 r = red[threadIdx.x + q * gpp];
                t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
            }
            *reinterpret_cast<float4*>(partial + ((size_t)b * ntiles + tile) * c + g * 4) = t;
        }
        __syncthreads();
    }
}

// mean[b][k] = sum_tiles partial / HW  (second stage of the deterministic global average pool)
__global__ __launch_bounds__(256) void sca_mean_kernel(const float* __restrict__ partial, const int ntiles,
                                                       float* __restrict__ mean, const int c, const float inv_hw) {
    __shared__ float red[256];
    const int b = blockIdx.y;
    const int k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int tl = threadIdx.x >> 6;
    float t = 0.f;
    if (k < c)
        for (int q = tl; q < ntiles; q += 4) t += partial[((size_t)b * ntiles + q) * c + k];
    red[threadIdx.x] = t;
    __syncthreads();
    if (tl == 0 && k < c) mean[(size_t)b * c + k] = ((red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192])) * inv_hw;
}

// s[b][o] = bias[o] + sum_k W[o][k] * mean[b][k]   (AdaptiveAvgPool2d(1) -> 1x1 conv, DenoisingNAFNet_arch.py:29-33)
// r06: both steps in one launch for the small feature maps (the latent network's levels: every launch there costs ~5 us whatever it does, profiles/r06_notes.md 3b).
// Every block sums the image's tile partials itself — the same four strided partial sums and the same order as sca_mean_kernel: the same bits — and
// computes four output channels.
__global__ __launch_bounds__(256) void sca_fused_kernel(const float* __restrict__ partial, const int ntiles, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ s_out, const int c, const float inv_hw) {
    extern __shared__ float sca_mean_lds[];
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < c; k += 256) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int q0 = 0; q0 < ntiles; q0 += 4)
#pragma unroll
            for (int tl = 0; tl < 4; ++tl)
                if (q0 + tl < ntiles) t[tl] += partial[((size_t)b * ntiles + q0 + tl) * c + k];
        sca_mean_lds[k] = ((t[0] + t[1]) + (t[2] + t[3])) * inv_hw;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= c) return;
    const float* wr = W + (size_t)o * c;
    float t = 0.f;
    for (int k = lane; k < c; k += 64) t = fmaf(wr[k], sca_mean_lds[k], t);
    t = wave_xor_sum(t, 64);
    if (lane == 0) s_out[(size_t)b * c + o] = t + bias[o];
}
__global__ __launch_bounds__(256) void sca_kernel(const float* __restrict__ mean, const float* __restrict__ W,
                                                  const float* __restrict__ bias, float* __restrict__ s_out, const int c) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= c) return;
    const float* wr = W + (size_t)o * c;
    const float* mb = mean + (size_t)b * c;
    float t = 0.f;
    for (int k = lane; k < c; k += 64) t = fmaf(wr[k], mb[k], t);
    t = wave_xor_sum(t, 64);
    if (lane == 0) s_out[(size_t)b * c + o] = t + bias[o];
}

// ---------------------------------------------------------------------------------------------
// r06: NAFBlock norm1 + conv1 / norm2 + conv4 as ONE launch on the small-channel levels of the fp16 operand mode (c = 64 / 128 / 256: the latent network outside
// the chain): LayerNorm + time FiLM (DenoisingNAFNet_arch.py:63-64,74-75) of a 64-pixel tile straight into the fp16 A operand in LDS, the whole K = c in one
// piece (no K loop, every load of the tile in flight at once), v_mfma_f32_32x32x16_f16 against a 64-column weight tile, then bias (conv1) or bias + SimpleGate
// (+ lens FiLM) (conv4).  Why: at these sizes a launch costs ~5 us whatever it does and the two-kernel form pays three or four serialised memory latencies
// (profiles/r06_notes.md 3b).  The LayerNorm arithmetic is layernorm_kernel's (KV = 1: L = c / 4 lanes per pixel, one float4 each, the same summation order,
// this file is compiled without FMA contraction) and the rounding to fp16 is the consuming convolution's: the same operand bits as the two-kernel path.
// ---------------------------------------------------------------------------------------------
struct NafLnConvArgs {
    const float* x;          // [M][c] fp32
    const float* g;          // LayerNorm gain [c]
    const float* fscale;     // FiLM rows (row of image b at + b * film_bstride)
    const float* fshift;
    int film_bstride;
    long long ppi;           // pixels per image
    const unsigned short* w; // fp16 [Cout][c]
    const float* bias;       // [Cout]
    float* out;              // [M][Cout] or (gate) [M][Cout / 2]
    const float* gate_film;  // gate only: per-image [scale (Cout / 2) | shift (Cout / 2)] or nullptr
    int gate_film_bstride;
    int gate;
    int M, Cout;
    // PRO 1 / 2 (conv3 / conv5: no LayerNorm): the A operand is x itself, PRO 1 times the per-image channel scale in_scale[b][c] (SCA, :68) before the rounding;
    // epilogue out = res + (acc + bias) * ch_scale (beta / gamma, :70,81)
    const float* in_scale;
    const float* ch_scale;
    const float* res;
};

// PRO: 0 LayerNorm + FiLM, 1 per-image channel scale, 2 plain
template <int C, int PRO = 0>
__global__ __launch_bounds__(256, 2) void naf_lnconv_kernel(const NafLnConvArgs a) {
    constexpr int L = C / 4;                 // lanes per pixel (16 / 32 / 64)
    constexpr int RPP = 256 / L;             // pixels per pass of the block
    constexpr int NPASS = 64 / RPP;          // passes (4 / 8 / 16): one float4 per lane and pass
    constexpr int ROWB = C * 2 + 16;         // LDS row stride (bytes): K halves + 16-byte pad (conflict-free 16-byte fragment reads)
    extern __shared__ __attribute__((aligned(16))) char lnc_lds[];
    char* As = lnc_lds;                      // [64 pixels][ROWB]
    char* Bs = lnc_lds + 64 * ROWB;          // [64 output channels][ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk_n = a.Cout / 64;
    // XCD-aware remap: the column tiles of a pixel tile run back to back on one XCD (they share the pixel tile in L2)
    int wgid;
    {
        const int orig = blockIdx.x, nwg = gridDim.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int mblk = wgid / nblk_n, nblk = wgid - mblk * nblk_n;
    const int m0 = mblk * 64, n0 = nblk * 64;
    // ---- every load of the tile first: the pixel rows, the weight tile, the LayerNorm / FiLM vectors ----
    const int li = tid % L, sub = tid / L;
    float4 v[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int m = m0 + ps * RPP + sub;
        v[ps] = *reinterpret_cast<const float4*>(a.x + (size_t)(m < a.M ? m : a.M - 1) * C + 4 * li);
    }
    constexpr int BCH = (64 * C * 2 / 16) / 256;   // 16-byte chunks of the weight tile per thread (2 / 4 / 8)
    floatx4 wv[BCH];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int idx = tid + 256 * i, row = idx / (C / 8), ch = idx % (C / 8);
        wv[i] = *reinterpret_cast<const floatx4*>(reinterpret_cast<const char*>(a.w) + ((size_t)(n0 + row) * C) * 2 + ch * 16);
    }
    float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (PRO == 0) gg = reinterpret_cast<const float4*>(a.g)[li];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int idx = tid + 256 * i, row = idx / (C / 8), ch = idx % (C / 8);
        *reinterpret_cast<floatx4*>(Bs + row * ROWB + ch * 16) = wv[i];
    }
    // ---- LayerNorm + FiLM of the 64 pixels (layernorm_kernel, KV = 1), rounded to fp16 into the A tile ----
    const float invC = 1.0f / (float)C;
    if constexpr (PRO != 0) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            floatx4 f = {v[ps].x, v[ps].y, v[ps].z, v[ps].w};
            if constexpr (PRO == 1) {   // (conv_igemm's INSCALE staging: fp32 product, then the rounding)
                const int m = m0 + ps * RPP + sub;
                const float4 sc = *reinterpret_cast<const float4*>(a.in_scale + (size_t)((m < a.M ? m : a.M - 1) / a.ppi) * C + 4 * li);
                f = f * floatx4{sc.x, sc.y, sc.z, sc.w};
            }
            *reinterpret_cast<f16x4*>(As + (ps * RPP + sub) * ROWB + li * 8) = __builtin_convertvector(f, f16x4);
        }
    } else
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int m = m0 + ps * RPP + sub;
        float s = (v[ps].x + v[ps].y) + (v[ps].z + v[ps].w);
        for (int o = L >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * invC;
        float4 d = v[ps];
        d.x -= mean; d.y -= mean; d.z -= mean; d.w -= mean;
        float q = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        for (int o = L >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = 1.0f / sqrtf(q * invC + 1e-5f);
        const size_t frow = (size_t)(a.film_bstride ? (m < a.M ? m : a.M - 1) / a.ppi : 0) * a.film_bstride;
        const float4 s4 = reinterpret_cast<const float4*>(a.fscale + frow)[li];
        const float4 h4 = reinterpret_cast<const float4*>(a.fshift + frow)[li];
        float4 o4;
        o4.x = d.x * rstd * gg.x; o4.y = d.y * rstd * gg.y; o4.z = d.z * rstd * gg.z; o4.w = d.w * rstd * gg.w;
        o4.x = o4.x * (s4.x + 1.0f) + h4.x; o4.y = o4.y * (s4.y + 1.0f) + h4.y;
        o4.z = o4.z * (s4.z + 1.0f) + h4.z; o4.w = o4.w * (s4.w + 1.0f) + h4.w;
        const floatx4 f = {o4.x, o4.y, o4.z, o4.w};
        *reinterpret_cast<f16x4*>(As + (ps * RPP + sub) * ROWB + li * 8) = __builtin_convertvector(f, f16x4);
    }
    __syncthreads();
    // ---- 64 x 64 x C on the matrix pipe: wave (wm, wn) owns a 32 x 32 tile ----
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const char* ap = As + (wm * 32 + l31) * ROWB + h * 16;
    const char* bp = Bs + (wn * 32 + l31) * ROWB + h * 16;
#pragma unroll
    for (int ks = 0; ks < C / 16; ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(ap + ks * 32), *reinterpret_cast<const f16x8*>(bp + ks * 32), acc, 0, 0, 0);
    // ---- epilogue: acc[r] = row 8 (r >> 2) + 4 h + (r & 3), column l31 of the wave's tile ----
    const int n = n0 + wn * 32 + l31;
    const float bn = a.bias[n];
    if constexpr (PRO != 0) {   // out = res + (acc + bias) * ch_scale
        const float cs = a.ch_scale[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            if (m < a.M) {
                float t = (acc[r] + bn) * cs;
                t += a.res[(size_t)m * a.Cout + n];
                a.out[(size_t)m * a.Cout + n] = t;
            }
        }
    } else if (!a.gate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            if (m < a.M) a.out[(size_t)m * a.Cout + n] = acc[r] + bn;
        }
    } else {   // SimpleGate: the packed weight rows pair columns (2 j, 2 j + 1); product j, then the per-image lens FiLM
        const int ch = a.Cout >> 1, oc = n >> 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            const float t = acc[r] + bn;
            const float other = __shfl_xor(t, 1, 64);
            float gv = (lane & 1) ? other * t : t * other;   // v[2 j] * v[2 j + 1]
            if (a.gate_film) {
                const float* f = a.gate_film + (size_t)((m < a.M ? m : a.M - 1) / a.ppi) * a.gate_film_bstride;
                gv = gv * (f[oc] + 1.0f) + f[ch + oc];
            }
            if (!(lane & 1) && m < a.M) a.out[(size_t)m * ch + oc] = gv;
        }
    }
}

__global__ void row_gate_kernel(const float* __restrict__ in, float* __restrict__ out, const int rows, const int h) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * h) return;
    const int r = idx / h, j = idx - r * h;
    out[idx] = in[(size_t)r * 2 * h + j] * in[(size_t)r * 2 * h + h + j];
}

__global__ void fill_random_kernel(float* p, const size_t n, const unsigned seed, const float scale) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= n) return;
    float z[4];
    philox_normal4((uint32_t)q, (uint32_t)(q >> 32), seed, 0x5EEDULL, z);
    for (int k = 0; k < 4; ++k)
        if (q * 4 + k < n) p[q * 4 + k] = z[k] * scale;
}

__global__ void philox_normal_kernel(float* out, const int CHW, const int t, const uint64_t seed,
                                     const uint64_t image_offset) {
    const int b = blockIdx.y;
    const int quad = blockIdx.x * blockDim.x + threadIdx.x;
    if (quad * 4 >= CHW) return;
    float z[4];
    philox_normal4((uint32_t)quad, (uint32_t)t, (uint32_t)(image_offset + b), seed, z);
    for (int k = 0; k < 4; ++k)
        if (quad * 4 + k < CHW) out[(size_t)b * CHW + quad * 4 + k] = z[k];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------------
void launch_layernorm(const float* x, const float* g, const float* res, float* out, int64_t M, int C, float eps,
                      hipStream_t s, bool bf16) {
    if (bf16)
        launch_ln_t(reinterpret_cast<const bf16_t*>(x), g, reinterpret_cast<const bf16_t*>(res), reinterpret_cast<bf16_t*>(out), M,
                    C, eps, nullptr, nullptr, 0, 1, s);
    else
        launch_ln_t(x, g, res, out, M, C, eps, nullptr, nullptr, 0, 1, s);
}

void launch_layernorm_film(const float* x, const float* g, const float* scale, const float* shift, int film_bstride,
                           int64_t pixels_per_image, float* out, int64_t M, int C, float eps, hipStream_t s) {
    launch_ln_t(x, g, (const float*)nullptr, out, M, C, eps, scale, shift, film_bstride, pixels_per_image, s);
}

int dwgate_tiles(int H, int W, int c) {
    const DwGeom g = dw_geom(H, W, c);
    return g.tiles_x * g.tiles_y;
}

bool naf_lnconv_ok(int c, int Cout, long long M) { return (c == 64 || c == 128 || c == 256) && Cout % 64 == 0 && M > 0 && M < (1ll << 31); }

void naf_lnconv_global_init() {
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(naf_lnconv_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(naf_lnconv_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(naf_lnconv_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(naf_lnconv_kernel<256, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(naf_lnconv_kernel<256, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}

// norm + FiLM + 1x1 convolution (fp16 operands: w16 = the layer's fp16 weight copy [Cout][c]) in one launch; gate: SimpleGate epilogue (+ gate_film)
void launch_naf_lnconv(const float* x, const float* g, const float* fscale, const float* fshift, int film_bstride, int64_t pixels_per_image,
                       const unsigned short* w16, const float* bias, float* out, int64_t M, int c, int Cout, int gate, const float* gate_film,
                       int gate_film_bstride, hipStream_t s) {
    if (!naf_lnconv_ok(c, Cout, M)) throw HipError("naf_lnconv: unsupported shape");
    NafLnConvArgs a;
    a.x = x; a.g = g; a.fscale = fscale; a.fshift = fshift; a.film_bstride = film_bstride; a.ppi = pixels_per_image;
    a.w = w16; a.bias = bias; a.out = out; a.gate_film = gate_film; a.gate_film_bstride = gate_film_bstride; a.gate = gate;
    a.M = (int)M; a.Cout = Cout;
    a.in_scale = nullptr; a.ch_scale = nullptr; a.res = nullptr;
    const dim3 grid((unsigned)(((M + 63) / 64) * (Cout / 64)));
    const size_t lds = (size_t)128 * (c * 2 + 16);
    if (c == 64) hipLaunchKernelGGL(naf_lnconv_kernel<64>, grid, dim3(256), lds, s, a);
    else if (c == 128) hipLaunchKernelGGL(naf_lnconv_kernel<128>, grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL(naf_lnconv_kernel<256>, grid, dim3(256), lds, s, a);
    IRSDE_HIP_CHECK(hipGetLastError());
}

// conv3 / conv5 of a NAFBlock on the same kernel: out = res + (W (x * in_scale) + bias) * ch_scale, in_scale (per image and input channel) optional
void launch_naf_pwconv(const float* x, const float* in_scale, int64_t pixels_per_image, const unsigned short* w16, const float* bias, const float* ch_scale,
                       const float* res, float* out, int64_t M, int c, int Cout, hipStream_t s) {
    if (!naf_lnconv_ok(c, Cout, M) || !ch_scale || !res) throw HipError("naf_pwconv: unsupported shape");
    NafLnConvArgs a;
    a.x = x; a.g = nullptr; a.fscale = nullptr; a.fshift = nullptr; a.film_bstride = 0; a.ppi = pixels_per_image;
    a.w = w16; a.bias = bias; a.out = out; a.gate_film = nullptr; a.gate_film_bstride = 0; a.gate = 0;
    a.M = (int)M; a.Cout = Cout;
    a.in_scale = in_scale; a.ch_scale = ch_scale; a.res = res;
    const dim3 grid((unsigned)(((M + 63) / 64) * (Cout / 64)));
    const size_t lds = (size_t)128 * (c * 2 + 16);
#define IRSDE_PW(CC) do { if (in_scale) hipLaunchKernelGGL((naf_lnconv_kernel<CC, 1>), grid, dim3(256), lds, s, a); else hipLaunchKernelGGL((naf_lnconv_kernel<CC, 2>), grid, dim3(256), lds, s, a); } while (0)
    if (c == 64) IRSDE_PW(64); else if (c == 128) IRSDE_PW(128); else IRSDE_PW(256);
#undef IRSDE_PW
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_dwconv_gate(const float* u, const float* w, const float* bias, float* out, float* partial, int B, int H, int W,
                        int c, hipStream_t s) {
    if (c % 4) throw HipError("dwconv_gate: channel count must be a multiple of 4");
    const DwGeom g = dw_geom(H, W, c);
    const int nt = g.tiles_x * g.tiles_y;
    hipLaunchKernelGGL(dwconv_gate_kernel, dim3(nt, B), dim3(256), 0, s, u, w, bias, out, partial, H, W, c, g.tiles_x, nt, g.run);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_sca(const float* partial, int ntiles, const float* W, const float* bias, float* mean, float* s_out, int B,
                int c, int HW, hipStream_t s) {
    if ((long long)ntiles * c <= 65536 && c <= 4096) {   // small maps: one launch (every block re-sums <= 256 KB of partials from L2)
        hipLaunchKernelGGL(sca_fused_kernel, dim3((c + 3) / 4, B), dim3(256), (size_t)c * 4, s, partial, ntiles, W, bias, s_out, c, 1.0f / (float)HW);
    } else {
        hipLaunchKernelGGL(sca_mean_kernel, dim3((c + 63) / 64, B), dim3(256), 0, s, partial, ntiles, mean, c, 1.0f / (float)HW);
        hipLaunchKernelGGL(sca_kernel, dim3((c + 3) / 4, B), dim3(256), 0, s, mean, W, bias, s_out, c);
    }
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_row_gate(const float* in, float* out, int rows, int h, hipStream_t s) {
    const int total = rows * h;
    hipLaunchKernelGGL(row_gate_kernel, dim3((total + 255) / 256), dim3(256), 0, s, in, out, rows, h);
    IRSDE_HIP_CHECK(hipGetLastError());
}

// pixels per context chunk: 32 chunks per image, more for small batches so that the chunk grid (chunks x B blocks) still fills
// the 256 CUs (r03: at B = 2 the 64-block k/v kernels ran 4-5x slower per image than at B = 16); never below one 128-pixel tile
static int attn_chunk_len(int N, int B) {
    const int want = std::max(32, (512 + B - 1) / std::max(B, 1));
    int len = (N + want - 1) / want;
    if (len < 128) len = 128;
    return (len + 7) & ~7;
}
int attn_num_chunks(int N, int B) {
    const int len = attn_chunk_len(N, B);
    return (N + len - 1) / len;
}

template <typename T>
static void linear_attention_t(const T* qkv, T* out, int B, int N, const AttnWorkspace& ws, hipStream_t s) {
    const int len = attn_chunk_len(N, B);
    const int nch = attn_num_chunks(N, B);
    if (nch != ws.nch) throw HipError("attention workspace chunk mismatch");
    hipLaunchKernelGGL(attn_ctx_partial_kernel<T>, dim3(nch, B * kHeads), dim3(256), 0, s, qkv, ws.pmax, ws.pctx, ws.psum,
                       N, len, nch);
    hipLaunchKernelGGL(attn_ctx_finalize_kernel, dim3(B * kHeads, 4), dim3(1024), 0, s, ws.pctx, ws.psum, ws.pmax, ws.ctx, nch,
                       1.0f / (float)N, 1.0f / sqrtf((float)kDh));
    const int tiles = (N + 31) / 32;
    hipLaunchKernelGGL(attn_out_kernel<T>, dim3((tiles + kOutTilesPerBlock - 1) / kOutTilesPerBlock, B), dim3(256), 0, s,
                       qkv, ws.ctx, out, N, kQkv);
    IRSDE_HIP_CHECK(hipGetLastError());
}
// > 64 KB of dynamic LDS needs the attribute on the CURRENT device: called from conv_global_init() (engine finalize, per
// engine and so per device, outside any stream capture) like every other kernel of the library
void attention_global_init() {
#define IRSDE_KV_ATTR(CC) IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kv_ctx_kernel<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    IRSDE_KV_ATTR(32); IRSDE_KV_ATTR(64); IRSDE_KV_ATTR(96); IRSDE_KV_ATTR(128); IRSDE_KV_ATTR(160); IRSDE_KV_ATTR(192); IRSDE_KV_ATTR(224); IRSDE_KV_ATTR(256);
#undef IRSDE_KV_ATTR
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_q_out_fused_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_q_out_fused_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_q_out_fused_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kv_ctx_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kv_ctx_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kv_ctx_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#define IRSDE_A16_ATTR(CC, MM, AA)                                                                                                                  \
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kv_ctx_kernel<CC, MM, AA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_q_out_fused_kernel<CC, MM, AA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    IRSDE_A16_ATTR(64, 2, false); IRSDE_A16_ATTR(128, 2, false); IRSDE_A16_ATTR(256, 2, false);
    IRSDE_A16_ATTR(64, 2, true); IRSDE_A16_ATTR(128, 2, true); IRSDE_A16_ATTR(256, 2, true);
    IRSDE_A16_ATTR(64, 3, false); IRSDE_A16_ATTR(128, 3, false); IRSDE_A16_ATTR(256, 3, false);
#undef IRSDE_A16_ATTR
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_q_out_fused_kernel<64>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_q_out_fused_kernel<128>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_q_out_fused_kernel<256>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}
// fp32 fused form: context from xn and the k / v weight rows (attn_kv_ctx_kernel), then q (its own [B][N][128] tensor) -> out
void launch_attention_kv_context(const float* xn, const float* wkv, int B, int N, int C, const AttnWorkspace& ws, hipStream_t s,
                                 const float* ln_g, float ln_eps, const unsigned short* wkv_pair, float w_inv_scale, int op16, bool act_bf16) {
    if (ln_g && (C & (C - 1))) throw HipError("attention_kv_context: the fused LayerNorm needs a power-of-two channel count");
    if (C % 32 || C > 256 || C < 32) throw HipError("attention_kv_context: C must be a multiple of 32, <= 256");
    const int len = attn_chunk_len(N, B);
    const int nch = attn_num_chunks(N, B);
    if (nch != ws.nch) throw HipError("attention workspace chunk mismatch");
    if (op16) {   // one bf16 / fp16 operand plane (IRSDE_FLAG_BF16 / IRSDE_FLAG_FP16): C = 64 / 128 / 256
        if (!wkv_pair || (op16 != 2 && op16 != 3) || (act_bf16 && op16 != 2) || (C != 64 && C != 128 && C != 256))
            throw HipError("attention_kv_context: the 16-bit operand modes need their weight plane, C = 64 / 128 / 256; bf16 storage goes with bf16 operands");
        const size_t lds1 = (size_t)kKvTile * (2 * C + 16);
#define IRSDE_KV16_LAUNCH(CC, MM, AA) hipLaunchKernelGGL((attn_kv_ctx_kernel<CC, MM, AA>), dim3(nch, B), dim3(256), lds1, s, xn, wkv, ws.pmax, ws.pctx, ws.psum, N, len, nch, ln_g, ln_eps, wkv_pair, 1.f)
#define IRSDE_KV16_C(CC) { if (act_bf16) IRSDE_KV16_LAUNCH(CC, 2, true); else if (op16 == 2) IRSDE_KV16_LAUNCH(CC, 2, false); else IRSDE_KV16_LAUNCH(CC, 3, false); }
        if (C == 64) IRSDE_KV16_C(64) else if (C == 128) IRSDE_KV16_C(128) else IRSDE_KV16_C(256)
#undef IRSDE_KV16_C
#undef IRSDE_KV16_LAUNCH
        hipLaunchKernelGGL(attn_ctx_finalize_kernel, dim3(B * kHeads, 4), dim3(1024), 0, s, ws.pctx, ws.psum, ws.pmax, ws.ctx, nch,
                           1.0f / (float)N, 1.0f / sqrtf((float)kDh));
        IRSDE_HIP_CHECK(hipGetLastError());
        return;
    }
    if (wkv_pair) {   // fp16-pair projection (IRSDE_FLAG_SPLIT_F16X2): C = 64 / 128 / 256
        const size_t ldsp = (size_t)2 * kKvTile * (2 * C + 16);
#define IRSDE_KVP_LAUNCH(CC) hipLaunchKernelGGL((attn_kv_ctx_kernel<CC, true>), dim3(nch, B), dim3(256), ldsp, s, xn, wkv, ws.pmax, ws.pctx, ws.psum, N, len, nch, ln_g, ln_eps, wkv_pair, w_inv_scale)
        switch (C) {
            case 64: IRSDE_KVP_LAUNCH(64); break;
            case 128: IRSDE_KVP_LAUNCH(128); break;
            case 256: IRSDE_KVP_LAUNCH(256); break;
            default: throw HipError("attention_kv_context: the fp16-pair projection exists for C = 64, 128, 256");
        }
#undef IRSDE_KVP_LAUNCH
        hipLaunchKernelGGL(attn_ctx_finalize_kernel, dim3(B * kHeads, 4), dim3(1024), 0, s, ws.pctx, ws.psum, ws.pmax, ws.ctx, nch,
                           1.0f / (float)N, 1.0f / sqrtf((float)kDh));
        IRSDE_HIP_CHECK(hipGetLastError());
        return;
    }
    const size_t lds = (size_t)kKvTile * (C + 4) * sizeof(float);
#define IRSDE_KV_LAUNCH(CC) hipLaunchKernelGGL(attn_kv_ctx_kernel<CC>, dim3(nch, B), dim3(256), lds, s, xn, wkv, ws.pmax, ws.pctx, ws.psum, N, len, nch, ln_g, ln_eps)
    switch (C) {
        case 32: IRSDE_KV_LAUNCH(32); break;
        case 64: IRSDE_KV_LAUNCH(64); break;
        case 96: IRSDE_KV_LAUNCH(96); break;
        case 128: IRSDE_KV_LAUNCH(128); break;
        case 160: IRSDE_KV_LAUNCH(160); break;
        case 192: IRSDE_KV_LAUNCH(192); break;
        case 224: IRSDE_KV_LAUNCH(224); break;
        case 256: IRSDE_KV_LAUNCH(256); break;
        default: throw HipError("attention_kv_context: unsupported channel count");
    }
#undef IRSDE_KV_LAUNCH
    hipLaunchKernelGGL(attn_ctx_finalize_kernel, dim3(B * kHeads, 4), dim3(1024), 0, s, ws.pctx, ws.psum, ws.pmax, ws.ctx, nch,
                       1.0f / (float)N, 1.0f / sqrtf((float)kDh));
    IRSDE_HIP_CHECK(hipGetLastError());
}
void launch_attention_q_out(const float* q, float* out, int B, int N, const AttnWorkspace& ws, hipStream_t s) {
    const int tiles = (N + 31) / 32;
    hipLaunchKernelGGL(attn_out_kernel<float>, dim3((tiles + kOutTilesPerBlock - 1) / kOutTilesPerBlock, B), dim3(256), 0, s, q, ws.ctx,
                       out, N, kHid);
    IRSDE_HIP_CHECK(hipGetLastError());
}

// y = LayerNorm(to_out(softmax(q) . ctx)) * g + x with q = Wq . xn computed in the kernel (C = 64, 128 or 256)
void launch_attention_q_out_fused(const float* xn, const float* x, const float* wq, const float* wout, const float* bias,
                                  const float* g2, float* y, int B, int N, int C, float eps, const AttnWorkspace& ws, hipStream_t s,
                                  const float* ln_g, const unsigned short* wq_pair, const unsigned short* wout_pair, float wq_inv,
                                  float wout_inv, int op16, bool act_bf16) {
    if (C != 64 && C != 128 && C != 256) throw HipError("attention_q_out_fused: C must be 64, 128 or 256");
    if (op16) {   // one bf16 / fp16 operand plane per projection
        if (!wq_pair || !wout_pair || (op16 != 2 && op16 != 3) || (act_bf16 && op16 != 2))
            throw HipError("attention_q_out_fused: the 16-bit operand modes need their weight planes; bf16 storage goes with bf16 operands");
        const size_t plane_x = (size_t)128 * (2 * C + 16), plane_o = (size_t)128 * (2 * kHid + 16);
        const size_t ys = (size_t)128 * ((C > kHid ? C : kHid) + 4) * sizeof(float);
        const size_t lds1 = std::max(std::max(plane_x, plane_o), ys);
        const dim3 grid1((N + 127) / 128, B);
#define IRSDE_QO16_LAUNCH(CC, MM, AA) hipLaunchKernelGGL((attn_q_out_fused_kernel<CC, MM, AA>), grid1, dim3(256), lds1, s, xn, x, wq, ws.ctx, wout, bias, g2, y, N, eps, ln_g, wq_pair, wout_pair, 1.f, 1.f)
#define IRSDE_QO16_C(CC) { if (act_bf16) IRSDE_QO16_LAUNCH(CC, 2, true); else if (op16 == 2) IRSDE_QO16_LAUNCH(CC, 2, false); else IRSDE_QO16_LAUNCH(CC, 3, false); }
        if (C == 64) IRSDE_QO16_C(64) else if (C == 128) IRSDE_QO16_C(128) else IRSDE_QO16_C(256)
#undef IRSDE_QO16_C
#undef IRSDE_QO16_LAUNCH
        IRSDE_HIP_CHECK(hipGetLastError());
        return;
    }
    if (wq_pair && wout_pair) {   // fp16-pair projections (IRSDE_FLAG_SPLIT_F16X2)
        const size_t planes_x = (size_t)2 * 128 * (2 * C + 16), planes_o = (size_t)2 * 128 * (2 * kHid + 16);
        const size_t ys = (size_t)128 * ((C > kHid ? C : kHid) + 4) * sizeof(float);
        const size_t ldsp = std::max(std::max(planes_x, planes_o), ys);
        const dim3 gridp((N + 127) / 128, B);
        if (C == 64)
            hipLaunchKernelGGL((attn_q_out_fused_kernel<64, true>), gridp, dim3(256), ldsp, s, xn, x, wq, ws.ctx, wout, bias, g2, y, N, eps, ln_g,
                               wq_pair, wout_pair, wq_inv, wout_inv);
        else if (C == 128)
            hipLaunchKernelGGL((attn_q_out_fused_kernel<128, true>), gridp, dim3(256), ldsp, s, xn, x, wq, ws.ctx, wout, bias, g2, y, N, eps, ln_g,
                               wq_pair, wout_pair, wq_inv, wout_inv);
        else
            hipLaunchKernelGGL((attn_q_out_fused_kernel<256, true>), gridp, dim3(256), ldsp, s, xn, x, wq, ws.ctx, wout, bias, g2, y, N, eps, ln_g,
                               wq_pair, wout_pair, wq_inv, wout_inv);
        IRSDE_HIP_CHECK(hipGetLastError());
        return;
    }
    const size_t lds = (size_t)128 * ((C > kHid ? C : kHid) + 4) * sizeof(float);  // one region: max(xn tile, out tile)
    const dim3 grid((N + 127) / 128, B);
    if (C == 64)
        hipLaunchKernelGGL(attn_q_out_fused_kernel<64>, grid, dim3(256), lds, s, xn, x, wq, ws.ctx, wout, bias, g2, y, N, eps, ln_g);
    else if (C == 128)
        hipLaunchKernelGGL(attn_q_out_fused_kernel<128>, grid, dim3(256), lds, s, xn, x, wq, ws.ctx, wout, bias, g2, y, N, eps, ln_g);
    else
        hipLaunchKernelGGL(attn_q_out_fused_kernel<256>, grid, dim3(256), lds, s, xn, x, wq, ws.ctx, wout, bias, g2, y, N, eps, ln_g);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_linear_attention(const float* qkv, float* out, int B, int N, const AttnWorkspace& ws, hipStream_t s, bool bf16) {
    if (bf16)
        linear_attention_t(reinterpret_cast<const bf16_t*>(qkv), reinterpret_cast<bf16_t*>(out), B, N, ws, s);
    else
        linear_attention_t(qkv, out, B, N, ws, s);
}

void launch_full_attention(const float* qkv, float* out, int B, int N, hipStream_t s) {
    const int qtiles = (N + 31) / 32;
    hipLaunchKernelGGL(full_attn_kernel, dim3((qtiles + 3) / 4, B * kHeads), dim3(256), 0, s, qkv, out, N,
                       1.0f / sqrtf((float)kDh));
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_prep_input(const float* xt, const float* cond, float* x0, int B, int in_nc, int H, int W, int Hp, int Wp,
                       hipStream_t s, int reflect) {
    const int P = ((cond ? 2 : 1) * in_nc + 3) & ~3;
    if (P > 16) throw HipError("prep_input: more than 8 (conditional) / 16 (unconditional) input channels unsupported");
    const size_t total = (size_t)B * (Hp + 6) * (Wp + 6);
    hipLaunchKernelGGL(prep_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xt, cond, x0, B,
                       in_nc, P, H, W, Hp, Wp, reflect);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_sinusoid(const float* tvals, const float* freqs, float* out, int rows, int half, hipStream_t s) {
    const int total = rows * half;
    hipLaunchKernelGGL(sinusoid_kernel, dim3((total + 255) / 256), dim3(256), 0, s, tvals, freqs, out, rows, half);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_row_linear(const float* in, int in_stride, const float* W, const float* b, float* out, int out_stride,
                       int rows, int in_dim, int out_dim, int act_in, int act_out, hipStream_t s) {
    hipLaunchKernelGGL(row_linear_kernel, dim3((out_dim + 3) / 4), dim3(256), 0, s, in, in_stride, W, b, out,
                       out_stride, rows, in_dim, out_dim, act_in, act_out);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_step_begin(StepState* st, const float* film_table, int film_row, float* film_cur, const float* coef_table,
                       hipStream_t s) {
    hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(1024), 0, s, st, film_table, film_row, film_cur, coef_table);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_set_step(StepState* st, int t_next, hipStream_t s) {
    hipLaunchKernelGGL(set_step_kernel, dim3(1), dim3(1), 0, s, st, t_next);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_sde_update(const UpdateParams& p, hipStream_t s) {
    const int CHW = p.C * p.H * p.W;
    const int quads = (CHW + 3) / 4;
    hipLaunchKernelGGL(sde_update_kernel, dim3((quads + 255) / 256, p.B), dim3(256), 0, s, p);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_set_ctl(SampleCtl* ctl, int mode, const float* noise, long long noise_tstride, unsigned long long seed,
                    unsigned long long image_offset, hipStream_t s) {
    hipLaunchKernelGGL(set_ctl_kernel, dim3(1), dim3(1), 0, s, ctl, mode, noise, noise_tstride, seed, image_offset);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_unpack_pred(const float* pred, float* out, int B, int C, int H, int W, int Hp, int Wp, int stride, hipStream_t s) {
    const size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(unpack_pred_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pred, out, B, C, H,
                       W, Hp, Wp, stride);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int H, int W, hipStream_t s, bool bf16) {
    const size_t total = (size_t)B * C * H * W;
    if (bf16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const bf16_t*>(in), out, B, C, H, W);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, B, C, H,
                           W);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_fill_random(float* p, size_t n, unsigned seed, float scale, hipStream_t s) {
    const size_t quads = (n + 3) / 4;
    hipLaunchKernelGGL(fill_random_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, p, n, seed, scale);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_philox_normal(float* out, int B, int CHW, int t, uint64_t seed, uint64_t image_offset, hipStream_t s) {
    const int quads = (CHW + 3) / 4;
    hipLaunchKernelGGL(philox_normal_kernel, dim3((quads + 255) / 256, B), dim3(256), 0, s, out, CHW, t, seed,
                       image_offset);
    IRSDE_HIP_CHECK(hipGetLastError());
}

__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
void launch_bf16_to_f32(const unsigned short* in, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(in), out, n);
    IRSDE_HIP_CHECK(hipGetLastError());
}

// out = a + b (latent NAFNet: ending(x + intro); latent UNet: final_conv(x + h[0]))
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}
void launch_add(const float* a, const float* b, float* out, size_t n, hipStream_t s) {
    if (n % 4) throw HipError("launch_add: element count must be a multiple of 4");
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, a, b, out, n / 4);
    IRSDE_HIP_CHECK(hipGetLastError());
}

// NCHW [B][C][H][W] -> NHWC [B][Hp][Wp][Cp]: channels >= C are zero; rows/cols >= H/W are reflect-padded (reflect=1,
// F.pad 'reflect' on the right/bottom) or zero
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ in, float* __restrict__ out, const int B, const int C,
                                        const int H, const int W, const int Hp, const int Wp, const int Cp, const int reflect) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * Hp * Wp * Cp;
    if (idx >= total) return;
    const int c = (int)(idx % Cp);
    size_t r = idx / Cp;
    const int x = (int)(r % Wp); r /= Wp;
    const int y = (int)(r % Hp);
    const int b = (int)(r / Hp);
    float v = 0.f;
    if (c < C && (reflect || (y < H && x < W))) {
        const int sy = y < H ? y : 2 * (H - 1) - y, sx = x < W ? x : 2 * (W - 1) - x;
        v = in[(((size_t)b * C + c) * H + sy) * W + sx];
    }
    out[idx] = v;
}
void launch_nchw_to_nhwc_pad(const float* in, float* out, int B, int C, int H, int W, int Hp, int Wp, int Cp, int reflect,
                             hipStream_t s) {
    const size_t total = (size_t)B * Hp * Wp * Cp;
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, B, C, H, W, Hp,
                       Wp, Cp, reflect);
    IRSDE_HIP_CHECK(hipGetLastError());
}

// fp32 -> bf16 (round to nearest even) copy of packed conv weights for the bf16-MFMA mode
__global__ void f32_to_bf16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const __bf16 b = (__bf16)in[i];
    out[i] = __builtin_bit_cast(unsigned short, b);
}
void launch_f32_to_bf16(const float* in, unsigned short* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
    IRSDE_HIP_CHECK(hipGetLastError());
}

// fp32 -> IEEE fp16 (round to nearest even) copy of packed conv weights for the fp16-MFMA mode (IRSDE_FLAG_FP16)
__global__ void f32_to_f16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const _Float16 b = (_Float16)in[i];
    out[i] = __builtin_bit_cast(unsigned short, b);
}
void launch_f32_to_f16(const float* in, unsigned short* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
    IRSDE_HIP_CHECK(hipGetLastError());
}

}  // namespace irsde
