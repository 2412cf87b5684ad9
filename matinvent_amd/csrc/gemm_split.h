// fp32-class GEMM on the bf16 matrix pipe:  C[M,N] = epi( A[M,K] * W[N,K]^T ),  A, W, C fp32.
//
// On gfx950 the f32-input MFMA runs at the f32 VECTOR rate (1/16 of bf16 MFMA) and, as measured on the
// edge kernel, does not overlap VALU work.  Here every fp32 operand is split into three bf16 planes
// x = x0 + x1 + x2 (each the RNE-bf16 of the running remainder; the remainders are exact in fp32, so
// |x - (x0+x1+x2)| <= 2^-27 |x|) and the product is evaluated with the six terms whose weight is
// >= 2^-16 relative:  a0b0 + (a0b1 + a1b0) + (a1b1 + a0b2 + a2b0); the three dropped terms are <= 2^-24
// relative, i.e. fp32 round-off class.  bf16 x bf16 products are exact in fp32 and accumulate in fp32.
// Cost: 6 MFMAs of K=16 (32 cycles each) per 16 k  =  12 cycles per unit k per 32x32 tile, against 32 for
// v_mfma_f32_32x32x2_f32: 2.67x, on the real matrix pipe, leaving the VALU to split/stage.
//
// Tile: 128 x 128 outputs per 256-thread workgroup (4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles), BK = 32.
#pragma once
#include <type_traits>
#include "gemm.h"

namespace mi {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // (HIP's uint4 is a struct: arrays of it end up in scratch)

// RNE float -> bf16 bits (finite inputs)
__device__ __forceinline__ u16 f2bf(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }

__device__ __forceinline__ void split3(float x, u16& p0, u16& p1, u16& p2) {
    p0 = f2bf(x);
    float r = x - bf2f(p0);
    p1 = f2bf(r);
    r = r - bf2f(p1);
    p2 = f2bf(r);
}

// RNE fp32 -> bf16 pair in one instruction (gfx950 v_cvt_pk_bf16_f32): result = bf16(lo) | bf16(hi) << 16
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// three-plane split of two values at once: p[k] = plane k of (x, y) packed as (x | y << 16)
__device__ __forceinline__ void split3_pair(float x, float y, unsigned (&p)[3]) {
    p[0] = cvt_pk_bf16(x, y);
    float rx = x - __uint_as_float(p[0] << 16), ry = y - __uint_as_float(p[0] & 0xFFFF0000u);
    p[1] = cvt_pk_bf16(rx, ry);
    rx -= __uint_as_float(p[1] << 16);
    ry -= __uint_as_float(p[1] & 0xFFFF0000u);
    p[2] = cvt_pk_bf16(rx, ry);
}

// ------------------------------------------------------------------------------------------------------------------------
// Element format of the PRE-SPLIT plane sets (the operands of gemm_planes*).
//   MI_PLANES_FP16 = 1:  two fp16 planes,  x * s = h0 + h1  with h0 = fp16(x s), h1 = fp16(x s - h0): 22 mantissa bits in two
//     16-bit words, so a product needs THREE MFMA terms (h0 g0 + h0 g1 + h1 g0; the dropped h1 g1 is 2^-22 relative) instead
//     of the six of the three-plane bf16 split -- half the matrix-pipe work, two thirds of the operand bytes.  fp16 has a
//     narrow exponent range, so each operand class carries a fixed power-of-two scale (below) chosen for the magnitudes it can
//     take; the accumulator is multiplied by 1 / (sA sW) before the epilogue (a power of two: exact).  Where the scaled value
//     is small the residual plane goes subnormal and the representation error becomes ABSOLUTE (2^-25 / scale) instead of
//     relative; values beyond 65504 / scale saturate (finite, wrong) -- use the three-plane build for such networks.
//   MI_PLANES_FP16 = 0:  three bf16 planes, six terms, no range limits.
// Plane 2 of the tile layout is unused in the fp16 format (neither written nor read).
// ------------------------------------------------------------------------------------------------------------------------
#ifndef MI_PLANES_FP16
#define MI_PLANES_FP16 1
#endif
// Ablation-only instantiations that SPILL registers (the segmented-sum epilogue on the 256 x 256 LDS-DMA kernel: 576 B of scratch; the
// four-waves-per-SIMD build of the 128-row loop: 15-16 spilled registers) are compiled only with -DMI_ABLATION_KERNELS: the default library
// ships no kernel with scratch on any path, and the switches that select them (mi_debug_set_planes_big_seg, mi_debug_set_planes_dma(4))
// return MI_EINVAL without it.
// TF32-CLASS BUILD (-DMI_TF32_CLASS=1 -> lib/libmatinvent_hip_tf32.so, `python -m matinvent_amd.build --tf32`; never the default, never the
// headline): every product of two PLANE SETS keeps its leading term only -- fp16(s a) x fp16(s b), i.e. 11-bit significands with f32
// accumulation, the arithmetic class the reference itself runs after torch.set_float32_matmul_precision("high") (pipeline/mat_invent.py:127,
// TF32 on its hardware) -- a third of the matrix-pipe work.  Plane sets, scales, saturation accounting, the fp32 sums (segmented means,
// reductions) and the fp32-operand products that split on the fly are unchanged.
#ifndef MI_TF32_CLASS
#define MI_TF32_CLASS 0
#endif
#define MI_TERM0 (MI_TF32_CLASS ? 2 : 0)   // first term of the (a1 b0), (a0 b1), (a0 b0) sequences
#ifdef MI_ABLATION_KERNELS
#define MI_HAVE_ABLATION_KERNELS 1
#else
#define MI_HAVE_ABLATION_KERNELS 0
#endif
constexpr int NPL = MI_PLANES_FP16 ? 2 : 3;
// scales by operand class (a plane set carries its scale in Planes::scale; the GEMM multiplies the accumulator by 1 / (sA sW)):
constexpr float PL_SW = MI_PLANES_FP16 ? 64.f : 1.f;      // weights: exact up to |w| = 1023, residual plane normal down to |w| ~ 2e-3
constexpr float PL_S_UNIT = MI_PLANES_FP16 ? 64.f : 1.f;  // Fourier features, |x| <= 1
constexpr float PL_S_LN = MI_PLANES_FP16 ? 8.f : 1.f;     // LayerNorm outputs (|x| <= sqrt(H) |w_ln|): up to 8188
constexpr float PL_S_ACT = MI_PLANES_FP16 ? 0.125f : 1.f;  // unbounded activations (SiLU outputs, aggregated messages): up to 5.2e5,
                                                           // absolute error <= 2.4e-7 below |x| ~ 1 (the residual plane goes subnormal)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_f16(float lo, float hi) {
    f16x2 h = {(_Float16)lo, (_Float16)hi};  // round to nearest even
    return __builtin_bit_cast(unsigned, h);
}
// Saturation guard of the fp16 plane format: every conversion that had to clamp (or met a NaN) counts itself here, so finite
// garbage can never be silent -- mi_saturation_events() reads the counters of all translation units (one copy per unit: the
// library is built without relocatable device code).  The test is one compare on the hot path; the atomic runs only on a hit.
static __device__ unsigned g_sat_events;
__device__ __forceinline__ void sat_note(float a, float b = 0.f) {
    if (!(fabsf(a) <= 65504.f) | !(fabsf(b) <= 65504.f)) atomicAdd(&g_sat_events, 1u);  // (NaN and inf count too)
}
static inline int sat_fetch(unsigned* out, bool reset) {
    unsigned v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_sat_events), sizeof(v)) != hipSuccess) return MI_EHIP;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_sat_events), &zero, sizeof(zero)) != hipSuccess) return MI_EHIP;
    *out = v;
    return MI_OK;
}
// Every translation unit that includes this header owns a copy of the counter (no relocatable device code) and registers its
// reader at library load, so that mi_saturation_events() (cspnet.hip) sums ALL of them -- a unit cannot be forgotten.
void sat_register(int (*fetch)(unsigned*, bool));
namespace {
struct SatRegistrar {
    SatRegistrar() { sat_register(&sat_fetch); }
};
static SatRegistrar g_sat_registrar;
}  // namespace
// plane words of the element pair (x, y): p[k] = plane k of x | plane k of y << 16; `scale` = the destination plane set's scale
__device__ __forceinline__ void pl_split_pair(float x, float y, float scale, unsigned (&p)[3]) {
#if MI_PLANES_FP16
    // (saturate instead of overflowing to inf: inf x 0 in a padded column would poison the whole row -- and say so)
    const float xr = x * scale, yr = y * scale;
    sat_note(xr, yr);
    const float xs = fminf(fmaxf(xr, -65504.f), 65504.f), ys = fminf(fmaxf(yr, -65504.f), 65504.f);
    const f16x2 h0 = {(_Float16)xs, (_Float16)ys};
    p[0] = __builtin_bit_cast(unsigned, h0);
    p[1] = pack_f16(xs - (float)h0[0], ys - (float)h0[1]);
    p[2] = 0u;
#else
    (void)scale;
    split3_pair(x, y, p);
#endif
}
// The same with the saturation test ACCUMULATED into `sat` instead of a branch with an atomic per conversion: lets the compiler schedule a
// whole epilogue as straight-line code; the caller reports once with sat_report().  fp16 format, per element pair: clamp (2 x v_med3),
// ONE packed conversion for the leading plane, the residuals x - (float)h straight from its halves (2 x v_fma_mix_f32: the compiler's own
// form converted every element twice -- v_cvt_f16_f32 for the residual chain next to the v_cvt_pk_f16_f32 for the store -- and back with
// v_cvt_f32_f16 + v_sub: 7 operations per element where this takes 4), one packed conversion for the second plane; and the saturation test
// on the BITS of the leading plane: a half that was clamped (|h| = 65504 = 0x7BFF) or is a NaN (> 0x7C00) carries into bit 15 when 0x0401
// is added to its magnitude -- and / add / or per PAIR instead of compare / select / or per element.  `sat` is meaningful in bits 15 and 31.
__device__ __forceinline__ void pl_split_scaled_pair_acc(float xr, float yr, unsigned (&p)[3], unsigned& sat) {   // (xr, yr: already multiplied by the scale)
#if MI_PLANES_FP16 && defined(MI_AB_SPLIT_R3)   // (A/B builds, scripts/gpu_epi_ab.sh: round 3's form of the split, the compiler's own instruction choice)
    sat |= ((unsigned)(!(fabsf(xr) <= 65504.f)) | (unsigned)(!(fabsf(yr) <= 65504.f))) << 15;
    const float xs0 = fminf(fmaxf(xr, -65504.f), 65504.f), ys0 = fminf(fmaxf(yr, -65504.f), 65504.f);
    const f16x2 h0 = {(_Float16)xs0, (_Float16)ys0};
    p[0] = __builtin_bit_cast(unsigned, h0);
    p[1] = pack_f16(xs0 - (float)h0[0], ys0 - (float)h0[1]);
    p[2] = 0u;
#elif MI_PLANES_FP16
    // (the instruction itself: fminf(fmaxf()) makes the compiler canonicalise -- v_max x, x -- a value that reaches it through a branch
    //  merge; a NaN comes out as -65504 -- v_med3 returns the minimum of the non-NaN operands -- and is counted like a clamp)
    const float xs = __builtin_amdgcn_fmed3f(xr, -65504.f, 65504.f), ys = __builtin_amdgcn_fmed3f(yr, -65504.f, 65504.f);
    const unsigned hi = pack_f16(xs, ys);
    sat |= (hi & 0x7fff7fffu) + 0x04010401u;   // (no carry between the halves: a masked half + 0x0401 < 0x10000)
    float lx, ly;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lx) : "v"(hi), "v"(xs));   // xs - (float)hi.lo, exact
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ly) : "v"(hi), "v"(ys));   // ys - (float)hi.hi
    p[0] = hi;
    p[1] = pack_f16(lx, ly);
    p[2] = 0u;
#else
    (void)sat;
    split3_pair(xr, yr, p);
#endif
}
__device__ __forceinline__ void pl_split_pair_acc(float x, float y, float scale, unsigned (&p)[3], unsigned& sat) {
#if MI_PLANES_FP16
    pl_split_scaled_pair_acc(x * scale, y * scale, p, sat);
#else
    (void)scale;
    (void)sat;
    split3_pair(x, y, p);
#endif
}
__device__ __forceinline__ void sat_report(unsigned sat) {
    if (sat & 0x80008000u) atomicAdd(&g_sat_events, 1u);
}
__device__ __forceinline__ void pl_split(float x, float scale, u16& p0, u16& p1, u16& p2) {
#if MI_PLANES_FP16
    const float xr = x * scale;
    sat_note(xr);
    const float xs = fminf(fmaxf(xr, -65504.f), 65504.f);
    const _Float16 h0 = (_Float16)xs, h1 = (_Float16)(xs - (float)h0);
    p0 = __builtin_bit_cast(u16, h0);
    p1 = __builtin_bit_cast(u16, h1);
    p2 = 0;
#else
    (void)scale;
    split3(x, p0, p1, p2);
#endif
}

// scale of an on-the-fly two-plane fp16 split from the operand's exact absmax (bit pattern): 2^floor(log2(16384 / absmax)), so that
// the largest stored magnitude stays below 32768 -- saturation is impossible by construction
__device__ __forceinline__ float f16_scale_from_absmax(unsigned bits) {
    const float m = __uint_as_float(bits);
    if (!(m > 0.f) || !(m < 3e38f)) return 1.f;
    int e = 14 - (int)ceilf(log2f(m));
    e = e > 40 ? 40 : (e < -100 ? -100 : e);
    return exp2f((float)e);
}
// max |x| of a tensor as a bit pattern (order-independent: deterministic); the slot must be zero before the launch
// (spread_mask: the waves' atomics go to out[blockIdx & mask] -- same-address atomics serialise at ~8.5 ns each in the L2, so a
// launch of thousands of waves spreads them over a power-of-two row of sub-slots that the reader folds; 0 = one slot)
// (mdev / per_row: optional device-side row count -- the tensor has *mdev rows of per_row elements, n is the capacity)
template <int MI_UNUSED = 0>   // (a template so that only the units that launch it emit it)
static __global__ void absmax_bits_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out, int spread_mask = 0,
                                          const int* __restrict__ mdev = nullptr, int per_row = 0) {
    float m = 0.f;
    if (mdev) {
        const int64_t nd = (int64_t)*mdev * per_row;
        n = nd < n ? nd : n;
    }
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {   // 16-byte loads (4-byte ones ran at 2.3 TB/s on a [64k, 512] tensor)
        const int64_t n4 = n >> 2;
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        int64_t i = tid;
        for (; i + 3 * nthr < n4; i += 4 * nthr) {   // four independent 16-byte loads in flight per lane
            const f32x4 a = x4[i], b = x4[i + nthr], c = x4[i + 2 * nthr], d = x4[i + 3 * nthr];
            const float ma = fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3])));
            const float mb = fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3])));
            const float mc = fmaxf(fmaxf(fabsf(c[0]), fabsf(c[1])), fmaxf(fabsf(c[2]), fabsf(c[3])));
            const float md = fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fmaxf(fabsf(d[2]), fabsf(d[3])));
            m = fmaxf(m, fmaxf(fmaxf(ma, mb), fmaxf(mc, md)));
        }
        for (; i < n4; i += nthr) {
            const f32x4 v = x4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nthr) m = fmaxf(m, fabsf(x[i]));
    } else {
        for (int64_t i = tid; i < n; i += nthr) m = fmaxf(m, fabsf(x[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out + (blockIdx.x & spread_mask), __float_as_uint(m));
}

// F16 = true: the fp32 operands are split on the fly into TWO fp16 planes (three MFMA terms instead of six) with power-of-two scales
// derived from their exact absmax (amax_a / amax_w: device bit patterns); the accumulator is rescaled before the epilogue.
template <int BM, int BN, bool F16 = false>
__global__ __launch_bounds__(256) void gemm_nt_split_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                            float* __restrict__ C, int ldc, int M, int N, int K, GemmEpilogue ep, int kchunk,
                                                            float* __restrict__ part, const unsigned* __restrict__ amax_a = nullptr,
                                                            const unsigned* __restrict__ amax_w = nullptr) {
    constexpr int BK = 32, ROWB = 80;            // bytes per row per plane (64 + 16 pad: conflict-free b128 reads)
    constexpr int TM = BM / 64, TN = BN / 64;    // 32x32 tiles per wave along M / N
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                    // [3 planes][BM rows][ROWB]
    unsigned char* Ws = smem + 3 * BM * ROWB;    // [3 planes][BN rows][ROWB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kg = lane >> 5;
    const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;
    // split-K: slice z of the reduction; with more than one slice the raw partial sums go to part[z][M][N] and
    // splitk_reduce_kernel adds them in a fixed order and applies the epilogue
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int A_V = BM * BK / 4 / 256, W_V = BN * BK / 4 / 256;
    // two register stages: the fp32 tiles of k-step t+2 are requested while step t is split, staged and multiplied, so a
    // lone workgroup on a CU (node-level products) is not serialised on the global-load latency
    f32x4 ra[2][A_V], rw[2][W_V];
    auto load_tiles = [&](int k0, f32x4 (&qa)[A_V], f32x4 (&qw)[W_V]) {
#pragma unroll
        for (int v = 0; v < A_V; ++v) {
            int f = tid + v * 256, r = f >> 3, c = (f & 7) * 4, gr = row0 + r, gk = k0 + c;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (gr < M && gk < kend) val = *reinterpret_cast<const f32x4*>(A + (size_t)gr * lda + gk);
            qa[v] = val;
        }
#pragma unroll
        for (int v = 0; v < W_V; ++v) {
            int f = tid + v * 256, r = f >> 3, c = (f & 7) * 4, gr = col0 + r, gk = k0 + c;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (gr < N && gk < kend) val = *reinterpret_cast<const f32x4*>(W + (size_t)gr * ldw + gk);
            qw[v] = val;
        }
    };
    float sc_a = 1.f, sc_w = 1.f;
    if (F16) {
        sc_a = f16_scale_from_absmax(*amax_a);
        sc_w = f16_scale_from_absmax(*amax_w);
    }
    auto store_split = [&](unsigned char* base, int rows, const f32x4& val, int f, float sc) {
        int r = f >> 3, c = (f & 7) * 4;
        unsigned lo[3], hi[3];
        if (F16) {
            const float x0 = val[0] * sc, x1 = val[1] * sc, x2 = val[2] * sc, x3 = val[3] * sc;
            const f16x2 h0 = {(_Float16)x0, (_Float16)x1}, h1 = {(_Float16)x2, (_Float16)x3};
            lo[0] = __builtin_bit_cast(unsigned, h0);
            hi[0] = __builtin_bit_cast(unsigned, h1);
            lo[1] = pack_f16(x0 - (float)h0[0], x1 - (float)h0[1]);
            hi[1] = pack_f16(x2 - (float)h1[0], x3 - (float)h1[1]);
        } else {
            split3_pair(val[0], val[1], lo);
            split3_pair(val[2], val[3], hi);
        }
#pragma unroll
        for (int pl = 0; pl < (F16 ? 2 : 3); ++pl) {
            uint2 pk = {lo[pl], hi[pl]};
            *reinterpret_cast<uint2*>(base + ((size_t)pl * rows + r) * ROWB + c * 2) = pk;
        }
    };
    auto step = [&](int k0, f32x4 (&qa)[A_V], f32x4 (&qw)[W_V]) {
#pragma unroll
        for (int v = 0; v < A_V; ++v) store_split(As, BM, qa[v], tid + v * 256, sc_a);
#pragma unroll
        for (int v = 0; v < W_V; ++v) store_split(Ws, BN, qw[v], tid + v * 256, sc_w);
        __syncthreads();
        load_tiles(k0 + 2 * BK, qa, qw);  // past-the-end tiles load as zeros
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 a[TM][3], b[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < (F16 ? 2 : 3); ++pl)
                    a[i][pl] = *reinterpret_cast<const bf16x8*>(As + ((size_t)pl * BM + (wm * TM + i) * 32 + l31) * ROWB + (16 * s + 8 * kg) * 2);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < (F16 ? 2 : 3); ++pl)
                    b[j][pl] = *reinterpret_cast<const bf16x8*>(Ws + ((size_t)pl * BN + (wn * TN + j) * 32 + l31) * ROWB + (16 * s + 8 * kg) * 2);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (F16) {
#if !MI_TF32_CLASS   // (the TF32-class build keeps the leading term only: 11-bit operands, f32 accumulate)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][1]), __builtin_bit_cast(f16x8, b[j][0]), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][0]), __builtin_bit_cast(f16x8, b[j][1]), acc[i][j], 0, 0, 0);
#endif
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][0]), __builtin_bit_cast(f16x8, b[j][0]), acc[i][j], 0, 0, 0);
                        continue;
                    }
                    // smallest terms first, so that they are not lost against a large accumulator
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    };

    load_tiles(kbeg, ra[0], rw[0]);
    load_tiles(kbeg + BK, ra[1], rw[1]);
    for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {  // steps in pairs (an odd last step multiplies a zero tile)
        step(k0, ra[0], rw[0]);
        step(k0 + BK, ra[1], rw[1]);
    }

    // (Recorded ablation, round 5: issuing every optional addend load of the tile before the arithmetic -- the element-by-element form below cannot move
    //  element r + 1's loads above element r's store, C may alias the addend arrays: 129 tight `s_waitcnt vmcnt(0)` in the 64 x 64 instantiation -- measured
    //  +0.2 % on the fine-tune line and -2 % on the MatterGen-shaped sampler, profiles/r5_nt_split_epi_ab.log: the launches that carry addends are
    //  split-K ones, whose epilogue runs in splitk_reduce_kernel.  Not kept.)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int col = col0 + (wn * TN + j) * 32 + l31;
            if (col >= N) continue;
            float bcol = ep.bias ? ep.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = row0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (row >= M) continue;
                const float av = F16 ? acc[i][j][r] * (1.0f / (sc_a * sc_w)) : acc[i][j][r];   // (powers of two: exact)
                if (gridDim.z > 1) {
                    part[((size_t)blockIdx.z * M + row) * N + col] = av;
                    continue;
                }
                float v = av + bcol;
                v = apply_epilogue(ep, v, row, col);
                C[(size_t)row * ldc + col] = v;
            }
        }
}

#ifdef MI_GEMM_OWNER   // (launched by the owning unit's launchers only: a static kernel is emitted by EVERY unit that sees its definition)
static __global__ void splitk_reduce_kernel(const float* __restrict__ part, int S, int M, int N, float* __restrict__ C, int ldc, GemmEpilogue ep) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int row = (int)(idx / N), col = (int)(idx % N);
    float v = 0.f;
    int z = 0;
    const size_t MN = (size_t)M * N;
    for (; z + 4 <= S; z += 4) {   // (four slices requested together, added in slice order: one memory latency per four slices instead of one per slice)
        const float p0 = part[(size_t)z * MN + idx], p1 = part[(size_t)(z + 1) * MN + idx], p2 = part[(size_t)(z + 2) * MN + idx],
                    p3 = part[(size_t)(z + 3) * MN + idx];
        v += p0;
        v += p1;
        v += p2;
        v += p3;
    }
    for (; z < S; ++z) v += part[(size_t)z * MN + idx];
    v += ep.bias ? ep.bias[col] : 0.f;
    C[(size_t)row * ldc + col] = apply_epilogue(ep, v, row, col);
}
#endif

#ifdef MI_GEMM_OWNER   // (defined once, in the owning translation unit: every unit that defined it also carried its kernels)
int gemm_nt_split(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, const GemmEpilogue& ep,
                         hipStream_t s, const SplitK* sk = nullptr, const unsigned* amax_a = nullptr, const unsigned* amax_w = nullptr) {
    MI_CHECK(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, MI_EINVAL, "gemm_nt_split: K/lda/ldw must be multiples of 4");
    if (M <= 0 || N <= 0) return MI_OK;
    count_mfma(M, N, K, (amax_a && amax_w && (int64_t)cdiv(M, 128) * cdiv(N, 128) >= 256) ? (MI_TF32_CLASS ? 1 : 3) : (MI_TF32_CLASS ? 1 : 6));
    if ((int64_t)cdiv(M, 128) * cdiv(N, 128) >= 256) {
        constexpr int BM = 128, BN = 128;
        if (amax_a && amax_w)   // two fp16 planes split on the fly, three MFMA terms (scales from the operands' exact absmax)
            hipLaunchKernelGGL((gemm_nt_split_kernel<BM, BN, true>), dim3(cdiv(N, BN), cdiv(M, BM)), dim3(256), 3 * (BM + BN) * 80, s, A, lda, W, ldw, C,
                               ldc, M, N, K, ep, K, (float*)nullptr, amax_a, amax_w);
        else
            hipLaunchKernelGGL((gemm_nt_split_kernel<BM, BN>), dim3(cdiv(N, BN), cdiv(M, BM)), dim3(256), 3 * (BM + BN) * 80, s, A, lda, W, ldw, C, ldc,
                               M, N, K, ep, K, (float*)nullptr, (const unsigned*)nullptr, (const unsigned*)nullptr);
    } else {
        constexpr int BM = 64, BN = 64;
        // few output tiles and a long reduction: the serial k-loop is the latency -- cut it into slices
        const int tiles = cdiv(M, BM) * cdiv(N, BN);
        int S = 1;
        if (sk && sk->buf && tiles <= 128 && K >= 256) {
            S = std::min(8, K / 128);
            while (S > 1 && (tiles * S > 512 || (size_t)S * M * N > sk->floats)) --S;
        }
        const int kchunk = S > 1 ? cdiv(cdiv(K, S), 64) * 64 : K;
        S = cdiv(K, kchunk);
        hipLaunchKernelGGL((gemm_nt_split_kernel<BM, BN>), dim3(cdiv(N, BN), cdiv(M, BM), S), dim3(256), 3 * (BM + BN) * 80, s, A, lda, W, ldw, C,
                           ldc, M, N, K, ep, kchunk, S > 1 ? sk->buf : (float*)nullptr, (const unsigned*)nullptr, (const unsigned*)nullptr);
        if (S > 1) hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv((int64_t)M * N, 256)), dim3(256), 0, s, sk->buf, S, M, N, C, ldc, ep);
    }
    MI_KERNEL_CHECK();
    return MI_OK;
}
#else
int gemm_nt_split(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, const GemmEpilogue& ep,
                         hipStream_t s, const SplitK* sk = nullptr, const unsigned* amax_a = nullptr, const unsigned* amax_w = nullptr);
#endif

// Which matrix path the node- and edge-level GEMMs take (process-wide; set by mi_set_gemm_mode).
//   MI_GEMM_SPLIT (default): three-plane bf16 split, six terms -- fp32-class accuracy, bf16 matrix pipe
//   MI_GEMM_F32            : v_mfma_f32_32x32x2_f32 -- bit-for-bit an fp32 fma chain
extern int g_gemm_mode;
extern int g_planes_variant;  // 0 = 128x128 tiles, 1 = 256x128 double-buffered (large M)
extern int g_planes_db_min_tiles;
extern int g_planes_small_tiles;  // plain plane GEMMs with fewer 128-row tiles than this use 64-row tiles
extern int g_planes_big;          // 1: row-major-epilogue products with M >= g_planes_big_min_rows and N % 256 == 0 on the 256 x 256 LDS-DMA kernel
extern int g_planes_big_min_rows;
extern int g_planes_rt;           // row-major-epilogue products with a fragment-order W, N % 256 == 0, K % 64 == 0 on the 128 x 256 register-tile kernel: 0 off, 1 = those with epilogue extensions, 2 = all
extern int g_planes_rt_min_rows;
extern int g_planes_big_seg_min_rows;  // > 0: products with the fused segmented sum (second edge GEMM, inference) from this many rows up on the 256 x 256 kernel too
extern int g_planes_dma;             // 128 x 128 tiles fed by LDS-DMA: 0 = never, 1 = launches of at most g_planes_lat_max_blocks workgroups, 2 = every launch
extern int g_planes_lat_max_blocks;  // plane GEMMs of at most this many workgroups run the latency form (deep operand prefetch); 0 = never
extern int g_pair_kernel;  // 0 = 128-row kernel for pair mode (default), 1 = size-based choice
inline int gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, const GemmEpilogue& ep,
                   hipStream_t s, const SplitK* sk = nullptr) {
    return g_gemm_mode == 0 ? gemm_nt_f32(A, lda, W, ldw, C, ldc, M, N, K, ep, s) : gemm_nt_split(A, lda, W, ldw, C, ldc, M, N, K, ep, s, sk);
}

}  // namespace mi

namespace mi {

// ------------------------------------------------------------------------------------------------
// Pre-split operands.  A "plane set" stores an fp32 matrix [rows][cols] as three bf16 matrices.
// Splitting happens ONCE where the data is produced (weights at pack time, Fourier features in their
// generator, M1 in the epilogue of the first edge GEMM) instead of in every consumer tile.
//
// Layout: TILE-BLOCKED.  The matrix is cut into 128-row x 32-column tiles; tile (rt, kt) is one contiguous
// block [plane 3][row 128][col 32] of bf16 (24 KiB).  A GEMM workgroup therefore fetches each operand
// tile as 24 perfectly coalesced 1 KiB wave loads -- the row-major form touched only 64 B of every
// 128 B line and ran at ~5 TB/s of L2 traffic, which is what bounded the kernel.
//   element (r, c, plane p) -> base[ ((r/128) * KT + c/32) * 12288 + p * 4096 + (r%128) * 32 + c%32 ],  KT = ceil(cols/32)
// Rows are padded to a multiple of 128, columns to a multiple of 32; pads must be zero for the K direction.
// ------------------------------------------------------------------------------------------------
struct Planes {
    u16* base = nullptr;
    int KT = 0;  // column tiles per row tile
    float scale = 1.f;  // power-of-two scale of the stored values (fp16 two-plane format; 1 for the bf16 format)
    const float* dscale = nullptr;  // optional DEVICE-side {scale, 1 / scale}: per-evaluation scale of an unbounded activation class
    // optional, W operands only: the same [N x K] values in MFMA FRAGMENT order [N / 32][K / 16][plane][lane][8] (pack_frag_from_planes),
    // which the register-tile GEMM (gemm_rt, edge_stage.hip) streams from L2 straight into registers
    const u16* frag = nullptr;
    __device__ __forceinline__ float s() const { return dscale ? dscale[0] : scale; }
    // +2048 elements (4 KiB) per row tile: without the skew every row tile starts a multiple of 64 KiB apart,
    // i.e. on the same memory channel, and workgroups marching through k in lockstep hammer a few channels
    // (measured: 2x slower than the row-major form).
    __host__ __device__ size_t tile(int rt, int kt) const { return (size_t)rt * ((size_t)KT * 12288 + 2048) + (size_t)kt * 12288; }
    __host__ __device__ size_t elem(int r, int c, int p) const { return tile(r >> 7, c >> 5) + (size_t)p * 4096 + (r & 127) * 32 + (c & 31); }
};
static inline size_t planes_elems(int64_t rows, int cols) { return (size_t)((rows + 127) / 128) * ((size_t)((cols + 31) / 32) * 12288 + 2048); }
static inline Planes make_planes(u16* base, int cols, float scale = PL_SW, const float* dscale = nullptr) {
    return Planes{base, (cols + 31) / 32, scale, MI_PLANES_FP16 ? dscale : nullptr};
}

// fp32 [rows][cols] (row stride ld_src) -> plane set (pads written as zero); one thread per column pair
// (row0: first destination row -- lets several matrices share one plane set; the zero row padding then runs to the next
// multiple of 128 destination rows, so stack them in increasing row order)
template <int MI_UNUSED = 0>   // (a template so that only the units that launch it emit it)
static __global__ void split_planes_kernel(const float* __restrict__ src, int ld_src, int rows, int cols, Planes dst, int row0 = 0) {
    const int cp = dst.KT * 16;  // column pairs per padded row
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows_pad = (int64_t)(row0 + rows + 127) / 128 * 128 - row0;
    if (idx >= rows_pad * cp) return;
    int r = (int)(idx / cp), c = (int)(idx % cp) * 2;
    float x = (r < rows && c < cols) ? src[(size_t)r * ld_src + c] : 0.f;
    float y = (r < rows && c + 1 < cols) ? src[(size_t)r * ld_src + c + 1] : 0.f;
    unsigned p[3];
    pl_split_pair(x, y, dst.s(), p);
#pragma unroll
    for (int k = 0; k < NPL; ++k) *reinterpret_cast<unsigned*>(dst.base + dst.elem(row0 + r, c, k)) = p[k];
}

// buffer descriptor over [p, p + bytes) with every field forced into scalar registers (the compiler otherwise
// treats block-uniform pointer arithmetic as divergent and wraps each buffer load in a waterfall loop)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, int bytes) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// the same descriptor as four scalar words (an operand of inline-asm buffer loads)
__device__ __forceinline__ u32x4 rsrc_words(const void* p, int bytes) {
    const uint64_t a = (uint64_t)p;
    return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((uint32_t)a), (unsigned)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) & 0xffffu,
                 (unsigned)__builtin_amdgcn_readfirstlane(bytes), 0x00020000u};
}

struct PlanesEpilogue {
    GemmEpilogue ep;       // bias / gathers / pre_act / act / residual as for the fp32 kernels
    float out_scale = 1.f; // 1 / (A.scale * W.scale), set by gemm_planes: applied to the accumulator first
    const float* a_dinv = nullptr;  // device-side 1 / (dynamic scale of A), multiplied in as well (set by gemm_planes)
    unsigned* absmax = nullptr;     // optional: atomicMax of the bit pattern of max |output| (row-major epilogue only)
    int absmax_mask = 0;            // the atomics of workgroup w go to absmax[w & mask] (see absmax_bits_kernel); 0 = one slot
    __device__ __forceinline__ float oscale() const { return a_dinv ? out_scale * a_dinv[0] : out_scale; }
    float* C = nullptr;    // optional fp32 output [M][ldc]
    int ldc = 0;
    Planes Cp;             // optional plane-set output (the A operand of the next GEMM)
    // optional (row-major epilogue only): the residual given as a PLANE SET instead of ep.residual -- inference runs that keep an
    // edge-level tensor only in the format its consumers read (x = (h0 + h1) / scale: exact in fp32)
    Planes res_pl;
    // optional (row-major epilogue only) second merge behind the first:  y = ((act(z) + residual) * ep.out_scale + residual2) * out_scale2
    // (a skip connection folded into the last layer of the residual stack it closes); residual2 as fp32 rows or as a plane set
    const float* residual2 = nullptr;
    int ld_res2 = 0;
    Planes res2_pl;
    float out_scale2 = 1.f;
    const int* res2_rows = nullptr;   // optional row map of residual2 (fp32 form): row r adds residual2[res2_rows[r]]
    // optional (row-major epilogue only) multiplicand applied right after the activation:  act(z) * post_mul[row][col]
    const float* post_mul = nullptr;
    int ld_post_mul = 0;
    // The four extensions above (plane-set residual, second merge, row map, multiplicand) live in the EXT instantiation of the
    // kernel only, so that the launches that do not use them keep their register budget (gemm_planes picks the instantiation).
    __host__ __device__ bool extended() const { return res_pl.base || residual2 || res2_pl.base || post_mul; }
    // optional fused segmented sum over rows (edge -> node aggregation, cspnet.py:79): rows are edges sorted by
    // `seg_src`; every 32-row block writes the partial sum of each node run it contains to
    // seg_part[slot][node][col], slot = block - first block of that node (fixed order, no atomics);
    // finalize_agg_kernel adds the slots and divides by the degree.
    float* seg_part = nullptr;
    const int* seg_src = nullptr;
    const int* seg_rowptr = nullptr;
    int seg_nodes = 0;
    // optional PAIR mode (first edge GEMM over a symmetric edge list): rows are unordered node pairs (i, j); the first half of
    // K holds the sine features, the second half the cosine features of d = x_j - x_i.  The reversed edge sees 1 - d, i.e.
    // -sin and +cos, so ONE row of MFMA work yields both edges:  Z(i->j) = C + S + ...,  Z(j->i) = C - S + ...
    // (S, C = the two half-K sums).  pair_e1 / pair_e2 = output rows (edge ids) of the two directions; row-gathered addends:
    // ep.row_bias [pair_i] + ep.row_bias2 [pair_j] for i->j and the swapped pair for j->i, plus ep.row_bias3 [pair_graph].
    const int* pair_i = nullptr;
    const int* pair_j = nullptr;
    const int* pair_e1 = nullptr;
    const int* pair_e2 = nullptr;
    const int* pair_graph = nullptr;
    // optional, PAIR mode only: work folded into the same launch (two kernel boundaries less per layer).
    //  * the power-of-two scales of this layer's unbounded activation plane sets (see act_scales_eval): every workgroup derives
    //    the scale of the output plane set from sc_pq / sc_gmax / sc_wb itself, workgroup 0 publishes all of them in sc_dsc for
    //    the kernels that follow in the stream;
    //  * the SELF edges (i, i): d = 0, so the Fourier term is the constant diag_C0 -- extra workgroups from diag_block0 on,
    //    eight nodes each, write  M1[diag_e[i]] = SiLU(diag_C0 + P_i[i] + P_j[i] + G[graph])  (cspnet.py:59-79).
    const unsigned* sc_pq = nullptr;
    const unsigned* sc_gmax = nullptr;
    const float* sc_wb = nullptr;
    float* sc_dsc = nullptr;
    float* sc_dsc2 = nullptr;  // optional second copy (the training tape keeps each layer's scales for the backward pass)
    // pair mode: true (the safe default) = 64-bit address arithmetic in the epilogue; false = every gathered array, the pre-activation
    // array and the output plane set are addressed with 32-bit byte offsets off their (scalar) base -- the caller has checked the sizes
    // (pairs_fit_32bit).  In pair-mode kernels the EXT template flag selects the form (the extended row epilogue does not exist there).
    bool pair_wide = true;
    const float* diag_C0 = nullptr;
    const int* diag_node2graph = nullptr;
    const int* diag_e = nullptr;
    int diag_nodes = 0;
    int diag_block0 = 0;
    // optional DEVICE-side row count: the launch is sized by an upper bound M (a capacity), the kernel works on min(M, *m_dev) rows.
    // Lets a chain of launches whose row count is produced on the device (the periodic graph's edge count) run without a host
    // round trip per evaluation (gemnet.hip, the sampler's forwards).
    const int* m_dev = nullptr;
    int m_hint = 0;         // with m_dev: the row count the host expects (the last one it has seen) -- chooses the kernel form, never the rows processed
    __device__ __forceinline__ int rows(int M) const {
        if (!m_dev) return M;
        const int md = __builtin_amdgcn_readfirstlane(*m_dev);
        return md < M ? md : M;
    }
};

// Power-of-two scales of the three unbounded activation plane sets of a layer from RIGOROUS bounds (fp16 plane format):
//   |M1| <= |Z1| <= sum|Wff| + max|P_i| + max|P_j| + max|G|            (|Fourier features| <= 1, |silu(z)| <= |z|)
//   |agg| <= max|Z2| <= max|b2| + rowsum|W2| * bound(M1)
//   |X|   <= |pre|   <= max|b| + rowsum|W[:, H:]| * bound(agg) + max|X_part|
// (mpq = max over the whole [P_i | P_j | X_part] block).  scale = 2^floor(log2(16384 / bound)): the largest stored magnitude
// stays below 32768, elements down to bound * 2^-18 keep all 22 bits.  dsc = {scale, 1 / scale} x 3.
__device__ __forceinline__ void act_scales_eval(float mpq, float mg, const float* __restrict__ wb, float (&dsc)[6]) {
    float bound[3];
    bound[0] = wb[0] + 2.f * mpq + mg;
    bound[1] = wb[2] + wb[1] * bound[0];
    bound[2] = wb[4] + wb[3] * bound[1] + mpq;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float bnd = bound[c];
        int e = 14 - (int)ceilf(log2f(fmaxf(bnd, 1e-30f)));
        if (!(bnd == bnd) || bnd > 3e38f) e = -100;  // NaN / inf upstream: everything saturates, nothing overflows
        e = e > 14 ? 14 : (e < -100 ? -100 : e);
        dsc[2 * c] = exp2f((float)e);
        dsc[2 * c + 1] = exp2f(-(float)e);
    }
}

// Epilogue of one wave's TM x TN block of 32x32 accumulator tiles whose first row / column are row_w / col_w:
// bias, gathers, activation, optional fp32 / plane-set stores and the fused segmented row sum.
template <int TM, int TN>
__device__ __forceinline__ void planes_epilogue(const PlanesEpilogue& pe, f32x16 (&acc)[TM][TN], int row_w, int col_w, int M, int N, int l31,
                                                int kg) {
    const GemmEpilogue& ep = pe.ep;
    const float os = pe.oscale(), cps = pe.Cp.base ? pe.Cp.s() : 1.f;
    // segment structure of every row block of this wave, requested together (one chain of two dependent loads instead of one per block)
    int h_srcv[TM], h_rpv[TM], h_nvalid[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        h_srcv[i] = h_rpv[i] = h_nvalid[i] = 0;
        if (pe.seg_part) {
            const int rb = row_w + i * 32;
            h_nvalid[i] = M - rb < 32 ? (M - rb > 0 ? M - rb : 0) : 32;
            // (unconditional loads of clamped rows -- `nvalid > 0 ? table[..] : 0` compiles to a branch per block with its own s_waitcnt vmcnt(0):
            //  TM dependent latencies instead of the one chain of two this block is written for; an empty block's values are unused)
            int r = rb + (l31 < h_nvalid[i] ? l31 : h_nvalid[i] - 1);
            r = r < M ? r : M - 1;
            r = r > 0 ? r : 0;   // (M may be a device-side count of 0: the tables are capacity-sized)
            h_srcv[i] = pe.seg_src[r];
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
        if (pe.seg_part) h_rpv[i] = pe.seg_rowptr[h_srcv[i]];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rb = row_w + i * 32;  // first row of this 32-row block
        // segment structure of the block (runs of equal seg_src), wave-uniform
        uint32_t starts = 0;
        const int srcv = h_srcv[i], rpv = h_rpv[i], nvalid = h_nvalid[i];
        if (pe.seg_part) {
            // (rpv = the run's first edge, per lane and up front: looked up inside the segment loop it was one dependent global load per
            // segment and column tile -- a serial chain of ~12 load latencies, a quarter of this kernel's time)
            const int prev = __shfl_up(srcv, 1, 64);
            starts = (uint32_t)__ballot(kg == 0 && l31 < nvalid && (l31 == 0 || srcv != prev));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col_w + j * 32 + l31;
            const bool col_ok = col < N;
            const float bcol = (ep.bias && col_ok) ? ep.bias[col] : 0.f;
            float val[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb + (r & 3) + 8 * (r >> 2) + 4 * kg;
                float v = 0.f;
                if (row < M && col_ok) {
                    v = apply_epilogue<true>(ep, acc[i][j][r] * os + bcol, row, col);
                    if (pe.C) pe.C[(size_t)row * pe.ldc + col] = v;
                    if (pe.Cp.base) {
                        u16 p0, p1, p2;
                        pl_split(v, cps, p0, p1, p2);
                        pe.Cp.base[pe.Cp.elem(row, col, 0)] = p0;
                        pe.Cp.base[pe.Cp.elem(row, col, 1)] = p1;
                        if (NPL > 2) pe.Cp.base[pe.Cp.elem(row, col, 2)] = p2;
                    }
                }
                val[r] = v;
            }
            if (pe.seg_part) {
                uint32_t rem = starts;
                while (rem) {
                    const int sg = __builtin_ctz(rem);
                    rem &= rem - 1;
                    const int end = rem ? __builtin_ctz(rem) : nvalid;
                    const int node = __builtin_amdgcn_readlane(srcv, sg), rp = __builtin_amdgcn_readlane(rpv, sg);
                    // rows this lane holds: c + 4*kg for the 16 compile-time offsets c; in-range test as one unsigned compare
                    const unsigned lo = (unsigned)(sg - 4 * kg), span = (unsigned)(end - sg);
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned c = (unsigned)((r & 3) + 8 * (r >> 2));
                        sum += (c - lo < span) ? val[r] : 0.f;
                    }
                    sum += __shfl_xor(sum, 32, 64);
                    if (kg == 0 && col_ok) {
                        const int slot = (rb >> 5) - (rp >> 5);
                        pe.seg_part[((size_t)slot * pe.seg_nodes + node) * N + col] = sum;
                    }
                }
            }
        }
    }
}

// dst[0..7] += the eight consecutive elements (row, col .. col + 7) of a plane set (col % 8 == 0: one 16-byte load per plane)
__device__ __forceinline__ void pl_add8(float (&dst)[8], const Planes& P, int row, int col) {
    const float inv = P.dscale ? P.dscale[1] : 1.f / P.scale;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pl = NPL - 1; pl >= 0; --pl) {   // smallest plane first: the sum of the planes is exact in fp32
        const u32x4 w = *reinterpret_cast<const u32x4*>(P.base + P.elem(row, col, pl));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned wk = w[k];   // (a scalar copy: __builtin_bit_cast of the vector ELEMENT expression read element 0 every time)
#if MI_PLANES_FP16
            const f16x2 h = __builtin_bit_cast(f16x2, wk);
            acc[2 * k] += (float)h[0];
            acc[2 * k + 1] += (float)h[1];
#else
            acc[2 * k] += __uint_as_float(wk << 16);
            acc[2 * k + 1] += __uint_as_float(wk & 0xFFFF0000u);
#endif
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] += acc[k] * inv;
}

// Row-major variant of the epilogue for GEMMs that write a plane set (the first edge GEMM).  In the MFMA result layout a lane
// owns ONE column of 16 rows, so the three row-gathered addends cost 48 four-byte loads and the plane output 48 two-byte
// stores per 32x32 tile and lane.  Here each tile goes through a per-wave LDS patch (32 x 36 floats) and comes back as
// 8 consecutive columns of one row per lane: every gather, the optional pre-activation save and the three plane stores are
// 16-byte accesses (12 loads + 6..8 stores per tile and lane).  Needs N % 8 == 0; `stage` = this wave's 4608-byte patch.
template <int TM, int TN, bool EXT = false>
__device__ __forceinline__ void planes_epilogue_rows(const PlanesEpilogue& pe, f32x16 (&acc)[TM][TN], int row_w, int col_w, int M, int N,
                                                     int lane, float* stage) {
    const GemmEpilogue& ep = pe.ep;
    const float os = pe.oscale(), cps = pe.Cp.base ? pe.Cp.s() : 1.f;
    const int l31 = lane & 31, kg = lane >> 5;
    float amax = 0.f;
    unsigned sat = 0;
    // products that only write fp32 rows (the data gradients of the backward pass): the scaled tile goes through the patch once and leaves as
    // whole 128-byte lines -- none of the general path's per-element feature tests, second patch round trip or plane conversion
    if (pe.C && !pe.Cp.base && !ep.bias && !ep.row_bias && !ep.pre_add && !ep.pre_act && ep.act == ACT_NONE && !ep.residual && ep.out_scale == 1.f && !pe.extended()) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int rb = row_w + i * 32, cb = col_w + j * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * kg) * 36 + l31] = acc[i][j][r] * os;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int rl = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
                    const int row = rb + rl, col = cb + c4;
                    const f32x4 z = *reinterpret_cast<const f32x4*>(stage + rl * 36 + c4);
                    if (row < M && col < N) {
                        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(z[0]), fabsf(z[1]))), fmaxf(fabsf(z[2]), fabsf(z[3])));
                        *reinterpret_cast<f32x4*>(pe.C + (size_t)row * pe.ldc + col) = z;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        if (pe.absmax) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
            if (lane == 0) atomicMax(pe.absmax + (blockIdx.x & pe.absmax_mask), __float_as_uint(amax));
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int rb = row_w + i * 32, cb = col_w + j * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * kg) * 36 + l31] = acc[i][j][r] * os;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = lane + 64 * u, rl = q >> 2, c8 = (q & 3) * 8;
                const int row = rb + rl, col = cb + c8;
                const f32x4 z0 = *reinterpret_cast<const f32x4*>(stage + rl * 36 + c8);
                const f32x4 z1 = *reinterpret_cast<const f32x4*>(stage + rl * 36 + c8 + 4);
                if (row < M && col < N) {
                    float v[8] = {z0[0], z0[1], z0[2], z0[3], z1[0], z1[1], z1[2], z1[3]};
                    auto add8 = [&](float (&dst)[8], const float* src) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            dst[k] += a[k];
                            dst[4 + k] += b[k];
                        }
                    };
                    if (ep.bias) add8(v, ep.bias + col);
                    if (ep.row_bias) {
                        float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        add8(g, ep.row_bias + (size_t)ep.row_group[row] * ep.ld_row_bias + col);
                        if (ep.row_bias2) add8(g, ep.row_bias2 + (size_t)ep.row_group2[row] * ep.ld_row_bias2 + col);
                        if (ep.row_bias3) add8(g, ep.row_bias3 + (size_t)ep.row_group3[row] * ep.ld_row_bias3 + col);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += g[k];
                    }
                    if (ep.pre_add) add8(v, ep.pre_add + (size_t)row * ep.ld_pre_add + col);
                    if (ep.pre_act) {
                        float* d = ep.pre_act + (size_t)row * ep.ld_pre + col;
                        *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    }
                    if (ep.act == ACT_SILU) {  // hardware exp2/rcp form (~3 ulp): E*H activations per launch, not hidden behind MFMAs
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = silu_fast(v[k]);
                    } else if (ep.act == ACT_SSILU) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = silu_fast(v[k]) * 1.66666666666666667f;
                    }
                    if constexpr (EXT) {
                        if (pe.post_mul) {
                            const float* src = pe.post_mul + (size_t)row * pe.ld_post_mul + col;
                            const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                v[k] *= a[k];
                                v[4 + k] *= b[k];
                            }
                        }
                    }
                    if (ep.residual) add8(v, ep.residual + (size_t)row * ep.ld_res + col);
                    else if constexpr (EXT) {
                        if (pe.res_pl.base) pl_add8(v, pe.res_pl, row, col);
                    }
                    if (ep.out_scale != 1.f) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] *= ep.out_scale;
                    }
                    if constexpr (EXT) {
                        if (pe.residual2 || pe.res2_pl.base) {
                            if (pe.residual2) add8(v, pe.residual2 + (size_t)(pe.res2_rows ? pe.res2_rows[row] : row) * pe.ld_res2 + col);
                            else pl_add8(v, pe.res2_pl, row, col);
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] *= pe.out_scale2;
                        }
                    }
                    if (pe.absmax) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(v[k]));
                    }
                    if (pe.C) {   // back into the patch (this lane's own eight slots); stored below in whole 128-byte lines
                        *reinterpret_cast<f32x4*>(stage + rl * 36 + c8) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(stage + rl * 36 + c8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    }
                    if (pe.Cp.base) {
                        u32x4 o[3];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            unsigned pr[3];
                            pl_split_pair_acc(v[2 * k], v[2 * k + 1], cps, pr, sat);
                            o[0][k] = pr[0];
                            o[1][k] = pr[1];
                            o[2][k] = pr[2];
                        }
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x4*>(pe.Cp.base + pe.Cp.elem(row, col, pl)) = o[pl];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (pe.C) {
                // fp32 rows: eight lanes write one 128-byte line (a row of the 32 x 32 tile), a wave instruction eight whole lines.
                // In the eight-columns-per-lane mapping above the same data left as 2 x 64 sixteen-byte pieces spread over 16 rows,
                // and a K = 16 layer that only writes its [E, 512] result ran at 1.3 TB/s (393 us at 256k edges).
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int rl = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
                    const int row = rb + rl, col = cb + c4;
                    const f32x4 z = *reinterpret_cast<const f32x4*>(stage + rl * 36 + c4);
                    if (row < M && col < N) *reinterpret_cast<f32x4*>(pe.C + (size_t)row * pe.ldc + col) = z;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    sat_report(sat);
    if (pe.absmax) {  // max |output| of this wave's block: order-independent, so the atomic keeps the result deterministic
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0) atomicMax(pe.absmax + (blockIdx.x & pe.absmax_mask), __float_as_uint(amax));
    }
}
// LEAN row epilogue of the register-tile GEMM (gemm_rt_kernel<EXT, true>, edge_stage.hip): the dense layers of an inference forward that only
// write the plane set their consumer reads -- y = ((act(z + G1[g1[row]] + G2[g2[row]]) * post_mul + res) * scale + res2[rows2[row]]) * scale2 with
// res / res2 given as plane sets or fp32 rows, the exact max |y| recorded, nothing else (no column bias, third gather, pre-activation or
// fp32 output rows: those launches keep planes_epilogue_rows).
// The kernel multiplies with its operands SWAPPED (acc = W A^T tile: a lane holds ONE ROW of the output and the sixteen columns
// (r & 3) + 8 (r >> 2) + 4 kg), and eight v_permlane32_swap per 32 x 32 tile exchange column groups between the two lanes of a row, after which
// lane (l31, kg) owns columns 16 kg .. 16 kg + 15 of row l31: the transposition that planes_epilogue_rows does through an LDS patch
// (16 ds_write_b32 + 4 ds_read_b128 + two wave barriers per tile, and the tiles serialised behind them) costs eight VALU slots and no wait.
// Every scale folds into two constants: with I = 1 / (os ga post) the scaled activation is a rcp(fma(e, I, I)) (silu_fast_scaled's identity;
// os = accumulator scale, ga = 1 / 0.6 for ScaledSiLU, post = scale scale2 cps) straight from the raw accumulator a, and a plane-set residual
// enters as two v_fma_mix_f32 per element.  Addresses: the row tile is the workgroup's, so every plane access is a scalar base + one per-lane
// offset computed once.  Measured instruction count per element: ~10 against ~25 in the general epilogue, which under a partner wave's MFMAs
// issue at ~16 cycles each (scripts/rt_phases.py: 29 k cycles of epilogue per workgroup, 60 k with the extensions).
// (FULL = false: without the gathered addends and the fp32 residual rows -- the persistent ablation kernel's budget, see planes_epilogue_is_lean)
template <int TM, int TN, bool FULL = true>
__device__ __forceinline__ void planes_epilogue_lean(const PlanesEpilogue& pe, f32x16 (&acc)[TM][TN], int tile, int col_w, int M, int lane) {
#if MI_PLANES_FP16
    const GemmEpilogue& ep = pe.ep;
    const int l31 = lane & 31, kg = lane >> 5;
    const float os = pe.oscale(), cps = pe.Cp.s();
    const bool has_res = pe.res_pl.base != nullptr, has_res2 = pe.res2_pl.base != nullptr, has_pm = pe.post_mul != nullptr;
    const bool has_g = FULL && ep.row_bias != nullptr, has_g2 = FULL && ep.row_bias2 != nullptr, has_rf = FULL && ep.residual != nullptr,
               has_rf2 = FULL && pe.residual2 != nullptr;
    const float s1 = ep.out_scale, s2 = (has_res2 || has_rf2) ? pe.out_scale2 : 1.f;
    const float post = s1 * s2 * cps;
    const float ga = ep.act == ACT_SSILU ? 1.66666666666666667f : 1.f;
    // activation / no activation; with gathered addends the pre-activation is formed first (x = a os + g) and os leaves the constants
    const float osx = has_g ? 1.f : os;
    const float I = 1.0f / (osx * ga * post), lin = osx * post;
    const float R1 = has_res ? (pe.res_pl.dscale ? pe.res_pl.dscale[1] : 1.f / pe.res_pl.scale) * post : 0.f;
    const float R2 = has_res2 ? (pe.res2_pl.dscale ? pe.res2_pl.dscale[1] : 1.f / pe.res2_pl.scale) * (s2 * cps) : 0.f;
    float amax = 0.f;
    unsigned sat = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = tile * 128 + i * 32 + l31;
        const bool row_ok = row < M;
        const unsigned loff = (unsigned)((i * 32 + l31) * 32 + kg * 16);   // element offset of this lane's sixteen columns inside a (row tile, column tile, plane) block
        int g1 = 0, g2 = 0, r2row = row;   // this row's gather / row-map indices: one load each per row block
        if (row_ok) {
            if (has_g) g1 = ep.row_group[row];
            if (has_g2) g2 = ep.row_group2[row];
            if (has_rf2 && pe.res2_rows) r2row = pe.res2_rows[row];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ct = (col_w >> 5) + j;
            // column groups X_g = acc[4 g .. 4 g + 3]: swap (X0, X2) and (X1, X3) between the halves of the wave -> (X0, X2, X1, X3) = sixteen consecutive columns
            float x[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const auto a02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][k]), __float_as_uint(acc[i][j][8 + k]), false, false);
                const auto a13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][4 + k]), __float_as_uint(acc[i][j][12 + k]), false, false);
                x[k] = __uint_as_float(a02[0]);
                x[4 + k] = __uint_as_float(a02[1]);
                x[8 + k] = __uint_as_float(a13[0]);
                x[12 + k] = __uint_as_float(a13[1]);
            }
            if (row_ok) {
                u32x4 rh[2][2], qh[2][2];   // residual planes [plane][half of the sixteen columns]
                if (has_res) {
                    const u16* rb = pe.res_pl.base + pe.res_pl.tile(tile, ct) + loff;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int h = 0; h < 2; ++h) rh[pl][h] = *reinterpret_cast<const u32x4*>(rb + pl * 4096 + h * 8);
                }
                if (has_res2) {
                    const u16* qb = pe.res2_pl.base + pe.res2_pl.tile(tile, ct) + loff;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int h = 0; h < 2; ++h) qh[pl][h] = *reinterpret_cast<const u32x4*>(qb + pl * 4096 + h * 8);
                }
                f32x4 pm[4];
                if (has_pm) {
                    const float* pb = pe.post_mul + (size_t)row * pe.ld_post_mul + ct * 32 + kg * 16;
#pragma unroll
                    for (int q = 0; q < 4; ++q) pm[q] = *reinterpret_cast<const f32x4*>(pb + 4 * q);
                }
                if (has_g) {   // x = a os + G1[g1] (+ G2[g2])
                    const float* gp = ep.row_bias + (size_t)g1 * ep.ld_row_bias + ct * 32 + kg * 16;
                    f32x4 ga4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) ga4[q] = *reinterpret_cast<const f32x4*>(gp + 4 * q);
                    if (has_g2) {
                        const float* gq = ep.row_bias2 + (size_t)g2 * ep.ld_row_bias2 + ct * 32 + kg * 16;
#pragma unroll
                        for (int q = 0; q < 4; ++q) ga4[q] += *reinterpret_cast<const f32x4*>(gq + 4 * q);
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) x[k] = __builtin_fmaf(x[k], os, ga4[k >> 2][k & 3]);
                }
                float v[16];
                if (ep.act == ACT_NONE) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] = x[k] * lin;
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        v[k] = x[k] * __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(x[k] * (osx * -1.44269504088896340736f)), I, I));
                }
                if (has_pm) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] *= pm[k >> 2][k & 3];
                }
                // v += (float)half * R for the two planes of a residual (smallest plane first): v_fma_mix_f32 reads the half in place
                auto add_planes = [&](const u32x4 (&w)[2][2], float R) {
#pragma unroll
                    for (int pl = 1; pl >= 0; --pl)
#pragma unroll
                        for (int k = 0; k < 16; k += 2) {
                            const unsigned word = w[pl][k >> 3][(k >> 1) & 3];
                            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(v[k]) : "v"(word), "v"(R));
                            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v[k + 1]) : "v"(word), "v"(R));
                        }
                };
                if (has_res) add_planes(rh, R1);
                if (has_rf) {   // fp32 residual rows: v += r post
                    const float* rp = ep.residual + (size_t)row * ep.ld_res + ct * 32 + kg * 16;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 r = *reinterpret_cast<const f32x4*>(rp + 4 * q);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[4 * q + k] = __builtin_fmaf(r[k], post, v[4 * q + k]);
                    }
                }
                if (has_res2) add_planes(qh, R2);
                if (has_rf2) {   // fp32 second merge, optionally through its row map: v += r2 (scale2 cps)
                    const float* rp = pe.residual2 + (size_t)r2row * pe.ld_res2 + ct * 32 + kg * 16;
                    const float c2 = s2 * cps;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 r = *reinterpret_cast<const f32x4*>(rp + 4 * q);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[4 * q + k] = __builtin_fmaf(r[k], c2, v[4 * q + k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < 16; k += 2) amax = fmaxf(amax, fmaxf(fabsf(v[k]), fabsf(v[k + 1])));
                u32x4 o[2][2];
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    unsigned pr[3];
                    pl_split_scaled_pair_acc(v[k], v[k + 1], pr, sat);
                    o[0][k >> 3][(k >> 1) & 3] = pr[0];
                    o[1][k >> 3][(k >> 1) & 3] = pr[1];
                }
                u16* ob = pe.Cp.base + pe.Cp.tile(tile, ct) + loff;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int h = 0; h < 2; ++h) *reinterpret_cast<u32x4*>(ob + pl * 4096 + h * 8) = o[pl][h];
            }
        }
    }
    sat_report(sat);
    if (pe.absmax) {  // max |y| of this wave's block (v is y cps: cps is a power of two)
        amax *= 1.0f / cps;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0) atomicMax(pe.absmax + (blockIdx.x & pe.absmax_mask), __float_as_uint(amax));
    }
#endif
}
// whether a launch's epilogue is the lean one's (host side, gemm_rt); full = false: its form without gathers and fp32 residual rows
inline bool planes_epilogue_is_lean(const PlanesEpilogue& pe, bool full = true) {
    const GemmEpilogue& ep = pe.ep;
    if (!full && (ep.row_bias || ep.row_bias2 || ep.residual || pe.residual2)) return false;
    return MI_PLANES_FP16 && pe.Cp.base && !pe.C && !ep.bias && (ep.row_bias || !ep.row_bias2) && !ep.row_bias3 && !ep.pre_add && !ep.pre_act &&
           !(ep.residual && pe.res_pl.base) && !(pe.residual2 && pe.res2_pl.base) && (pe.residual2 || !pe.res2_rows) && !pe.seg_part && !pe.pair_i &&
           (ep.ld_row_bias & 3) == 0 && (ep.ld_row_bias2 & 3) == 0 && (ep.ld_res & 3) == 0 && (pe.ld_res2 & 3) == 0 && (pe.ld_post_mul & 3) == 0 &&
           (ep.act == ACT_NONE || ep.act == ACT_SILU || ep.act == ACT_SSILU);
}

// PAIR-mode epilogue (see PlanesEpilogue): accS / accC = the sine-half and cosine-half sums of one wave's tiles.  Row-major
// through two per-wave LDS patches; each lane handles 8 consecutive columns of one pair and emits BOTH directed edges.
// Instruction budget (this epilogue is 40 % of the kernel's issue slots and does not overlap other waves' MFMAs): per output element
// one add / subtract of the raw accumulators and one FMA with the (power-of-two, hence exact) output scale; two adds for the gathered
// addends; SiLU with the destination plane scale folded into the reciprocal's argument (silu_fast_scaled); the plane split at four
// operations (pl_split_scaled_pair_acc).  Addresses: WIDE = false takes every gather / store offset in 32 bits off a scalar base
// (v_mad_u32_u24 + the saddr form of the access: one operation per address where 64-bit pointer arithmetic took three to twelve);
// the host picks WIDE = true when an operand is too large for that (pairs_fit_32bit).
template <int TM, int TN, bool WIDE = false>
__device__ __forceinline__ void planes_epilogue_pairs(const PlanesEpilogue& pe, f32x16 (&accS)[TM][TN], f32x16 (&accC)[TM][TN], int row_w,
                                                      int col_w, int M, int N, int lane, float* stage, float cps_in = 0.f) {
    const GemmEpilogue& ep = pe.ep;
    const float os = pe.oscale(), cps = cps_in != 0.f ? cps_in : pe.Cp.base ? pe.Cp.s() : 1.f;
    const float inv_cps = 1.0f / cps;   // (a power of two: exact)
    const int l31 = lane & 31, kg = lane >> 5;
    float* stS = stage;
    float* stC = stage + 1152;
    unsigned sat = 0;   // saturation flag, accumulated: a branch with an atomic per conversion kept the gathers of the next piece from being scheduled under this one's arithmetic
    // the index rows of this lane's pairs (they depend on the row block and the half only, not on the column tile), all requested up
    // front: inside the tile loop each tile paid an index-load latency ahead of its gathers -- eight times per wave in a launch of one
    // workgroup per CU, where nothing else covers it (a third of a 33 us product at 265 edges)
    int h_ni[TM][2], h_nj[TM][2], h_gr[TM][2], h_e1[TM][2], h_e2[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // (UNCONDITIONAL loads of a clamped row: `ok ? table[row] : 0` compiled to one exec-masked branch per (i, u) with its own
            //  s_waitcnt vmcnt(0) inside -- eight dependent memory latencies at the head of every epilogue, 10-15 k cycles of the 33 k
            //  (found in the ISA); rows past M are masked where they are used)
            const int row = row_w + i * 32 + ((lane + 64 * u) >> 2);
            const int rc = row < M ? row : (M > 0 ? M - 1 : 0);
            h_ni[i][u] = pe.pair_i[rc];
            h_nj[i][u] = pe.pair_j[rc];
            h_gr[i][u] = pe.pair_graph[rc];
            h_e1[i][u] = pe.pair_e1[rc];
            h_e2[i][u] = pe.pair_e2[rc];
        }
    // row r, column c of a row-major fp32 array / of the output plane set, as a pointer
#ifdef MI_AB_PAIRS_R3   // (A/B builds: round 3's addressing and arithmetic)
    constexpr bool W64 = true;
#else
    constexpr bool W64 = WIDE;
#endif
    auto frow = [&](const float* base, int r, int ld, int c) -> const float* {
        if constexpr (W64) return base + (size_t)r * ld + c;
        else return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (__umul24((unsigned)r, (unsigned)ld * 4u) + (unsigned)c * 4u));
    };
    const unsigned tile_stride_b = ((unsigned)pe.Cp.KT * 12288u + 2048u) * 2u;
    auto prow = [&](int r, int c, int pl) -> u16* {
        if constexpr (W64) return pe.Cp.base + pe.Cp.elem(r, c, pl);
        else return reinterpret_cast<u16*>(reinterpret_cast<char*>(pe.Cp.base + pl * 4096) +
                                           (__umul24((unsigned)r >> 7, tile_stride_b) + (((unsigned)r & 127u) << 6) + ((unsigned)(c >> 5) * 24576u + ((unsigned)c & 31u) * 2u)));
    };
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int rb = row_w + i * 32, cb = col_w + j * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {   // (raw sums: the output scale is applied by the FMA that adds the gathered terms -- exact, it is a power of two)
                const int o = ((r & 3) + 8 * (r >> 2) + 4 * kg) * 36 + l31;
                stS[o] = accS[i][j][r];
                stC[o] = accC[i][j][r];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = lane + 64 * u, rl = q >> 2, c8 = (q & 3) * 8;
                const int row = rb + rl, col = cb + c8;
                if (row < M && col < N) {
                    float sv[8], cv[8];
                    {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(stS + rl * 36 + c8), b = *reinterpret_cast<const f32x4*>(stS + rl * 36 + c8 + 4);
                        const f32x4 c = *reinterpret_cast<const f32x4*>(stC + rl * 36 + c8), d = *reinterpret_cast<const f32x4*>(stC + rl * 36 + c8 + 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            sv[k] = a[k];
                            sv[4 + k] = b[k];
                            cv[k] = c[k];
                            cv[4 + k] = d[k];
                        }
                    }
                    const int ni = h_ni[i][u], nj = h_nj[i][u], gr = h_gr[i][u];
                    auto ld8 = [&](float (&dst)[8], const float* src) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            dst[k] = a[k];
                            dst[4 + k] = b[k];
                        }
                    };
                    float pii[8], pjj[8], pij[8], pji[8], gg[8];   // (no column bias in pair mode: the caller folds it into the per-crystal addend -- a run-time test of it cost a select per element)
#if defined(MI_DBG_PAIRS_SKIP) && (MI_DBG_PAIRS_SKIP & 2)   // timing diagnostic (wrong results): no row gathers
#pragma unroll
                    for (int k = 0; k < 8; ++k) pii[k] = pjj[k] = pij[k] = pji[k] = gg[k] = 0.25f * (float)(ni + nj + gr);
#else
                    ld8(pii, frow(ep.row_bias, ni, ep.ld_row_bias, col));     // P_i[i]
                    ld8(pjj, frow(ep.row_bias2, nj, ep.ld_row_bias2, col));   // P_j[j]
                    ld8(pij, frow(ep.row_bias, nj, ep.ld_row_bias, col));     // P_i[j]
                    ld8(pji, frow(ep.row_bias2, ni, ep.ld_row_bias2, col));   // P_j[i]
                    ld8(gg, frow(ep.row_bias3, gr, ep.ld_row_bias3, col));
#endif
#pragma unroll
                    for (int dir = 0; dir < 2; ++dir) {
#if defined(MI_DBG_PAIRS_SKIP) && (MI_DBG_PAIRS_SKIP & 16)  // (timing diagnostic, wrong results: every store lands in the first 1024 rows -- the same instructions, no HBM-side traffic)
                        const int erow = (dir == 0 ? h_e1[i][u] : h_e2[i][u]) & 1023;
#elif defined(MI_DBG_PAIRS_SKIP) && (MI_DBG_PAIRS_SKIP & 8)   // (timing diagnostic, wrong results: both directions of a pair in adjacent rows 2 p, 2 p + 1)
                        const int erow = 2 * row + dir;
#else
                        const int erow = dir == 0 ? h_e1[i][u] : h_e2[i][u];
#endif
                        float v[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float acc = dir == 0 ? cv[k] + sv[k] : cv[k] - sv[k];
                            const float g = dir == 0 ? (pii[k] + pjj[k]) + gg[k] : (pij[k] + pji[k]) + gg[k];
#ifdef MI_AB_PAIRS_R3
                            v[k] = acc * os + g;
#else
                            v[k] = __builtin_fmaf(acc, os, g);   // = (acc os) + g: acc os is exact
#endif
                        }
                        if (ep.pre_act) {
                            float* d = const_cast<float*>(frow(ep.pre_act, erow, ep.ld_pre, col));
                            *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
                        }
#if !(defined(MI_DBG_PAIRS_SKIP) && (MI_DBG_PAIRS_SKIP & 4))   // (timing diagnostic, wrong results: 4 = no SiLU)
                        if (ep.act == ACT_SILU) {
#pragma unroll
#ifdef MI_AB_PAIRS_R3
                            for (int k = 0; k < 8; ++k) v[k] = silu_fast(v[k]) * cps;
#else
                            for (int k = 0; k < 8; ++k) v[k] = silu_fast_scaled(v[k], inv_cps);
#endif
                        } else
#endif
                        {
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] *= cps;
                        }
                        u32x4 o[3];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            unsigned pr[3];
                            pl_split_scaled_pair_acc(v[2 * k], v[2 * k + 1], pr, sat);
                            o[0][k] = pr[0];
                            o[1][k] = pr[1];
                            o[2][k] = pr[2];
                        }
#if defined(MI_DBG_PAIRS_SKIP) && (MI_DBG_PAIRS_SKIP & 1)   // (timing diagnostic, wrong results: 1 = one plane store per tile and lane instead of all)
                        if ((o[0][0] ^ o[1][1]) == 0x12345677u)
#endif
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) {
                            // (non-temporal stores measured equal: 44.4 / 48.9-50.6 against 44.6 / 49.8-50.6 structures/s on one / four chains)
                            *reinterpret_cast<u32x4*>(prow(erow, col, pl)) = o[pl];
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    sat_report(sat);
}

// ONE direction of the pair-mode epilogue (edge_gemm1e_kernel, edge_stage.hip): `acc` = the whole Fourier term of direction `dir` of each
// pair (0: i -> j = C + S, 1: j -> i = C - S, both accumulated on the matrix pipe); the gathers, activation and plane stores of
// planes_epilogue_pairs for that direction only.  The accumulators are read, not consumed: the caller goes on accumulating into them.
template <int TM, int TN>
__device__ __forceinline__ void planes_epilogue_pairs_dir(const PlanesEpilogue& pe, const f32x16 (&acc)[TM][TN], int dir, int row_w, int col_w, int M, int N,
                                                          int lane, float* stage, float cps_in) {
    const GemmEpilogue& ep = pe.ep;
    const float os = pe.oscale(), cps = cps_in != 0.f ? cps_in : pe.Cp.base ? pe.Cp.s() : 1.f;
    const int l31 = lane & 31, kg = lane >> 5;
    unsigned sat = 0;
    int h_a[TM][2], h_b[TM][2], h_gr[TM][2], h_e[TM][2];   // dir 0: P_i[i] + P_j[j] -> row e1;  dir 1: P_i[j] + P_j[i] -> row e2
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = row_w + i * 32 + ((lane + 64 * u) >> 2);
            const bool ok = row < M;
            const int ni = ok ? pe.pair_i[row] : 0, nj = ok ? pe.pair_j[row] : 0;
            h_a[i][u] = dir == 0 ? ni : nj;
            h_b[i][u] = dir == 0 ? nj : ni;
            h_gr[i][u] = ok ? pe.pair_graph[row] : 0;
            h_e[i][u] = ok ? (dir == 0 ? pe.pair_e1[row] : pe.pair_e2[row]) : 0;
        }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int rb = row_w + i * 32, cb = col_w + j * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * kg) * 36 + l31] = acc[i][j][r] * os;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = lane + 64 * u, rl = q >> 2, c8 = (q & 3) * 8;
                const int row = rb + rl, col = cb + c8;
                if (row < M && col < N) {
                    const f32x4 z0 = *reinterpret_cast<const f32x4*>(stage + rl * 36 + c8), z1 = *reinterpret_cast<const f32x4*>(stage + rl * 36 + c8 + 4);
                    float v[8] = {z0[0], z0[1], z0[2], z0[3], z1[0], z1[1], z1[2], z1[3]};
                    auto ld8 = [&](float (&dst)[8], const float* src) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            dst[k] = a[k];
                            dst[4 + k] = b[k];
                        }
                    };
                    float pa[8], pb[8], gg[8];
                    ld8(pa, ep.row_bias + (size_t)h_a[i][u] * ep.ld_row_bias + col);
                    ld8(pb, ep.row_bias2 + (size_t)h_b[i][u] * ep.ld_row_bias2 + col);
                    ld8(gg, ep.row_bias3 + (size_t)h_gr[i][u] * ep.ld_row_bias3 + col);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = v[k] + ((pa[k] + pb[k]) + gg[k]);
                    const int erow = h_e[i][u];
                    if (ep.pre_act) {
                        float* d = ep.pre_act + (size_t)erow * ep.ld_pre + col;
                        *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    }
                    if (ep.act == ACT_SILU) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = silu_fast(v[k]);
                    }
                    u32x4 o[3];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        unsigned pr[3];
                        pl_split_pair_acc(v[2 * k], v[2 * k + 1], cps, pr, sat);
                        o[0][k] = pr[0];
                        o[1][k] = pr[1];
                        o[2][k] = pr[2];
                    }
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x4*>(pe.Cp.base + pe.Cp.elem(erow, col, pl)) = o[pl];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    sat_report(sat);
}

// Pre-activation save of the result-layout epilogue (training forward of the second edge linear: Z2 = acc + bias is kept for the
// backward pass while the activation and the fused segmented sum go on in registers).  Through the per-wave LDS patch so that the
// stores are 16-byte row-major ones instead of 16 four-byte column stores per tile and lane.  No row-gathered addends here.
template <int TM, int TN>
__device__ __forceinline__ void planes_store_preact_rows(const PlanesEpilogue& pe, f32x16 (&acc)[TM][TN], int row_w, int col_w, int M, int N,
                                                         int lane, float* stage) {
    const GemmEpilogue& ep = pe.ep;
    const float os = pe.oscale(), cps = pe.Cp.base ? pe.Cp.s() : 1.f;
    const int l31 = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int rb = row_w + i * 32, cb = col_w + j * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * kg) * 36 + l31] = acc[i][j][r] * os;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = lane + 64 * u, rl = q >> 2, c8 = (q & 3) * 8;
                const int row = rb + rl, col = cb + c8;
                f32x4 z0 = *reinterpret_cast<const f32x4*>(stage + rl * 36 + c8);
                f32x4 z1 = *reinterpret_cast<const f32x4*>(stage + rl * 36 + c8 + 4);
                if (row < M && col < N) {
                    if (ep.bias) {
                        z0 += *reinterpret_cast<const f32x4*>(ep.bias + col);
                        z1 += *reinterpret_cast<const f32x4*>(ep.bias + col + 4);
                    }
                    float* d = ep.pre_act + (size_t)row * ep.ld_pre + col;
                    *reinterpret_cast<f32x4*>(d) = z0;
                    *reinterpret_cast<f32x4*>(d + 4) = z1;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
}
__device__ __forceinline__ bool planes_preact_rows_applies(const PlanesEpilogue& pe, int N) {
    return pe.seg_part != nullptr && pe.ep.pre_act != nullptr && pe.ep.row_bias == nullptr && pe.ep.pre_add == nullptr && (N & 7) == 0 &&
           (pe.ep.ld_pre & 3) == 0;
}

// whether the row-major epilogue applies (otherwise the result-layout one, which also carries the fused segmented sum)
__host__ __device__ __forceinline__ bool planes_epilogue_is_rows(const PlanesEpilogue& pe, int N) {
    return (pe.Cp.base != nullptr || pe.C != nullptr) && pe.seg_part == nullptr && (N & 7) == 0 && (pe.ldc & 3) == 0 && (pe.ep.ld_res & 3) == 0 && (pe.ld_res2 & 3) == 0 &&
           (pe.ep.ld_pre_add & 3) == 0 &&
           (pe.ep.ld_pre & 3) == 0 && (pe.ep.ld_row_bias & 3) == 0 && (pe.ep.ld_row_bias2 & 3) == 0 && (pe.ep.ld_row_bias3 & 3) == 0;
}

// C = epi(A W^T) with both operands given as tile-blocked plane sets; main loop = loads + ds + MFMA only.
//
// LDS image per plane: 128 rows x 64 B, unpadded, 16-byte chunk index XOR-swizzled with (row >> 2) & 3:
//   chunk c of row r lives at r*64 + ((c ^ ((r >> 2) & 3)) * 16).
// Fragment reads (ds_read_b128, 16-lane groups of rows distinct mod 16, same chunk) and staging writes
// (ds_write_b128, 8 consecutive lanes = 2 rows x 4 chunks) are both conflict-free; a padded-row layout
// was 2-way on the writes (SQ_LDS_BANK_CONFLICT = 33 % of LDS cycles).
// Global -> register prefetch runs one k-tile ahead (issued right after the staging barrier).
// This is the 128x128-tile, two-barriers-per-k-step structure.  V = 0: plain products -- 135 registers, so THREE workgroups
// share a CU (3 x 48 KiB LDS) and cover each other's staging, barriers and epilogues; in situ this beats the 256-row
// double-buffered kernel below (one workgroup per CU), which is kept as an option.  V = 1: PAIR mode (see PlanesEpilogue) -- a
// second accumulator set, 196 registers, two workgroups per CU.
// PF > 1: the LATENCY form for launches of at most one round of workgroups (short edge lists, node-level products of small batches): PF
// register sets, the loads of k-tiles kt + 1 .. kt + PF in flight while k-tile kt is multiplied.  With one workgroup per CU nothing else
// covers the load -> stage -> barrier chain, and a k-tile then costs one memory latency (~1.5 us: a K = 768 product over 265 edges took
// 38 us, as long as over 7.5k edges).  Same tiles, same LDS image, same accumulation order: results are bit-identical to PF = 1.
template <int V, int TM, bool EXT = false, int PF = 1>
__device__ __forceinline__ void gemm_planes_body(Planes A, Planes W, int M, int N, int K, const PlanesEpilogue& pe, int rt_base) {
    constexpr int BM = 64 * TM, BN = 128, BK = 32, TN = 2, PLA = BM * 64, PLB = 128 * 64;  // bytes per plane tile in LDS (A, W)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    M = pe.rows(M);
    unsigned char* As = smem;
    unsigned char* Ws = smem + NPL * PLA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kg = lane >> 5;
    // XCD-aware tile mapping (workgroups go round-robin to the 8 XCDs, each with its own L2): all column tiles of one row tile run
    // on the same XCD, so the streamed A operand crosses the fabric once (measured on the pair-mode GEMM: FETCH_SIZE 4x the
    // operand size without it)
    const int nct_ = (N + BN - 1) / BN, id_ = blockIdx.x, slot_ = id_ >> 3;
    float cps_local = 0.f;  // PAIR mode with folded-in scales: this layer's M1 scale, derived here instead of read from Cp.dscale
    if constexpr (V == 1) {
        if (pe.sc_pq) {
            float dsc[6];
            act_scales_eval(__uint_as_float(pe.sc_pq[0]), __uint_as_float(pe.sc_gmax[0]), pe.sc_wb, dsc);
            cps_local = dsc[0];
            if (id_ == 0 && tid < 6) {
                pe.sc_dsc[tid] = dsc[tid];
                if (pe.sc_dsc2) pe.sc_dsc2[tid] = dsc[tid];
            }
        }
        if (pe.diag_C0 && id_ >= pe.diag_block0) {  // self edges: eight nodes per workgroup, a thread per column pair
            const float cps = cps_local != 0.f ? cps_local : pe.Cp.s();
            const int n0 = (id_ - pe.diag_block0) * 8, n1 = n0 + 8 < pe.diag_nodes ? n0 + 8 : pe.diag_nodes;
            const float* PQ = pe.ep.row_bias;
            const int ldpq = pe.ep.ld_row_bias;
            for (int f = 2 * tid; f < N; f += 512) {
                const float c0 = pe.diag_C0[f], c1 = pe.diag_C0[f + 1];
                for (int i = n0; i < n1; ++i) {
                    const int e = pe.diag_e[i], g = pe.diag_node2graph[i];
                    const float* G = pe.ep.row_bias3 + (size_t)g * pe.ep.ld_row_bias3;
                    const float v0 = c0 + ((PQ[(size_t)i * ldpq + f] + PQ[(size_t)i * ldpq + N + f]) + G[f]);
                    const float v1 = c1 + ((PQ[(size_t)i * ldpq + f + 1] + PQ[(size_t)i * ldpq + N + f + 1]) + G[f + 1]);
                    if (pe.ep.pre_act) {
                        pe.ep.pre_act[(size_t)e * pe.ep.ld_pre + f] = v0;
                        pe.ep.pre_act[(size_t)e * pe.ep.ld_pre + f + 1] = v1;
                    }
                    unsigned p[3];
                    pl_split_pair(silu_fast(v0), silu_fast(v1), cps, p);
#pragma unroll
                    for (int k = 0; k < NPL; ++k) *reinterpret_cast<unsigned*>(pe.Cp.base + pe.Cp.elem(e, f, k)) = p[k];
                }
            }
            return;
        }
    }
    const int ct = slot_ % nct_, rt = rt_base + (slot_ / nct_) * 8 + (id_ & 7);
    if (rt * BM >= M) return;
    const int row0 = rt * BM, col0 = ct * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // a plane tile = 128 rows x 64 B = 512 chunks of 16 B; two per thread per plane.  Buffer loads: the descriptor covers
    // this block's row tile (uniform), the per-thread offset is constant and every tile/plane offset is scalar, so the
    // loop carries no per-load vector address arithmetic.
    u32x4 ra0[3][TM], rw0[3][2];
    const int KT = (K + 31) / 32;
    // (TM = 1: 64-row tiles, i.e. one half of a 128-row plane tile -- 4 KiB into each plane)
    const int ahalf = TM == 1 ? (rt & 1) * 4096 : 0;
    const __amdgpu_buffer_rsrc_t rsa = uniform_rsrc(A.base + A.tile(TM == 1 ? rt >> 1 : rt, 0), KT * 24576);
    const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(W.base + W.tile(ct, 0), KT * 24576);
    auto load_tiles = [&](int kt, u32x4 (&ra)[3][TM], u32x4 (&rw)[3][2]) {
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                if (v < TM) ra[p][v] = __builtin_amdgcn_raw_buffer_load_b128(rsa, tid * 16, kt * 24576 + p * 8192 + v * 4096 + ahalf, 0);
                rw[p][v] = __builtin_amdgcn_raw_buffer_load_b128(rsw, tid * 16, kt * 24576 + p * 8192 + v * 4096, 0);
            }
    };
    auto store_tiles = [&](const u32x4 (&ra)[3][TM], const u32x4 (&rw)[3][2]) {
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int f = v * 256 + tid, r = f >> 2, c = (f & 3) ^ ((r >> 2) & 3);
                if (v < TM) *reinterpret_cast<u32x4*>(As + p * PLA + r * 64 + c * 16) = ra[p][v];
                *reinterpret_cast<u32x4*>(Ws + p * PLB + r * 64 + c * 16) = rw[p][v];
            }
    };
    auto compute = [&]() {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
#if MI_PLANES_FP16
            f16x8 a[TM][2], b[TN][2];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = (wm * TM + i) * 32 + l31, c = (2 * s + kg) ^ ((r >> 2) & 3);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) a[i][pl] = *reinterpret_cast<const f16x8*>(As + pl * PLA + r * 64 + c * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = (wn * TN + j) * 32 + l31, c = (2 * s + kg) ^ ((r >> 2) & 3);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) b[j][pl] = *reinterpret_cast<const f16x8*>(Ws + pl * PLB + r * 64 + c * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // the two residual cross terms first, the leading term last
#if !MI_TF32_CLASS   // (the TF32-class build keeps the leading term only: 11-bit operands, f32 accumulate)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
#endif
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
#else
            bf16x8 a[TM][3], b[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = (wm * TM + i) * 32 + l31, c = (2 * s + kg) ^ ((r >> 2) & 3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PLA + r * 64 + c * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = (wn * TN + j) * 32 + l31, c = (2 * s + kg) ^ ((r >> 2) & 3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[j][pl] = *reinterpret_cast<const bf16x8*>(Ws + pl * PLB + r * 64 + c * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // smallest terms first, so that they are not lost against a large accumulator
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
#endif
        }
    };

    if constexpr (PF == 0) {
#if MI_PLANES_FP16
        // LDS-DMA form of the 128 x 128 tile (the large-M kernel's pipeline at this tile size): operand tiles by `buffer_load ... lds`
        // straight into two 32 KiB stages [A plane 0 | A plane 1 | W plane 0 | W plane 1] (8 KiB blocks, the XOR swizzle applied on the
        // source side), wave w brings in block w; no staging registers and no ds_write pass; fragments software-pipelined over two
        // register sets (F0 = first 16-deep half of a k-tile, F1 = second), ONE barrier per k-tile, DMA two k-tiles ahead.  A lone
        // workgroup on a CU then overlaps its own LDS reads, DMA and MFMAs instead of running them in turn (0.8 us per k-tile in the
        // register-staged loop).  Same k order and term order per output: bit-identical results.
        static_assert(TM == 2, "the LDS-DMA form exists for 128-row tiles");
        constexpr int STG = 4 * 8192;
        const int KT_ = (K + 31) / 32;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int blk_op = wv >> 1, blk_pl = wv & 1;
        const Planes& X = blk_op ? W : A;
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(X.base + X.tile(blk_op ? ct : rt, 0) + blk_pl * 4096, KT_ * 24576 - blk_pl * 8192);
        const int voff = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
        auto dma_piece = [&](int kt, int stage, int sub) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + stage * STG + wv * 8192 + sub * 1024), 16, voff,
                                                     kt * 24576 + sub * 1024, 0, 0);
        };
        auto dma = [&](int kt, int stage) {
#pragma unroll
            for (int sub = 0; sub < 8; ++sub) dma_piece(kt, stage, sub);
        };
        struct Frag {
            f16x8 a[TM][2], b[TN][2];
        };
        auto read_a = [&](const unsigned char* st, int s2, Frag& f, int i) {
            const int r = (wm * TM + i) * 32 + l31, c = (2 * s2 + kg) ^ ((r >> 2) & 3);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) f.a[i][pl] = *reinterpret_cast<const f16x8*>(st + pl * 8192 + r * 64 + c * 16);
        };
        auto read_b = [&](const unsigned char* st, int s2, Frag& f, int j) {
            const int r = (wn * TN + j) * 32 + l31, c = (2 * s2 + kg) ^ ((r >> 2) & 3);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) f.b[j][pl] = *reinterpret_cast<const f16x8*>(st + (2 + pl) * 8192 + r * 64 + c * 16);
        };
        auto mma3 = [&](const Frag& f, int i, int j) {   // the three terms in the register-staged loop's order
#if !MI_TF32_CLASS   // (the TF32-class build keeps the leading term only: 11-bit operands, f32 accumulate)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][1], f.b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][0], f.b[j][1], acc[i][j], 0, 0, 0);
#endif
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][0], f.b[j][0], acc[i][j], 0, 0, 0);
        };
        f32x16 accS[V == 1 ? TM : 1][V == 1 ? TN : 1];
        const int khalf = V == 1 ? KT_ / 2 : -1;
        dma(0, 0);
        if (KT_ > 1) {
            dma(1, 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // (loads retire in order: the eight pieces of k-tile 0 are in)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        Frag F0, F1;
#pragma unroll
        for (int j = 0; j < TN; ++j) read_b(smem, 0, F0, j);
#pragma unroll
        for (int i = 0; i < TM; ++i) read_a(smem, 0, F0, i);
        for (int kt = 0; kt < KT_; ++kt) {
            const unsigned char* st = smem + (kt & 1) * STG;
            const unsigned char* stn = smem + ((kt + 1) & 1) * STG;
            if constexpr (V == 1) {
                if (kt == khalf) {   // pair mode: the cosine half of K goes into the second accumulator set
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            accS[i][j] = acc[i][j];
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) read_b(st, 1, F1, j);
#pragma unroll
            for (int i = 0; i < TM; ++i) read_a(st, 1, F1, i);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma3(F0, i, j);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool more = kt + 1 < KT_, more2 = kt + 2 < KT_;
#pragma unroll
            for (int q = 0; q < TM * TN; ++q) {
                if (more2) {
                    dma_piece(kt + 2, kt & 1, 2 * q);
                    dma_piece(kt + 2, kt & 1, 2 * q + 1);
                }
                if (more) {
                    if (q < TN) read_b(stn, 0, F0, q);
                    else read_a(stn, 0, F0, q - TN);
                }
                mma3(F1, q / TN, q % TN);
            }
        }
        if constexpr (V == 1) {
            __syncthreads();   // the epilogue's per-wave patches overlay the operand stages
            planes_epilogue_pairs<TM, TN, EXT>(pe, accS, acc, row0 + wm * TM * 32, col0 + wn * TN * 32, M, N, lane, reinterpret_cast<float*>(smem) + wave * 2304,
                                          cps_local);
            return;
        }
#endif
    } else if constexpr (PF > 1) {
        u32x4 ra[PF][3][TM], rw[PF][3][2];
#pragma unroll
        for (int u = 0; u < PF; ++u) load_tiles(u, ra[u], rw[u]);
        // at_half(k): pair mode switches accumulator sets when k reaches the cosine half of K
        auto run_pf = [&](auto&& at_half) {
            int k0 = 0;
            for (; k0 + PF <= KT; k0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    at_half(k0 + u);
                    store_tiles(ra[u], rw[u]);
                    __syncthreads();
                    load_tiles(k0 + u + PF, ra[u], rw[u]);  // past the last k-tile: outside the descriptor (zeros), never staged
                    compute();
                    __syncthreads();
                }
            }
#pragma unroll
            for (int u = 0; u < PF - 1; ++u) {
                if (k0 + u < KT) {
                    at_half(k0 + u);
                    store_tiles(ra[u], rw[u]);
                    __syncthreads();
                    compute();
                    __syncthreads();
                }
            }
        };
        if constexpr (V == 1) {
            f32x16 accS[TM][TN];
            const int khalf = KT / 2;
            run_pf([&](int k) {
                if (k == khalf) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            accS[i][j] = acc[i][j];
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                        }
                }
            });
            planes_epilogue_pairs<TM, TN, EXT>(pe, accS, acc, row0 + wm * TM * 32, col0 + wn * TN * 32, M, N, lane, reinterpret_cast<float*>(smem) + wave * 2304,
                                          cps_local);
            return;
        } else {
            run_pf([](int) {});
        }
    } else {
    load_tiles(0, ra0, rw0);
    int kt = 0;
    // k-tiles kt .. kend-1.  One register set, loads issued one k-tile ahead (right after the staging barrier, so they have the
    // whole compute phase to land); the second workgroup of the CU covers what is left.  A two-set, two-tiles-ahead loop measured
    // 2 % slower here -- and with the pair mode's second accumulator set live it spilled 120 registers inside the loop; tried again with
    // the three-term fp16 loop (whose compute phase is half as long): 46.7 vs 46.7 structures/s, so the loads are not what it waits for.
    auto run = [&](int kend) {
        for (; kt < kend; ++kt) {
            store_tiles(ra0, rw0);
            __syncthreads();
            load_tiles(kt + 1, ra0, rw0);  // past the last k-tile: outside the descriptor (zeros), never staged
            compute();
            __syncthreads();
        }
    };
    if constexpr (V == 1) {  // PAIR mode: sine half of K into one accumulator set, cosine half into the other (see PlanesEpilogue)
        f32x16 accS[TM][TN];
        run(KT / 2);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                accS[i][j] = acc[i][j];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        run(KT);
        planes_epilogue_pairs<TM, TN, EXT>(pe, accS, acc, row0 + wm * TM * 32, col0 + wn * TN * 32, M, N, lane, reinterpret_cast<float*>(smem) + wave * 2304,
                                      cps_local);
        return;
    }
    run(KT);
    }

    if (planes_epilogue_is_rows(pe, N)) {  // block-uniform
        __syncthreads();                   // the staging patches overlay the operand tiles
        planes_epilogue_rows<TM, TN, EXT>(pe, acc, row0 + wm * TM * 32, col0 + wn * TN * 32, M, N, lane, reinterpret_cast<float*>(smem) + wave * 1152);
    } else if (planes_preact_rows_applies(pe, N)) {  // training forward: pre-activation rows through LDS, the rest in registers
        __syncthreads();
        planes_store_preact_rows<TM, TN>(pe, acc, row0 + wm * TM * 32, col0 + wn * TN * 32, M, N, lane, reinterpret_cast<float*>(smem) + wave * 1152);
        PlanesEpilogue pq = pe;
        pq.ep.pre_act = nullptr;
        planes_epilogue<TM, TN>(pq, acc, row0 + wm * TM * 32, col0 + wn * TN * 32, M, N, l31, kg);
    } else {
        planes_epilogue<TM, TN>(pe, acc, row0 + wm * TM * 32, col0 + wn * TN * 32, M, N, l31, kg);
    }
}

// Occupancy targets (registers per lane follow from them): the plain kernel at MI_PLANES_OCC0 workgroups per CU, the pair-mode kernel
// (two accumulator sets) at MI_PLANES_OCC1.  Measured in the fp16 format, same box, structures/s: (3, 2) 46.5, (4, 2) 46.6 (10
// registers spilled), (3, 3) 44.9 and (4, 3) 45.0 (29 spilled in the pair kernel).
#ifndef MI_PLANES_OCC0
#define MI_PLANES_OCC0 3
#endif
#ifndef MI_PLANES_OCC1
#define MI_PLANES_OCC1 2
#endif
template <int V, int TM = 2, bool EXT = false>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(V == 1 ? MI_PLANES_OCC1 : MI_PLANES_OCC0, V == 1 ? MI_PLANES_OCC1 : MI_PLANES_OCC0)))
void gemm_planes_kernel(Planes A, Planes W, int M, int N, int K, PlanesEpilogue pe, int rt_base) {
    gemm_planes_body<V, TM, EXT>(A, W, M, N, K, pe, rt_base);
}
// the register-staged loop compiled for FOUR waves per SIMD (<= 128 registers): a node-level launch of this form fits beside two resident
// pair-mode workgroups (2 x 190 registers) or two LDS-DMA workgroups (2 x 64 KiB LDS + 32 KiB), instead of waiting for one of them to retire
template <bool EXT = false>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gemm_planes_slim_kernel(Planes A, Planes W, int M, int N, int K, PlanesEpilogue pe, int rt_base) {
    gemm_planes_body<0, 2, EXT, 1>(A, W, M, N, K, pe, rt_base);
}
// the LDS-DMA form of the 128 x 128 tile (see gemm_planes_body, PF = 0): two 32 KiB stages, two workgroups per CU
constexpr int PLANES_DMA_LDS = 2 * 4 * 8192;
template <int V, bool EXT = false>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_planes_dma_kernel(Planes A, Planes W, int M, int N, int K, PlanesEpilogue pe, int rt_base) {
    gemm_planes_body<V, 2, EXT, 0>(A, W, M, N, K, pe, rt_base);
}
// the latency form (see gemm_planes_body): at most two workgroups per CU, i.e. up to 256 registers for the PF operand sets
#ifndef MI_PLANES_PF
#define MI_PLANES_PF 3
#endif
template <int V, bool EXT = false>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
void gemm_planes_lat_kernel(Planes A, Planes W, int M, int N, int K, PlanesEpilogue pe, int rt_base) {
    gemm_planes_body<V, 2, EXT, MI_PLANES_PF>(A, W, M, N, K, pe, rt_base);
}
// dynamic LDS of gemm_planes_kernel: the operand tiles, overlaid after the loop by the epilogue's per-wave staging patches
constexpr int planes_lds_bytes(int V, int TM) {
    const int tiles = NPL * (64 * TM + 128) * 64, stage = 4 * (V == 1 ? 2304 : 1152) * 4;
    return tiles > stage ? tiles : stage;
}

// ------------------------------------------------------------------------------------------------------------------------
// Large-M form (fp16 two-plane format): 256 x 256 tile, 8 waves (2 along M x 4 along N, 128 x 64 outputs per wave), operand tiles
// brought in by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write pass), two 64 KiB LDS stages, ONE barrier
// per 32-deep k-tile.  Per unit of matrix work it moves half the operand bytes through L2 and LDS of the 128 x 128 kernel (a wave
// reads 12 fragments per 24 MFMAs instead of 8 per 12, and nothing is written to LDS by the waves), which is what bounds that kernel
// on the [256k, 512] x [512, 512] products of the MatterGen-shaped network (LDS array ~87 % busy at the full matrix rate).
//   LDS image of a stage: eight 8 KiB blocks [A rows 0-127 | A rows 128-255 | W rows 0-127 | W rows 128-255] x [plane 0 | plane 1],
//   each 128 rows x 64 B with the same XOR swizzle as above (chunk c of row r at r*64 + ((c ^ ((r >> 2) & 3)) * 16)).  LDS-DMA writes
//   lane-linear (M0 base + lane * 16), so the swizzle is applied on the SOURCE side: lane l of a 1 KiB piece (16 rows) fetches
//   chunk (l & 3) ^ ((l >> 4) & 3) of row l >> 2 -- a permutation inside each 64-byte row, the fetch stays one contiguous KiB.
//   Wave w brings in block w (its eight pieces), so every DMA address is wave-uniform + the fixed lane offset.
//   Order per k-tile:  wait own DMA (vmcnt 0) -> barrier -> issue the DMA of k-tile kt + 1 into the other stage -> MFMAs of kt.
//   (RAW: wait, then barrier, then read.  WAR: a wave reaches the barrier of kt only after its fragment reads of kt - 1 returned.)
// Accumulation order per output = the 128 x 128 kernel's (k-tiles in order, the three terms in the same order): bit-identical results.
// ------------------------------------------------------------------------------------------------------------------------
#if MI_PLANES_FP16
constexpr int GEMM_BIG_STAGE = 8 * 8192, GEMM_BIG_LDS = 2 * GEMM_BIG_STAGE;
// SEG: the MFMA-layout epilogue with the fused segmented sum instead of the row-major one -- its own instantiation (an ablation, DESIGN
// 15.3): as a run-time branch inside the default instantiation it cost that one 20-30 % (197 -> 258 us on [102 400, 512] x [512, 512])
template <bool EXT, bool SEG = false>
static __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_planes_big_kernel(Planes A, Planes W, int M, int N, int K,
                                                                                                               PlanesEpilogue pe) {
    constexpr int TM = 4, TN = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    M = pe.rows(M);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, kg = lane >> 5;
    // XCD-aware map: both 256-column tiles of a row tile on one XCD (ids go round-robin over the eight XCDs)
    const int nct = (N + 255) >> 8, id = blockIdx.x, slot = id >> 3;
    const int ct = slot % nct, RT = (slot / nct) * 8 + (id & 7);
    if (RT * 256 >= M) return;
    const int row0 = RT * 256, col0 = ct * 256;
    const int KT = (K + 31) >> 5;

    // this wave's DMA block: operand, 128-row half, plane
    const int blk_op = wave >> 2, blk_half = (wave >> 1) & 1, blk_pl = wave & 1;
    const Planes& X = blk_op ? W : A;
    const int xrt = (blk_op ? ct : RT) * 2 + blk_half, xtiles = ((blk_op ? N : M) + 127) >> 7;
    const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(X.base + X.tile(xrt < xtiles ? xrt : 0, 0) + blk_pl * 4096, xrt < xtiles ? KT * 24576 - blk_pl * 8192 : 0);
    const int voff = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    auto dma = [&](int kt, int stage) {
        unsigned char* dst = smem + stage * GEMM_BIG_STAGE + wave * 8192;
#pragma unroll
        for (int sub = 0; sub < 8; ++sub)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + sub * 1024), 16, voff, kt * 24576 + sub * 1024, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragments of one 16-deep half of a k-tile: a[i][plane] (four 32-row tiles of this wave's A half), b[j][plane]
    struct Frag {
        f16x8 a[TM][2], b[TN][2];
    };
    auto read_b = [&](const unsigned char* st, int s2, Frag& f, int j) {
        const unsigned char* Wb = st + (4 + (wn >> 1) * 2) * 8192;   // W half wn >> 1
        const int r = (wn & 1) * 64 + j * 32 + l31, c = (2 * s2 + kg) ^ ((r >> 2) & 3);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) f.b[j][pl] = *reinterpret_cast<const f16x8*>(Wb + pl * 8192 + r * 64 + c * 16);
    };
    auto read_a = [&](const unsigned char* st, int s2, Frag& f, int i) {
        const unsigned char* Ab = st + (wm * 2) * 8192;               // A half wm
        const int r = i * 32 + l31, c = (2 * s2 + kg) ^ ((r >> 2) & 3);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) f.a[i][pl] = *reinterpret_cast<const f16x8*>(Ab + pl * 8192 + r * 64 + c * 16);
    };
    auto mma3 = [&](const Frag& f, int i, int j) {   // the three terms of one accumulator tile, in the 128 x 128 kernel's order
#if !MI_TF32_CLASS   // (the TF32-class build keeps the leading term only: 11-bit operands, f32 accumulate)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][1], f.b[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][0], f.b[j][1], acc[i][j], 0, 0, 0);
#endif
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][0], f.b[j][0], acc[i][j], 0, 0, 0);
    };
    auto dma_piece = [&](int kt, int stage, int sub) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + stage * GEMM_BIG_STAGE + wave * 8192 + sub * 1024), 16, voff,
                                                 kt * 24576 + sub * 1024, 0, 0);
    };

    // Software pipeline (two register sets of fragments, F0 = first half of a k-tile, F1 = second half):
    //   top of k-tile kt:    read F1(kt) | MFMAs of F0(kt)                         -- the reads land behind the MFMAs
    //   middle:              wait (F1 here, own DMA(kt+1) landed) -> barrier ->
    //                        per accumulator tile: one DMA piece of k-tile kt+2 into the stage just released, part of F0(kt+1), 3 MFMAs of F1(kt)
    // so the matrix pipe only idles at that one barrier, with the next MFMAs' operands already in registers; DMA(kt+2) has a whole k-tile to land.
    dma(0, 0);
    if (KT > 1) {
        dma(1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // (loads retire in order: the eight pieces of k-tile 0 are in)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    Frag F0, F1;
#pragma unroll
    for (int j = 0; j < TN; ++j) read_b(smem, 0, F0, j);
#pragma unroll
    for (int i = 0; i < TM; ++i) read_a(smem, 0, F0, i);
    for (int kt = 0; kt < KT; ++kt) {
        const unsigned char* st = smem + (kt & 1) * GEMM_BIG_STAGE;
        const unsigned char* stn = smem + ((kt + 1) & 1) * GEMM_BIG_STAGE;
#pragma unroll
        for (int j = 0; j < TN; ++j) read_b(st, 1, F1, j);
#pragma unroll
        for (int i = 0; i < TM; ++i) read_a(st, 1, F1, i);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mma3(F0, i, j);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool more = kt + 1 < KT, more2 = kt + 2 < KT;
#pragma unroll
        for (int q = 0; q < TM * TN; ++q) {
            if (more2) dma_piece(kt + 2, kt & 1, q);
            if (more) {
                if (q < TN) read_b(stn, 0, F0, q);
                else if (q < TN + TM) read_a(stn, 0, F0, q - TN);
            }
            mma3(F1, q / TN, q % TN);
        }
    }
    __syncthreads();   // the epilogue's per-wave patches overlay the operand stages
    if constexpr (SEG)   // MFMA-layout epilogue: the second edge GEMM's SiLU + fused segmented sum (inference: no pre-activation rows)
        planes_epilogue<TM, TN>(pe, acc, row0 + wm * 128, col0 + wn * 64, M, N, l31, kg);
    else
        planes_epilogue_rows<TM, TN, EXT>(pe, acc, row0 + wm * 128, col0 + wn * 64, M, N, lane, reinterpret_cast<float*>(smem) + wave * 1152);
}
#endif

// The same contraction on a 256 x 128 tile with 8 waves and DOUBLE-BUFFERED LDS (2 x 72 KiB): one barrier per k-tile instead
// of two, and the staging writes of tile k+1 / the global loads of tile k+2 sit between the two MFMA halves of tile k, so the
// matrix pipe only idles at that one barrier.  One workgroup per CU (LDS), two waves per SIMD as before.
constexpr int GEMM_DB_LDS = 2 * (3 * 256 * 64 + 3 * 128 * 64);
template <bool PAIR>
static __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_planes_db_kernel(Planes A, Planes W, int M, int N,
                                                                                                              int K, PlanesEpilogue pe, int Mlim) {
    constexpr int TM = 2, TN = 2, PLA = 256 * 64, PLW = 128 * 64, BUF = 3 * PLA + 3 * PLW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kg = lane >> 5;
    // Workgroups are dispatched round-robin over the 8 XCDs (id % 8), each with its own L2.  Map ids so that all column tiles
    // of one 256-row tile run on the SAME XCD at about the same time: the streamed A operand then crosses the fabric once
    // instead of once per column tile.
    const int nct = (N + 127) / 128, id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int ct = slot % nct, rt2 = (slot / nct) * 8 + xcd;
    if (rt2 * 256 >= Mlim) return;  // this launch covers rows [0, Mlim)
    const int row0 = rt2 * 256, col0 = ct * 128;
    const int KT = (K + 31) / 32, RT = (M + 127) / 128;
    const int rowtile_bytes = (A.KT * 12288 + 2048) * 2;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // 72 KiB per k-tile = 9 x 16 B per thread: six of A (two 128-row tiles x three planes), three of W
    const __amdgpu_buffer_rsrc_t rsa = uniform_rsrc(A.base + A.tile(2 * rt2, 0), (2 * rt2 + 1 < RT ? 2 : 1) * rowtile_bytes);
    const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(W.base + W.tile(ct, 0), KT * 24576);
    u32x4 ra[6], rw[3];
    auto load_tiles = [&](int kt) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            ra[p] = __builtin_amdgcn_raw_buffer_load_b128(rsa, tid * 16, kt * 24576 + p * 8192, 0);
            ra[3 + p] = __builtin_amdgcn_raw_buffer_load_b128(rsa, tid * 16, rowtile_bytes + kt * 24576 + p * 8192, 0);
            rw[p] = __builtin_amdgcn_raw_buffer_load_b128(rsw, tid * 16, kt * 24576 + p * 8192, 0);
        }
    };
    const int sr = tid >> 2, sc = ((tid & 3) ^ ((sr >> 2) & 3)) * 16;
    auto store_tiles = [&](unsigned char* buf) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            *reinterpret_cast<u32x4*>(buf + p * PLA + sr * 64 + sc) = ra[p];
            *reinterpret_cast<u32x4*>(buf + p * PLA + (128 + sr) * 64 + sc) = ra[3 + p];
            *reinterpret_cast<u32x4*>(buf + 3 * PLA + p * PLW + sr * 64 + sc) = rw[p];
        }
    };
    // fragment sets are software-pipelined: while the MFMAs of one 16-wide k-half run, the ds_reads of the next half
    // (and the staging writes / global loads) are already in flight
    struct Frag {
        bf16x8 a[TM][3], b[TN][3];
    };
    auto read_frag = [&](Frag& f, const unsigned char* buf, int s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = (wm * TM + i) * 32 + l31, c = (2 * s + kg) ^ ((r >> 2) & 3);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) f.a[i][pl] = *reinterpret_cast<const bf16x8*>(buf + pl * PLA + r * 64 + c * 16);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = (wn * TN + j) * 32 + l31, c = (2 * s + kg) ^ ((r >> 2) & 3);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) f.b[j][pl] = *reinterpret_cast<const bf16x8*>(buf + 3 * PLA + pl * PLW + r * 64 + c * 16);
        }
    };
    // the six product terms of one k-half, smallest first (they must not be lost against a large accumulator).  Consecutive
    // MFMAs go to DIFFERENT accumulators: a filler between two dependent MFMAs costs the accumulator-forwarding fast path.
    auto mfma_terms = [&](const Frag& f, int t0, int t1) {
#pragma unroll
        for (int t = t0; t < t1; ++t) {
            const int pa = t == 0 ? 2 : (t == 1 || t == 4 || t == 5) ? 0 : 1;   // (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
            const int pb = t == 0 ? 0 : t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][pa], f.b[j][pb], acc[i][j], 0, 0, 0);
        }
    };
    Frag f0, f1;
    // one k-tile.  On entry: f0 = first-half fragments of tile kt (from `cur`), registers = tile kt+1 (if any).
    // Half 0: second-half fragments of tile kt are fetched and tile kt+1 is staged into `nxt` under the first 24 MFMAs;
    // barrier; half 1: tile kt+2 is requested and the first-half fragments of tile kt+1 are fetched under the other 24.
    auto step = [&](int kt, auto has_next, auto has_next2) {
        const unsigned char* cur = smem + (kt & 1) * BUF;
        unsigned char* nxt = smem + ((kt & 1) ^ 1) * BUF;
        // every wave is past the previous barrier, so nobody still reads `nxt`: stage tile kt+1 at once and re-issue the
        // registers for tile kt+2 -- its loads get a whole k-tile of MFMAs to land
        if constexpr (decltype(has_next)::value) store_tiles(nxt);
        if constexpr (decltype(has_next2)::value) load_tiles(kt + 2);
        read_frag(f1, cur, 1);
        mfma_terms(f0, 0, 6);
        __syncthreads();
        if constexpr (decltype(has_next)::value) read_frag(f0, nxt, 0);
        mfma_terms(f1, 0, 6);
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    load_tiles(0);
    store_tiles(smem);
    if (KT > 1) load_tiles(1);
    __syncthreads();
    read_frag(f0, smem, 0);
    int kt = 0;
    auto advance = [&](int kend) {
        for (; kt < kend && kt + 2 < KT; ++kt) step(kt, yes(), yes());  // steady state: branch-free
        for (; kt < kend; ++kt) {
            if (kt + 1 < KT) step(kt, yes(), no());
            else step(kt, no(), no());
        }
    };
    if constexpr (PAIR) {  // sine half of K into one accumulator set, cosine half into the other (see PlanesEpilogue)
        f32x16 accS[TM][TN];
        advance(KT / 2);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                accS[i][j] = acc[i][j];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        advance(KT);
        __syncthreads();  // the staging patches overlay the operand tiles
        planes_epilogue_pairs<TM, TN, true>(pe, accS, acc, row0 + wm * TM * 32, col0 + wn * TN * 32, M, N, lane, reinterpret_cast<float*>(smem) + wave * 2304);
        return;
    } else {
        advance(KT);
    }
    if (planes_epilogue_is_rows(pe, N)) {  // block-uniform
        __syncthreads();                   // the staging patches overlay the operand tiles
        planes_epilogue_rows<TM, TN>(pe, acc, row0 + wm * TM * 32, col0 + wn * TN * 32, M, N, lane, reinterpret_cast<float*>(smem) + wave * 1152);
    } else {
        planes_epilogue<TM, TN>(pe, acc, row0 + wm * TM * 32, col0 + wn * TN * 32, M, N, l31, kg);
    }
}

// Weight-gradient product  P[split][Na][Kx] = sum over the split's rows m of A[m][Na]^T X[m][Kx]  on the bf16 matrix pipe:
// the fp32 operands are split into three planes ON THE WAY into LDS and transposed there, so that the contraction index m is
// the contiguous one -- the LDS image of a 32-row slab is exactly the plane GEMM's ([plane][128 output rows][32 k] bf16, 64-byte
// rows, 16-byte chunks XOR-swizzled), and the MFMA loop is the same six-term one (staging: see below).  Needs even leading
// dimensions and 8-byte aligned operands.  128 x 128 output tile, next slab prefetched into registers during the MFMAs,
// XCD-aware tile order as in gemm_tn128_kernel, partial tiles reduced in fixed order by tn_reduce_kernel.
// XSILU: the X operand is silu(X) of what is stored (the backward pass keeps the pre-activation Z1; M1 = silu(Z1) is formed here
// instead of by a separate pass over [E, H]).
// F16 (fp16 plane format only): both operands carry a DEVICE-side power-of-two scale {s, 1 / s} from a rigorous bound of their
// magnitude (sa, sx); they are split into TWO fp16 planes and multiplied with three terms, and the tile is scaled back by
// 1 / (sa sx) (exact) before it is written -- half the matrix-pipe work of the six-term form.
template <bool XSILU, bool F16 = false>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void gemm_tn_split_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ X, int ldx, float* __restrict__ P, int M, int Na, int Kx, int rows_per_split,
    int gx, int gy, int nsplit, const float* __restrict__ sa = nullptr, const float* __restrict__ sx = nullptr) {
    constexpr int PLB = 128 * 64;
    __shared__ __attribute__((aligned(16))) unsigned char smem[6 * PLB];
    unsigned char* As = smem;
    unsigned char* Xs = smem + 3 * PLB;
    const int id_ = blockIdx.x, slot_ = id_ >> 3, tile_ = slot_ % (gx * gy), bz = (slot_ / (gx * gy)) * 8 + (id_ & 7);
    if (bz >= nsplit) return;
    const int bx = tile_ % gx, by = tile_ / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kg = lane >> 5;
    const int n0 = by * 128, k0 = bx * 128;
    const int m_begin = bz * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging: a thread owns two adjacent operand columns (8-byte global loads; a wave covers 128 consecutive floats of a row)
    // and 8 consecutive rows of the slab = ONE 16-byte chunk (8 k positions) of two LDS rows
    const int nl = (tid & 63) * 2, mq = tid >> 6;
    const float sca = F16 ? sa[0] : 1.f, scx = F16 ? sx[0] : 1.f, sc_out = F16 ? sa[1] * sx[1] : 1.f;
    const bool a_ok = n0 + nl < Na, x_ok = k0 + nl < Kx;  // (Na, Kx even: a column pair is in range or not as a whole)
    const float* pa = A + n0 + nl;
    const float* px = X + k0 + nl;
    f32x2 ra[8], rx[8];
    auto load_slab = [&](int m0) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int gm = m0 + mq * 8 + t;
            ra[t] = (a_ok && gm < m_end) ? *reinterpret_cast<const f32x2*>(pa + (size_t)gm * lda) : f32x2{0.f, 0.f};
            rx[t] = (x_ok && gm < m_end) ? *reinterpret_cast<const f32x2*>(px + (size_t)gm * ldx) : f32x2{0.f, 0.f};
        }
    };
    auto store_slab = [&]() {
        // (8 consecutive lanes = 8 even rows, same chunk index: conflict-free under the XOR swizzle)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            u32x4 va[3], vx[3];
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) {
                unsigned p[3], q[3];
                float x0 = rx[2 * t2][u], x1 = rx[2 * t2 + 1][u];
                if constexpr (XSILU) {
                    x0 = silu_fast(x0);
                    x1 = silu_fast(x1);
                }
                if constexpr (F16) {
                    pl_split_pair(ra[2 * t2][u], ra[2 * t2 + 1][u], sca, p);
                    pl_split_pair(x0, x1, scx, q);
                } else {
                    split3_pair(ra[2 * t2][u], ra[2 * t2 + 1][u], p);
                    split3_pair(x0, x1, q);
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    va[pl][t2] = p[pl];
                    vx[pl][t2] = q[pl];
                }
            }
            const int row = nl + u, off = row * 64 + ((mq ^ ((row >> 2) & 3)) * 16);
#pragma unroll
            for (int pl = 0; pl < (F16 ? 2 : 3); ++pl) {
                *reinterpret_cast<u32x4*>(As + pl * PLB + off) = va[pl];
                *reinterpret_cast<u32x4*>(Xs + pl * PLB + off) = vx[pl];
            }
        }
    };
    load_slab(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += 32) {
        store_slab();
        __syncthreads();
        if (m0 + 32 < m_end) load_slab(m0 + 32);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if constexpr (F16) {
#if MI_PLANES_FP16
                f16x8 a[2][2], b[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = (wm * 2 + i) * 32 + l31, c = (2 * s + kg) ^ ((r >> 2) & 3);
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) a[i][pl] = *reinterpret_cast<const f16x8*>(As + pl * PLB + r * 64 + c * 16);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int r = (wn * 2 + j) * 32 + l31, c = (2 * s + kg) ^ ((r >> 2) & 3);
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) b[j][pl] = *reinterpret_cast<const f16x8*>(Xs + pl * PLB + r * 64 + c * 16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#if !MI_TF32_CLASS   // (the TF32-class build keeps the leading term only: 11-bit operands, f32 accumulate)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
#endif
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                    }
#endif
            } else {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = (wm * 2 + i) * 32 + l31, c = (2 * s + kg) ^ ((r >> 2) & 3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PLB + r * 64 + c * 16);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = (wn * 2 + j) * 32 + l31, c = (2 * s + kg) ^ ((r >> 2) & 3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[j][pl] = *reinterpret_cast<const bf16x8*>(Xs + pl * PLB + r * 64 + c * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    const int PK = gx * 128;
    float* Pt = P + (size_t)bz * (gy * 128) * PK;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg, k = k0 + wn * 64 + j * 32 + l31;
                Pt[(size_t)n * PK + k] = F16 ? acc[i][j][r] * sc_out : acc[i][j][r];
            }
}

extern int g_tn_split_min_rows;  // shortest row list for which the bf16-pipe weight-gradient kernel is used
extern int g_tn_split;  // long weight-gradient contractions on the bf16 matrix pipe (1, split path only) or the f32 MFMA (0)
// C[Na,Kx] (ldc) += A^T X: picks the kernel by shape and arithmetic path (see gemm_tn_acc for the scratch contract)
// whether gemm_tn_auto will take the bf16-pipe kernel for this shape (the only one that can apply silu to X on the fly)
inline bool gemm_tn_is_split(const float* A, int lda, const float* X, int ldx, int M, int Na, int Kx) {
    return g_gemm_mode != 0 && g_tn_split && M >= g_tn_split_min_rows && Na >= 128 && Kx >= 128 && ((lda | ldx | Na | Kx) & 1) == 0 &&
           ((((uintptr_t)A) | ((uintptr_t)X)) & 7) == 0;
}
// sa / sx (fp16 plane format): optional device-side {scale, 1 / scale} of the two operands -> two-plane fp16 split, three terms
#ifdef MI_GEMM_OWNER   // (defined once, in the owning translation unit: every unit that defined it also carried its kernels)
int gemm_tn_auto(const float* A, int lda, const float* X, int ldx, float* C, int ldc, int M, int Na, int Kx, float* scratch,
                        size_t scratch_floats, hipStream_t s, bool x_silu = false, const float* sa = nullptr, const float* sx = nullptr) {
    if (gemm_tn_is_split(A, lda, X, ldx, M, Na, Kx)) {
        const int gy = cdiv(Na, 128), gx = cdiv(Kx, 128);
        int nsplit = std::max(1, std::min(cdiv(M, 256), cdiv(tn_target_tiles(), gx * gy)));
        while (nsplit > 1 && (size_t)nsplit * gy * 128 * gx * 128 > scratch_floats) --nsplit;
        MI_CHECK((size_t)nsplit * gy * 128 * gx * 128 <= scratch_floats, MI_ENOMEM, "gemm_tn scratch too small");
        const int rows = cdiv(cdiv(M, nsplit), 32) * 32;
        nsplit = cdiv(M, rows);
        const dim3 grid(gx * gy * ((nsplit + 7) / 8 * 8));
        const bool f16 = MI_PLANES_FP16 && sa && sx;
        count_mfma(M, Na, Kx, MI_TF32_CLASS ? 1 : f16 ? 3 : 6);
        if (f16 && x_silu) hipLaunchKernelGGL((gemm_tn_split_kernel<true, true>), grid, dim3(256), 0, s, A, lda, X, ldx, scratch, M, Na, Kx, rows, gx, gy, nsplit, sa, sx);
        else if (f16) hipLaunchKernelGGL((gemm_tn_split_kernel<false, true>), grid, dim3(256), 0, s, A, lda, X, ldx, scratch, M, Na, Kx, rows, gx, gy, nsplit, sa, sx);
        else if (x_silu) hipLaunchKernelGGL((gemm_tn_split_kernel<true, false>), grid, dim3(256), 0, s, A, lda, X, ldx, scratch, M, Na, Kx, rows, gx, gy, nsplit, sa, sx);
        else hipLaunchKernelGGL((gemm_tn_split_kernel<false, false>), grid, dim3(256), 0, s, A, lda, X, ldx, scratch, M, Na, Kx, rows, gx, gy, nsplit, sa, sx);
        tn_reduce(scratch, nsplit, gy * 128, gx * 128, C, ldc, Na, Kx, s);
        MI_KERNEL_CHECK();
        return MI_OK;
    }
    MI_CHECK(!x_silu, MI_EINVAL, "gemm_tn_auto: silu on the X operand needs the bf16-pipe kernel (check gemm_tn_is_split first)");
    return gemm_tn_acc(A, lda, X, ldx, C, ldc, M, Na, Kx, scratch, scratch_floats, s);
}
#else
int gemm_tn_auto(const float* A, int lda, const float* X, int ldx, float* C, int ldc, int M, int Na, int Kx, float* scratch,
                        size_t scratch_floats, hipStream_t s, bool x_silu = false, const float* sa = nullptr, const float* sx = nullptr);
#endif

// edge_stage.hip: C = A W^T on 128-row x 256-column register tiles (four waves, W in fragment order from L2, A by LDS-DMA), row-major epilogue
int gemm_rt(const Planes& A, const u16* Wfrag, int M, int N, int K, const PlanesEpilogue& pe, bool ext, hipStream_t s);
size_t frag_elems(int N, int K);
int pack_frag_from_planes(const Planes& W, int N, int K, u16* dst, hipStream_t s);

// whether the pair-mode epilogue may take its addresses in 32 bits (see PlanesEpilogue::pair_wide): `nodes` x ld_node floats and `graphs` x
// ld_graph floats in the gathered arrays, `edges` rows of H columns in the pre-activation array and the output plane set
inline bool pairs_fit_32bit(int64_t nodes, int ld_node, int64_t graphs, int ld_graph, int64_t edges, int H) {
    const int64_t lim = (int64_t)1 << 32, u24 = (int64_t)1 << 24;
    const int64_t plane_bytes = (int64_t)planes_elems(edges, H) * 2;
    return nodes < u24 && graphs < u24 && edges < u24 && (int64_t)ld_node * 4 < u24 && (int64_t)ld_graph * 4 < u24 && (int64_t)H * 4 < u24 &&
           nodes * ld_node * 4 < lim && graphs * ld_graph * 4 < lim && edges * H * 4 < lim && plane_bytes < lim && ((int64_t)((H + 31) / 32) * 12288 + 2048) * 2 < u24;
}

#ifdef MI_GEMM_OWNER   // (defined once, in the owning translation unit: every unit that defined it also carried its kernels)
int gemm_planes(const Planes& A, const Planes& W, int M, int N, int K, const PlanesEpilogue& pe_in, hipStream_t s) {
    // A may be a wider plane set of which the first K columns are used (A.KT is then only the row-tile stride)
    MI_CHECK(A.KT >= (K + 31) / 32 && W.KT == (K + 31) / 32 && (A.KT == W.KT || K % 32 == 0), MI_EINVAL, "gemm_planes: operand plane sets do not match K");
    if (M <= 0 || N <= 0) return MI_OK;
    count_mfma(M, N, K, MI_PLANES_TERMS);
    PlanesEpilogue pe = pe_in;
    pe.out_scale = 1.f / ((A.dscale ? 1.f : A.scale) * W.scale);
    pe.a_dinv = A.dscale ? A.dscale + 1 : nullptr;
    const bool pair = pe.pair_i != nullptr;
    const bool ext = pe.extended();
    MI_CHECK(!ext || (!pair && planes_epilogue_is_rows(pe, N) && (pe.ld_post_mul & 3) == 0 && (MI_PLANES_FP16 || g_planes_variant != 1)), MI_EINVAL,
             "gemm_planes: plane-set residuals / the second merge / the multiplicand exist in the row-major epilogue of the 128-row kernel only");
    MI_CHECK(!pair || (A.KT % 2 == 0 && pe.Cp.base && pe.ep.row_bias && pe.ep.row_bias2 && pe.ep.row_bias3 && (N & 7) == 0 && !pe.ep.bias), MI_EINVAL,
             "gemm_planes: pair mode needs an even k-tile count, a plane-set output, the three gathered addends and no column bias");
    const int nct = cdiv(N, 128);
    // Ms: the row count the FORM of the product is chosen for.  With a device-side count (pe.m_dev) M is a capacity -- several times the rows that exist
    // for a knn list -- and the caller's expectation pe.m_hint (the last count the host has seen) picks the form; the grids below still cover M.
    const int Ms = (pe.m_dev && pe.m_hint > 0) ? std::min(M, pe.m_hint) : M;
#if MI_PLANES_FP16
    {
        static bool dma_attr_set = false;
        if (!dma_attr_set) {
            MI_HIP(hipFuncSetAttribute((const void*)gemm_planes_dma_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PLANES_DMA_LDS));
            MI_HIP(hipFuncSetAttribute((const void*)gemm_planes_dma_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PLANES_DMA_LDS));
            MI_HIP(hipFuncSetAttribute((const void*)gemm_planes_dma_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PLANES_DMA_LDS));
            MI_HIP(hipFuncSetAttribute((const void*)gemm_planes_dma_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PLANES_DMA_LDS));
            dma_attr_set = true;
        }
    }
#endif
    // pair mode has twice the epilogue per row of MFMA work: two workgroups per CU (128-row kernel) hide it behind each other's main
    // loop, which measured faster than the one-workgroup-per-CU kernel at every size tried
    // (the 256-row double-buffered kernel only exists for the three-plane bf16 format)
#if !MI_PLANES_FP16   // (the two-plane fp16 build never takes this branch: its instantiations -- one of which spills -- are not compiled there)
    if (g_planes_variant == 1 && (int64_t)cdiv(M, 256) * nct >= g_planes_db_min_tiles && !(pair && g_pair_kernel == 0)) {
        static bool attr_set = false;
        if (!attr_set) {
            MI_HIP(hipFuncSetAttribute((const void*)gemm_planes_db_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_DB_LDS));
            MI_HIP(hipFuncSetAttribute((const void*)gemm_planes_db_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_DB_LDS));
            attr_set = true;
        }
        // (measured: peeling the partial last round off to half-size tiles does not pay -- workgroups are dispatched
        // dynamically, so the remainder overlaps the stragglers of the last full round)
        const dim3 grid(nct * ((cdiv(M, 256) + 7) / 8 * 8));
        if (pair) hipLaunchKernelGGL(gemm_planes_db_kernel<true>, grid, dim3(512), GEMM_DB_LDS, s, A, W, M, N, K, pe, M);
        else hipLaunchKernelGGL(gemm_planes_db_kernel<false>, grid, dim3(512), GEMM_DB_LDS, s, A, W, M, N, K, pe, M);
    } else
#endif
    if (pair) {
        int nblk = nct * ((cdiv(M, 128) + 7) / 8 * 8);
        const bool lat = nblk <= g_planes_lat_max_blocks;
        const bool dma = MI_PLANES_FP16 && (g_planes_dma == 2 || (g_planes_dma == 1 && lat));
        if (pe.diag_C0) {  // self edges ride along as extra workgroups behind the GEMM tiles
            pe.diag_block0 = nblk;
            nblk += cdiv(pe.diag_nodes, 8);
        }
        if (pe.pair_wide) {
            if (dma) hipLaunchKernelGGL((gemm_planes_dma_kernel<1, true>), dim3(nblk), dim3(256), PLANES_DMA_LDS, s, A, W, M, N, K, pe, 0);
            else if (lat) hipLaunchKernelGGL((gemm_planes_lat_kernel<1, true>), dim3(nblk), dim3(256), planes_lds_bytes(1, 2), s, A, W, M, N, K, pe, 0);
            else hipLaunchKernelGGL((gemm_planes_kernel<1, 2, true>), dim3(nblk), dim3(256), planes_lds_bytes(1, 2), s, A, W, M, N, K, pe, 0);
        } else {
            if (dma) hipLaunchKernelGGL((gemm_planes_dma_kernel<1>), dim3(nblk), dim3(256), PLANES_DMA_LDS, s, A, W, M, N, K, pe, 0);
            else if (lat) hipLaunchKernelGGL((gemm_planes_lat_kernel<1>), dim3(nblk), dim3(256), planes_lds_bytes(1, 2), s, A, W, M, N, K, pe, 0);
            else hipLaunchKernelGGL((gemm_planes_kernel<1, 2>), dim3(nblk), dim3(256), planes_lds_bytes(1, 2), s, A, W, M, N, K, pe, 0);
        }
    } else if (MI_PLANES_FP16 && W.frag && g_planes_rt && (g_planes_rt > 1 || ext) && (N & 255) == 0 && (K & 63) == 0 && K >= 128 && Ms >= g_planes_rt_min_rows &&
               planes_epilogue_is_rows(pe, N)) {
        return gemm_rt(A, W.frag, M, N, K, pe, ext, s);
    } else if (MI_PLANES_FP16 && (N & 255) == 0 &&
               ((g_planes_big && (g_planes_big > 1 || !ext) && Ms >= g_planes_big_min_rows && planes_epilogue_is_rows(pe, N)) ||
                (MI_HAVE_ABLATION_KERNELS && g_planes_big_seg_min_rows > 0 && Ms >= g_planes_big_seg_min_rows && !ext && pe.seg_part && !pe.ep.pre_act &&
                 !planes_epilogue_is_rows(pe, N)))) {
#if MI_PLANES_FP16
        static bool attr_set = false;
        if (!attr_set) {
            MI_HIP(hipFuncSetAttribute((const void*)gemm_planes_big_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_BIG_LDS));
            MI_HIP(hipFuncSetAttribute((const void*)gemm_planes_big_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_BIG_LDS));
#if MI_HAVE_ABLATION_KERNELS
            MI_HIP(hipFuncSetAttribute((const void*)gemm_planes_big_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_BIG_LDS));
#endif
            attr_set = true;
        }
        const dim3 grid((N >> 8) * ((cdiv(M, 256) + 7) / 8 * 8));
#if MI_HAVE_ABLATION_KERNELS
        if (!planes_epilogue_is_rows(pe, N)) hipLaunchKernelGGL((gemm_planes_big_kernel<false, true>), grid, dim3(512), GEMM_BIG_LDS, s, A, W, M, N, K, pe);
        else
#endif
        if (ext) hipLaunchKernelGGL(gemm_planes_big_kernel<true>, grid, dim3(512), GEMM_BIG_LDS, s, A, W, M, N, K, pe);
        else hipLaunchKernelGGL(gemm_planes_big_kernel<false>, grid, dim3(512), GEMM_BIG_LDS, s, A, W, M, N, K, pe);
#endif
    } else if (cdiv(Ms, 128) * nct < g_planes_small_tiles) {
        // few tiles (node-level products): 64-row tiles -- twice the workgroups, half the serial MFMA work in each
        if (ext) hipLaunchKernelGGL((gemm_planes_kernel<0, 1, true>), dim3(nct * ((cdiv(M, 64) + 7) / 8 * 8)), dim3(256), planes_lds_bytes(0, 1), s, A, W, M, N, K, pe, 0);
        else hipLaunchKernelGGL((gemm_planes_kernel<0, 1>), dim3(nct * ((cdiv(M, 64) + 7) / 8 * 8)), dim3(256), planes_lds_bytes(0, 1), s, A, W, M, N, K, pe, 0);
#if MI_HAVE_ABLATION_KERNELS
    } else if (g_planes_dma >= 4 && nct * ((cdiv(Ms, 128) + 7) / 8 * 8) <= 256) {   // mode 4: one-round launches on the four-waves-per-SIMD build
        if (ext) hipLaunchKernelGGL((gemm_planes_slim_kernel<true>), dim3(nct * ((cdiv(M, 128) + 7) / 8 * 8)), dim3(256), planes_lds_bytes(0, 2), s, A, W, M, N, K, pe, 0);
        else hipLaunchKernelGGL((gemm_planes_slim_kernel<false>), dim3(nct * ((cdiv(M, 128) + 7) / 8 * 8)), dim3(256), planes_lds_bytes(0, 2), s, A, W, M, N, K, pe, 0);
#endif
    } else if (MI_PLANES_FP16 && (g_planes_dma == 2 || (g_planes_dma == 1 && nct * ((cdiv(Ms, 128) + 7) / 8 * 8) <= g_planes_lat_max_blocks) ||
                                  (g_planes_dma >= 3 && nct * ((cdiv(Ms, 128) + 7) / 8 * 8) > 256))) {   // (modes 3 / 4: the LDS-DMA form for the LARGE launches only)
        if (ext) hipLaunchKernelGGL((gemm_planes_dma_kernel<0, true>), dim3(nct * ((cdiv(M, 128) + 7) / 8 * 8)), dim3(256), PLANES_DMA_LDS, s, A, W, M, N, K, pe, 0);
        else hipLaunchKernelGGL((gemm_planes_dma_kernel<0>), dim3(nct * ((cdiv(M, 128) + 7) / 8 * 8)), dim3(256), PLANES_DMA_LDS, s, A, W, M, N, K, pe, 0);
    } else if (nct * ((cdiv(Ms, 128) + 7) / 8 * 8) <= g_planes_lat_max_blocks) {   // at most one round: the latency form
        if (ext) hipLaunchKernelGGL((gemm_planes_lat_kernel<0, true>), dim3(nct * ((cdiv(M, 128) + 7) / 8 * 8)), dim3(256), planes_lds_bytes(0, 2), s, A, W, M, N, K, pe, 0);
        else hipLaunchKernelGGL((gemm_planes_lat_kernel<0>), dim3(nct * ((cdiv(M, 128) + 7) / 8 * 8)), dim3(256), planes_lds_bytes(0, 2), s, A, W, M, N, K, pe, 0);
    } else {
        if (ext) hipLaunchKernelGGL((gemm_planes_kernel<0, 2, true>), dim3(nct * ((cdiv(M, 128) + 7) / 8 * 8)), dim3(256), planes_lds_bytes(0, 2), s, A, W, M, N, K, pe, 0);
        else hipLaunchKernelGGL((gemm_planes_kernel<0, 2>), dim3(nct * ((cdiv(M, 128) + 7) / 8 * 8)), dim3(256), planes_lds_bytes(0, 2), s, A, W, M, N, K, pe, 0);
    }
    MI_KERNEL_CHECK();
    return MI_OK;
}
#else
int gemm_planes(const Planes& A, const Planes& W, int M, int N, int K, const PlanesEpilogue& pe_in, hipStream_t s);
#endif

}  // namespace mi
