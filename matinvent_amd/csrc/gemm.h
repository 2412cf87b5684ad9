// Node-level fp32 GEMM on MFMA:  C[M,N] = epi( A[M,K] * W[N,K]^T ).
//
// Both operands are K-contiguous (nn.Linear keeps weight as [out,in]), so A and W tiles are
// staged through LDS with 16-byte loads along K and fed to v_mfma_f32_32x32x2_f32.  The MFMA
// k-dimension is a free permutation: lane (row = l&31, hi = l>>5) reads the float4 at
// k = 8m + 4hi .. +3 of BOTH operands, which gives four k-steps per ds_read_b128 with the
// pairing {8m+j, 8m+4+j}.  Exact fp32 (fma chain), one rounding per product.
//
// Tile: BM x BN outputs per 256-thread workgroup (4 waves as 2x2), BK = 32.
#pragma once
#include <algorithm>

#include "common.h"

namespace mi {

extern int g_tn128;  // weight-gradient products over long row lists on 128 x 128 tiles (1) or always 64 x 64 (0)

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_SSILU = 2 };  // ACT_SSILU: ScaledSiLU = silu(x) / 0.6 (GemNet's activation)

// scratch for split-K partial sums (small-M products: more workgroups, shorter serial k-loops)
struct SplitK {
    float* buf = nullptr;
    size_t floats = 0;
};

struct GemmEpilogue {
    const float* bias = nullptr;      // [N] added to every row
    const float* row_bias = nullptr;  // [G, ld_row_bias]; row r gets row_bias[row_group[r]]
    const int* row_group = nullptr;   // [M]
    int ld_row_bias = 0;
    const float* row_bias2 = nullptr;  // second / third row-gathered addends (edge stage: P_j[dst], G[graph])
    const int* row_group2 = nullptr;
    int ld_row_bias2 = 0;
    const float* row_bias3 = nullptr;
    const int* row_group3 = nullptr;
    int ld_row_bias3 = 0;
    const float* pre_add = nullptr;   // [M, ld_pre_add] added BEFORE the activation (a partial product computed earlier)
    int ld_pre_add = 0;
    const float* residual = nullptr;  // [M, ld_res] added AFTER the activation
    int ld_res = 0;
    float out_scale = 1.f;            // applied last: y = (act(z) + residual) * out_scale  (GemNet's (x + f(x)) / sqrt(2) merges)
    int act = ACT_NONE;
    float* pre_act = nullptr;         // optional [M, ld_pre]: value before the activation (saved for backward)
    int ld_pre = 0;
};

// value after the accumulator + per-column bias: row-gathered addends, optional pre-activation save,
// activation, residual
template <bool FAST = false>
__device__ __forceinline__ float apply_epilogue(const GemmEpilogue& ep, float v, int row, int col) {
    if (ep.row_bias) {
        float g = ep.row_bias[(size_t)ep.row_group[row] * ep.ld_row_bias + col];
        if (ep.row_bias2) g += ep.row_bias2[(size_t)ep.row_group2[row] * ep.ld_row_bias2 + col];
        if (ep.row_bias3) g += ep.row_bias3[(size_t)ep.row_group3[row] * ep.ld_row_bias3 + col];
        v += g;
    }
    if (ep.pre_add) v += ep.pre_add[(size_t)row * ep.ld_pre_add + col];
    if (ep.pre_act) ep.pre_act[(size_t)row * ep.ld_pre + col] = v;
    if (ep.act == ACT_SILU) v = FAST ? silu_fast(v) : silu(v);
    else if (ep.act == ACT_SSILU) v = (FAST ? silu_fast(v) : silu(v)) * 1.66666666666666667f;
    if (ep.residual) v += ep.residual[(size_t)row * ep.ld_res + col];
    if (ep.out_scale != 1.f) v *= ep.out_scale;
    return v;
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                      int ldw, float* __restrict__ C, int ldc, int M, int N, int K,
                                                      GemmEpilogue ep) {
    constexpr int BK = 32, LDS_LD = BK + 4;
    constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 sub-tiles per wave along M / N
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Ws[BN * LDS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int A_V = BM * BK / 4 / 256, W_V = BN * BK / 4 / 256;  // float4 per thread per tile
    f32x4 ra[A_V], rw[W_V];

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int v = 0; v < A_V; ++v) {
            int f = tid + v * 256, r = f >> 3, c = (f & 7) * 4;
            int gr = row0 + r, gk = k0 + c;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (gr < M && gk < K) val = *reinterpret_cast<const f32x4*>(A + (size_t)gr * lda + gk);
            ra[v] = val;
        }
#pragma unroll
        for (int v = 0; v < W_V; ++v) {
            int f = tid + v * 256, r = f >> 3, c = (f & 7) * 4;
            int gr = col0 + r, gk = k0 + c;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (gr < N && gk < K) val = *reinterpret_cast<const f32x4*>(W + (size_t)gr * ldw + gk);
            rw[v] = val;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int v = 0; v < A_V; ++v) {
            int f = tid + v * 256, r = f >> 3, c = (f & 7) * 4;
            *reinterpret_cast<f32x4*>(&As[r * LDS_LD + c]) = ra[v];
        }
#pragma unroll
        for (int v = 0; v < W_V; ++v) {
            int f = tid + v * 256, r = f >> 3, c = (f & 7) * 4;
            *reinterpret_cast<f32x4*>(&Ws[r * LDS_LD + c]) = rw[v];
        }
    };

    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        store_tiles();
        __syncthreads();
        if (k0 + BK < K) load_tiles(k0 + BK);  // global loads in flight under the MFMAs
#pragma unroll
        for (int m4 = 0; m4 < BK / 8; ++m4) {
            f32x4 a4[TM], b4[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a4[i] = *reinterpret_cast<const f32x4*>(&As[((wm * TM + i) * 32 + l31) * LDS_LD + m4 * 8 + hi * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b4[j] = *reinterpret_cast<const f32x4*>(&Ws[((wn * TN + j) * 32 + l31) * LDS_LD + m4 * 8 + hi * 4]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][q], b4[j][q], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: acc[r] -> row (r&3) + 8*(r>>2) + 4*hi, col l31  (32x32 C/D map)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int col = col0 + (wn * TN + j) * 32 + l31;
            if (col >= N) continue;
            float bcol = ep.bias ? ep.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = row0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row >= M) continue;
                float v = acc[i][j][r] + bcol;
                v = apply_epilogue(ep, v, row, col);
                C[(size_t)row * ldc + col] = v;
            }
        }
}

// host launcher.  Requirements: K % 4 == 0, lda % 4 == 0, ldw % 4 == 0, 16-byte aligned bases.
#ifdef MI_GEMM_OWNER   // (defined once, in the owning translation unit: every unit that defined it also carried its kernels)
int gemm_nt_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                       const GemmEpilogue& ep, hipStream_t s) {
    MI_CHECK(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, MI_EINVAL, "gemm_nt: K/lda/ldw must be multiples of 4 (%d,%d,%d)", K, lda, ldw);
    MI_CHECK((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0, MI_EINVAL, "gemm_nt: operands must be 16-byte aligned");
    if (M <= 0 || N <= 0) return MI_OK;
    count_mfma(M, N, K, 0);
    // 128x64 tiles once there are enough of them to fill 256 CUs twice; 64x64 otherwise
    if ((int64_t)cdiv(M, 128) * cdiv(N, 64) >= 512) {
        dim3 grid(cdiv(N, 64), cdiv(M, 128));
        hipLaunchKernelGGL((gemm_nt_kernel<128, 64>), grid, dim3(256), 0, s, A, lda, W, ldw, C, ldc, M, N, K, ep);
    } else {
        dim3 grid(cdiv(N, 64), cdiv(M, 64));
        hipLaunchKernelGGL((gemm_nt_kernel<64, 64>), grid, dim3(256), 0, s, A, lda, W, ldw, C, ldc, M, N, K, ep);
    }
    MI_KERNEL_CHECK();
    return MI_OK;
}
#else
int gemm_nt_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                       const GemmEpilogue& ep, hipStream_t s);
#endif

// ------------------------------------------------------------------------------------------------
// Weight-gradient GEMM ("TN"):  C[Na, Kx] (+)= sum_m A[m, Na] * X[m, Kx]   (contraction over rows).
// Both operands are read in their natural row-major layout; the contraction index is the slow one,
// so LDS rows are m and the MFMA fragments are lane-contiguous reads (conflict-free ds_read_b32).
// The row range is split into `nsplit` chunks (grid.z) whose 64x64 partial tiles go to a scratch
// buffer; tn_reduce_kernel adds them in fixed order into the destination (deterministic; gradients
// ACCUMULATE across micro-steps, so the destination is always += ).
// ------------------------------------------------------------------------------------------------
#ifdef MI_GEMM_OWNER   // (launched by the owning unit's launchers only: a static kernel is emitted by EVERY unit that sees its definition)
static __global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, int lda, const float* __restrict__ X, int ldx,
                                                      float* __restrict__ P, int M, int Na, int Kx, int rows_per_split) {
    constexpr int BM = 32, LD = 68;
    __shared__ __attribute__((aligned(16))) float As[BM * LD];
    __shared__ __attribute__((aligned(16))) float Xs[BM * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1, l31 = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    const int m_begin = blockIdx.z * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool a_vec = (lda % 4 == 0) && ((((uintptr_t)A) & 15) == 0);
    const bool x_vec = (ldx % 4 == 0) && ((((uintptr_t)X) & 15) == 0);
    for (int m0 = m_begin; m0 < m_end; m0 += BM) {
        // 32 rows x 64 cols per operand = 512 float4 -> 2 per thread
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            int f = tid + v * 256, r = f >> 4, c = (f & 15) * 4;
            int gm = m0 + r;
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vx = {0.f, 0.f, 0.f, 0.f};
            if (gm < m_end) {
                const float* pa = A + (size_t)gm * lda + n0 + c;
                const float* px = X + (size_t)gm * ldx + k0 + c;
                if (a_vec && n0 + c + 3 < Na) va = *reinterpret_cast<const f32x4*>(pa);
                else
                    for (int u = 0; u < 4; ++u) va[u] = (n0 + c + u < Na) ? pa[u] : 0.f;
                if (x_vec && k0 + c + 3 < Kx) vx = *reinterpret_cast<const f32x4*>(px);
                else
                    for (int u = 0; u < 4; ++u) vx[u] = (k0 + c + u < Kx) ? px[u] : 0.f;
            }
            *reinterpret_cast<f32x4*>(&As[r * LD + c]) = va;
            *reinterpret_cast<f32x4*>(&Xs[r * LD + c]) = vx;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BM / 2; ++s) {
            float av = As[(2 * s + hi) * LD + wn * 32 + l31];
            float xv = Xs[(2 * s + hi) * LD + wk * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // partial tile: P[split][n][k] over the padded [gridDim.y*64][gridDim.x*64] matrix
    const int PK = gridDim.x * 64;
    float* Pt = P + (size_t)blockIdx.z * (gridDim.y * 64) * PK;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, k = k0 + wk * 32 + l31;
        Pt[(size_t)n * PK + k] = acc[r];
    }
}
#endif

// The same product on 128 x 128 output tiles (each wave a 64 x 64 quadrant = four accumulator tiles): half the LDS reads and a
// quarter of the operand re-reads per MFMA of the 64 x 64 kernel, next 32-row slab prefetched into registers during the MFMAs.
// Used for the large weight-gradient products (contraction over the edge or pair list).
#ifdef MI_GEMM_OWNER   // (launched by the owning unit's launchers only: a static kernel is emitted by EVERY unit that sees its definition)
static __global__ __launch_bounds__(256) void gemm_tn128_kernel(const float* __restrict__ A, int lda, const float* __restrict__ X, int ldx,
                                                                float* __restrict__ P, int M, int Na, int Kx, int rows_per_split, int gx,
                                                                int gy, int nsplit) {
    constexpr int BM = 32, LD = 132;
    // XCD-aware mapping (workgroup id % 8 = XCD, each with its own L2): all gx*gy output tiles of one row range run on the SAME
    // XCD, so each operand slab comes from HBM once instead of once per XCD (measured: 3x the algorithmic fetch without it)
    const int id_ = blockIdx.x, slot_ = id_ >> 3, tile_ = slot_ % (gx * gy), bz = (slot_ / (gx * gy)) * 8 + (id_ & 7);
    if (bz >= nsplit) return;
    const int bx = tile_ % gx, by = tile_ / gx;
    __shared__ __attribute__((aligned(16))) float As[BM * LD];
    __shared__ __attribute__((aligned(16))) float Xs[BM * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1, l31 = lane & 31, hi = lane >> 5;
    const int n0 = by * 128, k0 = bx * 128;
    const int m_begin = bz * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool a_vec = (lda % 4 == 0) && ((((uintptr_t)A) & 15) == 0);
    const bool x_vec = (ldx % 4 == 0) && ((((uintptr_t)X) & 15) == 0);
    // a 32-row slab = 32 x 128 floats per operand = 1024 float4: four per thread and operand
    f32x4 va[4], vx[4];
    auto load_slab = [&](int m0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int f = tid + v * 256, r = f >> 5, c = (f & 31) * 4, gm = m0 + r;
            va[v] = f32x4{0.f, 0.f, 0.f, 0.f};
            vx[v] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (gm < m_end) {
                const float* pa = A + (size_t)gm * lda + n0 + c;
                const float* px = X + (size_t)gm * ldx + k0 + c;
                if (a_vec && n0 + c + 3 < Na) va[v] = *reinterpret_cast<const f32x4*>(pa);
                else
                    for (int u = 0; u < 4; ++u) va[v][u] = (n0 + c + u < Na) ? pa[u] : 0.f;
                if (x_vec && k0 + c + 3 < Kx) vx[v] = *reinterpret_cast<const f32x4*>(px);
                else
                    for (int u = 0; u < 4; ++u) vx[v][u] = (k0 + c + u < Kx) ? px[u] : 0.f;
            }
        }
    };
    load_slab(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += BM) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int f = tid + v * 256, r = f >> 5, c = (f & 31) * 4;
            *reinterpret_cast<f32x4*>(&As[r * LD + c]) = va[v];
            *reinterpret_cast<f32x4*>(&Xs[r * LD + c]) = vx[v];
        }
        __syncthreads();
        if (m0 + BM < m_end) load_slab(m0 + BM);
#pragma unroll
        for (int s = 0; s < BM / 2; ++s) {
            float av[2], xv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                av[i] = As[(2 * s + hi) * LD + wn * 64 + i * 32 + l31];
                xv[i] = Xs[(2 * s + hi) * LD + wk * 64 + i * 32 + l31];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], xv[j], acc[i][j], 0, 0, 0);
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
                const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, k = k0 + wk * 64 + j * 32 + l31;
                Pt[(size_t)n * PK + k] = acc[i][j][r];
            }
}
#endif

// (a thread owns four adjacent columns when the output allows 16-byte accesses -- V4 -- and keeps eight partial tiles' loads in flight;
//  the sum runs over the splits in index order whatever the form, so the result does not depend on it)
template <bool V4>
static __global__ void tn_reduce_kernel(const float* __restrict__ P, int nsplit, int PN, int PK, float* __restrict__ C, int ldc, int Na,
                                 int Kx, float scale) {
    constexpr int W = V4 ? 4 : 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int kw = Kx / W;
    if (idx >= Na * kw) return;
    const int n = idx / kw, k = (idx % kw) * W;
    const float* p = P + (size_t)n * PK + k;
    const size_t zs = (size_t)PN * PK;
    float s[W];
#pragma unroll
    for (int c = 0; c < W; ++c) s[c] = 0.f;
    int z = 0;
    for (; z + 8 <= nsplit; z += 8) {
        float v[8][W];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (V4) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(p + (size_t)(z + u) * zs);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[u][c] = q[c];
            } else {
                v[u][0] = p[(size_t)(z + u) * zs];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < W; ++c) s[c] += v[u][c];
    }
    for (; z < nsplit; ++z) {
        if constexpr (V4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p + (size_t)z * zs);
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] += q[c];
        } else {
            s[0] += p[(size_t)z * zs];
        }
    }
    float* c = C + (size_t)n * ldc + k;
    if constexpr (V4) {
        f32x4 o = *reinterpret_cast<const f32x4*>(c);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] += s[i] * scale;
        *reinterpret_cast<f32x4*>(c) = o;
    } else {
        c[0] += s[0] * scale;
    }
}
#ifdef MI_GEMM_OWNER   // (defined once, in the owning translation unit: every unit that defined it also carried its kernels)
void tn_reduce(const float* P, int nsplit, int PN, int PK, float* C, int ldc, int Na, int Kx, hipStream_t s) {
    if ((Kx & 3) == 0 && (ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0)
        hipLaunchKernelGGL(tn_reduce_kernel<true>, dim3(cdiv((int64_t)Na * (Kx / 4), 256)), dim3(256), 0, s, P, nsplit, PN, PK, C, ldc, Na, Kx, 1.0f);
    else
        hipLaunchKernelGGL(tn_reduce_kernel<false>, dim3(cdiv((int64_t)Na * Kx, 256)), dim3(256), 0, s, P, nsplit, PN, PK, C, ldc, Na, Kx, 1.0f);
}
#else
void tn_reduce(const float* P, int nsplit, int PN, int PK, float* C, int ldc, int Na, int Kx, hipStream_t s);
#endif
extern int g_tn_target_tiles;  // workgroups a long weight-gradient contraction is split into (over its row list); the partial tiles are summed by tn_reduce
extern int g_concurrent_groups;
// ... when it has the chip to itself; with n crystal groups fine-tuned concurrently (mi_set_concurrent_groups) a launch's share is 1 / n of that
static inline int tn_target_tiles() { return std::max(64, g_tn_target_tiles / std::max(1, g_concurrent_groups)); }

// C[Na,Kx] (ldc) += A^T X.  `scratch` must hold nsplit * ceil64(Na) * ceil64(Kx) floats.
#ifdef MI_GEMM_OWNER   // (defined once, in the owning translation unit: every unit that defined it also carried its kernels)
int gemm_tn_acc(const float* A, int lda, const float* X, int ldx, float* C, int ldc, int M, int Na, int Kx, float* scratch,
                       size_t scratch_floats, hipStream_t s) {
    if (M <= 0 || Na <= 0 || Kx <= 0) return MI_OK;
    count_mfma(M, Na, Kx, 0);   // (gemm_tn_kernel / gemm_tn128_kernel: f32-input MFMA)
    if (g_tn128 && Na >= 128 && Kx >= 128 && M >= 8192) {  // the edge / pair-list contractions
        const int gy = cdiv(Na, 128), gx = cdiv(Kx, 128);
        int nsplit = std::max(1, std::min(cdiv(M, 256), cdiv(tn_target_tiles(), gx * gy)));
        while (nsplit > 1 && (size_t)nsplit * gy * 128 * gx * 128 > scratch_floats) --nsplit;
        MI_CHECK((size_t)nsplit * gy * 128 * gx * 128 <= scratch_floats, MI_ENOMEM, "gemm_tn scratch too small");
        const int rows = cdiv(cdiv(M, nsplit), 32) * 32;
        nsplit = cdiv(M, rows);
        hipLaunchKernelGGL(gemm_tn128_kernel, dim3(gx * gy * ((nsplit + 7) / 8 * 8)), dim3(256), 0, s, A, lda, X, ldx, scratch, M, Na, Kx, rows, gx, gy, nsplit);
        tn_reduce(scratch, nsplit, gy * 128, gx * 128, C, ldc, Na, Kx, s);
        MI_KERNEL_CHECK();
        return MI_OK;
    }
    const int gy = cdiv(Na, 64), gx = cdiv(Kx, 64);
    int nsplit = std::max(1, std::min(cdiv(M, 256), cdiv(1024, gx * gy)));
    while (nsplit > 1 && (size_t)nsplit * gy * 64 * gx * 64 > scratch_floats) --nsplit;
    MI_CHECK((size_t)nsplit * gy * 64 * gx * 64 <= scratch_floats, MI_ENOMEM, "gemm_tn scratch too small");
    int rows = cdiv(cdiv(M, nsplit), 32) * 32;
    nsplit = cdiv(M, rows);
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(gx, gy, nsplit), dim3(256), 0, s, A, lda, X, ldx, scratch, M, Na, Kx, rows);
    tn_reduce(scratch, nsplit, gy * 64, gx * 64, C, ldc, Na, Kx, s);
    MI_KERNEL_CHECK();
    return MI_OK;
}
#else
int gemm_tn_acc(const float* A, int lda, const float* X, int ldx, float* C, int ldc, int M, int Na, int Kx, float* scratch,
                       size_t scratch_floats, hipStream_t s);
#endif

// out[c] += sum_z P[z][c]  (P row stride ldp): the second stage of the column-sum style reductions.  32 columns (one 128-byte line
// per row) x 8 row groups per block, four independent chains per thread, combined through LDS in a fixed order (deterministic);
// tn_reduce_kernel's one-thread-per-output loop over a few hundred partial rows in two to four workgroups was pure latency (77 us
// for the LayerNorm weight gradients), and 64 columns x 4 row groups took 55 us over the 1600 partial rows of the dZ2 column sums.
// Launch with cdiv(Nc, PART_REDUCE_COLS) blocks of 256 threads.
constexpr int PART_REDUCE_COLS = 32;
template <int MI_UNUSED = 0>   // (a template so that only the units that launch it emit it)
static __global__ __launch_bounds__(256) void part_reduce_kernel(const float* __restrict__ P, int nsplit, int ldp, float* __restrict__ out, int Nc) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, c = blockIdx.x * 32 + cl, rg = threadIdx.x >> 5;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < Nc) {
        int z = rg;
        for (; z + 24 < nsplit; z += 32) {
            s0 += P[(size_t)z * ldp + c];
            s1 += P[(size_t)(z + 8) * ldp + c];
            s2 += P[(size_t)(z + 16) * ldp + c];
            s3 += P[(size_t)(z + 24) * ldp + c];
        }
        for (; z < nsplit; z += 8) s0 += P[(size_t)z * ldp + c];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < Nc)
        out[c] += ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]));
}

// out[c] += sum_m A[m][c]   (bias gradients), two stages through `scratch` like gemm_tn_acc
template <int MI_UNUSED = 0>   // (a template so that only the units that launch it emit it)
static __global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, int lda, float* __restrict__ P, int M, int Nc,
                                                     int rows_per_split, const int* __restrict__ row_idx = nullptr) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int m_begin = blockIdx.y * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    float s = 0.f;
    if (c < Nc)
        for (int m = m_begin + rg; m < m_end; m += 4) s += A[(size_t)(row_idx ? row_idx[m] : m) * lda + c];
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < Nc) P[(size_t)blockIdx.y * (gridDim.x * 64) + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

#ifdef MI_GEMM_OWNER   // (defined once, in the owning translation unit: every unit that defined it also carried its kernels)
int colsum_acc(const float* A, int lda, float* out, int M, int Nc, float* scratch, size_t scratch_floats, hipStream_t s,
                      const int* row_idx = nullptr) {
    if (M <= 0 || Nc <= 0) return MI_OK;
    const int gx = cdiv(Nc, 64);
    int nsplit = std::max(1, std::min(cdiv(M, 64), 64));
    MI_CHECK((size_t)nsplit * gx * 64 <= scratch_floats, MI_ENOMEM, "colsum scratch too small");
    int rows = cdiv(M, nsplit);
    nsplit = cdiv(M, rows);
    hipLaunchKernelGGL(colsum_kernel<>, dim3(gx, nsplit), dim3(256), 0, s, A, lda, scratch, M, Nc, rows, row_idx);
    hipLaunchKernelGGL(part_reduce_kernel<>, dim3(cdiv(Nc, PART_REDUCE_COLS)), dim3(256), 0, s, scratch, nsplit, gx * 64, out, Nc);
    MI_KERNEL_CHECK();
    return MI_OK;
}
#else
int colsum_acc(const float* A, int lda, float* out, int M, int Nc, float* scratch, size_t scratch_floats, hipStream_t s,
                      const int* row_idx = nullptr);
#endif

}  // namespace mi
