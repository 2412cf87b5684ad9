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
#include "common.h"

namespace mi {

enum { ACT_NONE = 0, ACT_SILU = 1 };

struct GemmEpilogue {
    const float* bias = nullptr;      // [N] added to every row
    const float* row_bias = nullptr;  // [G, ld_row_bias]; row r gets row_bias[row_group[r]]
    const int* row_group = nullptr;   // [M]
    int ld_row_bias = 0;
    const float* residual = nullptr;  // [M, ld_res] added AFTER the activation
    int ld_res = 0;
    int act = ACT_NONE;
    float* pre_act = nullptr;         // optional [M, ld_pre]: value before the activation (saved for backward)
    int ld_pre = 0;
};

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
                if (ep.row_bias) v += ep.row_bias[(size_t)ep.row_group[row] * ep.ld_row_bias + col];
                if (ep.pre_act) ep.pre_act[(size_t)row * ep.ld_pre + col] = v;
                if (ep.act == ACT_SILU) v = silu(v);
                if (ep.residual) v += ep.residual[(size_t)row * ep.ld_res + col];
                C[(size_t)row * ldc + col] = v;
            }
        }
}

// host launcher.  Requirements: K % 4 == 0, lda % 4 == 0, ldw % 4 == 0, 16-byte aligned bases.
inline int gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                   const GemmEpilogue& ep, hipStream_t s) {
    MI_CHECK(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, MI_EINVAL, "gemm_nt: K/lda/ldw must be multiples of 4 (%d,%d,%d)", K, lda, ldw);
    MI_CHECK((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0, MI_EINVAL, "gemm_nt: operands must be 16-byte aligned");
    if (M <= 0 || N <= 0) return MI_OK;
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

}  // namespace mi
