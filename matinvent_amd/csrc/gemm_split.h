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

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_split_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                            float* __restrict__ C, int ldc, int M, int N, int K, GemmEpilogue ep) {
    constexpr int BK = 32, ROWB = 80;            // bytes per row per plane (64 + 16 pad: conflict-free b128 reads)
    constexpr int TM = BM / 64, TN = BN / 64;    // 32x32 tiles per wave along M / N
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                    // [3 planes][BM rows][ROWB]
    unsigned char* Ws = smem + 3 * BM * ROWB;    // [3 planes][BN rows][ROWB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kg = lane >> 5;
    const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int A_V = BM * BK / 4 / 256, W_V = BN * BK / 4 / 256;
    f32x4 ra[A_V], rw[W_V];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int v = 0; v < A_V; ++v) {
            int f = tid + v * 256, r = f >> 3, c = (f & 7) * 4, gr = row0 + r, gk = k0 + c;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (gr < M && gk < K) val = *reinterpret_cast<const f32x4*>(A + (size_t)gr * lda + gk);
            ra[v] = val;
        }
#pragma unroll
        for (int v = 0; v < W_V; ++v) {
            int f = tid + v * 256, r = f >> 3, c = (f & 7) * 4, gr = col0 + r, gk = k0 + c;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (gr < N && gk < K) val = *reinterpret_cast<const f32x4*>(W + (size_t)gr * ldw + gk);
            rw[v] = val;
        }
    };
    auto store_split = [&](unsigned char* base, int rows, const f32x4& val, int f) {
        int r = f >> 3, c = (f & 7) * 4;
        u16 p[3][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) split3(val[u], p[0][u], p[1][u], p[2][u]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            uint2 pk = {(unsigned)p[pl][0] | ((unsigned)p[pl][1] << 16), (unsigned)p[pl][2] | ((unsigned)p[pl][3] << 16)};
            *reinterpret_cast<uint2*>(base + ((size_t)pl * rows + r) * ROWB + c * 2) = pk;
        }
    };

    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
        for (int v = 0; v < A_V; ++v) store_split(As, BM, ra[v], tid + v * 256);
#pragma unroll
        for (int v = 0; v < W_V; ++v) store_split(Ws, BN, rw[v], tid + v * 256);
        __syncthreads();
        if (k0 + BK < K) load_tiles(k0 + BK);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 a[TM][3], b[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[i][pl] = *reinterpret_cast<const bf16x8*>(As + ((size_t)pl * BM + (wm * TM + i) * 32 + l31) * ROWB + (16 * s + 8 * kg) * 2);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    b[j][pl] = *reinterpret_cast<const bf16x8*>(Ws + ((size_t)pl * BN + (wn * TN + j) * 32 + l31) * ROWB + (16 * s + 8 * kg) * 2);
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
        }
        __syncthreads();
    }

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
                float v = acc[i][j][r] + bcol;
                v = apply_epilogue(ep, v, row, col);
                C[(size_t)row * ldc + col] = v;
            }
        }
}

inline int gemm_nt_split(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, const GemmEpilogue& ep,
                         hipStream_t s) {
    MI_CHECK(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, MI_EINVAL, "gemm_nt_split: K/lda/ldw must be multiples of 4");
    if (M <= 0 || N <= 0) return MI_OK;
    if ((int64_t)cdiv(M, 128) * cdiv(N, 128) >= 256) {
        constexpr int BM = 128, BN = 128;
        hipLaunchKernelGGL((gemm_nt_split_kernel<BM, BN>), dim3(cdiv(N, BN), cdiv(M, BM)), dim3(256), 3 * (BM + BN) * 80, s, A, lda, W, ldw, C, ldc,
                           M, N, K, ep);
    } else {
        constexpr int BM = 64, BN = 64;
        hipLaunchKernelGGL((gemm_nt_split_kernel<BM, BN>), dim3(cdiv(N, BN), cdiv(M, BM)), dim3(256), 3 * (BM + BN) * 80, s, A, lda, W, ldw, C, ldc,
                           M, N, K, ep);
    }
    MI_KERNEL_CHECK();
    return MI_OK;
}

// Which matrix path the node- and edge-level GEMMs take (process-wide; set by mi_set_gemm_mode).
//   MI_GEMM_SPLIT (default): three-plane bf16 split, six terms -- fp32-class accuracy, bf16 matrix pipe
//   MI_GEMM_F32            : v_mfma_f32_32x32x2_f32 -- bit-for-bit an fp32 fma chain
extern int g_gemm_mode;
inline int gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, const GemmEpilogue& ep,
                   hipStream_t s) {
    return g_gemm_mode == 0 ? gemm_nt_f32(A, lda, W, ldw, C, ldc, M, N, K, ep, s) : gemm_nt_split(A, lda, W, ldw, C, ldc, M, N, K, ep, s);
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
    // +2048 elements (4 KiB) per row tile: without the skew every row tile starts a multiple of 64 KiB apart,
    // i.e. on the same memory channel, and workgroups marching through k in lockstep hammer a few channels
    // (measured: 2x slower than the row-major form).
    __host__ __device__ size_t tile(int rt, int kt) const { return (size_t)rt * ((size_t)KT * 12288 + 2048) + (size_t)kt * 12288; }
    __host__ __device__ size_t elem(int r, int c, int p) const { return tile(r >> 7, c >> 5) + (size_t)p * 4096 + (r & 127) * 32 + (c & 31); }
};
static inline size_t planes_elems(int64_t rows, int cols) { return (size_t)((rows + 127) / 128) * ((size_t)((cols + 31) / 32) * 12288 + 2048); }
static inline Planes make_planes(u16* base, int cols) { return Planes{base, (cols + 31) / 32}; }

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

// fp32 [rows][cols] (row stride ld_src) -> plane set (pads written as zero); one thread per column pair
static __global__ void split_planes_kernel(const float* __restrict__ src, int ld_src, int rows, int cols, Planes dst) {
    const int cp = dst.KT * 16;  // column pairs per padded row
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows_pad = (int64_t)(rows + 127) / 128 * 128;
    if (idx >= rows_pad * cp) return;
    int r = (int)(idx / cp), c = (int)(idx % cp) * 2;
    float x = (r < rows && c < cols) ? src[(size_t)r * ld_src + c] : 0.f;
    float y = (r < rows && c + 1 < cols) ? src[(size_t)r * ld_src + c + 1] : 0.f;
    unsigned p[3];
    split3_pair(x, y, p);
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<unsigned*>(dst.base + dst.elem(r, c, k)) = p[k];
}

struct PlanesEpilogue {
    GemmEpilogue ep;       // bias / gathers / pre_act / act / residual as for the fp32 kernels
    float* C = nullptr;    // optional fp32 output [M][ldc]
    int ldc = 0;
    Planes Cp;             // optional plane-set output (the A operand of the next GEMM)
    // optional fused segmented sum over rows (edge -> node aggregation, cspnet.py:79): rows are edges sorted by
    // `seg_src`; every 32-row block writes the partial sum of each node run it contains to
    // seg_part[slot][node][col], slot = block - first block of that node (fixed order, no atomics);
    // finalize_agg_kernel adds the slots and divides by the degree.
    float* seg_part = nullptr;
    const int* seg_src = nullptr;
    const int* seg_rowptr = nullptr;
    int seg_nodes = 0;
};

// C = epi(A W^T) with both operands given as tile-blocked plane sets; main loop = loads + ds + MFMA only.
//
// LDS image per plane: 128 rows x 64 B, unpadded, 16-byte chunk index XOR-swizzled with (row >> 2) & 3:
//   chunk c of row r lives at r*64 + ((c ^ ((r >> 2) & 3)) * 16).
// Fragment reads (ds_read_b128, 16-lane groups of rows distinct mod 16, same chunk) and staging writes
// (ds_write_b128, 8 consecutive lanes = 2 rows x 4 chunks) are both conflict-free; a padded-row layout
// was 2-way on the writes (SQ_LDS_BANK_CONFLICT = 33 % of LDS cycles).
// Global -> register prefetch runs TWO k-tiles ahead (the A operand streams from HBM/MALL).
// This is the 128x128-tile, two-barriers-per-k-step structure: ~870 TF/s of bf16 MFMA issue (35 % of peak),
// which is its known ceiling on this chip; a 256x256 multi-phase schedule is the next step.
static __global__ __launch_bounds__(256) void gemm_planes_kernel(Planes A, Planes W, int M, int N, int K, PlanesEpilogue pe) {
    constexpr int BM = 128, BN = 128, BK = 32, TM = 2, TN = 2, PLB = 128 * 64;  // bytes per plane tile in LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Ws = smem + 3 * PLB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kg = lane >> 5;
    const int rt = blockIdx.y, ct = blockIdx.x, row0 = rt * BM, col0 = ct * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // a plane tile = 128 rows x 64 B = 512 chunks of 16 B; two per thread per plane
    u32x4 ra0[3][2], rw0[3][2], ra1[3][2], rw1[3][2];
    auto load_tiles = [&](int kt, u32x4 (&ra)[3][2], u32x4 (&rw)[3][2]) {
        const u32x4* ga = reinterpret_cast<const u32x4*>(A.base + A.tile(rt, kt));
        const u32x4* gw = reinterpret_cast<const u32x4*>(W.base + W.tile(ct, kt));
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                ra[p][v] = ga[p * 512 + v * 256 + tid];
                rw[p][v] = gw[p * 512 + v * 256 + tid];
            }
    };
    auto store_tiles = [&](const u32x4 (&ra)[3][2], const u32x4 (&rw)[3][2]) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int f = v * 256 + tid, r = f >> 2, c = (f & 3) ^ ((r >> 2) & 3);
                *reinterpret_cast<u32x4*>(As + p * PLB + r * 64 + c * 16) = ra[p][v];
                *reinterpret_cast<u32x4*>(Ws + p * PLB + r * 64 + c * 16) = rw[p][v];
            }
    };
    auto compute = [&]() {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 a[TM][3], b[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = (wm * TM + i) * 32 + l31, c = (2 * s + kg) ^ ((r >> 2) & 3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PLB + r * 64 + c * 16);
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
        }
    };

    const int KT = (K + 31) / 32;
    load_tiles(0, ra0, rw0);
    if (KT > 1) load_tiles(1, ra1, rw1);
    for (int kt = 0; kt < KT; kt += 2) {
        store_tiles(ra0, rw0);
        __syncthreads();
        if (kt + 2 < KT) load_tiles(kt + 2, ra0, rw0);
        compute();
        __syncthreads();
        if (kt + 1 < KT) {
            store_tiles(ra1, rw1);
            __syncthreads();
            if (kt + 3 < KT) load_tiles(kt + 3, ra1, rw1);
            compute();
            __syncthreads();
        }
    }

    const GemmEpilogue& ep = pe.ep;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rb = row0 + (wm * TM + i) * 32;  // first row of this 32-row block
        // segment structure of the block (runs of equal seg_src), wave-uniform
        uint32_t starts = 0;
        int srcv = 0, nvalid = 0;
        if (pe.seg_part) {
            nvalid = M - rb < 32 ? (M - rb > 0 ? M - rb : 0) : 32;
            srcv = nvalid > 0 ? pe.seg_src[rb + (l31 < nvalid ? l31 : nvalid - 1)] : 0;
            const int prev = __shfl_up(srcv, 1, 64);
            starts = (uint32_t)__ballot(kg == 0 && l31 < nvalid && (l31 == 0 || srcv != prev));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + (wn * TN + j) * 32 + l31;
            const bool col_ok = col < N;
            const float bcol = (ep.bias && col_ok) ? ep.bias[col] : 0.f;
            float val[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb + (r & 3) + 8 * (r >> 2) + 4 * kg;
                float v = 0.f;
                if (row < M && col_ok) {
                    v = apply_epilogue(ep, acc[i][j][r] + bcol, row, col);
                    if (pe.C) pe.C[(size_t)row * pe.ldc + col] = v;
                    if (pe.Cp.base) {
                        u16 p0, p1, p2;
                        split3(v, p0, p1, p2);
                        pe.Cp.base[pe.Cp.elem(row, col, 0)] = p0;
                        pe.Cp.base[pe.Cp.elem(row, col, 1)] = p1;
                        pe.Cp.base[pe.Cp.elem(row, col, 2)] = p2;
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
                    const int node = __builtin_amdgcn_readlane(srcv, sg);
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
                        const int slot = (rb >> 5) - (pe.seg_rowptr[node] >> 5);
                        pe.seg_part[((size_t)slot * pe.seg_nodes + node) * N + col] = sum;
                    }
                }
            }
        }
    }
}

inline int gemm_planes(const Planes& A, const Planes& W, int M, int N, int K, const PlanesEpilogue& pe, hipStream_t s) {
    MI_CHECK(A.KT == (K + 31) / 32 && W.KT == A.KT, MI_EINVAL, "gemm_planes: operand plane sets do not match K");
    if (M <= 0 || N <= 0) return MI_OK;
    hipLaunchKernelGGL(gemm_planes_kernel, dim3(cdiv(N, 128), cdiv(M, 128)), dim3(256), 6 * 128 * 64, s, A, W, M, N, K, pe);
    MI_KERNEL_CHECK();
    return MI_OK;
}

}  // namespace mi
