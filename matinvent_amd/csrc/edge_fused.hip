// Both products of a layer's edge MLP and the edge -> node reduction in ONE launch, inference form, fc pair mode, hidden_dim 512
// (cspnet.py:59-79):
//
//     Z1(i->j) = C + S + P_i[i] + P_j[j] + G[g],   Z1(j->i) = C - S + P_i[j] + P_j[i] + G[g]      (C / S: cosine / sine halves of the Fourier block)
//     M1 = SiLU(Z1)  ->  M2 = SiLU(M1 W2^T + b2)  ->  part[slot][node] = sum of the node's rows inside the tile
//
// The three-launch form writes M1 as plane sets (54 MB per 64-crystal launch) and reads it back; timing builds put those stores alone
// at 12-16 % of a denoising step.  Here a workgroup owns 64 unordered pairs = 128 directed edges and M1 never leaves the CU:
//   * the second product's 128 x 512 accumulators live in the registers of eight waves (a wave: 128 rows x 64 columns, as in
//     edge_stage.hip's eight-wave form), its weights stream from L2 in fragment order;
//   * the first product runs in FOUR column chunks of 128, TRANSPOSED (weights as the first MFMA operand): in the result layout a lane
//     then owns ONE pair and four consecutive columns per register group, so the pair epilogue (five 16-byte gathers per group, both
//     SiLUs, the plane split) writes M1 as 8-byte pieces straight into the LDS buffer the second product reads its A fragments from.
//     Both of its operands come through three LDS stages by LDS-DMA from the plane sets the plane GEMM uses (Fourier planes of the
//     pairs, pair-layout weight planes): a wave owns one 32 x 32 tile (two accumulator sets), so every fragment is shared by 2 or 4 waves;
//   * rows inside a tile are in PAIR order (2p: i -> j, 2p + 1: j -> i): the 0/1 matrix of the segmented sum absorbs any row order;
//   * a node's edges now sit in every pair tile that holds one of its pairs plus one self-edge tile, so the partial sums go to
//     part[slot][node] with slot = tile - first tile of the node's crystal (self-edge tiles: the last slot) and the node chain adds the
//     slots named by the per-node mask (mi_batch::ef_mask) in increasing order: deterministic, no atomics;
//   * self edges (d = 0: the Fourier term is the constant C0) are extra tiles of 128 nodes whose M1 rows are formed directly.
// Same plane format, same scales, same product terms in the same k order as the three-launch form: M1 is bit-identical, the partial sums
// are the same values summed over other row groups (fp32 round-off).
//
// MEASURED (round 3): parity green (tests/test_gpu_forward.py, form 7) and 27 % SLOWER end to end -- 32.9 / 38.5 against
// 45.3 / 49.0 structures/s on one / four chains.  Phase clock (scripts/edge_fused_phases.py), cycles per 128-edge tile: first product 40 k
// per 128-column chunk against 9 k of MFMA work (24 k-tiles of 6 MFMAs per wave: 1.7 k cycles each -- a barrier, 64 KiB of LDS fragment
// reads and a dependent accumulator chain per k-tile; at 64 pairs per workgroup no staging gives a wave more than one 32 x 32 tile per
// fragment), pair epilogue 16.6 k per chunk, second product 13.5 k per chunk (at the pipe's rate), final epilogue 19 k: 309 k against
// 205 k CU-cycles per 128 edges for the three-launch form.  Kept as a recorded experiment in -DMI_ABLATION_KERNELS builds (the kernel sits at
// the 256-register limit with 100 bytes of scratch); mi_debug_set_edge_fused(1) is refused otherwise.
#include <mutex>

#include "net.h"
#include "gemm_split.h"

namespace mi {

int g_edge_fused = 0;   // 1: inference forwards take this launch instead of the pair GEMM + the second edge GEMM (mi_debug_set_edge_fused)

#if MI_PLANES_FP16 && MI_HAVE_ABLATION_KERNELS

struct EdgeFusedArgs {
    Planes F;                 // Fourier planes of the pairs [Np x 2 Kh], scale PL_S_UNIT
    Planes W1;                // pair-layout weight planes [H x 2 Kh]
    const u16* W2f;           // fragment-order pack of edge_mlp.2.weight [H x H]
    const float* b2;
    const float* PQ;          // [N][ldpq]: P_i | P_j (| X_part)
    int ldpq;
    const float* G;           // [B][H] gram term + bias of this layer
    const float* C0;          // [H] the Fourier term of a self edge
    const int *pair_i, *pair_j, *pair_graph, *node2graph;
    const int* tile0;         // [N] first pair tile of the node's crystal
    const unsigned* sc_pq;    // absmax slots and weight bounds -> this layer's activation scales (act_scales_eval)
    const unsigned* sc_gmax;
    const float* sc_wb;
    float* sc_dsc;            // [6] published by workgroup 0 for the node chain
    float* part;              // [nslots][N][H]
    int Np, N, npt, nslots, K1;   // pairs, nodes, pair tiles (workgroups npt ..: self-edge tiles), slots, 2 Kh
    unsigned long long* clk;  // optional phase clock: [workgroup][16] s_memtime stamps (mi_debug_edge_fused_clock)
};

constexpr int EF_STAGE = 16384 + 8192;               // one k-tile of both operands: W [2 planes][128 rows][64 B] | F [2 planes][64 rows][64 B]
constexpr int EF_NST = 3;
constexpr int EF_ROWB = 272;                         // M1 chunk row: 128 halfs + 16 B pad (conflict-free 16-byte fragment reads)
constexpr int EF_PLB = 128 * EF_ROWB;                // one plane of the M1 chunk
constexpr int EF_M1 = EF_NST * EF_STAGE;             // offset of the M1 chunk
constexpr int EF_TAB = EF_M1 + 2 * EF_PLB;           // srcl[128] | slotb[128]
constexpr int EF_LDS = EF_TAB + 1024;

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void edge_fused_kernel(EdgeFusedArgs a) {
    constexpr int H = 512, KS2 = H / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* srcl = reinterpret_cast<int*>(smem + EF_TAB);
    int* slotb = srcl + 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    const int tile = blockIdx.x;
    const bool diag = tile >= a.npt;
    const int KT1 = a.K1 >> 5;   // k-tiles of the first product (sine half, then cosine half)

    // this layer's scales (every workgroup derives them; workgroup 0 publishes them for the kernels that follow)
    float dsc[6];
    act_scales_eval(__uint_as_float(a.sc_pq[0]), __uint_as_float(a.sc_gmax[0]), a.sc_wb, dsc);
    if (tile == 0 && tid < 6) a.sc_dsc[tid] = dsc[tid];
    const float cps = dsc[0];                                   // scale of the M1 planes
    const float os1 = 1.f / (PL_S_UNIT * PL_SW);                // first product: accumulator -> value
    const float os2 = dsc[1] * (1.f / PL_SW), s_m2 = dsc[2], inv_m2 = dsc[3];
    unsigned sat = 0;
    auto stamp = [&](int i) {
        if (a.clk && tid == 0) a.clk[(size_t)tile * 16 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    // ---- rows of this tile: local row r -> source node; first node, node count ----
    int node_first, cnt, nrows;
    if (!diag) {
        const int p0 = tile * 64, np = a.Np - p0 < 64 ? a.Np - p0 : 64;
        nrows = 2 * np;
        // pairs are sorted by (crystal, i, j): the smallest source is the first pair's i, the largest some pair's j
        node_first = a.pair_i[p0];
        int hi = 0;
        if (tid < 64) hi = tid < np ? a.pair_j[p0 + tid] : 0;
        // (wave 0 reduces the maximum over its 64 lanes)
        if (wave == 0) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) hi = max(hi, __shfl_xor(hi, o, 64));
            if (lane == 0) srcl[0] = hi;
        }
        __syncthreads();
        cnt = srcl[0] - node_first + 1;
        __syncthreads();
        if (tid < 128) {
            const int pl = tid >> 1, dir = tid & 1;
            srcl[tid] = pl < np ? (dir == 0 ? a.pair_i[p0 + pl] : a.pair_j[p0 + pl]) - node_first : -1;
            const int node = node_first + tid;
            const int sl = (tid < cnt && node < a.N) ? tile - a.tile0[node] : -1;   // (nodes of pairless crystals inside the range: tile0 = 2^30)
            slotb[tid] = (sl >= 0 && sl < a.nslots - 1) ? sl : -1;
        }
    } else {
        const int n0 = (tile - a.npt) * 128;
        nrows = a.N - n0 < 128 ? a.N - n0 : 128;
        node_first = n0;
        cnt = nrows;
        if (tid < 128) {
            srcl[tid] = tid < nrows ? tid : -1;
            slotb[tid] = a.nslots - 1;
        }
    }

    // ---- second product: W2 fragments from L2, ring of two k-steps; the wave's columns 64 w .. 64 w + 63 ----
    const __amdgpu_buffer_rsrc_t rsw2 = uniform_rsrc(a.W2f, H * H * 4);
    int voffw2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) voffw2[t] = lane * 16 + ((2 * wave + t) * KS2) * 2048;
    u32x4 ring[2][2][2];
    auto ring_load = [&](int ks, u32x4 (&w)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) w[t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsw2, voffw2[t] + pl * 1024, ks * 2048, 0);
    };
    f32x16 acc2[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;

    // ---- first product's operands by LDS-DMA: k-tile kt of chunk c = W rows 128 c .. + 127 (16 pieces of 1 KiB) and the tile's 64
    // pairs (8 pieces); 24 pieces, three per wave.  LDS image of both: 64-byte rows, 16-byte piece q of row r at position q ^ ((r >> 2) & 3)
    const int prow0 = (tile & 1) * 64;   // the tile's rows inside its 128-row plane tile
    const __amdgpu_buffer_rsrc_t rsf = uniform_rsrc(a.F.base + a.F.tile(diag ? 0 : tile >> 1, 0), a.F.KT * 24576);
    const int voffd = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    auto dma_tile = [&](int c, int kt, int st) {
        const __amdgpu_buffer_rsrc_t rsw1 = uniform_rsrc(a.W1.base + a.W1.tile(c, 0), a.W1.KT * 24576);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int piece = wave * 3 + q;   // 0 .. 15: W (plane = piece >> 3, rows 16 (piece & 7) ..); 16 .. 23: F (plane = (piece - 16) >> 2, rows 16 ((piece - 16) & 3) ..)
            if (piece < 16) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw1, (__attribute__((address_space(3))) void*)(smem + st * EF_STAGE + piece * 1024), 16, voffd,
                                                         kt * 24576 + (piece >> 3) * 8192 + (piece & 7) * 1024, 0, 0);
            } else {
                const int pf = piece - 16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (__attribute__((address_space(3))) void*)(smem + st * EF_STAGE + 16384 + pf * 1024), 16, voffd,
                                                         kt * 24576 + (pf >> 2) * 8192 + prow0 * 64 + (pf & 3) * 1024, 0, 0);
            }
        }
    };
    const int ct1 = wave & 3, pt1 = wave >> 2;   // the wave's tile of the first product: W rows 32 ct1 .., pairs 32 pt1 ..
    // the lane's pair (first product's result layout: column = pair) and its gather rows
    const int plocal = pt1 * 32 + l31;
    const int pglob = tile * 64 + plocal;
    const bool pok = !diag && pglob < a.Np;
    const int ni = pok ? a.pair_i[pglob] : 0, nj = pok ? a.pair_j[pglob] : 0, gr = pok ? a.pair_graph[pglob] : 0;

    int gk = 0;   // running k-tile counter over the chunks: stage = gk % 3
    if (!diag) {
        dma_tile(0, 0, 0);
        dma_tile(0, 1, 1);
    }
    ring_load(0, ring[0]);
    ring_load(1, ring[1]);

#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        if (!diag) {
            // ================= first product, chunk c: D[32 columns x 32 pairs] per wave, two accumulator sets =================
            f32x16 accS, accC;
#pragma unroll
            for (int r = 0; r < 16; ++r) accC[r] = 0.f;
#pragma unroll 1
            for (int kt = 0; kt < KT1; ++kt, ++gk) {
                if (kt == KT1 / 2) {   // the cosine half of K goes into the second accumulator set
                    accS = accC;
#pragma unroll
                    for (int r = 0; r < 16; ++r) accC[r] = 0.f;
                }
                // k-tile kt's three pieces were issued two k-tiles ago; the younger ones (k-tile kt + 1) may stay in flight
                if (kt + 1 < KT1 || c + 1 < 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();   // landed for every wave; every wave is done with the stage the next DMA goes to
                {
                    int cn = c, kn = kt + 2;
                    if (kn >= KT1) { kn -= KT1; ++cn; }
                    if (cn < 4) dma_tile(cn, kn, (gk + 2) % EF_NST);
                }
                const unsigned char* st = smem + (gk % EF_NST) * EF_STAGE;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    f16x8 wf[2], ff[2];
                    {
                        const int r = ct1 * 32 + l31, q = (2 * s2 + kg) ^ ((r >> 2) & 3);
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) wf[pl] = *reinterpret_cast<const f16x8*>(st + pl * 8192 + r * 64 + q * 16);
                    }
                    {
                        const int r = pt1 * 32 + l31, q = (2 * s2 + kg) ^ ((r >> 2) & 3);
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) ff[pl] = *reinterpret_cast<const f16x8*>(st + 16384 + pl * 4096 + r * 64 + q * 16);
                    }
                    // the plane GEMM's terms and order: (activation lo, weight hi), (activation hi, weight lo), (hi, hi) -- transposed operands
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0], ff[1], accC, 0, 0, 0);
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1], ff[0], accC, 0, 0, 0);
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0], ff[0], accC, 0, 0, 0);
                }
            }
            stamp(1 + 3 * c);
            // ================= pair epilogue of the chunk: both directed edges of the lane's pair, into the M1 chunk =================
            // result layout: register r <-> column 32 ct1 + (r & 3) + 8 (r >> 2) + 4 kg of the chunk; lane <-> pair
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cc = ct1 * 32 + 8 * g + 4 * kg, col = c * 128 + cc;
                f32x4 pii = {0.f, 0.f, 0.f, 0.f}, pjj = pii, pij = pii, pji = pii, gg = pii;
                if (pok) {
                    pii = *reinterpret_cast<const f32x4*>(a.PQ + (size_t)ni * a.ldpq + col);
                    pjj = *reinterpret_cast<const f32x4*>(a.PQ + (size_t)nj * a.ldpq + H + col);
                    pij = *reinterpret_cast<const f32x4*>(a.PQ + (size_t)nj * a.ldpq + col);
                    pji = *reinterpret_cast<const f32x4*>(a.PQ + (size_t)ni * a.ldpq + H + col);
                    gg = *reinterpret_cast<const f32x4*>(a.G + (size_t)gr * H + col);
                }
#pragma unroll
                for (int dir = 0; dir < 2; ++dir) {
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float sv = accS[4 * g + k] * os1, cv = accC[4 * g + k] * os1;
                        const float ac = dir == 0 ? cv + sv : cv - sv;
                        const float gsum = dir == 0 ? (pii[k] + pjj[k]) + gg[k] : (pij[k] + pji[k]) + gg[k];
                        v[k] = pok ? silu_fast(ac + gsum) : 0.f;
                    }
                    unsigned p0[3], p1[3];
                    pl_split_pair_acc(v[0], v[1], cps, p0, sat);
                    pl_split_pair_acc(v[2], v[3], cps, p1, sat);
                    unsigned char* dst = smem + EF_M1 + (2 * plocal + dir) * EF_ROWB + cc * 2;
                    *reinterpret_cast<uint2*>(dst) = make_uint2(p0[0], p1[0]);
                    *reinterpret_cast<uint2*>(dst + EF_PLB) = make_uint2(p0[1], p1[1]);
                }
            }
        } else {
            // ================= self-edge tile: M1 rows of the chunk directly (d = 0: the Fourier term is C0) =================
            const int r = tid >> 2, node = node_first + r;
            const bool ok = r < nrows;
            const int g0 = ok ? a.node2graph[node] : 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int cc = (tid & 3) * 32 + q * 4, col = c * 128 + cc;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(a.C0 + col);
                    const f32x4 pi = *reinterpret_cast<const f32x4*>(a.PQ + (size_t)node * a.ldpq + col);
                    const f32x4 pj = *reinterpret_cast<const f32x4*>(a.PQ + (size_t)node * a.ldpq + H + col);
                    const f32x4 gg = *reinterpret_cast<const f32x4*>(a.G + (size_t)g0 * H + col);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = silu_fast(c0[k] + ((pi[k] + pj[k]) + gg[k]));
                }
                unsigned p0[3], p1[3];
                pl_split_pair_acc(v[0], v[1], cps, p0, sat);
                pl_split_pair_acc(v[2], v[3], cps, p1, sat);
                unsigned char* dst = smem + EF_M1 + r * EF_ROWB + cc * 2;
                *reinterpret_cast<uint2*>(dst) = make_uint2(p0[0], p1[0]);
                *reinterpret_cast<uint2*>(dst + EF_PLB) = make_uint2(p0[1], p1[1]);
            }
        }
        __syncthreads();   // the M1 chunk is complete
        stamp(2 + 3 * c);
        // ================= second product, k-steps 8 c .. 8 c + 7: acc2 += M1[:, chunk] W2[:, chunk]^T =================
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            // (two row tiles at a time: 16 fragment registers live instead of 32 -- the kernel sits at the 256-register limit; every accumulator
            // still receives its three terms in the plane GEMM's order)
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
                f16x8 af[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = (2 * ih + i) * 32 + l31;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) af[i][pl] = *reinterpret_cast<const f16x8*>(smem + EF_M1 + pl * EF_PLB + r * EF_ROWB + (16 * s + 8 * kg) * 2);
                }
#pragma unroll
                for (int term = MI_TERM0; term < 3; ++term)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc2[2 * ih + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][term == 0 ? 1 : 0], __builtin_bit_cast(f16x8, ring[s & 1][j][term == 1 ? 1 : 0]),
                                                                                        acc2[2 * ih + i][j], 0, 0, 0);
            }
            if (c * 8 + s + 2 < KS2) ring_load(c * 8 + s + 2, ring[s & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();   // every wave is done with the M1 chunk before the next chunk's epilogue overwrites it
        stamp(3 + 3 * c);
    }

    // ---- epilogue (edge_stage.hip's): M2 = SiLU(acc2 / (s_A s_W) + b2) -> two fp16 planes in registers -> part = S x M2 on the matrix pipe ----
    u16* sfr = reinterpret_cast<u16*>(smem);   // S fragments overlay the stages
    const int nlb = (cnt + 31) >> 5;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int f = tid; f < nlb * 8 * 64; f += 512) {
        const int ln = f & 63, fu = (f >> 6) & 1, frb = (f >> 7) & 3, fq = f >> 9;
        const int me = fq * 32 + (ln & 31), fkg = ln >> 5;
        u32x4 w;
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            const int ia = 2 * i2, ib = 2 * i2 + 1;
            const int ra = frb * 32 + 4 * fkg + 8 * (2 * fu + (ia >> 2)) + (ia & 3), rbb = frb * 32 + 4 * fkg + 8 * (2 * fu + (ib >> 2)) + (ib & 3);
            w[i2] = (srcl[ra] == me ? 0x3C00u : 0u) | (srcl[rbb] == me ? 0x3C000000u : 0u);
        }
        *reinterpret_cast<u32x4*>(sfr + (size_t)f * 8) = w;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        const int col = wave * 64 + j * 32 + l31;
        const float bcol = a.b2[col];
#pragma unroll 1
        for (int lb0 = 0; lb0 < nlb; lb0 += 2) {
            const bool two = lb0 + 1 < nlb;
            f32x16 ps[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) ps[q][r] = 0.f;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const f32x16 av = acc2[rb][j];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f16x8 mh, ml;
#pragma unroll
                    for (int idx = 0; idx < 8; idx += 2) {
                        const float v0 = silu_fast(av[8 * u + idx] * os2 + bcol), v1 = silu_fast(av[8 * u + idx + 1] * os2 + bcol);
                        unsigned p[3];
                        pl_split_pair_acc(v0, v1, s_m2, p, sat);
                        const f16x2 h = __builtin_bit_cast(f16x2, p[0]), lo = __builtin_bit_cast(f16x2, p[1]);
                        mh[idx] = h[0]; mh[idx + 1] = h[1];
                        ml[idx] = lo[0]; ml[idx + 1] = lo[1];
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q == 1 && !two) break;
                        const f16x8 sf = *reinterpret_cast<const f16x8*>(sfr + (size_t)((((lb0 + q) * 4 + rb) * 2 + u) * 64 + lane) * 8);
                        ps[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sf, ml, ps[q], 0, 0, 0);
                        ps[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sf, mh, ps[q], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int loc = (lb0 + q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    if (loc < cnt && slotb[loc] >= 0) a.part[((size_t)slotb[loc] * a.N + node_first + loc) * H + col] = ps[q][r] * inv_m2;
                }
        }
    }
    stamp(13);
    sat_report(sat);
}

unsigned long long* g_edge_fused_clk = nullptr;

bool edge_fused_supported(const mi_net* net, const mi_batch* b) {
    return g_edge_fused && net->H == 512 && net->Wnc != nullptr && net->Wffpl_pair != nullptr && b->ef_ok && !b->knn && b->Np > 0 && (2 * net->Kh) % 64 == 0;
}

int edge_fused(mi_net* net, mi_batch* b, int l, hipStream_t s) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] { attr_err = hipFuncSetAttribute((const void*)edge_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, EF_LDS); });
    MI_HIP(attr_err);
    const int H = net->H, Kp = 2 * net->Kh;
    EdgeFusedArgs a;
    a.F = make_planes(b->FFpl, Kp, PL_S_UNIT);
    a.W1 = make_planes(net->Wffpl_pair + (size_t)l * planes_elems(H, Kp), Kp);
    a.W2f = net->Wnc + (size_t)l * node_chain_pack_elems(H) + (size_t)5 * H * H * 2;
    a.b2 = net->p("csp_layer_" + std::to_string(l) + ".edge_mlp.2.bias");
    a.PQ = (l == 0 && b->PQ0) ? b->PQ0 : b->PQ;   // (inference only: layer 0's projections live in mi_batch::PQ0)
    a.ldpq = 3 * H;
    a.G = b->G + (size_t)l * b->B * H;
    a.C0 = net->C0 + (size_t)l * H;
    a.pair_i = b->pair_i;
    a.pair_j = b->pair_j;
    a.pair_graph = b->pair_graph;
    a.node2graph = b->node2graph;
    a.tile0 = b->ef_tile0;
    a.sc_pq = b->absmax + 2 * l;
    a.sc_gmax = b->absmax + 2 * l + 1;
    a.sc_wb = net->wbounds + (size_t)l * 8;
    a.sc_dsc = b->dsc;
    a.part = b->part;
    a.Np = (int)b->Np;
    a.N = b->N;
    a.npt = (int)((b->Np + 63) / 64);
    a.nslots = b->ef_nslots;
    a.K1 = Kp;
    a.clk = g_edge_fused_clk;
    const int ndt = (b->N + 127) / 128;
    hipLaunchKernelGGL(edge_fused_kernel, dim3(a.npt + ndt), dim3(512), EF_LDS, s, a);
    MI_KERNEL_CHECK();
    return MI_OK;
}

#else

bool edge_fused_supported(const mi_net*, const mi_batch*) { return false; }
int edge_fused(mi_net*, mi_batch*, int, hipStream_t) { return MI_ESTATE; }

#endif

}  // namespace mi

extern "C" int mi_debug_edge_fused_clock(void* dev_buffer) {
#if MI_PLANES_FP16 && MI_HAVE_ABLATION_KERNELS
    mi::g_edge_fused_clk = (unsigned long long*)dev_buffer;
#endif
    return MI_OK;
}

extern "C" int mi_debug_set_edge_fused(int on) {
    const int was = mi::g_edge_fused;
    if (on && !(MI_PLANES_FP16 && MI_HAVE_ABLATION_KERNELS)) return was;   // (an ablation instantiation: stays off in the default library)
    mi::g_edge_fused = on;
    return was;
}
