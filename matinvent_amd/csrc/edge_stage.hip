// Second linear of the edge MLP with the edge -> node reduction, inference form (cspnet.py:73-79):
//
//     M2 = SiLU(M1 W2^T + b2)   over the E directed edges (rows in CSR order: sorted by source node),
//     part[slot][node] = sum of the node's rows inside one 128-row tile       (finalize / node chain: sum of slots / degree)
//
// What differs from the 128 x 128-tile plane GEMM (gemm_split.h) that runs this product everywhere else:
//   * ONE workgroup owns 128 rows and ALL H = 512 output columns: the 128 x 512 fp32 accumulators live in the registers of its eight
//     waves (128 per lane), so the A operand (the M1 planes) is read from HBM / L2 exactly once instead of once per column tile, and a
//     wave multiplies 4 x 2 MFMA tiles per fragment set -- 8 LDS fragment reads per 24 MFMAs instead of 8 per 12;
//   * the W operand never touches LDS: a weight element is used by exactly one wave of the workgroup, so the weights come pre-split in
//     MFMA FRAGMENT ORDER straight from L2 into a register ring (node_chain.hip's scheme);
//   * the A operand streams through LDS in four 128-deep k-chunks (64 KiB each, two buffers) by LDS-DMA (`buffer_load ... lds`), the XOR
//     swizzle applied on the source side;
//   * the segmented row sum is an MFMA product too:  part = S x M2  with S[node][row] = 1 iff the row's source is that node.  In the MFMA
//     result layout a lane owns one column of 4 x 4 rows -- exactly a B-operand fragment once the rows' k order is permuted, which the
//     0/1 matrix S absorbs -- so SiLU(M2) is split into its two fp16 planes in registers and multiplied by S (exact), instead of the
//     per-run select-and-add loops over 16 registers that made the epilogue a quarter of the old kernel.
// Same plane format, same scales, same three-term products in the same k order as the plane GEMM: M2 agrees to fp32 round-off; the
// partial sums additionally round M2 to the 22 bits of the plane format (scale = the rigorous bound of |Z2| the layer's aggregated
// messages use anyway).
#include <mutex>

#include "net.h"
#include "gemm_split.h"

namespace mi {

int g_edge2_fused = 1;   // inference forwards at hidden_dim 512: the second edge GEMM on the 128 x 512 register-tile kernel (0: plane GEMM)

#if MI_PLANES_FP16

struct EdgeGemm2Args {
    Planes A;                 // M1 planes [E x H] (tile-blocked, scale dsc[0])
    const u16* W2f;           // fragment-order pack of edge_mlp.2.weight [H x H]
    const float* b2;
    const float* dsc;         // {s_M1, 1/s_M1, s_agg, 1/s_agg, ...} (act_scales_eval)
    const int* src;           // [E] source node of each row
    const int* rowptr;        // [N + 1]
    float* part;              // [nslots][N][H], slot = tile - (first row of the node >> 7)
    int E, N;
};

constexpr int EG2_CHUNK = 2 * 128 * 256;            // bytes of one k-chunk in LDS: [plane][row 128][k 128 halfs], 16-byte pieces XOR-swizzled by row
constexpr int EG2_LDS = 2 * EG2_CHUNK + 128 * 4 + 128 * 4;   // two chunks + per-row local source + per-local-node slot base

template <int D>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void edge_gemm2_kernel(EdgeGemm2Args a) {
    constexpr int H = 512, KS = H / 16;   // 32 k-steps of 16
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* srcl = reinterpret_cast<int*>(smem + 2 * EG2_CHUNK);   // [128] local source index of each row (-1: no row)
    int* slotb = srcl + 128;                                     // [128] per local node: tile - first tile of the node
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    const int tile = blockIdx.x, row0 = tile * 128;
    const int nrows = a.E - row0 < 128 ? a.E - row0 : 128;

    // ---- A operand by LDS-DMA: chunk kc = k-tiles 4 kc .. 4 kc + 3 of this row tile, 64 pieces of 1 KiB, eight per wave ----
    // LDS image: [plane][row][16 pieces of 16 B], piece c of row r stored at position c ^ (r & 15)  (conflict-free ds_read_b128 fragments)
    const __amdgpu_buffer_rsrc_t rsa = uniform_rsrc(a.A.base + a.A.tile(tile, 0), a.A.KT * 24576);
    auto dma_chunk = [&](int kc, int buf) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int piece = wave * 8 + q;                   // 0 .. 63: plane = piece >> 5, rows 4 (piece & 31) .. + 3
            const int pl = piece >> 5, r = (piece & 31) * 4 + (lane >> 4), cs = lane & 15, c = cs ^ (r & 15);
            // source: k-tile 4 kc + (c >> 2), plane pl, row r, 16-byte piece (c & 3) of the row's 64 bytes
            const int voff = ((4 * kc + (c >> 2)) * 12288 + pl * 4096 + r * 32 + (c & 3) * 8) * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (__attribute__((address_space(3))) void*)(smem + buf * EG2_CHUNK + piece * 1024), 16, voff, 0, 0, 0);
        }
    };
    dma_chunk(0, 0);

    // ---- W operand: register ring over the k-steps, the wave's two column tiles (64 w .. 64 w + 63) ----
    const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(a.W2f, H * H * 4);
    int voffw[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) voffw[t] = lane * 16 + ((2 * wave + t) * KS) * 2048;
    u32x4 ring[D][2][2];
    auto ring_load = [&](int ks, u32x4 (&w)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) w[t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsw, voffw[t] + pl * 1024, ks * 2048, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) ring_load(d, ring[d]);

    // ---- per-tile segment structure (under the first chunk's flight): local source of every row, slot of every local node ----
    const int node_first = a.src[row0];
    if (tid < 128) {
        const int r = row0 + tid;
        srcl[tid] = r < a.E ? a.src[r] - node_first : -1;
        const int node = node_first + tid;
        slotb[tid] = node < a.N ? tile - (a.rowptr[node] >> 7) : 0;
    }
    const float os = a.dsc[1] * (1.f / PL_SW), s_m2 = a.dsc[2], inv_m2 = a.dsc[3];

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto read_a = [&](int buf, int s, f16x8 (&af)[4][2]) {   // fragments of k-step s (0 .. 7) of the chunk in `buf`
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 32 + l31, c = (2 * s + kg) ^ (r & 15);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[i][pl] = *reinterpret_cast<const f16x8*>(smem + buf * EG2_CHUNK + pl * 32768 + r * 256 + c * 16);
        }
    };
    auto mma = [&](const u32x4 (&w)[2][2], const f16x8 (&af)[4][2]) {   // terms (a1, b0), (a0, b1), (a0, b0): the plane GEMM's order
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][term == 0 ? 1 : 0], __builtin_bit_cast(f16x8, w[j][term == 1 ? 1 : 0]), acc[i][j], 0, 0, 0);
    };
    static_assert(8 % D == 0, "the ring index is static inside a chunk");
#pragma unroll 1
    for (int kc = 0; kc < 4; ++kc) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this chunk's pieces have landed (and the ring's first sets with them)
        __syncthreads();                                    // ... for every wave; and every wave has finished reading the other buffer
        if (kc + 1 < 4) dma_chunk(kc + 1, (kc + 1) & 1);
        f16x8 af[4][2];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            read_a(kc & 1, s, af);
            mma(ring[s % D], af);
            if (kc * 8 + s + D < KS) ring_load(kc * 8 + s + D, ring[s % D]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: M2 = SiLU(acc / (s_A s_W) + b2) -> two fp16 planes in registers -> part = S x M2 on the matrix pipe ----
    // k order of a 32-row block's two k-steps (u = 0, 1): lane group kg holds rows 4 kg + 8 (2 u + (idx >> 2)) + (idx & 3), idx = 0 .. 7.
    // slk[kg][rb][u] = the local sources of those eight rows as bytes (255: no row), read back as one broadcast 8-byte load per fragment.
    const int cnt = (nrows > 0 ? a.src[row0 + nrows - 1] - node_first + 1 : 0);   // local nodes of this tile (<= 128)
    __syncthreads();   // (srcl / slotb written above; every wave is out of the main loop, so the chunk buffers are free)
    unsigned char* slk = smem;   // 16 x 8 bytes, overlays the first chunk buffer
    if (tid < 128) {
        const int kg_ = tid >> 6, rb_ = (tid >> 4) & 3, u_ = (tid >> 3) & 1, idx = tid & 7;
        const int v = srcl[rb_ * 32 + 4 * kg_ + 8 * (2 * u_ + (idx >> 2)) + (idx & 3)];
        slk[tid] = (unsigned char)(v < 0 || v > 254 ? 255 : v);
    }
    __syncthreads();
    unsigned sat = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        const int col = wave * 64 + j * 32 + l31;
        const float bcol = a.b2[col];
#pragma unroll 1
        for (int lb0 = 0; lb0 * 32 < cnt; lb0 += 2) {   // two 32-node blocks of partial sums at a time (one pass for all but degenerate tiles)
            f32x16 ps[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) ps[q][r] = 0.f;
            const bool two = (lb0 + 1) * 32 < cnt;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const f32x16 av = acc[rb][j];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f16x8 mh, ml;
#pragma unroll
                    for (int idx = 0; idx < 8; idx += 2) {
                        const float v0 = silu_fast(av[8 * u + idx] * os + bcol), v1 = silu_fast(av[8 * u + idx + 1] * os + bcol);
                        unsigned p[3];
                        pl_split_pair_acc(v0, v1, s_m2, p, sat);
                        const f16x2 h = __builtin_bit_cast(f16x2, p[0]), lo = __builtin_bit_cast(f16x2, p[1]);
                        mh[idx] = h[0]; mh[idx + 1] = h[1];
                        ml[idx] = lo[0]; ml[idx + 1] = lo[1];
                    }
                    const uint2 sb = *reinterpret_cast<const uint2*>(slk + ((kg * 4 + rb) * 2 + u) * 8);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q == 1 && !two) break;
                        const unsigned me = (unsigned)((lb0 + q) * 32 + l31);
                        f16x8 sf;
#pragma unroll
                        for (int idx = 0; idx < 8; ++idx) {
                            const unsigned byte = ((idx < 4 ? sb.x : sb.y) >> (8 * (idx & 3))) & 255u;
                            sf[idx] = byte == me ? (_Float16)1.0f : (_Float16)0.0f;
                        }
                        ps[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sf, ml, ps[q], 0, 0, 0);
                        ps[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sf, mh, ps[q], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // (keeps the conversions of one fragment from being hoisted over all the others)
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int loc = (lb0 + q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    if (loc < cnt) a.part[((size_t)slotb[loc] * a.N + node_first + loc) * H + col] = ps[q][r] * inv_m2;
                }
        }
    }
    sat_report(sat);
}

int edge_gemm2(mi_net* net, mi_batch* b, int layer, hipStream_t s) {
    constexpr int D = 4;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] { attr_err = hipFuncSetAttribute((const void*)edge_gemm2_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2_LDS); });
    MI_HIP(attr_err);
    const int H = net->H;
    EdgeGemm2Args a;
    a.A = make_planes(b->M1pl, H, PL_S_ACT, b->dsc);
    a.W2f = net->Wnc + (size_t)layer * node_chain_pack_elems(H) + (size_t)5 * H * H * 2;
    a.b2 = net->p("csp_layer_" + std::to_string(layer) + ".edge_mlp.2.bias");
    a.dsc = b->dsc;
    a.src = b->src;
    a.rowptr = b->rowptr;
    a.part = b->part;
    a.E = (int)b->E;
    a.N = b->N;
    hipLaunchKernelGGL(edge_gemm2_kernel<D>, dim3(cdiv(b->E, 128)), dim3(512), EG2_LDS, s, a);
    MI_KERNEL_CHECK();
    return MI_OK;
}

bool edge_gemm2_supported(const mi_net* net) { return g_edge2_fused && net->H == 512 && net->Wnc != nullptr; }

#else

int edge_gemm2(mi_net*, mi_batch*, int, hipStream_t) { return MI_ESTATE; }
bool edge_gemm2_supported(const mi_net*) { return false; }

#endif

}  // namespace mi

extern "C" int mi_debug_set_edge2_fused(int on) {
    const int was = mi::g_edge2_fused;
    mi::g_edge2_fused = on;
    return was;
}
