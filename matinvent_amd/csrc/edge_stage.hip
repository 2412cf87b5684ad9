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

#include <cstdio>
#include <cstdlib>
#include "net.h"
#include "gemm_split.h"
// The k-loops of edge_gemm2b_kernel and gemm_rt_kernel with EVERY memory operation under manual control (2, default): LDS fragment reads and
// the weight ring as inline asm with counted lgkmcnt / vmcnt waits, bare s_barrier.  With compiler-visible LDS reads and __syncthreads() (0) the
// compiler drains the LDS-DMA in flight in front of every barrier (`s_waitcnt vmcnt(1)`: it cannot tell that the k-tiles requested two and
// three iterations ahead target other stages than the one being read); with the reads hidden but the ring loads visible (1) it waits for the
// ring with vmcnt(1) / vmcnt(0), i.e. for the LDS-DMA just issued.  Measured, builds alternating on one box (scripts/gpu_asm_lds_ab.sh):
// 2 vs 0 = 51.8 / 51.3 vs 51.2 / 51.3 structures/s on four chains, 46.1 / 46.4 vs 45.9 / 45.9 on one, MatterGen-shaped sampler 3.12 / 3.16
// vs 3.04 / 2.99; 1 is slower than both.
#ifndef MI_ASM_LDS
#define MI_ASM_LDS 2
#endif
#if MI_ASM_LDS
#define MI_LOOP_BARRIER() __builtin_amdgcn_s_barrier()
#else
#define MI_LOOP_BARRIER() __syncthreads()
#endif
#if MI_ASM_LDS >= 2
// Weight slice of a k-step: requested four steps ahead; the vector-memory operations this wave issues after it and before its use are the
// other three slices in flight (12) and the LDS-DMA pieces of the two k-tile heads in between (8) -- vector loads retire in order, so at most
// 20 outstanding means it has arrived.  (The last three k-tiles and the first drain the counter at their head: fewer operations behind a slice
// there.)  The slice rides through the wait as read-write operands, so that no MFMA can be scheduled in front of it.
#define MI_RING_WAIT(w) asm volatile("s_waitcnt vmcnt(20)" : "+v"((w)[0][0]), "+v"((w)[0][1]), "+v"((w)[1][0]), "+v"((w)[1][1]))
#else
#define MI_RING_WAIT(w) (void)0
#endif

// Cache policy of the LDS-DMA operand streams (the `aux` word of buffer_load ... lds: 2 = nt, "non-temporal"): M1 rows are read by the two column halves of a row
// tile and are dead afterwards; the Fourier operand is read by the four column quarters of a row tile and again by the next layer ~400 us later.  A/B builds only
// (scripts/build_variant.py): measured, see DESIGN 19.7.
#ifndef MI_DMA_AUX_M1
#define MI_DMA_AUX_M1 0
#endif
#ifndef MI_DMA_AUX_FF
#define MI_DMA_AUX_FF 0
#endif

namespace mi {

extern int g_edge2_train;
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
    const int* tab = nullptr; // [tiles][EG2_TAB] per-tile tables {first node, local nodes, srcl[128], slotb[128]} (edge2_tables_kernel; edge_gemm2b_kernel)
    float* Z2 = nullptr;      // optional (training forward): the pre-activation M1 W2^T + b2, fp32 [E][H], kept for the backward pass
    unsigned long long* clk;  // optional phase clock: [tile][8] s_memtime stamps (mi_debug_edge2_clock)
};

// Per-tile tables of the second edge GEMM (a 128-row tile's first source node, its number of local nodes, each row's local source index and each
// local node's slot): a function of the edge list alone, so they are built ONCE per graph (fc: once per batch; knn: once per forward) instead of
// by every workgroup of every layer -- where they were a chain of three dependent loads, each behind an s_waitcnt vmcnt(0) that also waited for
// the twelve LDS-DMA pieces and the weight ring issued in front of it, plus one more at the head of the epilogue (ISA).
constexpr int EG2_TAB = 2 + 128 + 128;
__global__ __launch_bounds__(128) void edge2_tables_kernel(const int* __restrict__ src, const int* __restrict__ rowptr, int E, int N, int* __restrict__ tab) {
    const int tile = blockIdx.x, tid = threadIdx.x, row0 = tile * 128;
    if (row0 >= E) return;
    const int nrows = E - row0 < 128 ? E - row0 : 128;
    const int node_first = src[row0];
    int* t = tab + (size_t)tile * EG2_TAB;
    const int r = row0 + tid, node = node_first + tid;
    t[2 + tid] = r < E ? src[r] - node_first : -1;
    t[130 + tid] = node < N ? tile - (rowptr[node] >> 7) : 0;
    if (tid == 0) {
        t[0] = node_first;
        t[1] = src[row0 + nrows - 1] - node_first + 1;
    }
}

constexpr int EG2_CHUNK = 2 * 128 * 256;            // bytes of one k-chunk in LDS: [plane][row 128][k 128 halfs], 16-byte pieces XOR-swizzled by row
constexpr int EG2_LDS = 2 * EG2_CHUNK + 128 * 4 + 128 * 4;   // two chunks + per-row local source + per-local-node slot base

// D: depth of the weight ring in k-steps; AFB: 2 = the activation fragments of k-step s + 1 are read while k-step s multiplies
#if MI_HAVE_ABLATION_KERNELS   // (the eight-wave 128 x 512 form: superseded by edge_gemm2b_kernel, kept for the recorded A/B)
template <int D, int AFB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void edge_gemm2_kernel(EdgeGemm2Args a) {
    constexpr int H = 512, KS = H / 16;   // 32 k-steps of 16
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* srcl = reinterpret_cast<int*>(smem + 2 * EG2_CHUNK);   // [128] local source index of each row (-1: no row)
    int* slotb = srcl + 128;                                     // [128] per local node: tile - first tile of the node
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    const int tile = blockIdx.x, row0 = tile * 128;
    const int nrows = a.E - row0 < 128 ? a.E - row0 : 128;
    int stamp_i = 0;
    auto stamp = [&]() {
        if (a.clk && tid == 0) a.clk[(size_t)tile * 8 + stamp_i] = __builtin_amdgcn_s_memtime();
        ++stamp_i;
    };
    stamp();

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
        for (int term = MI_TERM0; term < 3; ++term)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][term == 0 ? 1 : 0], __builtin_bit_cast(f16x8, w[j][term == 1 ? 1 : 0]), acc[i][j], 0, 0, 0);
    };
    static_assert(8 % D == 0, "the ring index is static inside a chunk");
#pragma unroll 1
    for (int kc = 0; kc < 4; ++kc) {
        // this chunk's pieces have landed: vector loads retire in order and the pieces were issued BEFORE the chunk's ring loads, so leaving
        // the youngest 4 D ring loads in flight is enough (a full drain exposed one L2 latency per chunk)
        if (kc == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (D == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();                                    // ... for every wave; and every wave has finished reading the other buffer
        if (kc == 0) stamp();
        if (kc + 1 < 4) dma_chunk(kc + 1, (kc + 1) & 1);
        f16x8 af[AFB][4][2];
        if constexpr (AFB == 2) read_a(kc & 1, 0, af[0]);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if constexpr (AFB == 2) {
                if (s + 1 < 8) read_a(kc & 1, s + 1, af[(s + 1) & 1]);
                mma(ring[s % D], af[s & 1]);
            } else {
                read_a(kc & 1, s, af[0]);
                mma(ring[s % D], af[0]);
            }
            if (kc * 8 + s + D < KS) ring_load(kc * 8 + s + D, ring[s % D]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    stamp();

    // ---- epilogue: M2 = SiLU(acc / (s_A s_W) + b2) -> two fp16 planes in registers -> part = S x M2 on the matrix pipe ----
    // k order of a 32-row block's two k-steps (u = 0, 1): lane group kg holds rows 4 kg + 8 (2 u + (idx >> 2)) + (idx & 3), idx = 0 .. 7.
    // The S fragments -- S[32 lb + l31][those rows] = 1 iff the row's local source is that node -- are the same for every wave and both
    // column tiles: built once into LDS in fragment order, [block][rb][u][lane][8 halfs] (8 KiB per 32-node block, overlays the chunk
    // buffers), and read back with one ds_read_b128 each.
    const int cnt = (nrows > 0 ? a.src[row0 + nrows - 1] - node_first + 1 : 0);   // local nodes of this tile (<= 128)
    u16* sfr = reinterpret_cast<u16*>(smem);
    unsigned sat = 0;
    const int nlb = (cnt + 31) >> 5;   // 32-node blocks of partial sums (1 for all but degenerate tiles; <= 4)
    __syncthreads();   // every wave is out of the main loop: the overlay is free; srcl / slotb are visible
    for (int f = tid; f < nlb * 8 * 64; f += 512) {   // one (fragment, lane) per iteration: 16 bytes
        const int ln = f & 63, fu = (f >> 6) & 1, frb = (f >> 7) & 3, fq = f >> 9;
        const int me = fq * 32 + (ln & 31), fkg = ln >> 5;
        u32x4 w;
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            const int ia = 2 * i2, ib = 2 * i2 + 1;
            const int ra = frb * 32 + 4 * fkg + 8 * (2 * fu + (ia >> 2)) + (ia & 3), rbb = frb * 32 + 4 * fkg + 8 * (2 * fu + (ib >> 2)) + (ib & 3);
            w[i2] = (srcl[ra] == me ? 0x3C00u : 0u) | (srcl[rbb] == me ? 0x3C000000u : 0u);   // fp16 1.0 = 0x3C00
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
        for (int lb0 = 0; lb0 < nlb; lb0 += 2) {   // two blocks at a time (one pass for all but degenerate tiles)
            const bool two = lb0 + 1 < nlb;
            f32x16 ps[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) ps[q][r] = 0.f;
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
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q == 1 && !two) break;
                        const f16x8 sf = *reinterpret_cast<const f16x8*>(sfr + (size_t)((((lb0 + q) * 4 + rb) * 2 + u) * 64 + lane) * 8);
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
    stamp();
    sat_report(sat);
}
#endif

// ------------------------------------------------------------------------------------------------------------------------------------
// Form B (default): 128 rows x 256 columns per FOUR-wave workgroup, TWO workgroups per CU.  A wave does the same work as in the form
// above (4 x 2 MFMA tiles per fragment set, 128 accumulator registers, its weights straight from L2), but the two workgroups of a CU are
// independent, so they drift apart and one's epilogue -- SiLU and the plane split are ~11k VALU cycles per wave, a quarter of a tile's
// time in the eight-wave form, where both waves of a SIMD reach it together -- runs under the other's MFMA loop; so do the operand fills
// at a tile's start.  The A operand goes through four 16 KiB LDS stages of one 32-deep k-tile each (the plane GEMM's LDS image and
// source-side swizzle), three k-tiles ahead; the two column halves of a row tile run on the same XCD (its second read of A hits L2).
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int EG2B_STAGE = 2 * 128 * 64, EG2B_NST = 4;                   // [plane][row 128][32 k halfs]
constexpr int EG2B_LDS = EG2B_NST * EG2B_STAGE + 128 * 4 + 128 * 4;      // stages (overlaid by the S fragments in the epilogue) + srcl + slotb

template <int D, bool SAVE_Z = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void edge_gemm2b_kernel(EdgeGemm2Args a) {
    constexpr int H = 512, KS = H / 16, KT = H / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* srcl = reinterpret_cast<int*>(smem + EG2B_NST * EG2B_STAGE);
    int* slotb = srcl + 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    // XCD-aware mapping (workgroups go round-robin to the 8 XCDs): both column halves of a row tile on the same XCD
    const int id = blockIdx.x, slot = id >> 3, half = slot & 1, tile = (slot >> 1) * 8 + (id & 7);
    const int row0 = tile * 128;
    if (row0 >= a.E) return;
    const int nrows = a.E - row0 < 128 ? a.E - row0 : 128;
    int stamp_i = 0;
    auto stamp = [&]() {
        if (a.clk && tid == 0) a.clk[(size_t)(tile * 2 + half) * 8 + stamp_i] = __builtin_amdgcn_s_memtime();
        ++stamp_i;
    };
    stamp();
    if (a.clk && tid == 0) a.clk[(size_t)(tile * 2 + half) * 8 + 4] = __builtin_amdgcn_s_memrealtime();   // (100 MHz: stamps 4 / 5 give the shader clock the other four ran at)

    // per-tile tables (edge2_tables_kernel), requested FIRST: independent loads that land under the operand stream; consumed after it is issued
    const int* tab = a.tab + (size_t)tile * EG2_TAB;
    int t_srcl = -1, t_slotb = 0;
    if (tid < 128) {
        t_srcl = tab[2 + tid];
        t_slotb = tab[130 + tid];
    }
    const int node_first = tab[0], cnt_tab = tab[1];

    // ---- A operand: k-tile kt = one 16 KiB block [plane][128 rows][64 B] of the plane set, 16 pieces of 1 KiB, four per wave ----
    // LDS image of a stage = the plane GEMM's: 64-byte rows, 16-byte piece c of row r at position c ^ ((r >> 2) & 3)
#if defined(MI_DBG_PAIRS_SKIP) && (MI_DBG_PAIRS_SKIP & 32)   // (timing diagnostic, wrong results: M1 is READ from its first eight row tiles -- with bit 16, the whole M1 round trip stays inside the L2s)
    const __amdgpu_buffer_rsrc_t rsa = uniform_rsrc(a.A.base + a.A.tile(tile & 7, 0), a.A.KT * 24576);
#else
    const __amdgpu_buffer_rsrc_t rsa = uniform_rsrc(a.A.base + a.A.tile(tile, 0), a.A.KT * 24576);
#endif
    const int voffa = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    auto dma_tile = [&](int kt, int st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = wave * 4 + q;   // plane = piece >> 3, rows 16 (piece & 7) .. + 15
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (__attribute__((address_space(3))) void*)(smem + st * EG2B_STAGE + piece * 1024), 16, voffa,
                                                     kt * 24576 + (piece >> 3) * 8192 + (piece & 7) * 1024, 0, MI_DMA_AUX_M1);
        }
    };
    dma_tile(0, 0);
    dma_tile(1, 1);
    dma_tile(2, 2);

    // ---- W operand: register ring over the k-steps, the wave's two column tiles: columns 256 half + 64 wave .. + 63 ----
    const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(a.W2f, H * H * 4);
    int voffw[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) voffw[t] = lane * 16 + ((8 * half + 2 * wave + t) * KS) * 2048;
    u32x4 ring[D][2][2];
#if MI_ASM_LDS >= 2
    // (the weight ring as inline asm too, with a counted wait in front of every step's MFMAs: see MI_RING_WAIT)
    const u32x4 rsw_s = rsrc_words(a.W2f, H * H * 4);
    auto ring_load = [&](int ks, u32x4 (&w)[2][2]) {
        const int soff = __builtin_amdgcn_readfirstlane(ks * 2048);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(w[t][0]) : "v"(voffw[t]), "s"(rsw_s), "s"(soff));
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(w[t][1]) : "v"(voffw[t]), "s"(rsw_s), "s"(soff));
        }
    };
#else
    auto ring_load = [&](int ks, u32x4 (&w)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) w[t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsw, voffw[t] + pl * 1024, ks * 2048, 0);
    };
#endif
#pragma unroll
    for (int d = 0; d < D; ++d) ring_load(d, ring[d]);

    if (tid < 128) {   // (the tables were requested before the operand loads: this waits for them alone)
        srcl[tid] = t_srcl;
        slotb[tid] = t_slotb;
    }
    const float os = a.dsc[1] * (1.f / PL_SW), s_m2 = a.dsc[2], inv_m2 = a.dsc[3];

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#if MI_ASM_LDS
    // The fragment reads as inline asm with their own lgkmcnt wait, and the bare s_barrier instead of __syncthreads(): the compiler cannot tell
    // that the LDS-DMA in flight targets other stages than the one being read, so it drains it -- in this loop `s_waitcnt vmcnt(1)` in front of
    // every barrier, i.e. the k-tiles requested two and three iterations ahead had to LAND before the current one was used (found in
    // gemm_tn_planes_kernel, backward.hip, where the same cost 30 %).  The fragments ride through the wait as read-write operands, so that no
    // MFMA can be scheduled in front of it.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned rd_off[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) rd_off[s2] = lds0 + (unsigned)(l31 * 64 + (((2 * s2 + kg) ^ ((l31 >> 2) & 3)) * 16));
    auto read_a = [&](int st, int s2, f16x8 (&af)[4][2]) {
        const unsigned va = rd_off[s2] + (unsigned)st * EG2B_STAGE;
#define MI_RD128(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(va), "n"(off))
        MI_RD128(af[0][0], 0 * 2048);        MI_RD128(af[0][1], 0 * 2048 + 8192);
        MI_RD128(af[1][0], 1 * 2048);        MI_RD128(af[1][1], 1 * 2048 + 8192);
        MI_RD128(af[2][0], 2 * 2048);        MI_RD128(af[2][1], 2 * 2048 + 8192);
        MI_RD128(af[3][0], 3 * 2048);        MI_RD128(af[3][1], 3 * 2048 + 8192);
#undef MI_RD128
        // (no wait here: MI_AF_WAIT in front of the MFMAs that read the fragments -- LDS reads return in order, so the first two row blocks' four
        //  reads are complete when at most four are outstanding)
    };
#else
    auto read_a = [&](int st, int s2, f16x8 (&af)[4][2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 32 + l31, c = (2 * s2 + kg) ^ ((r >> 2) & 3);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[i][pl] = *reinterpret_cast<const f16x8*>(smem + st * EG2B_STAGE + pl * 8192 + r * 64 + c * 16);
        }
    };
#endif
#if MI_ASM_LDS
    auto mma = [&](const u32x4 (&w)[2][2], f16x8 (&af)[4][2]) {   // two row blocks at a time, each pair behind the wait for its own fragments (same term order per accumulator)
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            if (ip == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[2][0]), "+v"(af[2][1]), "+v"(af[3][0]), "+v"(af[3][1]));
#pragma unroll
            for (int term = MI_TERM0; term < 3; ++term)
#pragma unroll
                for (int i = 2 * ip; i < 2 * ip + 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][term == 0 ? 1 : 0], __builtin_bit_cast(f16x8, w[j][term == 1 ? 1 : 0]), acc[i][j], 0, 0, 0);
        }
    };
#else
    auto mma = [&](const u32x4 (&w)[2][2], const f16x8 (&af)[4][2]) {
#pragma unroll
        for (int term = MI_TERM0; term < 3; ++term)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][term == 0 ? 1 : 0], __builtin_bit_cast(f16x8, w[j][term == 1 ? 1 : 0]), acc[i][j], 0, 0, 0);
    };
#endif
    static_assert(D == 4, "two k-tiles of ring per unrolled pair of iterations");
    // vector loads retire in order.  Per k-tile this wave issues 4 DMA pieces (k-tile kt + 3) and then 8 ring loads; k-tile kt's pieces were
    // issued three iterations ago, i.e. at least 8 + 12 + 12 operations ago: vmcnt(24) leaves the younger ones in flight.
#pragma unroll 1
    for (int kt = 0; kt < KT; kt += 2) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int k = kt + h2;
            // (the last three k-tiles have fewer loads behind them -- no further DMA, the ring runs dry: drain)
            if (k == 0 || k >= KT - 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            MI_LOOP_BARRIER();   // k-tile k has landed for every wave; every wave is done with the stage k-tile k + 3 goes to (= that of k - 1)
            if (k == 0) stamp();
            if (k + 3 < KT) dma_tile(k + 3, (k + 3) & 3);
            f16x8 af[4][2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                read_a(k & 3, s2, af);
                MI_RING_WAIT(ring[2 * h2 + s2]);
                mma(ring[2 * h2 + s2], af);
                if (2 * k + s2 + D < KS) ring_load(2 * k + s2 + D, ring[2 * h2 + s2]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    stamp();

    if constexpr (SAVE_Z) {
        // training forward: Z2 = acc / (s_A s_W) + b2 is kept for the backward pass.  In the result layout that would be 128 four-byte
        // stores per lane; through a per-wave LDS patch (the plane GEMM's planes_store_preact_rows) every store is a 16-byte row-major piece.
        __syncthreads();   // every wave is out of the main loop: the patches overlay the operand stages
        float* stage = reinterpret_cast<float*>(smem) + wave * 1152;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rb = row0 + i * 32, cb = half * 256 + wave * 64 + j * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * kg) * 36 + l31] = acc[i][j][r] * os;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = lane + 64 * u, rl = q >> 2, c8 = (q & 3) * 8;
                    const int row = rb + rl, col = cb + c8;
                    f32x4 z0 = *reinterpret_cast<const f32x4*>(stage + rl * 36 + c8);
                    f32x4 z1 = *reinterpret_cast<const f32x4*>(stage + rl * 36 + c8 + 4);
                    if (row < a.E) {
                        z0 += *reinterpret_cast<const f32x4*>(a.b2 + col);
                        z1 += *reinterpret_cast<const f32x4*>(a.b2 + col + 4);
                        float* d = a.Z2 + (size_t)row * H + col;
                        *reinterpret_cast<f32x4*>(d) = z0;
                        *reinterpret_cast<f32x4*>(d + 4) = z1;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
    }

    // ---- epilogue (as in the eight-wave form): SiLU -> two fp16 planes -> part = S x M2 on the matrix pipe ----
    const int cnt = nrows > 0 ? cnt_tab : 0;
    u16* sfr = reinterpret_cast<u16*>(smem);
    unsigned sat = 0;
    const int nlb = (cnt + 31) >> 5;
    __syncthreads();
    for (int f = tid; f < nlb * 8 * 64; f += 256) {
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
        const int col = half * 256 + wave * 64 + j * 32 + l31;
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
                    if (loc < cnt) a.part[((size_t)slotb[loc] * a.N + node_first + loc) * H + col] = ps[q][r] * inv_m2;
                }
        }
    }
    stamp();
    if (a.clk && tid == 0) a.clk[(size_t)(tile * 2 + half) * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    sat_report(sat);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The same main loop as a general product  C = epilogue(A W^T)  with the plane GEMM's row-major epilogue (gemm_rt, picked by
// gemm_planes for operands that carry a fragment-order W): 128 rows x 256 columns per four-wave workgroup, any K % 64 == 0, any
// N % 256 == 0; the column blocks of a row tile run on the same XCD.  Same products, same k order, same epilogue function as the
// 128 x 128 plane kernel: bit-identical results.
// ------------------------------------------------------------------------------------------------------------------------------------
#ifndef MI_RT_PF_AT
#define MI_RT_PF_AT 0   // k-tiles before the end of the loop at which the epilogue's operands are touched (0: never -- measured: no gain, see below)
#endif
template <bool EXT, bool LEAN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_rt_kernel(Planes A, const u16* __restrict__ Wf, int M, int N, int K,
                                                                                                 PlanesEpilogue pe, unsigned long long* __restrict__ clk) {
    const int KS = K >> 4, KT = K >> 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    M = pe.rows(M);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // optional phase clock (mi_debug_rt_clock): per workgroup [8] = s_memtime at entry / first k-tile in LDS / end of the main loop / exit,
    // s_memrealtime (100 MHz) at entry and exit
    int stamp_i = 0;
    auto stamp = [&]() {
        if (clk && tid == 0) clk[(size_t)blockIdx.x * 8 + stamp_i] = __builtin_amdgcn_s_memtime();
        ++stamp_i;
    };
    stamp();
    if (clk && tid == 0) clk[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();
    const int l31 = lane & 31, kg = lane >> 5;
    const int ncb = N >> 8;
    const int id = blockIdx.x, slot = id >> 3, cb = slot % ncb, tile = (slot / ncb) * 8 + (id & 7);
    const int row0 = tile * 128;
    if (row0 >= M) return;

    const __amdgpu_buffer_rsrc_t rsa = uniform_rsrc(A.base + A.tile(tile, 0), A.KT * 24576);
    const int voffa = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    auto dma_tile = [&](int kt, int st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = wave * 4 + q;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (__attribute__((address_space(3))) void*)(smem + st * EG2B_STAGE + piece * 1024), 16, voffa,
                                                     kt * 24576 + (piece >> 3) * 8192 + (piece & 7) * 1024, 0, 0);
        }
    };
    dma_tile(0, 0);
    dma_tile(1, 1);
    dma_tile(2, 2);

    const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(Wf, N * K * 4);
    int voffw[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) voffw[t] = lane * 16 + ((8 * cb + 2 * wave + t) * KS) * 2048;
    u32x4 ring[4][2][2];
#if MI_ASM_LDS >= 2
    // (the weight ring as inline asm too, with a counted wait in front of every step's MFMAs: see MI_RING_WAIT)
    const u32x4 rsw_s = rsrc_words(Wf, N * K * 4);
    auto ring_load = [&](int ks, u32x4 (&w)[2][2]) {
        const int soff = __builtin_amdgcn_readfirstlane(ks * 2048);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(w[t][0]) : "v"(voffw[t]), "s"(rsw_s), "s"(soff));
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(w[t][1]) : "v"(voffw[t]), "s"(rsw_s), "s"(soff));
        }
    };
#else
    auto ring_load = [&](int ks, u32x4 (&w)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) w[t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsw, voffw[t] + pl * 1024, ks * 2048, 0);
    };
#endif
#pragma unroll
    for (int d = 0; d < 4; ++d) ring_load(d, ring[d]);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#if MI_ASM_LDS
    // The fragment reads as inline asm with their own lgkmcnt wait, and the bare s_barrier instead of __syncthreads(): the compiler cannot tell
    // that the LDS-DMA in flight targets other stages than the one being read, so it drains it -- in this loop `s_waitcnt vmcnt(1)` in front of
    // every barrier, i.e. the k-tiles requested two and three iterations ahead had to LAND before the current one was used (found in
    // gemm_tn_planes_kernel, backward.hip, where the same cost 30 %).  The fragments ride through the wait as read-write operands, so that no
    // MFMA can be scheduled in front of it.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned rd_off[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) rd_off[s2] = lds0 + (unsigned)(l31 * 64 + (((2 * s2 + kg) ^ ((l31 >> 2) & 3)) * 16));
    auto read_a = [&](int st, int s2, f16x8 (&af)[4][2]) {
        const unsigned va = rd_off[s2] + (unsigned)st * EG2B_STAGE;
#define MI_RD128(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(va), "n"(off))
        MI_RD128(af[0][0], 0 * 2048);        MI_RD128(af[0][1], 0 * 2048 + 8192);
        MI_RD128(af[1][0], 1 * 2048);        MI_RD128(af[1][1], 1 * 2048 + 8192);
        MI_RD128(af[2][0], 2 * 2048);        MI_RD128(af[2][1], 2 * 2048 + 8192);
        MI_RD128(af[3][0], 3 * 2048);        MI_RD128(af[3][1], 3 * 2048 + 8192);
#undef MI_RD128
        // (no wait here: MI_AF_WAIT in front of the MFMAs that read the fragments -- LDS reads return in order, so the first two row blocks' four
        //  reads are complete when at most four are outstanding)
    };
#else
    auto read_a = [&](int st, int s2, f16x8 (&af)[4][2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 32 + l31, c = (2 * s2 + kg) ^ ((r >> 2) & 3);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[i][pl] = *reinterpret_cast<const f16x8*>(smem + st * EG2B_STAGE + pl * 8192 + r * 64 + c * 16);
        }
    };
#endif
    auto mma = [&](const u32x4 (&w)[2][2], f16x8 (&af)[4][2]) {   // (MI_ASM_LDS: two row blocks at a time, each pair behind the wait for its own fragments)
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
#if MI_ASM_LDS
            if (ip == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[2][0]), "+v"(af[2][1]), "+v"(af[3][0]), "+v"(af[3][1]));
#endif
#pragma unroll
            for (int term = MI_TERM0; term < 3; ++term)
#pragma unroll
                for (int i = 2 * ip; i < 2 * ip + 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if constexpr (LEAN)   // operands swapped: the tile arrives transposed, a lane holds one output ROW (planes_epilogue_lean)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[j][term == 1 ? 1 : 0]), af[i][term == 0 ? 1 : 0], acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][term == 0 ? 1 : 0], __builtin_bit_cast(f16x8, w[j][term == 1 ? 1 : 0]), acc[i][j], 0, 0, 0);
        }
    };
    // EXT (a recorded ablation, off: MI_RT_PF_AT = 0): the epilogue's row-wise operands -- the residual and the second merge as plane sets,
    // the multiplicand as fp32 rows; 128 KB per workgroup each -- are first touched in the epilogue (590 us against 454 us for the same
    // product without them).  Touching one word of each of their 128-byte lines 3 / 5 / 8 k-tiles before the loop ends, so that the
    // epilogue finds them in this XCD's L2, measured 2.54 / 2.52 / 2.52 structures/s against 2.56 on one chain of the MatterGen-shaped
    // sampler and 2.67-2.76 against 2.71-2.79 on four: the epilogue is not waiting for these lines (two workgroups per CU cover each
    // other's loads); the extra time of the EXT form is its arithmetic and its stores.
    unsigned pfv[4][4];   // (raw words, consumed by an empty asm at the kernel's end: nothing waits for them before that)
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) pfv[o][q] = 0;
    int pf_row2 = 0;
    if constexpr (EXT) {
        if (pe.res2_rows && pe.residual2) {
            const int r = row0 + (tid >> 1);
            pf_row2 = r < M ? pe.res2_rows[r] : 0;
        }
    }
    auto prefetch_epilogue_operands = [&]() {
        if constexpr (EXT) {
            const int col0 = cb * 256;
            auto touch_planes = [&](const Planes& P, unsigned (&w)[4]) {   // 8 column tiles x 2 planes x 64 lines (128 rows x 64 B) = 1024 lines
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int line = tid + 256 * q, blk = line >> 6, ct = blk >> 1, plane = blk & 1, ln = line & 63;
                    w[q] = *reinterpret_cast<const unsigned*>(P.base + P.tile(tile, (col0 >> 5) + ct) + (size_t)plane * 4096 + ln * 64);
                }
            };
            auto touch_rows = [&](const float* X, int ld, int row, unsigned (&w)[4]) {   // this thread's half of one row: 4 of its 8 lines
                if (row0 + (tid >> 1) < M)
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[q] = __float_as_uint(X[(size_t)row * ld + col0 + ((tid & 1) * 4 + q) * 32]);
            };
            if (pe.res_pl.base) touch_planes(pe.res_pl, pfv[0]);
            if (pe.res2_pl.base) touch_planes(pe.res2_pl, pfv[1]);
            if (pe.post_mul) touch_rows(pe.post_mul, pe.ld_post_mul, row0 + (tid >> 1), pfv[2]);
            if (pe.residual2) touch_rows(pe.residual2, pe.ld_res2, pe.res2_rows ? pf_row2 : row0 + (tid >> 1), pfv[3]);
        }
    };
    // (waits as in edge_gemm2b_kernel: k-tile k's DMA pieces were issued at least 8 + 12 + 12 vector-memory operations ago)
#pragma unroll 1
    for (int kt = 0; kt < KT; kt += 2) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int k = kt + h2;
            if (k == 0 || k >= KT - 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            MI_LOOP_BARRIER();
            if (k == 0) stamp();
            // (extra loads only put more operations behind a k-tile's DMA pieces: the counted waits stay sufficient)
            if (MI_RT_PF_AT > 0 && k == KT - MI_RT_PF_AT) prefetch_epilogue_operands();
            if (k + 3 < KT) dma_tile(k + 3, (k + 3) & 3);
            f16x8 af[4][2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                read_a(k & 3, s2, af);
                MI_RING_WAIT(ring[2 * h2 + s2]);
                mma(ring[2 * h2 + s2], af);
                if (2 * k + s2 + 4 < KS) ring_load(2 * k + s2 + 4, ring[2 * h2 + s2]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();   // every wave is done with the stages: they become the epilogue's per-wave patches
    stamp();
    if constexpr (LEAN) planes_epilogue_lean<4, 2>(pe, acc, tile, cb * 256 + wave * 64, M, lane);
    else planes_epilogue_rows<4, 2, EXT>(pe, acc, row0, cb * 256 + wave * 64, M, N, lane, reinterpret_cast<float*>(smem) + wave * 1152);
    stamp();
    if (clk && tid == 0) clk[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    if constexpr (EXT && MI_RT_PF_AT > 0) {
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" ::"v"(pfv[o][q]));
    }
}

// The lean launches as a PERSISTENT grid (two workgroups per CU, each walking virtual block ids blockIdx.x, + gridDim.x, ...; the id -> (row tile,
// column block) map is gemm_rt_kernel's, and the stride is a multiple of 8, so a workgroup stays on its XCD's row tiles): the lean epilogue
// uses no LDS, so the NEXT tile's first three k-tiles (LDS-DMA from HBM) are requested before the epilogue starts -- a workgroup of
// gemm_rt_kernel waits ~8 k cycles for those at its start (scripts/rt_phases.py).  The first wait of a tile is vmcnt(0) as before; it now also
// covers the previous tile's plane stores, which were issued after the requests.
#if MI_HAVE_ABLATION_KERNELS   // (the persistent form of the lean launch: did not recover its tail)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_rt_lean_kernel(Planes A, const u16* __restrict__ Wf, int M, int N, int K,
                                                                                                      PlanesEpilogue pe, unsigned long long* __restrict__ clk, int nvb) {
    const int KS = K >> 4, KT = K >> 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    M = pe.rows(M);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int stamp_i = 0;   // phase clock: the FIRST tile's phases (entry / first k-tile in LDS / end of the main loop / end of the epilogue), [5] = exit, [6] = tiles done
    auto stamp = [&]() {
        if (clk && tid == 0 && stamp_i < 4) clk[(size_t)blockIdx.x * 8 + stamp_i] = __builtin_amdgcn_s_memtime();
        ++stamp_i;
    };
    stamp();
    if (clk && tid == 0) clk[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();
    const int l31 = lane & 31, kg = lane >> 5;
    const int ncb = N >> 8;
    int vid = blockIdx.x, tile, cb;
    auto decode = [&](int v) {
        const int slot = v >> 3;
        cb = slot % ncb;
        tile = (slot / ncb) * 8 + (v & 7);
    };
    if (vid >= nvb) return;
    decode(vid);
    if (tile * 128 >= M) return;   // (a workgroup's tiles ascend: nothing behind an empty one)

    __amdgpu_buffer_rsrc_t rsa;
    const int voffa = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    auto dma_tile = [&](int kt, int st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = wave * 4 + q;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (__attribute__((address_space(3))) void*)(smem + st * EG2B_STAGE + piece * 1024), 16, voffa,
                                                     kt * 24576 + (piece >> 3) * 8192 + (piece & 7) * 1024, 0, 0);
        }
    };
    const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(Wf, N * K * 4);
    int voffw[2];
    u32x4 ring[4][2][2];
    auto ring_load = [&](int ks, u32x4 (&w)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) w[t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsw, voffw[t] + pl * 1024, ks * 2048, 0);
    };
    // the first three k-tiles of (tile, cb) -- from HBM, into LDS: no registers -- and its first four weight slices (from L2, 64 registers: those
    // are requested AFTER the epilogue; live across it they spilled 83 registers)
    auto request_a = [&]() {
        rsa = uniform_rsrc(A.base + A.tile(tile, 0), A.KT * 24576);
        dma_tile(0, 0);
        dma_tile(1, 1);
        dma_tile(2, 2);
    };
    auto request_w = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t) voffw[t] = lane * 16 + ((8 * cb + 2 * wave + t) * KS) * 2048;
#pragma unroll
        for (int d = 0; d < 4; ++d) ring_load(d, ring[d]);
    };
    request_a();
    request_w();

    f32x16 acc[4][2];
    auto read_a = [&](int st, int s2, f16x8 (&af)[4][2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 32 + l31, c = (2 * s2 + kg) ^ ((r >> 2) & 3);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[i][pl] = *reinterpret_cast<const f16x8*>(smem + st * EG2B_STAGE + pl * 8192 + r * 64 + c * 16);
        }
    };
    auto mma = [&](const u32x4 (&w)[2][2], const f16x8 (&af)[4][2]) {   // operands swapped: transposed tiles (planes_epilogue_lean)
#pragma unroll
        for (int term = MI_TERM0; term < 3; ++term)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[j][term == 1 ? 1 : 0]), af[i][term == 0 ? 1 : 0], acc[i][j], 0, 0, 0);
    };
    int done = 0;
#pragma unroll 1
    for (;;) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
        for (int kt = 0; kt < KT; kt += 2) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int k = kt + h2;
                if (k == 0 || k >= KT - 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                __syncthreads();
                if (k == 0) stamp();
                if (k + 3 < KT) dma_tile(k + 3, (k + 3) & 3);
                f16x8 af[4][2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    read_a(k & 3, s2, af);
                    mma(ring[2 * h2 + s2], af);
                    if (2 * k + s2 + 4 < KS) ring_load(2 * k + s2 + 4, ring[2 * h2 + s2]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();   // every wave is done with the stages: the next tile's k-tiles may land in them
        stamp();
        const int cur_tile = tile, cur_cb = cb;
        vid += gridDim.x;
        bool more = vid < nvb;
        if (more) {
            decode(vid);
            more = tile * 128 < M;
        }
        if (more) request_a();
        __builtin_amdgcn_sched_barrier(0);   // (the requests go out before the epilogue's first instruction)
        planes_epilogue_lean<4, 2, false>(pe, acc, cur_tile, cur_cb * 256 + wave * 64, M, lane);
        stamp();
        ++done;
        if (!more) break;
        __builtin_amdgcn_sched_barrier(0);
        request_w();
    }
    if (clk && tid == 0) {
        clk[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
        clk[(size_t)blockIdx.x * 8 + 6] = (unsigned long long)done;
        clk[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime();
    }
}
#endif

// plane set [N x K] -> fragment order (exact copy): one thread per (row, 8-k chunk)
__global__ void pack_frag_from_planes_kernel(Planes W, int N, int K, u16* __restrict__ dst) {
    const int KS = K / 16;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * (K / 8)) return;
    const int r = (int)(idx / (K / 8)), ch = (int)(idx % (K / 8));
    const int ct = r >> 5, l31 = r & 31, ks = ch >> 1, kg = ch & 1;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
        *reinterpret_cast<u32x4*>(dst + ((((size_t)ct * KS + ks) * 2 + pl) * 64 + kg * 32 + l31) * 8) = *reinterpret_cast<const u32x4*>(W.base + W.elem(r, ch * 8, pl));
}

// ------------------------------------------------------------------------------------------------------------------------------------
// First edge GEMM, pair mode (see PlanesEpilogue: one operand row per unordered atom pair, the sine half of K into one accumulator set,
// the cosine half into another, both directed edges emitted by the epilogue), in the same form: 128 pairs x 128 columns per four-wave
// workgroup, a wave owns 128 pairs x 32 columns (4 x 1 MFMA tiles x two accumulator sets = 128 registers), the Fourier operand through
// four LDS stages by LDS-DMA, the weights in fragment order straight from L2 into a register ring, two workgroups per CU.  The plane
// GEMM's form of this product staged BOTH operands through registers into LDS (48 KiB of ds_write_b128 and 64 fragment reads per
// k-tile for 96 MFMAs): its main loop ran at 42 % of the matrix pipe's floor.  Epilogue, folded-in activation scales and self-edge
// workgroups are the plane GEMM's (planes_epilogue_pairs / act_scales_eval), so M1 is bit-identical to its output.
// ------------------------------------------------------------------------------------------------------------------------------------
int g_edge1_fused = 9;   // 9 = form b (edge_gemm1b_kernel) for launches beyond the plane GEMM's latency forms, 0 = the plane GEMM always, 1 .. 4 = a form whatever the size.
                         // (Round 3 measured form b equal to the plane GEMM, 191.8 vs 186.6 us at B = 256, and concluded "bound by its epilogue, not by its
                         //  loop"; its loop was being drained by compiler-inserted waits -- MI_ASM_LDS -- and with those gone it is 3-6 % ahead end to end.)

// fragment-order pack of the Fourier block of edge_mlp.0 in the pair-mode column layout [sin block | pad | cos block | pad] (2 Kh columns)
__global__ void pack_frag_wff_pair_kernel(const float* __restrict__ W1, int edge_in, int H, int F, int Kh, u16* __restrict__ dst) {
    const int K = 2 * Kh, KS = K / 16, F3 = 3 * F;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one (row, 8-k chunk) per thread
    if (idx >= (int64_t)H * (K / 8)) return;
    const int r = (int)(idx / (K / 8)), ch = (int)(idx % (K / 8));
    const int ct = r >> 5, l31 = r & 31, ks = ch >> 1, kg = ch & 1;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = ch * 8 + i, blk = c >= Kh, cc = blk ? c - Kh : c;
        v[i] = cc < F3 ? W1[(size_t)r * edge_in + 2 * H + 9 + blk * F3 + cc] : 0.f;
    }
    u32x4 pk[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned p[3];
        pl_split_pair(v[2 * i], v[2 * i + 1], PL_SW, p);
        pk[0][i] = p[0];
        pk[1][i] = p[1];
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<u32x4*>(dst + ((((size_t)ct * KS + ks) * 2 + pl) * 64 + kg * 32 + l31) * 8) = pk[pl];
}

// NJ: 32-column MFMA tiles per wave.  1: 128 pairs x 128 columns per workgroup, two workgroups per CU (128 accumulator registers: two sets
// of 4 x 1 tiles) -- every activation fragment read from LDS feeds ONE column tile, 8 ds_read_b128 per 12 MFMAs: 170 B per cycle and CU,
// above what LDS delivers (128), which is why this form's loop ran at 60 % of the matrix pipe's floor.  2: 128 pairs x 256 columns per
// workgroup, ONE workgroup per CU with 512 registers per lane (two accumulator sets of 4 x 2 tiles = 256): half the LDS reads per MFMA.
// MI: 32-row MFMA tiles per wave.  4: a wave owns all 128 pairs of the tile; 2 (with NJ = 2): the four waves are 2 x 2 -- a wave owns 64 pairs x 64
// columns, the same 128 accumulator registers and the same 128 x 128 workgroup tile as (4, 1), but each activation fragment feeds TWO column
// tiles (4 LDS reads per 12 MFMAs instead of 8) at the price of each weight fragment being fetched by two waves (L2 -> CU traffic doubles).
template <int D, int MI, int NJ, bool WIDE = true>
__device__ __forceinline__ void edge_gemm1_body(const Planes& A, const u16* __restrict__ Wf, int M, int N, int K, PlanesEpilogue& pe, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    constexpr int WM = 4 / MI, WN = 4 / WM, WGC = 32 * NJ * WN;   // wave rows x wave columns of the workgroup, its columns
    const int KT = K / 32, KS = K / 16, nq = N / WGC;   // k-tiles, k-steps, column blocks of a row tile
    const int wm = wave / WN, wn = wave % WN;
    const int id = blockIdx.x;
    // this layer's M1 scale from the absmax slots and the weight bounds; workgroup 0 publishes all six scales (as the plane GEMM does)
    float cps_local = 0.f;
    auto eval_scales = [&]() {
        if (pe.sc_pq) {
            float dsc[6];
            act_scales_eval(__uint_as_float(pe.sc_pq[0]), __uint_as_float(pe.sc_gmax[0]), pe.sc_wb, dsc);
            cps_local = dsc[0];
            if (id == 0 && tid < 6) {
                pe.sc_dsc[tid] = dsc[tid];
                if (pe.sc_dsc2) pe.sc_dsc2[tid] = dsc[tid];
            }
        }
    };
    // (evaluated in front of the operand loads: behind them -- so that its four dependent scalar loads would run under the LDS-DMA latency --
    //  measured 2 k cycles SLOWER per workgroup, 4.9 k -> 6.9 k to the first barrier: profiles/r5_index_loads_ab.log)
    eval_scales();
    if (pe.diag_C0 && id >= pe.diag_block0) {  // self edges: eight nodes per workgroup, a thread per column pair (d = 0: the Fourier term is C0)
        const float cps = cps_local != 0.f ? cps_local : pe.Cp.s();
        const int n0 = (id - pe.diag_block0) * 8, n1 = n0 + 8 < pe.diag_nodes ? n0 + 8 : pe.diag_nodes;
        const float* PQ = pe.ep.row_bias;
        const int ldpq = pe.ep.ld_row_bias;
        for (int f = 2 * tid; f < N; f += 512) {
            const float c0 = pe.diag_C0[f], c1 = pe.diag_C0[f + 1];
            for (int i = n0; i < n1; ++i) {
                const int e = pe.diag_e[i], g = pe.diag_node2graph[i];
                const float* G = pe.ep.row_bias3 + (size_t)g * pe.ep.ld_row_bias3;
                const float v0 = c0 + ((PQ[(size_t)i * ldpq + f] + PQ[(size_t)i * ldpq + N + f]) + G[f]);
                const float v1 = c1 + ((PQ[(size_t)i * ldpq + f + 1] + PQ[(size_t)i * ldpq + N + f + 1]) + G[f + 1]);
                if (pe.ep.pre_act) {   // (training forward: the self edges' pre-activation is on the tape too)
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
    // XCD-aware mapping: the column quarters of a row tile run on the same XCD
    const int slot = id >> 3, qt = slot % nq, tile = (slot / nq) * 8 + (id & 7);
    const int row0 = tile * 128;
    if (row0 >= M) return;
#ifdef MI_E1_STAGGER   // (experiment: the workgroups that fill the CUs' SECOND slots start this many cycles late, so that co-resident workgroups are out of phase)
    if ((id >> 8) & 1) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)(MI_E1_STAGGER)) __builtin_amdgcn_s_sleep(64);
    }
#endif
    int stamp_i = 0;
    auto stamp = [&]() {
        if (clk && tid == 0) clk[(size_t)(tile * nq + qt) * 8 + stamp_i] = __builtin_amdgcn_s_memtime();
        ++stamp_i;
    };
    stamp();

    const __amdgpu_buffer_rsrc_t rsa = uniform_rsrc(A.base + A.tile(tile, 0), A.KT * 24576);
    const int voffa = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
#ifndef MI_DBG_E1_LOOP
#define MI_DBG_E1_LOOP 0   // timing diagnostics (wrong results): 1 = no MFMAs, 2 = no LDS-DMA of the Fourier operand, 4 = no weight ring loads, 8 = no LDS fragment reads
#endif
    auto dma_tile = [&](int kt, int st) {
        if constexpr ((MI_DBG_E1_LOOP & 2) != 0) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = wave * 4 + q;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (__attribute__((address_space(3))) void*)(smem + st * EG2B_STAGE + piece * 1024), 16, voffa,
                                                     kt * 24576 + (piece >> 3) * 8192 + (piece & 7) * 1024, 0, MI_DMA_AUX_FF);
        }
    };
    dma_tile(0, 0);
    dma_tile(1, 1);
    dma_tile(2, 2);
    const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(Wf, N * K * 4);
    int voffw[NJ];   // the wave's column tiles: columns WGC qt + 32 NJ wn + 32 t .. + 31
#pragma unroll
    for (int t = 0; t < NJ; ++t) voffw[t] = lane * 16 + (((WN * qt + wn) * NJ + t) * KS) * 2048;
    u32x4 ring[D][NJ][2];
    // (MI_ASM_LDS >= 2, the 4 x 1 form: every memory operation of the k-loop under manual control, as in edge_gemm2b_kernel -- a weight slice is two
    //  loads here, so 6 ring loads and 8 LDS-DMA pieces are issued behind one before its use: vmcnt(14))
    constexpr bool MAN = MI_ASM_LDS >= 2 && MI == 4 && NJ == 1;
    const u32x4 rsw_s = rsrc_words(Wf, N * K * 4);
    auto ring_load = [&](int ks, u32x4 (&w)[NJ][2]) {
        if constexpr ((MI_DBG_E1_LOOP & 4) != 0) {
            asm volatile("" : "+v"(w[0][0]), "+v"(w[0][1]));
            return;
        }
        if constexpr (MAN) {
            const int soff = __builtin_amdgcn_readfirstlane(ks * 2048);
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(w[0][0]) : "v"(voffw[0]), "s"(rsw_s), "s"(soff));
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(w[0][1]) : "v"(voffw[0]), "s"(rsw_s), "s"(soff));
        } else {
#pragma unroll
            for (int t = 0; t < NJ; ++t)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) w[t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsw, voffw[t] + pl * 1024, ks * 2048, 0);
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) ring_load(d, ring[d]);

    f32x16 acc[MI][NJ], accS[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned rd_off[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) rd_off[s2] = lds0 + (unsigned)(l31 * 64 + (((2 * s2 + kg) ^ ((l31 >> 2) & 3)) * 16));
    auto read_a = [&](int st, int s2, f16x8 (&af)[MI][2]) {
        if constexpr ((MI_DBG_E1_LOOP & 8) != 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(af[i][0]), "+v"(af[i][1]));
            return;
        }
        if constexpr (MAN) {
            const unsigned va = rd_off[s2] + (unsigned)st * EG2B_STAGE;
#define MI_RD128(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(va), "n"(off))
            MI_RD128(af[0][0], 0 * 2048);        MI_RD128(af[0][1], 0 * 2048 + 8192);
            MI_RD128(af[1][0], 1 * 2048);        MI_RD128(af[1][1], 1 * 2048 + 8192);
            MI_RD128(af[2][0], 2 * 2048);        MI_RD128(af[2][1], 2 * 2048 + 8192);
            MI_RD128(af[3][0], 3 * 2048);        MI_RD128(af[3][1], 3 * 2048 + 8192);
#undef MI_RD128
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = (wm * MI + i) * 32 + l31, c = (2 * s2 + kg) ^ ((r >> 2) & 3);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) af[i][pl] = *reinterpret_cast<const f16x8*>(smem + st * EG2B_STAGE + pl * 8192 + r * 64 + c * 16);
            }
        }
    };
    auto mma = [&](u32x4 (&w)[NJ][2], f16x8 (&af)[MI][2]) {   // terms (a1, b0), (a0, b1), (a0, b0)
        if constexpr (MAN) {
            asm volatile("s_waitcnt vmcnt(14)" : "+v"(w[0][0]), "+v"(w[0][1]));
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                if (ip == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[2][0]), "+v"(af[2][1]), "+v"(af[3][0]), "+v"(af[3][1]));
#pragma unroll
                for (int term = (MI_DBG_E1_LOOP & 1) ? 3 : MI_TERM0; term < 3; ++term)
#pragma unroll
                    for (int i = 2 * ip; i < 2 * ip + 2; ++i)
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][term == 0 ? 1 : 0], __builtin_bit_cast(f16x8, w[0][term == 1 ? 1 : 0]), acc[i][0], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int term = MI_TERM0; term < 3; ++term)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][term == 0 ? 1 : 0], __builtin_bit_cast(f16x8, w[j][term == 1 ? 1 : 0]), acc[i][j], 0, 0, 0);
        }
    };
    static_assert(D == 4, "two k-tiles of ring per unrolled pair of iterations");
    const int khalf = KT / 2;
    // (vmcnt: per k-tile this wave issues 4 DMA pieces and then 4 ring loads; k-tile kt's pieces were issued three iterations ago, i.e. at
    //  least 4 + 8 + 8 operations ago: vmcnt(16) leaves the younger ones in flight; the last three k-tiles drain)
#pragma unroll 1
    for (int kt = 0; kt < KT; kt += 2) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int k = kt + h2;
            if (k == khalf) {   // the cosine half of K goes into the second accumulator set
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        accS[i][j] = acc[i][j];
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                    }
            }
            if (k == 0 || k >= KT - 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (NJ == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");   // (4 DMA pieces + 8 ring loads per k-tile, as in edge_gemm2b_kernel)
            if constexpr (MAN) __builtin_amdgcn_s_barrier();
            else __syncthreads();
            if (k == 0) stamp();
            if (k + 3 < KT) dma_tile(k + 3, (k + 3) & 3);
            f16x8 af[MI][2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                read_a(k & 3, s2, af);
                mma(ring[2 * h2 + s2], af);
                if (2 * k + s2 + D < KS) ring_load(2 * k + s2 + D, ring[2 * h2 + s2]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    stamp();
    __syncthreads();   // the epilogue's per-wave patches overlay the operand stages
    planes_epilogue_pairs<MI, NJ, WIDE>(pe, accS, acc, row0 + wm * MI * 32, qt * WGC + wn * 32 * NJ, M, N, lane, reinterpret_cast<float*>(smem) + wave * 2304, cps_local);
    stamp();
}

template <int D, bool WIDE = true>   // (WIDE: the pair epilogue's 64-bit addressing, see PlanesEpilogue::pair_wide)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void edge_gemm1b_kernel(Planes A, const u16* __restrict__ Wf, int M, int N, int K,
                                                                                                     PlanesEpilogue pe, unsigned long long* clk) {
    edge_gemm1_body<D, 4, 1, WIDE>(A, Wf, M, N, K, pe, clk);
}
#if MI_HAVE_ABLATION_KERNELS   // (2 x 2 waves of 64 x 64: 256 registers and 64 bytes of scratch; measured 12 % SLOWER end to end -- every weight fragment is fetched by two waves)
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void edge_gemm1d_kernel(Planes A, const u16* __restrict__ Wf, int M, int N, int K,
                                                                                                     PlanesEpilogue pe, unsigned long long* clk) {
    edge_gemm1_body<D, 2, 2>(A, Wf, M, N, K, pe, clk);
}
#endif
#if MI_HAVE_ABLATION_KERNELS   // (512 registers and 36 bytes of scratch; measured 11-13 % SLOWER end to end than the plane GEMM: DESIGN 16.3a)
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void edge_gemm1c_kernel(Planes A, const u16* __restrict__ Wf, int M, int N, int K,
                                                                                                     PlanesEpilogue pe, unsigned long long* clk) {
    edge_gemm1_body<D, 4, 2>(A, Wf, M, N, K, pe, clk);
}
#endif

// ------------------------------------------------------------------------------------------------------------------------------------
// Form E (mi_debug_set_edge1_fused(4)): ONE accumulator set on 128 pairs x 256 columns per four-wave workgroup -- the second edge GEMM's
// register tile (a wave owns 128 pairs x 64 columns: 8 LDS fragment reads per 24 MFMAs, the ratio whose loop runs at the matrix pipe's
// rate, where the two-set forms above read 8 per 12 and run at 1.7 x its floor).  The price of one set is half a pass more of MFMA work:
//   pass 1  acc  = [sin | cos] x [Wsin ; Wcos]   (the whole K)      = C + S   -> epilogue of direction i -> j
//   pass 2  acc += sin x (-2 Wsin)               (the sine half again) = C - S -> epilogue of direction j -> i
// (-2 Wsin is exact in the fp16 plane format and packed beside the weights, edge_gemm1_pack).  1.5 x the MFMAs at the pipe's rate against
// 1 x at 1.7 x its floor; the epilogue work per emitted edge is unchanged.  Same products up to the order of the fp32 accumulation: M1
// agrees with the two-set forms to fp32 round-off, not bit for bit.
// ------------------------------------------------------------------------------------------------------------------------------------
#if MI_HAVE_ABLATION_KERNELS   // (form E -- one accumulator set, the sine half twice: measured no faster than form b)
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void edge_gemm1e_kernel(Planes A, const u16* __restrict__ Wf, const u16* __restrict__ Wf2, int M,
                                                                                                     int N, int K, PlanesEpilogue pe, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    const int nq = N / 256;
    const int id = blockIdx.x;
    float cps_local = 0.f;
    if (pe.sc_pq) {
        float dsc[6];
        act_scales_eval(__uint_as_float(pe.sc_pq[0]), __uint_as_float(pe.sc_gmax[0]), pe.sc_wb, dsc);
        cps_local = dsc[0];
        if (id == 0 && tid < 6) {
            pe.sc_dsc[tid] = dsc[tid];
            if (pe.sc_dsc2) pe.sc_dsc2[tid] = dsc[tid];
        }
    }
    if (pe.diag_C0 && id >= pe.diag_block0) {  // self edges, as in edge_gemm1_body
        const float cps = cps_local != 0.f ? cps_local : pe.Cp.s();
        const int n0 = (id - pe.diag_block0) * 8, n1 = n0 + 8 < pe.diag_nodes ? n0 + 8 : pe.diag_nodes;
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
    const int slot = id >> 3, qt = slot % nq, tile = (slot / nq) * 8 + (id & 7);   // the column halves of a row tile on one XCD
    const int row0 = tile * 128;
    if (row0 >= M) return;
    int stamp_i = 0;
    auto stamp = [&]() {
        if (clk && tid == 0) clk[(size_t)(tile * nq + qt) * 8 + stamp_i] = __builtin_amdgcn_s_memtime();
        ++stamp_i;
    };
    stamp();
    const __amdgpu_buffer_rsrc_t rsa = uniform_rsrc(A.base + A.tile(tile, 0), A.KT * 24576);
    const int voffa = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    auto dma_tile = [&](int kt, int st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = wave * 4 + q;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (__attribute__((address_space(3))) void*)(smem + st * EG2B_STAGE + piece * 1024), 16, voffa,
                                                     kt * 24576 + (piece >> 3) * 8192 + (piece & 7) * 1024, 0, 0);
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#if MI_ASM_LDS >= 2   // (every memory operation of the k-loop under manual control, as in edge_gemm2b_kernel: see MI_ASM_LDS)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned rd_off[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) rd_off[s2] = lds0 + (unsigned)(l31 * 64 + (((2 * s2 + kg) ^ ((l31 >> 2) & 3)) * 16));
    auto read_a = [&](int st, int s2, f16x8 (&af)[4][2]) {
        const unsigned va = rd_off[s2] + (unsigned)st * EG2B_STAGE;
#define MI_RD128(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(va), "n"(off))
        MI_RD128(af[0][0], 0 * 2048);        MI_RD128(af[0][1], 0 * 2048 + 8192);
        MI_RD128(af[1][0], 1 * 2048);        MI_RD128(af[1][1], 1 * 2048 + 8192);
        MI_RD128(af[2][0], 2 * 2048);        MI_RD128(af[2][1], 2 * 2048 + 8192);
        MI_RD128(af[3][0], 3 * 2048);        MI_RD128(af[3][1], 3 * 2048 + 8192);
#undef MI_RD128
    };
    auto mma = [&](const u32x4 (&w)[2][2], f16x8 (&af)[4][2]) {   // terms (a1, b0), (a0, b1), (a0, b0); two row blocks at a time behind the wait for their fragments
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            if (ip == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[2][0]), "+v"(af[2][1]), "+v"(af[3][0]), "+v"(af[3][1]));
#pragma unroll
            for (int term = MI_TERM0; term < 3; ++term)
#pragma unroll
                for (int i = 2 * ip; i < 2 * ip + 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][term == 0 ? 1 : 0], __builtin_bit_cast(f16x8, w[j][term == 1 ? 1 : 0]), acc[i][j], 0, 0, 0);
        }
    };
#else
    auto read_a = [&](int st, int s2, f16x8 (&af)[4][2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 32 + l31, c = (2 * s2 + kg) ^ ((r >> 2) & 3);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[i][pl] = *reinterpret_cast<const f16x8*>(smem + st * EG2B_STAGE + pl * 8192 + r * 64 + c * 16);
        }
    };
    auto mma = [&](const u32x4 (&w)[2][2], const f16x8 (&af)[4][2]) {   // terms (a1, b0), (a0, b1), (a0, b0)
#pragma unroll
        for (int term = MI_TERM0; term < 3; ++term)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][term == 0 ? 1 : 0], __builtin_bit_cast(f16x8, w[j][term == 1 ? 1 : 0]), acc[i][j], 0, 0, 0);
    };
#endif
    static_assert(D == 4, "two k-tiles of ring per unrolled pair of iterations");
    // one pass over the k-tiles 0 .. KTp - 1 of the Fourier operand against the fragment-order weights Wp (KSp = 2 KTp k-steps per column tile);
    // the pipeline of edge_gemm2b_kernel: DMA three k-tiles ahead, four-deep weight ring, counted waits
    auto run_pass = [&](const u16* __restrict__ Wp, int KTp) {
        const int KSp = 2 * KTp;
        const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(Wp, N * KSp * 16 * 4);
        int voffw[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) voffw[t] = lane * 16 + ((8 * qt + 2 * wave + t) * KSp) * 2048;
        u32x4 ring[D][2][2];
#if MI_ASM_LDS >= 2
        const u32x4 rsw_s = rsrc_words(Wp, N * KSp * 16 * 4);
        auto ring_load = [&](int ks, u32x4 (&w)[2][2]) {
            const int soff = __builtin_amdgcn_readfirstlane(ks * 2048);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(w[t][0]) : "v"(voffw[t]), "s"(rsw_s), "s"(soff));
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(w[t][1]) : "v"(voffw[t]), "s"(rsw_s), "s"(soff));
            }
        };
#else
        auto ring_load = [&](int ks, u32x4 (&w)[2][2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) w[t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsw, voffw[t] + pl * 1024, ks * 2048, 0);
        };
#endif
        dma_tile(0, 0);
        dma_tile(1, 1);
        dma_tile(2, 2);
#pragma unroll
        for (int d = 0; d < D; ++d) ring_load(d, ring[d]);
#pragma unroll 1
        for (int kt = 0; kt < KTp; kt += 2) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int k = kt + h2;
                if (k == 0 || k >= KTp - 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");   // (4 DMA pieces + 8 ring loads per k-tile, as in edge_gemm2b_kernel)
                MI_LOOP_BARRIER();
                if (k + 3 < KTp) dma_tile(k + 3, (k + 3) & 3);
                f16x8 af[4][2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    read_a(k & 3, s2, af);
                    MI_RING_WAIT(ring[2 * h2 + s2]);
                    mma(ring[2 * h2 + s2], af);
                    if (2 * k + s2 + D < KSp) ring_load(2 * k + s2 + D, ring[2 * h2 + s2]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    float* patch = reinterpret_cast<float*>(smem) + wave * 1152;
    const int KT = K / 32;
    run_pass(Wf, KT);                                   // C + S
    stamp();
    __syncthreads();                                    // the epilogue's per-wave patches overlay the operand stages
    planes_epilogue_pairs_dir<4, 2>(pe, acc, 0, row0, qt * 256 + wave * 64, M, N, lane, patch, cps_local);
    stamp();
    __syncthreads();                                    // every wave is done with its patch: the stages take operand tiles again
    run_pass(Wf2, KT / 2);                              // ... - 2 S
    stamp();
    __syncthreads();
    planes_epilogue_pairs_dir<4, 2>(pe, acc, 1, row0, qt * 256 + wave * 64, M, N, lane, patch, cps_local);
    stamp();
}
#endif

// fragment-order pack of -2 x the SINE block of the Fourier weights (Kh columns): the second pass of edge_gemm1e_kernel
__global__ void pack_frag_wff_sin_neg2_kernel(const float* __restrict__ W1, int edge_in, int H, int F, int Kh, u16* __restrict__ dst) {
    const int KS = Kh / 16, F3 = 3 * F;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one (row, 8-k chunk) per thread
    if (idx >= (int64_t)H * (Kh / 8)) return;
    const int r = (int)(idx / (Kh / 8)), ch = (int)(idx % (Kh / 8));
    const int ct = r >> 5, l31 = r & 31, ks = ch >> 1, kg = ch & 1;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = ch * 8 + i;
        v[i] = c < F3 ? -2.f * W1[(size_t)r * edge_in + 2 * H + 9 + c] : 0.f;
    }
    u32x4 pk[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned p[3];
        pl_split_pair(v[2 * i], v[2 * i + 1], PL_SW, p);
        pk[0][i] = p[0];
        pk[1][i] = p[1];
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<u32x4*>(dst + ((((size_t)ct * KS + ks) * 2 + pl) * 64 + kg * 32 + l31) * 8) = pk[pl];
}

unsigned long long* g_edge1_clk = nullptr;

// (timing experiment, mi_debug_set_skip bit 5: an EMPTY launch in front of every edge GEMM -- 24 more kernel boundaries per step on a chain's serial path and no
//  work: what a boundary costs the four-chain step, i.e. what removing launches from the path can buy: DESIGN 19.9)
__global__ void empty_boundary_kernel() {}

int edge_gemm1(mi_net* net, const Planes& A, int layer, int M, PlanesEpilogue pe, hipStream_t s) {
    if (g_ablate_skip & 2) return MI_OK;
    if (g_ablate_skip & 32) hipLaunchKernelGGL(empty_boundary_kernel, dim3(1), dim3(64), 0, s);
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)edge_gemm1b_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2B_LDS);
        if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)edge_gemm1b_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2B_LDS);
#if MI_HAVE_ABLATION_KERNELS
        if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)edge_gemm1d_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2B_LDS);
        if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)edge_gemm1c_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2B_LDS);
#endif
    });
    MI_HIP(attr_err);
    const int H = net->H, K = 2 * net->Kh;
    count_mfma(M, H, K, MI_PLANES_TERMS);   // (pair rows x [sine block | cosine block] of the Fourier columns)
    pe.out_scale = 1.f / (A.scale * PL_SW);
#if MI_HAVE_ABLATION_KERNELS
    if (g_edge1_fused == 4 && H % 256 == 0 && net->Wffc2 && (K / 32) % 4 == 0) {   // form E: one accumulator set, 128 x 256 tiles, the sine half twice
        static std::once_flag once_e;
        static hipError_t attr_e = hipSuccess;
        std::call_once(once_e, [] { attr_e = hipFuncSetAttribute((const void*)edge_gemm1e_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2B_LDS); });
        MI_HIP(attr_e);
        int nblk = (H / 256) * ((cdiv(M, 128) + 7) / 8 * 8);
        if (pe.diag_C0) {
            pe.diag_block0 = nblk;
            nblk += cdiv(pe.diag_nodes, 8);
        }
        hipLaunchKernelGGL((edge_gemm1e_kernel<4>), dim3(nblk), dim3(256), EG2B_LDS, s, A, net->Wffc + (size_t)layer * ((size_t)H * K * 2),
                           net->Wffc2 + (size_t)layer * ((size_t)H * net->Kh * 2), M, H, K, pe, g_edge1_clk);
        MI_KERNEL_CHECK();
        return MI_OK;
    }
#endif
    const bool wide = MI_HAVE_ABLATION_KERNELS && g_edge1_fused == 2 && H % 256 == 0;   // 128 x 256 tiles, one four-wave workgroup per CU with 512 registers per lane
    int nblk = (H / (wide ? 256 : 128)) * ((cdiv(M, 128) + 7) / 8 * 8);
    if (pe.diag_C0) {
        pe.diag_block0 = nblk;
        nblk += cdiv(pe.diag_nodes, 8);
    }
    const u16* Wf = net->Wffc + (size_t)layer * ((size_t)H * K * 2);
#if MI_HAVE_ABLATION_KERNELS
    if (wide) hipLaunchKernelGGL((edge_gemm1c_kernel<4>), dim3(nblk), dim3(256), EG2B_LDS, s, A, Wf, M, H, K, pe, g_edge1_clk);
    else if (g_edge1_fused == 3) hipLaunchKernelGGL((edge_gemm1d_kernel<4>), dim3(nblk), dim3(256), EG2B_LDS, s, A, Wf, M, H, K, pe, g_edge1_clk);   // 2 x 2 waves of 64 x 64
    else
#endif
    if (pe.pair_wide) hipLaunchKernelGGL((edge_gemm1b_kernel<4>), dim3(nblk), dim3(256), EG2B_LDS, s, A, Wf, M, H, K, pe, g_edge1_clk);
    else hipLaunchKernelGGL((edge_gemm1b_kernel<4, false>), dim3(nblk), dim3(256), EG2B_LDS, s, A, Wf, M, H, K, pe, g_edge1_clk);
    MI_KERNEL_CHECK();
    return MI_OK;
}

// whether layer launches of M pair rows take edge_gemm1: a forced form (1 .. 4: tests, ablations) whatever the size; by default (9) form b for the
// launches beyond the plane GEMM's latency forms -- with its k-loop under manual control it beats the plane GEMM there
bool edge_gemm1_supported(const mi_net* net, int64_t M) {
    if (!(net->H % 128 == 0 && net->Wffc != nullptr && (2 * net->Kh) % 64 == 0) || g_edge1_fused == 0) return false;
    if (g_edge1_fused != 9) return true;
    return (int64_t)(net->H / 128) * ((cdiv(M, 128) + 7) / 8 * 8) > g_planes_lat_max_blocks;
}

int edge_gemm1_pack(mi_net* net, int l, const float* W1, hipStream_t s) {
    const int H = net->H, K = 2 * net->Kh;
    hipLaunchKernelGGL(pack_frag_wff_pair_kernel, dim3(cdiv((int64_t)H * (K / 8), 256)), dim3(256), 0, s, W1, net->edge_in, H, net->F, net->Kh,
                       net->Wffc + (size_t)l * ((size_t)H * K * 2));
    if (net->Wffc2)
        hipLaunchKernelGGL(pack_frag_wff_sin_neg2_kernel, dim3(cdiv((int64_t)H * (net->Kh / 8), 256)), dim3(256), 0, s, W1, net->edge_in, H, net->F, net->Kh,
                           net->Wffc2 + (size_t)l * ((size_t)H * net->Kh * 2));
    MI_KERNEL_CHECK();
    return MI_OK;
}

unsigned long long* g_edge2_clk = nullptr;

int edge_gemm2(mi_net* net, mi_batch* b, int layer, hipStream_t s, float* Z2) {
    if (g_ablate_skip & 4) return MI_OK;
    if (g_ablate_skip & 32) hipLaunchKernelGGL(empty_boundary_kernel, dim3(1), dim3(64), 0, s);
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
#if MI_HAVE_ABLATION_KERNELS
        attr_err = hipFuncSetAttribute((const void*)edge_gemm2_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2_LDS);
        if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)edge_gemm2_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2_LDS);
#endif
        if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)edge_gemm2b_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2B_LDS);
        if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)edge_gemm2b_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2B_LDS);
    });
    MI_HIP(attr_err);
    const int H = net->H;
    count_mfma(b->E, H, H, MI_PLANES_TERMS);
    EdgeGemm2Args a;
    a.A = make_planes(b->m1_cur ? b->m1_cur : b->M1pl, H, PL_S_ACT, b->dsc);
    a.W2f = net->Wnc + (size_t)layer * node_chain_pack_elems(H) + (size_t)5 * H * H * 2;
    a.b2 = net->p("csp_layer_" + std::to_string(layer) + ".edge_mlp.2.bias");
    a.dsc = b->dsc;
    a.src = b->src;
    a.rowptr = b->rowptr;
    a.part = b->part;
    a.E = (int)b->E;
    a.N = b->N;
    {   // the per-tile tables of this graph (fc: built once per batch handle; knn: the forward that rebuilt the edge list bumped graph_epoch)
        const int tiles_cap = cdiv(b->E_cap, 128);
        if (!b->e2_tab) MI_TRY(dev_alloc(b, &b->e2_tab, (size_t)tiles_cap * EG2_TAB));
        if (b->e2_tab_epoch != b->graph_epoch) {
            hipLaunchKernelGGL(edge2_tables_kernel, dim3(cdiv(b->E, 128)), dim3(128), 0, s, b->src, b->rowptr, (int)b->E, b->N, b->e2_tab);
            MI_KERNEL_CHECK();
            b->e2_tab_epoch = b->graph_epoch;
        }
        a.tab = b->e2_tab;
    }
    a.Z2 = Z2;
    a.clk = g_edge2_clk;
    if (Z2) {   // the training forward: form B with the pre-activation kept (written row-major through per-wave LDS patches)
        hipLaunchKernelGGL((edge_gemm2b_kernel<4, true>), dim3(2 * ((cdiv(b->E, 128) + 7) / 8 * 8)), dim3(256), EG2B_LDS, s, a);
        MI_KERNEL_CHECK();
        return MI_OK;
    }
    // 1 (default): form B -- 128 x 256 tiles, four waves, two workgroups per CU; 2 / 3: the eight-wave 128 x 512 form (ablations)
#if MI_HAVE_ABLATION_KERNELS
    if (g_edge2_fused == 2) hipLaunchKernelGGL((edge_gemm2_kernel<2, 2>), dim3(cdiv(b->E, 128)), dim3(512), EG2_LDS, s, a);
    else if (g_edge2_fused == 3) hipLaunchKernelGGL((edge_gemm2_kernel<4, 1>), dim3(cdiv(b->E, 128)), dim3(512), EG2_LDS, s, a);
    else
#endif
    hipLaunchKernelGGL((edge_gemm2b_kernel<4>), dim3(2 * ((cdiv(b->E, 128) + 7) / 8 * 8)), dim3(256), EG2B_LDS, s, a);
    MI_KERNEL_CHECK();
    return MI_OK;
}

bool edge_gemm2_supported(const mi_net* net) { return g_edge2_fused && net->H == 512 && net->Wnc != nullptr; }

unsigned long long* g_rt_clk = nullptr;   // phase clock of gemm_rt launches (mi_debug_rt_clock): [workgroup][8]
int g_rt_clk_ext = -1;                    // -1: every launch writes it (the last one stays); 0 / 1: launches of the plain / the extended epilogue only
int g_rt_lean_grid = 512;                 // workgroups of the persistent form (two per CU; a multiple of 8)
int g_rt_lean = 1;                        // the lean epilogue for the launches that qualify (planes_epilogue_is_lean); 0: the general one for all
int g_rt_clk_skip = 0;                    // matching launches to let pass before the one that is clocked (then the clock switches itself off)

int gemm_rt(const Planes& A, const u16* Wfrag, int M, int N, int K, const PlanesEpilogue& pe, bool ext, hipStream_t s) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)gemm_rt_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2B_NST * EG2B_STAGE);
        if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)gemm_rt_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2B_NST * EG2B_STAGE);
        if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)gemm_rt_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, EG2B_NST * EG2B_STAGE);
#if MI_HAVE_ABLATION_KERNELS
        if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)gemm_rt_lean_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, EG2B_NST * EG2B_STAGE);
#endif
        if (const char* e = getenv("MI_RT_LEAN")) g_rt_lean = atoi(e) & 3;   // (A/B runs of whole test files: scripts/gpu_rt_lean_ab.sh)
    });
    MI_HIP(attr_err);
    MI_CHECK(Wfrag && (N & 255) == 0 && (K & 63) == 0 && K >= 128 && A.KT >= K / 32, MI_EINVAL, "gemm_rt: N % 256, K % 64, K >= 128 and a fragment-order W operand");
    if (M <= 0) return MI_OK;
    count_mfma(M, N, K, MI_PLANES_TERMS);
    const dim3 grid((N >> 8) * ((cdiv(M, 128) + 7) / 8 * 8));
    static const bool trace = getenv("MI_RT_TRACE") != nullptr;   // (debugging aid: which epilogue features each launch carries)
    if (trace)
        fprintf(stderr, "gemm_rt M=%d N=%d K=%d ext=%d act=%d bias=%d rb=%d%d%d pre_add=%d pre_act=%d res=%d res_pl=%d os=%g res2=%d res2_pl=%d rows2=%d post_mul=%d C=%d Cp=%d absmax=%d\n", M, N, K,
                (int)ext, pe.ep.act, !!pe.ep.bias, !!pe.ep.row_bias, !!pe.ep.row_bias2, !!pe.ep.row_bias3, !!pe.ep.pre_add, !!pe.ep.pre_act, !!pe.ep.residual, !!pe.res_pl.base,
                (double)pe.ep.out_scale, !!pe.residual2, !!pe.res2_pl.base, !!pe.res2_rows, !!pe.post_mul, !!pe.C, !!pe.Cp.base, !!pe.absmax);
    unsigned long long* clk = nullptr;
    if (g_rt_clk && (g_rt_clk_ext < 0 || g_rt_clk_ext == (int)ext)) {
        if (g_rt_clk_skip > 0) {
            --g_rt_clk_skip;
        } else {
            clk = g_rt_clk;
            if (g_rt_clk_ext >= 0) g_rt_clk = nullptr;   // one launch
        }
    }
#if MI_HAVE_ABLATION_KERNELS
    if (g_rt_lean >= 2 && planes_epilogue_is_lean(pe, false))
        hipLaunchKernelGGL(gemm_rt_lean_kernel, dim3(std::min<unsigned>(grid.x, (unsigned)g_rt_lean_grid)), dim3(256), EG2B_NST * EG2B_STAGE, s, A, Wfrag, M, N, K, pe, clk, (int)grid.x);
    else
#endif
    if (g_rt_lean && planes_epilogue_is_lean(pe)) hipLaunchKernelGGL((gemm_rt_kernel<false, true>), grid, dim3(256), EG2B_NST * EG2B_STAGE, s, A, Wfrag, M, N, K, pe, clk);
    else if (ext) hipLaunchKernelGGL(gemm_rt_kernel<true>, grid, dim3(256), EG2B_NST * EG2B_STAGE, s, A, Wfrag, M, N, K, pe, clk);
    else hipLaunchKernelGGL(gemm_rt_kernel<false>, grid, dim3(256), EG2B_NST * EG2B_STAGE, s, A, Wfrag, M, N, K, pe, clk);
    MI_KERNEL_CHECK();
    return MI_OK;
}

size_t frag_elems(int N, int K) { return (size_t)N * K * 2; }

int pack_frag_from_planes(const Planes& W, int N, int K, u16* dst, hipStream_t s) {
    MI_CHECK((N & 31) == 0 && (K & 15) == 0, MI_EINVAL, "pack_frag_from_planes: N % 32, K % 16");
    hipLaunchKernelGGL(pack_frag_from_planes_kernel, dim3((unsigned)cdiv((int64_t)N * (K / 8), 256)), dim3(256), 0, s, W, N, K, dst);
    MI_KERNEL_CHECK();
    return MI_OK;
}

#else

int gemm_rt(const Planes&, const u16*, int, int, int, const PlanesEpilogue&, bool, hipStream_t) { return MI_ESTATE; }
size_t frag_elems(int N, int K) { return (size_t)N * K * 2; }
int pack_frag_from_planes(const Planes&, int, int, u16*, hipStream_t) { return MI_ESTATE; }

int edge_gemm2(mi_net*, mi_batch*, int, hipStream_t, float*) { return MI_ESTATE; }
bool edge_gemm2_supported(const mi_net*) { return false; }
int edge_gemm1(mi_net*, const Planes&, int, int, PlanesEpilogue, hipStream_t) { return MI_ESTATE; }
bool edge_gemm1_supported(const mi_net*, int64_t) { return false; }
int edge_gemm1_pack(mi_net*, int, const float*, hipStream_t) { return MI_OK; }
int g_edge1_fused = 0;

#endif

}  // namespace mi

extern "C" int mi_debug_set_rt_lean(int on) {
#if MI_PLANES_FP16
    const int was = mi::g_rt_lean;
    mi::g_rt_lean = on & 3;                                  // 2: the persistent grid (gemm_rt_lean_kernel)
    if ((on >> 2) > 0) mi::g_rt_lean_grid = (on >> 2) * 8;   // (optional: its size in units of 8 workgroups, in the bits above)
    return was;
#else
    (void)on;
    return 0;
#endif
}

extern "C" int mi_debug_rt_clock(void* dev_buffer, int ext, int skip) {
#if MI_PLANES_FP16
    mi::g_rt_clk = (unsigned long long*)dev_buffer;
    mi::g_rt_clk_ext = ext;
    mi::g_rt_clk_skip = skip;
#else
    (void)dev_buffer; (void)ext; (void)skip;
#endif
    return MI_OK;
}

extern "C" int mi_debug_edge2_clock(void* dev_buffer) {
#if MI_PLANES_FP16
    mi::g_edge2_clk = (unsigned long long*)dev_buffer;
#endif
    return MI_OK;
}

extern "C" int mi_debug_edge1_clock(void* dev_buffer) {
#if MI_PLANES_FP16
    mi::g_edge1_clk = (unsigned long long*)dev_buffer;
#endif
    return MI_OK;
}

extern "C" int mi_debug_set_edge2_fused(int on) {
    // (returns the previous MODE, so that save-and-restore through the return value is exact: 4 = inference forwards only)
    const int was = (mi::g_edge2_fused == 1 && !mi::g_edge2_train) ? 4 : mi::g_edge2_fused;
    mi::g_edge2_fused = on == 4 ? 1 : on;
    mi::g_edge2_train = on == 1;   // (4: inference forwards only -- the training forward keeps the 128 x 128 plane GEMM)
    return was;
}

extern "C" int mi_debug_set_edge1_fused(int on) {
    const int was = mi::g_edge1_fused;
    mi::g_edge1_fused = on;
    return was;
}
