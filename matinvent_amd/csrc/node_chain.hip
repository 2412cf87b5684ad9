// The node-level chain between two edge stages of an inference forward as ONE launch (cspnet.py:80-91 of layer l-1, :87-88 and the
// h_i / h_j projections of :61 for layer l):
//
//   phase A (finishes layer l-1)   agg = segmented mean of the edge stage's partial sums          (cspnet.py:79)
//                                  X   = SiLU(agg W0b^T + X_part + b0)                            (node_mlp.0, the agg half; :80-82)
//                                  h'  = h + SiLU(X W2^T + b2)                                    (node_mlp.2 + residual, :91)
//   LayerNorm                      y   = LN_l(h')                                                 (:87-88; the final LayerNorm after layer L)
//   phase B (begins layer l)       [P_i | P_j | X_part] = y [W1[:, :H]; W1[:, H:2H]; W0[:, :H]]^T (everything LayerNorm(h) feeds)
//
// Every step is row-local (a node's row never meets another node's), so a workgroup that owns 32 nodes runs the whole chain with the
// activations in LDS and only h', the projections and the absmax leave the chip.  This replaces seven launches on every chain's
// serial path (finalize_agg, two node-MLP products, LayerNorm, the 3H-wide projection product) by one: a denoising step is the SUM of one
// chain's kernel durations, and those launches were 38 % of the kernel time for 5 % of the flops.
//
// Arithmetic: the fp16 two-plane format of gemm_split.h (three MFMA terms, f32 accumulate, the same scales as the plane GEMMs: the
// rigorous per-layer activation scales published by the pair-mode edge GEMM for agg and X, 2^3 for LayerNorm outputs, 2^6 for weights).
// Weights come pre-split in MFMA FRAGMENT ORDER and go from L2 straight into registers through a D-deep ring (a weight element is used by
// exactly one wave of the workgroup, so staging it in LDS would only add traffic); the activation planes sit in LDS (32 rows x H).
// Products whose result feeds the next product are computed TRANSPOSED (weights as the first MFMA operand): a lane then owns four
// consecutive columns of one row, i.e. whole 8-byte pieces of the next operand's LDS rows.
#include <mutex>

#include "net.h"
#include "gemm_split.h"

namespace mi {

unsigned long long* g_node_clk = nullptr;
int g_node_fused = 1;  // inference forwards: the node-level chain as one launch per layer boundary (0: the seven-launch form)
int g_node_split = 1;  // small / medium batches: phase A and LayerNorm + projections as two launches, the second on three workgroups per row block (0: one launch)
int g_node_split_max_blocks = 256;   // ... while those fit the chip in one round (one workgroup per CU)
int g_ablate_skip = 0;   // TIMING ABLATIONS ONLY (results are garbage): bit 0 = skip the node chain's launches, 1 = the first edge GEMM, 2 = the second (mi_debug_set_skip)
int g_node_touch_min_blocks = 64;   // row blocks from which a row-block chain launch warms the L2 with its weights first: measured +4.3 % on one chain of 256 crystals
                                    // (160 row blocks: node chain 88 -> 74 us), -1 % on four chains of 64 (40 row blocks: each workgroup then touches a fifth of the
                                    // operands and waits for it, A1 8.7 k -> 17 k cycles, while the products -- bound by the CU's L2 port -- barely gain); profiles/r5_node_chain_l2_touch_ab.log
int g_node_cols = 3;   // the column-split form of the chain (node_cols_kernel): 3 = automatic (default: one launch per stage for chains of at most g_node_cols_auto_blocks
                       // row blocks -- the reference's default sampling batch, +3 %; the row-block forms above for larger ones -- the headline's chains measured equal, one
                       // chain of 256 crystals 2.5 % slower: DESIGN 19.1), 1 = one launch per stage always, 2 = one launch per layer boundary with in-launch hand-overs, 0 = off
int g_node_cols_auto_blocks = 36;
int g_node_cols_b_max = 0;         // LayerNorm + projections: one 128-column group per workgroup up to this many workgroups, NCG groups each above (0: always NCG -- 3 x NCG workgroups
                                   // per row block each re-reading h' and recomputing its LayerNorm measured 85 against 62 us per launch under four chains)
int g_node_cols_fused_max = 256;   // the one-launch form only while every workgroup of the launch is resident at once (a waiting workgroup holds its slot)

#if MI_PLANES_FP16

// fragment-order pack of W[rows][K] (row stride ld) into dst: out-column tile ct (32 rows of W), k-step ks (16 k):
//   [ct][ks][plane][lane = kg * 32 + l31][8 halfs] = plane(W[32 ct + l31][16 ks + 8 kg + 0..7] * PL_SW);  rows >= `rows` are zero
// (row0: first destination row, so that several matrices stack into one packed operand)
__global__ void pack_frag_kernel(const float* __restrict__ W, int ld, int rows, int K, u16* __restrict__ dst, int row0) {
    const int KS = K / 16;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one (row, 8-k chunk) per thread
    if (idx >= (int64_t)rows * (K / 8)) return;
    const int r = (int)(idx / (K / 8)), ch = (int)(idx % (K / 8));
    const int R = row0 + r, ct = R >> 5, l31 = R & 31, ks = ch >> 1, kg = ch & 1;
    u32x4 pk[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned p[3];
        pl_split_pair(W[(size_t)r * ld + ch * 8 + 2 * i], W[(size_t)r * ld + ch * 8 + 2 * i + 1], PL_SW, p);
        pk[0][i] = p[0];
        pk[1][i] = p[1];
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
        *reinterpret_cast<u32x4*>(dst + ((((size_t)ct * KS + ks) * 2 + pl) * 64 + kg * 32 + l31) * 8) = pk[pl];
}

struct NodeChainArgs {
    int N = 0;
    // phase A (nullptr part: skipped -- the chain starts at h_in, the embedding's output)
    const float* part = nullptr;    // [nslots][N][H] partial sums of the edge -> node reduction
    const int* rowptr = nullptr;    // [N + 1] CSR rows (degree, slots)
    int seg_shift = 5;              // log2 of the row-block size the partial sums were formed over (32 rows: plane GEMM; 128: edge_stage.hip)
    const unsigned* slotmask = nullptr;   // optional [N]: bit s set = slot s holds a partial sum of this node (edge_fused.hip: a node's edges sit in
    int nslots = 0;                       // several 64-pair tiles + the self-edge tile); summed in increasing slot order instead of the CSR slot range
    const float* xpart = nullptr;   // LayerNorm(h) W0[:, :H]^T of layer l-1 (computed with its P_i / P_j), row stride ld_xpart
    int ld_xpart = 0;
    const u16* Wagg = nullptr;      // fragment-order packs (H x H)
    const u16* Wn2 = nullptr;
    const float* b0 = nullptr;
    const float* b2 = nullptr;
    const float* dsc = nullptr;     // this layer's activation scales {M1, 1/M1, agg, 1/agg, X, 1/X} (act_scales_eval)
    const float* h_in = nullptr;    // [N][H]
    float* h_out = nullptr;         // [N][H] (phase A only)
    // LayerNorm
    const float* ln_w = nullptr;
    const float* ln_b = nullptr;
    float* hf = nullptr;            // final LayerNorm: fp32 rows out, no phase B
    // phase B (nullptr Wln: skipped)
    const u16* Wln = nullptr;       // fragment-order pack (3H x H)
    float* PQ = nullptr;            // [N][3H]
    unsigned* absmax = nullptr;     // atomicMax of the bit pattern of max |PQ|
    unsigned long long* clk = nullptr;  // optional phase clock: [workgroup][16] s_memtime stamps (mi_debug_node_chain_clock)
    // optional (TRAINING forward): what the backward pass reads of this chain, written on the way -- the same values the seven-launch form
    // keeps on its tape (cspnet.hip: finalize_agg / the two node-MLP products' pre-activations / layernorm_kernel)
    float* t_agg = nullptr;     // phase A: the aggregated messages, rows of stride ld_agg (cat[:, H:2H] of layer l-1)
    int ld_agg = 0;
    float* t_xpre = nullptr;    // phase A: node_mlp.0's pre-activation [N][H]
    float* t_ypre = nullptr;    // phase A: node_mlp.2's pre-activation [N][H]
    float* t_ln = nullptr;      // LayerNorm output, rows of stride ld_ln (cat[:, 0:H] of layer l)
    int ld_ln = 0;
    float* t_lnstat = nullptr;  // LayerNorm statistics [N][2] = {mean, 1 / sqrt(var + eps)}
    // the chain as TWO launches (node_chain(): small and medium batches): `a_only` = phase A, h' written, nothing else; `split_b` = a launch of
    // LayerNorm + phase B with gridDim.y = 3, a workgroup running the projection pass blockIdx.y only (the passes are independent given y)
    int a_only = 0, split_b = 0;
    int touch = 0;                  // node_chain_kernel: warm the L2 with this launch's weight operands first (large launches: see the kernel)
    // the COLUMN-SPLIT form (node_cols_kernel): `stages` = which of {1: agg + node_mlp.0 -> X planes, 2: node_mlp.2 + residual -> h', 4: LayerNorm
    // + projections} this launch runs, on `gy` workgroups per 32-row block (each owning 128-column groups gy apart); the X planes and h' cross
    // from one stage to the next through `xpl` / `h_out` -- at a kernel boundary (one launch per stage) or, with `flags`, inside ONE launch
    // behind an agent-scope release / acquire hand-over (all gy workgroups of a row block arrive at flags[2 rb + seam])
    int stages = 0, gy = 1, npad = 0;
    u16* xpl = nullptr;             // [2 planes][npad][H] X = SiLU(node_mlp.0) as fp16 planes, row-major
    unsigned* flags = nullptr;
};

template <int H>
struct NodeChainCfg {
    static constexpr int KS = H / 16, ROWB = 2 * H + 16, PLB = 32 * ROWB, HLD = H + 4;
    static constexpr int LDS = 2 * PLB + 32 * HLD * 4;
};

// NW waves (4 or 8) share the H output columns of a product: wave w owns columns [w CW, (w + 1) CW), CW = H / NW = TW tiles of 32.
// D = depth of the weight ring in 16-deep k-steps.
// PQ: the epilogues' row addends (X_part, the residual h) requested IN FRONT of their product (true) or behind it (false: 64 registers fewer, which a ring of
// depth 8 needs -- twice the weight bytes in flight per CU)
template <int H, int NW, int D, bool PQ = true>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void node_chain_kernel(NodeChainArgs a) {
    using C = NodeChainCfg<H>;
    constexpr int KS = C::KS, ROWB = C::ROWB, PLB = C::PLB, HLD = C::HLD, CW = H / NW, TW = CW / 32, RPW = 32 / NW;
    static_assert(KS % D == 0 && KS >= 2 * D && CW % 32 == 0, "ring depth / column split");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* P = smem;                                    // activation planes [2][32][ROWB]
    float* Hs = reinterpret_cast<float*>(smem + 2 * PLB);       // h' rows [32][HLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    const int row0 = blockIdx.x * 32, N = a.N;

    int stamp_i = 0;
    auto stamp = [&]() {
        if (a.clk && tid == 0) a.clk[(size_t)blockIdx.x * 16 + stamp_i] = __builtin_amdgcn_s_memtime();
        ++stamp_i;
    };
    stamp();
    // L2 warm-up of THIS launch's weight operands (NodeChainArgs::touch: launches of at least g_node_touch_min_blocks row blocks).  Between two node-chain launches the edge GEMMs stream > 100 MB through the L2s, so every
    // product's weight ring runs at the miss latency (the same launch twice in a row: 90 -> 66 us at 5 120 nodes,
    // profiles/r5_node_chain_cold_vs_warm_l2.log).  The workgroups of an XCD (block b runs on XCD b % 8 -- for speed only) share out the lines of all
    // operands and touch one dword per 128-byte line, fire and forget, before anything else: the later products then find their weights in L2.
    float touch_dummy = 0.f;
    if (a.touch) {
        const int bl = (int)(blockIdx.x + gridDim.x * blockIdx.y), nb = (int)(gridDim.x * gridDim.y);
        const int part = bl >> 3, nparts = (nb - (bl & 7) + 7) >> 3;
        auto touch = [&](const void* base, int bytes) {
            const int nlines = bytes >> 7;
            for (int l0 = part * (64 * NW); l0 < nlines; l0 += nparts * (64 * NW)) {
                const int line = l0 + tid;
                // ("+v": EVERY touch writes the one VGPR that stays reserved until the closing s_waitcnt below -- a plain output would be a dead value whose
                // register the allocator may reuse while the load is still in flight; scripts/scan_touch_regs.py checks the disassembly)
                if (line < nlines) asm volatile("global_load_dword %0, %1, off" : "+v"(touch_dummy) : "v"(reinterpret_cast<const char*>(base) + (size_t)line * 128));
            }
        };
        const int wbytes = H * H * 4;
        if (a.part != nullptr) {
            touch(a.Wagg, wbytes);
            touch(a.Wn2, wbytes);
        }
        if (a.Wln != nullptr && !a.a_only) touch(a.Wln + (size_t)(a.split_b ? blockIdx.y : 0) * H * H * 2, (a.split_b ? 1 : 3) * wbytes);
    }
    u32x4 ring[D][TW][2];
    f32x16 acc[TW];
    // ---- weight ring: k-steps ks .. ks + D - 1 of the wave's TW column tiles (first tile ct0) in flight ----
    // (per-tile offsets live in the vector offset, the plane in the instruction's immediate: one scalar offset per k-step)
    int voff[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) voff[t] = lane * 16 + t * KS * 2048;
    auto ring_load = [&](const __amdgpu_buffer_rsrc_t& rs, int ct0, int ks, u32x4 (&w)[TW][2]) {
        const int so = (ct0 * KS + ks) * 2048;
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) w[t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[t] + pl * 1024, so, 0);
    };
    auto ring_fill = [&](const __amdgpu_buffer_rsrc_t& rs, int ct0) {
#pragma unroll
        for (int d = 0; d < D; ++d) ring_load(rs, ct0, d, ring[d]);
    };
    auto read_act = [&](int ks, f16x8 (&af)[2]) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[pl] = *reinterpret_cast<const f16x8*>(P + pl * PLB + l31 * ROWB + (2 * ks + kg) * 16);
    };
    // one product over K = H: acc[t] (+)= the wave's TW tiles.  TR: weights as the first operand (lane = row, registers = columns).
    // Term order of the plane GEMMs: (a1, b0), (a0, b1), (a0, b0) with a = activation, b = weight.
    auto mma_step = [&](auto tr, const u32x4 (&w)[TW][2], const f16x8 (&af)[2]) {
        constexpr bool TR = decltype(tr)::value;
#pragma unroll
        for (int term = MI_TERM0; term < 3; ++term)
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                const f16x8 wv = __builtin_bit_cast(f16x8, w[t][term == 1 ? 1 : 0]);
                const f16x8 av = af[term == 0 ? 1 : 0];
                if constexpr (TR) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv, av, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, wv, acc[t], 0, 0, 0);
            }
    };
    auto run = [&](auto tr, const __amdgpu_buffer_rsrc_t& rs, int ct0) {   // the ring holds k-steps 0 .. D-1 on entry, nothing on exit
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        f16x8 af[2][2];
        read_act(0, af[0]);
#pragma unroll 1
        for (int ks0 = 0; ks0 < KS - D; ks0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                read_act(ks0 + d + 1, af[(d + 1) & 1]);
                mma_step(tr, ring[d], af[d & 1]);
                ring_load(rs, ct0, ks0 + d + D, ring[d]);
                // (pin the order: left alone, the scheduler moves every refill to the end of the unrolled body, which empties the ring)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d + 1 < D) read_act(KS - D + d + 1, af[(d + 1) & 1]);
            mma_step(tr, ring[d], af[d & 1]);
        }
    };
    using TRt = std::true_type;
    using TRf = std::false_type;
    static_assert(D % 2 == 0, "the activation fragments alternate between two register sets");

    const bool phaseA = a.part != nullptr, phaseB = a.Wln != nullptr;
    const int wsz = H * H * 4;  // bytes of one packed H x H operand
    if (phaseA) {
        const __amdgpu_buffer_rsrc_t rs_agg = uniform_rsrc(a.Wagg, wsz);
        // ---- A1: agg = (sum of the node's slots) / degree -> planes (finalize_agg_kernel's arithmetic) ----
        const float s_agg = a.dsc[2];
        unsigned sat = 0;
        {
            // all of the wave's rows at once: the row pointers, then the first two slots of every row (the common case: a node's edges
            // span at most two 32-row blocks), then the arithmetic -- one chain of two dependent load latencies instead of one per row
            const int c0 = lane * 8;
            const bool act = c0 < H;
            int e0[RPW], e1[RPW];
            int rp;
            {
                const int idx = row0 + wave * RPW + lane;
                rp = a.rowptr[idx < N ? idx : N];   // (lanes 0 .. RPW hold rowptr[i0 .. i0 + RPW]; the table has N + 1 entries)
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                // (the wave's RPW + 1 consecutive row pointers come from ONE lane-parallel load, `rp`, below: `i < N ? rowptr[i] : 0` per row
                //  compiled to RPW wave-uniform loads, each with its own s_waitcnt vmcnt(0) + v_readfirstlane -- RPW dependent memory
                //  latencies in front of everything else)
                const int i = row0 + wave * RPW + r;
                e0[r] = i < N ? __builtin_amdgcn_readlane(rp, r) : 0;
                e1[r] = i < N ? __builtin_amdgcn_readlane(rp, r + 1) : 0;
            }
            f32x4 xsum[RPW], ysum[RPW];
            bool any[RPW];
            if (a.slotmask) {   // slots by mask (a node's partial sums come from every tile that holds one of its edges)
                unsigned mk[RPW];
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int i = row0 + wave * RPW + r;
                    mk[r] = i < N ? a.slotmask[i] : 0u;
                    xsum[r] = ysum[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                    any[r] = mk[r] != 0u && e1[r] > e0[r];
                }
                for (int sl = 0; sl < a.nslots; ++sl) {
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const int i = row0 + wave * RPW + r;
                        if (act && ((mk[r] >> sl) & 1u)) {
                            const float* p = a.part + ((size_t)sl * N + i) * H + c0;
                            xsum[r] += *reinterpret_cast<const f32x4*>(p);
                            ysum[r] += *reinterpret_cast<const f32x4*>(p + 4);
                        }
                    }
                }
                ring_fill(rs_agg, wave * TW);
            } else {
            f32x4 x[RPW], y[RPW], x1[RPW], y1[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int i = row0 + wave * RPW + r;
                const int ns = e1[r] > e0[r] ? ((e1[r] - 1) >> a.seg_shift) - (e0[r] >> a.seg_shift) + 1 : 0;
                // (these stay conditional: loading both slots of every row unconditionally and masking afterwards measured SLOWER -- A1 8.6 k -> 10.3 k
                //  cycles, profiles/r5_node_a1_ab.log: twice the bytes for the rows that have one slot)
                x[r] = y[r] = x1[r] = y1[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (act && ns > 0) {
                    const float* p = a.part + (size_t)i * H + c0;
                    x[r] = *reinterpret_cast<const f32x4*>(p);
                    y[r] = *reinterpret_cast<const f32x4*>(p + 4);
                    if (ns > 1) {
                        x1[r] = *reinterpret_cast<const f32x4*>(p + (size_t)N * H);
                        y1[r] = *reinterpret_cast<const f32x4*>(p + (size_t)N * H + 4);
                    }
                }
            }
            // (vector loads return in order: the ring fill goes BEHIND the gathers, so that the arithmetic waits for the gathers only and
            // the weights land while it runs)
            ring_fill(rs_agg, wave * TW);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int i = row0 + wave * RPW + r;
                const int ns = e1[r] > e0[r] ? ((e1[r] - 1) >> a.seg_shift) - (e0[r] >> a.seg_shift) + 1 : 0;
                f32x4 xs = x[r], ys = y[r];
                if (act && ns > 1) {
                    xs += x1[r];
                    ys += y1[r];
                    for (int sl = 2; sl < ns; ++sl) {
                        const float* p = a.part + ((size_t)sl * N + i) * H + c0;
                        xs += *reinterpret_cast<const f32x4*>(p);
                        ys += *reinterpret_cast<const f32x4*>(p + 4);
                    }
                }
                xsum[r] = xs;
                ysum[r] = ys;
                any[r] = ns > 0;
            }
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wave * RPW + r;
                if (!act) continue;
                f32x4 xs = xsum[r], ys = ysum[r];
                if (any[r]) {
                    const float d = (float)(e1[r] - e0[r]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        xs[k] = xs[k] / d;
                        ys[k] = ys[k] / d;
                    }
                }
                if (a.t_agg && row0 + row < N) {
                    float* o = a.t_agg + (size_t)(row0 + row) * a.ld_agg + c0;
                    *reinterpret_cast<f32x4*>(o) = xs;
                    *reinterpret_cast<f32x4*>(o + 4) = ys;
                }
                u32x4 pk[2];
                unsigned pr[3];
                pl_split_pair_acc(xs[0], xs[1], s_agg, pr, sat); pk[0][0] = pr[0]; pk[1][0] = pr[1];
                pl_split_pair_acc(xs[2], xs[3], s_agg, pr, sat); pk[0][1] = pr[0]; pk[1][1] = pr[1];
                pl_split_pair_acc(ys[0], ys[1], s_agg, pr, sat); pk[0][2] = pr[0]; pk[1][2] = pr[1];
                pl_split_pair_acc(ys[2], ys[3], s_agg, pr, sat); pk[0][3] = pr[0]; pk[1][3] = pr[1];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<u32x4*>(P + pl * PLB + row * ROWB + c0 * 2) = pk[pl];
            }
        }
        __syncthreads();
        stamp();
        // (the epilogue's row gathers do not depend on the product: requested BEFORE it, they land under its MFMAs instead of costing one
        //  exposed memory latency per phase -- the phase clock put A3 at 18k cycles against 12k for the product itself)
        f32x4 xq[TW][4];
        auto load_xq = [&]() {
            const int i = row0 + l31;
            const int ic = i < N ? i : N - 1;   // (clamped, UNCONDITIONAL gathers: `if (i < N) x = load` compiled to a branch per load and, through a register copy, an s_waitcnt vmcnt(0) in the middle of them -- DESIGN 19.3; rows past N are masked where they are stored)
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xq[t][q] = *reinterpret_cast<const f32x4*>(a.xpart + (size_t)ic * a.ld_xpart + wave * CW + t * 32 + 8 * q + 4 * kg);
                }
        };
        if constexpr (PQ) load_xq();
        // ---- A2: Z = agg W0b^T (transposed) ----
        run(TRt{}, rs_agg, wave * TW);
        stamp();
        if constexpr (!PQ) load_xq();
        const __amdgpu_buffer_rsrc_t rs_n2 = uniform_rsrc(a.Wn2, wsz);
        f32x4 hq[TW][4];   // the residual rows of A5, requested here for the same reason
        auto load_hq = [&]() {
            const int i = row0 + l31;
            const int ic = i < N ? i : N - 1;   // (clamped, UNCONDITIONAL gathers: `if (i < N) x = load` compiled to a branch per load and, through a register copy, an s_waitcnt vmcnt(0) in the middle of them -- DESIGN 19.3; rows past N are masked where they are stored)
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    hq[t][q] = *reinterpret_cast<const f32x4*>(a.h_in + (size_t)ic * H + wave * CW + t * 32 + 8 * q + 4 * kg);
                }
        };
        if constexpr (PQ) load_hq();
        __syncthreads();                // every wave has read the agg planes
        // ---- A3: X = SiLU(Z + b0 + X_part) -> planes ----
        {
            const float os = a.dsc[3] * (1.f / PL_SW), s_x = a.dsc[4];
            const int i = row0 + l31;
            f32x4 bq[TW][4];
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) bq[t][q] = *reinterpret_cast<const f32x4*>(a.b0 + wave * CW + t * 32 + 8 * q + 4 * kg);
            ring_fill(rs_n2, wave * TW);    // behind the gathers, in flight under the arithmetic
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c4 = wave * CW + t * 32 + 8 * q + 4 * kg;
                    float v[4];
                    f32x4 zpre;
#pragma unroll
                    for (int k = 0; k < 4; ++k) zpre[k] = (acc[t][4 * q + k] * os + bq[t][q][k]) + xq[t][q][k];
                    if (a.t_xpre && i < N) *reinterpret_cast<f32x4*>(a.t_xpre + (size_t)i * H + c4) = zpre;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = i < N ? silu_fast(zpre[k]) : 0.f;
                    unsigned p01[3], p23[3];
                    pl_split_pair_acc(v[0], v[1], s_x, p01, sat);
                    pl_split_pair_acc(v[2], v[3], s_x, p23, sat);
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const uint2 wv = make_uint2(p01[pl], p23[pl]);
                        *reinterpret_cast<uint2*>(P + pl * PLB + l31 * ROWB + c4 * 2) = wv;
                    }
                }
            sat_report(sat);
        }
        __syncthreads();
        stamp();
        // ---- A4: Y = X W2^T (transposed) ----
        run(TRt{}, rs_n2, wave * TW);
        stamp();
        if constexpr (!PQ) load_hq();
        // ---- A5: h' = h + SiLU(Y + b2) -> Hs ----
        {
            const float os = a.dsc[5] * (1.f / PL_SW);
            const int i = row0 + l31;
            f32x4 bq[TW][4];
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) bq[t][q] = *reinterpret_cast<const f32x4*>(a.b2 + wave * CW + t * 32 + 8 * q + 4 * kg);
            if (phaseB) ring_fill(uniform_rsrc(a.Wln, 3 * wsz), wave * TW);   // pass 0 of phase B, in flight under the epilogue and the LayerNorm
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c4 = wave * CW + t * 32 + 8 * q + 4 * kg;
                    f32x4 o, ypre;
#pragma unroll
                    for (int k = 0; k < 4; ++k) ypre[k] = acc[t][4 * q + k] * os + bq[t][q][k];
                    if (a.t_ypre && i < N) *reinterpret_cast<f32x4*>(a.t_ypre + (size_t)i * H + c4) = ypre;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = silu_fast(ypre[k]) + hq[t][q][k];
                    *reinterpret_cast<f32x4*>(Hs + l31 * HLD + c4) = o;
                }
        }
        __syncthreads();
        stamp();
    } else if (phaseB) {
        ring_fill(uniform_rsrc(a.Wln, 3 * wsz), ((a.split_b ? (int)blockIdx.y : 0) * H + wave * CW) / 32);
    }
    if (a.touch) asm volatile("s_waitcnt vmcnt(0)" : "+v"(touch_dummy));   // (the touches have landed long ago: this only ends the register's reservation)
    if (a.a_only) {   // (the first of two launches: h' leaves, LayerNorm and the projections follow on three times the CUs)
        const int c0 = lane * 8;
        if (c0 < H)
#pragma unroll 2
            for (int r = 0; r < RPW; ++r) {
                const int row = wave * RPW + r, i = row0 + row;
                if (i < N) {
                    *reinterpret_cast<f32x4*>(a.h_out + (size_t)i * H + c0) = *reinterpret_cast<const f32x4*>(Hs + row * HLD + c0);
                    *reinterpret_cast<f32x4*>(a.h_out + (size_t)i * H + c0 + 4) = *reinterpret_cast<const f32x4*>(Hs + row * HLD + c0 + 4);
                }
            }
        return;
    }
    const int pass_lo = a.split_b ? (int)blockIdx.y : 0, pass_hi = a.split_b ? (int)blockIdx.y + 1 : 3;
    const bool tape_w = !a.split_b || blockIdx.y == 0;   // (the three workgroups of a row block compute the same LayerNorm: one writes the tape)
    // ---- LayerNorm (layernorm_kernel's arithmetic: a wave per row, a lane owns eight consecutive columns) ----
    unsigned sat_ln = 0;
    {
        const int c0 = lane * 8;
        const bool act = c0 < H;
        f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0, b0 = w0, b1 = w0;
        if (act) {
            w0 = *reinterpret_cast<const f32x4*>(a.ln_w + c0);
            w1 = *reinterpret_cast<const f32x4*>(a.ln_w + c0 + 4);
            b0 = *reinterpret_cast<const f32x4*>(a.ln_b + c0);
            b1 = *reinterpret_cast<const f32x4*>(a.ln_b + c0 + 4);
        }
#pragma unroll 2
        for (int r = 0; r < RPW; ++r) {
            const int row = wave * RPW + r, i = row0 + row;
            f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = {0.f, 0.f, 0.f, 0.f};
            if (act && i < N) {
                if (phaseA) {
                    x = *reinterpret_cast<const f32x4*>(Hs + row * HLD + c0);
                    y = *reinterpret_cast<const f32x4*>(Hs + row * HLD + c0 + 4);
                    *reinterpret_cast<f32x4*>(a.h_out + (size_t)i * H + c0) = x;
                    *reinterpret_cast<f32x4*>(a.h_out + (size_t)i * H + c0 + 4) = y;
                } else {
                    x = *reinterpret_cast<const f32x4*>(a.h_in + (size_t)i * H + c0);
                    y = *reinterpret_cast<const f32x4*>(a.h_in + (size_t)i * H + c0 + 4);
                }
            }
            const float mean = wave_sum(((x[0] + x[1]) + (x[2] + x[3])) + ((y[0] + y[1]) + (y[2] + y[3]))) / (float)H;
            float q = 0.f;
            if (act) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float da = x[k] - mean, db = y[k] - mean;
                    q += da * da + db * db;
                }
            }
            const float var = wave_sum(q) / (float)H;
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
            if (act && i < N) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o0[k] = (x[k] - mean) * rstd * w0[k] + b0[k];
                    o1[k] = (y[k] - mean) * rstd * w1[k] + b1[k];
                }
                if (a.hf) {
                    *reinterpret_cast<f32x4*>(a.hf + (size_t)i * H + c0) = o0;
                    *reinterpret_cast<f32x4*>(a.hf + (size_t)i * H + c0 + 4) = o1;
                }
                if (a.t_ln && tape_w) {
                    *reinterpret_cast<f32x4*>(a.t_ln + (size_t)i * a.ld_ln + c0) = o0;
                    *reinterpret_cast<f32x4*>(a.t_ln + (size_t)i * a.ld_ln + c0 + 4) = o1;
                }
            }
            if (a.t_lnstat && tape_w && lane == 0 && i < N) {
                a.t_lnstat[2 * (size_t)i] = mean;
                a.t_lnstat[2 * (size_t)i + 1] = rstd;
            }
            if (act && phaseB) {
                u32x4 pk[2];
                unsigned pr[3];
                pl_split_pair_acc(o0[0], o0[1], PL_S_LN, pr, sat_ln); pk[0][0] = pr[0]; pk[1][0] = pr[1];
                pl_split_pair_acc(o0[2], o0[3], PL_S_LN, pr, sat_ln); pk[0][1] = pr[0]; pk[1][1] = pr[1];
                pl_split_pair_acc(o1[0], o1[1], PL_S_LN, pr, sat_ln); pk[0][2] = pr[0]; pk[1][2] = pr[1];
                pl_split_pair_acc(o1[2], o1[3], PL_S_LN, pr, sat_ln); pk[0][3] = pr[0]; pk[1][3] = pr[1];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<u32x4*>(P + pl * PLB + row * ROWB + c0 * 2) = pk[pl];
            }
        }
    }
    sat_report(sat_ln);
    if (!phaseB) return;
    __syncthreads();
    stamp();
    // ---- phase B: [P_i | P_j | X_part] = y Wln^T, three passes of H columns (lane = column, registers = rows) ----
    {
        const __amdgpu_buffer_rsrc_t rs_ln = uniform_rsrc(a.Wln, 3 * wsz);
        constexpr float os = 1.f / (PL_S_LN * PL_SW);
        float m = 0.f;
#pragma unroll 1
        for (int pass = pass_lo; pass < pass_hi; ++pass) {
            const int ct0 = (pass * H + wave * CW) / 32;
            run(TRf{}, rs_ln, ct0);
            stamp();
            if (pass + 1 < pass_hi) ring_fill(rs_ln, ((pass + 1) * H + wave * CW) / 32);
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                const int col = pass * H + wave * CW + t * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = row0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    const float v = acc[t][r] * os;
                    if (i < N) {
                        a.PQ[(size_t)i * (3 * H) + col] = v;
                        m = fmaxf(m, fabsf(v));
                    }
                }
            }
        }
        if (a.absmax) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            if (lane == 0) atomicMax(a.absmax, __float_as_uint(m));
        }
        stamp();
    }
}

template <int H, int NW, int D, bool PQ = true>
static int node_chain_launch(const NodeChainArgs& a, hipStream_t s) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] { attr_err = hipFuncSetAttribute((const void*)node_chain_kernel<H, NW, D, PQ>, hipFuncAttributeMaxDynamicSharedMemorySize, NodeChainCfg<H>::LDS); });
    MI_HIP(attr_err);
    hipLaunchKernelGGL((node_chain_kernel<H, NW, D, PQ>), dim3(cdiv(a.N, 32), a.split_b ? 3 : 1), dim3(64 * NW), NodeChainCfg<H>::LDS, s, a);
    MI_KERNEL_CHECK();
    return MI_OK;
}


// ------------------------------------------------------------------------------------------------------------------------------------------
// The same chain with every product's COLUMNS split over workgroups.  The one-launch kernel above is bound by what ONE CU's L2
// port delivers: a workgroup owning 32 rows streams each product's whole weight operand (1 MB at H = 512, five times per layer) and, at 8
// waves x 228 registers + 133 KB of LDS, needs a CU to itself -- beside other chains' edge GEMMs (two 4-wave workgroups per CU) it waits for
// one to drain completely.  Here a workgroup is 4 waves x one 32 x 32 tile = 32 rows x 128 columns of ONE product: a quarter of the weight
// bytes per CU, 1 / NCG of the time per product, 67 KB of LDS and < 128 registers, so it is placed next to a running edge-GEMM workgroup.  The
// price is the exchange of the intermediates' column slices between the workgroups of a row block (X planes, h'): through L2, either at a
// kernel boundary (three launches per layer boundary: ~1.5 us each, MI355X guide "boundary") or inside one launch behind an agent-scope
// flag hand-over (`flags`; the group sits on one XCD by block-id arithmetic, so the slices are served by the L2 they were written to).
// Arithmetic, k order and term order per output element are those of node_chain_kernel: results are bit-identical.
template <int H>
struct NodeColsCfg {
    static constexpr int KS = H / 16, ROWB = 2 * H + 16, PLB = 32 * ROWB, NCG = H / 128;
    static constexpr int LDS = 2 * PLB + 32;   // (+ 32: the fragment read one k-step ahead of the last row's last step lands here)
};

template <int H, int D, int ST>   // ST: the stages this instantiation runs (1, 2, 4: one per launch; 7: all three behind hand-overs)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void node_cols_kernel(NodeChainArgs a) {
    using C = NodeColsCfg<H>;
    constexpr int KS = C::KS, ROWB = C::ROWB, PLB = C::PLB, NCG = C::NCG, RPW = 8;
    static_assert(KS % D == 0 && KS >= 2 * D && D % 2 == 0 && H % 128 == 0, "ring depth / column groups");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* P = smem;                                    // activation planes [2][32][ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    const int N = a.N, nblk = (N + 31) >> 5, gy = a.gy;
    // block id -> (row block, column-group worker) with the gy workgroups of a row block on ONE XCD (dispatch is round-robin over the 8 XCDs)
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int cgw = j % gy, rb = (j / gy) * 8 + xcd;
    if (rb >= nblk) return;
    const int row0 = rb * 32;

    int stamp_i = 0;
    auto stamp = [&]() {
        if (a.clk && tid == 0) a.clk[(size_t)bid * 16 + stamp_i] = __builtin_amdgcn_s_memtime();
        ++stamp_i;
    };
    stamp();
    u32x4 ring[D][2];
    f32x16 acc;
    const int voff = lane * 16;
    // One descriptor per (operand, column tile): exactly the tile's KS k-steps.  The k-loop has no tail -- every step refills its ring slot D
    // steps ahead, and the refills past the tile's end are out of range: they return zeros without a memory access (the k-step offset sits in
    // the VECTOR offset, which the range check covers), so the loop body is one uniform block whatever D is.
    auto tile_rsrc = [&](const u16* W, int ct) { return uniform_rsrc(W + (size_t)ct * KS * 1024, KS * 2048); };
    // (ks = kv + d: the part that can run past the tile's end, kv, in the vector offset -- ONE register per D steps --, the slot d in the scalar offset)
    auto ring_load = [&](const __amdgpu_buffer_rsrc_t& rs, int vo, int d, u32x4 (&w)[2]) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) w[pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + pl * 1024, d * 2048, 0);
    };
    auto ring_fill = [&](const __amdgpu_buffer_rsrc_t& rs) {
#pragma unroll
        for (int d = 0; d < D; ++d) ring_load(rs, voff, d, ring[d]);
    };
    auto read_act = [&](int ks, f16x8 (&af)[2]) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[pl] = *reinterpret_cast<const f16x8*>(P + pl * PLB + l31 * ROWB + (2 * ks + kg) * 16);
    };
    // term order of the plane GEMMs: (a1, b0), (a0, b1), (a0, b0) with a = activation, b = weight
    auto mma_step = [&](auto tr, const u32x4 (&w)[2], const f16x8 (&af)[2]) {
        constexpr bool TR = decltype(tr)::value;
#pragma unroll
        for (int term = MI_TERM0; term < 3; ++term) {
            const f16x8 wv = __builtin_bit_cast(f16x8, w[term == 1 ? 1 : 0]);
            const f16x8 av = af[term == 0 ? 1 : 0];
            if constexpr (TR) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv, av, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, wv, acc, 0, 0, 0);
        }
    };
    auto run = [&](auto tr, const __amdgpu_buffer_rsrc_t& rs) {   // the ring holds k-steps 0 .. D-1 on entry
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        f16x8 af[2][2];
        read_act(0, af[0]);
#pragma unroll 1
        for (int ks0 = 0; ks0 < KS; ks0 += D) {
            const int vo = voff + (ks0 + D) * 2048;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                read_act(ks0 + d + 1, af[(d + 1) & 1]);
                mma_step(tr, ring[d], af[d & 1]);
                ring_load(rs, vo, d, ring[d]);
                __builtin_amdgcn_sched_barrier(0);   // (pins the refill behind its slot's use: see node_chain_kernel)
            }
        }
    };
    using TRt = std::true_type;
    using TRf = std::false_type;
    // hand-over between two stages of ONE launch: every workgroup of the row block has published its slice (MI355X guide, Guideline 16:
    // plain stores -> barrier -> one lane's agent-scope release -> drained -> relaxed arrive; ONE relaxed poll loop -> agent acquire -> barrier)
    auto handover = [&](int seam) {
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned* f = a.flags + 2 * rb + seam;
            __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)gy) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    };

    const int cg = cgw;         // stages 1 and 2: this workgroup's 128-column group (gy == NCG)
    const int ct_a = cg * 4 + wave;
    const int ccol = cg * 128 + wave * 32;   // first column of the wave's tile
    if constexpr ((ST & 1) != 0) {
        const __amdgpu_buffer_rsrc_t rs_agg = tile_rsrc(a.Wagg, ct_a);
        // ---- A1: agg = (sum of the node's slots) / degree -> planes, all H columns (every workgroup of the row block needs the whole K) ----
        const float s_agg = a.dsc[2];
        unsigned sat = 0;
        {
            const int c0 = lane * 8;
            const bool act = c0 < H;
            int rp;
            {
                const int idx = row0 + wave * RPW + lane;
                rp = a.rowptr[idx < N ? idx : N];   // (lanes 0 .. RPW hold the wave's RPW + 1 consecutive row pointers)
            }
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {   // (two rounds of four rows: the gathers of a round in flight together, 64 registers)
                constexpr int RH = RPW / 2;
                int e0[RH], e1[RH];
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    const int i = row0 + wave * RPW + half * RH + r;   // (from the one lane-parallel load `rp`: see node_chain_kernel)
                    e0[r] = i < N ? __builtin_amdgcn_readlane(rp, half * RH + r) : 0;
                    e1[r] = i < N ? __builtin_amdgcn_readlane(rp, half * RH + r + 1) : 0;
                }
                f32x4 x[RH], y[RH], x1[RH], y1[RH];
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    const int i = row0 + wave * RPW + half * RH + r;
                    const int ns = e1[r] > e0[r] ? ((e1[r] - 1) >> a.seg_shift) - (e0[r] >> a.seg_shift) + 1 : 0;
                    x[r] = y[r] = x1[r] = y1[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (act && ns > 0) {
                        const float* p = a.part + (size_t)i * H + c0;
                        x[r] = *reinterpret_cast<const f32x4*>(p);
                        y[r] = *reinterpret_cast<const f32x4*>(p + 4);
                        if (ns > 1) {
                            x1[r] = *reinterpret_cast<const f32x4*>(p + (size_t)N * H);
                            y1[r] = *reinterpret_cast<const f32x4*>(p + (size_t)N * H + 4);
                        }
                    }
                }
                if (half == 1) ring_fill(rs_agg);   // behind the last gathers (vector loads return in order), in flight under the arithmetic
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    const int row = wave * RPW + half * RH + r, i = row0 + row;
                    const int ns = e1[r] > e0[r] ? ((e1[r] - 1) >> a.seg_shift) - (e0[r] >> a.seg_shift) + 1 : 0;
                    f32x4 xs = x[r], ys = y[r];
                    if (act && ns > 1) {
                        xs += x1[r];
                        ys += y1[r];
                        for (int sl = 2; sl < ns; ++sl) {
                            const float* p = a.part + ((size_t)sl * N + i) * H + c0;
                            xs += *reinterpret_cast<const f32x4*>(p);
                            ys += *reinterpret_cast<const f32x4*>(p + 4);
                        }
                    }
                    if (!act) continue;
                    if (ns > 0) {
                        const float d = (float)(e1[r] - e0[r]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            xs[k] = xs[k] / d;
                            ys[k] = ys[k] / d;
                        }
                    }
                    if (a.t_agg && cgw == 0 && i < N) {
                        float* o = a.t_agg + (size_t)i * a.ld_agg + c0;
                        *reinterpret_cast<f32x4*>(o) = xs;
                        *reinterpret_cast<f32x4*>(o + 4) = ys;
                    }
                    u32x4 pk[2];
                    unsigned pr[3];
                    pl_split_pair_acc(xs[0], xs[1], s_agg, pr, sat); pk[0][0] = pr[0]; pk[1][0] = pr[1];
                    pl_split_pair_acc(xs[2], xs[3], s_agg, pr, sat); pk[0][1] = pr[0]; pk[1][1] = pr[1];
                    pl_split_pair_acc(ys[0], ys[1], s_agg, pr, sat); pk[0][2] = pr[0]; pk[1][2] = pr[1];
                    pl_split_pair_acc(ys[2], ys[3], s_agg, pr, sat); pk[0][3] = pr[0]; pk[1][3] = pr[1];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<u32x4*>(P + pl * PLB + row * ROWB + c0 * 2) = pk[pl];
                }
            }
        }
        __syncthreads();
        stamp();
        f32x4 xq[4];   // the epilogue's row addends, requested in front of the product
        {
            const int i = row0 + l31;
            const int ic = i < N ? i : N - 1;   // (clamped, UNCONDITIONAL gathers: `if (i < N) x = load` compiled to a branch per load and, through a register copy, an s_waitcnt vmcnt(0) in the middle of them -- DESIGN 19.3; rows past N are masked where they are stored)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xq[q] = *reinterpret_cast<const f32x4*>(a.xpart + (size_t)ic * a.ld_xpart + ccol + 8 * q + 4 * kg);
            }
        }
        // ---- A2: Z = agg W0b^T (transposed), this workgroup's 128 columns ----
        run(TRt{}, rs_agg);
        stamp();
        // ---- A3: X = SiLU(Z + b0 + X_part) -> planes, to the exchange buffer ----
        {
            const float os = a.dsc[3] * (1.f / PL_SW), s_x = a.dsc[4];
            const int i = row0 + l31;
            f32x4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(a.b0 + ccol + 8 * q + 4 * kg);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c4 = ccol + 8 * q + 4 * kg;
                float v[4];
                f32x4 zpre;
#pragma unroll
                for (int k = 0; k < 4; ++k) zpre[k] = (acc[4 * q + k] * os + bq[q][k]) + xq[q][k];
                if (a.t_xpre && i < N) *reinterpret_cast<f32x4*>(a.t_xpre + (size_t)i * H + c4) = zpre;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = i < N ? silu_fast(zpre[k]) : 0.f;
                unsigned p01[3], p23[3];
                pl_split_pair_acc(v[0], v[1], s_x, p01, sat);
                pl_split_pair_acc(v[2], v[3], s_x, p23, sat);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    *reinterpret_cast<uint2*>(a.xpl + ((size_t)pl * a.npad + i) * H + c4) = make_uint2(p01[pl], p23[pl]);
            }
            sat_report(sat);
        }
        stamp();
        if constexpr ((ST & 2) != 0) handover(0);
    }
    if constexpr ((ST & 2) != 0) {
        const __amdgpu_buffer_rsrc_t rs_n2 = tile_rsrc(a.Wn2, ct_a);
        // ---- the row block's X planes (all H columns) -> LDS: 2 x 32 rows of 2H bytes, 16-byte pieces ----
        {
            constexpr int PPR = H / 8, NP = 32 * PPR / 256;   // pieces per row, pieces per thread and plane
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {   // (a plane at a time: 32 registers of staging next to the weight ring)
                u32x4 v[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int pc = tid + 256 * k, r = pc / PPR, c = pc % PPR;
                    v[k] = *reinterpret_cast<const u32x4*>(a.xpl + ((size_t)pl * a.npad + row0 + r) * H + c * 8);
                }
                if (pl == 1) ring_fill(rs_n2);   // behind the last piece loads (vector loads return in order)
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int pc = tid + 256 * k, r = pc / PPR, c = pc % PPR;
                    *reinterpret_cast<u32x4*>(P + pl * PLB + r * ROWB + c * 16) = v[k];
                }
                __builtin_amdgcn_sched_barrier(0);   // (keeps the second plane's loads behind the first plane's LDS writes: both in registers at once spill)
            }
        }
        f32x4 hq[4];   // the residual rows of A5
        {
            const int i = row0 + l31;
            const int ic = i < N ? i : N - 1;   // (clamped, UNCONDITIONAL gathers: `if (i < N) x = load` compiled to a branch per load and, through a register copy, an s_waitcnt vmcnt(0) in the middle of them -- DESIGN 19.3; rows past N are masked where they are stored)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hq[q] = *reinterpret_cast<const f32x4*>(a.h_in + (size_t)ic * H + ccol + 8 * q + 4 * kg);
            }
        }
        __syncthreads();
        stamp();
        // ---- A4: Y = X W2^T (transposed) ----
        run(TRt{}, rs_n2);
        stamp();
        // ---- A5: h' = h + SiLU(Y + b2) -> h_out ----
        {
            const float os = a.dsc[5] * (1.f / PL_SW);
            const int i = row0 + l31;
            f32x4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(a.b2 + ccol + 8 * q + 4 * kg);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c4 = ccol + 8 * q + 4 * kg;
                f32x4 o, ypre;
#pragma unroll
                for (int k = 0; k < 4; ++k) ypre[k] = acc[4 * q + k] * os + bq[q][k];
                if (a.t_ypre && i < N) *reinterpret_cast<f32x4*>(a.t_ypre + (size_t)i * H + c4) = ypre;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = silu_fast(ypre[k]) + hq[q][k];
                if (i < N) *reinterpret_cast<f32x4*>(a.h_out + (size_t)i * H + c4) = o;
            }
        }
        stamp();
        if constexpr ((ST & 4) != 0) handover(1);
    }
    if constexpr ((ST & 4) == 0) return;
    const bool phaseB = a.Wln != nullptr;
    const float* hsrc = (ST & 2) ? a.h_out : a.h_in;   // (a launch of this stage alone is handed h' as h_in)
    const bool tape_w = cgw == 0;   // (the gy workgroups of a row block compute the same LayerNorm: one writes the tape / the final rows)
    // ---- LayerNorm (layernorm_kernel's arithmetic: a wave per row, a lane owns eight consecutive columns) ----
    unsigned sat_ln = 0;
    {
        const int c0 = lane * 8;
        const bool act = c0 < H;
        f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0, b0 = w0, b1 = w0;
        if (act) {
            w0 = *reinterpret_cast<const f32x4*>(a.ln_w + c0);
            w1 = *reinterpret_cast<const f32x4*>(a.ln_w + c0 + 4);
            b0 = *reinterpret_cast<const f32x4*>(a.ln_b + c0);
            b1 = *reinterpret_cast<const f32x4*>(a.ln_b + c0 + 4);
        }
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
        constexpr int RH = RPW / 2;
        f32x4 xr[RH], yr[RH];
#pragma unroll
        for (int r = 0; r < RH; ++r) {   // four rows of the wave requested at once
            const int i = row0 + wave * RPW + half * RH + r;
            xr[r] = yr[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (act && i < N) {
                xr[r] = *reinterpret_cast<const f32x4*>(hsrc + (size_t)i * H + c0);
                yr[r] = *reinterpret_cast<const f32x4*>(hsrc + (size_t)i * H + c0 + 4);
            }
        }
        if (phaseB && half == 1) ring_fill(tile_rsrc(a.Wln, cgw * 4 + wave));   // behind the last row loads, in flight under the arithmetic
#pragma unroll 2
        for (int r = 0; r < RH; ++r) {
            const int row = wave * RPW + half * RH + r, i = row0 + row;
            const f32x4 x = xr[r], y = yr[r];
            const float mean = wave_sum(((x[0] + x[1]) + (x[2] + x[3])) + ((y[0] + y[1]) + (y[2] + y[3]))) / (float)H;
            float q = 0.f;
            if (act) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float da = x[k] - mean, db = y[k] - mean;
                    q += da * da + db * db;
                }
            }
            const float var = wave_sum(q) / (float)H;
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
            if (act && i < N) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o0[k] = (x[k] - mean) * rstd * w0[k] + b0[k];
                    o1[k] = (y[k] - mean) * rstd * w1[k] + b1[k];
                }
                if (a.hf && tape_w) {
                    *reinterpret_cast<f32x4*>(a.hf + (size_t)i * H + c0) = o0;
                    *reinterpret_cast<f32x4*>(a.hf + (size_t)i * H + c0 + 4) = o1;
                }
                if (a.t_ln && tape_w) {
                    *reinterpret_cast<f32x4*>(a.t_ln + (size_t)i * a.ld_ln + c0) = o0;
                    *reinterpret_cast<f32x4*>(a.t_ln + (size_t)i * a.ld_ln + c0 + 4) = o1;
                }
            }
            if (a.t_lnstat && tape_w && lane == 0 && i < N) {
                a.t_lnstat[2 * (size_t)i] = mean;
                a.t_lnstat[2 * (size_t)i + 1] = rstd;
            }
            if (act && phaseB) {
                u32x4 pk[2];
                unsigned pr[3];
                pl_split_pair_acc(o0[0], o0[1], PL_S_LN, pr, sat_ln); pk[0][0] = pr[0]; pk[1][0] = pr[1];
                pl_split_pair_acc(o0[2], o0[3], PL_S_LN, pr, sat_ln); pk[0][1] = pr[0]; pk[1][1] = pr[1];
                pl_split_pair_acc(o1[0], o1[1], PL_S_LN, pr, sat_ln); pk[0][2] = pr[0]; pk[1][2] = pr[1];
                pl_split_pair_acc(o1[2], o1[3], PL_S_LN, pr, sat_ln); pk[0][3] = pr[0]; pk[1][3] = pr[1];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<u32x4*>(P + pl * PLB + row * ROWB + c0 * 2) = pk[pl];
            }
        }
        }
    }
    sat_report(sat_ln);
    if (!phaseB) return;
    __syncthreads();
    stamp();
    // ---- phase B: [P_i | P_j | X_part] = y Wln^T: the 128-column groups cgw, cgw + gy, ... of the 3H columns (lane = column, registers = rows) ----
    {
        constexpr float os = 1.f / (PL_S_LN * PL_SW);
        float m = 0.f;
#pragma unroll 1
        for (int g = cgw; g < 3 * NCG; g += gy) {
            const int ct = g * 4 + wave;
            run(TRf{}, tile_rsrc(a.Wln, ct));
            if (g + gy < 3 * NCG) ring_fill(tile_rsrc(a.Wln, (g + gy) * 4 + wave));
            const int col = ct * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = row0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                const float v = acc[r] * os;
                if (i < N) {
                    a.PQ[(size_t)i * (3 * H) + col] = v;
                    m = fmaxf(m, fabsf(v));
                }
            }
        }
        if (a.absmax) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            if (lane == 0) atomicMax(a.absmax, __float_as_uint(m));
        }
        stamp();
    }
}

template <int H, int D, int ST>
static int node_cols_launch_st(const NodeChainArgs& a, hipStream_t s) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] { attr_err = hipFuncSetAttribute((const void*)node_cols_kernel<H, D, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, NodeColsCfg<H>::LDS); });
    MI_HIP(attr_err);
    const int nblk = cdiv(a.N, 32);
    hipLaunchKernelGGL((node_cols_kernel<H, D, ST>), dim3(8 * a.gy * cdiv(nblk, 8)), dim3(256), NodeColsCfg<H>::LDS, s, a);
    MI_KERNEL_CHECK();
    return MI_OK;
}
template <int H, int D>
static int node_cols_launch(const NodeChainArgs& a, hipStream_t s) {
    switch (a.stages) {
        case 1: return node_cols_launch_st<H, D, 1>(a, s);
        case 2: return node_cols_launch_st<H, D, 2>(a, s);
        case 4: return node_cols_launch_st<H, D, 4>(a, s);
        case 7: return node_cols_launch_st<H, D, 7>(a, s);
    }
    return MI_EINVAL;
}

bool node_chain_supported(const mi_net* net) { return g_node_fused && net->cfg.ln && (net->H == 128 || net->H == 256 || net->H == 512) && net->Wnc != nullptr; }

size_t node_chain_pack_elems(int H) { return (size_t)6 * H * H * 2; }  // per layer: [Wagg | Wn2 | Wln (3H rows) | W2 (edge_mlp.2, edge_stage.hip)] x two planes (u16 elements)

// packs of layer l (called by mi_net_set_params): Wagg = node_mlp.0.weight[:, H:], Wn2 = node_mlp.2.weight, Wln = [W1[:, :H]; W1[:, H:2H]; node_mlp.0.weight[:, :H]]
int node_chain_pack(mi_net* net, int l, const float* W1, const float* Wn0, const float* Wn2, const float* W2, hipStream_t s) {
    const int H = net->H;
    u16* base = net->Wnc + (size_t)l * node_chain_pack_elems(H);
    const int nb = cdiv((int64_t)H * (H / 8), 256);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(nb), dim3(256), 0, s, Wn0 + H, 2 * H, H, H, base, 0);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(nb), dim3(256), 0, s, Wn2, H, H, H, base + (size_t)H * H * 2, 0);
    u16* wln = base + (size_t)2 * H * H * 2;
    hipLaunchKernelGGL(pack_frag_kernel, dim3(nb), dim3(256), 0, s, W1, net->edge_in, H, H, wln, 0);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(nb), dim3(256), 0, s, W1 + H, net->edge_in, H, H, wln, H);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(nb), dim3(256), 0, s, Wn0, 2 * H, H, H, wln, 2 * H);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(nb), dim3(256), 0, s, W2, H, H, H, base + (size_t)5 * H * H * 2, 0);
    MI_KERNEL_CHECK();
    return MI_OK;
}

// The chain in front of layer l's edge stage (l = 0 .. L; l = L: finishes the last layer and applies the final LayerNorm into b->hf).
static int node_chain_once(mi_net* net, mi_batch* b, int l, hipStream_t s, bool train);
int node_chain(mi_net* net, mi_batch* b, int l, hipStream_t s, bool train) {
    if (g_ablate_skip & 1) return MI_OK;
    if ((g_ablate_skip & 64) && !train) return MI_OK;   // (timing ablation: the INFERENCE forwards' chains only -- in a fine-tune micro-step: the frozen prior's)
    if (g_ablate_skip & 8) MI_TRY(node_chain_once(net, b, l, s, train));   // (timing experiment: every chain launched TWICE -- the second finds its weights in L2)
    return node_chain_once(net, b, l, s, train);
}
static int node_chain_once(mi_net* net, mi_batch* b, int l, hipStream_t s, bool train) {
    const int H = net->H, L = net->L, N = b->N;
    if (l > 0) count_mfma(N, 2 * H, H, MI_PLANES_TERMS);   // agg . Wagg^T and X . Wn2^T of layer l - 1
    if (l < L) count_mfma(N, 3 * H, H, MI_PLANES_TERMS);   // LayerNorm(h) . [W_Pi; W_Pj; W_node0[:, :H]]^T of layer l
    const size_t NH = (size_t)N * H;
    NodeChainArgs a;
    a.N = N;
    a.clk = g_node_clk;
    if (train) {   // the training forward: this chain writes what the backward reads of it (see NodeChainArgs)
        Tape& tp = b->tape;
        auto cat_of = [&](int layer) {
            return tp.wslots > 0 ? tp.w_cat + ((size_t)layer * tp.wslots * N + (size_t)tp.wcur * N) * 2 * H : tp.cat + (size_t)layer * N * 2 * H;
        };
        if (l > 0) {
            a.t_agg = cat_of(l - 1) + H;
            a.ld_agg = 2 * H;
            a.t_xpre = tp.Xpre + (size_t)(l - 1) * NH;
            a.t_ypre = tp.Ypre + (size_t)(l - 1) * NH;
        }
        if (l < L) {
            a.t_ln = cat_of(l);
            a.ld_ln = 2 * H;
        }
        a.t_lnstat = tp.lnstat + (size_t)l * N * 2;
    }
    if (l > 0) {
        const std::string p = "csp_layer_" + std::to_string(l - 1) + ".";
        const u16* base = net->Wnc + (size_t)(l - 1) * node_chain_pack_elems(H);
        a.part = b->part;
        a.rowptr = b->rowptr;
        a.seg_shift = b->seg_shift;
        if (b->seg_shift < 0) {   // (edge_fused.hip wrote the partial sums: slots by mask)
            a.slotmask = b->ef_mask;
            a.nslots = b->ef_nslots;
        }
        a.xpart = ((!train && l == 1 && b->PQ0) ? b->PQ0 : b->PQ) + 2 * H;   // (layer l - 1's X_part: layer 0's is in its own buffer on the inference path, mi_batch::PQ0)
        a.ld_xpart = 3 * H;
        a.Wagg = base;
        a.Wn2 = base + (size_t)H * H * 2;
        a.b0 = net->p(p + "node_mlp.0.bias");
        a.b2 = net->p(p + "node_mlp.2.bias");
        a.dsc = b->dsc;
        a.h_in = b->h + (size_t)(l - 1) * NH;
        a.h_out = b->h + (size_t)l * NH;
    } else {
        a.h_in = b->h;
    }
    if (l < L) {
        const std::string p = "csp_layer_" + std::to_string(l) + ".";
        a.ln_w = net->p(p + "layer_norm.weight");
        a.ln_b = net->p(p + "layer_norm.bias");
        a.Wln = net->Wnc + (size_t)l * node_chain_pack_elems(H) + (size_t)2 * H * H * 2;
        a.PQ = (!train && l == 0 && b->PQ0) ? b->PQ0 : b->PQ;
        a.absmax = b->absmax + 2 * l;
    } else {
        a.ln_w = net->p("final_layer_norm.weight");
        a.ln_b = net->p("final_layer_norm.bias");
        a.hf = b->hf;
    }
#if MI_HAVE_ABLATION_KERNELS   // (the deeper weight ring spills registers: an ablation instantiation, see gemm_split.h)
    if (H == 512 && g_node_fused == 2) return node_chain_launch<512, 8, 8>(a, s);
#endif
    auto launch = [&](const NodeChainArgs& x) {
        if (H == 512) return node_chain_launch<512, 8, 4>(x, s);
        if (H == 256) return node_chain_launch<256, 8, 4>(x, s);
        return node_chain_launch<128, 4, 4>(x, s);
    };
    // The column-split form (mi_debug_set_node_cols; node_cols_kernel): three launches per layer boundary -- or one with in-launch hand-overs -- of light
    // workgroups that own 32 rows x 128 columns of one product each.
    if (g_node_cols && (g_node_cols != 3 || cdiv(N, 32) <= g_node_cols_auto_blocks) && !a.slotmask && b->Xpl != nullptr) {   // (H = 128 / 256 / 512: node_chain_supported)
        constexpr int NCGmax = 4;
        const int nblk = cdiv(N, 32), NCG = H / 128;
        (void)NCGmax;
        auto launch_cols = [&](const NodeChainArgs& x) {
            if (H == 512) return node_cols_launch<512, 16>(x, s);
            if (H == 256) return node_cols_launch<256, 8>(x, s);
            return node_cols_launch<128, 4>(x, s);
        };
        a.xpl = b->Xpl;
        a.npad = nblk * 32;
        // workgroups per row block of the LayerNorm + projection stage: one 128-column group each while the launch still fits the chip at two
        // workgroups per CU, otherwise NCG (three groups each)
        const int gyB = l < L ? (nblk * 3 * NCG <= g_node_cols_b_max ? 3 * NCG : NCG) : 1;
        const bool one_launch = g_node_cols == 2 && l > 0 && l < L && b->nc_flags != nullptr && nblk * NCG <= g_node_cols_fused_max;
        if (one_launch) {   // stages 1 | 2 | 4 behind two hand-overs; the row block's arrival counters were zeroed with the absmax slots
            a.stages = 7;
            a.gy = NCG;
            a.flags = b->nc_flags + (size_t)l * 2 * nblk;
            return launch_cols(a);
        }
        if (l > 0) {
            NodeChainArgs a1 = a;
            a1.stages = 1;
            a1.gy = NCG;
            a1.Wln = nullptr;
            MI_TRY(launch_cols(a1));
            a1.stages = 2;
            MI_TRY(launch_cols(a1));
        }
        NodeChainArgs a2 = a;
        a2.stages = 4;
        a2.gy = gyB;
        a2.part = nullptr;
        a2.h_in = b->h + (size_t)l * NH;
        a2.t_agg = a2.t_xpre = a2.t_ypre = nullptr;
        return launch_cols(a2);
    }
    // The chain is bound by the weight planes each workgroup streams through ITS CU's L2 port (1 MB per product: DESIGN 18.6); the three
    // projection passes are independent given LayerNorm(h'), so when three workgroups per row block still fit the chip in one round they
    // run as a second launch on three times the CUs (each recomputes the LayerNorm of its 32 rows).  Larger batches keep the one launch.
    const int nblk = cdiv(N, 32);
    a.touch = g_node_touch_min_blocks > 0 && nblk >= g_node_touch_min_blocks;
    if (g_node_split && l < L && 3 * nblk <= g_node_split_max_blocks) {
        if (l > 0) {
            NodeChainArgs a1 = a;
            a1.a_only = 1;
            a1.Wln = nullptr;
            a1.t_ln = nullptr;
            a1.t_lnstat = nullptr;
            MI_TRY(launch(a1));
        }
        NodeChainArgs a2 = a;
        a2.part = nullptr;                       // no phase A: the chain starts at h' (l = 0: at the embedding's output, as before)
        a2.h_in = b->h + (size_t)l * NH;
        a2.t_agg = a2.t_xpre = a2.t_ypre = nullptr;
        a2.split_b = 1;
        return launch(a2);
    }
    return launch(a);
}

#else  // three-plane bf16 build: the chain stays on the plane GEMMs

bool node_chain_supported(const mi_net*) { return false; }
size_t node_chain_pack_elems(int) { return 0; }
int node_chain_pack(mi_net*, int, const float*, const float*, const float*, const float*, hipStream_t) { return MI_OK; }
int node_chain(mi_net*, mi_batch*, int, hipStream_t, bool) { return MI_ESTATE; }

#endif

}  // namespace mi

extern "C" int mi_debug_node_chain_clock(void* dev_buffer) {
    mi::g_node_clk = (unsigned long long*)dev_buffer;
    return MI_OK;
}

extern "C" int mi_debug_set_node_split(int on) {
    const int was = mi::g_node_split;
    mi::g_node_split = on != 0;
    if (on > 1) mi::g_node_split_max_blocks = on;   // (experiments: the largest 3 x row-block count that still takes the two-launch form; default 256)
    return was;
}

extern "C" int mi_debug_set_node_touch(int min_blocks) {
    const int was = mi::g_node_touch_min_blocks;
    mi::g_node_touch_min_blocks = min_blocks;
    return was;
}

extern "C" int mi_debug_set_skip(int mask) {
    const int was = mi::g_ablate_skip;
    mi::g_ablate_skip = mask;
    return was;
}

extern "C" int mi_debug_set_node_cols(int mode) {
    const int was = mi::g_node_cols;
    if (mode < 0 || (mode & 255) > 3) return MI_EINVAL;
    mi::g_node_cols = mode & 255;
    if (mode >> 8) mi::g_node_cols_b_max = (mode >> 8) - 1;   // (experiments: the workgroup count up to which LayerNorm + projections take one column group per workgroup)
    return was;
}

extern "C" int mi_debug_set_node_train(int on) {
    const int was = mi::g_node_train;
    mi::g_node_train = on != 0;
    return was;
}

extern "C" int mi_debug_set_node_fused(int on) {
    const int was = mi::g_node_fused;
    mi::g_node_fused = on;  // (2: the deeper weight ring at hidden_dim 512 -- ablation)
    return was;
}
