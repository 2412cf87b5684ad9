// Host-side objects behind the opaque C handles.
#pragma once
#include <mutex>

#include "common.h"
#include "gemm.h"

struct ParamInfo {
    std::string name;
    int64_t off, numel;
    int rows, cols;
};

struct mi_net {
    mi_net_config cfg;
    int H, L, F, TD, KP, NT;  // KP = 3*FP (sin,cos) pairs, FP = F rounded up to 8; NT = H/32
    int edge_in;              // 2H + 9 + 6F
    std::vector<ParamInfo> params;
    int64_t nparams = 0;
    const float* theta = nullptr;  // caller-owned flat parameters (device)
    uint64_t param_epoch = 0;      // bumped by every mi_net_set_params: what an evaluation left behind for a later one (mi_batch::G, PQ0) is valid for ONE parameter version
    float* freqs = nullptr;        // [F] device
    bool have_freqs = false;
    // packed copies, rebuilt by mi_net_set_params
    float* Whh = nullptr;    // [L][2H][H]   rows [0,H) = W1[:, :H], rows [H,2H) = W1[:, H:2H]
    float* Wff_p = nullptr;  // [L][KP/4][NT][64][4]
    float* W2_p = nullptr;   // [L][NT][NT][4][64][4]
    float* Wff = nullptr;    // [L][H][6F]  Fourier column block of edge_mlp.0, contiguous (GEMM edge path)
    int edge_mode = 1;       // MI_EDGE_GEMM (default) or MI_EDGE_FUSED_F32
    unsigned short* Wffpl = nullptr;  // [L][3][H][ld(6F)] bf16 planes of Wff
    unsigned short* W2pl = nullptr;   // [L][3][H][H]      bf16 planes of edge_mlp.2.weight
    // [L] plane sets of the node-level weights, grouped by the operand they multiply:
    unsigned short* Wlnpl = nullptr;   // (3H x H): [P_i block; P_j block; node_mlp.0.weight[:, :H]] -- everything LayerNorm(h) feeds
    unsigned short* Waggpl = nullptr;  // (H x H):  node_mlp.0.weight[:, H:]  (multiplies the aggregated messages)
    unsigned short* Wn2pl = nullptr;   // (H x H):  node_mlp.2.weight
    unsigned short* Wffc = nullptr;    // [L] the pair-layout Fourier block of edge_mlp.0 in MFMA fragment order (edge_stage.hip: first edge GEMM)
    unsigned short* Wffc2 = nullptr;   // [L] -2 x its sine block in the same order (the second pass of edge_gemm1e_kernel)
    unsigned short* Wnc = nullptr;     // [L] the three operands above in MFMA fragment order: [Wagg | Wn2 | Wln] (node_chain.hip; H = 128 / 256 / 512 with LayerNorm)
    float* wbounds = nullptr;          // [L][8] row-sum / bias bounds of the layer's weights (fp16 plane format: activation scales)
    // pair mode of the first edge GEMM (symmetric edge lists): K' = 2*Kh columns = [sin block | pad | cos block | pad],
    // Kh = 3F rounded up to 32, so that each block is a whole number of k-tiles
    int Kh = 0;
    unsigned short* Wffpl_pair = nullptr;  // [L] plane sets of the Fourier block of edge_mlp.0 in that column layout
    float* C0 = nullptr;                   // [L][H] sum of the cosine-block weights = the Fourier term of a self edge (d = 0)
    // transposed copies for the data-gradient GEMMs (training), rebuilt with the packs
    float* W2T = nullptr;    // [L][H][H]
    unsigned short* W2Tpl = nullptr;  // [L] plane sets of W2T (fp16 plane format): the W operand of the dM1 data gradient
    unsigned short* W2Tf = nullptr;   // [L] the same in MFMA fragment order (Planes::frag: the register-tile GEMM, hidden_dim % 256 == 0)
    float* Wn2T = nullptr;   // [L][H][H]
    float* Wn1T = nullptr;   // [L][2H][H]
    float* WhhT = nullptr;   // [L][H][2H]
    float* WaT = nullptr;    // [H][H]   (atom_latent_emb.weight[:, :H])^T
    unsigned short* Wbw = nullptr;   // [L] fragment-order packs of the TRANSPOSED node-level weights [WhhA | WhhB | Wn2^T | Wn0^T]: the fused backward chain (node_bwd.hip)
    float* WheadT = nullptr; // [H][104] coord_out.weight (3 rows) and type_out.weight (100 rows) transposed side by side (inference heads)
    // profiling of the dominant kernel (event pairs; launches may come from several host threads / streams)
    bool prof = false;
    std::vector<hipEvent_t> ev;  // pairs
    size_t ev_used = 0;
    hipEvent_t ev_origin = nullptr;
    std::mutex prof_mu;

    int64_t off(const std::string& name) const;
    const float* p(const std::string& name) const { return theta + off(name); }
    size_t whh_stride() const { return (size_t)2 * H * H; }
    size_t wff_stride() const { return (size_t)(KP / 4) * NT * 256; }
    size_t w2_stride() const { return (size_t)NT * NT * 4 * 256; }
};

// Saved activations of one training forward + backward scratch (allocated on first use).
struct Tape {
    bool allocated = false, valid = false;
    float *cat = nullptr, *Z1 = nullptr, *Z2 = nullptr, *Xpre = nullptr, *Ypre = nullptr, *lnstat = nullptr, *gf = nullptr;
    float *atom_types = nullptr, *t_emb = nullptr, *lattices = nullptr, *frac = nullptr;
    // what the backward reads of the forward's INPUTS: the tape's copies above (mi_cspnet_forward_train: the caller may overwrite its arrays before the
    // backward), or -- `borrow_inputs`, set by the fused micro-step around its own training forward -- the caller's arrays themselves (they are the
    // tape's noised inputs and the batch's time embedding, alive until the micro-step's backward has run: four copy launches fewer per micro-step)
    const float *in_types = nullptr, *in_temb = nullptr, *in_lat = nullptr, *in_frac = nullptr;
    bool borrow_inputs = false;
    float *dh = nullptr, *dY = nullptr, *dXa = nullptr, *Xa = nullptr, *dcat = nullptr, *dPQ = nullptr, *dG = nullptr, *dgf = nullptr,
          *dlo = nullptr, *dtproj = nullptr, *M1 = nullptr, *dM1 = nullptr, *FF = nullptr, *scratch = nullptr;
    // fused fine-tune micro-step: noised inputs, targets, gradient seeds, per-crystal losses
    float *nz_lat = nullptr, *nz_frac = nullptr, *nz_types = nullptr, *tar_x = nullptr, *rnd_l = nullptr, *rnd_t = nullptr, *d_l = nullptr,
          *d_x = nullptr, *d_t = nullptr, *Lb = nullptr, *KLb = nullptr;
    float* lnpart = nullptr;      // [L][ceil(N / 32)][2H] per-workgroup partial sums of d ln_w | d ln_b of the fused backward chain (node_bwd.hip), reduced for all layers at once
    float* dsc_layers = nullptr;  // [L][8] each layer's activation scales of the training forward (fp16 plane format, folded launch)
    // fp16 plane format: every layer's M1 plane set of the training forward, kept for edge_mlp.2's weight gradient (gemm_tn_planes, backward.hip:
    // the product reads both operands as the plane sets their producers wrote instead of re-splitting -- and re-evaluating SiLU on -- fp32 rows)
    unsigned short* M1pl_l = nullptr;
    size_t m1pl_stride = 0;       // elements per layer
    // and the pair differences / sums of dZ1 (the A operands of the Fourier block's weight gradient) as plane sets, written by the fused pair pass
    unsigned short *DmPl = nullptr, *DpPl = nullptr;
    bool dsc_layers_valid = false;
    size_t scratch_floats = 0;
    // Deferred node-level weight gradients (mi_batch_set_wgrad_window).  Between two optimizer steps the weights do not change, so the
    // weight gradient of a node-level linear over `wslots` micro-steps is ONE contraction over wslots x N rows instead of wslots
    // contractions over N rows (N = 1.7k atoms per crystal group: 24 launches of ~20 us plus as many partial reductions per evaluation,
    // each leaving most of the chip idle).  The operand pairs of every layer are kept per micro-step, slot after slot, rows contiguous:
    //   w_dY | w_Xa  -> node_mlp.2.weight / bias      w_dXa | w_cat -> node_mlp.0.weight / bias      w_dPQ | w_cat[:, :H] -> edge_mlp.0.weight[:, :2H]
    // layout [L][wslots x N][width]; the training forward writes its `cat` rows straight into slot `wcur`, the backward its dY / Xa / dXa /
    // dPQ; net_wgrad_flush contracts rows 0 .. wcur x N and resets wcur (automatically when the window is full).
    int wslots = 0, wcap = 0, wcur = 0;
    float *w_dY = nullptr, *w_Xa = nullptr, *w_dXa = nullptr, *w_cat = nullptr, *w_dPQ = nullptr;
    // ... and (round 6) the HEAD and EMBEDDING weight gradients in the same window -- six short contractions + three column sums per micro-step, each with its
    // own partial reduction (18 launches), become the same nine over wslots x the rows.  Their operand rows, layout [wslots][rows][width], slot `wcur`:
    //   w_dlo [B,12] | w_gf [B,H] -> lattice_out.weight      w_dtype [N,100] | w_hf [N,H] -> type_out.weight / bias      w_dcoord [N,3] | w_hf -> coord_out.weight
    //   w_dh [N,H] | w_x1 [N,H] -> atom_latent_emb.weight[:, :H] / bias      w_dtproj [B,H] | w_temb [B,TD] -> atom_latent_emb.weight[:, H:]
    //   w_eXa [N,H] | w_types [N,100] -> node_embedding.weight / bias
    // The training forward and the backward write them IN PLACE: while a window is open, the batch's x1 / hf, the tape's gf / dlo / dtproj / d h and the
    // micro-step's noised types, time embedding and gradient seeds LIVE in the slot (the pointer accessors below); nothing is copied on the fused route.
    float *w_dlo = nullptr, *w_gf = nullptr, *w_dtype = nullptr, *w_dcoord = nullptr, *w_hf = nullptr, *w_dh = nullptr, *w_x1 = nullptr, *w_dtproj = nullptr,
          *w_temb = nullptr, *w_eXa = nullptr, *w_types = nullptr;
    bool head_window() const { return wslots > 0 && w_hf != nullptr; }
};

struct mi_batch {
    int B = 0, N = 0, H = 0, L = 0;
    int64_t E = 0;      // edges of the current graph (fixed for fc; rewritten by every forward for knn)
    int64_t E_cap = 0;  // edge capacity every per-edge buffer is sized for (= E for fc)
    const int* e_dev = nullptr;   // knn lists rebuilt WITHOUT a host round trip (knn_build nosync): E above is the capacity the launches are sized for, *e_dev the edge count
    int64_t e_hint = 0;           // the last edge count the HOST has seen (a synchronising build, mi_knn_graph_status): picks kernel forms for capacity-sized launches
    bool knn_nosync = false;      // set by mi_sampler_run around its forwards: the chain's graph builds do not synchronise (mi_knn_graph_status reads the verdict behind it)
    int nslots = 1;
    int seg_shift = 5;  // log2 of the row-block size behind the partial sums currently in `part` (5: plane GEMM epilogue; 7: edge_stage.hip; -1: edge_fused.hip, slots by mask)
    // edge_fused.hip (fc pair mode): 64-pair tiles.  ef_tile0[v] = first pair tile of v's crystal, ef_mask[v] = slots (tile - ef_tile0, and
    // ef_nslots - 1 for the self-edge tile) that hold a partial sum of v; ef_ok: every tile's node range fits 128 and ef_nslots <= 32
    int* ef_tile0 = nullptr;
    unsigned* ef_mask = nullptr;
    int ef_nslots = 0;
    bool ef_ok = false;
    // knn edge style (CSPNet.gen_edges knn branch, graph.hip)
    int knn = 0, max_neighbors = 0, cap_per_node = 0, deg_cap = 0, nmax = 0;
    // pair tables of the fc edge list (unordered node pairs i < j of each crystal): the first edge GEMM runs over pairs
    int64_t Np = 0;
    int *pair_i = nullptr, *pair_j = nullptr, *pair_e1 = nullptr /*edge i->j*/, *pair_e2 = nullptr /*edge j->i*/, *pair_graph = nullptr,
        *e_diag = nullptr /*[N] self edge of node i*/;
    int* pair_off = nullptr;  // [B+1] first pair row of each crystal (fc list); nmax_fc = largest atom count
    int nmax_fc = 0;
    float* fd = nullptr;    // [E][3] explicit frac_diff per edge (CSR order); nullptr = fc: (x_dst - x_src) % 1
    int* inedge = nullptr;  // [E] edge ids grouped by destination node, node v at rowptr[v]..rowptr[v+1] (degrees are symmetric)
    int *kn_ent = nullptr, *kn_acnt = nullptr, *kn_deg = nullptr, *kn_mcount = nullptr, *kn_eoff = nullptr, *kn_meta = nullptr,
        *kn_refpos = nullptr, *r_src = nullptr, *r_dst = nullptr;
    float* r_vec = nullptr;  // reference-order copy of the list: edges (r_src, r_dst), attribute r_vec
    int64_t node_offset = 0, graph_offset = 0;
    std::vector<int> num_atoms_h, node_off_h;
    // index tables (device)
    int *num_atoms = nullptr, *node_off = nullptr /*[B+1]*/, *node2graph = nullptr, *src = nullptr, *dst = nullptr,
        *rowptr = nullptr /*[N+1]*/, *edge_graph = nullptr /*[E]*/;
    // forward workspace (device)
    float *x1_base = nullptr, *hf_base = nullptr;   // the handle's own x1 / hf buffers (x1 / hf point into the weight-gradient window's slot while a training forward with an open window runs)
    float* h = nullptr;      // [L+1][N][H] node features before layer l / after the last
    float* cat = nullptr;    // [N][2H]  (LN(h) | agg)
    float* PQ = nullptr;     // [N][2H]
    float* PQ0 = nullptr;    // layer 0's [P_i | P_j | X_part] of the INFERENCE node chain, in a buffer of its own: the sampler's predictor evaluation starts from
                             // the same embedding as the corrector evaluation in front of it, so layer 0's LayerNorm + projections launch is not repeated (pq0_valid)
    bool pq0_valid = false;
    const void* reuse_net = nullptr;   // the network whose evaluation left G / PQ0 behind (another network's evaluation never reuses them)
    uint64_t reuse_epoch = 0;          // ... and its parameter version (mi_net::param_epoch): an optimizer step in between invalidates both
    float* G = nullptr;      // [B][H]
    float* part = nullptr;   // [nslots][N][H]
    float* FFp = nullptr;    // [tiles][KP/4][64][4] Fourier operand, B-fragment order (fused path)
    float* FF = nullptr;     // [E][6F] Fourier features, reference column order (GEMM path)
    float* M1 = nullptr;     // [E][H]
    float* M2 = nullptr;     // [E][H]
    unsigned short* FFpl = nullptr;  // [3][E][ld(6F)] bf16 planes of the Fourier features
    unsigned short* M1pl = nullptr;  // [3][E][H]      bf16 planes of M1
    unsigned short* m1_cur = nullptr;  // the plane set this layer's edge products write / read: M1pl, or the layer's slot of Tape::M1pl_l in a training forward
    unsigned short* lnpl = nullptr;  // plane sets of LayerNorm(h) and of the aggregated messages (N x H each)
    unsigned short* aggpl = nullptr;
    unsigned short* Xpl = nullptr;   // plane set of the node MLP's hidden activation (N x H)
    float* dsc = nullptr;            // [6][2] {scale, 1/scale}: this layer's M1 / agg / X plane sets; backward pass: dZ2, the pair differences, the Fourier features (fp16 plane format)
    unsigned* absmax = nullptr;      // [2 L] bit patterns: [2l] = max |P_i, P_j, X_part| of layer l, [2l + 1] = max |G[l]| (zeroed per evaluation); [2L], [2L + 1] = max |d cat|, max |dM1| of the layer the backward pass is in
    int* e2_tab = nullptr;           // per-tile tables of the second edge GEMM (edge_stage.hip: EG2_TAB ints per 128-row tile), valid for graph_epoch == e2_tab_epoch
    bool gram_valid = false;         // b->G and its absmax slots hold the lattice term of the previous (inference) evaluation of this handle
    bool ff_built_once = false;      // (timing ablation mi_debug_set_skip bit 4 only: the Fourier operand exists)
    int e2_tab_epoch = -1, graph_epoch = 0;   // graph_epoch: bumped whenever the edge list changes (knn: every forward)
    unsigned* nc_flags = nullptr;    // [L + 1][2 x row blocks] arrival counters of node_cols_kernel's in-launch hand-overs (zeroed per evaluation)
    float* X = nullptr;      // [N][H] node-MLP hidden
    float* x1 = nullptr;     // [N][H] node_embedding output
    float* tproj = nullptr;  // [B][H]
    float* hf = nullptr;     // [N][H] after the final LayerNorm
    // sampler scratch
    float* temb = nullptr;     // [B][TD]
    int* times = nullptr;      // [B]
    float* pred_l = nullptr;   // [B][9]
    float* pred_x = nullptr;   // [N][3]
    float* pred_t = nullptr;   // [N][100]
    float* x_mid = nullptr;    // [N][3]
    float* lp_corr = nullptr;  // [B]
    float* coef = nullptr;     // [T+1][MI_NCOEF]
    int coef_T = -1;
    std::vector<float> coef_h;  // host copy of what `coef` holds (re-uploaded only when the caller's table differs)
    int keep_lattice = 0, keep_coords = 0;  // CSP mode of the sampler (mi_sampler_set_keep)
    mi::SplitK sk;  // split-K scratch of the node-level products (small batches only)
    // optional helper stream of higher priority for the node-level kernels of an inference forward (see net_forward) + its join events
    hipStream_t hi_stream = nullptr;
    hipEvent_t hi_ev[2] = {nullptr, nullptr};
    Tape tape;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;  // fork / join of work put on an auxiliary stream (mi_ft_micro_step)
    std::vector<void*> allocs;
};

namespace mi {
// reuse_embedding -- the CONTRACT (internal; the only caller is mi_sampler_run's predictor evaluation, diffusion.py:320-322): this evaluation has the same
// t_emb, atom_types AND lattices as the previous evaluation of this batch handle by this network at this parameter version; only the coordinates moved.
// It keeps h[0], and (inference, pair mode) the lattice term G of every layer and layer 0's [P_i | P_j | X_part].  The handle remembers the network and
// its parameter epoch, so another network or an mi_net_set_params in between falls back to recomputing G / PQ0; unchanged lattices are the caller's promise.
int net_forward(mi_net* net, mi_batch* b, const float* t_emb, const float* atom_types, const float* frac,
                const float* lattices, float* lattice_out, float* coord_out, float* type_out, hipStream_t s, bool train = false,
                bool reuse_embedding = false, bool coords_only = false);
int net_tape_prepare(mi_net* net, mi_batch* b);
int net_pack_transposes(mi_net* net, hipStream_t s);
int net_backward(mi_net* net, mi_batch* b, const float* d_lat, const float* d_coord, const float* d_type, float* grad, hipStream_t s);
int net_wgrad_window(mi_net* net, mi_batch* b, int slots);                   // 0: every backward contracts its own rows (default)
int net_wgrad_flush(mi_net* net, mi_batch* b, float* grad, hipStream_t s);   // grad += the pending micro-steps' node-level weight gradients
template <typename T>
int dev_alloc(mi_batch* b, T** p, size_t n);
int knn_alloc(mi_batch* b, int max_neighbors, int cap_per_node);
// node_bwd.hip: the node-level BACKWARD chain between two edge stages of the backward pass as one launch per layer boundary
extern int g_node_bwd, g_node_bwd_min_blocks;
size_t node_bwd_pack_elems(int H);
bool node_bwd_supported(const mi_net* net, const mi_batch* b);
int node_bwd_pack(mi_net* net, int l, const float* W1, const float* Wn0, const float* Wn2, hipStream_t s);
int node_bwd(mi_net* net, mi_batch* b, int l, const float* dPQ, float* dh, float* dY, float* dXa, float* Xa, float* lnpart, unsigned* dcat_absmax, hipStream_t s);
// node_chain.hip: the node-level chain between two edge stages of an inference forward as one launch
bool node_chain_supported(const mi_net* net);
size_t node_chain_pack_elems(int H);
int node_chain_pack(mi_net* net, int l, const float* W1, const float* Wn0, const float* Wn2, const float* W2, hipStream_t s);
// edge_stage.hip: the second edge GEMM + edge -> node reduction on 128-row x H-column register tiles (inference, hidden_dim 512)
bool edge_gemm1_supported(const mi_net* net, int64_t M);
int edge_gemm1_pack(mi_net* net, int l, const float* W1, hipStream_t s);
bool edge_gemm2_supported(const mi_net* net);
extern int g_edge2_train;
extern int g_node_train;
extern int g_node_cols;
extern int g_ablate_skip;
int edge_fused(mi_net* net, mi_batch* b, int layer, hipStream_t s);   // edge_fused.hip: both edge products of a layer in one launch (M1 stays in LDS)
bool edge_fused_supported(const mi_net* net, const mi_batch* b);
int edge_gemm2(mi_net* net, mi_batch* b, int layer, hipStream_t s, float* Z2 = nullptr);   // Z2: optional pre-activation output (training forward)
int node_chain(mi_net* net, mi_batch* b, int l, hipStream_t s, bool train = false);
extern int g_knn_nosync;
int knn_build(mi_batch* b, const float* frac, const float* lattices, hipStream_t s, bool nosync = false);
}  // namespace mi
