/* Debug / experiment entry points of libmatinvent_hip.so -- NOT part of the drop-in boundary.
 *
 * A maintainer binding the library for the reference binds include/matinvent_hip.h only.  Everything here selects between kernel
 * forms that compute the same result (A/B measurements, the parity tests' cross-checks), installs phase clocks, or -- where the
 * comment says so -- skips work for TIMING ablations.  The defaults are what the library ships with; tests/test_abi.py checks that
 * every name declared here is exported and bound by matinvent_amd/_lib.py like the product entry points.
 */
#ifndef MATINVENT_HIP_DEBUG_H
#define MATINVENT_HIP_DEBUG_H

#include "matinvent_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics: C[M,N] = A[M,K] W[N,K]^T through the node-level GEMM kernels.  kind 0 = f32-input MFMA,
 * kind 1 = three-plane bf16 split (six product terms, fp32-class) on the bf16 matrix pipe; kinds 2 / 3 = the same on pre-split
 * tile-blocked plane sets (128-row / 256-row double-buffered kernel); kind 4 = the weight-gradient form C[M,N] += A^T W with
 * A [K,M] and W [K,N] (contraction over rows, f32 MFMA, deterministic split reduction); kind 5 = the same product from fp16 plane
 * sets of both operands (split here with scale 2^6; LDS-DMA slabs + transposing LDS reads: the edge-level weight gradients of
 * `loss.backward()`, pipeline/mat_invent.py:164; M % 256 == 0, N % 128 == 0, K >= 4096). */
int mi_debug_gemm(int kind, const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M,
                  int N, int K, void* stream);
/* bench.py's roofline hook: HIP events bracket the dominant stage (the per-edge MLP of one layer) on the stream it is
 * launched on.  mi_profile_read returns the number of bracketed launches, the sum of their durations and (optional) the
 * UNION of their execution intervals -- with several chains running concurrently on different streams the launches overlap,
 * and total flops / union time is the rate the stage sustains while at least one instance is executing. */
/* One wave busy-waiting ~`cycles` shader clocks on `stream`: lets the host side test whether two HIP streams really execute
 * concurrently (streams may share a hardware queue, which serialises them). */
int mi_debug_spin(long long cycles, void* stream);
/* Test hook for SinusoidsEmbedding (models/diffcsp/cspnet.py:12-24) as the edge stage consumes it: the pair-mode Fourier operand of
 * Np atom pairs (i, j), d = (frac[j] - frac[i]) % 1, rebuilt as fp32 from its plane set -- out[Np][6F] = [sin(2 pi k d_c) | cos(...)],
 * column c * F + k.  The product path never materialises these values in fp32; the tests compare them with the reference's golden
 * embedding.  All pointers are device pointers; synchronises `stream`. */
int mi_debug_fourier_pairs(const float* frac, const int* pair_i, const int* pair_j, int64_t Np, int F, float* out, void* stream);
/* Tuning knob: smallest number of 256-row tiles for which the double-buffered plane GEMM is used (default 256). */
int mi_debug_set_db_min_tiles(int n);
/* Tuning knob: smallest node count for which the node-level products (P_i/P_j projections, node MLP) run on the plane-set
 * GEMM kernel (pre-split weights, producer-written activation planes); smaller batches use the fp32-operand split-K kernel. */
int mi_debug_set_node_planes_min_rows(int n);
/* Tuning knob for the weight-gradient products over long row lists (edge / pair list): 3 (default) = bf16 three-plane split on
 * the matrix pipe (split arithmetic path only), 1 = f32 MFMA on 128 x 128 tiles, 0 = f32 MFMA on 64 x 64 tiles; +8 = the separate
 * dZ1-consumer kernels instead of the fused fc pair-mode backward pass, +16 = a separate silu(Z1) pass instead of forming M1 inside
 * the weight-gradient product's operand load, +32 = the dM1 data gradient on the on-the-fly three-plane bf16 split instead of the
 * pre-split fp16 plane GEMM, +64 = the edge-level weight gradients on three bf16 planes / six terms instead of two fp16 planes /
 * three, +128 = the thread-per-column form of the fused pair-mode backward pass instead of the LDS-tile form, +256 = edge_mlp.2's weight gradient
 * from fp32 rows instead of the M1 / dZ2 plane sets, +512 = the head / embedding weight gradients contracted in every backward instead of in the
 * deferred window of mi_batch_set_wgrad_window (applies to windows sized afterwards) (ablations). */
int mi_debug_set_tn128(int on);
/* Tuning knob: shortest row list (contraction length) for which the bf16-pipe weight-gradient kernel is used (default 4096). */
int mi_debug_set_tn_split_min_rows(int n);
/* Workgroups a long weight-gradient contraction C += A^T X (the backward of every linear layer over the edge / pair / node lists:
 * pipeline/mat_invent.py:164 `loss.backward()`) is split into along its row list; every split writes a partial tile that a fixed-order
 * reduction adds into C.  Default 768 (three per CU for a product that has the chip to itself).  Returns the previous value. */
int mi_debug_set_tn_target_tiles(int n);
/* The pair-mode first edge GEMM's epilogue (models/diffcsp/cspnet.py:59-79) addresses its gathered operands and its output plane set with
 * 32-bit offsets off scalar bases when their sizes allow (below ~1.38 M edges per chain) and with 64-bit pointers otherwise, same results:
 * 1 = the 64-bit form whatever the sizes (tests), 0 (default) = by size.  Returns the previous setting. */
int mi_debug_set_pair_wide(int on);
/* Tuning knob: plain plane GEMMs with fewer 128x128 output tiles than this run on 64-row tiles (more, shorter workgroups); default 0 = never. */
int mi_debug_set_planes_small_tiles(int n);
/* Plain plane-set products (row-major epilogue) with at least `min_rows` rows (default 65536; <= 0 keeps the limit) and N % 256 == 0
 * run on the 256 x 256-tile kernel that stages its operands by LDS-DMA (fp16 two-plane build): 1 (default) / 0 = the 128 x 128
 * kernel everywhere.  Same accumulation order per output: bit-identical results (tests/test_gpu_gemm.py). */
int mi_debug_set_planes_big(int on, int min_rows);
/* Plane-set products with a row-major epilogue whose W operand carries a fragment-order copy (the MatterGen-shaped network's
 * edge-level dense layers), N % 256 == 0, K % 64 == 0, at least `min_rows` rows (default 16384; <= 0 keeps the limit), on the
 * 128-row x 256-column register-tile kernel of csrc/edge_stage.hip (W straight from L2 into registers, A by LDS-DMA):
 * 0 = never, 1 = the products with epilogue extensions (plane-set residuals, second merge, multiplicand), 2 (default) = all of
 * them (MatterGen-shaped sampler at the benchmark size: 2.53 / 2.56 / 2.58 structures/s with 0 / 1 / 2).  Same products in the same k order, same epilogue function: bit-identical to the 128 x 128 kernel. */
int mi_debug_set_planes_rt(int mode, int min_rows);
/* Plane-set products whose launch is at most `max_blocks` workgroups (default 256 = one per CU; 0 = never) run the LATENCY form of the
 * 128 x 128 kernel: three operand register sets, loads three k-tiles ahead.  Such launches (short edge lists -- the reference's default
 * sampling and fine-tune batches, models/diffcsp/sample.py:42-62 -- and node-level products) are one round of workgroups whose k-loop
 * is a chain of memory latencies.  Same accumulation order per output: bit-identical results (tests/test_gpu_gemm.py). */
int mi_debug_set_planes_latency(int max_blocks);
/* The second edge GEMM of an inference forward (SiLU + fused segmented sum epilogue, cspnet.py:79) with at least `min_rows` edges on
 * the 256 x 256 LDS-DMA kernel as well; 0 (default) = never.  Bit-identical partial sums (same accumulation order). */
int mi_debug_set_planes_big_seg(int min_rows);
/* Experiment: 1 = the node-level kernels of an inference forward (LayerNorm, the node-level products, the aggregation's last pass) run on
 * a helper stream of the highest priority owned by the batch handle, joined to the caller's stream by events at every hand-over;
 * 0 (default) = everything on the caller's stream.  Same kernels, same order of dependent work: identical results.
 * +2 = the coordinate / type heads of an inference forward as two fp32-operand GEMM launches instead of the fused heads kernel (ablation). */
int mi_debug_set_node_priority(int on);
/* The 128 x 128-tile plane product with its operands staged by LDS-DMA (`buffer_load ... lds` into two 32 KiB stages, fragments
 * software-pipelined over two register sets, one barrier per k-tile): 0 = never, 1 (default) = launches of at most the latency
 * limit above, 2 = every launch of the 128-row kernel.  Bit-identical to the register-staged loop (tests/test_gpu_gemm.py). */
int mi_debug_set_planes_dma(int mode);
/* The node-level chain between two edge stages of an inference forward (segmented mean, node MLP with residual, LayerNorm, the
 * projections LayerNorm(h) feeds: models/diffcsp/cspnet.py:79-91,61) as ONE launch per layer boundary (csrc/node_chain.hip):
 * 1 (default) = on for hidden_dim 128 / 256 / 512 with LayerNorm, 0 = the seven-launch form.  Returns the previous setting. */
int mi_debug_set_node_fused(int on);
/* The same launch in the TRAINING forward, which then also writes what the backward pass reads of it (the aggregated messages and
 * LayerNorm(h) into the tape's cat rows, the two node-MLP pre-activations, the LayerNorm statistics): 1 (default) = on wherever the
 * inference chain is, 0 = layernorm + PQ product + finalize_agg + two node-MLP products (seven launches per layer).  Returns the
 * previous setting. */
int mi_debug_set_node_train(int on);
/* The node-level BACKWARD chain of the fine-tune step (node_bwd.hip): 1 (default) = one launch per layer boundary for batches of at least
 * `min_blocks` 32-row blocks (<= 0: keep the current threshold), 0 = the seven-launch form (three fp32-operand products, three streaming
 * passes and the LayerNorm gradient per layer).  Both are fp32-class; the gradient tests run both.  Returns the previous setting. */
int mi_debug_set_node_bwd(int on, int min_blocks);
/* knn edge style inside mi_sampler_run: 1 (default) = every evaluation rebuilds the list without a host round trip (consumers sized by capacity, edge
 * count read on the device; the FIRST build of a batch handle still synchronises once, so that the host knows the list's size when it picks kernel forms),
 * 0 = the synchronising build of rounds 1-5, 2 = no synchronisation at all (forms picked for the capacity; the tests use it to reach the deferred capacity
 * report).  Same results bit for bit; the tests run all three.  Returns the previous setting. */
int mi_debug_set_knn_nosync(int on);
/* The same chain as TWO launches for small and medium batches (at most 85 row blocks of 32 atoms per chain): phase A, then LayerNorm + the
 * three projection passes on three workgroups per row block -- the chain is bound by the weight planes a workgroup streams through its CU's
 * L2 port, and the passes are independent given LayerNorm(h') (models/diffcsp/cspnet.py:87-88,61).  1 (default) = on, 0 = one launch.
 * Returns the previous setting. */
int mi_debug_set_node_split(int on);
/* The chain with every product's COLUMNS split over workgroups (csrc/node_chain.hip, node_cols_kernel: 32 rows x 128 columns of one product
 * per 4-wave workgroup, the intermediates' slices exchanged through L2): 3 (default) = automatic -- one launch per stage (agg + node_mlp.0,
 * node_mlp.2 + residual, LayerNorm + projections) for chains of at most 36 row blocks, the row-block forms above for larger ones; 1 = one
 * launch per stage always; 2 = one launch per layer boundary with agent-scope flag hand-overs between the stages; 0 = off.  Bit-identical to
 * the row-block forms (models/diffcsp/cspnet.py:79-91,61).  Returns the previous setting, MI_EINVAL for other values. */
int mi_debug_set_node_cols(int mode);
/* Row-block chain launches of at least `min_blocks` 32-row blocks first warm the L2 with their weight operands (one dword per 128-byte line, shared
 * out among the workgroups of an XCD): between two node-chain launches the edge GEMMs stream > 100 MB through the L2s, so the weight rings otherwise
 * run at the miss latency.  Default 64 (one chain of 256 crystals: +4.3 %; chains of 64 crystals measured -1 % and stay off); 0 = never.  Returns the
 * previous setting. */
int mi_debug_set_node_touch(int min_blocks);
/* Atom count from which the fused output heads (coordinate + type read-outs, models/diffcsp/cspnet.py:291-294) run on 16 rows per workgroup instead of 4
 * (every workgroup streams the whole 213 KB weight block; same fp32 FMA chains per output, so the results do not depend on it).  Default: never (measured neutral / -2 %,
 * profiles/r5_heads_rows16_ab.log).  Returns the previous setting. */
int mi_debug_set_heads_rows16(int min_nodes);
/* What the sampler's predictor evaluation keeps from the corrector evaluation in front of it (models/diffcsp/diffusion.py:320-322: the corrector moves the
 * coordinates only, so everything that depends on the lattice, the types and the time alone is still valid): bit 0 = the lattice term G of every layer, bit 1 =
 * layer 0's LayerNorm + projections; bit 2 = the CORRECTOR evaluation computes the coordinate head alone (the Langevin corrector reads nothing else of it,
 * diffusion.py:310-322).  Default 7; 0 = everything evaluated every time (the A/B).  Results are identical either way.  Returns the previous mask. */
int mi_debug_set_eval_reuse(int mask);
/* TIMING ABLATIONS ONLY -- the results of a forward are garbage while a bit is set: 1 = skip the node chain's launches, 2 = the first edge GEMM,
 * 4 = the second (what a chain's serial path and the chip's occupancy cost each other: DESIGN 19.1), 8 = every node chain launched twice, 16 = the
 * pair-mode Fourier operand built once per batch handle and then left stale, 32 = an EMPTY launch in front of every edge GEMM (what a kernel boundary on a
 * chain's serial path costs; results unaffected).  Returns the previous mask. */
int mi_debug_set_skip(int mask);
/* The second linear of the edge MLP with the edge -> node reduction (models/diffcsp/cspnet.py:73-79) of an inference forward at
 * hidden_dim 512 on 128-row x 512-column register tiles with the segmented sum as an MFMA product (csrc/edge_stage.hip):
 * 1 (default) = on (inference forwards, next to the node-chain launch above; training forwards too, with the pre-activation kept
 * for the backward pass), 4 = inference forwards only, 0 = the 128 x 128-tile plane GEMM.  Returns the previous setting. */
int mi_debug_set_edge2_fused(int on);
/* Both edge products of a layer and the edge -> node sums in ONE launch (csrc/edge_fused.hip: a workgroup owns 64 atom pairs, M1 stays
 * in LDS; inference forwards, fc pair mode, hidden_dim 512, next to the node-chain launch): 1 = on, 0 (default) = the pair GEMM + the
 * second edge GEMM.  M1 is bit-identical; the partial sums are formed over other row groups.  A recorded experiment (parity green, 27 %
 * slower end to end: DESIGN 16.4) that exists in -DMI_ABLATION_KERNELS builds only; the default library ignores 1.  Returns the previous
 * setting. */
int mi_debug_set_edge_fused(int on);
/* Phase clock of that launch: dev_buffer = [workgroups][16] uint64 s_memtime stamps (0 entry, 1 + 3c / 2 + 3c / 3 + 3c: first product /
 * pair epilogue / second product of column chunk c, 13 exit), NULL = off. */
int mi_debug_edge_fused_clock(void* dev_buffer);
/* The pair-mode first edge GEMM (Fourier block over unordered atom pairs, models/diffcsp/cspnet.py:59-74) on the same form -- 128 x 128
 * tiles per four-wave workgroup, the Fourier operand by LDS-DMA, the weights in fragment order straight from L2: 9 (default) = that form
 * for hidden_dim multiples of 128 and launches beyond the plane GEMM's small-launch forms (with its k-loop under manual control it is
 * 3-6 % ahead end to end: DESIGN 18.4e), 1 = that form whatever the size, 0 = the plane GEMM always; 2 = 128 x 256 tiles, one
 * workgroup per CU with 512 registers per lane (half the LDS reads per MFMA; measured 11-13 % slower end to end; exists only in a
 * -DMI_ABLATION_KERNELS build, otherwise 2 runs form 1); 3 = the 128 x 128 tile as 2 x 2 waves of 64 pairs x 64 columns (half the LDS reads,
 * twice the weight fetches; 12 % slower; ablation build only).  Same epilogue: bit-identical M1.  Returns the previous setting. */
int mi_debug_set_edge1_fused(int on);
/* Phase clock of ONE launch of the register-tile GEMM (csrc/edge_stage.hip gemm_rt_kernel: the dense layers of the MatterGen-shaped network,
 * the dM1 data gradient of `loss.backward()`, pipeline/mat_invent.py:164): dev_buffer = [workgroups][8] uint64 -- s_memtime at entry / first
 * k-tile in LDS / end of the main loop / exit, then s_memrealtime (100 MHz) at entry and exit.  ext = 0 / 1: the next launch with the plain /
 * the extended epilogue after `skip` such launches (the clock then switches itself off); ext = -1: every launch (the last one stays) until
 * called with a null buffer. */
int mi_debug_rt_clock(void* dev_buffer, int ext, int skip);
/* The lean epilogue of that kernel for the launches that only write a plane set (the dense layers of an inference forward of the
 * MatterGen-shaped network, models/mattergen/pl_module.py:73: activation, plane-set residuals, multiplicand, exact max |y|): transposed
 * accumulator tiles + v_permlane32_swap instead of the LDS patch, every scale folded into two constants.  1 (default) = on, 0 = the general
 * row epilogue for every launch.  Returns the previous setting. */
int mi_debug_set_rt_lean(int on);
/* Phase clock of that kernel (measurement only): device buffer of [row tiles][8] 64-bit s_memtime stamps (start, first operand chunk
 * landed, main loop done, epilogue done); nullptr = off.  `on` = 2 above selects the variant with a two-deep weight ring and
 * double-buffered activation fragments (ablation). */
int mi_debug_edge2_clock(void* dev_buffer);
int mi_debug_edge1_clock(void* dev_buffer);   /* the same for the first edge GEMM: [row tiles x column quarters][8] stamps */
/* Phase clock of that launch (measurement only): a device buffer of [workgroups][16] 64-bit words that every workgroup fills with
 * s_memtime stamps at its phase boundaries; nullptr (default) = off. */
int mi_debug_node_chain_clock(void* dev_buffer);
/* Arithmetic of the large dense layers of the forward pass: 0 (default) = three bf16 planes split on the fly (six MFMA terms),
 * 1 = TWO fp16 planes with power-of-two scales from the operands' exact absmax (three terms; saturation impossible by
 * construction; measured no faster: the fp32-operand kernel is bound by its operand path).  Both are fp32-class; the tests run both. */
int mi_debug_set_mg_f16(int on);
/* Edge-level dense layers (from 4096 edges up) on the pre-split plane-set kernel: 1 (default) / 0 = the fp32-operand kernel everywhere;
 * 3 = the forward on the plane-set kernel, the backward's data-gradient products (dX += dZ W, dZ written as a plane set by the
 * activation-gradient pass) on the fp32-operand kernel (ablation). */
int mi_debug_set_mg_planes(int on);
/* Inference forwards in plane mode keep each edge-level tensor in ONE format (the plane set where a dense layer reads it, fp32 rows
 * otherwise; other consumers reconstruct the exact value from the planes), fold the skip-connection merges into the last layer of the
 * residual stack they close and the radial weighting into the edge -> atom sum: 1 (default) / 0 = both formats and separate passes,
 * as the training forward always does.  Same results to fp32 rounding; the tests run both. */
int mi_debug_set_mg_lean(int on);
/* The sampler's forwards without a host round trip per evaluation (mi_gemnet_forward, bit 2): 1 (default) / 0 = the synchronising
 * form.  Returns the previous setting.  Same results bit for bit; the tests run both. */
int mi_debug_set_mg_nosync(int on);
/* In-degree capacity of the periodic graph (default and maximum 128 = the triplet kernels' LDS capacity; <= 0 restores it): a crystal
 * holding an atom with more in-edges is taken out of the graph (flag 4).  Lowering it shrinks the triplet kernels' LDS image of the
 * sampler's forwards (more workgroups per CU) at the price of flagging denser crystals; the tests use it to exercise the flag path.
 * Returns the previous value. */
int mi_debug_set_mg_deg_cap(int cap);
/* Matrix-pipe work ISSUED by every product this process has launched since the last reset (measurement only; host-side counters, so
 * nothing is synchronised): *flops16 = sum over the products on the 16-bit pipe of 2 M N K x the MFMA terms each fp32 product is issued
 * as (3: two pre-split fp16 planes; 6: three bf16 planes split on the fly; 1: the TF32-class build), *flops32 = 2 M N K of the
 * f32-input MFMA products.  bench.py divides them by the elapsed time of a timed region: the whole step's matrix-pipe rate. */
int mi_debug_mfma_flops(double* flops16, double* flops32, int reset);
#ifdef __cplusplus
}
#endif
#endif /* MATINVENT_HIP_DEBUG_H */
