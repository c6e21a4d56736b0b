/*
 * melissa_hip.h - C ABI of libmelissa_hip.so: the MI355X (gfx950) hot path of Melissa.
 *
 * The drop-in boundary.  Every pointer marked "device" is a DEVICE pointer
 * borrowed from the caller (e.g. torch.Tensor.data_ptr() of a ROCm tensor); the library never
 * allocates, frees or retains caller memory, keeps no global mutable state except a thread-local
 * error string, launches everything on the caller's HIP stream and never synchronises.  `stream`
 * is a hipStream_t passed as void* so this header needs no HIP include.
 *
 * Every entry point returns mel_status (0 = ok, negative = error, see mel_last_error()).
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   mel_ldgn_forward   -> LDGNNetwork.forward        graph_env/env/utils/networks/l_dgn.py:92-151
 *   mel_hldgn_forward  -> HLDGNNetwork.forward       graph_env/env/utils/networks/hl_dgn.py:82-119
 *   mel_dgnr_forward   -> DGNRNetwork.forward        graph_env/env/utils/networks/dgn_r.py:82-129
 *                         (both include build_pyg_batch_time, networks/common.py:6-64, and the
 *                          [3P] radius_graph / GATv2Conv / global_*_pool / tianshou MLP they call)
 *   mel_select_action  -> [3P] tianshou DQNPolicy.forward mask + argmax and exploration_noise,
 *                         called at policies/multi_agent_managers/shared_policy.py:81-91,154
 *   mel_env_reset      -> GraphEnv.reset + World.reset      graph.py:222-248, core.py:343-437
 *   mel_env_step       -> GraphEnv.step (+ World.step ...)  graph.py:303-389,402-463, core.py:225-341,
 *                         selector.py:25-48
 *   mel_env_observe    -> GraphEnv.observe / last() + [3P] PettingZooEnv.step packing
 *                         graph.py:181-216, SURVEY.md Appendix A.6
 */
#ifndef MELISSA_HIP_H
#define MELISSA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t mel_status;
#define MEL_OK                 0
#define MEL_ERR_INVALID_ARG   -1   /* null pointer, bad size, N outside [1, 128] ...             */
#define MEL_ERR_SHAPE         -2   /* obs width != N*(in_dim+3)+1 (networks/common.py:24-29)      */
#define MEL_ERR_UNSUPPORTED   -3   /* layer sizes the kernels are not built for                  */
#define MEL_ERR_WORKSPACE     -4   /* ws_bytes < mel_workspace_bytes(...)                        */
#define MEL_ERR_LAUNCH        -5   /* hipGetLastError() != hipSuccess after a launch             */

/* Node sets.  A set of nodes of an N-node graph is MEL_SET_WORDS(N) consecutive uint64 words, word k holding nodes
 * 64 k .. 64 k + 63 (bit i of word k = node 64 k + i).  For N <= 64 (the sizes BASELINE quotes, 20 and 50) that is ONE
 * 64-bit mask and every layout below is exactly "one uint64 per set"; for 64 < N <= 128 (the reference CLI's third size,
 * --n-agents 100, common.py:49) every array documented as uint64 [..] of node sets gains a trailing [2]: one_hop is
 * [B, N, 2], node_sets [B, 8, 2], live [B, 2], adjacency taps [bs, N, 2] and so on.  One wavefront still steps one env;
 * a lane then holds the two nodes lane and lane + 64. */
#define MEL_MAX_NODES        128
#define MEL_SET_WORDS(n)     (((n) + 63) / 64)
#define MEL_NODE_COLS          8   /* x, y, 5 features, dm flag (graph.py:80)                    */
#define MEL_MAX_HEAD_LAYERS    6

#define MEL_MODEL_LDGN         0
#define MEL_MODEL_HLDGN        1
#define MEL_MODEL_DGNR         2   /* dgn_r.py: the L-DGN skeleton with TransformerConv layers       */

#define MEL_CONV_GATV2         0
#define MEL_CONV_TRANSFORMER   1

#define MEL_AGG_MAX            0   /* hl_dgn.py:56-60 */
#define MEL_AGG_MEAN           1
#define MEL_AGG_ADD            2

/* One torch.nn.Linear: weight [out_dim, in_dim] row-major fp32, bias [out_dim] fp32 (device). */
typedef struct mel_linear {
    const float* weight;
    const float* bias;
    int32_t in_dim;
    int32_t out_dim;
} mel_linear;

/* One attention conv layer.
 * kind MEL_CONV_GATV2 - [3P] PyG GATv2Conv: lin_l (sources) / lin_r (targets)
 *   [heads*C, in], att [1, heads, C], bias [heads*C]; negative_slope 0.2, self-loops added, eps 1e-16.
 * kind MEL_CONV_TRANSFORMER - [3P] PyG TransformerConv(root_weight=False) (A.2): lin_l = lin_key and
 *   lin_v = lin_value (sources), lin_r = lin_query (targets); score q.k/sqrt(C), NO self-loops, no output
 *   bias (att, bias NULL); lin_skip is a parameter of the module but never used. */
typedef struct mel_gatv2 {
    mel_linear lin_l;
    mel_linear lin_r;
    const float* att;
    const float* bias;
    int32_t heads;
    int32_t channels;      /* C, per head */
    mel_linear lin_v;
    int32_t kind;
    int32_t reserved;
} mel_gatv2;

/* [3P] tianshou MLP: Linear, ReLU, ..., Linear (no activation after the last layer). */
typedef struct mel_mlp {
    mel_linear layer[MEL_MAX_HEAD_LAYERS];
    int32_t n_layers;
} mel_mlp;

/* Borrowed views of the nn.Parameter storages of LDGNNetwork / HLDGNNetwork (state_dict keys
 * encoder.model.{0,2}.*, conv1.*, conv2.* (L-DGN only), Q.model.*, V.model.*; l_dgn.py:49-86). */
typedef struct mel_weights {
    int32_t model;         /* MEL_MODEL_*                                            */
    int32_t in_dim;        /* node features fed to the encoder (5)                   */
    int32_t n_actions;     /* Q head output (2)                                      */
    int32_t dueling;       /* 1: Q/V dueling heads; 0: q_head is the single out_linear (l_dgn.py:88) */
    mel_mlp   encoder;     /* 2 layers: in_dim -> hidden -> hidden                   */
    mel_gatv2 conv1;
    mel_gatv2 conv2;       /* unused for HL-DGN                                      */
    mel_mlp   q_head;      /* latent -> ... -> n_actions                             */
    mel_mlp   v_head;      /* latent -> ... -> 1                                     */
    int32_t precision;     /* MEL_PREC_F32 (reference arithmetic, logits <= 1e-4), MEL_PREC_BF16, MEL_PREC_F32_SPLIT or MEL_PREC_F32_AUTO */
    int32_t flags;         /* MEL_FWD_PLAN_READY: the plan masks of this call were written by mel_env_round (plan_* sink);
                            * MEL_FWD_INTEGER_FEATURES: see below */
    const void* prepared;  /* optional (MEL_PREC_BF16 / MEL_PREC_F32_SPLIT): device buffer filled by mel_prepare_weights for THESE
                            * weights - the forward then reads the converted projection weights from it and launches no
                            * conversion.  NULL: every call converts the fp32 parameters into its workspace (stateless). */
    const void* tables;    /* optional: device buffer filled by mel_prepare_feature_tables for THESE weights and tables_nodes
                            * nodes per graph - a forward that takes the node-feature table (MEL_FWD_INTEGER_FEATURES) then reads
                            * the encoder / conv1-projection rows of every feature tuple from it instead of evaluating them
                            * (they depend on the weights only).  NULL (the default everywhere, bench.py's headline included):
                            * the table rows are evaluated by every call. */
    int32_t tables_nodes;
    int32_t reserved;
} mel_weights;
#define MEL_FWD_PLAN_READY 1
/* MEL_FWD_INTEGER_FEATURES: the caller guarantees that the five node features of every observation row are the integers
 * GraphEnv writes for policy agents (graph.py:261-269: degree in [0, N), messages transmitted in [0, 4], last action /
 * interested / has message in {0, 1}) - true for everything mel_env_* produces when no agent is scripted.  The encoder and the conv1 projections are then functions of
 * one of N * 40 feature TUPLES: when the receptive-field row lists are at least twice that long, the forward evaluates them
 * once per tuple (table rows, on every call - nothing is kept between calls) and the conv1 attention gathers rows by tuple
 * id.  Same arithmetic per row, bit-identical logits.  A feature outside the ranges is clamped into the table and flagged
 * (mel_forward_tap kind 3).  Without the flag the general row-list path runs (arbitrary float features). */
#define MEL_FWD_INTEGER_FEATURES 2

/* Feature precision of the L-DGN / DGN-R forward (BASELINE config "bf16 feature path").  MEL_PREC_BF16: the
 * node-feature rows between layers (encoder output, lin_l / lin_r projections, conv outputs, head input and
 * hidden layers) and the weight matrices of the dense projections are bf16 (the fp32 nn.Parameters are
 * converted into the workspace by every call), contractions run on the bf16 MFMA with fp32 accumulation;
 * attention scores / softmax / aggregation, biases, the encoder's first layer, the last head layer and the
 * logits stay fp32.  Not the reference's arithmetic: expect ~1e-2 absolute on logits (tests state the bound). */
#define MEL_PREC_F32  0
#define MEL_PREC_BF16 1
/* MEL_PREC_F32_SPLIT: fp32 features and fp32-accurate results, with the dense projections evaluated on the bf16 matrix
 * cores by operand splitting (x = hi + mid + lo in bf16, exactly; six of the nine partial products, each exact in
 * fp32, accumulated in fp32).  As close to the exact dot product as a native fp32 GEMM (csrc/gemm_split.hpp); the
 * reference's parity bar (logits <= 1e-4) holds with the same margin as MEL_PREC_F32. */
#define MEL_PREC_F32_SPLIT 2
/* MEL_PREC_F32_AUTO: fp32 features and fp32-accurate results like the two above, with the arithmetic chosen PER LAUNCH: the
 * large projections of a forward (conv2's lin_l + lin_r and the heads' first layer once their row lists fill the chip with
 * 128 x 128 tiles: from a few thousand rows on) take the split kernels of MEL_PREC_F32_SPLIT, everything else - small batches
 * entirely - the exact-fp32 matrix instruction of MEL_PREC_F32.  The choice depends on the expected row counts only, never on
 * the data.  Weights: the fp32 parameters plus their bf16 planes (prepared or converted per call exactly as for
 * MEL_PREC_F32_SPLIT, same buffer size).  L-DGN 50-node, 1024 envs: 0.211 -> 0.18 ms per round step. */
#define MEL_PREC_F32_AUTO 3

/* Convert the projection weights once per weight VERSION instead of once per call (bf16 feature path: bf16 copies; split
 * path: three bf16 planes per weight).  The caller owns the buffer (mel_prepared_weights_bytes; 0 for MEL_PREC_F32), calls
 * mel_prepare_weights after every change of the parameters (optimizer step, load_state_dict) and points
 * mel_weights.prepared at it; the library keeps no state. */
size_t mel_prepared_weights_bytes(const mel_weights* w);
mel_status mel_prepare_weights(const mel_weights* w, void* prepared, size_t bytes, void* stream);

/* The node-feature table of MEL_FWD_INTEGER_FEATURES (encoder row and conv1 projections of each of the n_nodes * 40 feature
 * tuples) evaluated ONCE per weight version into a caller-owned buffer, like the prepared weights above: same kernels, same
 * rows, same bits as the per-call evaluation.  Opt-in (mel_weights.tables); precision follows mel_weights.precision and,
 * for bf16 / split, needs mel_weights.prepared. */
size_t mel_feature_tables_bytes(const mel_weights* w, int32_t n_nodes);
mel_status mel_prepare_feature_tables(const mel_weights* w, int32_t n_nodes, void* tables, size_t bytes, void* stream);

/* Bytes of scratch the forward needs for `bs` observation rows of `n_nodes`-node graphs. */
size_t mel_workspace_bytes(const mel_weights* w, int64_t bs, int32_t n_nodes);

/* obs: device fp32 [bs, obs_width] row-major, obs_width must equal n_nodes*(in_dim+3)+1;
 * logits: device fp32 [bs, n_actions].  L-DGN evaluates exactly the receptive field of the
 * controlling agent (rows of conv1/conv2 that can reach logits), which yields the same logits as the
 * full-graph evaluation of l_dgn.py:117-135. */
mel_status mel_ldgn_forward(const mel_weights* w, const float* obs, int64_t bs, int32_t n_nodes,
                            int32_t obs_width, float* logits, void* workspace, size_t ws_bytes,
                            void* stream);

/* L-DGN for a SET of controlling agents per env (the round-batched loop): within one env round every
 * active agent observes the same obs_matrix (graph.py:186-188 - rows differ only in the index column), so
 * all of a round's agents are evaluated together and the encoder / conv1 work on the union of their
 * receptive fields is shared.  obs: device fp32 [bs, obs_stride], row b = obs_matrix of env b (the index
 * column is not read; obs_stride >= n_nodes*(in_dim+3)); agent_mask: device uint64 [bs], bit i = agent i
 * is evaluated.  logits: device fp32 [rows_cap, n_actions], rows ordered by env then agent id; only the
 * first sum(popcount(agent_mask)) rows are written (rows_cap >= that sum, <= bs*n_nodes).
 * row_offsets (optional, device int32 [bs+1]): first logits row of each env, total at [bs].
 * select (optional): also write each row's action (see mel_select).
 * Each row equals mel_ldgn_forward on (obs_matrix, agent) within fp32 rounding. */
/* Optional fused action selection for mel_ldgn_forward_agents: argmax of each logits row (+ eps-greedy with
 * the same counter-based stream as mel_select_action_rows) written by the kernel that produces the logits. */
typedef struct mel_select {
    int32_t*        act;       /* device int32 [rows_cap]                                          */
    float           eps;
    uint32_t        seed;
    const uint32_t* step_dev;  /* optional device counter added to the stream position             */
    /* per-env logits (HL-DGN, mel_hldgn_forward_envs_select): every agent i in live[b] takes its action from row b,
     * act is the dense [bs, n_nodes] layout of mel_select_action_envs and the stream is keyed on
     * b * 64 * MEL_SET_WORDS(n_nodes) + i                                                                           */
    const uint64_t* live;      /* node sets [bs]; NULL: one action per logits row                   */
    int32_t         n_nodes;
    int32_t         reserved;
} mel_select;

size_t mel_workspace_bytes_agents(const mel_weights* w, int64_t bs, int32_t n_nodes, int64_t rows_cap);
/* Where, inside `workspace`, a forward of these dimensions keeps its plan masks (rows_cap = 0: the single-agent / HL-DGN
 * layout of mel_workspace_bytes): out5 = {adj, live, u1, u2, cnt} (u1 / u2 / cnt NULL for HL-DGN) - the values for
 * mel_env_batch.plan_*. */
mel_status mel_plan_pointers(const mel_weights* w, int64_t bs, int32_t n_nodes, int64_t rows_cap, void* workspace,
                             void** out5);
mel_status mel_ldgn_forward_agents(const mel_weights* w, const float* obs, int64_t bs, int32_t n_nodes,
                                   int32_t obs_stride, const uint64_t* agent_mask, int64_t rows_cap,
                                   float* logits, int32_t* row_offsets, const mel_select* select,
                                   void* workspace, size_t ws_bytes, void* stream);

/* DGNRNetwork.forward (graph_env/env/utils/networks/dgn_r.py:82-129): same contracts as the two L-DGN
 * entry points above, for weights with model == MEL_MODEL_DGNR (TransformerConv layers). */
mel_status mel_dgnr_forward(const mel_weights* w, const float* obs, int64_t bs, int32_t n_nodes,
                            int32_t obs_width, float* logits, void* workspace, size_t ws_bytes,
                            void* stream);
mel_status mel_dgnr_forward_agents(const mel_weights* w, const float* obs, int64_t bs, int32_t n_nodes,
                                   int32_t obs_stride, const uint64_t* agent_mask, int64_t rows_cap,
                                   float* logits, int32_t* row_offsets, const mel_select* select,
                                   void* workspace, size_t ws_bytes, void* stream);

mel_status mel_hldgn_forward(const mel_weights* w, int32_t aggregator, const float* obs, int64_t bs,
                             int32_t n_nodes, int32_t obs_width, float* logits, void* workspace,
                             size_t ws_bytes, void* stream);

/* The dense projection used by every layer above, exposed for tests and tuning:
 *   Y[m, n] = act(sum_k A[m, k] * W[n, k] + bias[n]),  A [M, lda], W [N, K] (nn.Linear layout), Y [M, ldy],
 * all device fp32; K % 32 == 0, N % 64 == 0.  tile: 0 = automatic, 1 = 64x64, 2 = 128x128 workgroup tile, 11 = the
 * persistent 64x64 kernel the ragged launches of the round step use, 31 = the specialised-wavefront (loader / MFMA waves)
 * kernel the heads' long-K first layer uses. */
mel_status mel_gemm_f32(const float* A, int32_t lda, const float* W, const float* bias, float* Y, int32_t ldy,
                        int64_t M, int32_t N, int32_t K, int32_t relu, int32_t tile, void* stream);

/* The learn path's backward products without transposed copies of their operands: Y[m, n] = sum_k A'[m, k] W'[n, k] with
 * W' = W^T read in place (W'[n, k] = W[k * ldw + n]) and, a_t != 0, A' = A^T likewise (A'[m, k] = A[k * lda + m]).
 * dX [rows, in] = dY [rows, out] . W [out, in]:  mel_gemm_f32_t(dY, out, 0, W, in, 1, dX, in, rows, in, out);
 * dW [out, in]  = dY^T . X [rows, in]:           mel_gemm_f32_t(dY, out, 1, X, in, 1, dW, in, out, in, rows).
 * N % 64 == 0, K % 32 == 0, lda % 4 == ldw % 4 == 0, M % 4 == 0 with a_t; w_t must be 1 (the plain form is mel_gemm_f32).  Same
 * tile, K order and MFMA sequence as mel_gemm_f32's 64 x 64 kernel on transposed copies: bit-identical sums. */
mel_status mel_gemm_f32_t(const float* A, int32_t lda, int32_t a_t, const float* W, int32_t ldw, int32_t w_t, float* Y, int32_t ldy,
                          int64_t M, int32_t N, int32_t K, void* stream);

/* The same product with the contraction cut into `ksplit` equal chunks that run as independent work items and are summed in
 * chunk order afterwards: for the learn path's weight gradients dW = dY^T X, a few dozen output tiles over a contraction as
 * long as the batch has rows (one tile per workgroup would leave most of the chip idle).  K / 32 must be a multiple of
 * ksplit with at least two 32-steps per chunk; parts: device scratch of ksplit * M * N floats. */
mel_status mel_gemm_f32_splitk(const float* A, int32_t lda, const float* W, const float* bias, float* Y, int32_t ldy,
                               int64_t M, int32_t N, int32_t K, int32_t relu, int32_t ksplit, float* parts,
                               int64_t parts_floats, void* stream);

/* The same product at MEL_PREC_F32_SPLIT (fp32 operands and results, six exact bf16 partial products per term on the bf16
 * matrix cores): W is split into bf16 planes in `scratch` first.  tile: 0 = the library's choice, 1 = 64 x 64, 2 = 128 x 128
 * (N % 128 == 0), 3 = 128 x 256 with BOTH operands converted to bf16 planes in scratch first (the kernel conv2's projections
 * run on in large forwards, where the conv1 attention stores its rows in that form; N % 256 == 0, N <= 1536, lda == K, ldy % 4 == 0,
 * no split-K; bit-identical to tile 2), + 100 = W's planes are already in scratch (an earlier call with the same W: benchmarks;
 * not with tile 3); ksplit > 1: the 128 x 128 kernel's split-K (K / 16 a multiple of ksplit, >= 4 steps per chunk, ldy % 4 == 0).
 * K % 32 == 0, K >= 128, N % 64 == 0; scratch: device, >= round_up(6 N K, 256) + (ksplit > 1 ? 4 ksplit M N : 0)
 * + (tile 3 ? 6 K M : 0) bytes. */
mel_status mel_gemm_f32_split(const float* A, int32_t lda, const float* W, const float* bias, float* Y, int32_t ldy,
                              int64_t M, int32_t N, int32_t K, int32_t relu, int32_t tile, int32_t ksplit, void* scratch,
                              int64_t scratch_bytes, void* stream);

/* The same projection on the bf16 feature path: A [M, lda] and W [N, K] device bf16, bias fp32, fp32
 * accumulation, Y [M, ldy] bf16 (y_f32 = 0) or fp32 (y_f32 = 1); K % 64 == 0, N % 64 == 0, lda % 8 == 0.  tile: 0 = automatic,
 * 2 = 128 x 128, 3 = the 128 x 256 kernel of the large launches (specialised loader / MFMA wavefronts; N % 256 == 0, N <= 1536).
 * mel_convert_bf16: count (multiple of 8) fp32 values -> bf16, round to nearest even. */
mel_status mel_gemm_bf16(const void* A, int32_t lda, const void* W, const float* bias, void* Y, int32_t ldy,
                         int64_t M, int32_t N, int32_t K, int32_t relu, int32_t y_f32, int32_t tile, void* stream);
mel_status mel_convert_bf16(const float* src, void* dst, int64_t count, void* stream);

/* dst[c, r] = src[r, c] (device fp32; src [rows, ld_src >= cols], dst [cols, ld_dst >= rows]; columns r >= rows of dst are
 * left untouched, so a zero-filled dst with ld_dst = rows rounded up to 32 is a zero-padded transpose).  The learn path's
 * dense backward (policies/dgn.py:49-67, [3P] DQNPolicy.learn) runs on mel_gemm_f32, which wants both operands contiguous
 * along the contraction index:  dX = dY W = gemm(dY, W^T),  dW = dY^T X = gemm(dY^T, X^T). */
mel_status mel_transpose_f32(const float* src, int32_t ld_src, int64_t rows, int32_t cols, float* dst, int32_t ld_dst,
                             void* stream);

/* ---- learn path: attention and pool with hand-written backward ---------------------------
 * Wrapped by torch.autograd.Function in melissa_amd/networks/autograd_ops.py; all buffers device fp32, row-major,
 * rows = bs * n_nodes (row b*n + i = node i of graph b), HC = heads * channels in {128, 256, 512, 1024}.
 *
 * mel_radius_graph: adj[b*n + i] = sources of target i ([3P] radius_graph on the fp32 obs positions,
 *   networks/common.py:47-48: strict <, first 33 hits, self excluded) - the mask the forward kernels build.
 * mel_gat_forward: out = relu(conv(x) + bias) given the projections (kind MEL_CONV_GATV2: xl = lin_l(x) sources,
 *   xr = lin_r(x) targets, att [HC], self-loops added, l_dgn.py:125-126; MEL_CONV_TRANSFORMER: xl = keys,
 *   xv = values, xr = queries, no self-loops, bias null, dgn_r.py:103-104).
 * mel_gat_backward: grad_out -> dxl (dxv) dxr datt dbias, all WRITTEN, all deterministic sums (the source rows' gradients are
 *   gathered target by target; d att / d bias are summed per workgroup and then over the workgroups in a fixed order: no
 *   atomics).  stats: device scratch, bs * n_nodes * heads * 4 + MEL_GAT_PARTIAL_GROUPS * 2 * HC floats.
 * mel_pool_forward / backward: hl_dgn.py:105-108, pooled[b] = max / mean / add over nodes of x * dm;
 *   arg [bs, HC] int32 = node of the first maximum (max only). */
#define MEL_GAT_PARTIAL_GROUPS 256
mel_status mel_radius_graph(const float* obs, int64_t bs, int32_t n_nodes, int32_t obs_stride, int32_t in_dim,
                            uint64_t* adj, void* stream);
mel_status mel_gat_forward(const float* xl, const float* xv, const float* xr, const float* att, const float* bias,
                           const uint64_t* adj, int64_t bs, int32_t n_nodes, int32_t heads, int32_t channels,
                           int32_t kind, float* out, void* stream);
mel_status mel_gat_backward(const float* xl, const float* xv, const float* xr, const float* att, const uint64_t* adj,
                            const float* out, const float* grad_out, int64_t bs, int32_t n_nodes, int32_t heads,
                            int32_t channels, int32_t kind, float* dxl, float* dxv, float* dxr, float* datt,
                            float* dbias, float* stats, void* stream);
mel_status mel_pool_forward(const float* x, const float* dm, int64_t bs, int32_t n_nodes, int32_t hc,
                            int32_t aggregator, float* pooled, int32_t* arg, void* stream);
mel_status mel_pool_backward(const float* grad_pooled, const float* dm, const int32_t* arg, int64_t bs,
                             int32_t n_nodes, int32_t hc, int32_t aggregator, float* dx, void* stream);

/* HL-DGN for the round-batched loop: logits depend only on the env (hl_dgn.py:108 pools over the graph and
 * ignores the controlling index), so one row per env serves all of a round's agents.  obs: device fp32
 * [bs, obs_stride], row b = obs_matrix of env b, obs_stride >= n_nodes*(in_dim+3), index column not read. */
/* mel_hldgn_forward_envs with the per-(env, agent) selection of mel_select_action_envs fused into the launch that
 * produces the logits (select->live / n_nodes / act as documented at mel_select). */
mel_status mel_hldgn_forward_envs_select(const mel_weights* w, int32_t aggregator, const float* obs, int64_t bs,
                                         int32_t n_nodes, int32_t obs_stride, float* logits, const mel_select* select,
                                         void* workspace, size_t ws_bytes, void* stream);
mel_status mel_hldgn_forward_envs(const mel_weights* w, int32_t aggregator, const float* obs, int64_t bs,
                                  int32_t n_nodes, int32_t obs_stride, float* logits, void* workspace,
                                  size_t ws_bytes, void* stream);

/* Debug/parity taps: copies of intermediates after a forward with the same workspace.
 * kind: 0 = adjacency masks uint64 [bs, n_nodes] (bit j of row i set <=> edge j -> i, radius rule),
 *       1 = head input [bs, latent], fp32 (bf16 when w->precision is MEL_PREC_BF16) (L-DGN: x_1|x_2|x_3,
 *           l_dgn.py:139; HL-DGN: pooled, hl_dgn.py:108),
 *       2 = int32 [3]: rows the L-DGN GEMMs processed (sum |U1|, sum |U2|, agent rows),
 *       3 = int32 [1 + bs]: [0] rows of the node-feature table the last forward used (0 = row-list path),
 *           [1 + b] = 1 if env b had a node feature outside the integer ranges of MEL_FWD_INTEGER_FEATURES.
 * rows_cap: 0 for a workspace used by mel_ldgn_forward / mel_hldgn_forward, else the rows_cap given to
 * mel_ldgn_forward_agents.  `out` is a device pointer with room for the requested tensor. */
mel_status mel_forward_tap(const mel_weights* w, int32_t kind, int64_t bs, int32_t n_nodes, int64_t rows_cap,
                           const void* workspace, void* out, void* stream);

/* [3P] DQNPolicy.forward + exploration_noise:
 *   q = logits + (1 - mask) * (min(logits) - max(logits) - 1)   (batch-wide min / max)
 *   act = argmax(q);  if rand_u[b] < eps: act = argmax(rand_q[b, :] + mask)
 * mask: device uint8 [bs, n_actions] or NULL; rand_u [bs], rand_q [bs, n_actions] device fp32 or NULL
 * (eps-greedy off).  act: device int32 [bs].  scratch: device, >= 8 bytes. */
mel_status mel_select_action(const float* logits, const uint8_t* mask, int64_t bs, int32_t n_actions,
                             float eps, const float* rand_u, const float* rand_q, int32_t* act,
                             void* scratch, void* stream);

/* Row-wise argmax + eps-greedy for the round-batched loop: the row count lives on the device
 * (rows_dev, may be NULL = rows_cap) and the exploration stream is a counter-based hash of
 * (seed, step + *step_dev, row) instead of host numpy draws (step_dev: optional device counter, e.g. the
 * one mel_env_round advances, so the stream moves on when the launches are replayed from a HIP graph).  logit_row (optional, device int32 [rows_cap]) maps an
 * action row to its logits row (HL-DGN: all agents of an env share the env's logits). */
mel_status mel_select_action_rows(const float* logits, const int32_t* logit_row, int64_t rows_cap,
                                  const int32_t* rows_dev, int32_t n_actions, float eps, uint32_t seed,
                                  uint32_t step, const uint32_t* step_dev, int32_t* act, void* stream);

/* Per-(env, agent) action selection from per-env logits (HL-DGN in the round loop): for every agent i in
 * live[b], act[b, i] = argmax(logits[b]) or, with probability eps, a uniformly random action (same
 * counter-based stream as mel_select_action_rows, keyed on (seed, step + *step_dev, b * 64 * MEL_SET_WORDS(n_nodes) + i)).
 * act: device int32 [bs, n_nodes] (dense layout accepted by mel_env_round when row_offsets is NULL). */
mel_status mel_select_action_envs(const float* logits, const uint64_t* live, int64_t bs, int32_t n_nodes,
                                  int32_t n_actions, float eps, uint32_t seed, const uint32_t* step_dev,
                                  int32_t* act, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Environment half.  State of B independent envs lives in caller-owned device memory laid out as
 * struct-of-arrays (one wavefront steps one env, lane = node - two nodes per lane beyond 64 -, node sets as defined at
 * MEL_SET_WORDS);
 * mel_env_state_bytes() gives the size, mel_env_bind() carves the pointer table.
 * ------------------------------------------------------------------------------------------------ */
#define MEL_ENV_LOGGER_STATS  10   /* graph.py:167-177, in dict order                            */

/* node_sets[b][k] */
#define MEL_SET_HAS_MESSAGE    0   /* State.has_message                        core.py:18        */
#define MEL_SET_ORIGIN         1   /* State.message_origin                     core.py:16        */
#define MEL_SET_INTERESTED     2   /* Agent.is_interested                      core.py:71        */
#define MEL_SET_SCRIPTED       3   /* Agent.is_scripted                        core.py:69,404    */
#define MEL_SET_TRUNCATED      4   /* Agent.truncated                          graph.py:333      */
#define MEL_SET_ALIVE          5   /* keys of GraphEnv.terminations            graph.py:282-286  */
#define MEL_SET_TERMINATED     6   /* terminations[a] == True                  graph.py:334      */
#define MEL_SET_AGENTS         7   /* GraphEnv.agents (always id-ordered)      graph.py:336-341  */
/* sel_sets[b][k] */
#define MEL_SEL_ACTIVE         0   /* CustomSelector active                    selector.py:43-44 */
#define MEL_SEL_SELECTED       1   /* CustomSelector selected_round            selector.py:30    */
#define MEL_SEL_INFO_VALID     2   /* infos[a] holds logger_stats              graph.py:358      */
#define MEL_SEL_TAKEN_ACTION   3   /* State.has_taken_action                   core.py:272       */
/* scalars[b][k] */
#define MEL_S_ORIGIN           0   /* World.origin_agent                                         */
#define MEL_S_SELECTION        1   /* GraphEnv.agent_selection (-1 = False)                      */
#define MEL_S_SKIP             2   /* _skip_agent_selection (-2 = None, -1 = False)              */
#define MEL_S_NUM_MOVES        3   /* GraphEnv.num_moves                                         */
#define MEL_S_WORLD_MSGS       4   /* World.messages_transmitted                                 */
#define MEL_S_NEW_ROUND        5   /* GraphEnv.is_new_round (-1 = None, 0, 1)                    */
#define MEL_S_EPISODE          6   /* pool episode currently loaded                              */
#define MEL_S_MOVE_CURSOR      7   /* movement draws consumed this episode                       */
#define MEL_S_DECISIONS        8   /* live (non-dead) agent decisions since bind                 */
#define MEL_S_DONE_COUNT       9   /* done observations this episode (multi_agent_collector.py:261-263) */
#define MEL_S_EPISODES_DONE   10   /* episodes finished (auto-reset mode)                        */
#define MEL_S_ERROR           11   /* MEL_ENV_ERR_* bits                                         */
#define MEL_S_EP_CURSOR       12   /* episodes started (indexes the auto-reset episode table)    */
#define MEL_ENV_SCALARS       16

typedef struct mel_env_batch {
    int32_t n_envs;
    int32_t n_nodes;
    int32_t dynamic_graph;     /* core.py:256                                                    */
    int32_t has_local_ratio;   /* graph.py:376,380                                               */
    double  local_ratio;
    int32_t heuristic;         /* MEL_HEURISTIC_* run by the scripted agents (core.py:226-234)    */
    int32_t is_testing;        /* evaluation mode: scripted agents stay in the active set (graph.py:244,340) */
    double*   pos;             /* [B, N, 2] float64 positions (graph node attr "pos")            */
    uint64_t* one_hop;         /* [B, N]    Agent.one_hop_neighbours_ids as bit masks            */
    uint64_t* two_hop;         /* [B, N]    Agent.two_hop_neighbours_ids                         */
    uint64_t* node_sets;       /* [B, 8]    MEL_SET_*                                            */
    uint64_t* sel_sets;        /* [B, 4]    MEL_SEL_*                                            */
    int32_t*  scalars;         /* [B, 16]   MEL_S_*                                              */
    int32_t*  agent_msgs;      /* [B, N]    Agent.messages_transmitted                           */
    int32_t*  received;        /* [B, N]    sum(State.received_from)                             */
    int32_t*  two_hop_cover;   /* [B, N]    Agent.two_hop_cover                                  */
    int8_t*   agent_action;    /* [B, N]    Agent.action (-1 = None)                             */
    int8_t*   current_actions; /* [B, N]    GraphEnv.current_actions (-1 = None)                 */
    int8_t*   steps_taken;     /* [B, N]    Agent.steps_taken                                    */
    int8_t*   sel_steps;       /* [B, N]    CustomSelector steps                                 */
    double*   rewards;         /* [B, N]    GraphEnv.rewards                                     */
    double*   pz_rewards;      /* [B, N]    [3P] PettingZooEnv.rewards (sticky)   */
    double*   episode_rewards; /* [B]       GraphEnv.episode_rewards_sum                         */
    float*    obs_matrix;      /* [B, N, 8] GraphEnv.obs_matrix                                  */
    double*   info_stats;      /* [B, N, 10] infos[agent]['logger_stats']                        */
    /* Optional episode log (caller-owned; log_capacity 0 = off): whenever an env's episode ends inside
     * mel_env_step / mel_env_round with on-device reset, one row is appended - the `logger_stats` of the final
     * observation (graph.py:166-178: what the reference's collectors gather into `episode_info`,
     * multi_agent_collector.py:283) plus env id, episode id and num_moves.  Rows beyond the capacity are dropped
     * (the cursor keeps counting). */
    int32_t   log_capacity;
    int32_t   log_reserved;
    int32_t*  log_cursor;      /* [1]  episodes logged so far (device, atomically advanced)      */
    double*   log_stats;       /* [capacity, 10]                                                 */
    int32_t*  log_meta;        /* [capacity, 3] env, pool episode, num_moves                     */
    /* Optional forward-plan sink (mel_env_round only; all NULL = off): device pointers into the forward workspace the
     * NEXT mel_*_forward_agents / mel_hldgn_forward_envs call will use (mel_plan_pointers).  The round kernel then also
     * writes what that forward's first launch would compute from the obs it just wrote and the next active sets -
     * the fp32 radius-graph adjacency (networks/common.py:48) and, for L-DGN / DGN-R, the agent / one-hop / two-hop
     * node sets and their sizes - and the forward is called with MEL_FWD_PLAN_READY in mel_weights.flags. */
    uint64_t* plan_adj;        /* [B*N] */
    uint64_t* plan_live;       /* [B]   (NULL for HL-DGN: adjacency only) */
    uint64_t* plan_u1;         /* [B]   */
    uint64_t* plan_u2;         /* [B]   */
    int32_t*  plan_cnt;        /* [3*B] */
} mel_env_batch;

/* An episode pool: what World.reset samples (core.py:372-394), pre-drawn on the host with the
 * reference's RNG calls and packed for the device (device pointers). */
typedef struct mel_episode_pool {
    int32_t n_episodes;
    int32_t n_nodes;
    int32_t max_moves;         /* movement offsets stored per episode                            */
    int32_t reserved;
    const double*   pos;       /* [E, N, 2] initial positions                                    */
    const uint64_t* one_hop;   /* [E, N]    initial adjacency (graph edges)                      */
    const uint64_t* interested;/* [E]                                                            */
    const int32_t*  origin;    /* [E]                                                            */
    const double*   moves;     /* [E, max_moves, 2, N] 0.06*U(-1,1): all x then all y (core.py:316-319) */
    const uint64_t* scripted;  /* [E] World.scripted_indices (core.py:197-221), or NULL = no scripted agents */
    /* Optional reset snapshots (mel_env_round only): HOST pointer to an env batch of >= n_episodes envs in which
     * mel_env_reset has been run once per episode (env e <- episode e, same dynamic_graph / local_ratio / heuristic /
     * is_testing settings).  GraphEnv.reset + World.reset are a pure function of the pre-drawn episode, so an env
     * whose episode ends loads that state instead of recomputing it (positions after the reset's move, one- / two-hop
     * masks, the source's first transmission); per-env counters and the sticky reward vector are carried over.
     * NULL: resets are computed in the round kernel. */
    const struct mel_env_batch* snapshot;
    /* Optional (device int32 [n_envs of the env batch]): how many episodes of env b the table holds so far.  An env that
     * starts episode number ep_cursor >= produced[b] sets MEL_ENV_ERR_EPISODE_UNDERRUN (the table would hand it an episode
     * it has already played: a static table that wraps, or a stream that was not refilled in time).  NULL: no check. */
    const int32_t* produced;
} mel_episode_pool;

/* bits of scalars[b][MEL_S_ERROR] */
#define MEL_ENV_ERR_MOVES_EXHAUSTED   1   /* an episode needed more than max_moves movement draws               */
#define MEL_ENV_ERR_NO_SELECTION      2   /* step with no agent selected (the reference would raise KeyError)   */
#define MEL_ENV_ERR_UNCOVERED_AGENT   4   /* mel_env_round: an agent acts that the action rows do not cover     */
#define MEL_ENV_ERR_EPISODE_UNDERRUN  8   /* see mel_episode_pool.produced                                      */

/* ------------------------------------------------------------------------------------------------
 * Continuous episode supply: World.reset's sampling (core.py:343-395) ON THE DEVICE.
 *
 * The reference draws, on every reset, an episode seed and a graph from the env's own generator
 * (np_random = Generator(PCG64), core.py:372,378) and - from RandomState(episode_seed), MT19937 - the movement seed,
 * the source, the interest density and the interested set (core.py:381-394); node movement then consumes
 * RandomState(movement_seed).uniform(-1, 1) (core.py:316-319).  mel_episode_refill performs exactly those draws, bit for
 * bit (numpy's PCG64 next_uint32 buffering + Lemire bounded integers; legacy MT19937 seeding, masked rejection
 * sampling, random_sample doubles and the Fisher-Yates shuffle of RandomState.choice(replace=False)), into a RING of
 * `ring` episode slots per env: episode j of env b lives in pool slot b*ring + j % ring.  The graph comes from a packed
 * device-resident dataset (mel_graph_pool: positions + adjacency masks of the G graphs that stand for
 * graph_topologies/training_N/*.pickle, core.py:165-175,450-452).  Each call also runs GraphEnv.reset + World.reset
 * for the new slots into the pool's snapshot batch (same code as mel_env_reset), so an ending episode loads its
 * successor's state.  Not covered (use a host-sampled table): is_testing's fixed seed list, scripted_agents_ratio > 0
 * (a Generator.choice without replacement), a fixed graph that moves (its positions carry over between episodes).
 * ------------------------------------------------------------------------------------------------ */
typedef struct mel_graph_pool {
    int32_t n_graphs;
    int32_t n_nodes;
    const double*   pos;       /* [G, N, 2] device float64 node positions ("pos" node attribute)              */
    const uint64_t* one_hop;   /* [G, N]    device adjacency masks (graph edges)                              */
} mel_graph_pool;

typedef struct mel_episode_stream {
    int32_t n_envs;            /* B                                                                            */
    int32_t ring;              /* K >= 3 episode slots per env                                                  */
    int32_t fixed_graph;       /* 1: GraphEnv(graph=...) - no graph draw (core.py:377), graphs->n_graphs == 1   */
    int32_t has_density;       /* 1: fixed_interest_density is used instead of ep_rng.uniform(0.1, 1.0) (:385) */
    double  fixed_interest_density;
    uint64_t* pcg;             /* [B, 4] device: PCG64 state lo, state hi, inc lo, inc hi of env b's np_random  */
    uint32_t* pcg_half;        /* [B, 2] device: has_uint32, uinteger (numpy buffers the upper half of a draw)  */
    int32_t*  produced;        /* [B]    device: episodes drawn into the ring so far (= mel_episode_pool.produced) */
    uint32_t* draw_seed;       /* [B, K] device scratch: episode_seed per slot (core.py:372)                    */
    int32_t*  draw_graph;      /* [B, K] device scratch: graph index per slot (core.py:378)                     */
    int32_t*  work;            /* [1 + 2*B*K] device scratch: work-item count, then (pool slot, unused) pairs   */
    int32_t*  new_count;       /* [B]    device scratch                                                         */
} mel_episode_stream;

/* For every env b draw episodes produced[b], produced[b]+1, ... while the slot they go to is free - i.e. up to episode
 * ep_cursor[b] + ring - 2 (the episode the env is playing keeps its slot) - and at most max_new of them; fill their
 * pool slots (pos / one_hop / origin / interested / scripted = 0 / moves), run their reset into pool->snapshot, then
 * publish produced[b].  `discard` episodes are drawn and dropped first (the samplings the reference performs while an
 * env is CONSTRUCTED, so that streams line up with a reference run).  `pool` must have n_episodes == B*ring, device
 * arrays the library may WRITE, a snapshot batch of B*ring envs, and produced == stream->produced.  ep_cursor is read
 * from env->scalars.  Launches on `stream`; the caller orders it against the env launches (a refill may overlap env
 * rounds on another stream as long as no env can reach an episode >= the produced[] value of the previous refill). */
/* A pacing gate for `stream`: whatever is enqueued behind it runs once the device counter `counter` (e.g. the round_counter
 * mel_env_round advances on another stream) has reached `target` (wrap-around compare), or after timeout_us.  It lets the
 * episode refill follow the main stream's progress without an event on the main stream (events between HIP-graph replays
 * cost the step several microseconds on this stack); one wavefront polls with agent-scope relaxed loads and s_sleep. */
mel_status mel_wait_counter(const uint32_t* counter, uint32_t target, uint32_t timeout_us, void* stream);

mel_status mel_episode_refill(const mel_episode_stream* st, const mel_graph_pool* graphs, const mel_episode_pool* pool,
                              const mel_env_batch* env, int32_t max_new, int32_t discard, void* stream);

/* Scripted agents (scripted_agents_ratio > 0): the deterministic heuristics of
 * graph_env/env/utils/heuristics/core.py.  The probabilistic ones (probabilistic_gossip / _relay) draw from the
 * process-global np.random and "mpr" does not return a HeuristicResult: not offered. */
#define MEL_HEURISTIC_NONE                  0
#define MEL_HEURISTIC_SIMPLE_BROADCAST      1   /* action = 0 if has_taken_action else 1     heuristics/core.py:13-18 */
#define MEL_HEURISTIC_BROADCAST_IF_INTERESTED 2 /* action = number_interested_neighbors > 0  :45-53 */
#define MEL_HEURISTIC_SILENT                3   /* action = 0                                :56-62 */

/* Outputs of last() + [3P] PettingZooEnv packing, one row per listed env (device; any may be NULL). */
typedef struct mel_env_obs {
    float*    obs;             /* [n, 8N+1] obs_matrix flattened + controlling index (graph.py:186-188) */
    int64_t   obs_stride;      /* floats between rows (>= 8N+1)                                  */
    int32_t*  agent_id;        /* [n]                                                            */
    uint8_t*  action_mask;     /* [n, 2]    graph.py:190-192                                     */
    double*   rew;             /* [n, N]    sticky reward vector                                 */
    uint8_t*  terminated;      /* [n]                                                            */
    int32_t*  flags;           /* [n, 4]    env_step, environment_step, explicit_reset, has_stats */
    uint64_t* active_nb;       /* [n]       info['active_one_hop_neighbors'] (graph.py:198-203)  */
    double*   stats;           /* [n, 10]   info['logger_stats']                                 */
} mel_env_obs;

size_t mel_env_state_bytes(int32_t n_envs, int32_t n_nodes);
/* Carves `state` (device, mel_env_state_bytes() bytes, 256-byte aligned) into the pointer table. */
mel_status mel_env_bind(mel_env_batch* env, int32_t n_envs, int32_t n_nodes, void* state);

/* GraphEnv.reset for the envs listed in env_ids (device int32 [n], or NULL = envs 0..n-1): loads pool
 * episode episode_ids[k] (device int32 [n]), runs the forced source transmission (core.py:246,437)
 * and, if `out` is not NULL, observes.  keep_graph != 0 keeps the env's current positions/edges
 * (the reference's fixed-graph mode mutates one graph across episodes, core.py:130,303-314). */
mel_status mel_env_reset(mel_env_batch* env, const mel_episode_pool* pool, const int32_t* env_ids,
                         const int32_t* episode_ids, int64_t n, int32_t keep_graph,
                         const mel_env_obs* out, void* stream);

/* One AEC step per listed env (GraphEnv.step; action ignored when the selected agent is dead), the
 * sticky reward copy of PettingZooEnv.step, then - if `out` is not NULL - last().
 * actions: device int32 [n].  If episode_table is not NULL (device int32 [B, table_stride]) an env
 * whose observation says the episode is over (terminated and (explicit_reset or N agents reported
 * done), multi_agent_collector.py:261-264) is reset in the same launch to pool episode
 * episode_table[b, ep_cursor % table_stride] and observed again. */
mel_status mel_env_step(mel_env_batch* env, const mel_episode_pool* pool, const int32_t* actions,
                        const int32_t* env_ids, int64_t n, const mel_env_obs* out,
                        const int32_t* episode_table, int32_t table_stride, void* stream);

/* Device-resident replay of the round-batched loop (the counterpart of the per-(env, agent) sub-buffers the
 * reference collector routes transitions into, multi_agent_collector.py:229-271).  A round of env b is ONE
 * record: the obs_matrix its agents decided on, who acted, their actions, the rewards GraphEnv.rewards holds
 * after the round's world step (graph.py:373-389), who was terminated by it (graph.py:330-334) and the
 * obs_matrix every one of them observes next - i.e. the transition (obs, act, rew, done, obs_next) of each
 * acting agent i is (obs|i, act[i], rew[i], done bit i, obs_next|i).  Records of an env are consecutive ring
 * slots, so n-step returns follow an agent by walking slots while it keeps acting. */
typedef struct mel_round_replay {
    int32_t   capacity;        /* K ring slots per env                                            */
    int32_t   reserved;
    float*    obs;             /* [B, K, 8N]                                                      */
    float*    obs_next;        /* [B, K, 8N]                                                      */
    uint64_t* acted;           /* [B, K]                                                          */
    uint64_t* done;            /* [B, K]                                                          */
    int8_t*   act;             /* [B, K, N]                                                       */
    float*    rew;             /* [B, K, N]                                                       */
    int32_t*  episode;         /* [B, K] episode ordinal of the env when the round was played     */
    int32_t*  cursor;          /* [B]    rounds recorded so far (slot = cursor % K)               */
} mel_round_replay;

/* Replay sampling in one launch (the learn half's first step, l_dgn.py:246-261 -> [3P] tianshou ReplayBuffer.sample +
 * compute_nstep_return with estimation_step = n_step): `batch` (record, acting agent) pairs drawn uniformly with replacement
 * from the records the ring holds, each followed through consecutive slots while the env's episode is the same and the agent
 * keeps acting.  Outputs (device): obs / boot_obs float [batch, 8N+1] (the record's obs_matrix | agent id; the observation to
 * bootstrap from), act int64, ret float (sum_j discount[j] * rew_j), boot_w float (discount[steps], 0 when the agent
 * terminated inside the window), env / slot / agent int64.  discount: HOST float [n_step + 1] = gamma^j (copied into the launch).
 * Draws are a counter-based function of (seed, *draw_counter, sample index); the launch increments *draw_counter (device
 * uint64), so replaying it from a HIP graph keeps drawing new batches.  scratch: device int32 [n_envs * capacity + 1]. */
#define MEL_REPLAY_MAX_NSTEP 16
typedef struct mel_replay_batch {
    float*   obs;
    float*   boot_obs;
    int64_t* act;
    float*   ret;
    float*   boot_w;
    int64_t* env;
    int64_t* slot;
    int64_t* agent;
} mel_replay_batch;
mel_status mel_replay_sample(const mel_round_replay* replay, int64_t n_envs, int32_t n_nodes, int32_t batch, int32_t n_step,
                             const float* discount, uint64_t seed, uint64_t* draw_counter, int32_t* scratch,
                             const mel_replay_batch* out, void* stream);

/* One Adam update of up to MEL_ADAM_MAX_TENSORS parameter tensors in one launch ([3P] torch.optim.Adam as the reference
 * configures it, l_dgn.py:207: no amsgrad, L2 weight decay): exp_avg <- lerp(exp_avg, g, 1 - beta1); exp_avg_sq <- beta2 exp_avg_sq
 * + (1 - beta2) g^2; param <- param - lr / (1 - beta1^t) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^t) + eps).  All pointers
 * device fp32.  step[i]: the tensor's step counter as torch keeps it in capturable mode (device fp32 scalar, the number of updates
 * DONE: the launch uses t = *step + 1 and a second small launch stores it back) - or NULL for all tensors, then t = host_step. */
#define MEL_ADAM_MAX_TENSORS 64
typedef struct mel_adam_tensors {
    int32_t      count;
    int32_t      reserved;
    float*       param[MEL_ADAM_MAX_TENSORS];
    const float* grad[MEL_ADAM_MAX_TENSORS];
    float*       exp_avg[MEL_ADAM_MAX_TENSORS];
    float*       exp_avg_sq[MEL_ADAM_MAX_TENSORS];
    float*       step[MEL_ADAM_MAX_TENSORS];
    int64_t      numel[MEL_ADAM_MAX_TENSORS];
} mel_adam_tensors;
mel_status mel_adam_step(const mel_adam_tensors* t, float lr, float beta1, float beta2, float eps, float weight_decay,
                         double host_step, void* stream);

/* One whole env ROUND per launch for every env of the batch (round-batched loop): replays, in the
 * reference's AEC order, the dead-agent steps and one GraphEnv.step per active agent with that agent's
 * action until the world step fires or the episode ends (then the env is reset to
 * episode_table[b, ep_cursor % table_stride]).  State after the call is identical to issuing the same
 * steps one at a time through mel_env_step.
 *   actions     device int32 [rows], one per (env, active agent), ordered by env then agent id
 *   row_offsets device int32 [B+1], first action row of each env (from mel_ldgn_forward_agents); NULL =
 *               actions is the dense layout [B, N] indexed by agent id (mel_select_action_envs)
 *   live        device uint64 [B]; in: the active sets the actions belong to, out: the next round's
 *   first != 0  only publishes the current active sets (call once after mel_env_reset).
 *   round_counter (optional, device uint32): incremented once per call.
 *   replay (optional): every env that had acting agents appends one record (see mel_round_replay). */
mel_status mel_env_round(mel_env_batch* env, const mel_episode_pool* pool, const int32_t* actions,
                         const int32_t* row_offsets, uint64_t* live, const int32_t* episode_table,
                         int32_t table_stride, int32_t first, uint32_t* round_counter,
                         const mel_round_replay* replay, void* stream);

/* last() only (mutates is_new_round exactly like GraphEnv.observe, graph.py:205-211). */
mel_status mel_env_observe(mel_env_batch* env, const int32_t* env_ids, int64_t n,
                           const mel_env_obs* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optional stage timer: while a profiler is attached to the calling thread, every launch group of the
 * entry points above is bracketed by a pair of HIP events recorded on the caller's stream
 * (measurement only - bench.py uses it for the live per-kernel roofline; nothing else in the library
 * creates events or synchronises).  mel_prof_read synchronises on the recorded events.
 * ------------------------------------------------------------------------------------------------ */
#define MEL_STAGE_PLAN          0   /* plan_masks + plan_scan + plan_lists                       */
#define MEL_STAGE_ENCODER       1   /* encoder GEMM (layer 0 fused into the A-tile producer)     */
#define MEL_STAGE_CONV1_LIN     2   /* conv1.lin_l (+ lin_r for HL-DGN, one launch)              */
#define MEL_STAGE_CONV1_LIN_R   3   /* conv1.lin_r (L-DGN one-hop rows)                          */
#define MEL_STAGE_CONV1_ATT     4   /* conv1 edge-softmax / aggregate (+ pool for HL-DGN)        */
#define MEL_STAGE_CONV2_LIN     5   /* conv2.lin_l                                               */
#define MEL_STAGE_CONV2_LIN_R   6   /* conv2.lin_r                                               */
#define MEL_STAGE_CONV2_ATT     7   /* conv2 attention for the controlling agent                 */
#define MEL_STAGE_HEAD_HIDDEN   8   /* dueling hidden layers                                     */
#define MEL_STAGE_HEAD_TAIL     9   /* last Linear + q - mean(q) + v                             */
#define MEL_STAGE_SELECT       10   /* mask + argmax + eps-greedy                                */
#define MEL_STAGE_ENV_STEP     11   /* env step (+ observe, + auto reset)                        */
#define MEL_STAGE_ENV_RESET    12
#define MEL_STAGE_ENV_OBSERVE  13
#define MEL_N_STAGES           14

void*      mel_prof_create(int32_t capacity);            /* capacity = max recorded launch groups */
void       mel_prof_destroy(void* prof);
void       mel_prof_attach(void* prof);                   /* NULL detaches                         */
void       mel_prof_reset(void* prof);
/* ms_sum / count: host arrays [MEL_N_STAGES]; returns number of records read (negative on error). */
int32_t    mel_prof_read(void* prof, double* ms_sum, int64_t* count);

const char* mel_last_error(void);
/* sizeof() of the structs of this header as the library was compiled, for binding authors to check their mirrors
 * against: which = 0 mel_linear, 1 mel_gatv2, 2 mel_mlp, 3 mel_weights, 4 mel_select, 5 mel_env_batch,
 * 6 mel_episode_pool, 7 mel_env_obs, 8 mel_round_replay, 9 mel_graph_pool, 10 mel_episode_stream, 11 mel_replay_batch, 12 mel_adam_tensors;
 * 0 for anything else. */
size_t mel_abi_sizeof(int32_t which);
const char* mel_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MELISSA_HIP_H */
