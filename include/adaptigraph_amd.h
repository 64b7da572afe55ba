/*
 * adaptigraph_amd.h - C-ABI of the MI355X-native GNN-dynamics rollout engine.
 *
 * Drop-in boundary for ONE path of jhyau/AdaptiGraph: the GNN dynamics forward /
 * rollout the MPC planner calls.  The reference has no FFI for this path - the
 * boundary there is a Python callable (src/planning/real_world/planner.py:246,270
 * -> src/planning/forward_dynamics.py:12).  This header is the C boundary placed
 * UNDER that callable; adaptigraph_amd/ (Python, ctypes) keeps the reference's
 * Python signatures on top of it.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - Every function returns 0 on success, a negative AG_ERR_* code otherwise;
 *     ag_last_error(ctx) gives the message of the last failure on that ctx.
 *   - Pointers named d_* are DEVICE pointers (e.g. torch tensor.data_ptr()),
 *     h_* are host pointers.  All floating data is fp32, indices int32, masks
 *     uint8 (0/1, the memory layout of a torch.bool tensor).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *     work is enqueued on it; no function synchronises the stream unless its
 *     comment says so.
 *   - The caller owns every input/output buffer.  The library owns only its
 *     ctx workspace (grown monotonically, freed by ag_ctx_destroy).
 *   - One ctx per (process, device); a ctx is not re-entrant (one host thread at a time).  Calls on ONE stream are ordered by
 *     the stream.  Calls issued on DIFFERENT streams run side by side: the launch plan, the repeat table, the workspace and the
 *     pinned read-back buffers of a call belong to a per-stream "call slot" (up to 8 per ctx; r05).  A ninth stream takes over the
 *     least recently used slot and first waits (GPU side, an event) for that slot's last call.  That is what lets a caller deal
 *     the 40 independent dynamics() calls of the planner's chunk loop (plan.py:241-247) to a few streams
 *     (adaptigraph_amd/planner.py).  Early returns before any work was enqueued (argument errors) record nothing.  A call that is
 *     being captured into a hipGraph neither waits nor records: the caller serialises around a capture.
 *   - WHICH ENTRY POINTS BLOCK THE HOST, and when (everything else only enqueues):
 *       ag_ctx_load_weights, ag_ctx_set_precision     always (host repack + copies)
 *       ag_forward, ag_rollout                        once, at the end: they return the overflow verdict (AG_ERR_MAX_NR)
 *       ag_rollout_work                               for its plan (and a base rollout, if none is kept): it returns host numbers
 *       ag_ctx_rollout_counts (after a device-planned call without prefix sharing), ag_ctx_share_counts   wait for the device
 *       ag_rollout_async, ag_rollout_actions          only when the contact-free prefix is in play (option "share_prefix";
 *           y_mode 0, by default batches of >= 64 candidates and >= 32768 rows), and then for SMALL plan kernels at the start
 *           of the call, never for the rollout: (a) a base rollout of this start state is kept in the ctx: ONE wait for census +
 *           state compare + contact plan, enqueued together; (b) the last census of this shape said "not worth it" (e.g. every
 *           push starts on the object): NO wait - a census goes out that a later call reads; (c) otherwise: one wait for the
 *           census, and if it keeps the sharing a second one for the contact plan, the GPU running the base rollout meanwhile.
 *           An event wait on the caller's stream: it also covers whatever the caller enqueued on that stream before the call.
 *           "share_prefix" 0 keeps both purely asynchronous.
 *     Device memory, pinned memory, events and streams are created when a call slot first sees a shape and kept: a repeated call
 *     of the same shape on the same stream allocates nothing (ag_ctx_alloc_counts; hipMalloc / hipFree are device-wide syncs).
 *   - No float atomics anywhere: results are bit-reproducible and independent
 *     of how candidates are chunked or sharded across GPUs.
 */
#ifndef ADAPTIGRAPH_AMD_H
#define ADAPTIGRAPH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AG_OK 0
#define AG_ERR_INVALID -1      /* bad argument (mirrors the reference's asserts, model.py:89,187,222,240,250) */
#define AG_ERR_HIP -2          /* a HIP runtime call failed                                                  */
#define AG_ERR_MAX_NR -3       /* a graph had more edges than max_nR: reference raises Exception("Exceeds max dims"),
                                  src/dynamics/utils.py:63-65 via forward_dynamics.py:127-128,173-174          */
#define AG_ERR_UNSUPPORTED -4  /* configuration outside what the kernels implement                            */
#define AG_ERR_NO_WEIGHTS -5   /* forward/rollout before ag_ctx_load_weights                                   */

#define AG_ABI_VERSION 7
#define AG_NUM_WEIGHT_TENSORS 22

typedef struct ag_ctx ag_ctx;

/* Model dimensions = what DynamicsPredictor.__init__ derives from model_config
 * (src/dynamics/gnn/model.py:78-123).  Every shipped config has nf=150, in_dim=6 (attr 2 + physics 1 + action 3) and
 * rel_dim = 2*attr 2 + group 1 + 3*n_his: 17 with n_his=4 (rope, granular, cloth, ... and every planner task config),
 * 20 with n_his=5 (config/dynamics/softbody.yaml:29).  Both are served by ag_forward and by the rollout driver (dynamics()
 * takes n_his from the task config it is handed, forward_dynamics.py:16); the bf16x3 arithmetic and the latency-mode chains
 * are built for n_his=4 (ag_ctx_set_precision returns AG_ERR_UNSUPPORTED for an n_his=5 model; small n_his=5 launches
 * simply run the throughput kernels). */
typedef struct ag_dims {
    int32_t nf;            /* nf_particle == nf_relation == nf_effect; kernels are built for 150 */
    int32_t n_his;         /* history frames: 4 or 5                                             */
    int32_t pstep;         /* message-passing rounds (3; softbody.yaml uses 4)                   */
    int32_t in_dim;        /* particle-encoder input width, must be 6                            */
    int32_t rel_dim;       /* relation-encoder input width, must be 5 + 3*n_his                  */
    float motion_clamp;    /* model.py:86, 100.0                                                 */
} ag_dims;

/* Pusher/tool description for the rollout driver (forward_dynamics.py:40-81,163-168). */
typedef struct ag_rollout_params {
    int32_t B;             /* candidates                                                          */
    int32_t H;             /* look-ahead steps (n_look_forward); 1 for the masked variant         */
    int32_t N_o;           /* object particles (max_nobj)                                         */
    int32_t M;             /* tool particles (eef_num)                                            */
    int32_t topk;
    int32_t connect_tools_all;
    int32_t max_nR;        /* reference raises when a graph has more edges                        */
    int32_t y_mode;        /* 0: tool y = min object y (dynamics, :40,:163); 1: masked mean (dynamics_masked, :235,:359) */
    float adj_thresh;
    float gripper_offset;  /* fp32(0.01*sim_real_ratio) if gripper_enable else 0 (:80-81,:167-168) */
    int32_t gripper_enable;
    float physics_param;   /* homogeneous physics parameter (forward_dynamics.py:151)             */
} ag_rollout_params;

uint32_t ag_abi_version(void);

/* Create / destroy.  device_id is the HIP ordinal.  Does not touch the GPU beyond hipSetDevice + small mallocs. */
int ag_ctx_create(int32_t device_id, const ag_dims* dims, ag_ctx** out_ctx);
int ag_ctx_destroy(ag_ctx* ctx);
const char* ag_last_error(const ag_ctx* ctx);

/* Upload the 22 state_dict tensors of DynamicsPredictor (model.py:104-123), HOST pointers, torch layout
 * (weight = (out,in) row-major), in this order:
 *   0..5   particle_encoder.model.{0,2,4}.{weight,bias}      (150x6,150 | 150x150,150 | 150x150,150)
 *   6..11  relation_encoder.model.{0,2,4}.{weight,bias}      (150x17,150 | 150x150,150 | 150x150,150)
 *   12,13  particle_propagator.linear.{weight,bias}          (150x300,150)
 *   14,15  relation_propagator.linear.{weight,bias}          (150x450,150)
 *   16..21 non_rigid_predictor.linear_{0,1,2}.{weight,bias}  (150x150,150 | 150x150,150 | 3x150,3)
 * Synchronous (repacks on the host, copies, waits). */
int ag_ctx_load_weights(ag_ctx* ctx, const float* const* h_tensors, int32_t n_tensors);

/* Arithmetic of the MLP chains.  0 (default): exact fp32 on v_mfma_f32_32x32x2_f32.  1: "bf16x3" - every fp32 operand
 * is split exactly into three bf16 pieces and each product is rebuilt from its six leading partial products on the
 * bf16 matrix pipe with fp32 accumulation (error per product ~2^-24, i.e. fp32-grade; ~2.7x the fp32-MFMA rate).
 * Edge construction is unaffected (always exact).  Re-derives the self-loop constant rows; synchronous. */
int ag_ctx_set_precision(ag_ctx* ctx, int32_t mode);

/* Tuning: candidates per launch wave of the rollout (0 = automatic). */
int ag_ctx_set_chunk(ag_ctx* ctx, int32_t candidates_per_chunk);

/* Per-context switches between bit-identical execution paths (A/B measurements, tests) - with ONE exception, "device_decode",
 * which moves the cos / sin of the action decode to the device (see there).  A context takes its defaults
 * from the environment ONCE, at ag_ctx_create (AG_* name in brackets; values outside an option's range are clamped into it);
 * afterwards only these calls change them, so two contexts of one process can differ.  No call of the library reads the
 * environment after ag_ctx_create.
 *   "streams"        [AG_STREAMS]         in-library HIP streams of a rollout: 0 = by batch size (default), 1..4
 *   "chunk"          [AG_CHUNK]           candidates per launch chunk, 0 = automatic (ag_ctx_set_chunk takes precedence)
 *   "latency"        [AG_LATENCY]         latency-mode chains: -1 by launch size (default), 0 never, 1 always
 *   "ragged"         [AG_NO_RAGGED=1 -> 0]      masked rollouts walk a compact row list (default 1)
 *   "ell_graph"      [AG_NO_ELL_GRAPH=1 -> 0]   rollout graphs stay slot-indexed, no CSR emit pass (default 1)
 *   "self_dedupe"    [AG_NO_SELF_DEDUPE=1 -> 0] self-loop edges skip the relation encoder (default 1)
 *   "repeat_sort"    [AG_NO_REPEAT_SORT=1 -> 0] repeat-aware launch order of ag_rollout (default 1, see there)
 *   "edge_wgs"       [AG_EDGE_WGS]        workgroups the edge builder aims at per launch (default 256)
 *   "edge_block_min" [AG_EDGE_BLOCK_MIN]  rows per slice from which the 64-rows-per-wavefront schedule is used (-1 = built-in 256)
 *   "enc_persist"    [AG_ENC_PERSIST]     persistent workgroups of k_edge_enc (default 0 = one workgroup per tile)
 *   "stagger_us"     [AG_STAGGER_US]      start offset between the two workgroups of a CU in the propagate chains (default 0)
 *   "zigzag"         [AG_ZIGZAG]          odd message-passing rounds walk the row tiles backwards (default 1; Infinity-Cache reuse of
 *                                         the C rows the previous round read last)
 *   "device_decode"  [AG_DEVICE_DECODE]   consumed by the Python shim: dynamics() with GPU-resident actions goes through
 *                                         ag_rollout_actions: -1 when task_config bounds the repeat (default), 0 never, 1 always.
 *                                         NOT bit-neutral: the decode's cos / sin are then the device's, 'action_seqs' agrees
 *                                         with a host decode to ~1e-7 (and a rollout can part from the host-decoded one at a
 *                                         near-tie of the edge selection).  No reference fixture pins GPU-evaluated trig:
 *                                         parity unpinned for that decode; the goldens are compared on the host-decode path
 *   "share_first"    [AG_SHARE_FIRST]     first forward of an ag_rollout* call with y_mode 0 (one start state broadcast to all
 *                                         candidates, forward_dynamics.py:25): the relation encoder runs once over the
 *                                         object-object edges of the start state's tool-free graph, every candidate reads those
 *                                         C rows; -1 = batches of >= 8 candidates (default), 0 never, 1 whenever possible
 *   "share_prefix"   [AG_SHARE_PREFIX]    contact-free prefix of look-ahead step 0 (y_mode 0): a candidate whose tool has not yet come
 *                                         within adj_thresh of an object particle has no tool edge (graph.py:253-286), so its
 *                                         object particles evolve exactly like the start state without a tool.  That base rollout runs
 *                                         once per call; every candidate is stepped only from its first contact on and one that never
 *                                         touches takes the base state of its last step (same bits as stepping it).  -1 = batches of
 *                                         >= 64 candidates and >= 32768 particle rows of which at most half touch at the first forward
 *                                         (a census: one more tiny kernel and wait) (default), 0 never, 1 whenever possible.  The call
 *                                         WAITS once for the contact plan (the GPU is running the base rollout meanwhile), so
 *                                         ag_rollout_async / ag_rollout_actions are then not purely asynchronous
 *   "stream_min_rows" [AG_STREAM_MIN_ROWS] batches below this many rows (candidates x particles) stay on the caller's stream
 *                                         (default 32768: small batches are dispatch-bound, a second stream only doubles the launches)
 *   "pipeline_fork"  [AG_PIPELINE_FORK]   0 (default): a call that starts while a call issued on ANOTHER caller stream is still running
 *                                         does not fork onto in-library streams (the caller is already spreading independent calls
 *                                         over streams); 1: it forks as usual
 * Unknown names and values outside an option's range return AG_ERR_INVALID (ranges: stream_min_rows 0..INT32_MAX, pipeline_fork 0..1, streams 0..4, chunk 0..2^20, latency -1..1,
 * the 0/1 switches 0..1, edge_wgs 1..65536, edge_block_min -1..INT32_MAX, enc_persist 0..2^20, stagger_us 0..1000,
 * device_decode -1..1, share_first -1..1, share_prefix -1..1). */
int ag_ctx_set_option(ag_ctx* ctx, const char* name, int32_t value);
int ag_ctx_get_option(ag_ctx* ctx, const char* name, int32_t* out_value);

/* Candidate-forwards of the LAST ag_rollout / ag_rollout_async / ag_rollout_actions call on this context: executed (sum over
 * launches of the candidates each model forward was launched over) and needed (sum of action_repeat over the batch).  With
 * the repeat-aware launch order (default; masked batches included) the two are equal; with "repeat_sort" 0 every candidate of
 * a launch chunk is stepped to the chunk's maximum, as the reference steps the whole batch to the batch maximum
 * (forward_dynamics.py:156-161); with "share_prefix" active executed (which then includes the base rollout's forwards) is
 * SMALLER than needed: the forwards before a candidate's first contact are the base rollout's.  After an ag_rollout_actions
 * call without prefix sharing the call waits for the device (the sums live there). */
int ag_ctx_rollout_counts(ag_ctx* ctx, int64_t* out_executed, int64_t* out_needed);

/* Model forwards (per launch chunk and look-ahead step) ENQUEUED by the LAST ag_rollout / ag_rollout_async /
 * ag_rollout_actions call: out2[0] = enqueued, out2[1] = what the loop bounds alone give (host plan: the chunk maxima, equal
 * to out2[0]; device plan: max_repeat per chunk and look-ahead step).  On the device-planned path the chunk maxima come back
 * to the host asynchronously (pinned memory + event, never waited for); once they have landed the enqueue loop stops a
 * look-ahead step at the chunk's own maximum, so out2[0] <= out2[1].  Host bookkeeping only: no device access. */
int ag_ctx_launch_counts(ag_ctx* ctx, int64_t* out2);

/* out[0] = number of device / pinned allocations and frees, event and stream creations this context has made so far.  A
 * steady-state call - same shape, same stream as an earlier one - makes none (tests/test_gpu_share_prefix.py).  Host
 * bookkeeping only. */
int ag_ctx_alloc_counts(ag_ctx* ctx, int64_t* out1);

/* Shared first forward ("share_first") of the LAST ag_rollout / ag_rollout_async / ag_rollout_actions call on this context:
 * out3[0] = edges the once-per-call base encode ran over (0 when the call did not share), out3[1] = edge slots of all
 * candidates that took their C row from the shared table, out3[2] = edge slots the candidates encoded themselves at that
 * forward (edges with a tool at either end).  Without sharing the relation encoder would have run over out3[1] + out3[2]
 * edges.  Waits for the device. */
int ag_ctx_share_counts(ag_ctx* ctx, int64_t* out3);

/* Replaces construct_edges_from_states_batch (src/dynamics/dataset/graph.py:233-298).
 *   d_pos (B,N,3); d_mask,d_tool_mask (B,N) uint8; adj_thresh scalar, or d_adj_thresh_vec (B,) if non-NULL.
 * Outputs, per batch element b, in the reference's nonzero order (sorted by receiver, then sender):
 *   d_recv,d_send (B,edge_cap) int32 (entries past n_edges[b] are left untouched),
 *   d_row_ptr (B,N+1) int32 CSR offsets by receiver, d_n_edges (B,) int32 = TRUE edge count even when > edge_cap
 *   (then nothing is written for that element).  The caller compares n_edges with its max_nR (pad_torch semantics). */
int ag_build_edges(ag_ctx* ctx, void* stream, const float* d_pos, const uint8_t* d_mask, const uint8_t* d_tool_mask,
                   int32_t B, int32_t N, float adj_thresh, const float* d_adj_thresh_vec, int32_t topk,
                   int32_t connect_tools_all, int32_t edge_cap, int32_t* d_recv, int32_t* d_send,
                   int32_t* d_row_ptr, int32_t* d_n_edges);

/* Replaces the default-argument path of construct_edges_from_states (src/dynamics/dataset/graph.py:68-231, single
 * graph; the dataset / eval-rollout builder).  Differences from the batch builder that are reproduced: the squared
 * threshold is formed in double precision and then rounded (h: thr2 = (float)((double)adj*adj), graph.py:86,101 - one ulp
 * away from the batch builder's fp32 square for e.g. 0.4); connect_tools_all is unconditional and leaves no
 * tool<->tool edge (graph.py:119-122).  cull_radius: any float with cull_radius^2 >= thr2 (e.g. nextafter(adj)).
 * The tool-surface / kNN / non-fixed-particle options of that function (max_y, kNN, ...) are applied afterwards, one
 * rule at a time, with ag_edges_apply_tool_rule. */
int ag_build_edges_single(ag_ctx* ctx, void* stream, const float* d_pos, const uint8_t* d_mask, const uint8_t* d_tool_mask,
                          int32_t N, float thr2, float cull_radius, int32_t topk, int32_t connect_tools_all,
                          int32_t edge_cap, int32_t* d_recv, int32_t* d_send, int32_t* d_row_ptr, int32_t* d_n_edges);

/* One tool-attachment rule of construct_edges_from_states applied to an edge list produced by ag_build_edges_single
 * (src/dynamics/dataset/graph.py:144-170 "tool to all non-fixed particles", :208-218 "tool to the closest surfaces").
 * d_subset (N,) uint8 marks the rule's particle subset S (the kernel ANDs it with d_mask); forming S from max_y, the
 * plane bounds etc. is scalar host arithmetic (graph.py:134-143, :190-207) and stays with the caller.  Effect:
 *   edges (tool receiver <- sender in S) are removed; every (receiver in S <- tool) edge is present; if 0 < kNN < 1
 *   only the keepK = (int)(kNN * #pairs) of those pairs with the smallest fp32 distance survive, ranked over the flat
 *   row-major pair list, ties by pair position (graph.py:156-169); no tool<->tool edge remains.
 * n_tools must equal the number of set entries of d_tool_mask (else *d_n_out = -1 and nothing else is written).
 * Outputs like ag_build_edges_single: sorted by (receiver, sender); d_n_out is the TRUE count even when > edge_cap (then
 * d_recv_out/d_send_out are not written).  Input and output arrays must not overlap. */
int ag_edges_apply_tool_rule(ag_ctx* ctx, void* stream, const float* d_pos, const uint8_t* d_mask, const uint8_t* d_tool_mask,
                             int32_t N, int32_t n_tools, const int32_t* d_send_in, const int32_t* d_row_ptr_in,
                             const uint8_t* d_subset, double kNN, int32_t edge_cap, int32_t* d_recv_out, int32_t* d_send_out,
                             int32_t* d_row_ptr_out, int32_t* d_n_out);

/* Replaces DynamicsPredictor.forward (src/dynamics/gnn/model.py:130-342) on index-list graphs.
 *   d_state (B,n_his,N,3) with the ctx's n_his; d_attrs (B,N,2); d_action (B,N,3); d_phys (B,N) physics parameter per particle, zero
 *   for the trailing N-n_p tool particles (model.py:206-207); d_group (B,N,n_inst) = [p_instance ; 0] (model.py:264);
 *   edges as produced by ag_build_edges (must be sorted by receiver; row_ptr consistent).
 * Outputs d_pred_pos, d_pred_motion (B,n_p,3) (model.py:335-338).
 * A graph whose d_n_edges[b] exceeds edge_cap (ag_build_edges reports the true count and writes no indices then) is
 * never walked: the call returns AG_ERR_MAX_NR - the reference raises Exception("Exceeds max dims") at the pad_torch in
 * front of its forward (utils.py:63-65).  Synchronises the stream once at the end to read that flag. */
int ag_forward(ag_ctx* ctx, void* stream, const float* d_state, const float* d_attrs, const float* d_action,
               const float* d_phys, const float* d_group, int32_t n_inst, const int32_t* d_recv, const int32_t* d_send,
               const int32_t* d_row_ptr, const int32_t* d_n_edges, int32_t edge_cap, int32_t B, int32_t N, int32_t n_p,
               float* d_pred_pos, float* d_pred_motion);

/* Replaces the device side of dynamics() / dynamics_masked() (src/planning/forward_dynamics.py:12-205, 209-399):
 * the whole look-ahead x action-repeat loop, graph rebuilt every step, no host sync inside.
 *   d_state0      y_mode 0: (N_o,3) one start cloud broadcast to all candidates (:25);
 *                 y_mode 1: (B,N_o,3) per-candidate padded clouds (:225-227)
 *   d_obj_mask    (B,N_o) uint8 or NULL (= all valid)                                   (:107-115 / :302-309)
 *   d_eef_xz      (B,H,M,2) tool start x,z per look-ahead step; d_eef_delta (B,H,M,3)   (:42-75, computed by the shim
 *                 with torch CPU ops exactly as the reference does, so cos/sin bits match)
 *   h_repeat      (B,H) int32 HOST array = action_repeat (plan_utils.py:16).  A candidate is stepped exactly
 *                 h_repeat[b,h] times in look-ahead step h: per launch chunk the candidates are put in descending order
 *                 of their repeat count and every step is launched over the prefix that is still live (the reference steps
 *                 all of them to the batch maximum and discards the surplus, forward_dynamics.py:156-161; same outputs)
 *   d_phys_vec    NULL (use p->physics_param for every object particle), or (N_o,) per-particle physics parameters
 *                 shared by all candidates (the (B,n_p) branch of model.py:200-204 fed by forward_dynamics.py:151)
 *   d_state_seqs  (B,H,N_o,3) output, fully written (zeros where repeat==0, :32)
 * Synchronises the stream once at the end to read the overflow flag; returns AG_ERR_MAX_NR if any consumed graph
 * had more than max_nR edges (the shim re-raises Exception("Exceeds max dims")). */
int ag_rollout(ag_ctx* ctx, void* stream, const ag_rollout_params* p, const float* d_state0, const uint8_t* d_obj_mask,
               const float* d_eef_xz, const float* d_eef_delta, const int32_t* h_repeat, const float* d_phys_vec,
               float* d_state_seqs);

/* Same, but does not wait for the rollout: enqueue only.  *d_overflow_flag (int32, device, caller-zeroed) receives the max
 * edge count seen if it exceeded max_nR.  Used by bench.py to time the pure device path.  One exception: a call that shares the
 * contact-free prefix (option "share_prefix") waits for small plan kernels at its start, never for the rollout - when and how
 * often is stated once, at the top of this header ("WHICH ENTRY POINTS BLOCK THE HOST"). */
int ag_rollout_async(ag_ctx* ctx, void* stream, const ag_rollout_params* p, const float* d_state0,
                     const uint8_t* d_obj_mask, const float* d_eef_xz, const float* d_eef_delta,
                     const int32_t* h_repeat, const float* d_phys_vec, float* d_state_seqs,
                     int32_t* d_overflow_flag);

/* dynamics() for actions that are RESIDENT ON THE GPU (the planner samples them there): decode_action
 * (src/planning/plan_utils.py:11-20), the tool-keypoint layout (forward_dynamics.py:42-75) and the repeat-aware launch plan
 * all run in one device kernel; the host never reads an action, so nothing between the caller's sampling kernel and the
 * first rollout kernel waits for the GPU.  Enqueue only (like ag_rollout_async, incl. its "share_prefix" exception).
 *   d_action        (B,H,4) raw [x, z, theta, length]
 *   push_length     task_config push_length;  h_tool_offsets (M,) HOST: pusher_points[k][1] * sim_real_ratio (entry 0 unused;
 *                   may be NULL when M == 1)
 *   max_repeat      upper bound of action_repeat = int(length) the caller guarantees (e.g. its action_upper_lim[3]); a
 *                   look-ahead step is launched at most max_repeat times: steps past a chunk's own maximum find no live slot
 *                   and exit, and are no longer enqueued once the plan's maxima have reached the host (ag_ctx_launch_counts)
 *   d_action_seqs   (B,H,4) output: decoded actions [x_start, z_start, x_end, z_end] (the reference's 'action_seqs')
 *   d_flags         (>= 2 int32, device, caller-zeroed): [0] max edge count seen if it exceeded max_nR (as ag_rollout_async),
 *                   [1] largest action_repeat seen if it exceeded max_repeat (the results of such a candidate are invalid)
 * cos / sin are the device's: decoded values agree with a host decode to an ulp or two (a CUDA-resident reference would
 * use device transcendental functions too); everything downstream is the same arithmetic as ag_rollout. y_mode must be 0
 * (the masked variant takes host-decoded actions), M <= 8, 0 <= max_repeat <= 1024 (AG_ERR_INVALID / AG_ERR_UNSUPPORTED
 * otherwise). */
int ag_rollout_actions(ag_ctx* ctx, void* stream, const ag_rollout_params* p, const float* d_state0, const float* d_action,
                       float push_length, const float* h_tool_offsets, int32_t max_repeat, const float* d_phys_vec,
                       float* d_state_seqs, float* d_action_seqs, int32_t* d_flags);

/* The work an ag_rollout_actions call on (d_state0, d_action) would do, per candidate, WITHOUT rolling anything out:
 * h_work[b] (HOST, (B,) int32) = model forwards candidate b would be stepped = sum over look-ahead steps of min(action_repeat,
 * max_repeat), where look-ahead step 0 counts only the forwards from the candidate's first contact on when the contact-free
 * prefix applies to the batch (option "share_prefix": a candidate that never touches counts 0).  Runs the plan kernels and, if
 * none is kept for this start state, the tool-free base rollout - which then stays in the ctx for the rollout call that follows.
 * For cutting WORK-balanced shards of a candidate batch across GPUs (adaptigraph_amd/sharding.py): every rank calls
 * it on the full batch and gets the same numbers, no exchange.  The reference has no counterpart (one device, plan.py:87).
 * Arguments as ag_rollout_actions.  Synchronous (waits for the plan). */
int ag_rollout_work(ag_ctx* ctx, void* stream, const ag_rollout_params* p, const float* d_state0, const float* d_action,
                    float push_length, const float* h_tool_offsets, int32_t max_repeat, const float* d_phys_vec, int32_t* h_work);

/* ---- Per-candidate cost functions: SURVEY §8(f) rank 1 (reference src/planning/losses.py, src/planning/plan.py:27-59) ---- */

/* chamfer(x, y) (losses.py:4-10): d_x (R,N,3); d_y (By,M,3) with By == 1 (one target for all rows, plan.py:146) or
 * By == R; optional uint8 masks d_xmask (R,N), d_ymask (By,M) keep only masked-in points (mean_chamfer, losses.py:12-24).
 * d_out (R,).  N + M must fit the LDS tile (<= ~13k points). */
int ag_cost_chamfer(ag_ctx* ctx, void* stream, const float* d_x, const float* d_y, const uint8_t* d_xmask,
                    const uint8_t* d_ymask, int32_t R, int32_t N, int32_t M, int32_t By, float* d_out);

/* Particle statistics of d_state (R,N,3) -> d_out (R,5) = [box_loss, xmin, xmax, zmin, zmax]: box_loss (losses.py:26-35)
 * against h_box4 = {xmin, xmax, zmin, zmax} (NULL: entry 0 is 0), and the x/z bounds running_cost needs (plan.py:41-44). */
int ag_cost_state_stats(ag_ctx* ctx, void* stream, const float* d_state, int32_t R, int32_t N, const float* h_box4,
                        float* d_out);

/* Collision penalties (losses.py:37-92): kind 0 rope, 1 cloth, 2 granular.  d_state_pred (B,H,N,3), d_action (B,H,4)
 * raw [x,z,theta,len], d_state_init (N,3).  d_out (B,H,2) = [exp(-max(dmin - size,0)*100), min(dmax, 0.4*ratio)];
 * rope/granular: entry 0 is the penalty; cloth: 1 - e0 - 0.2 * e1 / max_batch(e1) (the caller owns the global max). */
int ag_cost_penalty(ag_ctx* ctx, void* stream, const float* d_state_pred, const float* d_action,
                    const float* d_state_init, int32_t B, int32_t H, int32_t N, int32_t kind, float sim_real_ratio,
                    float* d_out);

/* cloth_penalty's tail (losses.py:62-63) on the (B,H,2) output of ag_cost_penalty(kind 1): d_out[i] = 1 - e0 - 0.2 * e1 / max(e1),
 * n = B*H entries.  d_dmax: NULL = the maximum over this batch is formed here; else a device float holding it (a sharded batch
 * all-reduces it first).  One launch. */
int ag_cost_cloth_combine(ag_ctx* ctx, void* stream, const float* d_raw, const float* d_dmax, int64_t n, float* d_out);

/* What is left of running_cost (src/planning/plan.py:35-53) once the particle reductions are done, in one launch:
 *   error_weight = fp32(2 / (double(max error) + 1e-6)); box penalty from the x / z bounds of ag_cost_state_stats against
 *   h_bbox4 = {x_lo, x_hi, z_lo, z_hi} (doubles, rounded to fp32 as torch rounds a Python scalar); reward[b] =
 *   -error_weight * error[b,H-1] - 5 * mean_h penalty[b,h] - 5 * mean_h box_penalty[b,h].
 * d_error, d_penalty (B,H); d_stats (B*H,5) as ag_cost_state_stats writes it; d_error_max: NULL = batch maximum formed here,
 * else a device float holding it (sharded batches all-reduce it first); d_reward (B,). */
int ag_cost_reward(ag_ctx* ctx, void* stream, const float* d_error, const float* d_penalty, const float* d_stats,
                   const float* d_error_max, const double* h_bbox4, int32_t B, int32_t H, float* d_reward);

/* ---- MPPI sampling / update: SURVEY §8(f) rank 2 (reference src/planning/plan_utils.py:31-101) ---- */

/* sample_action_seq (plan_utils.py:42-77).  d_act_seq (H,4) nominal actions [x, z, theta, length]; d_lo, d_hi (4,)
 * action limits; d_out (S,H,4).
 *   mode 0 (iter_index == 0, :48-50): d_rnd (S,H,4) uniform [0,1) draws -> d_out = u*(hi-lo)+lo; d_act_seq, d_scale unused.
 *   mode 1 (:51-77): d_rnd (H,S,4) = for look-ahead step i the (S,4) draws of N(0, noise_level) in the reference's draw
 *     order; d_scale (H,) = fp32(0.1 * 10^i) (:62); start and end point of the nominal push are perturbed, re-encoded
 *     as (theta, length) and limited (:31-39); sample 0 keeps the nominal action (:75).
 * The random draws are an input so that the function is testable against the reference's vectors. */
int ag_mppi_sample(ag_ctx* ctx, void* stream, const float* d_act_seq, const float* d_lo, const float* d_hi,
                   const float* d_rnd, const float* d_scale, int32_t S, int32_t H, int32_t mode, float push_length,
                   float* d_out);

/* optimize_action_mppi (plan_utils.py:80-101): softmax(reward * reward_weight) over the B candidates, weighted mean of
 * the start and end points per look-ahead step, re-encoded and limited.  d_act_seqs (B,H,4), d_reward (B,), d_out (H,4).
 * One workgroup per look-ahead step, fixed-order reductions (deterministic). */
int ag_mppi_update(ag_ctx* ctx, void* stream, const float* d_act_seqs, const float* d_reward, const float* d_lo,
                   const float* d_hi, int32_t B, int32_t H, float reward_weight, float push_length, float* d_out);

/* clip_actions (plan_utils.py:35-39) on n actions: theta wrapped into [-pi, pi), every component clamped. */
int ag_mppi_clip(ag_ctx* ctx, void* stream, const float* d_in, const float* d_lo, const float* d_hi, int64_t n,
                 float* d_out);

/* Introspection for bench.py / tests: HIP-event time of every launch of a kernel family, recorded on the stream the
 * kernels run on.  family_mask bit i enables family i of: edge_count, edge_emit, prep, node_enc, edge_enc, mp,
 * node_prop, node_final, roll_init, roll_update, cost (0 = off; "mp" is kept for index stability and never records: the
 * message passing is fused into node_prop / node_final).  A non-zero mask pins the rollout to ONE stream so that a
 * duration measures the kernel alone; bit 30 keeps the streams instead (durations then include the other stream's
 * co-running kernels - what a kernel trace of a normal run shows).  ag_ctx_kernel_stats waits for the recorded
 * events and returns the total milliseconds and launch count since the last reset. */
int ag_ctx_set_profiling(ag_ctx* ctx, int32_t family_mask);
int ag_ctx_kernel_stats(ag_ctx* ctx, const char* kernel, double* out_total_ms, int64_t* out_launches);
int ag_ctx_reset_stats(ag_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* ADAPTIGRAPH_AMD_H */
