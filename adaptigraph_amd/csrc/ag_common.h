// Shared constants and host-side declarations for the gfx950 kernels (internal, not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ag {

// ---- model geometry (every shipped config: src/config/dynamics/*.yaml model_config) --------------------
constexpr int NF = 150;        // real feature width of every hidden layer
constexpr int NFP = 160;       // row pitch of every activation buffer (5 MFMA tiles of 32)
constexpr int ONE_F = 150;     // slot that carries the constant 1.0 (bias rides in weight column 150)
constexpr int N_HIS = 4;       // history frames of the ROLLOUT path (every planner task config: n_his 4)
constexpr int IN_DIM = 6;      // [attr_obj, attr_tool, phys, act_x, act_y, act_z]
constexpr int REL_DIM = 17;    // [attr_r(2), attr_s(2), group_diff, (res0,res1,res2,cur)_r - (...)_s]
constexpr int F12 = 3 * N_HIS; // per-particle history feature row
// The forward path (ag_forward) also serves the softbody model variant: n_his = 5, rel_input_dim = 20
// (src/config/dynamics/softbody.yaml:29).  Its feature rows are 15 floats at a 16-float pitch.
constexpr int N_HIS_MAX = 5;
constexpr int F15_PITCH = 16;
inline int feat_pitch(int n_his) { return n_his == 5 ? F15_PITCH : F12; }
constexpr int NODE_IN = 8;     // node input row: 6 features, 1.0, pad

// ---- packed weight geometry ----------------------------------------------------------------------------
// A hidden layer consumes K = 152 input slots = 76 MFMA k-steps (v_mfma_f32_32x32x2_f32 eats 2 k per step:
// lanes 0-31 supply k_a, lanes 32-63 supply k_b).  Step s reads accumulator register (tile t, reg r) of the
// previous layer: t = s/16, r = s%16 (tile 4 only has r < 12), i.e. features 32t + (r&3) + 8(r>>2) + 4h.
// Weights are packed [chunk q = s/4][m-block][lane][4 steps] so one ds_read_b128 feeds 4 MFMAs.
constexpr int KSTEPS = 76;
constexpr int KCH = 19;               // chunks of 4 steps
constexpr int KCH_H0 = 10;            // chunks staged in the first half-phase
constexpr int KCH_H1 = KCH - KCH_H0;  // 9
constexpr int CHUNK_FLOATS_MB5 = 5 * 64 * 4;  // 1280 floats per chunk for a 160-wide output
constexpr int CHUNK_FLOATS_MB1 = 1 * 64 * 4;
constexpr int HALF0_FLOATS = KCH_H0 * CHUNK_FLOATS_MB5;  // 12800
constexpr int HALF1_FLOATS = KCH_H1 * CHUNK_FLOATS_MB5;  // 11520
constexpr int LAYER_FLOATS = HALF0_FLOATS + HALF1_FLOATS;  // 24320
constexpr int EDGE_L1_CHUNKS = 3;     // 18 inputs -> 9 steps, padded to 12
constexpr int NODE_L1_CHUNKS = 1;     // 8 inputs -> 4 steps
constexpr int OUT3_FLOATS = KCH * CHUNK_FLOATS_MB1;  // predictor head (3 outputs, one m-block)

// Offsets (floats) into the packed weight blob, in the order the chains stage them.
struct WeightLayout {
    // edge chain: L1, L2, L3, W1(+b_rp)
    static constexpr int E_L1 = 0;
    static constexpr int E_L2 = E_L1 + EDGE_L1_CHUNKS * CHUNK_FLOATS_MB5;
    static constexpr int E_L3 = E_L2 + LAYER_FLOATS;
    static constexpr int E_W1 = E_L3 + LAYER_FLOATS;
    // node encode chain: L1, L2, L3, Wa(+b_pp), W2, W3
    static constexpr int N_L1 = E_W1 + LAYER_FLOATS;
    static constexpr int N_L2 = N_L1 + NODE_L1_CHUNKS * CHUNK_FLOATS_MB5;
    static constexpr int N_L3 = N_L2 + LAYER_FLOATS;
    static constexpr int N_WA = N_L3 + LAYER_FLOATS;
    static constexpr int N_W2 = N_WA + LAYER_FLOATS;
    static constexpr int N_W3 = N_W2 + LAYER_FLOATS;
    // node propagate chain: Wb, then (W2, W3) again, or the predictor P0, P1, P2
    static constexpr int P_WB = N_W3 + LAYER_FLOATS;
    static constexpr int P_P0 = P_WB + LAYER_FLOATS;
    static constexpr int P_P1 = P_P0 + LAYER_FLOATS;
    static constexpr int P_P2 = P_P1 + LAYER_FLOATS;
    static constexpr int TOTAL = P_P2 + OUT3_FLOATS;
};

// ---- kernel families (profiling ids) ------------------------------------------------------------------
enum Family { FAM_EDGE_COUNT = 0, FAM_EDGE_EMIT, FAM_PREP, FAM_NODE_ENC, FAM_EDGE_ENC, FAM_MP, FAM_NODE_PROP,
              FAM_NODE_FINAL, FAM_ROLL_INIT, FAM_ROLL_UPDATE, FAM_COST, FAM_COUNT };

// ---- per-context tuning / A-B switches.  Defaults come from the environment ONCE, at ag_ctx_create (the AG_* name in
// brackets); ag_ctx_set_option changes them per context afterwards.  None of them changes a result (bit-identical paths),
// with ONE exception: device_decode (see there).  Every option has a valid range (kOptions in ag_api.hip): ag_ctx_set_option
// refuses a value outside it, values from the environment are clamped into it.
struct Options {
    int streams = 0;          // [AG_STREAMS]        in-library streams of a rollout: 0 = by batch size, else 1..4
    int chunk = 0;            // [AG_CHUNK]          candidates per launch chunk: 0 = automatic (ag_ctx_set_chunk overrides)
    int latency = -1;         // [AG_LATENCY]        latency-mode chains: -1 by size, 0 never, 1 always
    int ragged = 1;           // [AG_NO_RAGGED]      masked rollouts walk a compact row list
    int ell_graph = 1;        // [AG_NO_ELL_GRAPH]   rollout graphs stay slot-indexed (no CSR emit pass)
    int self_dedupe = 1;      // [AG_NO_SELF_DEDUPE] self-loop edges skip the relation encoder
    int repeat_sort = 1;      // [AG_NO_REPEAT_SORT] candidates of a chunk ordered by action_repeat, finished ones dropped
    int edge_wgs = 256;       // [AG_EDGE_WGS]       edge-builder workgroups aimed at per launch
    int edge_block_min = -1;  // [AG_EDGE_BLOCK_MIN] rows per slice from which the 64-rows-per-wavefront schedule is used (-1: built-in)
    int enc_persist = 0;      // [AG_ENC_PERSIST]    persistent workgroups of k_edge_enc (0 = one per tile)
    int stagger_us = 0;       // [AG_STAGGER_US]     offset between the two workgroups of a CU in the propagate chains
    int zigzag = 1;           // [AG_ZIGZAG]         odd message-passing rounds walk the row tiles backwards: what the previous round
                              //                     read last from HBM is read first, while it is still in the 256-MB Infinity Cache
                              //                     (one stream: k_node_prop 170.5 -> 168.3, final round 83.4 -> 81.4 ms per rollout;
                              //                     four streams: 477.3 -> 475.8 ms - the other chunks' traffic evicts most of it)
    int device_decode = -1;   // [AG_DEVICE_DECODE]  consumed by the Python shim: dynamics() hands GPU-resident actions to
                              //                     ag_rollout_actions (-1: when the task config bounds the repeat, 0 never, 1 always).
                              //                     The one switch that is NOT bit-neutral: cos/sin of the decode are then the device's,
                              //                     so action_seqs agrees with a host decode to ~1e-7, not bit for bit
    int share_prefix = -1;    // [AG_SHARE_PREFIX]   contact-free prefix of look-ahead step 0: candidates whose tool has
                              //                     not touched the object yet follow ONE tool-free base rollout (-1: batches of >= 64
                              //                     candidates and >= 32768 rows, 0 never, 1 whenever possible).  Waits once per call
                              //                     for the contact plan (the GPU is busy with the base rollout meanwhile)
    int stream_min_rows = 32768;  // [AG_STREAM_MIN_ROWS] batches below this many rows (candidates x particles) stay on the caller's stream
                              //                     (r05 A/B, rope x 20 steps: 64 x 301 rows 9.99 ms on one stream / 11.3 on two,
                              //                     128 x 301 rows 14.2 / 13.4: the break-even lies between 19k and 38k rows)
    int pipeline_fork = 0;    // [AG_PIPELINE_FORK]  1: a call forks onto in-library streams even while a call issued on ANOTHER caller
                              //                     stream is still running (0: such a call stays on its stream - the caller is
                              //                     already spreading independent calls over streams, adaptigraph_amd/planner.py)
    int share_first = -1;     // [AG_SHARE_FIRST]    first forward of a dynamics() call: the relation encoder runs ONCE over the
                              //                     object-object edges of the start state's tool-free graph and every candidate reads
                              //                     those C rows (-1: batches of 8 candidates or more, 0 never, 1 whenever possible)
};

// ---- launchers (defined in the .hip files) ------------------------------------------------------------
struct EdgeArgs {
    const float* pos;           // (B,N,3) with `pos_bstride` floats between candidates
    long pos_bstride;
    const uint8_t* mask;        // (B,N)
    const uint8_t* tool;        // (B,N)
    const float* thr_vec;       // (B,) or null
    float thr;                  // also the chunk-culling radius: must satisfy thr*thr >= the squared threshold in use
    float thr2_override; int use_thr2;   // single-graph builder (graph.py:86,101)
    int B, N, topk, cta, edge_cap, slices;   // cta: 0 off, 1 batch rule (graph.py:276-286), 2 single-graph rule (:119-122)
    int* ell;                   // (B,N,min(topk,N)) scratch: kept senders per row (unused when topk >= N)
    int* deg;                   // (B,N) scratch
    int* slice_tot;             // (B,slices) scratch
    int* cta_flag;              // (B,) scratch: connect_tools_all batch flag
    int* recv; int* send;       // (B,edge_cap)
    int* row_ptr;               // (B,N+1)
    int* n_edges;               // (B,)
    int* overflow;              // single int: max edge count seen above max_nR (atomicMax), may be null
    int max_nR;
    int zero_on_overflow;       // internal rollout use: present an EMPTY graph downstream when E > edge_cap
    // Rollout fast path (top-k active, tool particles behind the object particles): the per-row sender lists ARE the
    // graph.  `ell` then points at the `send` array, a row owns `ell_stride` = topk + M slots (kept senders ascending,
    // then the tool senders of the connect_tools_all rule), a candidate `ell_bstride` = edge_cap slots; slot ids
    // replace CSR edge ids, `deg` replaces row_ptr, and k_edge_emit is skipped.  k_ell_index writes recv per slot,
    // the non-self-loop slot list (ns_edge, n_ns), n_edges, and applies the max_nR rule.
    int ell_full; int ell_stride; long ell_bstride;   // 0 / unset: ell_stride = min(topk,N), ell_bstride = N*ell_stride
    int* ns_edge; int* n_ns;
    int block_min_rows;         // Options::edge_block_min (-1: built-in threshold)
    const int* live;            // null, or device int: candidates [*live, B) of this launch have no forward left (device-planned
                                // rollout, RollPlan): their workgroups exit and their graphs are presented as empty
    // First forward of a dynamics() call (GraphBufs::send_pk): k_ell_index looks every object-object slot up in the BASE graph
    // (the start state's graph without the tool: base_send / base_deg, base_stride slots per row) and writes, per slot,
    // send_pk = sender | (position in the base row + 1) << 12 (0 in the upper bits: not in the base row); slots found there are
    // left out of ns_edge - their C row is the base table's.  share_stats (device, 2 x uint64, may be null): += slots served by
    // the base table, += slots this candidate encodes itself.
    int* send_pk; const int* base_send; const int* base_deg; int base_stride; int share_No;
    unsigned long long* share_stats;
    // with the contact-free prefix active only the candidates that start from the start state itself (start == 0) are at "the
    // first forward": share_start (per candidate, or null = all), share_cand (slot -> candidate, or null = share_b0 + slot)
    const int* share_start; const int* share_cand; int share_b0;
};
hipError_t launch_edge_build(const EdgeArgs& a, hipStream_t st, void (*mark)(void*, int, int), void* mark_ctx);
// list of non-self-loop edges per candidate (self-loop dedupe, see GraphBufs)
// one tool-attachment rule of the single-graph builder applied to a CSR edge list (ag_rules.hip)
struct RuleArgs {
    const float* pos; const uint8_t* mask; const uint8_t* tool; const uint8_t* subset;
    const int* send_in; const int* row_ptr_in;
    int N, n_tools, edge_cap, use_knn; double kNN;
    int* tlist; int* misc; float* pdis; uint8_t* keep; uint8_t* kept; int* deg;     // scratch
    int* recv; int* send; int* row_ptr; int* n_out;
};
hipError_t launch_tool_rule(const RuleArgs& a, hipStream_t st);
hipError_t launch_edge_nonself(const int* recv, const int* send, const int* row_ptr, int B, int N, int edge_cap,
                               int* ns_edge, int* n_ns, const int* live, hipStream_t st);

struct GraphBufs {
    // per-chunk activations; node rows = b*N + i, edge rows = b*c_cap + e, pitch NFP floats
    float* node_in;   // (B*N, NODE_IN)  [attr_obj, attr_tool, phys, act xyz, 1, 0]
    float* feat12;    // (B*N, F12)      [res0, res1, res2, cur] (model.py:156-166)
    float* group;     // (B*N, n_inst)   [p_instance ; 0]        (model.py:264)
    float* eff;       // particle effect, updated in place by the propagate chain
    float* P;
    float* UV[2][2];  // [parity][0 = U, 1 = V]: message-passing round r reads parity (r-1)&1 and writes parity r&1 (the gather
                      // is fused into the chain that rewrites U/V, so a launch must not read what it writes); k_node_enc
                      // writes parity 1, which round 0 reads
    float* C;         // (B*c_cap, NFP)  W1*rel_enc + b_rp
    const int* recv; const int* send; const int* row_ptr; const int* n_edges;
    const int* deg; int ell_stride;   // ell_stride > 0: slot-indexed graph of the rollout fast path (row i owns slots
                                      // [i*ell_stride, i*ell_stride + deg[i]) of send / C), row_ptr unused
    int B, N, n_p, n_inst;
    int edge_cap;     // pitch of recv/send per candidate
    int c_cap;        // pitch of C per candidate, multiple of 256
    // ---- class table (rollout only).  The particle-encoder chain depends only on [attrs, phys, action]: for an object
    // particle that row is the same for every candidate and every rollout step (action is zero for objects,
    // forward_dynamics.py:87-88; phys is broadcast, :151), for a tool particle it changes per look-ahead step.
    // So p_enc / P / U0 / V0 live in a small table: rows [0,N_o) valid object i, [N_o,2N_o) masked-out object i,
    // 2N_o + b*M + m tool m of candidate b.  cls_on = 0: plain per-(b,i) rows (ag_forward).
    int cls_on, N_o, M;
    const uint8_t* vmask;                  // (B,N) validity, selects the object variant
    float* c_node_in;                      // (2N_o + B*M, NODE_IN)
    float* c_eff; float* c_P; float* c_U; float* c_V;   // (2N_o + B*M, NFP)
    // ---- self-loop dedupe (rollout only).  A self-loop edge (i,i) has relation input [attrs_i, attrs_i, 0, 0..0]
    // (group and position differences of a particle with itself are exactly 0), so its C row is one of two
    // constants of the model: c_self[0] for an object particle (attrs 1,0), c_self[1] for a tool (attrs 0,1).
    // ns_edge lists the edges that are NOT self-loops; only those go through the relation encoder.
    const float* c_self;                   // (2, NFP) or null
    long self_row;                         // row of C where c_self[0..1] were copied (k_mp reads them from there)
    const int* ns_edge; const int* n_ns;   // (B,edge_cap), (B,) or null
    const float* wb3;                      // bf16x3 weight image: non-null selects the bf16x3 chains (ag_mlp.hip)
    const int* n_guard;                    // ag_forward: guarded per-candidate edge counts (k_edge_guard), else null
    // ---- ragged batches (masked rollouts, forward_dynamics.py:286-309: every candidate has its own number of valid
    // particles).  The propagate chains walk `rowlist` (dense rows b*N + i of the valid object particles and the tools,
    // ascending) instead of all B*N rows.  A masked-out particle takes part in no edge and its node input does not depend
    // on the candidate, so what the reference computes for it (model.py:338: pos + clamp(motion), a constant motion per
    // particle index) is computed ONCE, on the rows of a phantom candidate B (all particles masked out, no tools, no
    // edges) appended to the list; k_roll_update applies it to the masked-out rows of every real candidate.
    const int* rowlist; const int* n_rows; // (B*N + N_o,), device int; null = all rows
    // ---- first forward of a dynamics() call (forward_dynamics.py:25: ONE start state broadcast to all candidates, constant
    // history).  The relation input of an object-object edge is then the same in every candidate, so its C row is too: it is
    // encoded once per call into C_share (row = receiver * share_kb + position in the receiver's row of the tool-free base
    // graph), and the message passing of that forward reads send_pk (EdgeArgs) instead of send: sender in the low 12 bits,
    // position + 1 above them when the slot's C row is the base table's.  Null: every candidate's own C rows (all other forwards).
    const int* send_pk; const float* C_share; int share_kb;
    int n_his;                             // 4 (0 = 4), or 5 on the forward path (feature rows then have pitch F15_PITCH)
    int enc_persist, stagger_us, zigzag;   // Options of the owning context
    void* diag;                            // diagnostic build (-DAG_DIAG, ag_diag.hip) only: the context's probe state, else null
};
constexpr int B3_PHASE_BYTES = 2 * 5 * 3 * 64 * 16;   // 30,720
constexpr int B3_PHASES = 58;
inline long cls_rows(int N_o, int M, int B) { return 2L * N_o + (long)B * M; }
// row0/nrows select a slice of the class table when g.cls_on, else all B*N rows are encoded
hipError_t launch_node_enc(const float* wblob, const GraphBufs& g, long row0, long nrows, hipStream_t st);
hipError_t launch_edge_enc(const float* wblob, const GraphBufs& g, hipStream_t st);
// message-passing round `round` (gather fused into the chain); round 0: U/V/eff come from the class table (when g.cls_on)
hipError_t launch_node_prop(const float* wblob, const GraphBufs& g, int round, hipStream_t st);
hipError_t launch_node_final(const float* wblob, const GraphBufs& g, int round, float clamp, float* pred_pos,
                             float* pred_motion, hipStream_t st);

hipError_t launch_edge_guard(const int* n_edges, int B, int edge_cap, int* n_eff, int* overflow, hipStream_t st);
// model-input preparation for ag_forward: state (B,n_his,N,3) etc. -> node_in, feat12
hipError_t launch_prep(const float* state, const float* attrs, const float* action, const float* phys,
                       const GraphBufs& g, hipStream_t st);   // state has g.n_his frames

struct RollBufs {
    float* hist;        // (B, N_HIS, N, 3)
    float* pred;        // (B, N_o, 3) latest prediction
    float* motion;      // (B, N_o, 3) latest raw motion; ragged batches: row block B = the phantom candidate's motion, i.e.
                        // the constant motion of a masked-out particle per particle index
    int ragged;         // masked rollout with a work list (GraphBufs::rowlist)
    float clamp;        // model.py:86 motion clamp, for the masked-out rows
    uint8_t* mask;      // (B,N) valid particles incl. tools
    uint8_t* tool;      // (B,N)
};
struct RollArgs {
    int B, N_o, M, H, li, ai, y_mode, b0;  // B = slots this launch covers; b0 = first candidate of this chunk in the full batch
    int B_slots;                           // slots of the chunk (the phantom candidate of a ragged batch sits behind them)
    float grip; int grip_on; float phys;
    int write_obj_cls;                        // roll_init also writes the object rows of the class table
    const float* phys_vec;                    // null or (N_o,) per-particle physics parameter
    const float* state0; int state0_batched;  // (N_o,3) or (Bfull,N_o,3)
    const uint8_t* obj_mask;                  // (Bfull,N_o) or null
    const float* eef_xz; const float* eef_delta;  // (Bfull,H,M,2), (Bfull,H,M,3)
    const int* repeat;                        // device (Bfull,H)
    const int* live;                          // null, or device int: slots [*live, B) have no forward left (k_roll_update exits)
    const int* cand;                          // null, or (B,) device: slot b of this chunk holds candidate cand[b] of the full
                                              // batch (repeat-sorted launch order); null = candidate b0 + b
    float* state_seqs;                        // (Bfull,H,N_o,3)
    // ---- contact-free prefix (Options::share_prefix; look-ahead step 0 of dynamics() only).  Until a candidate's tool first
    // comes within the radius of an object particle its graph holds no tool edge, so its object particles evolve exactly as
    // in the tool-free BASE rollout of the start state, which is computed once per call: base_states (R+1, N_o, 3) = S_0
    // (the start state) .. S_R, base_y (R+1) = the tool height the reference derives from S_s (forward_dynamics.py:40,163).
    // start (Bfull,) = base step a candidate's own stepping starts from (its first contact is at forward start + 1); k_roll_init
    // then fills the slot's history with S_(start-3) .. S_start and the tool positions replayed up to there, and `repeat`
    // holds the forwards that are LEFT.  Null: every candidate starts from the start state.
    const float* base_states; const float* base_y; const int* start;
    // the base rollout itself: every step's prediction and tool height are recorded (null for ordinary candidates)
    float* all_states; float* all_y;
};
// Contact plan of the prefix sharing (ag_graph.hip: k_contact_plan).  Per candidate: replay the tool keypoints along the base
// rollout (x, z advance by fp32 adds exactly as k_roll_update advances them; y = base_y) and find the first forward whose graph
// would hold a tool-object pair inside the radius - the edge builder's own arithmetic (ag_edges.hip: dist_exact, (dis - thr^2) < 0),
// so "no contact" is exactly "no tool edge in either direction, whatever top-k keeps".
struct ContactPlan {
    const float* base_states; const float* base_y; int R;   // S_0..S_R, base_y[0..R]
    const float* eef_xz; const float* eef_delta; const int* repeat;   // (B,H,M,2), (B,H,M,3), (B,H): look-ahead step 0 is read
    int B, H, N_o, M; float thr;
    int* rep_eff;               // (B,H) out: forwards left per look-ahead step (step 0: repeat - first contact + 1, or 0; others: repeat)
    int* start;                 // (B,) out: base step the candidate starts from (0 when it never touches or touches at once)
    float* state_seqs;          // (B,H,N_o,3): candidates that never touch get S_repeat at look-ahead step 0 here
    // census mode (count != null; base_states = the start state, base_y = null, R = 1): nothing is planned or written except
    // count[0] += candidates with a forward to run whose tool touches at the FIRST forward, count[1] += candidates with a forward
    // to run, count[2] = max repeat of look-ahead step 0; the tool height of the start state is formed here (min object y +
    // gripper offset, forward_dynamics.py:40,80-81)
    int* count; float grip; int grip_on;
    // device-planned calls (ag_rollout_actions): the caller's bound of action_repeat.  A candidate beyond it is treated as the
    // plain device-planned path treats it - never captured (its rows stay zero; the shim marks them NaN) - and later look-ahead
    // steps are stepped at most R_bound times.  Host-planned calls: INT_MAX.
    int R_bound;
};
// count += number of 32-bit words in which a and b differ (bitwise)
hipError_t launch_count_diff(const float* a, const float* b, long n, int* count, hipStream_t st);
hipError_t launch_contact_plan(const ContactPlan& p, hipStream_t st);
// cost kernels (ag_cost.hip)
hipError_t launch_chamfer(const float* x, const float* y, const uint8_t* xm, const uint8_t* ym, int R, int N, int M,
                          int By, float* out, hipStream_t st);
hipError_t launch_state_stats(const float* state, int R, int N, const float* box4, float* out, hipStream_t st);
hipError_t launch_penalty(const float* state_pred, const float* action, const float* state_init, int B, int H, int N,
                          int kind, float ratio, float* out, hipStream_t st);
size_t chamfer_max_points();
hipError_t launch_reward(const float* error, const float* pen, const float* stats, const float* emax, const double* bbox4, int B,
                         int H, float* out, hipStream_t st);
hipError_t launch_cloth_combine(const float* raw, const float* dmax, long n, float* out, hipStream_t st);

// MPPI sampling / update (ag_mppi.hip)
hipError_t launch_mppi_sample(const float* act_seq, const float* lo, const float* hi, const float* rnd,
                              const float* scale, int S, int H, int mode, float pl, float* out, hipStream_t st);
hipError_t launch_mppi_update(const float* acts, const float* reward, const float* lo, const float* hi, int B, int H,
                              float rw, float pl, float* out, hipStream_t st);
hipError_t launch_mppi_clip(const float* in, const float* lo, const float* hi, float* out, long n, hipStream_t st);

// Device-side launch plan of a rollout whose actions are resident on the GPU (ag_rollout_actions): decode_action + tool
// keypoints (plan_utils.py:11-20, forward_dynamics.py:42-75) and the repeat-aware launch order, without the host ever seeing
// an action.  One wavefront per (launch chunk, look-ahead step): decodes the chunk's actions, orders its candidates by
// action_repeat (descending, stable) and tabulates for every step ai = 0..max_repeat how many of them are still live.
struct RollPlan {
    const float* action;        // (B,H,4) raw [x, z, theta, length]
    float push_length; int M; float tool_off[8];   // pusher_points[k][1] * sim_real_ratio, k < M (entry 0 unused)
    int B, H, Bc, N, max_repeat;
    float* decoded;             // (B,H,4) [x_start, z_start, x_end, z_end]        -> action_seqs
    float* eef_xz; float* eef_delta; int* repeat;  // (B,H,M,2), (B,H,M,3), (B,H)
    int* cand;                  // (H,B) slot -> candidate, per chunk segment
    int* live; int* rows;       // (n_chunks, H, max_repeat + 2): candidates live at step ai, and that times N
    int* sums;                  // (n_chunks, H, 2): sum of min(repeat, max_repeat), sum of live over the steps
    int* maxrep;                // (n_chunks, H): largest min(repeat, max_repeat) of the chunk = steps that find a live slot
    int* flags;                 // caller's flag words: [1] = atomicMax of a repeat beyond max_repeat
    int sort;                   // 0: keep the candidate order (live counts then stay at the chunk size while any candidate is live)
};
hipError_t launch_roll_plan(const RollPlan& p, hipStream_t st);

// work list of a ragged batch (see GraphBufs::rowlist): phantom rows (if any particle is masked out), then the valid rows of
// slots [0,B) in slot order; tab (B+1 ints): entries when only the first n slots are live; cand: slot -> candidate or null;
// also clears the phantom candidate's mask and degree rows
hipError_t launch_build_rowlist(const uint8_t* obj_mask, const int* cand, int b0, int B, int N_o, int M, int* rowlist, int* tab,
                                uint8_t* mask, int* deg, hipStream_t st);
// inputs of the shared first-forward edge encode (GraphBufs::C_share): the object rows of look-ahead step 0, as k_roll_init writes them
hipError_t launch_share_prep(const float* state0, int N_o, int n_his, float* node_in, float* feat, float* group, uint8_t* mask,
                             uint8_t* tool, hipStream_t st);
hipError_t launch_roll_init(const RollArgs& a, const RollBufs& r, const GraphBufs& g, hipStream_t st);
hipError_t launch_roll_update(const RollArgs& a, const RollBufs& r, const GraphBufs& g, hipStream_t st);

}  // namespace ag
