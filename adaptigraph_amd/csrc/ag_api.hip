// C-ABI of the MI355X-native GNN-dynamics rollout engine (see include/adaptigraph_amd.h).
// Host orchestration only: context, workspace, weight repacking, launch sequences.  No CPU compute fallback.
#include "../../include/adaptigraph_amd.h"
#include "ag_common.h"

#include <algorithm>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace ag {
size_t edge_build_max_particles();
int edge_ell_stride(int N, int topk);
size_t lat_weights_floats();
size_t lat_weights_offset(int which);
hipError_t launch_edge_enc_lat(const float* wl, const GraphBufs& g, hipStream_t st);
hipError_t launch_node_enc_lat(const float* wl, const GraphBufs& g, long row0, long nrows, hipStream_t st);
hipError_t launch_node_prop_lat(const float* wl, const GraphBufs& g, int round, bool last, float clamp, float* pred_pos,
                                float* pred_motion, hipStream_t st);
#ifdef AG_DIAG   // diagnostic build only (ag_diag.hip)
void* diag_create();
void diag_destroy(void* diag);
int diag_fail_at_chunk(void* diag);
int diag_timing_skip(void* diag);
#endif
}
using namespace ag;

namespace {

const char* kFamilyNames[FAM_COUNT] = {"edge_count", "edge_emit", "prep", "node_enc", "edge_enc",
                                       "mp", "node_prop", "node_final", "roll_init", "roll_update", "cost"};

struct Slab {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    template <typename T> T* take(size_t n) {
        used = (used + 255) & ~size_t(255);
        T* p = reinterpret_cast<T*>(base + used);
        used += n * sizeof(T);
        return p;
    }
};

struct ProfEvent { int fam; hipEvent_t e0, e1; };

}  // namespace

// what a kept base rollout of the prefix sharing (and a census verdict) is valid for; compared with memcmp, so always memset +
// field-wise filled + memcpy'd
struct BaseKey { int N_o, M, topk, cta, max_nR, n_his, precision, pstep, grip_on; float thr, grip, phys, clamp; const float* phys_vec;
                 unsigned long long weights_version; };

// Everything a call writes while it is in flight: workspace, launch plans, pinned read-back buffers, the events and streams of
// its fork / join.  A context keeps up to kMaxSlots of them, one per CALLER STREAM: calls issued on different streams then run
// side by side on the GPU (the planner's chunk loop, plan.py:241-247, is 40 independent calls on one start state;
// adaptigraph_amd/planner.py deals them to a few streams), calls on one stream stay ordered by the stream.  A stream that finds no
// free slot takes over the least recently used one after making itself wait for that slot's last call (an event recorded at the
// end of every call).  Created on first use, kept until ag_ctx_destroy: a call of a shape the slot has seen allocates nothing.
struct CallSlot {
    hipStream_t stream = nullptr; bool bound = false; unsigned long long tick = 0;
    Slab slab;
    int* d_repeat = nullptr; size_t repeat_cap = 0;   // device: [repeat (B*H) | launch order (H*B)]
    std::vector<int> h_repeat;   // slot-owned copy so the caller's array may die right after the call; same layout
    char* d_plan = nullptr; size_t plan_cap = 0;      // device-planned rollouts (ag_rollout_actions): decoded tool keypoints,
                                                      // repeats, launch order and per-step live counts
    int* h_rep_pin = nullptr; size_t rep_pin_cap = 0;     // pinned: [forwards left | action_repeat | flag, census x4] of a prefix-sharing call
    int* h_plan_max = nullptr; size_t plan_max_cap = 0;   // pinned host copy of RollPlan::maxrep of the call being enqueued
    int* h_census = nullptr;                            // pinned (8 ints): result of a census nobody waited for (see Decision)
    hipEvent_t ev_plan = nullptr;                       // fires when a read-back of this call has landed
    hipEvent_t ev_census = nullptr; bool census_pending = false;   // a census went out on this slot's stream that nobody waited for
    BaseKey census_key{}; int census_B = 0, census_H = 0, census_R = 0;
    hipEvent_t ev_done = nullptr; bool have_done = false;   // end of the slot's last call
    float* d_work = nullptr; size_t work_cap = 0;   // ag_rollout_work: scratch for the plan kernel's other outputs
    int* d_words = nullptr;      // 64 ints: [0] overflow word of the synchronous entry points, [8..11] census counters
    unsigned long long* d_share_stats = nullptr;      // shared first forward: [0] slots served by the base table, [1] slots encoded per candidate
    static constexpr int kMaxStreams = 4;
    hipStream_t aux_stream[kMaxStreams] = {nullptr, nullptr, nullptr, nullptr};   // [0] unused: the caller's stream
    hipEvent_t ev_fork = nullptr, ev_join[kMaxStreams] = {nullptr, nullptr, nullptr, nullptr};
};

struct ag_ctx {
    int device = 0;
    ag_dims dims{};
    std::string err;
    float* d_w = nullptr;
    float* d_wb3 = nullptr;      // bf16x3 weight image (58 phases of 30,720 B)
    float* d_wlat = nullptr;     // weight image of the latency-mode chains (ag_lat.hip), n_his = 4 models only
    int precision = 0;           // 0: exact fp32 MFMA (default), 1: bf16x3 split on the bf16 matrix pipe
    bool have_w = false;
    int chunk = 0;
    Options opt;                 // per-context switches: environment defaults read once at create, ag_ctx_set_option afterwards
    void* diag = nullptr;        // diagnostic build only: probe state of this context (ag_diag.hip)
    static constexpr int kMaxSlots = 8;
    static constexpr int kMaxStreams = CallSlot::kMaxStreams;
    CallSlot slots[kMaxSlots];
    unsigned long long slot_tick = 0;
    int last_slot = 0;           // slot of the last rollout call (the diagnostics below refer to it)
    long long n_allocs = 0;      // hipMalloc / hipHostMalloc / hipFree / hipHostFree / event and stream creations so far (ag_ctx_alloc_counts)
    long long fwd_executed = 0, fwd_needed = 0;       // candidate-forwards of the last rollout call (ag_ctx_rollout_counts)
    int* d_plan_sums = nullptr; int plan_sums_n = 0;  // device-planned call: sums pending a read-back
    // base rollout of the prefix sharing, kept across calls: the reference's planner calls dynamics() 40 times per planner call
    // with one start state (plan.py:241-247).  Valid for (start state bit-equal, same model / task scalars); [states | heights].
    // Shared by all slots: host-side validity (base_cache_R) is set only after the producing call has waited for its contact plan,
    // i.e. with the contents complete; a call that overwrites it first makes its stream wait for every other slot's last call.
    float* d_base_cache = nullptr; size_t base_cache_cap = 0; int base_cache_R = -1, base_cache_capR = 0;
    BaseKey base_key{};
    // the automatic mode's last census verdict "not worth a base rollout" (bench-like batches: every push starts on the object),
    // for batches of the same key and shape: such a call skips the blocking census, enqueues one that nobody waits for, and the
    // verdict is revisited when that one has landed (see rollout_impl).  A stale verdict costs time, never a result.
    struct Decision { bool decline = false; BaseKey key{}; int B = 0, H = 0; } decision;
    unsigned long long weights_version = 0;
    long long steps_enqueued = 0, steps_bound = 0;      // model forwards (per chunk) enqueued by the last rollout call / what the bound alone gives
    const int* d_share_nns = nullptr;                 // edges the base encode ran over (workspace of the last rollout call), or null
    float* d_cself = nullptr;    // (256, NFP): rows 0/1 = C of an object / tool self-loop edge (see GraphBufs)
    // in-library streams of a call: alternate chunks run on them so that the HBM-bound kernels of one chunk overlap the
    // MFMA-bound chains of the other (fork/join with events around every rollout call)
    int n_streams = 2;
    // profiling
    unsigned prof_mask = 0;
    std::vector<ProfEvent> prof_live;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[FAM_COUNT] = {0};
    long long prof_n[FAM_COUNT] = {0};
    hipStream_t prof_stream = nullptr;
};

namespace {

struct OptName { const char* name; const char* env; int Options::* field; bool env_negates; int lo, hi; };
const OptName kOptions[] = {
    {"streams", "AG_STREAMS", &Options::streams, false, 0, ag_ctx::kMaxStreams},
    {"chunk", "AG_CHUNK", &Options::chunk, false, 0, 1 << 20},
    {"latency", "AG_LATENCY", &Options::latency, false, -1, 1},
    {"ragged", "AG_NO_RAGGED", &Options::ragged, true, 0, 1},
    {"ell_graph", "AG_NO_ELL_GRAPH", &Options::ell_graph, true, 0, 1},
    {"self_dedupe", "AG_NO_SELF_DEDUPE", &Options::self_dedupe, true, 0, 1},
    {"repeat_sort", "AG_NO_REPEAT_SORT", &Options::repeat_sort, true, 0, 1},
    {"edge_wgs", "AG_EDGE_WGS", &Options::edge_wgs, false, 1, 1 << 16},
    {"edge_block_min", "AG_EDGE_BLOCK_MIN", &Options::edge_block_min, false, -1, 0x7fffffff},
    {"enc_persist", "AG_ENC_PERSIST", &Options::enc_persist, false, 0, 1 << 20},
    {"stagger_us", "AG_STAGGER_US", &Options::stagger_us, false, 0, 1000},
    {"device_decode", "AG_DEVICE_DECODE", &Options::device_decode, false, -1, 1},
    {"zigzag", "AG_ZIGZAG", &Options::zigzag, false, 0, 1},
    {"share_first", "AG_SHARE_FIRST", &Options::share_first, false, -1, 1},
    {"share_prefix", "AG_SHARE_PREFIX", &Options::share_prefix, false, -1, 1},
    {"stream_min_rows", "AG_STREAM_MIN_ROWS", &Options::stream_min_rows, false, 0, 0x7fffffff},
    {"pipeline_fork", "AG_PIPELINE_FORK", &Options::pipeline_fork, false, 0, 1},
};
void options_from_env(Options& o) {   // values from the environment are clamped into the option's range
    for (const OptName& n : kOptions)
        if (const char* e = getenv(n.env)) o.*(n.field) = n.env_negates ? (atoi(e) ? 0 : 1) : std::min(n.hi, std::max(n.lo, atoi(e)));
}

int fail(ag_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}
#define HIPCHK(c, expr)                                                                                   \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) return fail(c, AG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// every allocation / creation the library makes is counted (ag_ctx_alloc_counts): a steady-state call makes none
hipError_t dev_alloc(ag_ctx* c, void** p, size_t bytes) { ++c->n_allocs; return hipMalloc(p, bytes); }
hipError_t dev_free(ag_ctx* c, void* p) { ++c->n_allocs; return hipFree(p); }
hipError_t pin_alloc(ag_ctx* c, void** p, size_t bytes) { ++c->n_allocs; return hipHostMalloc(p, bytes, hipHostMallocDefault); }
hipError_t pin_free(ag_ctx* c, void* p) { ++c->n_allocs; return hipHostFree(p); }
hipError_t event_new(ag_ctx* c, hipEvent_t* e) { ++c->n_allocs; return hipEventCreateWithFlags(e, hipEventDisableTiming); }
hipError_t stream_new(ag_ctx* c, hipStream_t* s) { ++c->n_allocs; return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }

// The slot of caller stream `st` (see CallSlot).  capturing: the call is being recorded into a hipGraph - it may neither wait for
// nor record an event that lives outside the graph.
int slot_acquire(ag_ctx* c, hipStream_t st, bool capturing, CallSlot** out) {
    CallSlot* s = nullptr;
    for (CallSlot& k : c->slots) if (k.bound && k.stream == st) { s = &k; break; }
    if (!s) for (CallSlot& k : c->slots) if (!k.bound) { s = &k; break; }
    if (!s) {   // every slot belongs to another stream: take the least recently used one, after its last call
        s = &c->slots[0];
        for (CallSlot& k : c->slots) if (k.tick < s->tick) s = &k;
        if (!capturing && s->have_done) HIPCHK(c, hipStreamWaitEvent(st, s->ev_done, 0));
        s->census_pending = false;
    }
    if (!s->ev_done) {   // first use: everything whose size does not depend on the call
        HIPCHK(c, event_new(c, &s->ev_done));
        HIPCHK(c, event_new(c, &s->ev_plan));
        HIPCHK(c, event_new(c, &s->ev_census));
        HIPCHK(c, event_new(c, &s->ev_fork));
        HIPCHK(c, dev_alloc(c, reinterpret_cast<void**>(&s->d_words), 256));
        HIPCHK(c, dev_alloc(c, reinterpret_cast<void**>(&s->d_share_stats), 256));
        HIPCHK(c, hipMemset(s->d_share_stats, 0, 256));
        HIPCHK(c, pin_alloc(c, reinterpret_cast<void**>(&s->h_census), 64));
    }
    s->bound = true; s->stream = st; s->tick = ++c->slot_tick;
    *out = s;
    return AG_OK;
}
// end of a call that used the slot: later calls on OTHER streams that take the slot over wait for this point
void slot_release(CallSlot* s, hipStream_t st, bool capturing) {
    if (capturing || !s || !s->ev_done) return;
    if (hipEventRecord(s->ev_done, st) == hipSuccess) s->have_done = true;
}
// Records the slot's end-of-call event on EVERY exit of the call that acquired it (r06): an early `return rc` after work was
// enqueued used to leave ev_done marking an EARLIER call, so a later take-over of the slot by another stream (slot_acquire's LRU
// path, the wait-for-all-slots before d_base_cache is replaced) would not have waited for what the failed call had enqueued.
struct SlotGuard {
    CallSlot* s; hipStream_t st; bool capturing;
    SlotGuard(CallSlot* s_, hipStream_t st_, bool cap_) : s(s_), st(st_), capturing(cap_) {}
    SlotGuard(const SlotGuard&) = delete;
    SlotGuard& operator=(const SlotGuard&) = delete;
    ~SlotGuard() { slot_release(s, st, capturing); }
};
// is a call of another slot still running on the GPU?  (then this caller is pipelining calls over streams)
bool other_slot_busy(ag_ctx* c, const CallSlot* me) {
    bool busy = false;
    for (CallSlot& k : c->slots)
        if (&k != me && k.bound && k.have_done) {
            if (hipEventQuery(k.ev_done) == hipErrorNotReady) busy = true;
            (void)hipGetLastError();
        }
    return busy;
}
void slot_destroy(ag_ctx* c, CallSlot& s) {
    for (hipEvent_t e : {s.ev_plan, s.ev_census, s.ev_done, s.ev_fork}) if (e) (void)hipEventDestroy(e);
    for (int i = 1; i < CallSlot::kMaxStreams; ++i) {
        if (s.ev_join[i]) (void)hipEventDestroy(s.ev_join[i]);
        if (s.aux_stream[i]) (void)hipStreamDestroy(s.aux_stream[i]);
    }
    if (s.h_plan_max) (void)hipHostFree(s.h_plan_max);
    if (s.h_rep_pin) (void)hipHostFree(s.h_rep_pin);
    if (s.h_census) (void)hipHostFree(s.h_census);
    if (s.d_words) (void)hipFree(s.d_words);
    if (s.d_share_stats) (void)hipFree(s.d_share_stats);
    if (s.d_repeat) (void)hipFree(s.d_repeat);
    if (s.d_plan) (void)hipFree(s.d_plan);
    if (s.d_work) (void)hipFree(s.d_work);
    if (s.slab.base) (void)hipFree(s.slab.base);
    s = CallSlot();
}

void prof_mark(void* vc, int fam, int phase) {
    ag_ctx* c = static_cast<ag_ctx*>(vc);
    if (!(c->prof_mask & (1u << fam))) return;
    auto get = [&]() {
        hipEvent_t e;
        if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    if (phase == 0) {
        ProfEvent p{fam, get(), get()};
        (void)hipEventRecord(p.e0, c->prof_stream);
        c->prof_live.push_back(p);
    } else {
        for (auto it = c->prof_live.rbegin(); it != c->prof_live.rend(); ++it)
            if (it->fam == fam) { (void)hipEventRecord(it->e1, c->prof_stream); break; }
    }
}
struct Scoped {
    ag_ctx* c; int fam;
    Scoped(ag_ctx* c_, int f) : c(c_), fam(f) { prof_mark(c, fam, 0); }
    ~Scoped() { prof_mark(c, fam, 1); }
};

// ---------------------------------------------------------------------------------------------- weight packing
// MFMA A-operand image of a layer: [chunk q][m-block][lane][4 steps]; lane l supplies out-feature 32*mb + (l&31)
// for input slot k(s, l>>5).  See ag_mlp.hip header.
int slot_of(int s, int h) {
    const int t = s < 64 ? s / 16 : 4, r = s < 64 ? s % 16 : s - 64;
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
}
void pack_layer(float* dst, const float* W, int ld, int col0, int out_dim, int in_dim, const float* bias, int MB) {
    for (int q = 0; q < KCH; ++q)
        for (int mb = 0; mb < MB; ++mb)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int k = slot_of(4 * q + e, lane >> 5), m = 32 * mb + (lane & 31);
                    float v = 0.f;
                    if (m < out_dim) {
                        if (k < in_dim) v = W[(size_t)m * ld + col0 + k];
                        else if (k == ONE_F && bias) v = bias[m];
                    }
                    dst[((size_t)(q * MB + mb) * 64 + lane) * 4 + e] = v;
                }
}
void pack_first(float* dst, const float* W, int in_dim, const float* bias, int nch) {
    for (int q = 0; q < nch; ++q)
        for (int mb = 0; mb < 5; ++mb)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int k = 2 * (4 * q + e) + (lane >> 5), m = 32 * mb + (lane & 31);
                    float v = 0.f;
                    if (m < NF) {
                        if (k < in_dim) v = W[(size_t)m * in_dim + k];
                        else if (k == in_dim) v = bias[m];
                    }
                    dst[((size_t)(q * 5 + mb) * 64 + lane) * 4 + e] = v;
                }
}

// ---- latency-mode chains (ag_lat.hip): A-operand image of v_mfma_f32_16x16x4_f32, [chunk of 4 k-steps][tile][lane][4].
// Register r of tile T in lane group g stands for feature 16T + 8(r>>1) + 4(g&1) + 2(r&1) + (g>>1): the k sequence of the
// 32-row chains (slot_of) cut into steps of four, so that both kernel families round identically.  >= 152: dead slot.
int feat_lat(int T, int g, int r) {
    const int f = 16 * T + 8 * (r >> 1) + 4 * (g & 1) + 2 * (r & 1) + (g >> 1);
    return f < 152 ? f : -1;
}
void pack_layer_lat(float* dst, const float* W, int ld, int col0, int out_dim, int in_dim, const float* bias, bool head) {
    const int ntile = head ? 1 : 10;
    for (int c = 0; c < 10; ++c)
        for (int mt = 0; mt < ntile; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int s = 4 * c + e;
                    float v = 0.f;
                    if (s < 38) {
                        const int T = s < 36 ? s / 4 : 9, r = s < 36 ? s % 4 : s - 36;
                        const int k = feat_lat(T, lane >> 4, r), i = lane & 15;
                        // D row 4g + r of an output tile = A row i: the head keeps its 3 outputs in rows 0..2
                        const int m = head ? i : feat_lat(mt, i >> 2, i & 3);
                        if (m >= 0 && m < out_dim && k >= 0) {
                            if (k < in_dim) v = W[(size_t)m * ld + col0 + k];
                            else if (k == ONE_F && bias) v = bias[m];
                        }
                    }
                    dst[((size_t)(c * ntile + mt) * 64 + lane) * 4 + e] = v;
                }
}
void pack_first_lat(float* dst, const float* W, int in_dim, const float* bias, int nchunks) {
    for (int c = 0; c < nchunks; ++c)
        for (int mt = 0; mt < 10; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * (4 * c + e) + (lane >> 4), i = lane & 15, m = feat_lat(mt, i >> 2, i & 3);
                    float v = 0.f;
                    if (m >= 0 && m < NF) {
                        if (k < in_dim) v = W[(size_t)m * in_dim + k];
                        else if (k == in_dim) v = bias[m];
                    }
                    dst[((size_t)(c * 10 + mt) * 64 + lane) * 4 + e] = v;
                }
}

// ---- bf16x3 weight image (see ag_mlp.hip): every weight is split exactly into three bf16 pieces
uint16_t bf16_rn(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
void split3(float w, uint16_t out[3]) {
    out[0] = bf16_rn(w);
    const float r = w - bf16_f(out[0]);
    out[1] = bf16_rn(r);
    const float q = r - bf16_f(out[1]);
    out[2] = bf16_rn(q);
}
// image index (uint16 units) of element j of lane `lane`, part `part`, m-block mb, k-step ks (MB m-blocks per k-step)
size_t b3_idx(int ks, int MB, int mb, int part, int lane, int j) { return ((((size_t)ks * MB + mb) * 3 + part) * 64 + lane) * 8 + j; }
void pack_layer_b3(uint16_t* dst, const float* W, int ld, int col0, int out_dim, int in_dim, const float* bias, int MB) {
    for (int ks = 0; ks < 10; ++ks)                        // k-step ks = 2*tile + u
        for (int mb = 0; mb < MB; ++mb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int t = ks >> 1, u = ks & 1, h = lane >> 5;
                    const int k = 32 * t + 16 * u + (j & 3) + 8 * (j >> 2) + 4 * h, m = 32 * mb + (lane & 31);
                    float v = 0.f;
                    if (m < out_dim) {
                        if (k < in_dim) v = W[(size_t)m * ld + col0 + k];
                        else if (k == ONE_F && bias) v = bias[m];
                    }
                    uint16_t p3[3];
                    split3(v, p3);
                    for (int part = 0; part < 3; ++part) dst[b3_idx(ks, MB, mb, part, lane, j)] = p3[part];
                }
}
void pack_first_b3(uint16_t* dst, const float* W, int in_dim, const float* bias) {
    for (int ks = 0; ks < 2; ++ks)
        for (int mb = 0; mb < 5; ++mb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * ks + 8 * (lane >> 5) + j, m = 32 * mb + (lane & 31);
                    float v = 0.f;
                    if (m < NF) {
                        if (k < in_dim) v = W[(size_t)m * in_dim + k];
                        else if (k == in_dim) v = bias[m];
                    }
                    uint16_t p3[3];
                    split3(v, p3);
                    for (int part = 0; part < 3; ++part) dst[b3_idx(ks, 5, mb, part, lane, j)] = p3[part];
                }
}

// ---------------------------------------------------------------------------------------------- workspace
struct Work {
    GraphBufs g{};
    RollBufs r{};
    int* ell; int* deg; int* slice_tot; int* cta_flag;
    int* recv; int* send; int* row_ptr; int* n_edges;
    int* ns_edge; int* n_ns;
    int* send_pk;                // first forward of a dynamics() call (GraphBufs::send_pk)
    int* rowlist; int* n_rows;   // ragged batches (GraphBufs::rowlist)
};

size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

int ensure_slab(ag_ctx* c, CallSlot& sl, size_t bytes) {
    Slab& slab = sl.slab;
    if (slab.cap >= bytes) { slab.used = 0; return AG_OK; }
    if (slab.base) HIPCHK(c, dev_free(c, slab.base));
    slab.base = nullptr; slab.cap = 0;
    const size_t want = round_up(bytes + (bytes >> 3), 1 << 20);
    HIPCHK(c, dev_alloc(c, reinterpret_cast<void**>(&slab.base), want));
    slab.cap = want; slab.used = 0;
    return AG_OK;
}

// bytes of one workspace for Bc candidates.  own_edges: edge index arrays + builder scratch; roll: rollout state
size_t work_bytes(int Bc, int N, int n_inst, int edge_cap, int c_cap, int slices, bool own_edges, bool roll,
                  bool own_group, int N_o, int ell_stride) {
    const size_t rows = (size_t)Bc * N;
    size_t bytes = 16 * 256;
    bytes += rows * (NODE_IN + F15_PITCH + (own_group ? n_inst : 0)) * 4 + 6 * rows * NFP * 4 + ((size_t)Bc * c_cap + 256) * NFP * 4;
    if (own_edges) bytes += rows * (size_t)(ell_stride + 1) * 4 + (size_t)Bc * (slices + 3) * 4 + 3 * (size_t)Bc * edge_cap * 4 + (size_t)Bc * (N + 1) * 4;
    if (roll) bytes += (size_t)Bc * edge_cap * 4 + rows * 4 + 1024 + (size_t)Bc * 4 + (size_t)Bc * N_HIS_MAX * N * 3 * 4 + 2 * (size_t)Bc * N_o * 3 * 4 + 2 * rows +
                       (size_t)cls_rows(N_o, N - N_o, Bc) * (NODE_IN + 4 * NFP) * 4;
    return bytes + 64 * 256;
}

// carve one workspace from the slab (which must already be large enough; see work_bytes)
int carve_work(ag_ctx* c, Slab& s, Work& w, int Bc, int N, int n_inst, int edge_cap, int c_cap, int slices, bool own_edges,
               bool roll, bool own_group, int N_o, int ell_stride) {
    const size_t rows = (size_t)Bc * N;
    w.g.node_in = s.take<float>(rows * NODE_IN);
    w.g.feat12 = s.take<float>(rows * F15_PITCH);            // pitch 12 (n_his 4) or 16 (n_his 5, forward path)
    w.g.group = own_group ? s.take<float>(rows * n_inst) : nullptr;
    w.g.eff = s.take<float>(rows * NFP);
    w.g.P = s.take<float>(rows * NFP);
    for (int par = 0; par < 2; ++par)
        for (int k = 0; k < 2; ++k) w.g.UV[par][k] = s.take<float>(rows * NFP);
    w.g.C = s.take<float>(((size_t)Bc * c_cap + 256) * NFP);   // + room for the two self-loop constant rows
    w.g.B = Bc; w.g.N = N; w.g.n_inst = n_inst; w.g.edge_cap = edge_cap; w.g.c_cap = c_cap; w.g.n_p = N_o;
    w.g.enc_persist = c->opt.enc_persist; w.g.stagger_us = c->opt.stagger_us; w.g.zigzag = c->opt.zigzag; w.g.diag = c->diag;
    if (own_edges) {
        w.ell = s.take<int>(rows * (size_t)std::max(1, ell_stride));
        w.deg = s.take<int>(rows);
        w.slice_tot = s.take<int>((size_t)Bc * slices);
        w.cta_flag = s.take<int>(Bc);
        w.recv = s.take<int>((size_t)Bc * edge_cap);
        w.send = s.take<int>((size_t)Bc * edge_cap);
        w.row_ptr = s.take<int>((size_t)Bc * (N + 1));
        w.n_edges = s.take<int>(Bc);
        w.ns_edge = s.take<int>((size_t)Bc * edge_cap);
        w.n_ns = s.take<int>(Bc);
        w.g.recv = w.recv; w.g.send = w.send; w.g.row_ptr = w.row_ptr; w.g.n_edges = w.n_edges;
    }
    if (roll) {
        w.send_pk = s.take<int>((size_t)Bc * edge_cap);
        w.rowlist = s.take<int>(rows);
        w.n_rows = s.take<int>((size_t)Bc + 64);            // ragged batches: row count per number of live slots (k_build_rowlist)
        w.r.hist = s.take<float>((size_t)Bc * N_HIS_MAX * N * 3);   // (Bc, n_his, N, 3) with the model's n_his (4 or 5)
        w.r.pred = s.take<float>((size_t)Bc * N_o * 3);
        w.r.motion = s.take<float>((size_t)Bc * N_o * 3);
        w.r.mask = s.take<uint8_t>(rows);
        w.r.tool = s.take<uint8_t>(rows);
        const size_t cr = (size_t)cls_rows(N_o, N - N_o, Bc);
        w.g.cls_on = 1; w.g.N_o = N_o; w.g.M = N - N_o; w.g.vmask = w.r.mask;
        w.g.c_node_in = s.take<float>(cr * NODE_IN);
        w.g.c_eff = s.take<float>(cr * NFP);
        w.g.c_P = s.take<float>(cr * NFP);
        w.g.c_U = s.take<float>(cr * NFP);
        w.g.c_V = s.take<float>(cr * NFP);
    }
    if (s.used > s.cap) return fail(c, AG_ERR_INVALID, "internal: workspace carve overflow");
    return AG_OK;
}

int pick_slices(const ag_ctx* c, int B, int N) {
    // one sixteen-wave workgroup per CU: every workgroup re-reads its candidate's positions and re-derives the chunk
    // boxes, so fewer, longer row slices win (cloth, 128 candidates: 128 workgroups 56.8 ms per rollout, 256: 31.0,
    // 384: 43.1, 512: 35.2, 1024: 41.9)
    const int target = std::max(1, c->opt.edge_wgs);
    int s = (target + B - 1) / B;
    // a slice is at least 16 rows (one per wavefront of the workgroup): small batches are latency-bound, so a single
    // graph is spread over as many workgroups as that allows (one rope graph: 4 -> 18 workgroups, 43 -> 13 us per launch)
    s = std::min(s, std::max(1, N / 16));
    // (r06: up to 128 slices - one cloth-sized graph alone was cut into 64 slices of 32 rows, two rows per wavefront on a quarter
    // of the chip; 127 slices of 16 rows give every wavefront one row.  Which rows share a workgroup never changes a row's result.)
    return std::max(1, std::min(s, 128));
}

// the gather (ag_mlp.hip: gather_agg) addresses C, U and V with 32-bit element offsets: a launch chunk must keep every buffer below 2^32 floats
int clamp_chunk_for_offsets(int Bc, int N, int c_cap) {
    const long max_rows = ((1L << 32) / NFP) - 512;          // rows of NFP floats addressable with a 32-bit element offset
    const long by_c = max_rows / std::max(1, c_cap);
    const long by_n = max_rows / std::max(1, N);
    return (int)std::max(1L, std::min<long>(Bc, std::min(by_c, by_n)));
}

int auto_chunk(const ag_ctx* c, int B, int N) {
    if (c->chunk > 0) return std::min(c->chunk, B);
    if (c->opt.chunk > 0) return std::min(c->opt.chunk, B);
    // Node chains run 128-row workgroups, two per CU: the largest chunk whose workgroup count is <= 4 rounds of 512.
    // (Measured on the 1024 x 2026 cloth batch: 64 candidates/launch 574 ms, 96: 564, 128: 559, 192: 561, 256: 558 -
    // more rounds per launch dilute the lockstep store bursts and the launch tails; the workspace grows with it.)
    const long max_rows = 4L * 256 * 256;
    long bc = max_rows / N;
    return (int)std::max(1L, std::min<long>(bc, B));
}

// Small launches are latency-bound: below one chip-filling round of 128-row workgroups the latency-mode chains take over
// (ag_lat.hip: 32-row workgroups, every layer split over the four wavefronts; bit-identical results).  Options::latency:
// 0 never, 1 always, -1 = by size.  (Thresholds in 128-row workgroups of the throughput kernels: a latency workgroup reads
// its weight fragments from L2 itself - 200 KB per layer - so beyond about one latency workgroup per CU the L2 traffic eats
// the gain: rope 64 x 301 rows = 151 workgroups runs the same 71 us either way, one rope graph 66 -> 31 us.)
bool lat_available(const ag_ctx* c, const GraphBufs& g) { return c->d_wlat && !g.wb3 && g.n_his != 5; }
bool lat_edge_for(const ag_ctx* c, const GraphBufs& g) {
    const long edge_wgs = (long)g.B * g.c_cap / 128;
    return lat_available(c, g) && (c->opt.latency >= 0 ? c->opt.latency == 1 : edge_wgs <= 128);
}
// particle-encoder chain (class table: 2 N_o + B M rows per look-ahead step): the latency-mode kernel while its grid of 32-row
// workgroups fits one round of the chip (two per CU); beyond that the 128-row throughput kernel
hipError_t node_enc_for(const ag_ctx* c, const GraphBufs& g, long row0, long nrows, hipStream_t st) {
    const long rows = g.cls_on ? nrows : (long)g.B * g.N;
    const bool lat = lat_available(c, g) && (c->opt.latency >= 0 ? c->opt.latency == 1 : rows <= 512L * 32);
    return lat ? launch_node_enc_lat(c->d_wlat, g, row0, nrows, st) : launch_node_enc(c->d_w, g, row0, nrows, st);
}
bool lat_node_for(const ag_ctx* c, const GraphBufs& g) {
    const long node_wgs = ((long)g.B * g.N + 127) / 128;
    return lat_available(c, g) && (c->opt.latency >= 0 ? c->opt.latency == 1 : node_wgs <= 64);
}
// relation encoder + W1 over the graph's (non-self-loop) edges -> C
int run_edge_chain(ag_ctx* c, const GraphBufs& g, hipStream_t st) {
    Scoped p(c, FAM_EDGE_ENC);
    if (lat_edge_for(c, g)) HIPCHK(c, launch_edge_enc_lat(c->d_wlat, g, st));
    else HIPCHK(c, launch_edge_enc(c->d_w, g, st));
    return AG_OK;
}

// one model forward on a prepared workspace (node_in, feat12, group, edges all set).  With g.cls_on the particle
// encoder outputs already sit in the class table (encoded at look-ahead-step start) and k_node_enc is skipped.
int run_model(ag_ctx* c, const GraphBufs& g, float* pred_pos, float* pred_motion, hipStream_t st) {
    if (!g.cls_on) { Scoped p(c, FAM_NODE_ENC); HIPCHK(c, node_enc_for(c, g, 0, (long)g.B * g.N, st)); }
    const bool lat_node = lat_node_for(c, g);
    if (g.send_pk && lat_node) return fail(c, AG_ERR_INVALID, "internal: shared first forward on the latency-mode chains");
    int rc = run_edge_chain(c, g, st);
    if (rc) return rc;
    for (int ps = 0; ps < c->dims.pstep; ++ps) {
        const bool last = ps + 1 == c->dims.pstep;
        Scoped p(c, last ? FAM_NODE_FINAL : FAM_NODE_PROP);
        if (lat_node) HIPCHK(c, launch_node_prop_lat(c->d_wlat, g, ps, last, c->dims.motion_clamp, pred_pos, pred_motion, st));
        else if (!last) HIPCHK(c, launch_node_prop(c->d_w, g, ps, st));
        else HIPCHK(c, launch_node_final(c->d_w, g, ps, c->dims.motion_clamp, pred_pos, pred_motion, st));
    }
    return AG_OK;
}

// C rows of the two kinds of self-loop edge (object: attrs 1,0; tool: attrs 0,1), through the real edge chain of the
// ACTIVE precision mode on a 2-particle, 2-edge graph {(0,0),(1,1)} - bitwise what k_edge_enc produces for such edges.
int compute_self_rows(ag_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_cself) HIPCHK(c, dev_alloc(c, reinterpret_cast<void**>(&c->d_cself), 256 * NFP * 4));
    struct Mini { float node_in[2 * NODE_IN]; float feat12[2 * F15_PITCH]; float group[2]; int recv[2]; int send[2]; int n_edges; int pad; } h{};
    h.node_in[0] = 1.f; h.node_in[6] = 1.f;                         // object particle
    h.node_in[NODE_IN + 1] = 1.f; h.node_in[NODE_IN + 6] = 1.f;     // tool particle
    h.group[0] = 1.f;
    h.recv[0] = 0; h.recv[1] = 1; h.send[0] = 0; h.send[1] = 1; h.n_edges = 2;
    char* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), sizeof(Mini)));
    HIPCHK(c, hipMemcpy(d, &h, sizeof(Mini), hipMemcpyHostToDevice));
    GraphBufs g{};
    g.n_his = c->dims.n_his;
    g.node_in = reinterpret_cast<float*>(d + offsetof(Mini, node_in));
    g.feat12 = reinterpret_cast<float*>(d + offsetof(Mini, feat12));
    g.group = reinterpret_cast<float*>(d + offsetof(Mini, group));
    g.recv = reinterpret_cast<int*>(d + offsetof(Mini, recv));
    g.send = reinterpret_cast<int*>(d + offsetof(Mini, send));
    g.n_edges = reinterpret_cast<int*>(d + offsetof(Mini, n_edges));
    g.C = c->d_cself; g.B = 1; g.N = 2; g.n_p = 1; g.n_inst = 1; g.edge_cap = 2; g.c_cap = 256;
    g.wb3 = c->precision == 1 ? c->d_wb3 : nullptr;
    hipError_t e = launch_edge_enc(c->d_w, g, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, AG_ERR_HIP, "self-loop C rows: %s", hipGetErrorString(e));
    return AG_OK;
}

int check_topk(ag_ctx* c, int N, int topk) {
    if (N < 1 || topk < 1) return fail(c, AG_ERR_INVALID, "N and topk must be >= 1");
    if ((size_t)N > edge_build_max_particles()) return fail(c, AG_ERR_UNSUPPORTED, "N=%d exceeds the LDS-resident edge builder limit %zu", N, edge_build_max_particles());
    if (topk < N && topk > 128) return fail(c, AG_ERR_UNSUPPORTED, "topk=%d: 128 < topk < N is not implemented", topk);
    return AG_OK;
}

}  // namespace

// ================================================================================================ C-ABI
extern "C" {

uint32_t ag_abi_version(void) { return AG_ABI_VERSION; }

const char* ag_last_error(const ag_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int ag_ctx_create(int32_t device_id, const ag_dims* dims, ag_ctx** out) {
    if (!dims || !out) return AG_ERR_INVALID;
    *out = nullptr;
    ag_ctx* c = new ag_ctx();
    c->device = device_id; c->dims = *dims;
    options_from_env(c->opt);
    *out = c;   // returned even on failure so the caller can read ag_last_error, then destroy
    const bool his_ok = (dims->n_his == 4 || dims->n_his == N_HIS_MAX) && dims->rel_dim == 5 + 3 * dims->n_his;
    if (dims->nf != NF || !his_ok || dims->in_dim != IN_DIM)
        return fail(c, AG_ERR_UNSUPPORTED, "kernels are built for nf=150, in_dim=6 and n_his=4 (rel_dim 17) or n_his=5 "
                    "(rel_dim 20) - got nf %d, n_his %d, in_dim %d, rel_dim %d", dims->nf, dims->n_his, dims->in_dim, dims->rel_dim);
    if (dims->pstep < 1) return fail(c, AG_ERR_INVALID, "pstep must be >= 1");
    HIPCHK(c, hipSetDevice(device_id));
    HIPCHK(c, dev_alloc(c, reinterpret_cast<void**>(&c->d_w), (size_t)WeightLayout::TOTAL * 4));
#ifdef AG_DIAG
    c->diag = diag_create();
#endif
    return AG_OK;
}

int ag_ctx_destroy(ag_ctx* c) {
    if (!c) return AG_OK;
    (void)hipSetDevice(c->device);
#ifdef AG_DIAG
    diag_destroy(c->diag);
#endif
    for (auto& p : c->prof_live) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    for (auto e : c->prof_pool) (void)hipEventDestroy(e);
    for (CallSlot& k : c->slots) slot_destroy(c, k);
    if (c->d_base_cache) (void)hipFree(c->d_base_cache);
    if (c->d_w) (void)hipFree(c->d_w);
    if (c->d_wb3) (void)hipFree(c->d_wb3);
    if (c->d_wlat) (void)hipFree(c->d_wlat);
    if (c->d_cself) (void)hipFree(c->d_cself);
    delete c;
    return AG_OK;
}

int ag_ctx_set_precision(ag_ctx* c, int32_t mode) {
    if (!c) return AG_ERR_INVALID;
    if (mode != 0 && mode != 1) return fail(c, AG_ERR_INVALID, "precision mode must be 0 (fp32) or 1 (bf16x3)");
    if (mode == c->precision) return AG_OK;
    if (mode == 1 && c->dims.n_his != 4) return fail(c, AG_ERR_UNSUPPORTED, "the bf16x3 chains are built for n_his=4");
    c->precision = mode;
    ++c->weights_version;
    return c->have_w ? compute_self_rows(c) : AG_OK;
}

int ag_ctx_set_chunk(ag_ctx* c, int32_t n) {
    if (!c || n < 0) return AG_ERR_INVALID;
    c->chunk = n;
    return AG_OK;
}

int ag_ctx_set_option(ag_ctx* c, const char* name, int32_t value) {
    if (!c || !name) return AG_ERR_INVALID;
    for (const OptName& n : kOptions)
        if (!strcmp(name, n.name)) {
            if (value < n.lo || value > n.hi)
                return fail(c, AG_ERR_INVALID, "option '%s' = %d is outside [%d, %d]", name, value, n.lo, n.hi);
            c->opt.*(n.field) = value;
            return AG_OK;
        }
    return fail(c, AG_ERR_INVALID, "unknown option '%s'", name);
}

int ag_ctx_get_option(ag_ctx* c, const char* name, int32_t* out) {
    if (!c || !name || !out) return AG_ERR_INVALID;
    for (const OptName& n : kOptions)
        if (!strcmp(name, n.name)) { *out = c->opt.*(n.field); return AG_OK; }
    return fail(c, AG_ERR_INVALID, "unknown option '%s'", name);
}

int ag_ctx_rollout_counts(ag_ctx* c, int64_t* out_executed, int64_t* out_needed) {
    if (!c || !out_executed || !out_needed) return AG_ERR_INVALID;
    if (c->d_plan_sums) {                                    // device-planned call: the sums are still on the device
        std::vector<int> h((size_t)c->plan_sums_n * 2);
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, hipMemcpy(h.data(), c->d_plan_sums, h.size() * 4, hipMemcpyDeviceToHost));
        c->fwd_needed = 0; c->fwd_executed = 0;
        for (int i = 0; i < c->plan_sums_n; ++i) { c->fwd_needed += h[2 * i]; c->fwd_executed += h[2 * i + 1]; }
        c->d_plan_sums = nullptr;
    }
    *out_executed = c->fwd_executed; *out_needed = c->fwd_needed;
    return AG_OK;
}

int ag_ctx_launch_counts(ag_ctx* c, int64_t* out2) {
    if (!c || !out2) return AG_ERR_INVALID;
    out2[0] = c->steps_enqueued; out2[1] = c->steps_bound;
    return AG_OK;
}

int ag_ctx_alloc_counts(ag_ctx* c, int64_t* out1) {
    if (!c || !out1) return AG_ERR_INVALID;
    out1[0] = c->n_allocs;
    return AG_OK;
}

int ag_ctx_share_counts(ag_ctx* c, int64_t* out3) {
    if (!c || !out3) return AG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    unsigned long long h[2] = {0, 0};
    int nns = 0;
    if (c->slots[c->last_slot].d_share_stats) HIPCHK(c, hipMemcpy(h, c->slots[c->last_slot].d_share_stats, sizeof h, hipMemcpyDeviceToHost));
    if (c->d_share_nns) HIPCHK(c, hipMemcpy(&nns, c->d_share_nns, 4, hipMemcpyDeviceToHost));
    out3[0] = nns; out3[1] = (int64_t)h[0]; out3[2] = (int64_t)h[1];
    return AG_OK;
}

int ag_ctx_load_weights(ag_ctx* c, const float* const* t, int32_t n) {
    if (!c) return AG_ERR_INVALID;
    if (!t || n != AG_NUM_WEIGHT_TENSORS) return fail(c, AG_ERR_INVALID, "expected %d weight tensors", AG_NUM_WEIGHT_TENSORS);
    for (int i = 0; i < n; ++i) if (!t[i]) return fail(c, AG_ERR_INVALID, "weight tensor %d is null", i);
    using WL = WeightLayout;
    std::vector<float> blob((size_t)WL::TOTAL, 0.f);
    float* b = blob.data();
    // particle encoder 0..5, relation encoder 6..11
    pack_first(b + WL::N_L1, t[0], IN_DIM, t[1], NODE_L1_CHUNKS);
    pack_layer(b + WL::N_L2, t[2], NF, 0, NF, NF, t[3], 5);
    pack_layer(b + WL::N_L3, t[4], NF, 0, NF, NF, t[5], 5);
    pack_first(b + WL::E_L1, t[6], c->dims.rel_dim, t[7], EDGE_L1_CHUNKS);
    pack_layer(b + WL::E_L2, t[8], NF, 0, NF, NF, t[9], 5);
    pack_layer(b + WL::E_L3, t[10], NF, 0, NF, NF, t[11], 5);
    // particle propagator W_pp = [Wa | Wb] (150 x 300), bias with Wa
    pack_layer(b + WL::N_WA, t[12], 2 * NF, 0, NF, NF, t[13], 5);
    pack_layer(b + WL::P_WB, t[12], 2 * NF, NF, NF, NF, nullptr, 5);
    // relation propagator W_rp = [W1 | W2 | W3] (150 x 450), bias with W1
    pack_layer(b + WL::E_W1, t[14], 3 * NF, 0, NF, NF, t[15], 5);
    pack_layer(b + WL::N_W2, t[14], 3 * NF, NF, NF, NF, nullptr, 5);
    pack_layer(b + WL::N_W3, t[14], 3 * NF, 2 * NF, NF, NF, nullptr, 5);
    // predictor
    pack_layer(b + WL::P_P0, t[16], NF, 0, NF, NF, t[17], 5);
    pack_layer(b + WL::P_P1, t[18], NF, 0, NF, NF, t[19], 5);
    pack_layer(b + WL::P_P2, t[20], NF, 0, 3, NF, t[21], 1);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(c->d_w, b, blob.size() * 4, hipMemcpyHostToDevice));
    // bf16x3 image (opt-in precision mode, ag_ctx_set_precision; n_his = 4 models only)
    if (c->dims.n_his == 4) {
        std::vector<uint16_t> img((size_t)B3_PHASES * B3_PHASE_BYTES / 2, 0);
        auto ph = [&](int phase) { return img.data() + (size_t)phase * B3_PHASE_BYTES / 2; };
        // phase indices = WLB in ag_mlp.hip
        pack_first_b3(ph(0), t[6], REL_DIM, t[7]);
        pack_layer_b3(ph(1), t[8], NF, 0, NF, NF, t[9], 5);
        pack_layer_b3(ph(6), t[10], NF, 0, NF, NF, t[11], 5);
        pack_layer_b3(ph(11), t[14], 3 * NF, 0, NF, NF, t[15], 5);
        pack_first_b3(ph(16), t[0], IN_DIM, t[1]);
        pack_layer_b3(ph(17), t[2], NF, 0, NF, NF, t[3], 5);
        pack_layer_b3(ph(22), t[4], NF, 0, NF, NF, t[5], 5);
        pack_layer_b3(ph(27), t[12], 2 * NF, 0, NF, NF, t[13], 5);
        pack_layer_b3(ph(32), t[14], 3 * NF, NF, NF, NF, nullptr, 5);
        pack_layer_b3(ph(37), t[14], 3 * NF, 2 * NF, NF, NF, nullptr, 5);
        pack_layer_b3(ph(42), t[12], 2 * NF, NF, NF, NF, nullptr, 5);
        pack_layer_b3(ph(47), t[16], NF, 0, NF, NF, t[17], 5);
        pack_layer_b3(ph(52), t[18], NF, 0, NF, NF, t[19], 5);
        pack_layer_b3(ph(57), t[20], NF, 0, 3, NF, t[21], 1);
        if (!c->d_wb3) HIPCHK(c, dev_alloc(c, reinterpret_cast<void**>(&c->d_wb3), img.size() * 2));
        HIPCHK(c, hipMemcpy(c->d_wb3, img.data(), img.size() * 2, hipMemcpyHostToDevice));
    }
    if (c->dims.n_his == 4) {   // latency-mode image
        std::vector<float> wl(lat_weights_floats(), 0.f);
        float* L = wl.data();
        pack_first_lat(L + lat_weights_offset(0), t[6], REL_DIM, t[7], 2);
        pack_layer_lat(L + lat_weights_offset(1), t[8], NF, 0, NF, NF, t[9], false);
        pack_layer_lat(L + lat_weights_offset(2), t[10], NF, 0, NF, NF, t[11], false);
        pack_layer_lat(L + lat_weights_offset(3), t[14], 3 * NF, 0, NF, NF, t[15], false);        // W1 + b_rp
        pack_layer_lat(L + lat_weights_offset(4), t[12], 2 * NF, NF, NF, NF, nullptr, false);     // Wb
        pack_layer_lat(L + lat_weights_offset(5), t[14], 3 * NF, NF, NF, NF, nullptr, false);     // W2
        pack_layer_lat(L + lat_weights_offset(6), t[14], 3 * NF, 2 * NF, NF, NF, nullptr, false); // W3
        pack_layer_lat(L + lat_weights_offset(7), t[16], NF, 0, NF, NF, t[17], false);            // predictor 0
        pack_layer_lat(L + lat_weights_offset(8), t[18], NF, 0, NF, NF, t[19], false);            // predictor 1
        pack_layer_lat(L + lat_weights_offset(9), t[20], NF, 0, 3, NF, t[21], true);              // predictor 2 (3 outputs)
        pack_first_lat(L + lat_weights_offset(10), t[0], IN_DIM, t[1], 1);                         // particle encoder 0 (r06)
        pack_layer_lat(L + lat_weights_offset(11), t[2], NF, 0, NF, NF, t[3], false);             // particle encoder 2
        pack_layer_lat(L + lat_weights_offset(12), t[4], NF, 0, NF, NF, t[5], false);             // particle encoder 4
        pack_layer_lat(L + lat_weights_offset(13), t[12], 2 * NF, 0, NF, NF, t[13], false);       // Wa + b_pp
        if (!c->d_wlat) HIPCHK(c, dev_alloc(c, reinterpret_cast<void**>(&c->d_wlat), wl.size() * 4));
        HIPCHK(c, hipMemcpy(c->d_wlat, wl.data(), wl.size() * 4, hipMemcpyHostToDevice));
    }
    c->have_w = true;
    ++c->weights_version;
    return compute_self_rows(c);
}

int ag_build_edges(ag_ctx* c, void* stream, const float* d_pos, const uint8_t* d_mask, const uint8_t* d_tool,
                   int32_t B, int32_t N, float thr, const float* d_thr_vec, int32_t topk, int32_t cta, int32_t edge_cap,
                   int32_t* d_recv, int32_t* d_send, int32_t* d_row_ptr, int32_t* d_n_edges) {
    if (!c) return AG_ERR_INVALID;
    if (!d_pos || !d_mask || !d_tool || !d_recv || !d_send || !d_row_ptr || !d_n_edges || B < 1 || edge_cap < 1)
        return fail(c, AG_ERR_INVALID, "ag_build_edges: null pointer or empty batch");
    int rc = check_topk(c, N, topk);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const int slices = pick_slices(c, B, N);
    const size_t rows = (size_t)B * N;
    const int ell_stride = edge_ell_stride(N, topk);
    hipStream_t st = static_cast<hipStream_t>(stream);
    CallSlot* sl = nullptr;
    rc = slot_acquire(c, st, false, &sl);
    if (rc) return rc;
    SlotGuard slot_guard(sl, st, false);
    rc = ensure_slab(c, *sl, rows * (size_t)(ell_stride + 1) * 4 + (size_t)B * (slices + 1) * 4 + 4096);
    if (rc) return rc;
    EdgeArgs a{};
    a.pos = d_pos; a.pos_bstride = (long)N * 3; a.mask = d_mask; a.tool = d_tool; a.thr_vec = d_thr_vec; a.thr = thr;
    a.B = B; a.N = N; a.topk = topk; a.cta = cta ? 1 : 0; a.edge_cap = edge_cap; a.slices = slices;
    a.ell = sl->slab.take<int>(rows * (size_t)std::max(1, ell_stride));
    a.deg = sl->slab.take<int>(rows);
    a.slice_tot = sl->slab.take<int>((size_t)B * slices);
    a.cta_flag = sl->slab.take<int>(B);
    a.recv = d_recv; a.send = d_send; a.row_ptr = d_row_ptr; a.n_edges = d_n_edges; a.overflow = nullptr;
    a.max_nR = edge_cap; a.zero_on_overflow = 0; a.block_min_rows = c->opt.edge_block_min;
    c->prof_stream = st;
    HIPCHK(c, launch_edge_build(a, st, prof_mark, c));
    return AG_OK;
}

int ag_build_edges_single(ag_ctx* c, void* stream, const float* d_pos, const uint8_t* d_mask, const uint8_t* d_tool,
                          int32_t N, float thr2, float cull_radius, int32_t topk, int32_t cta, int32_t edge_cap,
                          int32_t* d_recv, int32_t* d_send, int32_t* d_row_ptr, int32_t* d_n_edges) {
    if (!c) return AG_ERR_INVALID;
    if (!d_pos || !d_mask || !d_tool || !d_recv || !d_send || !d_row_ptr || !d_n_edges || edge_cap < 1)
        return fail(c, AG_ERR_INVALID, "ag_build_edges_single: null pointer");
    if (!(cull_radius * cull_radius >= thr2)) return fail(c, AG_ERR_INVALID, "cull_radius^2 must be >= thr2");
    int rc = check_topk(c, N, topk);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const int slices = pick_slices(c, 1, N);
    const int ell_stride = edge_ell_stride(N, topk);
    hipStream_t st = static_cast<hipStream_t>(stream);
    CallSlot* sl = nullptr;
    rc = slot_acquire(c, st, false, &sl);
    if (rc) return rc;
    SlotGuard slot_guard(sl, st, false);
    rc = ensure_slab(c, *sl, (size_t)N * (size_t)(ell_stride + 1) * 4 + (size_t)(slices + 1) * 4 + 4096);
    if (rc) return rc;
    EdgeArgs a{};
    a.pos = d_pos; a.pos_bstride = (long)N * 3; a.mask = d_mask; a.tool = d_tool; a.thr_vec = nullptr; a.thr = cull_radius;
    a.thr2_override = thr2; a.use_thr2 = 1;
    a.B = 1; a.N = N; a.topk = topk; a.cta = cta ? 2 : 0; a.edge_cap = edge_cap; a.slices = slices;
    a.ell = sl->slab.take<int>((size_t)N * (size_t)std::max(1, ell_stride));
    a.deg = sl->slab.take<int>(N);
    a.slice_tot = sl->slab.take<int>(slices);
    a.cta_flag = sl->slab.take<int>(1);
    a.recv = d_recv; a.send = d_send; a.row_ptr = d_row_ptr; a.n_edges = d_n_edges; a.overflow = nullptr;
    a.max_nR = edge_cap; a.zero_on_overflow = 0; a.block_min_rows = c->opt.edge_block_min;
    c->prof_stream = st;
    HIPCHK(c, launch_edge_build(a, st, prof_mark, c));
    return AG_OK;
}

int ag_edges_apply_tool_rule(ag_ctx* c, void* stream, const float* d_pos, const uint8_t* d_mask, const uint8_t* d_tool,
                             int32_t N, int32_t n_tools, const int32_t* d_send_in, const int32_t* d_row_ptr_in,
                             const uint8_t* d_subset, double kNN, int32_t edge_cap, int32_t* d_recv, int32_t* d_send,
                             int32_t* d_row_ptr, int32_t* d_n_out) {
    if (!c) return AG_ERR_INVALID;
    if (!d_pos || !d_mask || !d_tool || !d_send_in || !d_row_ptr_in || !d_subset || !d_recv || !d_send || !d_row_ptr || !d_n_out)
        return fail(c, AG_ERR_INVALID, "ag_edges_apply_tool_rule: null pointer");
    if (N < 1 || n_tools < 0 || n_tools > N || edge_cap < 1)
        return fail(c, AG_ERR_INVALID, "ag_edges_apply_tool_rule: bad sizes N=%d n_tools=%d edge_cap=%d", N, n_tools, edge_cap);
    if (N > 4096) return fail(c, AG_ERR_UNSUPPORTED, "ag_edges_apply_tool_rule: N=%d exceeds 4096", N);
    if (d_send_in == d_send || d_row_ptr_in == d_row_ptr)
        return fail(c, AG_ERR_INVALID, "ag_edges_apply_tool_rule: input and output arrays must differ");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t pairs = (size_t)N * (size_t)std::max(1, n_tools);
    hipStream_t st = static_cast<hipStream_t>(stream);
    CallSlot* sl = nullptr;
    int rc = slot_acquire(c, st, false, &sl);
    if (rc) return rc;
    SlotGuard slot_guard(sl, st, false);
    rc = ensure_slab(c, *sl, pairs * 6 + (size_t)(N + n_tools + 16) * 4 + 8 * 256);
    if (rc) return rc;
    RuleArgs a{};
    a.pos = d_pos; a.mask = d_mask; a.tool = d_tool; a.subset = d_subset; a.send_in = d_send_in; a.row_ptr_in = d_row_ptr_in;
    a.N = N; a.n_tools = n_tools; a.edge_cap = edge_cap; a.use_knn = (kNN < 1.0 && kNN > 0.0) ? 1 : 0; a.kNN = kNN;   // graph.py:156
    a.tlist = sl->slab.take<int>(std::max(1, n_tools));
    a.misc = sl->slab.take<int>(16);
    a.pdis = sl->slab.take<float>(pairs);
    a.keep = sl->slab.take<uint8_t>(pairs);
    a.kept = sl->slab.take<uint8_t>(pairs);
    a.deg = sl->slab.take<int>(N);
    a.recv = d_recv; a.send = d_send; a.row_ptr = d_row_ptr; a.n_out = d_n_out;
    HIPCHK(c, launch_tool_rule(a, st));
    return AG_OK;
}

int ag_forward(ag_ctx* c, void* stream, const float* d_state, const float* d_attrs, const float* d_action,
               const float* d_phys, const float* d_group, int32_t n_inst, const int32_t* d_recv, const int32_t* d_send,
               const int32_t* d_row_ptr, const int32_t* d_n_edges, int32_t edge_cap, int32_t B, int32_t N, int32_t n_p,
               float* d_pred_pos, float* d_pred_motion) {
    if (!c) return AG_ERR_INVALID;
    if (!c->have_w) return fail(c, AG_ERR_NO_WEIGHTS, "ag_forward before ag_ctx_load_weights");
    if (!d_state || !d_attrs || !d_action || !d_phys || !d_group || !d_recv || !d_send || !d_row_ptr || !d_n_edges ||
        !d_pred_pos || !d_pred_motion)
        return fail(c, AG_ERR_INVALID, "ag_forward: null pointer");
    if (B < 1 || N < 1 || n_p < 1 || n_p > N || n_inst < 1 || edge_cap < 1)
        return fail(c, AG_ERR_INVALID, "ag_forward: bad sizes B=%d N=%d n_p=%d n_inst=%d edge_cap=%d", B, N, n_p, n_inst, edge_cap);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    c->prof_stream = st;
    const int c_cap = (int)round_up(edge_cap, 256);
    const int Bc = clamp_chunk_for_offsets(auto_chunk(c, B, N), N, c_cap);
    Work w{};
    CallSlot* sl = nullptr;
    int rc = slot_acquire(c, st, false, &sl);
    if (rc) return rc;
    SlotGuard slot_guard(sl, st, false);
    rc = ensure_slab(c, *sl, work_bytes(Bc, N, n_inst, edge_cap, c_cap, 1, false, false, false, n_p, 0) + (size_t)B * 4 + 512);
    if (rc) return rc;
    rc = carve_work(c, sl->slab, w, Bc, N, n_inst, edge_cap, c_cap, 1, false, false, false, n_p, 0);
    if (rc) return rc;
    // the caller's graphs may be overflowed (true count > edge_cap, indices never written): guard, then report
    int* n_eff = sl->slab.take<int>((size_t)B);
    if (sl->slab.used > sl->slab.cap) return fail(c, AG_ERR_INVALID, "internal: workspace carve overflow");
    HIPCHK(c, hipMemsetAsync(sl->d_words, 0, 4, st));
    HIPCHK(c, launch_edge_guard(d_n_edges, B, edge_cap, n_eff, sl->d_words, st));
    for (int b0 = 0; b0 < B; b0 += Bc) {
        const int nb = std::min(Bc, B - b0);
        GraphBufs g = w.g;
        g.B = nb; g.n_p = n_p; g.n_his = c->dims.n_his;
        g.wb3 = c->precision == 1 ? c->d_wb3 : nullptr;
        g.group = const_cast<float*>(d_group) + (size_t)b0 * N * n_inst;
        g.recv = d_recv + (size_t)b0 * edge_cap; g.send = d_send + (size_t)b0 * edge_cap;
        g.row_ptr = d_row_ptr + (size_t)b0 * (N + 1); g.n_edges = n_eff + b0; g.n_guard = n_eff + b0;
        { Scoped p(c, FAM_PREP);
          HIPCHK(c, launch_prep(d_state + (size_t)b0 * c->dims.n_his * N * 3, d_attrs + (size_t)b0 * N * 2,
                                d_action + (size_t)b0 * N * 3, d_phys + (size_t)b0 * N, g, st)); }
        rc = run_model(c, g, d_pred_pos + (size_t)b0 * n_p * 3, d_pred_motion + (size_t)b0 * n_p * 3, st);
        if (rc) return rc;
    }
    int seen = 0;
    HIPCHK(c, hipMemcpyAsync(&seen, sl->d_words, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (seen > 0) return fail(c, AG_ERR_MAX_NR, "Exceeds max dims: a graph had %d edges, edge_cap=%d", seen, edge_cap);
    return AG_OK;
}

}  // extern "C"

namespace {
// where a rollout's actions come from: decoded on the host by the caller (ag_rollout / ag_rollout_async), or raw on the
// device (ag_rollout_actions: decode + launch plan by k_roll_plan, the host never sees them)
struct ActionSrc {
    const float* d_eef_xz = nullptr; const float* d_eef_delta = nullptr; const int32_t* h_repeat = nullptr;   // host plan
    const float* d_action = nullptr; float push_length = 0.f; const float* h_tool_off = nullptr; int max_repeat = 0;
    float* d_action_seqs = nullptr;                                                                            // device plan
    int32_t* h_work = nullptr;    // ag_rollout_work: plan only - forwards each candidate would be stepped, to the host; nothing is rolled out
};

int rollout_impl(ag_ctx* c, void* stream, const ag_rollout_params* p, const float* d_state0, const uint8_t* d_obj_mask,
                 const ActionSrc& src, const float* d_phys_vec, float* d_state_seqs, int32_t* d_overflow_flag) {
    const bool dev_plan = src.d_action != nullptr;
    const float* d_eef_xz = src.d_eef_xz; const float* d_eef_delta = src.d_eef_delta; const int32_t* h_repeat = src.h_repeat;
    if (!c) return AG_ERR_INVALID;
    if (!c->have_w) return fail(c, AG_ERR_NO_WEIGHTS, "ag_rollout before ag_ctx_load_weights");
    const bool work_only = src.h_work != nullptr;
    if (!p || !d_state0 || (!d_state_seqs && !work_only) || !d_overflow_flag || (!dev_plan && (!d_eef_xz || !d_eef_delta || !h_repeat)) ||
        (dev_plan && (!src.d_action_seqs || (p->M > 1 && !src.h_tool_off))))
        return fail(c, AG_ERR_INVALID, "ag_rollout: null pointer");
    if (dev_plan && (src.max_repeat < 0 || src.max_repeat > 1024 || p->M > 8))
        return fail(c, AG_ERR_INVALID, "ag_rollout_actions: max_repeat must be in [0, 1024] and M <= 8 (got %d, %d)", src.max_repeat, p->M);
    if (dev_plan && p->y_mode != 0)
        return fail(c, AG_ERR_UNSUPPORTED, "ag_rollout_actions serves dynamics() (y_mode 0); the masked variant takes host-decoded actions");
    if (p->B < 1 || p->H < 1 || p->N_o < 1 || p->M < 1 || p->max_nR < 1)
        return fail(c, AG_ERR_INVALID, "ag_rollout: bad sizes B=%d H=%d N_o=%d M=%d max_nR=%d", p->B, p->H, p->N_o, p->M, p->max_nR);
    if (p->y_mode != 0 && p->y_mode != 1) return fail(c, AG_ERR_INVALID, "y_mode must be 0 or 1");
    if (p->y_mode == 1 && p->H != 1) return fail(c, AG_ERR_INVALID, "masked rollout has a single look-ahead step");
    const int n_his = c->dims.n_his;                          // 4 (every planner task config) or 5 (softbody.yaml:29)
    const int N = p->N_o + p->M;
    int rc = check_topk(c, N, p->topk);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    c->prof_stream = st;

    const size_t nrep = (size_t)p->B * p->H;
    // A caller may be capturing this call into a hipGraph (tools/graph_replay.py): nothing of it may then look at the host side of
    // an event or wait - no polling of the plan's maxima, no prefix sharing (both only save work; results are the same)
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap_status) == hipSuccess && cap_status == hipStreamCaptureStatusActive;
    // workspace, plans and read-back buffers of this call: the slot of the caller's stream (calls on other streams have their own
    // and may still be running; a taken-over slot has been waited for)
    CallSlot* slp = nullptr;
    rc = slot_acquire(c, st, capturing, &slp);
    if (rc) return rc;
    CallSlot& sl = *slp;
    SlotGuard slot_guard(slp, st, capturing);                // (a captured event could not be waited for outside its graph)
    c->last_slot = (int)(slp - c->slots);
    c->d_share_nns = nullptr;                                // pointed into a workspace of an earlier call
    if (d_state_seqs) HIPCHK(c, hipMemsetAsync(d_state_seqs, 0, (size_t)p->B * p->H * p->N_o * 3 * 4, st));   // forward_dynamics.py:32

    const int k = std::min(N, p->topk);
    const long bound = (long)N * (k + p->M);                 // in-degree <= topk + M (radius-AND-top-k, then tool rule)
    // Fast path (top-k active; the rollout keeps its tool particles behind the object particles): the count kernel's
    // per-row sender lists are used as the graph, slot-indexed (EdgeArgs::ell_full) - no emit pass, no CSR copy.
    // Every row then owns topk + M slots whatever max_nR is (the max_nR rule is applied by k_ell_index).
    const bool dedupe = c->opt.self_dedupe != 0;
    const bool ell_full = c->opt.ell_graph && dedupe && k < N;
    const int edge_cap = (int)round_up((size_t)(ell_full ? bound : std::min<long>(bound, p->max_nR)), 256);
    int ns = std::max(1, std::min(c->n_streams, (int)ag_ctx::kMaxStreams));
    {   // batches of eight or more full-size chunks run on four streams (two chunks each): the memory-bound phases of
        // three chunks then hide under the MFMA-bound k_edge_enc of a fourth (1024 x 2026 cloth: 497.8 ms on two streams,
        // 491.8 on three, 488.6 on four; with fewer chunks the streams would only cut them smaller)
        const int full = auto_chunk(c, p->B, N);
        if (c->n_streams == 2 && (p->B + full - 1) / full >= 8) ns = 4;
    }
    if ((long)p->B * N < c->opt.stream_min_rows) ns = 1;   // small batches are dispatch-bound: a second stream only doubles the launches
                                          // (rope 64 x 301 rows x 20 steps: 9.99 ms on one stream, 11.3 on two; 128 x 301: 14.2 / 13.4)
    // a caller that pipelines independent calls over several streams (the planner's chunk loop) already fills the chip across
    // calls: no fork inside a call that starts while a call of another stream is still running
    if (ns > 1 && !capturing && c->opt.pipeline_fork == 0 && other_slot_busy(c, &sl)) ns = 1;
    if (c->opt.streams > 0) ns = std::min(c->opt.streams, (int)ag_ctx::kMaxStreams);
    // per-kernel event times are only meaningful without cross-stream interference; bit 30 of the mask keeps the
    // streams (the durations then include whatever the other stream ran beside the kernel)
    if ((c->prof_mask & 0x3fffffffu) && !(c->prof_mask & (1u << 30))) ns = 1;
    // Ragged batches (the masked variant: every candidate has its own number of valid particles): the propagate chains
    // walk a compact row list, and one extra candidate slot per workspace - the phantom candidate, see GraphBufs - stands
    // for every masked-out particle.  Options::ragged = 0 keeps the dense rows (A/B measurements).
    const bool ragged = c->opt.ragged && p->y_mode == 1 && d_obj_mask != nullptr;
    // (the phantom candidate's rows must stay inside the 32-bit element offsets too)
    int Bc = ragged ? std::max(1, clamp_chunk_for_offsets(auto_chunk(c, p->B, N) + 1, N, edge_cap) - 1)
                    : clamp_chunk_for_offsets(auto_chunk(c, p->B, N), N, edge_cap);
    if (ns > 1) Bc = std::min(Bc, (p->B + ns - 1) / ns);      // at least one chunk per stream
    {   // equal-sized chunks, a multiple of the stream count of them (no short last chunk, no idle stream at the end)
        int n_chunks = (p->B + Bc - 1) / Bc;
        if (ns > 1) n_chunks = (n_chunks + ns - 1) / ns * ns;
        Bc = (p->B + n_chunks - 1) / n_chunks;
    }
    if (p->B <= 1) ns = 1;
    if (work_only) { ns = 1; Bc = 1; }                       // plan only: the one workspace a base rollout needs
    const int slices = pick_slices(c, Bc, N);
    const int ell = edge_ell_stride(N, p->topk);
    const int Ba = Bc + (ragged ? 1 : 0);                    // candidate slots per workspace

    // Repeat-aware launch order (Options::repeat_sort).  The reference steps the WHOLE batch to the batch maximum of
    // action_repeat and discards the surplus forwards (forward_dynamics.py:156-161).  Here, per launch chunk and
    // look-ahead step, the chunk's candidates are put in descending order of their repeat count (stable): the candidates
    // that still have forwards to run at step ai are then a PREFIX of the chunk's slots, and every kernel of that step is
    // launched over that prefix only.  Executed candidate-forwards = sum of action_repeat, exactly.  A slot's candidate
    // may change between look-ahead steps: the state carried from one to the next lives in d_state_seqs, which k_roll_init
    // reads by candidate id.  Candidates are independent, so every candidate's result is bit-identical to the unsorted
    // order's.  Ragged batches (one look-ahead step) build their row list in the sorted slot order, with the row count of
    // every live prefix tabulated beside it.
    const bool sort_on = c->opt.repeat_sort != 0;
    const int n_chunks_all = (p->B + Bc - 1) / Bc;
    int* h_cand = nullptr;
    // device plan: pointers into sl.d_plan
    const int R = src.max_repeat;
    int *pl_repeat = nullptr, *pl_cand = nullptr, *pl_live = nullptr, *pl_rows = nullptr, *pl_sums = nullptr;
    float *pl_xz = nullptr, *pl_delta = nullptr;
    c->d_plan_sums = nullptr;
    // Contact-free prefix (Options::share_prefix; RollArgs::start).  A tool acts on the object only through the edges it takes
    // part in, and it takes part in none while no object particle is inside its radius.  Until then a
    // candidate's object particles evolve exactly - bit for bit: a row's result does not depend on the rest of its batch - like
    // the start state WITHOUT a tool.  That base rollout is computed once per call (one candidate, tool parked out of reach);
    // k_contact_plan replays every candidate's tool along it and finds the forward of its first contact; a candidate is then
    // stepped only from there on (its slot starts from the base state and history of that step), and one that never touches
    // takes the base state of its last step.  The reference's planner samples its pushes uniformly over the workspace
    // (plan_utils.py:48-50 with planning/*.yaml:28-29): most of them never reach the object.  Look-ahead step 0 only (later
    // steps start from per-candidate states).  The contact plan decides the launch sizes, so the call waits for it once - the GPU
    // is busy with the base rollout meanwhile.
    // (connect_tools_all does not change the argument: its tool -> object edges are all-or-nothing on "some object sits inside a
    // tool particle's radius", graph.py:276-286 - the very contact that is tested; shipped cloth pushes just start on the cloth)
    bool prefix = c->opt.share_prefix != 0 && p->y_mode == 0 && !d_obj_mask && p->M <= 8 && !capturing;
    if (c->opt.share_prefix < 0 && (p->B < 64 || (long)p->B * N < 32768)) prefix = false;
    int R_base = 0;                                          // steps of the base rollout = the largest repeat of look-ahead step 0
    if (prefix) {
        if (dev_plan) R_base = R;
        else for (int b = 0; b < p->B; ++b) R_base = std::max(R_base, (int)h_repeat[(size_t)b * p->H]);
        if (R_base < 1) prefix = false;
    }
    const bool auto_prefix = prefix && c->opt.share_prefix < 0;
    const bool base_in_ctx = auto_prefix && !d_phys_vec;      // automatic mode: the base rollout lives in the context, for later calls

    // Shared first forward (Options::share_first).  dynamics() broadcasts ONE start state to all candidates with a constant
    // history (forward_dynamics.py:25), then builds and encodes every candidate's graph separately (:125, model.py:303).  At
    // that forward the relation input of an object-object edge - attrs, group difference, position / residual differences
    // (model.py:249-282) - does not depend on the candidate, so neither does its C row; and the object senders a candidate's
    // receiver keeps are a subset of what it keeps in the start state's graph WITHOUT the tool (a tool can only push senders
    // out of a row's top-k).  So: build that base graph once per call, run the edge chain once over its non-self edges into
    // a shared table, and let the first forward's message passing take the C row of every slot found in the base row from
    // there (k_ell_index: send_pk); per candidate only the edges with a tool at either end are encoded.  Bit-identical: a
    // row's chain does not depend on the lane / workgroup / launch that computes it.
    const int kb = std::min(p->N_o, p->topk);
    // (with the prefix sharing only the candidates that touch at once start from the start state: EdgeArgs::share_start)
    bool share = c->opt.share_first != 0 && p->y_mode == 0 && !d_obj_mask && ell_full && p->topk < p->N_o && k <= 255;
    if (c->opt.share_first < 0 && p->B < 8) share = false;   // a handful of candidates: the base build costs more than it saves
    if (work_only) share = false;
    {   // launches small enough for the latency-mode propagate chains (ag_lat.hip) keep their own C rows
        GraphBufs gt{};
        gt.B = std::min(Bc, p->B); gt.N = N; gt.n_his = n_his; gt.wb3 = c->precision == 1 ? c->d_wb3 : nullptr;
        if (lat_node_for(c, gt)) share = false;
    }
    const int base_cap = (int)round_up((size_t)p->N_o * kb, 256);
    const int base_slices = pick_slices(c, 1, p->N_o);
    const size_t base_bytes = !share ? 0 : (size_t)base_cap * (NFP + 3) * 4 + (size_t)p->N_o * (NODE_IN + F15_PITCH + 2) * 4 +
                                           2 * (size_t)p->N_o + (size_t)(base_slices + 8) * 4 + 24 * 256;
    // (sized before the census below may still switch the sharing off: its scratch is carved first)
    const size_t prefix_bytes = !prefix ? 0 : ((base_in_ctx ? 0 : (size_t)(R_base + 1) * (p->N_o * 3 + 1)) + 2 * nrep + p->B + 5 * p->M + 64) * 4 + 16 * 256;
    const size_t wb = work_bytes(Ba, N, 1, edge_cap, edge_cap, slices, true, true, true, p->N_o, ell);
    rc = ensure_slab(c, sl, wb * ns + base_bytes + prefix_bytes);
    if (rc) return rc;
    float* b_states = nullptr; float* b_y = nullptr;
    int* b_rep_eff = nullptr; int* b_start = nullptr; float* b_eef = nullptr; int* b_zero = nullptr;
    if (prefix) {
        if (!base_in_ctx) { b_states = sl.slab.take<float>((size_t)(R_base + 1) * p->N_o * 3); b_y = sl.slab.take<float>(R_base + 1); }
        b_rep_eff = sl.slab.take<int>(nrep); b_start = sl.slab.take<int>(p->B);
        b_eef = sl.slab.take<float>((size_t)5 * p->M);       // parked tool: xz (M,2), delta (M,3)
        b_zero = sl.slab.take<int>(1);
        if (sl.slab.used > sl.slab.cap) return fail(c, AG_ERR_INVALID, "internal: workspace carve overflow");
    }
    if (prefix || work_only) {
        if (sl.rep_pin_cap < 2 * nrep + 8) {                 // pinned read-back of the contact plan: [forwards left | repeat | flag, census x4]
            if (sl.h_rep_pin) HIPCHK(c, pin_free(c, sl.h_rep_pin));
            sl.h_rep_pin = nullptr; sl.rep_pin_cap = 0;
            HIPCHK(c, pin_alloc(c, reinterpret_cast<void**>(&sl.h_rep_pin), (2 * nrep + 64) * 4));
            sl.rep_pin_cap = 2 * nrep + 64;
        }
    }

    // host plan: repeat counts -> per chunk and look-ahead step the launch order (descending repeat, stable), both uploaded
    auto host_plan = [&](const int32_t* rep_src) -> int {
        if (sl.repeat_cap < 2 * nrep) {
            if (sl.d_repeat) HIPCHK(c, dev_free(c, sl.d_repeat));
            sl.d_repeat = nullptr; sl.repeat_cap = 0;
            HIPCHK(c, dev_alloc(c, reinterpret_cast<void**>(&sl.d_repeat), (2 * nrep + (nrep >> 2)) * 4));
            sl.repeat_cap = 2 * nrep + (nrep >> 2);
        }
        sl.h_repeat.resize(2 * nrep);
        if (rep_src != sl.h_repeat.data()) std::copy(rep_src, rep_src + nrep, sl.h_repeat.begin());
        h_repeat = sl.h_repeat.data();
        h_cand = sl.h_repeat.data() + nrep;                  // [li][slot] -> candidate
        for (int li = 0; li < p->H; ++li)
            for (int b0 = 0; b0 < p->B; b0 += Bc) {
                const int nb = std::min(Bc, p->B - b0);
                int* seg = h_cand + (size_t)li * p->B + b0;
                for (int b = 0; b < nb; ++b) seg[b] = b0 + b;
                if (sort_on)
                    std::stable_sort(seg, seg + nb, [&](int x, int y) { return h_repeat[(size_t)x * p->H + li] > h_repeat[(size_t)y * p->H + li]; });
            }
        HIPCHK(c, hipMemcpyAsync(sl.d_repeat, h_repeat, 2 * nrep * 4, hipMemcpyHostToDevice, st));
        return AG_OK;
    };
    if (!dev_plan) {
        rc = host_plan(h_repeat);                            // (prefix sharing plans again, with the forwards that are left)
        if (rc) return rc;
        c->fwd_executed = 0; c->fwd_needed = 0;
        for (size_t i = 0; i < nrep; ++i) c->fwd_needed += std::max(0, h_repeat[i]);
    } else {
        // Device plan: one kernel decodes the actions (plan_utils.py:11-20, forward_dynamics.py:42-75), orders every
        // chunk's candidates by action_repeat and tabulates how many are live at every step; the launches below take their
        // live counts from that table (device memory), so nothing of the actions ever crosses to the host.
        const size_t tab = (size_t)n_chunks_all * p->H * (R + 2);
        const size_t n_int = 2 * nrep + 2 * tab + (size_t)n_chunks_all * p->H * 3;
        const size_t n_flt = nrep * p->M * 5;
        const size_t bytes = round_up(n_int * 4, 256) + n_flt * 4;
        if (sl.plan_cap < bytes) {
            if (sl.d_plan) HIPCHK(c, dev_free(c, sl.d_plan));
            sl.d_plan = nullptr; sl.plan_cap = 0;
            HIPCHK(c, dev_alloc(c, reinterpret_cast<void**>(&sl.d_plan), bytes + (bytes >> 2)));
            sl.plan_cap = bytes + (bytes >> 2);
        }
        pl_repeat = reinterpret_cast<int*>(sl.d_plan); pl_cand = pl_repeat + nrep; pl_live = pl_cand + nrep;
        pl_rows = pl_live + tab; pl_sums = pl_rows + tab;
        pl_xz = reinterpret_cast<float*>(sl.d_plan + round_up(n_int * 4, 256)); pl_delta = pl_xz + nrep * p->M * 2;
        RollPlan rp{};
        rp.action = src.d_action; rp.push_length = src.push_length; rp.M = p->M;
        for (int kk = 1; kk < p->M; ++kk) rp.tool_off[kk] = src.h_tool_off[kk];
        rp.B = p->B; rp.H = p->H; rp.Bc = Bc; rp.N = N; rp.max_repeat = R;
        rp.decoded = src.d_action_seqs; rp.eef_xz = pl_xz; rp.eef_delta = pl_delta; rp.repeat = pl_repeat; rp.cand = pl_cand;
        rp.live = pl_live; rp.rows = pl_rows; rp.sums = pl_sums; rp.flags = d_overflow_flag; rp.sort = sort_on ? 1 : 0;
        rp.maxrep = pl_sums + (size_t)n_chunks_all * p->H * 2;
        HIPCHK(c, launch_roll_plan(rp, st));
        // Every (chunk, look-ahead step)'s own maximum comes back into pinned host memory behind an event - asynchronously:
        // nothing waits for it.  The enqueue loop below polls the event (hipEventQuery) and, once it has fired, stops enqueuing
        // a look-ahead step's repeats at that maximum instead of at the caller's bound (whose surplus steps would find no live
        // slot: full grids of workgroups that exit).  Until it fires the loop goes by the bound, as before.
        const size_t n_max = (size_t)n_chunks_all * p->H;
        if (sl.plan_max_cap < n_max) {
            if (sl.h_plan_max) HIPCHK(c, pin_free(c, sl.h_plan_max));
            sl.h_plan_max = nullptr; sl.plan_max_cap = 0;
            HIPCHK(c, pin_alloc(c, reinterpret_cast<void**>(&sl.h_plan_max), (n_max + 64) * 4));
            sl.plan_max_cap = n_max + 64;
        }
        if (!capturing) {
            HIPCHK(c, hipMemcpyAsync(sl.h_plan_max, rp.maxrep, n_max * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipEventRecord(sl.ev_plan, st));
        }
        d_eef_xz = pl_xz; d_eef_delta = pl_delta;
        c->d_plan_sums = pl_sums; c->plan_sums_n = n_chunks_all * p->H;
        c->fwd_executed = -1; c->fwd_needed = -1;
    }
    const int* d_rep_orig = dev_plan ? pl_repeat : sl.d_repeat;
    const int rep_bound = dev_plan ? R : 0x7fffffff;          // a device-planned candidate beyond the caller's bound is never captured
    BaseKey key_now;
    memset(&key_now, 0, sizeof key_now);                     // (padding bytes too: the keys are compared with memcmp)
    key_now.N_o = p->N_o; key_now.M = p->M; key_now.topk = p->topk; key_now.cta = p->connect_tools_all;
    key_now.max_nR = p->max_nR; key_now.n_his = n_his; key_now.precision = c->precision;
    key_now.pstep = c->dims.pstep; key_now.grip_on = p->gripper_enable; key_now.thr = p->adj_thresh;
    key_now.grip = p->gripper_offset; key_now.phys = p->physics_param; key_now.clamp = c->dims.motion_clamp;
    key_now.phys_vec = d_phys_vec; key_now.weights_version = c->weights_version;
    // censuses that nobody waited for (below): one that has landed and finds enough free candidates lifts the standing "not worth
    // it" verdict, so that the next call of that shape takes a proper census again
    for (CallSlot& q : c->slots)
        if (q.census_pending && !capturing) {
            if (hipEventQuery(q.ev_census) == hipSuccess) {
                q.census_pending = false;
                const int free_now = q.h_census[1] - q.h_census[0], rb = std::min(q.census_R, std::max(1, q.h_census[2]));
                if (c->decision.decline && c->decision.B == q.census_B && c->decision.H == q.census_H &&
                    memcmp(&c->decision.key, &q.census_key, sizeof(BaseKey)) == 0 && free_now >= std::max(64, 8 * rb))
                    c->decision.decline = false;
            } else (void)hipGetLastError();
        }
    bool census = false, base_cached = false, plan_done = false;
    if (auto_prefix) {
        // Automatic mode: is the base rollout worth its latency-bound forwards?  Census of the FIRST forward (its graph needs
        // the start state only): how many candidates touch at once.  Sharing is kept when enough of them do not - a batch of
        // pushes aimed at the object (every candidate in contact from the first forward on) steps all of them anyway: worth it
        // when enough candidates are still free at the first forward to pay for the base rollout's latency-bound forwards (each
        // costs about as much as eight candidate-forwards of a full launch).
        ContactPlan cen{};
        cen.base_states = d_state0; cen.R = 1; cen.eef_xz = d_eef_xz; cen.eef_delta = d_eef_delta; cen.repeat = d_rep_orig;
        cen.B = p->B; cen.H = p->H; cen.N_o = p->N_o; cen.M = p->M; cen.thr = p->adj_thresh;
        cen.grip = p->gripper_offset; cen.grip_on = p->gripper_enable;
        int* d_cnt = sl.d_words + 8;                          // [0] touch at the first forward, [1] have a forward to run, [2] max repeat, [3] state words that differ
        cen.count = d_cnt;
        // is the base rollout of an earlier call still good?  Same model and task scalars: compared here; same start state:
        // compared bit for bit on the device ([3])
        const bool key_ok = c->base_cache_R >= 1 && !d_phys_vec && memcmp(&key_now, &c->base_key, sizeof key_now) == 0;
        const bool declined = c->decision.decline && c->decision.B == p->B && c->decision.H == p->H &&
                              memcmp(&key_now, &c->decision.key, sizeof key_now) == 0;
        if (key_ok) {
            // A base rollout is kept: census, state compare and the contact plan ALONG THE KEPT ROLLOUT go out together and the
            // call waits once.  (The planner calls dynamics() 40 times with one start state, plan.py:241-247: calls 2..40 come here.)
            HIPCHK(c, hipMemsetAsync(d_cnt, 0, 16, st));
            HIPCHK(c, launch_contact_plan(cen, st));
            HIPCHK(c, launch_count_diff(d_state0, c->d_base_cache, (long)p->N_o * 3, d_cnt + 3, st));
            ContactPlan cp{};
            cp.base_states = c->d_base_cache; cp.base_y = c->d_base_cache + (size_t)(c->base_cache_capR + 1) * p->N_o * 3;
            cp.R = c->base_cache_R; cp.R_bound = rep_bound;
            cp.eef_xz = d_eef_xz; cp.eef_delta = d_eef_delta; cp.repeat = d_rep_orig;
            cp.B = p->B; cp.H = p->H; cp.N_o = p->N_o; cp.M = p->M; cp.thr = p->adj_thresh; cp.rep_eff = b_rep_eff; cp.start = b_start;
            cp.state_seqs = d_state_seqs;
            HIPCHK(c, launch_contact_plan(cp, st));
            HIPCHK(c, hipMemcpyAsync(sl.h_rep_pin, b_rep_eff, nrep * 4, hipMemcpyDeviceToHost, st));
            if (dev_plan) HIPCHK(c, hipMemcpyAsync(sl.h_rep_pin + nrep, pl_repeat, nrep * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipMemcpyAsync(sl.h_rep_pin + 2 * nrep, d_overflow_flag, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipMemcpyAsync(sl.h_rep_pin + 2 * nrep + 1, d_cnt, 16, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipEventRecord(sl.ev_plan, st));
            HIPCHK(c, hipEventSynchronize(sl.ev_plan));
            const int* h_cnt = sl.h_rep_pin + 2 * nrep + 1;
            R_base = std::min(R_base, std::max(1, h_cnt[2]));     // the batch's own maximum (the device plan only knows the bound)
            census = true;
            if (h_cnt[3] == 0 && c->base_cache_R >= R_base) { base_cached = true; plan_done = true; }   // a kept base rollout is free: share
            else {
                // another start state (or a longer push than the kept rollout covers): what the plan above wrote is void
                if (d_state_seqs) HIPCHK(c, hipMemsetAsync(d_state_seqs, 0, (size_t)p->B * p->H * p->N_o * 3 * 4, st));
                if (h_cnt[1] - h_cnt[0] < std::max(64, 8 * R_base)) prefix = false;
            }
        } else if (declined) {
            // the last census of this shape found (nearly) every push on the object: no sharing, and no waiting either - a census
            // goes out that the call does not wait for (read by a later call, above)
            prefix = false;
            if (!sl.census_pending) {
                HIPCHK(c, hipMemsetAsync(d_cnt, 0, 16, st));
                HIPCHK(c, launch_contact_plan(cen, st));
                HIPCHK(c, hipMemcpyAsync(sl.h_census, d_cnt, 16, hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipEventRecord(sl.ev_census, st));
                sl.census_pending = true; sl.census_B = p->B; sl.census_H = p->H; sl.census_R = R_base;
                memcpy(&sl.census_key, &key_now, sizeof key_now);
            }
        } else {
            // one tiny kernel and one wait (for it and whatever the caller enqueued on this stream before the call)
            HIPCHK(c, hipMemsetAsync(d_cnt, 0, 16, st));
            HIPCHK(c, launch_contact_plan(cen, st));
            HIPCHK(c, hipMemcpyAsync(sl.h_census + 4, d_cnt, 16, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipEventRecord(sl.ev_plan, st));
            HIPCHK(c, hipEventSynchronize(sl.ev_plan));
            const int* h_cnt = sl.h_census + 4;
            R_base = std::min(R_base, std::max(1, h_cnt[2]));
            if (h_cnt[1] - h_cnt[0] < std::max(64, 8 * R_base)) prefix = false;
            census = true;
        }
        if (census) {
            c->decision.decline = !prefix;
            if (!prefix) { memcpy(&c->decision.key, &key_now, sizeof key_now); c->decision.B = p->B; c->decision.H = p->H; }
        }
    }
    const bool loop_dev = dev_plan && !prefix;               // the enqueue loop reads its live counts from the device plan's tables

    Work ws[ag_ctx::kMaxStreams] = {};
    for (int i = 0; i < ns; ++i) {
        rc = carve_work(c, sl.slab, ws[i], Ba, N, 1, edge_cap, edge_cap, slices, true, true, true, p->N_o, ell);
        if (rc) return rc;
    }
    const int* base_send = nullptr; const int* base_deg = nullptr; const float* C_share = nullptr;
    HIPCHK(c, hipMemsetAsync(sl.d_share_stats, 0, 16, st));   // counters of this call (ag_ctx_share_counts)
    if (share) {
        Slab& sb = sl.slab;
        float* b_C = sb.take<float>((size_t)base_cap * NFP);
        int* b_send = sb.take<int>(base_cap); int* b_recv = sb.take<int>(base_cap); int* b_ns = sb.take<int>(base_cap);
        float* b_node_in = sb.take<float>((size_t)p->N_o * NODE_IN);
        float* b_feat = sb.take<float>((size_t)p->N_o * F15_PITCH);
        float* b_group = sb.take<float>(p->N_o);
        int* b_deg = sb.take<int>(p->N_o);
        uint8_t* b_mask = sb.take<uint8_t>(p->N_o); uint8_t* b_tool = sb.take<uint8_t>(p->N_o);
        int* b_slice_tot = sb.take<int>(base_slices); int* b_cta = sb.take<int>(1);
        int* b_n_edges = sb.take<int>(1); int* b_n_ns = sb.take<int>(1);
        if (sb.used > sb.cap) return fail(c, AG_ERR_INVALID, "internal: workspace carve overflow");
        HIPCHK(c, launch_share_prep(d_state0, p->N_o, n_his, b_node_in, b_feat, b_group, b_mask, b_tool, st));
        EdgeArgs be{};
        be.pos = d_state0; be.pos_bstride = (long)p->N_o * 3; be.mask = b_mask; be.tool = b_tool; be.thr = p->adj_thresh;
        be.B = 1; be.N = p->N_o; be.topk = p->topk; be.cta = 0; be.edge_cap = base_cap; be.slices = base_slices;
        be.ell_full = 1; be.ell = b_send; be.ell_stride = kb; be.ell_bstride = base_cap; be.deg = b_deg;
        be.slice_tot = b_slice_tot; be.cta_flag = b_cta; be.recv = b_recv; be.send = b_send; be.n_edges = b_n_edges;
        be.ns_edge = b_ns; be.n_ns = b_n_ns; be.max_nR = 0x7fffffff; be.block_min_rows = c->opt.edge_block_min;
        HIPCHK(c, launch_edge_build(be, st, prof_mark, c));
        GraphBufs gb{};
        gb.node_in = b_node_in; gb.feat12 = b_feat; gb.group = b_group; gb.C = b_C; gb.recv = b_recv; gb.send = b_send;
        gb.n_edges = b_n_edges; gb.ns_edge = b_ns; gb.n_ns = b_n_ns; gb.B = 1; gb.N = p->N_o; gb.n_p = p->N_o; gb.n_inst = 1;
        gb.edge_cap = base_cap; gb.c_cap = base_cap; gb.n_his = n_his; gb.wb3 = c->precision == 1 ? c->d_wb3 : nullptr;
        gb.diag = c->diag;
        rc = run_edge_chain(c, gb, st);
        if (rc) return rc;
        base_send = b_send; base_deg = b_deg; C_share = b_C; c->d_share_nns = b_n_ns;
    }
    const int* d_start = nullptr; const float* d_base_states = nullptr; const float* d_base_y = nullptr;
    if (prefix) {
        if (base_in_ctx) {
            const size_t need = (size_t)(R_base + 1) * (p->N_o * 3 + 1);
            if (!base_cached) {
                // the kept rollout is about to be replaced: calls of other streams that still read it come first
                for (CallSlot& q : c->slots)
                    if (&q != &sl && q.bound && q.have_done) HIPCHK(c, hipStreamWaitEvent(st, q.ev_done, 0));
                c->base_cache_R = -1;
                if (c->base_cache_cap < need) {
                    if (c->d_base_cache) HIPCHK(c, dev_free(c, c->d_base_cache));
                    c->d_base_cache = nullptr; c->base_cache_cap = 0;
                    // (room for the longest push the caller's bound allows: a later call with longer pushes re-uses the buffer)
                    const size_t want = std::max(need, (size_t)((dev_plan ? R : R_base) + 1) * (p->N_o * 3 + 1));
                    HIPCHK(c, dev_alloc(c, reinterpret_cast<void**>(&c->d_base_cache), want * 4));
                    c->base_cache_cap = want;
                }
                c->base_cache_capR = (int)(c->base_cache_cap / (p->N_o * 3 + 1)) - 1;
            }
            b_states = c->d_base_cache; b_y = c->d_base_cache + (size_t)(c->base_cache_capR + 1) * p->N_o * 3;
        }
        if (!plan_done) {
        const float far = 1.0e6f;                            // out of every particle's reach; delta 0: it stays there
        int far_bits; memcpy(&far_bits, &far, 4);
        HIPCHK(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b_eef), far_bits, (size_t)2 * p->M, st));
        HIPCHK(c, hipMemsetAsync(b_eef + 2 * p->M, 0, (size_t)3 * p->M * 4, st));
        HIPCHK(c, hipMemsetAsync(b_zero, 0, 4, st));
        if (!base_cached) HIPCHK(c, hipMemcpyAsync(b_states, d_state0, (size_t)p->N_o * 3 * 4, hipMemcpyDeviceToDevice, st));   // S_0
        if (!base_cached) {   // ---- the base rollout: one candidate on workspace 0, R_base forwards, every state recorded
            Work& w = ws[0];
            GraphBufs g = w.g;
            g.B = 1; g.n_p = p->N_o; g.n_his = n_his; g.wb3 = c->precision == 1 ? c->d_wb3 : nullptr;
            if (dedupe) {
                g.c_self = c->d_cself; g.ns_edge = w.ns_edge; g.n_ns = w.n_ns; g.self_row = (long)Ba * edge_cap;
                HIPCHK(c, hipMemcpyAsync(w.g.C + (size_t)g.self_row * NFP, c->d_cself, 2 * NFP * 4, hipMemcpyDeviceToDevice, st));
            }
            RollArgs ra{};
            ra.B = 1; ra.B_slots = 1; ra.N_o = p->N_o; ra.M = p->M; ra.H = 1; ra.y_mode = 0; ra.b0 = 0; ra.li = 0; ra.ai = 0;
            ra.grip = p->gripper_offset; ra.grip_on = p->gripper_enable; ra.phys = p->physics_param; ra.phys_vec = d_phys_vec;
            ra.state0 = d_state0; ra.state0_batched = 0; ra.eef_xz = b_eef; ra.eef_delta = b_eef + 2 * p->M; ra.repeat = b_zero;
            ra.write_obj_cls = 1; ra.all_states = b_states; ra.all_y = b_y;
            EdgeArgs ea{};
            ea.pos = w.r.hist + (size_t)(n_his - 1) * N * 3; ea.pos_bstride = (long)n_his * N * 3;
            ea.mask = w.r.mask; ea.tool = w.r.tool; ea.thr = p->adj_thresh; ea.B = 1; ea.N = N; ea.topk = p->topk;
            ea.cta = p->connect_tools_all ? 1 : 0;             // (the parked tool has no object in reach: the rule's flag stays 0)
            ea.edge_cap = edge_cap; ea.slices = slices; ea.ell = w.ell; ea.deg = w.deg; ea.slice_tot = w.slice_tot; ea.cta_flag = w.cta_flag;
            ea.recv = w.recv; ea.send = w.send; ea.row_ptr = w.row_ptr; ea.n_edges = w.n_edges; ea.overflow = d_overflow_flag;
            ea.max_nR = p->max_nR; ea.zero_on_overflow = 1; ea.block_min_rows = c->opt.edge_block_min;
            if (ell_full) {
                ea.ell_full = 1; ea.ell = w.send; ea.ell_stride = k + p->M; ea.ell_bstride = edge_cap; ea.ns_edge = w.ns_edge; ea.n_ns = w.n_ns;
                g.deg = w.deg; g.ell_stride = k + p->M;
            }
            w.r.ragged = 0; w.r.clamp = c->dims.motion_clamp;
            { Scoped sc(c, FAM_ROLL_INIT); HIPCHK(c, launch_roll_init(ra, w.r, g, st)); }
            { Scoped sc(c, FAM_NODE_ENC); HIPCHK(c, node_enc_for(c, g, 0, 2L * p->N_o + p->M, st)); }
            for (int ai = 1; ai <= R_base; ++ai) {
                HIPCHK(c, launch_edge_build(ea, st, prof_mark, c));
                if (g.ns_edge && !ell_full) { Scoped sc(c, FAM_EDGE_EMIT); HIPCHK(c, launch_edge_nonself(w.recv, w.send, w.row_ptr, 1, N, edge_cap, w.ns_edge, w.n_ns, nullptr, st)); }
                rc = run_model(c, g, w.r.pred, w.r.motion, st);
                if (rc) return rc;
                ra.ai = ai;
                { Scoped sc(c, FAM_ROLL_UPDATE); HIPCHK(c, launch_roll_update(ra, w.r, g, st)); }
            }
        }
        // ---- contact plan -> forwards left per candidate, back on the host (the one wait of a prefix-sharing call)
        ContactPlan cp{};
        cp.base_states = b_states; cp.base_y = b_y; cp.R = R_base; cp.R_bound = rep_bound;
        cp.eef_xz = d_eef_xz; cp.eef_delta = d_eef_delta; cp.repeat = d_rep_orig;
        cp.B = p->B; cp.H = p->H; cp.N_o = p->N_o; cp.M = p->M; cp.thr = p->adj_thresh; cp.rep_eff = b_rep_eff; cp.start = b_start;
        cp.state_seqs = d_state_seqs;
        HIPCHK(c, launch_contact_plan(cp, st));
        HIPCHK(c, hipMemcpyAsync(sl.h_rep_pin, b_rep_eff, nrep * 4, hipMemcpyDeviceToHost, st));
        if (dev_plan) HIPCHK(c, hipMemcpyAsync(sl.h_rep_pin + nrep, pl_repeat, nrep * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(sl.h_rep_pin + 2 * nrep, d_overflow_flag, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipEventRecord(sl.ev_plan, st));
        HIPCHK(c, hipEventSynchronize(sl.ev_plan));
        }
        if (base_in_ctx && !base_cached) {
            // keep the base rollout for later calls - unless its graphs overflowed max_nR (that call must raise by itself).
            // (memcmp compares the keys, padding included: both sides are memset + field-wise filled and copied with memcpy; a
            // spurious mismatch could only cost a re-computation, never a wrong re-use)
            const bool clean = sl.h_rep_pin[2 * nrep] <= p->max_nR;
            c->base_cache_R = clean ? R_base : -1;
            memcpy(&c->base_key, &key_now, sizeof key_now);
        }
        if (dev_plan) {
            c->fwd_needed = 0;
            for (size_t i = 0; i < nrep; ++i) c->fwd_needed += std::min(std::max(0, sl.h_rep_pin[nrep + i]), R);
            c->d_plan_sums = nullptr;
        }
        c->fwd_executed = base_cached ? 0 : R_base;          // the base rollout's forwards (none when an earlier call's is re-used)
        if (!work_only) {
            rc = host_plan(sl.h_rep_pin);                     // launch order and sizes from the forwards that are LEFT
            if (rc) return rc;
        }
        d_start = b_start; d_base_states = b_states; d_base_y = b_y;
    }
    if (work_only) {
        // forwards candidate b would be stepped by the call this one stands for: what is left of look-ahead step 0 after its first
        // contact (prefix sharing in play) or all of it, plus the later steps' repeats, each at most the caller's bound
        if (!prefix) {
            HIPCHK(c, hipMemcpyAsync(sl.h_rep_pin, pl_repeat, nrep * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipEventRecord(sl.ev_plan, st));
            HIPCHK(c, hipEventSynchronize(sl.ev_plan));
        }
        for (int b = 0; b < p->B; ++b) {
            long w = 0;
            for (int li = 0; li < p->H; ++li) w += std::min(std::max(0, sl.h_rep_pin[(size_t)b * p->H + li]), R);
            src.h_work[b] = (int32_t)w;
        }
        c->d_plan_sums = nullptr;
        return AG_OK;
    }
    hipStream_t streams[ag_ctx::kMaxStreams] = {st, st, st, st};
    if (ns > 1) {
        HIPCHK(c, hipEventRecord(sl.ev_fork, st));            // inputs / memset / repeat upload are ordered before
        for (int i = 1; i < ns; ++i) {
            if (!sl.aux_stream[i]) {
                HIPCHK(c, stream_new(c, &sl.aux_stream[i]));
                HIPCHK(c, event_new(c, &sl.ev_join[i]));
            }
            streams[i] = sl.aux_stream[i];
            HIPCHK(c, hipStreamWaitEvent(sl.aux_stream[i], sl.ev_fork, 0));
        }
    }

    int fail_at = -1, timing_skip = 0;
#ifdef AG_DIAG   // AG_TEST_FAIL_AT_CHUNK=n (diagnostic build only): fail with AG_ERR_HIP before enqueuing chunk n, as a failed launch would
    fail_at = diag_fail_at_chunk(c->diag);
    timing_skip = diag_timing_skip(c->diag);         // AG_TIMING_SKIP (diagnostic build only): timing-only, wrong results
#endif
    // The chunk loop as a callable: whatever it returns, the forked streams are joined back into the caller's stream
    // below, so that a failure in the middle never leaves work of this call in flight on a stream the caller cannot see.
    auto enqueue_chunks = [&]() -> int {
    bool obj_cls_ready[ag_ctx::kMaxStreams] = {false, false, false, false};   // per workspace, per call
    bool plan_landed = false;                                // device plan: the chunk maxima are in sl.h_plan_max
    c->steps_enqueued = 0; c->steps_bound = 0;
    int ci = 0;
    for (int b0 = 0; b0 < p->B; b0 += Bc, ++ci) {
        if (ci == fail_at) return fail(c, AG_ERR_HIP, "test hook: injected failure before chunk %d", ci);
        const int nb = std::min(Bc, p->B - b0);
        Work& w = ws[ci % ns];
        hipStream_t cs = streams[ci % ns];
        c->prof_stream = cs;
        GraphBufs g = w.g;
        g.B = nb; g.n_p = p->N_o; g.n_his = n_his;
        g.wb3 = c->precision == 1 ? c->d_wb3 : nullptr;
        if (dedupe) {
            g.c_self = c->d_cself; g.ns_edge = w.ns_edge; g.n_ns = w.n_ns;
            g.self_row = (long)Ba * edge_cap;     // behind the last candidate's C rows of this workspace
            HIPCHK(c, hipMemcpyAsync(w.g.C + (size_t)g.self_row * NFP, c->d_cself, 2 * NFP * 4, hipMemcpyDeviceToDevice, cs));
        }
        RollArgs ra{};
        ra.B = nb; ra.B_slots = nb; ra.N_o = p->N_o; ra.M = p->M; ra.H = p->H; ra.y_mode = p->y_mode; ra.b0 = b0;
        ra.grip = p->gripper_offset; ra.grip_on = p->gripper_enable; ra.phys = p->physics_param; ra.phys_vec = d_phys_vec;
        ra.state0 = d_state0; ra.state0_batched = p->y_mode == 1; ra.obj_mask = d_obj_mask;
        ra.eef_xz = d_eef_xz; ra.eef_delta = d_eef_delta; ra.repeat = loop_dev ? pl_repeat : sl.d_repeat; ra.state_seqs = d_state_seqs;
        ra.start = d_start; ra.base_states = d_base_states; ra.base_y = d_base_y;
        EdgeArgs ea{};
        ea.pos = w.r.hist + (size_t)(n_his - 1) * N * 3; ea.pos_bstride = (long)n_his * N * 3;   // the newest frame
        ea.mask = w.r.mask; ea.tool = w.r.tool; ea.thr_vec = nullptr; ea.thr = p->adj_thresh;
        ea.B = nb; ea.N = N; ea.topk = p->topk; ea.cta = p->connect_tools_all ? 1 : 0; ea.edge_cap = edge_cap;
        ea.slices = slices; ea.ell = w.ell; ea.deg = w.deg; ea.slice_tot = w.slice_tot; ea.cta_flag = w.cta_flag;
        ea.recv = w.recv; ea.send = w.send; ea.row_ptr = w.row_ptr; ea.n_edges = w.n_edges;
        ea.overflow = d_overflow_flag; ea.max_nR = p->max_nR; ea.zero_on_overflow = 1; ea.block_min_rows = c->opt.edge_block_min;
        if (ell_full) {
            ea.ell_full = 1; ea.ell = w.send; ea.ell_stride = k + p->M; ea.ell_bstride = edge_cap;
            ea.ns_edge = w.ns_edge; ea.n_ns = w.n_ns;
            g.deg = w.deg; g.ell_stride = k + p->M;
        }
        w.r.ragged = ragged ? 1 : 0; w.r.clamp = c->dims.motion_clamp;
        if (ragged) {   // the mask does not change during a rollout: one work list per chunk and call, in slot order (H = 1)
            const int* d_cand0 = sort_on ? sl.d_repeat + nrep + b0 : nullptr;
            HIPCHK(c, launch_build_rowlist(d_obj_mask, d_cand0, b0, nb, p->N_o, p->M, w.rowlist, w.n_rows, w.r.mask, w.deg, cs));
            HIPCHK(c, hipMemsetAsync(w.row_ptr + (size_t)nb * (N + 1), 0, (size_t)(N + 1) * 4, cs));   // CSR path: no edges
            g.rowlist = w.rowlist; g.n_rows = w.n_rows + nb;
        }
        for (int li = 0; li < p->H; ++li) {
            const int* seg = loop_dev ? nullptr : h_cand + (size_t)li * p->B + b0;   // slot -> candidate of this chunk and look-ahead step
            int max_rep = loop_dev ? R : 0;                  // device plan: the caller's bound; steps past a chunk's own maximum find no live slot
            if (!loop_dev) for (int b = 0; b < nb; ++b) max_rep = std::max(max_rep, h_repeat[(size_t)seg[b] * p->H + li]);
            if (max_rep == 0 && !loop_dev) continue;          // nothing of this chunk is stepped in this look-ahead step
            ra.li = li; ra.ai = 0; ra.B = nb; ra.live = nullptr;
            ra.cand = loop_dev ? pl_cand + (size_t)li * p->B + b0 : sort_on ? sl.d_repeat + nrep + (size_t)li * p->B + b0 : nullptr;
            const int* live_row = loop_dev ? pl_live + ((size_t)ci * p->H + li) * (R + 2) : nullptr;
            const int* rows_row = loop_dev ? pl_rows + ((size_t)ci * p->H + li) * (R + 2) : nullptr;
            // masked variant: the object rows depend on nothing per-candidate either (both validity variants are
            // tabulated), so they are encoded once per call and workspace; tool rows once per look-ahead step
            ra.write_obj_cls = obj_cls_ready[ci % ns] ? 0 : 1;
            { Scoped s(c, FAM_ROLL_INIT); HIPCHK(c, launch_roll_init(ra, w.r, g, cs)); }
            { Scoped s(c, FAM_NODE_ENC);
              const long tool0 = 2L * p->N_o;
              if (!obj_cls_ready[ci % ns]) HIPCHK(c, node_enc_for(c, g, 0, tool0 + (long)nb * p->M, cs));
              else HIPCHK(c, node_enc_for(c, g, tool0, (long)nb * p->M, cs)); }
            obj_cls_ready[ci % ns] = true;
            int n_live = nb;
            c->steps_bound += max_rep;
            for (int ai = 1; ai <= max_rep; ++ai) {           // forward_dynamics.py:156
                if (loop_dev) {
                    // past this chunk's own maximum no slot is live: stop as soon as the plan's maxima are known (no waiting)
                    if (!plan_landed && ai > 1 && !capturing) {
                        if (hipEventQuery(sl.ev_plan) == hipSuccess) plan_landed = true;
                        else (void)hipGetLastError();       // "not ready" must not be taken for a failed launch by the next check
                    }
                    if (plan_landed && ai > sl.h_plan_max[(size_t)ci * p->H + li]) break;
                }
                ++c->steps_enqueued;
                if (loop_dev) {   // grids cover the whole chunk; the kernels read how many slots are live from the plan's table
                    ea.live = live_row + ai; ra.live = live_row + ai; g.n_rows = rows_row + ai;
                } else {
                    if (sort_on) while (n_live > 0 && h_repeat[(size_t)seg[n_live - 1] * p->H + li] < ai) --n_live;   // descending order: a prefix
                    c->fwd_executed += n_live;
                    if (ragged) g.n_rows = w.n_rows + n_live;   // rows of the live slots (+ the phantom candidate's)
                }
                ea.B = n_live; g.B = n_live; ra.B = n_live;
                // the call's first forward (start state, constant history): object-object C rows from the shared table
                const bool share_step = share && li == 0 && ai == 1 && !lat_node_for(c, g);
                ea.send_pk = share_step ? w.send_pk : nullptr; g.send_pk = ea.send_pk;
                if (share_step) {
                    ea.base_send = base_send; ea.base_deg = base_deg; ea.base_stride = kb; ea.share_No = p->N_o;
                    ea.share_stats = sl.d_share_stats; g.C_share = C_share; g.share_kb = kb;
                    ea.share_start = d_start; ea.share_cand = ra.cand; ea.share_b0 = b0;
                }
                if (!(timing_skip & 1) || ai == 1) {
                    HIPCHK(c, launch_edge_build(ea, cs, prof_mark, c));
                    if (g.ns_edge && !ell_full) { Scoped s(c, FAM_EDGE_EMIT); HIPCHK(c, launch_edge_nonself(w.recv, w.send, w.row_ptr, n_live, N, edge_cap, w.ns_edge, w.n_ns, ea.live, cs)); }
                }
                rc = run_model(c, g, w.r.pred, w.r.motion, cs);
                if (rc) return rc;
                ra.ai = ai;
                if (!(timing_skip & 2) || ai == max_rep) { Scoped s(c, FAM_ROLL_UPDATE); HIPCHK(c, launch_roll_update(ra, w.r, g, cs)); }
            }
        }
    }
    return AG_OK;
    };
    const int rc_loop = enqueue_chunks();
    int rc_join = AG_OK;
    for (int i = 1; i < ns; ++i) {                           // join on EVERY exit once the fork has happened
        hipError_t e = hipEventRecord(sl.ev_join[i], sl.aux_stream[i]);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, sl.ev_join[i], 0);
        if (e != hipSuccess && rc_join == AG_OK && rc_loop == AG_OK)
            rc_join = fail(c, AG_ERR_HIP, "joining stream %d failed: %s", i, hipGetErrorString(e));
    }
    c->prof_stream = st;
    return rc_loop ? rc_loop : rc_join;
}
}  // namespace

extern "C" {

int ag_rollout_async(ag_ctx* c, void* stream, const ag_rollout_params* p, const float* d_state0,
                     const uint8_t* d_obj_mask, const float* d_eef_xz, const float* d_eef_delta,
                     const int32_t* h_repeat, const float* d_phys_vec, float* d_state_seqs, int32_t* d_overflow_flag) {
    ActionSrc src;
    src.d_eef_xz = d_eef_xz; src.d_eef_delta = d_eef_delta; src.h_repeat = h_repeat;
    if (c && (!d_eef_xz || !d_eef_delta || !h_repeat)) return fail(c, AG_ERR_INVALID, "ag_rollout: null pointer");
    return rollout_impl(c, stream, p, d_state0, d_obj_mask, src, d_phys_vec, d_state_seqs, d_overflow_flag);
}

int ag_rollout_actions(ag_ctx* c, void* stream, const ag_rollout_params* p, const float* d_state0, const float* d_action,
                       float push_length, const float* h_tool_offsets, int32_t max_repeat, const float* d_phys_vec,
                       float* d_state_seqs, float* d_action_seqs, int32_t* d_flags) {
    if (!c) return AG_ERR_INVALID;
    if (!d_action || !d_action_seqs || !d_flags) return fail(c, AG_ERR_INVALID, "ag_rollout_actions: null pointer");
    ActionSrc src;
    src.d_action = d_action; src.push_length = push_length; src.h_tool_off = h_tool_offsets; src.max_repeat = max_repeat;
    src.d_action_seqs = d_action_seqs;
    return rollout_impl(c, stream, p, d_state0, nullptr, src, d_phys_vec, d_state_seqs, d_flags);
}

int ag_rollout_work(ag_ctx* c, void* stream, const ag_rollout_params* p, const float* d_state0, const float* d_action,
                    float push_length, const float* h_tool_offsets, int32_t max_repeat, const float* d_phys_vec, int32_t* h_work) {
    if (!c) return AG_ERR_INVALID;
    if (!p || !d_action || !h_work) return fail(c, AG_ERR_INVALID, "ag_rollout_work: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(c, hipSetDevice(c->device));
    CallSlot* sl = nullptr;
    int rc = slot_acquire(c, st, false, &sl);
    if (rc) return rc;
    SlotGuard slot_guard(sl, st, false);
    const size_t nrep = (size_t)p->B * p->H;
    // scratch for what the plan kernel writes besides the plan: decoded actions (B,H,4) and the two flag words
    if (sl->work_cap < nrep * 4 + 64) {
        if (sl->d_work) HIPCHK(c, dev_free(c, sl->d_work));
        sl->d_work = nullptr; sl->work_cap = 0;
        HIPCHK(c, dev_alloc(c, reinterpret_cast<void**>(&sl->d_work), (nrep * 4 + 64) * 4 * 2));
        sl->work_cap = (nrep * 4 + 64) * 2;
    }
    HIPCHK(c, hipMemsetAsync(sl->d_work, 0, 64 * 4, st));
    ActionSrc src;
    src.d_action = d_action; src.push_length = push_length; src.h_tool_off = h_tool_offsets; src.max_repeat = max_repeat;
    src.d_action_seqs = sl->d_work + 64; src.h_work = h_work;
    return rollout_impl(c, stream, p, d_state0, nullptr, src, d_phys_vec, nullptr, reinterpret_cast<int32_t*>(sl->d_work));
}

int ag_rollout(ag_ctx* c, void* stream, const ag_rollout_params* p, const float* d_state0, const uint8_t* d_obj_mask,
               const float* d_eef_xz, const float* d_eef_delta, const int32_t* h_repeat, const float* d_phys_vec,
               float* d_state_seqs) {
    if (!c) return AG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    CallSlot* sl = nullptr;
    int rc = slot_acquire(c, st, false, &sl);                 // (the call below finds the same slot: same stream)
    if (rc) return rc;
    SlotGuard slot_guard(sl, st, false);
    int* d_word = sl->d_words;
    HIPCHK(c, hipMemsetAsync(d_word, 0, 4, st));
    rc = ag_rollout_async(c, stream, p, d_state0, d_obj_mask, d_eef_xz, d_eef_delta, h_repeat, d_phys_vec, d_state_seqs, d_word);
    if (rc) return rc;
    int seen = 0;
    HIPCHK(c, hipMemcpyAsync(&seen, d_word, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (seen > p->max_nR) return fail(c, AG_ERR_MAX_NR, "Exceeds max dims: a graph had %d edges, max_nR=%d", seen, p->max_nR);
    return AG_OK;
}

int ag_cost_chamfer(ag_ctx* c, void* stream, const float* d_x, const float* d_y, const uint8_t* d_xmask,
                    const uint8_t* d_ymask, int32_t R, int32_t N, int32_t M, int32_t By, float* d_out) {
    if (!c) return AG_ERR_INVALID;
    if (!d_x || !d_y || !d_out || R < 1 || N < 1 || M < 1 || (By != 1 && By != R))
        return fail(c, AG_ERR_INVALID, "ag_cost_chamfer: bad arguments R=%d N=%d M=%d By=%d", R, N, M, By);
    if ((size_t)N + (size_t)M > chamfer_max_points())
        return fail(c, AG_ERR_UNSUPPORTED, "ag_cost_chamfer: N+M=%d exceeds the LDS tile (%zu points)", N + M, chamfer_max_points());
    HIPCHK(c, hipSetDevice(c->device));
    c->prof_stream = static_cast<hipStream_t>(stream);
    Scoped p(c, FAM_COST);
    HIPCHK(c, launch_chamfer(d_x, d_y, d_xmask, d_ymask, R, N, M, By, d_out, static_cast<hipStream_t>(stream)));
    return AG_OK;
}

int ag_cost_state_stats(ag_ctx* c, void* stream, const float* d_state, int32_t R, int32_t N, const float* h_box4,
                        float* d_out) {
    if (!c) return AG_ERR_INVALID;
    if (!d_state || !d_out || R < 1 || N < 1) return fail(c, AG_ERR_INVALID, "ag_cost_state_stats: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    c->prof_stream = static_cast<hipStream_t>(stream);
    Scoped p(c, FAM_COST);
    HIPCHK(c, launch_state_stats(d_state, R, N, h_box4, d_out, static_cast<hipStream_t>(stream)));
    return AG_OK;
}

int ag_cost_penalty(ag_ctx* c, void* stream, const float* d_state_pred, const float* d_action,
                    const float* d_state_init, int32_t B, int32_t H, int32_t N, int32_t kind, float ratio, float* d_out) {
    if (!c) return AG_ERR_INVALID;
    if (!d_state_pred || !d_action || !d_state_init || !d_out || B < 1 || H < 1 || N < 1)
        return fail(c, AG_ERR_INVALID, "ag_cost_penalty: bad arguments");
    if (kind < 0 || kind > 2) return fail(c, AG_ERR_UNSUPPORTED, "penalty kind %d not implemented", kind);
    HIPCHK(c, hipSetDevice(c->device));
    c->prof_stream = static_cast<hipStream_t>(stream);
    Scoped p(c, FAM_COST);
    HIPCHK(c, launch_penalty(d_state_pred, d_action, d_state_init, B, H, N, kind, ratio, d_out, static_cast<hipStream_t>(stream)));
    return AG_OK;
}

int ag_cost_reward(ag_ctx* c, void* stream, const float* d_error, const float* d_penalty, const float* d_stats,
                   const float* d_error_max, const double* h_bbox4, int32_t B, int32_t H, float* d_reward) {
    if (!c) return AG_ERR_INVALID;
    if (!d_error || !d_penalty || !d_stats || !h_bbox4 || !d_reward || B < 1 || H < 1)
        return fail(c, AG_ERR_INVALID, "ag_cost_reward: bad arguments B=%d H=%d", B, H);
    HIPCHK(c, hipSetDevice(c->device));
    c->prof_stream = static_cast<hipStream_t>(stream);
    Scoped p(c, FAM_COST);
    HIPCHK(c, launch_reward(d_error, d_penalty, d_stats, d_error_max, h_bbox4, B, H, d_reward, static_cast<hipStream_t>(stream)));
    return AG_OK;
}

int ag_cost_cloth_combine(ag_ctx* c, void* stream, const float* d_raw, const float* d_dmax, int64_t n, float* d_out) {
    if (!c) return AG_ERR_INVALID;
    if (!d_raw || !d_out || n < 1) return fail(c, AG_ERR_INVALID, "ag_cost_cloth_combine: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    c->prof_stream = static_cast<hipStream_t>(stream);
    Scoped p(c, FAM_COST);
    HIPCHK(c, launch_cloth_combine(d_raw, d_dmax, (long)n, d_out, static_cast<hipStream_t>(stream)));
    return AG_OK;
}

int ag_mppi_sample(ag_ctx* c, void* stream, const float* d_act_seq, const float* d_lo, const float* d_hi, const float* d_rnd,
                   const float* d_scale, int32_t S, int32_t H, int32_t mode, float push_length, float* d_out) {
    if (!c) return AG_ERR_INVALID;
    if (!d_lo || !d_hi || !d_rnd || !d_out || S < 1 || H < 1 || (mode != 0 && mode != 1) || (mode == 1 && (!d_act_seq || !d_scale)))
        return fail(c, AG_ERR_INVALID, "ag_mppi_sample: bad arguments S=%d H=%d mode=%d", S, H, mode);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_mppi_sample(d_act_seq, d_lo, d_hi, d_rnd, d_scale, S, H, mode, push_length, d_out, static_cast<hipStream_t>(stream)));
    return AG_OK;
}

int ag_mppi_update(ag_ctx* c, void* stream, const float* d_act_seqs, const float* d_reward, const float* d_lo,
                   const float* d_hi, int32_t B, int32_t H, float reward_weight, float push_length, float* d_out) {
    if (!c) return AG_ERR_INVALID;
    if (!d_act_seqs || !d_reward || !d_lo || !d_hi || !d_out || B < 1 || H < 1)
        return fail(c, AG_ERR_INVALID, "ag_mppi_update: bad arguments B=%d H=%d", B, H);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_mppi_update(d_act_seqs, d_reward, d_lo, d_hi, B, H, reward_weight, push_length, d_out, static_cast<hipStream_t>(stream)));
    return AG_OK;
}

int ag_mppi_clip(ag_ctx* c, void* stream, const float* d_in, const float* d_lo, const float* d_hi, int64_t n, float* d_out) {
    if (!c) return AG_ERR_INVALID;
    if (!d_in || !d_lo || !d_hi || !d_out || n < 1) return fail(c, AG_ERR_INVALID, "ag_mppi_clip: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_mppi_clip(d_in, d_lo, d_hi, d_out, (long)n, static_cast<hipStream_t>(stream)));
    return AG_OK;
}

int ag_ctx_set_profiling(ag_ctx* c, int32_t mask) {
    if (!c) return AG_ERR_INVALID;
    c->prof_mask = (unsigned)mask;
    return AG_OK;
}

static int prof_collect(ag_ctx* c) {
    for (auto& p : c->prof_live) {
        hipError_t e = hipEventSynchronize(p.e1);
        if (e != hipSuccess) return fail(c, AG_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, p.e0, p.e1);
        if (e != hipSuccess) return fail(c, AG_ERR_HIP, "hipEventElapsedTime: %s", hipGetErrorString(e));
        c->prof_ms[p.fam] += ms; c->prof_n[p.fam] += 1;
        c->prof_pool.push_back(p.e0); c->prof_pool.push_back(p.e1);
    }
    c->prof_live.clear();
    return AG_OK;
}

int ag_ctx_kernel_stats(ag_ctx* c, const char* kernel, double* out_ms, int64_t* out_n) {
    if (!c || !kernel || !out_ms || !out_n) return AG_ERR_INVALID;
    int rc = prof_collect(c);
    if (rc) return rc;
    for (int f = 0; f < FAM_COUNT; ++f)
        if (!strcmp(kernel, kFamilyNames[f])) { *out_ms = c->prof_ms[f]; *out_n = c->prof_n[f]; return AG_OK; }
    return fail(c, AG_ERR_INVALID, "unknown kernel family '%s'", kernel);
}

int ag_ctx_reset_stats(ag_ctx* c) {
    if (!c) return AG_ERR_INVALID;
    int rc = prof_collect(c);
    for (int f = 0; f < FAM_COUNT; ++f) { c->prof_ms[f] = 0; c->prof_n[f] = 0; }
    return rc;
}

}  // extern "C"
