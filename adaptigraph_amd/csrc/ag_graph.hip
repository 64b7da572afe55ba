// Small HBM-bound kernels of the rollout step: model-input preparation, the guard for caller-built graphs and the per-step
// rollout bookkeeping (tool keypoints, history shift, capture).  gfx950 only.  (The edge->node message passing lives in
// ag_mlp.hip, fused into the propagate chains: gather_agg.)
#include "ag_common.h"
#include <algorithm>

namespace ag {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ caller-graph guard
// ag_forward consumes edge lists the CALLER built.  ag_build_edges reports the TRUE edge count even when it exceeds the
// caller's edge_cap and then writes no indices (pad_torch semantics, src/dynamics/utils.py:54-56): such a graph must not
// be walked.  n_eff[b] = n_edges[b] if it fits, else 0 (the kernels then see an empty graph) and *overflow = max count.
__global__ void k_edge_guard(const int* __restrict__ n_edges, int B, int edge_cap, int* __restrict__ n_eff, int* overflow) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int n = n_edges[b];
    const bool bad = n > edge_cap || n < 0;
    n_eff[b] = bad ? 0 : n;
    if (bad) atomicMax(overflow, n < 0 ? 0x7fffffff : n);
}
hipError_t launch_edge_guard(const int* n_edges, int B, int edge_cap, int* n_eff, int* overflow, hipStream_t st) {
    hipLaunchKernelGGL(k_edge_guard, dim3((B + 255) / 256), dim3(256), 0, st, n_edges, B, edge_cap, n_eff, overflow);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ forward() input prep
// state (B,n_his,N,3) -> feat12 rows [res0,res1,res2,cur] (model.py:156-166); node_in rows [attrs, phys, action, 1, 0]
// (model.py:169, 206-210, 223).
struct PrepDev {
    const float* state; const float* attrs; const float* action; const float* phys;
    float* node_in; float* feat12; int B, N;
};
template <int NH>
__global__ void k_prep(PrepDev p) {
    constexpr int FP = NH == 5 ? F15_PITCH : F12;
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (long)p.B * p.N) return;
    const int b = (int)(row / p.N), i = (int)(row - (long)b * p.N);
    float s[NH][3];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int c = 0; c < 3; ++c) s[h][c] = p.state[(((long)b * NH + h) * p.N + i) * 3 + c];
    float* f = p.feat12 + row * FP;
#pragma unroll
    for (int h = 0; h < NH - 1; ++h)
#pragma unroll
        for (int c = 0; c < 3; ++c) f[3 * h + c] = s[h + 1][c] - s[h][c];
#pragma unroll
    for (int c = 0; c < 3; ++c) f[3 * (NH - 1) + c] = s[NH - 1][c];
#pragma unroll
    for (int k = 3 * NH; k < FP; ++k) f[k] = 0.0f;
    float* n = p.node_in + row * NODE_IN;
    n[0] = p.attrs[row * 2 + 0]; n[1] = p.attrs[row * 2 + 1]; n[2] = p.phys[row];
    n[3] = p.action[row * 3 + 0]; n[4] = p.action[row * 3 + 1]; n[5] = p.action[row * 3 + 2];
    n[6] = 1.0f; n[7] = 0.0f;
}
hipError_t launch_prep(const float* state, const float* attrs, const float* action, const float* phys,
                       const GraphBufs& g, hipStream_t st) {
    PrepDev p{state, attrs, action, phys, g.node_in, g.feat12, g.B, g.N};
    const long rows = (long)g.B * g.N;
    if (g.n_his == 5) hipLaunchKernelGGL(k_prep<5>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(k_prep<4>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ rollout bookkeeping
struct RollDev {
    RollArgs a;
    float* hist; float* pred; uint8_t* mask; uint8_t* tool;
    const float* motion_inv; float clamp;   // ragged batches: constant motion of a masked-out particle per index, or null
    float* node_in; float* feat12; float* group; int n_inst;
    float* c_node_in; int write_obj_cls;   // class-table inputs (GraphBufs): tool rows every init, object rows on demand
};
constexpr int RT = 1024;

// tool height: min object y (forward_dynamics.py:40,163) or masked mean (forward_dynamics.py:235,359).
// `src` = (N_o,3) cloud of this candidate.  Result broadcast to the whole workgroup.
__device__ float tool_y(const float* src, const uint8_t* om, int N_o, int y_mode, float* red, int* redi) {
    const int tid = threadIdx.x;
    if (y_mode == 0) {
        // (a minimum does not depend on the order it is taken in: wave shuffles, then one value per wavefront - two barriers
        // instead of the ten of a tree over 1024 LDS slots; r06)
        float m = 3.4e38f;
        for (int i = tid; i < N_o; i += RT) m = fminf(m, src[3 * i + 1]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        float y = red[0];
#pragma unroll
        for (int w = 1; w < RT / 64; ++w) y = fminf(y, red[w]);
        __syncthreads();
        return y;
    }
    float s = 0.0f; int c = 0;
    for (int i = tid; i < N_o; i += RT) {
        const bool v = om ? om[i] != 0 : true;
        s += v ? src[3 * i + 1] : 0.0f;
        c += v ? 1 : 0;
    }
    red[tid] = s; redi[tid] = c;
    __syncthreads();
    for (int o = RT / 2; o > 0; o >>= 1) {
        if (tid < o) { red[tid] += red[tid + o]; redi[tid] += redi[tid + o]; }
        __syncthreads();
    }
    const float y = red[0] / (float)redi[0];
    __syncthreads();
    return y;
}

// Start of look-ahead step li (forward_dynamics.py:37-123 / 225-317): object cloud = start state (li == 0) or the
// captured prediction of step li-1; all n_his frames equal; tool keypoints from the decoded action; attrs, masks,
// p_instance, physics parameter, action rows.
template <int NH>
__global__ __launch_bounds__(RT) void k_roll_init(RollDev d) {
    constexpr int FP = NH == 5 ? F15_PITCH : F12;           // feature-row pitch: 3 * NH floats (+ pad for NH = 5)
    __shared__ float red[RT];
    __shared__ int redi[RT];
    const RollArgs& a = d.a;
    const int b = blockIdx.x, bg = a.cand ? a.cand[b] : a.b0 + b, tid = threadIdx.x;
    const int N = a.N_o + a.M;
    // Contact-free prefix (RollArgs::start): the slot starts from the BASE rollout's state `s0` - the candidate's tool touches
    // the object first at forward s0 + 1 - with the history the reference would hold there: object frames S_(s0-NH+1) .. S_s0
    // (clamped at the start state, :25), tool frames replayed with the very additions k_roll_update performs.
    const bool pre = a.start != nullptr && a.li == 0;
    const int s0 = pre ? a.start[bg] : 0;
    const float* src = pre ? a.base_states + (long)s0 * a.N_o * 3
                     : a.li == 0 ? (a.state0_batched ? a.state0 + (long)bg * a.N_o * 3 : a.state0)
                                 : a.state_seqs + ((long)bg * a.H + (a.li - 1)) * a.N_o * 3;
    const uint8_t* om = a.obj_mask ? a.obj_mask + (long)bg * a.N_o : nullptr;
    float y;
    if (pre) y = a.base_y[s0];                               // = tool_y(S_s0) (+ gripper offset), recorded by the base rollout
    else {
        y = tool_y(src, om, a.N_o, a.y_mode, red, redi);
        if (a.grip_on) y = y + a.grip;                       // :80-81
    }
    if (a.all_y && tid == 0) a.all_y[0] = y;                 // the base rollout records its tool heights
    int count = a.N_o;
    if (om) {                                                // first-`count` rows carry p_instance (:294-300)
        int c = 0;
        for (int i = tid; i < a.N_o; i += RT) c += om[i] ? 1 : 0;
        redi[tid] = c;
        __syncthreads();
        for (int o = RT / 2; o > 0; o >>= 1) { if (tid < o) redi[tid] += redi[tid + o]; __syncthreads(); }
        count = redi[0];
        __syncthreads();
    }
    for (int i = tid; i < N; i += RT) {
        float p[3], act[3] = {0.f, 0.f, 0.f};
        const bool is_tool = i >= a.N_o;
        if (!is_tool) {
            p[0] = src[3 * i]; p[1] = src[3 * i + 1]; p[2] = src[3 * i + 2];
        } else {
            const int m = i - a.N_o;
            const float* xz = a.eef_xz + (((long)bg * a.H + a.li) * a.M + m) * 2;
            const float* dl = a.eef_delta + (((long)bg * a.H + a.li) * a.M + m) * 3;
            p[0] = xz[0]; p[1] = y; p[2] = xz[1];
            act[0] = dl[0]; act[1] = dl[1]; act[2] = dl[2];
        }
        const long row = (long)b * N + i;
        float* f = d.feat12 + row * FP;
#pragma unroll
        for (int c = 0; c < FP; ++c) f[c] = 0.0f;            // identical frames: residuals are exactly 0 (and the row's pad)
        if (!pre || s0 == 0) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int c = 0; c < 3; ++c) d.hist[(((long)b * NH + h) * N + i) * 3 + c] = p[c];
        } else {
            float fr[NH][3];                                 // frame k = time max(0, s0 - (NH-1-k))
            if (!is_tool) {
#pragma unroll
                for (int k = 0; k < NH; ++k) {
                    const float* sp = a.base_states + ((long)max(0, s0 - (NH - 1 - k)) * a.N_o + i) * 3;
                    fr[k][0] = sp[0]; fr[k][1] = sp[1]; fr[k][2] = sp[2];
                }
            } else {
                // time 0 = the keypoint of the decoded action (p holds it with the height of time s0: x, z are what count);
                // every step adds the per-step delta (forward_dynamics.py:164) - the same fp32 additions, in the same order
                float x = p[0], z = p[2];
                const int t_first = s0 - (NH - 1);
#pragma unroll
                for (int k = 0; k < NH; ++k) if (t_first + k <= 0) { fr[k][0] = x; fr[k][1] = a.base_y[0]; fr[k][2] = z; }
                for (int t = 1; t <= s0; ++t) {
                    x = x + act[0]; z = z + act[2];
#pragma unroll
                    for (int k = 0; k < NH; ++k) if (t_first + k == t) { fr[k][0] = x; fr[k][1] = a.base_y[t]; fr[k][2] = z; }
                }
            }
#pragma unroll
            for (int k = 0; k < NH; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) d.hist[(((long)b * NH + k) * N + i) * 3 + c] = fr[k][c];
#pragma unroll
            for (int k = 0; k < NH - 1; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) f[3 * k + c] = fr[k + 1][c] - fr[k][c];        // model.py:156, as k_roll_update forms it
            p[0] = fr[NH - 1][0]; p[1] = fr[NH - 1][1]; p[2] = fr[NH - 1][2];
        }
        f[3 * (NH - 1)] = p[0]; f[3 * (NH - 1) + 1] = p[1]; f[3 * (NH - 1) + 2] = p[2];
        const bool ov = is_tool ? false : (om ? om[i] != 0 : true);
        float* n = d.node_in + row * NODE_IN;
        n[0] = ov ? 1.0f : 0.0f;                             // attrs[:, :nobj, 0] (:92 / :287)
        n[1] = is_tool ? 1.0f : 0.0f;                        // attrs[:, nobj:, 1] (:93)
        n[2] = is_tool ? 0.0f : (a.phys_vec ? a.phys_vec[i] : a.phys);                      // model.py:197,206-207
        n[3] = act[0]; n[4] = act[1]; n[5] = act[2];         // states_delta (:87-88)
        n[6] = 1.0f; n[7] = 0.0f;
        for (int k = 0; k < d.n_inst; ++k) d.group[row * d.n_inst + k] = (k == 0 && !is_tool && i < count) ? 1.0f : 0.0f;
        d.mask[row] = (is_tool || ov) ? 1 : 0;               // state_mask (:107-109 / :302-304)
        d.tool[row] = is_tool ? 1 : 0;                       // eef_mask (:111-112)
        if (d.c_node_in) {                                   // class-table inputs (see GraphBufs)
            if (is_tool) {
                float* cn = d.c_node_in + (2L * a.N_o + (long)b * a.M + (i - a.N_o)) * NODE_IN;
#pragma unroll
                for (int k = 0; k < NODE_IN; ++k) cn[k] = n[k];
            } else if (d.write_obj_cls && b == 0) {          // same for every candidate: candidate 0 writes both variants
                float* cv = d.c_node_in + (long)i * NODE_IN;
                float* ci = d.c_node_in + ((long)a.N_o + i) * NODE_IN;
                cv[0] = 1.0f; ci[0] = 0.0f;
#pragma unroll
                for (int k = 1; k < NODE_IN; ++k) { cv[k] = n[k]; ci[k] = n[k]; }
            }
        }
    }
}

// After forward number ai of look-ahead step li (forward_dynamics.py:160-176 / 356-372): capture, tool advance,
// history shift, history features for the next forward.
template <int NH>
__global__ __launch_bounds__(RT) void k_roll_update(RollDev d) {
    constexpr int FP = NH == 5 ? F15_PITCH : F12;
    __shared__ float red[RT];
    __shared__ int redi[RT];
    const RollArgs& a = d.a;
    if (a.live && (int)blockIdx.x >= *a.live) return;         // device-planned rollout: no forward was run for this slot
    const int b = blockIdx.x, bg = a.cand ? a.cand[b] : a.b0 + b, tid = threadIdx.x;
    const int N = a.N_o + a.M;
    float* pred = d.pred + (long)b * a.N_o * 3;
    const uint8_t* om = a.obj_mask ? a.obj_mask + (long)bg * a.N_o : nullptr;
    if (d.motion_inv && om) {
        // ragged batch: the chains skipped the masked-out particles.  What the reference computes for them (model.py:338:
        // last position + clamped motion; they receive no edge, so the motion is a constant of the model per particle
        // index) was computed once on the phantom candidate's rows: apply it here.
        for (int i = tid; i < a.N_o; i += RT) {
            if (om[i]) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float cur = d.hist[(((long)b * NH + (NH - 1)) * N + i) * 3 + c];
                pred[3 * i + c] = cur + fminf(fmaxf(d.motion_inv[3 * i + c], -d.clamp), d.clamp);
            }
        }
        __syncthreads();
    }
    if (a.repeat[(long)bg * a.H + a.li] == a.ai) {           // :160-161
        float* out = a.state_seqs + ((long)bg * a.H + a.li) * a.N_o * 3;
        for (int k = tid; k < a.N_o * 3; k += RT) out[k] = pred[k];
    }
    float y = tool_y(pred, om, a.N_o, a.y_mode, red, redi);  // :163 / :359
    if (a.grip_on) y = y + a.grip;                           // :167-168
    if (a.all_states) {                                      // the base rollout of the prefix sharing: S_ai and its tool height
        float* out = a.all_states + (long)a.ai * a.N_o * 3;
        for (int k = tid; k < a.N_o * 3; k += RT) out[k] = pred[k];
        if (tid == 0) a.all_y[a.ai] = y;
    }
    for (int i = tid; i < N; i += RT) {
        const long row = (long)b * N + i;
        float h[NH][3];
#pragma unroll
        for (int k = 0; k < NH; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) h[k][c] = d.hist[(((long)b * NH + k) * N + i) * 3 + c];
        float nw[3];
        if (i < a.N_o) {
            nw[0] = pred[3 * i]; nw[1] = pred[3 * i + 1]; nw[2] = pred[3 * i + 2];     // :170
        } else {
            const float* act = d.node_in + row * NODE_IN + 3;
            nw[0] = h[NH - 1][0] + act[0];                                          // :164
            nw[1] = y;                                                                 // :166
            nw[2] = h[NH - 1][2] + act[2];
        }
#pragma unroll
        for (int k = 0; k < NH - 1; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) h[k][c] = h[k + 1][c];                         // :176
#pragma unroll
        for (int c = 0; c < 3; ++c) h[NH - 1][c] = nw[c];
#pragma unroll
        for (int k = 0; k < NH; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) d.hist[(((long)b * NH + k) * N + i) * 3 + c] = h[k][c];
        float* f = d.feat12 + row * FP;
#pragma unroll
        for (int k = 0; k < NH - 1; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) f[3 * k + c] = h[k + 1][c] - h[k][c];          // model.py:156
#pragma unroll
        for (int c = 0; c < 3; ++c) f[3 * (NH - 1) + c] = h[NH - 1][c];
    }
}

// Work list of a ragged batch (GraphBufs::rowlist).  One workgroup.  Layout: first - only if some particle of the chunk is
// masked out - the N_o object rows of the phantom candidate (slot B), then for every slot s = 0..B-1 (in slot order, i.e. the
// repeat-sorted launch order) its valid object particles and its tools.  tab[n] = number of entries when only the first n slots
// are live (phantom rows included), n = 0..B: the launches of a step take their row count from tab + n_live.  `cand`: slot ->
// candidate of the full batch (null: b0 + slot).  Also clears what the chains read of the phantom candidate (validity mask,
// in-degrees).
__global__ __launch_bounds__(RT) void k_build_rowlist(const uint8_t* __restrict__ obj_mask, const int* __restrict__ cand, int b0,
                                                       int B, int N_o, int M, int* __restrict__ rowlist, int* __restrict__ tab,
                                                       uint8_t* __restrict__ mask, int* __restrict__ deg) {
    __shared__ int scan[RT];
    __shared__ int base;
    __shared__ int any_invalid;
    const int tid = threadIdx.x, N = N_o + M;
    if (tid == 0) { base = 0; any_invalid = 0; }
    for (int i = tid; i < N; i += RT) { mask[(long)B * N + i] = 0; if (deg) deg[(long)B * N + i] = 0; }
    __syncthreads();
    const long total = (long)B * N;
    int bad = 0;
    for (long r = tid; r < total; r += RT) {
        const int b = (int)(r / N), i = (int)(r - (long)b * N);
        if (i < N_o && !obj_mask[(long)(cand ? cand[b] : b0 + b) * N_o + i]) bad = 1;
    }
    if (bad) any_invalid = 1;                                // (benign race: every writer stores 1)
    __syncthreads();
    const int n_ph = any_invalid ? N_o : 0;
    for (int i = tid; i < n_ph; i += RT) rowlist[i] = (int)(total + i);
    if (tid == 0) { base = n_ph; tab[0] = n_ph; }
    __syncthreads();
    for (long r0 = 0; r0 < total; r0 += RT) {
        const long r = r0 + tid;
        int v = 0;
        if (r < total) {
            const int b = (int)(r / N), i = (int)(r - (long)b * N);
            v = (i >= N_o || obj_mask[(long)(cand ? cand[b] : b0 + b) * N_o + i]) ? 1 : 0;
        }
        scan[tid] = v;
        __syncthreads();
        for (int off = 1; off < RT; off <<= 1) {
            int t = 0;
            if (tid >= off) t = scan[tid - off];
            __syncthreads();
            scan[tid] += t;
            __syncthreads();
        }
        if (v) rowlist[base + scan[tid] - 1] = (int)r;
        // the last row of a slot closes that slot's prefix
        if (r < total && (r + 1) % N == 0) tab[(r + 1) / N] = base + scan[tid];
        __syncthreads();
        if (tid == RT - 1) base += scan[RT - 1];
        __syncthreads();
    }
}
hipError_t launch_build_rowlist(const uint8_t* obj_mask, const int* cand, int b0, int B, int N_o, int M, int* rowlist, int* tab,
                                uint8_t* mask, int* deg, hipStream_t st) {
    hipLaunchKernelGGL(k_build_rowlist, dim3(1), dim3(RT), 0, st, obj_mask, cand, b0, B, N_o, M, rowlist, tab, mask, deg);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ device-side launch plan
// See RollPlan (ag_common.h).  fp32 operations spelled in the reference's order (plan_utils.py:13-16: end = start -
// push_length * (cos, sin); forward_dynamics.py:48-75: delta = end - start, side keypoints start +- c_k * (sin, cos)); the
// transcendental functions are the device's (a GPU-resident reference would use the device's too), so decoded values agree
// with a CPU decode to an ulp or two, not bit for bit.
constexpr int PLAN_MAXR = 1024;
__global__ __launch_bounds__(64) void k_roll_plan(RollPlan p) {
    __shared__ int hist[PLAN_MAXR + 2];
    __shared__ int base[PLAN_MAXR + 2];
    const int lane = threadIdx.x;
    const int n_chunks = (p.B + p.Bc - 1) / p.Bc;
    const int ci = blockIdx.x % n_chunks, li = blockIdx.x / n_chunks;
    const int b0 = ci * p.Bc, nb = min(p.Bc, p.B - b0);
    const int R = p.max_repeat;
    for (int v = lane; v <= R + 1; v += 64) hist[v] = 0;
    __syncthreads();
    int over = 0;
    for (int t = lane; t < nb; t += 64) {
        const long bh = (long)(b0 + t) * p.H + li;
        const float* a = p.action + bh * 4;
        const float x = a[0], z = a[1], th = a[2], len = a[3];
        const float c = cosf(th), s = sinf(th);
        const float xe = x - p.push_length * c, ze = z - p.push_length * s;            // plan_utils.py:13-15
        float* dc = p.decoded + bh * 4;
        dc[0] = x; dc[1] = z; dc[2] = xe; dc[3] = ze;
        int rep = (int)len;                                                            // .to(int32): truncation (:16)
        p.repeat[bh] = rep;
        over = max(over, rep);
        rep = min(max(rep, 0), R);
        atomicAdd(&hist[rep], 1);                                                      // integer counts: order-free
        const float dx = xe - x, dz = ze - z;                                          // forward_dynamics.py:48-50
        for (int k = 0; k < p.M; ++k) {
            float* xz = p.eef_xz + (bh * p.M + k) * 2;
            float* dl = p.eef_delta + (bh * p.M + k) * 3;
            xz[0] = k == 0 ? x : x + p.tool_off[k] * s;                                // :60-75
            xz[1] = k == 0 ? z : z - p.tool_off[k] * c;
            dl[0] = dx; dl[1] = 0.0f; dl[2] = dz;
        }
    }
    for (int o = 1; o < 64; o <<= 1) over = max(over, __shfl_xor(over, o, 64));
    if (lane == 0 && over > R) atomicMax(p.flags + 1, over);
    __syncthreads();
    if (lane == 0) {
        // base[v] = candidates with a larger repeat (descending order); live[ai] = candidates with repeat >= ai
        int acc = 0, sum_rep = 0, sum_live = 0;
        int* live = p.live + ((long)ci * p.H + li) * (R + 2);
        int* rows = p.rows + ((long)ci * p.H + li) * (R + 2);
        live[R + 1] = 0; rows[R + 1] = 0;
        for (int v = R; v >= 0; --v) {
            base[v] = acc;
            acc += hist[v];                                  // candidates with repeat >= v
            // sorted: the live candidates are the first `acc` slots; unsorted: every slot while any candidate is live
            live[v] = v == 0 ? nb : (p.sort ? acc : (acc > 0 ? nb : 0));
            rows[v] = live[v] * p.N;
            sum_rep += v * hist[v];
            if (v >= 1) sum_live += live[v];
        }
        p.sums[((long)ci * p.H + li) * 2 + 0] = sum_rep;
        p.sums[((long)ci * p.H + li) * 2 + 1] = sum_live;
        int mx = 0;
        for (int v = 1; v <= R; ++v) if (hist[v] > 0) mx = v;
        p.maxrep[(long)ci * p.H + li] = mx;
    }
    __syncthreads();
    for (int v = lane; v <= R; v += 64) hist[v] = 0;         // reused as the running fill of every bucket
    __syncthreads();
    int* seg = p.cand + (long)li * p.B + b0;
    for (int t0 = 0; t0 < nb; t0 += 64) {                    // tiles in order, buckets filled in candidate order: stable
        const int t = t0 + lane;
        const bool on = t < nb;
        const int rep = on ? min(max(p.repeat[(long)(b0 + t) * p.H + li], 0), R) : -1;
        if (!p.sort) { if (on) seg[t] = b0 + t; continue; }
        unsigned long long todo = __ballot(on);
        while (todo) {
            const int first = __ffsll((long long)todo) - 1;
            const int v = __shfl(rep, first, 64);
            const unsigned long long m = __ballot(on && rep == v);
            if (on && rep == v) seg[base[v] + hist[v] + __popcll(m & ((1ull << lane) - 1ull))] = b0 + t;
            __builtin_amdgcn_wave_barrier();
            if (lane == first) hist[v] += __popcll(m);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            todo &= ~m;
        }
    }
}
hipError_t launch_roll_plan(const RollPlan& p, hipStream_t st) {
    if (p.max_repeat > PLAN_MAXR) return hipErrorInvalidValue;
    const int n_chunks = (p.B + p.Bc - 1) / p.Bc;
    hipLaunchKernelGGL(k_roll_plan, dim3(n_chunks * p.H), dim3(64), 0, st, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ contact-free prefix
// See ContactPlan (ag_common.h).  One workgroup per candidate.  The pair test is the edge builder's (ag_edges.hip: dist_exact,
// pair_within): dis = ((dx*dx + dy*dy) + dz*dz) with separate roundings, adjacent <=> (dis - thr*thr) < 0 (graph.py:248-267).
__device__ __forceinline__ float contact_dis(float xi, float yi, float zi, float xj, float yj, float zj) {
    const float dx = __fsub_rn(xi, xj), dy = __fsub_rn(yi, yj), dz = __fsub_rn(zi, zj);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
constexpr int CT = 256;
__global__ __launch_bounds__(CT) void k_contact_plan(ContactPlan p) {
    __shared__ float red[CT];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (p.count) {                                           // census of the first forward (see ContactPlan::count)
        const int rep = p.repeat[(long)b * p.H];
        if (rep < 1) return;                                 // (uniform over the workgroup)
        float m = 3.4e38f;
        for (int i = tid; i < p.N_o; i += CT) m = fminf(m, p.base_states[3 * i + 1]);
        red[tid] = m;
        __syncthreads();
        for (int o = CT / 2; o > 0; o >>= 1) { if (tid < o) red[tid] = fminf(red[tid], red[tid + o]); __syncthreads(); }
        float ty = red[0];                                   // a minimum: the same bits in any order (k_roll_init: tool_y)
        if (p.grip_on) ty = ty + p.grip;
        const float thr2 = __fmul_rn(p.thr, p.thr);
        int hit = 0;
        for (int i = tid; i < p.N_o; i += CT) {
            const float x = p.base_states[3 * i], y = p.base_states[3 * i + 1], z = p.base_states[3 * i + 2];
            for (int m2 = 0; m2 < p.M; ++m2) {
                const float* xz = p.eef_xz + ((long)b * p.H * p.M + m2) * 2;
                hit |= __fsub_rn(contact_dis(x, y, z, xz[0], ty, xz[1]), thr2) < 0.0f ? 1 : 0;
            }
        }
        hit = __syncthreads_or(hit);
        if (tid == 0) { if (hit) atomicAdd(p.count, 1); atomicAdd(p.count + 1, 1); atomicMax(p.count + 2, rep); }
        return;
    }
    for (int li = 1 + tid; li < p.H; li += CT) p.rep_eff[(long)b * p.H + li] = min(p.repeat[(long)b * p.H + li], p.R_bound);
    if (p.repeat[(long)b * p.H] > p.R_bound) {               // beyond the caller's bound: not stepped, not captured (rows stay zero)
        if (tid == 0) { p.rep_eff[(long)b * p.H] = 0; p.start[b] = 0; }
        return;
    }
    const int rep = min(max(p.repeat[(long)b * p.H], 0), p.R);
    const float thr2 = __fmul_rn(p.thr, p.thr);
    float tx[8], tz[8], dx[8], dz[8];
    for (int m = 0; m < p.M; ++m) {
        const float* xz = p.eef_xz + ((long)b * p.H * p.M + m) * 2;          // look-ahead step 0
        const float* dl = p.eef_delta + ((long)b * p.H * p.M + m) * 3;
        tx[m] = xz[0]; tz[m] = xz[1]; dx[m] = dl[0]; dz[m] = dl[2];
    }
    int d = 0;
    for (int ai = 1; ai <= rep; ++ai) {                      // the graph of forward ai: objects S_(ai-1), tool after ai-1 advances
        const float* S = p.base_states + (long)(ai - 1) * p.N_o * 3;
        const float ty = p.base_y[ai - 1];
        int hit = 0;
        for (int i = tid; i < p.N_o; i += CT) {
            const float x = S[3 * i], y = S[3 * i + 1], z = S[3 * i + 2];
            for (int m = 0; m < p.M; ++m)
                hit |= __fsub_rn(contact_dis(x, y, z, tx[m], ty, tz[m]), thr2) < 0.0f ? 1 : 0;
        }
        if (__syncthreads_or(hit)) { d = ai; break; }
        for (int m = 0; m < p.M; ++m) { tx[m] = tx[m] + dx[m]; tz[m] = tz[m] + dz[m]; }   // forward_dynamics.py:164
    }
    if (tid == 0) {
        p.rep_eff[(long)b * p.H] = d ? rep - d + 1 : 0;
        p.start[b] = d ? d - 1 : 0;
    }
    if (!d && rep >= 1 && p.state_seqs) {                    // never touched: the capture of forward `rep` is the base state S_rep
        const float* S = p.base_states + (long)rep * p.N_o * 3;
        float* out = p.state_seqs + (long)b * p.H * p.N_o * 3;
        for (int k = tid; k < p.N_o * 3; k += CT) out[k] = S[k];
    }
}
__global__ void k_count_diff(const unsigned* __restrict__ a, const unsigned* __restrict__ b, long n, int* __restrict__ count) {
    int d = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d += a[i] != b[i] ? 1 : 0;
    if (d) atomicAdd(count, d);
}
hipError_t launch_count_diff(const float* a, const float* b, long n, int* count, hipStream_t st) {
    hipLaunchKernelGGL(k_count_diff, dim3((unsigned)std::min<long>(64, (n + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const unsigned*>(a), reinterpret_cast<const unsigned*>(b), n, count);
    return hipGetLastError();
}
hipError_t launch_contact_plan(const ContactPlan& p, hipStream_t st) {
    if (p.M > 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_contact_plan, dim3(p.B), dim3(CT), 0, st, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ shared first forward
// Inputs of the once-per-call edge encode (GraphBufs::C_share): what k_roll_init<NH> writes for the OBJECT rows of every
// candidate at look-ahead step 0 (forward_dynamics.py:25: one start state, n_his equal frames) - attrs (1,0), group 1,
// residuals exactly 0, current position = the start state - for the N_o object particles alone, all valid, no tool.
__global__ void k_share_prep(const float* __restrict__ state0, int N_o, int fp, int cur_off, float* __restrict__ node_in,
                             float* __restrict__ feat, float* __restrict__ group, uint8_t* __restrict__ mask,
                             uint8_t* __restrict__ tool) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N_o) return;
    float* n = node_in + (long)i * NODE_IN;
    n[0] = 1.0f; n[1] = 0.0f; n[2] = 0.0f; n[3] = 0.0f; n[4] = 0.0f; n[5] = 0.0f; n[6] = 1.0f; n[7] = 0.0f;   // the edge chain reads [0], [1]
    float* f = feat + (long)i * fp;
    for (int c = 0; c < fp; ++c) f[c] = 0.0f;
    f[cur_off] = state0[3 * i]; f[cur_off + 1] = state0[3 * i + 1]; f[cur_off + 2] = state0[3 * i + 2];
    group[i] = 1.0f;
    mask[i] = 1; tool[i] = 0;
}
hipError_t launch_share_prep(const float* state0, int N_o, int n_his, float* node_in, float* feat, float* group, uint8_t* mask,
                             uint8_t* tool, hipStream_t st) {
    hipLaunchKernelGGL(k_share_prep, dim3((unsigned)((N_o + 255) / 256)), dim3(256), 0, st, state0, N_o, feat_pitch(n_his),
                       3 * (n_his - 1), node_in, feat, group, mask, tool);
    return hipGetLastError();
}

static RollDev to_dev(const RollArgs& a, const RollBufs& r, const GraphBufs& g) {
    RollDev d;
    d.a = a; d.hist = r.hist; d.pred = r.pred; d.mask = r.mask; d.tool = r.tool;
    d.motion_inv = r.ragged ? r.motion + (long)a.B_slots * a.N_o * 3 : nullptr; d.clamp = r.clamp;   // the phantom slot's rows
    d.node_in = g.node_in; d.feat12 = g.feat12; d.group = g.group; d.n_inst = g.n_inst;
    d.c_node_in = g.cls_on ? g.c_node_in : nullptr; d.write_obj_cls = a.write_obj_cls;
    return d;
}
// g.n_his: history frames of the model (4: every planner task config; 5: config/dynamics/softbody.yaml:29)
hipError_t launch_roll_init(const RollArgs& a, const RollBufs& r, const GraphBufs& g, hipStream_t st) {
    if (g.n_his == 5) hipLaunchKernelGGL(k_roll_init<5>, dim3(a.B), dim3(RT), 0, st, to_dev(a, r, g));
    else hipLaunchKernelGGL(k_roll_init<4>, dim3(a.B), dim3(RT), 0, st, to_dev(a, r, g));
    return hipGetLastError();
}
hipError_t launch_roll_update(const RollArgs& a, const RollBufs& r, const GraphBufs& g, hipStream_t st) {
    if (g.n_his == 5) hipLaunchKernelGGL(k_roll_update<5>, dim3(a.B), dim3(RT), 0, st, to_dev(a, r, g));
    else hipLaunchKernelGGL(k_roll_update<4>, dim3(a.B), dim3(RT), 0, st, to_dev(a, r, g));
    return hipGetLastError();
}

}  // namespace ag
