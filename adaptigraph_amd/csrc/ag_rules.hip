// Tool-attachment rules of the single-graph edge builder, applied to an existing CSR edge list.  gfx950 only.
//
// construct_edges_from_states (reference src/dynamics/dataset/graph.py:68-231) has two optional rules that both
// have the form "given a particle SUBSET S (already restricted to valid particles)":
//     adj[tool receiver, sender in S]   = 0                                   graph.py:153 / :216
//     adj[receiver in S, tool sender]   = 1                                   graph.py:154 / :217
//     optional: of those (receiver in S, tool sender) pairs keep only the keepK = int(kNN * #pairs) with the
//               smallest distance, taken over the FLAT row-major list of pairs                  graph.py:156-169
//     adj[tool, tool]                   = 0                                   graph.py:170 / :218
// S = "non-fixed" particles (y above the bottom 10 %, graph.py:134-143) or the particles beyond the two surface
// planes closest to the tool (graph.py:190-207); forming S is scalar Python arithmetic and lives in the host shim.
// This file is the device mechanism: integer/byte work over one graph (eval-rollout path, B = 1), bit-exact.
//
//   k_rule_prep  : ascending tool index list, pair/keep counters
//   k_rule_dis   : fp32 distance of every (receiver in S, tool) pair, spelled like the edge builder's
//   k_rule_rank  : rank of each pair by (distance, flat index) -> keep flag            (only with 0 < kNN < 1)
//   k_rule_apply : per receiver row, merge the surviving base senders with the rule's tool senders in index order;
//                  count -> scan -> write (one workgroup, rows striped over its threads)
#include "ag_common.h"

namespace ag {

constexpr int RW = 1024;
constexpr float RULE_BIG = 1e10f;            // graph.py:92,96

struct RuleDev {
    const float* pos; const uint8_t* mask; const uint8_t* tool; const uint8_t* subset;
    const int* send_in; const int* row_ptr_in;
    int N, n_tools, edge_cap, use_knn; double kNN;
    int* tlist; int* misc;                   // misc: 0 ntool, 1 pair count, 2 tool count differs from the caller's n_tools
    float* pdis; uint8_t* keep; uint8_t* kept;   // (N, ntool) pair tables, row-major = the reference's flat order:
                                             // distance, 1 = pair of the rule / 2 = not, verdict of the kNN filter
    int* deg;                                // (N) output degrees
    int* recv; int* send; int* row_ptr; int* n_out;
};

__device__ __forceinline__ int block_excl_scan(int* sh, int v) {   // RW threads, returns exclusive prefix; sh[RW-1] = total
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < RW; off <<= 1) {
        int t = 0;
        if ((int)threadIdx.x >= off) t = sh[threadIdx.x - off];
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    return sh[threadIdx.x] - v;
}

__global__ __launch_bounds__(RW) void k_rule_prep(RuleDev a) {
    __shared__ int sh[RW];
    const int per = (a.N + RW - 1) / RW;
    const int j0 = min(a.N, (int)threadIdx.x * per), j1 = min(a.N, j0 + per);
    int c = 0;
    for (int j = j0; j < j1; ++j) c += a.tool[j] ? 1 : 0;
    int r = block_excl_scan(sh, c);
    for (int j = j0; j < j1; ++j)
        if (a.tool[j] && r < a.n_tools) a.tlist[r++] = j;             // the pair tables hold n_tools columns
    if (threadIdx.x == 0) { a.misc[0] = sh[RW - 1]; a.misc[1] = 0; a.misc[2] = sh[RW - 1] != a.n_tools; }
}

__device__ __forceinline__ bool in_subset(const RuleDev& a, int i) { return a.subset[i] && a.mask[i]; }

__global__ void k_rule_dis(RuleDev a) {
    const int M = a.misc[0];
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.misc[2] || M == 0 || p >= (long)a.N * M) return;
    const int i = (int)(p / M), j = a.tlist[p % M];
    if (!in_subset(a, i)) { a.keep[p] = 2; return; }                   // 2 = not a pair of the rule
    float d = RULE_BIG;
    if (a.mask[j] && !(a.tool[i] && a.tool[j])) {                      // graph.py:89-96
        const float dx = __fsub_rn(a.pos[3 * i], a.pos[3 * j]), dy = __fsub_rn(a.pos[3 * i + 1], a.pos[3 * j + 1]),
                    dz = __fsub_rn(a.pos[3 * i + 2], a.pos[3 * j + 2]);
        d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));   // graph.py:87-88
    }
    a.pdis[p] = d;
    a.keep[p] = 1;
    atomicAdd(&a.misc[1], 1);                                          // integer: order-independent
}

__global__ __launch_bounds__(256) void k_rule_rank(RuleDev a) {
    __shared__ float sd[1024];
    __shared__ uint8_t sk[1024];
    const int M = a.misc[0];
    const long L = (long)a.N * M;
    if (a.misc[2] || M == 0) return;                                    // uniform over the grid
    const int keepK = (int)(a.kNN * (double)a.misc[1]);                // graph.py:160 int(kNN * count)
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = p < L && a.keep[p] != 2;
    const float dp = mine ? a.pdis[p] : 0.0f;
    int rank = 0;
    for (long q0 = 0; q0 < L; q0 += 1024) {
        __syncthreads();
        for (int t = threadIdx.x; t < 1024; t += 256) {
            const long q = q0 + t;
            sk[t] = q < L ? a.keep[q] : 2;
            sd[t] = (q < L && sk[t] != 2) ? a.pdis[q] : 0.0f;
        }
        __syncthreads();
        if (mine) {
            const int n = (int)min(1024L, L - q0);
            for (int t = 0; t < n; ++t)
                rank += (sk[t] != 2 && (sd[t] < dp || (sd[t] == dp && q0 + t < p))) ? 1 : 0;
        }
    }
    if (mine) a.kept[p] = rank < keepK ? 1 : 0;
}

// One receiver row: walks the base senders (ascending) and the tool list (ascending) as one merged sequence.
template <bool WRITE>
__device__ __forceinline__ int rule_row(const RuleDev& a, int i, int M, int out) {
    const bool i_tool = a.tool[i], i_sub = in_subset(a, i);
    int e = a.row_ptr_in[i];
    const int e1 = a.row_ptr_in[i + 1];
    int m = 0, n = 0;
    while (e < e1 || m < M) {
        const int sj = e < e1 ? a.send_in[e] : 0x7fffffff;
        const int tj = m < M ? a.tlist[m] : 0x7fffffff;
        const int j = min(sj, tj);
        const bool base = sj == j;
        bool on;
        if (tj == j) {                                                  // tool sender
            on = base;
            if (i_sub) on = a.use_knn ? a.kept[(long)i * M + m] != 0 : true;   // graph.py:154,168 | :217
            if (i_tool) on = false;                                     // graph.py:170 | :218
            ++m;
        } else {
            on = base && !(i_tool && in_subset(a, j));                  // graph.py:153 | :216
        }
        if (base) ++e;
        if (on) {
            if (WRITE) { a.recv[out + n] = i; a.send[out + n] = j; }
            ++n;
        }
    }
    return n;
}

__global__ __launch_bounds__(RW) void k_rule_apply(RuleDev a) {
    __shared__ int sh[RW];
    if (a.misc[2]) {                                                    // caller's n_tools is wrong: refuse, loudly
        if (threadIdx.x == 0) *a.n_out = -1;
        return;
    }
    const int M = a.misc[0];
    const int per = (a.N + RW - 1) / RW;
    const int i0 = min(a.N, (int)threadIdx.x * per), i1 = min(a.N, i0 + per);
    int mine = 0;
    for (int i = i0; i < i1; ++i) {
        const int d = rule_row<false>(a, i, M, 0);
        a.deg[i] = d;
        mine += d;
    }
    int run = block_excl_scan(sh, mine);
    const int total = sh[RW - 1];
    const bool fits = total <= a.edge_cap;
    for (int i = i0; i < i1; ++i) {
        a.row_ptr[i] = run;
        if (fits) rule_row<true>(a, i, M, run);
        run += a.deg[i];
    }
    if (threadIdx.x == 0) { a.row_ptr[a.N] = total; *a.n_out = total; }   // TRUE count even when nothing was written
}

hipError_t launch_tool_rule(const RuleArgs& h, hipStream_t st) {
    RuleDev a{};
    a.pos = h.pos; a.mask = h.mask; a.tool = h.tool; a.subset = h.subset;
    a.send_in = h.send_in; a.row_ptr_in = h.row_ptr_in;
    a.N = h.N; a.n_tools = h.n_tools; a.edge_cap = h.edge_cap; a.use_knn = h.use_knn; a.kNN = h.kNN;
    a.tlist = h.tlist; a.misc = h.misc; a.pdis = h.pdis; a.keep = h.keep; a.kept = h.kept; a.deg = h.deg;
    a.recv = h.recv; a.send = h.send; a.row_ptr = h.row_ptr; a.n_out = h.n_out;
    hipLaunchKernelGGL(k_rule_prep, dim3(1), dim3(RW), 0, st, a);
    if (h.use_knn && h.n_tools > 0) {
        const long L = (long)h.N * h.n_tools;
        const unsigned grid = (unsigned)((L + 255) / 256);
        hipLaunchKernelGGL(k_rule_dis, dim3(grid), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_rule_rank, dim3(grid), dim3(256), 0, st, a);
    }
    hipLaunchKernelGGL(k_rule_apply, dim3(1), dim3(RW), 0, st, a);
    return hipGetLastError();
}

}  // namespace ag
