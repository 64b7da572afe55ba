// Radius-AND-top-k graph construction with tool rules, CSR-by-receiver output.  gfx950 only.
//
// Replaces construct_edges_from_states_batch (reference src/dynamics/dataset/graph.py:233-298), which
// materialises (B,N,N,3) broadcasts, an N x N distance matrix, torch.topk, six N x N masks and dense one-hot
// Rr/Rs.  Here: one workgroup per (candidate, row slice); the candidate's positions live in LDS as SoA x/y/z;
// each wavefront owns receiver rows and sweeps senders 64 at a time (one sender per lane), so every distance is
// computed in registers and never stored.  Integer output must equal the reference bit-for-bit, so the fp32
// arithmetic is spelled exactly (SURVEY.md §8 a5'):
//     dis = ((dx*dx + dy*dy) + dz*dz)  with separate mul/add (no FMA contraction)      graph.py:251-252
//     thr2 = fp32(thr)*fp32(thr);  adjacent  <=>  (dis - thr2) < 0                      graph.py:248-250,267
//     masked pairs and tool-tool pairs get dis = 1e10                                    graph.py:253-260
//     top-k per RECEIVER row over the whole row, self-loops included                     graph.py:270-274
//     ties at the k-th boundary: (distance, sender index) lexicographic - torch.topk's own choice is
//     implementation-defined; fixtures assert no such tie inside the radius.
//     connect_tools_all rules                                                            graph.py:276-286
//     edge order = row-major nonzero = sorted by (receiver, sender)                      graph.py:293
//
// Two kernels, no inter-workgroup communication inside a launch:
//   k_edge_count : per row the k-th smallest key T* (selection in an LDS candidate buffer) and the row degree
//   k_edge_emit  : scan of degrees -> row_ptr, then a second sweep that writes (recv, send) in order
#include "ag_common.h"

namespace ag {

constexpr int EW = 1024;          // threads per workgroup (16 wavefronts)
constexpr int EWAVES = EW / 64;
constexpr int CAP = 256;          // candidate-buffer entries per wavefront
constexpr unsigned long long KEY_INF = ~0ull;

struct EdgeDev {
    const float* pos; long pos_bstride;  // floats between candidates
    const uint8_t* mask; const uint8_t* tool; const float* thr_vec; float thr;
    int B, N, k, topk_active, cta, edge_cap, slices, rows_per_slice;
    unsigned long long* tstar; int* deg; int* slice_tot; int* cta_flag;
    int* recv; int* send; int* row_ptr; int* n_edges; int* overflow; int max_nR; int zero_on_overflow;
};

__device__ __forceinline__ float dist_exact(float xi, float yi, float zi, float xj, float yj, float zj) {
    const float dx = __fsub_rn(xi, xj), dy = __fsub_rn(yi, yj), dz = __fsub_rn(zi, zj);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// LDS carve: x[N] y[N] z[N] floats | keys[EWAVES][CAP] u64 | flags[N] bytes | scan ints
struct EdgeLds {
    float* x; float* y; float* z; unsigned long long* keys; uint8_t* fl; int* misc;
};
__device__ __forceinline__ EdgeLds carve(unsigned char* base, int N) {
    EdgeLds l;
    const int Np = (N + 3) & ~3;
    l.keys = reinterpret_cast<unsigned long long*>(base);
    l.x = reinterpret_cast<float*>(base + (size_t)EWAVES * CAP * 8);
    l.y = l.x + Np;
    l.z = l.y + Np;
    l.misc = reinterpret_cast<int*>(l.z + Np);       // 64 ints
    l.fl = reinterpret_cast<uint8_t*>(l.misc + 64);
    return l;
}
static size_t edge_lds_bytes(int N) {
    const int Np = (N + 3) & ~3;
    return (size_t)EWAVES * CAP * 8 + (size_t)Np * 12 + 64 * 4 + (size_t)Np + 16;
}

__device__ __forceinline__ void load_candidate(const EdgeDev& a, const EdgeLds& l, int b) {
    const float* p = a.pos + (long)b * a.pos_bstride;
    for (int i = threadIdx.x; i < a.N; i += EW) {
        l.x[i] = p[3 * i + 0];
        l.y[i] = p[3 * i + 1];
        l.z[i] = p[3 * i + 2];
        l.fl[i] = (a.mask[(long)b * a.N + i] ? 1 : 0) | (a.tool[(long)b * a.N + i] ? 2 : 0);
    }
}

// distance key of pair (i, j) for lane's sender j; returns `within` and the 64-bit (dis, j) key
__device__ __forceinline__ bool pair_key(const EdgeLds& l, int N, float xi, float yi, float zi, int fi, int j, float thr2,
                                         unsigned long long& key, int& fj) {
    const bool vj = j < N;
    const int jj = vj ? j : 0;
    fj = l.fl[jj];
    float d = dist_exact(xi, yi, zi, l.x[jj], l.y[jj], l.z[jj]);
    if (!((fi & 1) && (fj & 1))) d = 1e10f;          // graph.py:253-256
    if ((fi & 2) && (fj & 2)) d = 1e10f;             // graph.py:257-260
    key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)jj;
    return vj && (__fsub_rn(d, thr2) < 0.0f);        // graph.py:267
}

// k-th smallest key among keys[0..nb) of this wavefront (nb <= CAP); optionally compacts the k smallest to the
// front in sorted order.  Keys are distinct (the sender index is in the low word).
__device__ __forceinline__ unsigned long long select_kth(unsigned long long* keys, int nb, int k, bool compact) {
    const int lane = lane_id();
    unsigned long long mine[CAP / 64];
    int rank[CAP / 64];
#pragma unroll
    for (int u = 0; u < CAP / 64; ++u) {
        const int e = lane + 64 * u;
        mine[u] = e < nb ? keys[e] : KEY_INF;
        rank[u] = 0;
    }
    for (int t = 0; t < nb; ++t) {
        const unsigned long long other = keys[t];    // same address in every lane: LDS broadcast
#pragma unroll
        for (int u = 0; u < CAP / 64; ++u) rank[u] += other < mine[u] ? 1 : 0;
    }
    wave_lds_sync();
    unsigned long long kth = 0;
#pragma unroll
    for (int u = 0; u < CAP / 64; ++u) {
        const int e = lane + 64 * u;
        const bool is_kth = e < nb && rank[u] == k - 1;
        const unsigned long long bal = __ballot(is_kth);
        if (bal) {
            const int src = __ffsll((long long)bal) - 1;
            const unsigned lo = __shfl((unsigned)(mine[u] & 0xffffffffull), src);
            const unsigned hi = __shfl((unsigned)(mine[u] >> 32), src);
            kth = ((unsigned long long)hi << 32) | lo;
        }
        if (compact && e < nb && rank[u] < k) keys[rank[u]] = mine[u];
    }
    wave_lds_sync();
    return kth;
}

// Pass A for receiver row i: T* = k-th smallest (dis, j) key among in-radius senders, or KEY_INF when the top-k
// constraint does not bind (k >= N, or at most k senders in radius).
__device__ unsigned long long row_tstar(const EdgeDev& a, const EdgeLds& l, int i, float thr2) {
    if (!a.topk_active) return KEY_INF;
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    unsigned long long* keys = l.keys + wave * CAP;
    const float xi = l.x[i], yi = l.y[i], zi = l.z[i];
    const int fi = l.fl[i];
    int cnt = 0, nb = 0;
    unsigned long long T = KEY_INF;
    for (int c0 = 0; c0 < a.N; c0 += 64) {
        unsigned long long key; int fj;
        const bool within = pair_key(l, a.N, xi, yi, zi, fi, c0 + lane, thr2, key, fj);
        const unsigned long long bw = __ballot(within);
        cnt += __popcll(bw);
        const bool push = within && key < T;
        const unsigned long long bp = __ballot(push);
        if (bp) {
            const int pos = nb + __popcll(bp & ((1ull << lane) - 1ull));
            if (push) keys[pos] = key;
            nb += __popcll(bp);
            wave_lds_sync();
            if (nb > CAP - 64) {                     // keep only the k smallest so far; tighten T
                T = select_kth(keys, nb, a.k, true);
                nb = a.k;
            }
        }
    }
    if (cnt <= a.k) return KEY_INF;
    return select_kth(keys, nb, a.k, false);
}

// membership of sender j (this lane) in the final adjacency row i
__device__ __forceinline__ bool member(const EdgeDev& a, const EdgeLds& l, int i, int j, float xi, float yi, float zi,
                                       int fi, float thr2, unsigned long long tstar, int flag) {
    unsigned long long key; int fj;
    const bool within = pair_key(l, a.N, xi, yi, zi, fi, j, thr2, key, fj);
    const bool kept = within && key <= tstar;        // radius AND top-k      graph.py:267-274
    if (!a.cta) return kept;
    if (j >= a.N) return false;
    if (fj & 2) return (fi & 1) && flag;             // tool sender, valid receiver: all-or-nothing  graph.py:284,286
    return kept && !(fi & 2);                        // tool receiver loses its object senders        graph.py:283,285
}

// connect_tools_all batch flag: does any tool receiver keep a non-tool sender after radius AND top-k?  graph.py:277
__device__ void compute_cta_flag(const EdgeDev& a, const EdgeLds& l, float thr2) {
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) l.misc[0] = 0;
    __syncthreads();
    if (a.cta) {
        for (int i = wave; i < a.N; i += EWAVES) {
            if (!(l.fl[i] & 2)) continue;            // wave-uniform
            const unsigned long long ts = row_tstar(a, l, i, thr2);
            const float xi = l.x[i], yi = l.y[i], zi = l.z[i];
            const int fi = l.fl[i];
            bool any = false;
            for (int c0 = 0; c0 < a.N; c0 += 64) {
                unsigned long long key; int fj;
                const bool within = pair_key(l, a.N, xi, yi, zi, fi, c0 + lane, thr2, key, fj);
                any |= within && key <= ts && !(fj & 2);
            }
            if (__ballot(any) && lane == 0) atomicOr(&l.misc[0], 1);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ float thr2_of(const EdgeDev& a, int b) {
    const float t = a.thr_vec ? a.thr_vec[b] : a.thr;
    return __fmul_rn(t, t);                          // graph.py:250 fp32 * fp32
}

__global__ __launch_bounds__(EW) void k_edge_count(EdgeDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x / a.slices, sl = blockIdx.x % a.slices;
    const EdgeLds l = carve(smem, a.N);
    load_candidate(a, l, b);
    __syncthreads();
    const float thr2 = thr2_of(a, b);
    compute_cta_flag(a, l, thr2);
    const int flag = l.misc[0];
    if (threadIdx.x == 0) { l.misc[1] = 0; if (sl == 0) a.cta_flag[b] = flag; }
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int r0 = sl * a.rows_per_slice;
    const int r1 = min(a.N, r0 + a.rows_per_slice);
    int my_tot = 0;
    for (int i = r0 + wave; i < r1; i += EWAVES) {
        const unsigned long long ts = row_tstar(a, l, i, thr2);
        const float xi = l.x[i], yi = l.y[i], zi = l.z[i];
        const int fi = l.fl[i];
        int d = 0;
        for (int c0 = 0; c0 < a.N; c0 += 64)
            d += __popcll(__ballot(member(a, l, i, c0 + lane, xi, yi, zi, fi, thr2, ts, flag)));
        if (lane == 0) {
            a.tstar[(long)b * a.N + i] = ts;
            a.deg[(long)b * a.N + i] = d;
        }
        my_tot += d;
    }
    if (lane == 0 && my_tot) atomicAdd(&l.misc[1], my_tot);   // integer add: order-independent
    __syncthreads();
    if (threadIdx.x == 0) a.slice_tot[b * a.slices + sl] = l.misc[1];
}

__global__ __launch_bounds__(EW) void k_edge_emit(EdgeDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x / a.slices, sl = blockIdx.x % a.slices;
    const EdgeLds l = carve(smem, a.N);
    load_candidate(a, l, b);
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int r0 = sl * a.rows_per_slice;
    const int r1 = min(a.N, r0 + a.rows_per_slice);
    const int nrows = max(0, r1 - r0);
    // keys area is reused as the int scan buffer for this slice's degrees (nrows <= N ints <= 16*256*2 ints? no:
    // only 8192 ints fit, so scan in registers per thread and keep per-thread bases in LDS instead)
    int* tbase = reinterpret_cast<int*>(l.keys);     // EW ints
    int base = 0, total = 0;
    for (int s = 0; s < a.slices; ++s) {
        const int v = a.slice_tot[b * a.slices + s];
        if (s < sl) base += v;
        total += v;
    }
    // each thread owns a contiguous run of rows of the slice
    const int per = (nrows + EW - 1) / EW;
    const int t0 = min(nrows, (int)threadIdx.x * per), t1 = min(nrows, t0 + per);
    int mysum = 0;
    for (int t = t0; t < t1; ++t) mysum += a.deg[(long)b * a.N + r0 + t];
    tbase[threadIdx.x] = mysum;
    __syncthreads();
    // exclusive scan over EW partial sums (Hillis-Steele in LDS, integers)
    for (int off = 1; off < EW; off <<= 1) {
        int v = 0;
        if ((int)threadIdx.x >= off) v = tbase[threadIdx.x - off];
        __syncthreads();
        tbase[threadIdx.x] += v;
        __syncthreads();
    }
    int run = base + tbase[threadIdx.x] - mysum;
    const bool fits = total <= a.edge_cap;
    const bool hide = !fits && a.zero_on_overflow;          // downstream kernels then see an empty graph
    for (int t = t0; t < t1; ++t) {
        a.row_ptr[(long)b * (a.N + 1) + r0 + t] = hide ? 0 : run;
        run += a.deg[(long)b * a.N + r0 + t];
    }
    if (sl == a.slices - 1 && threadIdx.x == 0) {
        a.row_ptr[(long)b * (a.N + 1) + a.N] = hide ? 0 : total;
        a.n_edges[b] = hide ? 0 : total;
        if (a.overflow && total > a.max_nR) atomicMax(a.overflow, total);
    }
    __syncthreads();   // row_ptr of this slice is complete and visible inside the workgroup
    if (!fits) return;
    const float thr2 = thr2_of(a, b);
    const int flag = a.cta_flag[b];
    for (int i = r0 + wave; i < r1; i += EWAVES) {
        const unsigned long long ts = a.tstar[(long)b * a.N + i];
        const float xi = l.x[i], yi = l.y[i], zi = l.z[i];
        const int fi = l.fl[i];
        int off = a.row_ptr[(long)b * (a.N + 1) + i];
        for (int c0 = 0; c0 < a.N; c0 += 64) {
            const bool m = member(a, l, i, c0 + lane, xi, yi, zi, fi, thr2, ts, flag);
            const unsigned long long bm = __ballot(m);
            if (m) {
                const long p = (long)b * a.edge_cap + off + __popcll(bm & ((1ull << lane) - 1ull));
                a.recv[p] = i;
                a.send[p] = c0 + lane;
            }
            off += __popcll(bm);
        }
    }
}

hipError_t launch_edge_build(const EdgeArgs& h, hipStream_t st, void (*mark)(void*, int, int), void* mark_ctx) {
    EdgeDev a;
    a.pos = h.pos; a.pos_bstride = h.pos_bstride; a.mask = h.mask; a.tool = h.tool; a.thr_vec = h.thr_vec; a.thr = h.thr;
    a.B = h.B; a.N = h.N; a.k = min(h.N, h.topk); a.topk_active = a.k < h.N; a.cta = h.cta; a.edge_cap = h.edge_cap;
    a.slices = h.slices; a.rows_per_slice = (h.N + h.slices - 1) / h.slices;
    a.tstar = h.tstar; a.deg = h.deg; a.slice_tot = h.slice_tot; a.cta_flag = h.cta_flag;
    a.recv = h.recv; a.send = h.send; a.row_ptr = h.row_ptr; a.n_edges = h.n_edges; a.overflow = h.overflow;
    a.max_nR = h.max_nR; a.zero_on_overflow = h.zero_on_overflow;
    const size_t lds = edge_lds_bytes(h.N);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_edge_count),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_edge_emit), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 256);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (mark) mark(mark_ctx, FAM_EDGE_COUNT, 0);
    hipLaunchKernelGGL(k_edge_count, dim3(h.B * h.slices), dim3(EW), lds, st, a);
    if (mark) mark(mark_ctx, FAM_EDGE_COUNT, 1);
    if (mark) mark(mark_ctx, FAM_EDGE_EMIT, 0);
    hipLaunchKernelGGL(k_edge_emit, dim3(h.B * h.slices), dim3(EW), lds, st, a);
    if (mark) mark(mark_ctx, FAM_EDGE_EMIT, 1);
    return hipGetLastError();
}

size_t edge_build_max_particles() { return 4096; }

}  // namespace ag
