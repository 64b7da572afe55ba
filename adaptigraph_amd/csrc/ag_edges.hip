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
// Work avoidance that cannot change the result:
//   * senders are taken in chunks of 64 consecutive indices; each chunk has an axis-aligned bounding box of its
//     valid particles.  A chunk is skipped for receiver i when fl(x_i - max_x) >= thr (or the mirrored / y / z
//     tests): rounding is monotonic, so every sender j of the chunk has |fl(x_i - x_j)| >= thr, hence
//     fl(dx*dx) >= fl(thr*thr) = thr2 and dis >= thr2 - exactly the pairs the reference's test rejects.  One
//     lane tests one chunk, a ballot gives the survivor mask.  (Pays off when consecutive particle indices are
//     spatially coherent, e.g. grids and ropes; costs ~one sweep otherwise.)
//   * with top-k active, the in-radius candidates of a row are collected once into an LDS buffer; the k-th
//     smallest (dis, j) key is found by rank counting; the kept senders go to a per-row ELL scratch (<= k
//     entries) and the emit kernel is a scan plus a merge-copy (tool senders of connect_tools_all are merged in
//     index order).  Without top-k (topk >= N) rows are unbounded, so the emit kernel sweeps again.
//
// Two kernels, no inter-workgroup communication inside a launch:
//   k_edge_count : per-row kept list (ELL) + degree, slice totals, connect_tools_all flag
//   k_edge_emit  : scan of degrees -> row_ptr, then writes (recv, send) in order
#include "ag_common.h"
#include <cstdlib>

namespace ag {

constexpr int EW = 1024;          // threads per workgroup (16 wavefronts).  256-thread workgroups that could sit beside
                                  // the MLP chains of another stream were measured: 1.8x slower alone, no net gain.
constexpr int EWAVES = EW / 64;
constexpr int CAP = 256;          // candidate-buffer entries per wavefront
constexpr int MAXCH = 64;         // sender chunks per candidate (N <= 4096)
constexpr unsigned long long KEY_INF = ~0ull;

struct EdgeDev {
    const float* pos; long pos_bstride;  // floats between candidates
    const uint8_t* mask; const uint8_t* tool; const float* thr_vec; float thr;
    float thr2_override; int use_thr2;   // single-graph builder: threshold squared in double, then rounded (graph.py:86,101)
    int B, N, k, topk_active, cta, edge_cap, slices, rows_per_slice;
    int* ell;                            // (B, N, k) kept non-merged senders per row (top-k active only)
    int ell_full, ell_stride; long ell_bstride;   // see EdgeArgs
    int* ns_edge; int* n_ns;
    int* deg; int* slice_tot; int* cta_flag;
    int* recv; int* send; int* row_ptr; int* n_edges; int* overflow; int max_nR; int zero_on_overflow;
    int block_min_rows;                  // slices with at least this many rows take the 64-rows-per-wavefront path
    const int* live;                     // see EdgeArgs
    int* send_pk; const int* base_send; const int* base_deg; int base_stride, share_No;   // see EdgeArgs (first forward)
    unsigned long long* share_stats;
    const int* share_start; const int* share_cand; int share_b0;
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
__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// LDS carve
struct EdgeLds {
    unsigned long long* keys;   // [EWAVES][CAP]                                   (count kernel)
    int* scan;                  // [EW]                                            (emit kernel)
    unsigned short* tprefix;    // [Np+4] number of tools with index < j           (emit kernel)
    unsigned short* tlist;      // [Np]   tool indices in ascending order          (emit kernel)
    float* x; float* y; float* z;
    float* bb;                  // [6][MAXCH] chunk boxes: minx maxx miny maxy minz maxz
    int* misc;                  // [64]  0: cta flag, 1: slice total, 2: tool count
    uint8_t* fl;                // [Np] bit0 valid, bit1 tool
};
__host__ __device__ inline size_t edge_lds_bytes(int N) {
    const size_t Np = (size_t)((N + 3) & ~3), Nc = (size_t)((N + 63) & ~63);   // Nc: positions, whole sender chunks
    return (size_t)EWAVES * CAP * 8 + EW * 4 + (2 * Np + 8) * 2 + Nc * 12 + 6 * MAXCH * 4 + 64 * 4 + Np + 16;
}
__device__ __forceinline__ EdgeLds carve(unsigned char* base, int N) {
    EdgeLds l;
    const int Np = (N + 3) & ~3, Nc = (N + 63) & ~63;
    l.keys = reinterpret_cast<unsigned long long*>(base);
    l.scan = reinterpret_cast<int*>(base + (size_t)EWAVES * CAP * 8);
    l.x = reinterpret_cast<float*>(l.scan + EW);
    l.y = l.x + Nc;
    l.z = l.y + Nc;
    l.bb = l.z + Nc;
    l.misc = reinterpret_cast<int*>(l.bb + 6 * MAXCH);
    l.tprefix = reinterpret_cast<unsigned short*>(l.misc + 64);
    l.tlist = l.tprefix + Np + 4;
    l.fl = reinterpret_cast<uint8_t*>(l.tlist + Np + 4);
    return l;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// positions + flags -> LDS, chunk boxes, tool count.  Ends with a barrier.
__device__ void load_candidate(const EdgeDev& a, const EdgeLds& l, int b, bool want_pos) {
    const float* p = a.pos + (long)b * a.pos_bstride;
    if (threadIdx.x < 64) l.misc[threadIdx.x] = 0;
#pragma unroll 2
    for (int i = threadIdx.x; i < a.N; i += EW) {           // (two trips' loads in flight together: a cloth-sized graph takes two)
        if (want_pos) {
            l.x[i] = p[3 * i + 0];
            l.y[i] = p[3 * i + 1];
            l.z[i] = p[3 * i + 2];
        }
        l.fl[i] = (a.mask[(long)b * a.N + i] ? 1 : 0) | (a.tool[(long)b * a.N + i] ? 2 : 0);
    }
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int nch = (a.N + 63) >> 6;
    int ntool = 0;
    for (int c = wave; c < nch; c += EWAVES) {
        const int j = 64 * c + lane;
        const bool inb = j < a.N;
        const int f = inb ? l.fl[j] : 0;
        ntool += __popcll(__ballot(inb && (f & 2)));
        if (want_pos) {
            const bool v = inb && (f & 1);
            const float INF = __builtin_huge_valf();
            const float xj = v ? l.x[j] : 0.f, yj = v ? l.y[j] : 0.f, zj = v ? l.z[j] : 0.f;
            const float mnx = wave_min(v ? xj : INF), mxx = wave_max(v ? xj : -INF);
            const float mny = wave_min(v ? yj : INF), mxy = wave_max(v ? yj : -INF);
            const float mnz = wave_min(v ? zj : INF), mxz = wave_max(v ? zj : -INF);
            if (lane == 0) {
                l.bb[0 * MAXCH + c] = mnx; l.bb[1 * MAXCH + c] = mxx; l.bb[2 * MAXCH + c] = mny;
                l.bb[3 * MAXCH + c] = mxy; l.bb[4 * MAXCH + c] = mnz; l.bb[5 * MAXCH + c] = mxz;
            }
        }
    }
    if (lane == 0 && ntool) atomicAdd(&l.misc[2], ntool);
    if (want_pos) {
        // invalid senders and the pad up to a whole chunk are parked at x = +inf: their distance to any receiver is
        // +inf (or NaN), which fails the adjacency test like the reference's 1e10 (graph.py:253-256) - the block sweep
        // (block_topk) then needs no validity flag per pair.  Disjoint from what the box pass above reads (valid j only).
        const int Nc = (a.N + 63) & ~63;
        for (int j = threadIdx.x; j < Nc; j += EW)
            if (j >= a.N || !(l.fl[j] & 1)) { l.x[j] = __builtin_huge_valf(); l.y[j] = 0.f; l.z[j] = 0.f; }
    }
    __syncthreads();
}

// chunks that can contain a sender within the radius of (xi,yi,zi): bit c set = must sweep chunk c
__device__ __forceinline__ unsigned long long survivors(const EdgeLds& l, int N, float xi, float yi, float zi, float thr) {
    thr = fabsf(thr);                                      // thr enters the adjacency test only as thr*thr
    const int lane = lane_id();
    const int nch = (N + 63) >> 6;
    bool keep = false;
    if (lane < nch) {
        const float mnx = l.bb[0 * MAXCH + lane], mxx = l.bb[1 * MAXCH + lane];
        const float mny = l.bb[2 * MAXCH + lane], mxy = l.bb[3 * MAXCH + lane];
        const float mnz = l.bb[4 * MAXCH + lane], mxz = l.bb[5 * MAXCH + lane];
        // exact rejection (see header): every |fl(c_i - c_j)| >= thr.  Empty chunks have min=+inf, max=-inf.
        const bool out = (__fsub_rn(xi, mxx) >= thr) || (__fsub_rn(mnx, xi) >= thr) ||
                         (__fsub_rn(yi, mxy) >= thr) || (__fsub_rn(mny, yi) >= thr) ||
                         (__fsub_rn(zi, mxz) >= thr) || (__fsub_rn(mnz, zi) >= thr);
        keep = !out;
    }
    return __ballot(keep);
}

// lane's sender j against receiver i: distance with the reference's masking; returns `within`
__device__ __forceinline__ bool pair_within(const EdgeLds& l, int N, float xi, float yi, float zi, int fi, int j,
                                            float thr2, float& d, int& fj) {
    const bool vj = j < N;
    const int jj = vj ? j : 0;
    fj = l.fl[jj];
    d = dist_exact(xi, yi, zi, l.x[jj], l.y[jj], l.z[jj]);
    if (!(fj & 1) || ((fi & 2) && (fj & 2))) d = 1e10f;    // graph.py:253-260 (receiver validity handled by caller)
    return vj && (__fsub_rn(d, thr2) < 0.0f);              // graph.py:267
}

template <int U>
__device__ __forceinline__ unsigned long long select_kth_slots(unsigned long long* keys, int nb, int k, bool compact) {
    const int lane = lane_id();
    unsigned long long mine[U];
    int rank[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = lane + 64 * u;
        mine[u] = e < nb ? keys[e] : KEY_INF;
        rank[u] = 0;
    }
#pragma unroll 4
    for (int t = 0; t < nb; ++t) {
        const unsigned long long other = keys[t];          // same address in every lane: LDS broadcast
#pragma unroll
        for (int u = 0; u < U; ++u) rank[u] += other < mine[u] ? 1 : 0;
    }
    wave_lds_sync();
    unsigned long long kth = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
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

// rank counting over keys[0..nb): returns the k-th smallest key; optionally compacts the k smallest to the front.
__device__ __forceinline__ unsigned long long select_kth(unsigned long long* keys, int nb, int k, bool compact) {
    const int lane = lane_id();
    if (nb <= 64 && !compact) {                            // common case: one key per lane
        const unsigned long long mine = lane < nb ? keys[lane] : KEY_INF;
        int rank = 0;
        for (int t = 0; t < nb; ++t) rank += keys[t] < mine ? 1 : 0;
        const unsigned long long bal = __ballot(lane < nb && rank == k - 1);
        const int src = __ffsll((long long)bal) - 1;
        const unsigned lo = __shfl((unsigned)(mine & 0xffffffffull), src);
        const unsigned hi = __shfl((unsigned)(mine >> 32), src);
        return ((unsigned long long)hi << 32) | lo;
    }
    // general case: U = ceil(nb / 64) keys per lane (r05: the loops run over the slots that hold keys - a rope row has 60..120
    // in-radius senders, i.e. two slots, and paid for four; slots beyond nb hold KEY_INF and never matter)
    switch ((nb + 63) >> 6) {
    case 1: return select_kth_slots<1>(keys, nb, k, compact);
    case 2: return select_kth_slots<2>(keys, nb, k, compact);
    case 3: return select_kth_slots<3>(keys, nb, k, compact);
    default: return select_kth_slots<CAP / 64>(keys, nb, k, compact);
    }
}

// One receiver row, top-k active.  Collects in-radius senders, applies top-k, writes the kept senders that are not
// governed by the connect_tools_all rule to ell_row (ascending sender index) and returns their count.
// *nontool_raw receives the number of kept NON-tool senders before the tool-receiver rule (feeds graph.py:277).
__device__ int row_topk(const EdgeDev& a, const EdgeLds& l, int i, float thr, float thr2, int* ell_row, int* nontool_raw) {
    const int lane = lane_id();
    unsigned long long* keys = l.keys + (threadIdx.x >> 6) * CAP;
    const float xi = l.x[i], yi = l.y[i], zi = l.z[i];
    const int fi = l.fl[i];
    *nontool_raw = 0;
    if (!(fi & 1)) return 0;                               // invalid receiver: every pair is masked (graph.py:256)
    int cnt = 0, nb = 0;
    bool reordered = false;
    unsigned long long T = KEY_INF;
    unsigned long long todo = survivors(l, a.N, xi, yi, zi, thr);
    while (todo) {
        const int c = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        float d; int fj;
        const int j = 64 * c + lane;
        const bool within = pair_within(l, a.N, xi, yi, zi, fi, j, thr2, d, fj);
        const unsigned long long bw = __ballot(within);
        if (!bw) continue;
        cnt += __popcll(bw);
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j;
        const bool push = within && key < T;
        const unsigned long long bp = __ballot(push);
        if (bp) {
            if (push) keys[nb + __popcll(bp & lanes_below(lane))] = key;
            nb += __popcll(bp);
            wave_lds_sync();
            if (nb > CAP - 64) {                           // keep only the k smallest so far; tighten T
                T = select_kth(keys, nb, a.k, true);
                nb = a.k;
                reordered = true;
            }
        }
    }
    const unsigned long long tstar = cnt > a.k ? select_kth(keys, nb, a.k, false) : KEY_INF;
    // kept = key <= tstar.  Rank the kept entries by sender index (the buffer is already ascending unless a
    // compaction reordered it) and write them out.
    int total = 0, raw = 0;
    if (!reordered) {
        for (int e0 = 0; e0 < nb; e0 += 64) {
            const int e = e0 + lane;
            const unsigned long long key = e < nb ? keys[e] : KEY_INF;
            const int j = (int)(key & 0xffffffffull);
            const bool kept = e < nb && key <= tstar;
            const bool jt = kept && (l.fl[j] & 2);
            raw += __popcll(__ballot(kept && !jt));
            const bool out = kept && (!a.cta || (!jt && !(fi & 2)));   // graph.py:283-286 handled at emit
            const unsigned long long bo = __ballot(out);
            if (out) ell_row[total + __popcll(bo & lanes_below(lane))] = j;
            total += __popcll(bo);
        }
    } else {
        unsigned mine[CAP / 64]; bool outm[CAP / 64]; int pos[CAP / 64];
#pragma unroll
        for (int u = 0; u < CAP / 64; ++u) {
            const int e = lane + 64 * u;
            const unsigned long long key = e < nb ? keys[e] : KEY_INF;
            mine[u] = (unsigned)(key & 0xffffffffull);
            const bool kept = e < nb && key <= tstar;
            const bool jt = kept && (l.fl[mine[u]] & 2);
            raw += __popcll(__ballot(kept && !jt));
            outm[u] = kept && (!a.cta || (!jt && !(fi & 2)));
            pos[u] = 0;
        }
        for (int t = 0; t < nb; ++t) {
            const unsigned long long ko = keys[t];
            const unsigned jo = (unsigned)(ko & 0xffffffffull);
            const bool oo = ko <= tstar && (!a.cta || (!(l.fl[jo] & 2) && !(fi & 2)));
#pragma unroll
            for (int u = 0; u < CAP / 64; ++u) pos[u] += (oo && jo < mine[u]) ? 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < CAP / 64; ++u) {
            if (outm[u]) ell_row[pos[u]] = (int)mine[u];
            total += __popcll(__ballot(outm[u]));
        }
    }
    *nontool_raw = raw;
    return total;
}

// membership of sender j (this lane) in the final adjacency row i when top-k is NOT active
__device__ __forceinline__ bool member_radius(const EdgeDev& a, const EdgeLds& l, int i, int j, float xi, float yi,
                                              float zi, int fi, float thr2, int flag) {
    float d; int fj;
    const bool within = (fi & 1) && pair_within(l, a.N, xi, yi, zi, fi, j, thr2, d, fj);
    if (!a.cta) return within;
    if (j >= a.N) return false;
    if (fj & 2) return (fi & 1) && flag && !(a.cta == 2 && (fi & 2));   // graph.py:284,286 | single-graph: :121-122
    return within && !(fi & 2);                            // graph.py:283,285 | :120
}

// ---- 64 receiver rows per wavefront (one per lane), top-k active.  The per-row path above spends ~400 wave
// instructions of control per row for ~250 useful pair tests; here the control is per BLOCK of 64 consecutive rows:
//   * chunk culling with the bounding box of the block's receivers: fl(x_i - max_x) >= fl(rmin_x - max_x) >= thr for
//     every receiver of the block (rounding is monotonic), so a chunk culled for the box is culled for each receiver
//     by the exact per-receiver argument in the header; surviving chunks are a superset, every pair in them is tested
//     exactly, so the result cannot change;
//   * sweep: the senders of a surviving chunk are LDS broadcasts (four per ds_read_b128), every lane tests them against
//     ITS receiver with the same spelled-out arithmetic and shifts the outcome into a 2 x 32-bit hit mask;
//   * each lane then walks its own hits (recomputing the distance - same operations, same bits) and keeps the KMAX
//     smallest (distance, sender) keys sorted in registers: the same set the rank-counting selection keeps;
//   * the kept senders leave in ascending index order, as the per-row path writes them.
// Tool receivers (tool-tool masking, graph.py:257-260, and the connect_tools_all census) stay on the per-row path.
template <int KMAX>
__device__ __forceinline__ void topk_insert(unsigned long long (&best)[KMAX], unsigned long long x) {
    bool c[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) c[s] = x < best[s];
#pragma unroll
    for (int s = KMAX - 1; s > 0; --s) best[s] = c[s - 1] ? best[s - 1] : (c[s] ? x : best[s]);
    best[0] = c[0] ? x : best[0];
}
__device__ __forceinline__ unsigned long long survivors_box(const EdgeLds& l, int N, float lox, float hix, float loy,
                                                            float hiy, float loz, float hiz, float thr) {
    thr = fabsf(thr);
    const int lane = lane_id();
    const int nch = (N + 63) >> 6;
    bool keep = false;
    if (lane < nch) {
        const float mnx = l.bb[0 * MAXCH + lane], mxx = l.bb[1 * MAXCH + lane];
        const float mny = l.bb[2 * MAXCH + lane], mxy = l.bb[3 * MAXCH + lane];
        const float mnz = l.bb[4 * MAXCH + lane], mxz = l.bb[5 * MAXCH + lane];
        const bool out = (__fsub_rn(lox, mxx) >= thr) || (__fsub_rn(mnx, hix) >= thr) ||
                         (__fsub_rn(loy, mxy) >= thr) || (__fsub_rn(mny, hiy) >= thr) ||
                         (__fsub_rn(loz, mxz) >= thr) || (__fsub_rn(mnz, hiz) >= thr);
        keep = !out;
    }
    return __ballot(keep);
}
typedef float ef4 __attribute__((ext_vector_type(4)));
typedef float ef2 __attribute__((ext_vector_type(2)));
// 32 consecutive senders starting at LDS index j0 (a multiple of 4) against this lane's receiver; sender s -> bit 31-s.
// Two senders per instruction on the packed fp32 pipe: v_pk_add/v_pk_mul are the same IEEE operations per element as
// the scalar ones of dist_exact (no FMA: the file is built with -ffp-contract=off), so the distances are the same bits.
__device__ __forceinline__ unsigned sweep32(const EdgeLds& l, int j0, float xi, float yi, float zi, float thr2) {
    unsigned m = 0;
    const ef2 rx = {xi, xi}, ry = {yi, yi}, rz = {zi, zi};
#pragma unroll 2
    for (int q = 0; q < 8; ++q) {
        const ef4 X = *reinterpret_cast<const ef4*>(l.x + j0 + 4 * q);    // same address in every lane: broadcast
        const ef4 Y = *reinterpret_cast<const ef4*>(l.y + j0 + 4 * q);
        const ef4 Z = *reinterpret_cast<const ef4*>(l.z + j0 + 4 * q);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const ef2 sx = h ? X.zw : X.xy, sy = h ? Y.zw : Y.xy, sz = h ? Z.zw : Z.xy;
            const ef2 dx = rx - sx, dy = ry - sy, dz = rz - sz;
            const ef2 d = ((dx * dx) + (dy * dy)) + (dz * dz);            // graph.py:251-252, element by element
            m = m + m + ((__fsub_rn(d.x, thr2) < 0.0f) ? 1u : 0u);        // graph.py:267
            m = m + m + ((__fsub_rn(d.y, thr2) < 0.0f) ? 1u : 0u);
        }
    }
    return m;
}
template <int KMAX>
__device__ void block_topk(const EdgeDev& a, const EdgeLds& l, int b, int rbase, int r1, float thr, float thr2) {
    const int lane = lane_id();
    const int i = rbase + lane;
    const bool inr = i < r1;
    const int ii = inr ? i : rbase;
    const int fi = l.fl[ii];
    const bool act = inr && (fi & 1) && !(fi & 2);
    const float xi = l.x[ii], yi = l.y[ii], zi = l.z[ii];
    unsigned long long best[KMAX];
#pragma unroll
    for (int t = 0; t < KMAX; ++t) best[t] = KEY_INF;
    if (__ballot(act)) {
        const float INF = __builtin_huge_valf();
        const float lox = wave_min(act ? xi : INF), hix = wave_max(act ? xi : -INF);
        const float loy = wave_min(act ? yi : INF), hiy = wave_max(act ? yi : -INF);
        const float loz = wave_min(act ? zi : INF), hiz = wave_max(act ? zi : -INF);
        unsigned long long todo = survivors_box(l, a.N, lox, hix, loy, hiy, loz, hiz, thr);
        while (todo) {
            const int c = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            unsigned m0 = sweep32(l, 64 * c, xi, yi, zi, thr2);
            unsigned m1 = sweep32(l, 64 * c + 32, xi, yi, zi, thr2);
            if (!act) { m0 = 0; m1 = 0; }
            while (__ballot((m0 | m1) != 0)) {              // every lane walks its own hits of this chunk
                const bool has = (m0 | m1) != 0;
                const bool lo = m0 != 0;
                const unsigned w = lo ? m0 : m1;
                const int bit = has ? __builtin_ctz(w) : 0;
                const int sdr = has ? 64 * c + (lo ? 31 : 63) - bit : 64 * c;
                if (lo) m0 &= m0 - 1; else m1 &= m1 - 1;
                const float d = dist_exact(xi, yi, zi, l.x[sdr], l.y[sdr], l.z[sdr]);
                const unsigned long long key = has ? (((unsigned long long)__float_as_uint(d) << 32) | (unsigned)sdr) : KEY_INF;
                topk_insert<KMAX>(best, key);
            }
        }
    }
    if (!inr || (fi & 2)) return;                           // tool rows: per-row path writes their degree
    int n = 0;
    if (act) {
        int js[KMAX]; bool outm[KMAX];
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            const bool kept = t < a.k && best[t] != KEY_INF;
            js[t] = kept ? (int)(best[t] & 0xffffffffull) : 0;
            outm[t] = kept && (!a.cta || !(l.fl[js[t]] & 2));  // graph.py:283-286: tool senders are decided at emit
        }
        int* ell_row = a.ell + (long)b * a.ell_bstride + (long)i * a.ell_stride;
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            int pos = 0;
#pragma unroll
            for (int u = 0; u < KMAX; ++u) pos += (outm[u] && js[u] < js[t]) ? 1 : 0;
            if (outm[t]) { ell_row[pos] = js[t]; ++n; }
        }
    }
    a.deg[(long)b * a.N + i] = n;
}
constexpr int BLOCK_MIN_ROWS = 256;   // below: the per-row path (16 wavefronts on 16 rows) has the shorter critical path

__device__ __forceinline__ float thr_of(const EdgeDev& a, int b) { return a.thr_vec ? a.thr_vec[b] : a.thr; }
// squared threshold of the adjacency test: fp32*fp32 for the batch builder (graph.py:250), a caller-supplied value
// for the single-graph builder, whose Python squares in double before the fp32 subtraction (graph.py:86,101)
__device__ __forceinline__ float thr2_of(const EdgeDev& a, float thr) { return a.use_thr2 ? a.thr2_override : __fmul_rn(thr, thr); }

// number of final senders of row i when top-k is not active (full sweep with culling); nontool_raw as above
__device__ int row_radius_count(const EdgeDev& a, const EdgeLds& l, int i, float thr, float thr2, int flag, int* nontool_raw) {
    const int lane = lane_id();
    const float xi = l.x[i], yi = l.y[i], zi = l.z[i];
    const int fi = l.fl[i];
    int n = 0, raw = 0;
    if (fi & 1) {
        unsigned long long todo = survivors(l, a.N, xi, yi, zi, thr);
        while (todo) {
            const int c = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            float d; int fj;
            const bool within = pair_within(l, a.N, xi, yi, zi, fi, 64 * c + lane, thr2, d, fj);
            raw += __popcll(__ballot(within && !(fj & 2)));
            n += __popcll(__ballot(within && (!a.cta || (!(fj & 2) && !(fi & 2)))));
        }
    }
    *nontool_raw = raw;
    return n;
}

__global__ __launch_bounds__(EW) void k_edge_count(EdgeDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x / a.slices, sl = blockIdx.x % a.slices;
    if (a.live && b >= *a.live) return;                     // device-planned rollout: this slot has no forward left
    const EdgeLds l = carve(smem, a.N);
    load_candidate(a, l, b, true);
    const float thr = thr_of(a, b);
    const float thr2 = thr2_of(a, thr);
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int r0 = sl * a.rows_per_slice;
    const int r1 = min(a.N, r0 + a.rows_per_slice);
    const int ntool = l.misc[2];
    // pass 1: rows of this slice (+ every tool row, whose non-tool senders decide the batch flag graph.py:277)
    const bool blocks = a.topk_active && a.k <= 20 && r1 - r0 >= a.block_min_rows;
    if (blocks) {                                          // non-tool rows, 64 per wavefront
        for (int rb = r0 + 64 * wave; rb < r1; rb += 64 * EWAVES) {
            if (a.k <= 5) block_topk<5>(a, l, b, rb, r1, thr, thr2);
            else if (a.k <= 10) block_topk<10>(a, l, b, rb, r1, thr, thr2);
            else block_topk<20>(a, l, b, rb, r1, thr, thr2);
        }
    }
    // r06: the rows of the slice directly, then the tool rows outside it found 64 flags at a time (the r01 loop walked ALL N rows
    // per wavefront to find them: 127 dependent LDS reads per wave on a cloth-sized graph, whatever the slice held)
    auto one_row = [&](int i, bool mine, bool is_tool) {
        int raw = 0, n;
        if (a.topk_active) n = row_topk(a, l, i, thr, thr2, a.ell + (long)b * a.ell_bstride + (long)i * a.ell_stride, &raw);
        else n = row_radius_count(a, l, i, thr, thr2, 0, &raw);
        if (lane == 0) {
            if (mine) a.deg[(long)b * a.N + i] = n;        // senders not governed by the tool rule
            if (a.cta == 1 && is_tool && raw) atomicOr(&l.misc[0], 1);
        }
    };
    for (int i = r0 + wave; i < r1; i += EWAVES) {
        const bool is_tool = l.fl[i] & 2;                  // wave-uniform
        if (blocks && !is_tool) continue;                  // done above
        one_row(i, true, is_tool);
    }
    if (a.cta == 1 && ntool) {                             // every tool row decides the batch flag (graph.py:277), in every slice
        for (int c = wave; c < (a.N + 63) >> 6; c += EWAVES) {
            const int j = 64 * c + lane;
            unsigned long long tm = __ballot(j < a.N && (l.fl[j < a.N ? j : 0] & 2) && !(j >= r0 && j < r1));
            while (tm) {
                const int i = 64 * c + __ffsll((long long)tm) - 1;
                tm &= tm - 1;
                one_row(i, false, true);
            }
        }
    }
    __syncthreads();
    // batch builder: all-or-nothing flag (graph.py:277); single-graph builder (cta == 2): unconditional (graph.py:119-122)
    const int flag = a.cta == 2 ? 1 : l.misc[0];
    if (a.ell_full && a.cta && flag && ntool) {            // ascending tool index list for the slot-indexed rows
        if (wave == 0) {
            int m = 0;
            for (int c = 0; c < (a.N + 63) >> 6; ++c) {
                const int j = 64 * c + lane;
                const bool t = j < a.N && (l.fl[j] & 2);
                const unsigned long long bt = __ballot(t);
                if (t) l.tlist[m + __popcll(bt & lanes_below(lane))] = (unsigned short)j;
                m += __popcll(bt);
            }
        }
        __syncthreads();
    }
    // pass 2: add the all-or-nothing tool senders (graph.py:284,286) and total the slice
    int my = 0;
    for (int i = r0 + threadIdx.x; i < r1; i += EW) {
        int d = a.deg[(long)b * a.N + i];
        if (a.cta && (l.fl[i] & 1) && flag && !(a.cta == 2 && (l.fl[i] & 2))) {
            if (a.ell_full) {                               // slot-indexed graph: the tool senders follow the kept ones
                int* row = a.ell + (long)b * a.ell_bstride + (long)i * a.ell_stride + d;
                for (int m = 0; m < ntool; ++m) row[m] = l.tlist[m];
            }
            d += ntool;
        }
        a.deg[(long)b * a.N + i] = d;
        my += d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) my += __shfl_xor(my, o);
    if (lane == 0 && my) atomicAdd(&l.misc[1], my);        // integer add: order-independent
    __syncthreads();
    if (threadIdx.x == 0) {
        a.slice_tot[b * a.slices + sl] = l.misc[1];
        if (sl == 0) a.cta_flag[b] = flag;
    }
}

__global__ __launch_bounds__(EW) void k_edge_emit(EdgeDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x / a.slices, sl = blockIdx.x % a.slices;
    if (a.live && b >= *a.live) {                           // no forward left: present an empty graph downstream
        if (sl == 0 && threadIdx.x == 0) a.n_edges[b] = 0;
        return;
    }
    const EdgeLds l = carve(smem, a.N);
    load_candidate(a, l, b, !a.topk_active);
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int r0 = sl * a.rows_per_slice;
    const int r1 = min(a.N, r0 + a.rows_per_slice);
    const int nrows = max(0, r1 - r0);
    const int flag = a.cta_flag[b];
    const int ntool = l.misc[2];
    const bool tools_on = a.cta && flag && ntool > 0;
    int base = 0, total = 0;
    {   // (one load per thread + a reduction instead of a loop over the slices in every thread: see k_ell_index)
        int pb = 0, pt = 0;
        for (int s = threadIdx.x; s < a.slices; s += EW) {
            const int v = a.slice_tot[b * a.slices + s];
            pb += s < sl ? v : 0;
            pt += v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { pb += __shfl_xor(pb, o); pt += __shfl_xor(pt, o); }
        if (lane == 0) { l.scan[2 * wave] = pb; l.scan[2 * wave + 1] = pt; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < EWAVES; ++w) { base += l.scan[2 * w]; total += l.scan[2 * w + 1]; }
        __syncthreads();                                    // l.scan is the scan buffer below
    }
    // ---- exclusive scan of this slice's degrees: each thread owns a contiguous run of rows
    const int per = (nrows + EW - 1) / EW;
    const int t0 = min(nrows, (int)threadIdx.x * per), t1 = min(nrows, t0 + per);
    int mysum = 0;
    for (int t = t0; t < t1; ++t) mysum += a.deg[(long)b * a.N + r0 + t];
    l.scan[threadIdx.x] = mysum;
    __syncthreads();
    for (int off = 1; off < EW; off <<= 1) {               // Hillis-Steele, integers
        int v = 0;
        if ((int)threadIdx.x >= off) v = l.scan[threadIdx.x - off];
        __syncthreads();
        l.scan[threadIdx.x] += v;
        __syncthreads();
    }
    int run = base + l.scan[threadIdx.x] - mysum;
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
    // ---- tool index tables for the merge (tprefix[j] = tools with index < j; tlist ascending)
    if (tools_on) {
        __syncthreads();
        const int perN = (a.N + EW - 1) / EW;
        const int j0 = min(a.N, (int)threadIdx.x * perN), j1 = min(a.N, j0 + perN);
        int c = 0;
        for (int j = j0; j < j1; ++j) c += (l.fl[j] & 2) ? 1 : 0;
        l.scan[threadIdx.x] = c;
        __syncthreads();
        for (int off = 1; off < EW; off <<= 1) {
            int v = 0;
            if ((int)threadIdx.x >= off) v = l.scan[threadIdx.x - off];
            __syncthreads();
            l.scan[threadIdx.x] += v;
            __syncthreads();
        }
        int r = l.scan[threadIdx.x] - c;
        for (int j = j0; j < j1; ++j) {
            l.tprefix[j] = (unsigned short)r;
            if (l.fl[j] & 2) { l.tlist[r] = (unsigned short)j; ++r; }
        }
    }
    __syncthreads();   // row_ptr of this slice and the tool tables are complete and visible inside the workgroup
    if (!fits) return;
    int* recv = a.recv + (long)b * a.edge_cap;
    int* send = a.send + (long)b * a.edge_cap;
    if (a.topk_active) {
        for (int i = r0 + wave; i < r1; i += EWAVES) {
            const int off = a.row_ptr[(long)b * (a.N + 1) + i];
            const int d = a.deg[(long)b * a.N + i];
            const bool row_tools = tools_on && (l.fl[i] & 1) && !(a.cta == 2 && (l.fl[i] & 2));
            const int nt = row_tools ? ntool : 0;
            const int nk = d - nt;                          // kept senders from the ELL row
            const int* ell = a.ell + (long)b * a.ell_bstride + (long)i * a.ell_stride;
            for (int t = lane; t < nk; t += 64) {
                const int j = ell[t];
                const int p = off + t + (row_tools ? l.tprefix[j] : 0);
                recv[p] = i; send[p] = j;
            }
            for (int m = lane; m < nt; m += 64) {
                const int j = l.tlist[m];
                int before = 0;
                for (int t = 0; t < nk; ++t) before += ell[t] < j ? 1 : 0;
                const int p = off + m + before;
                recv[p] = i; send[p] = j;
            }
        }
    } else {
        const float thr = thr_of(a, b);
        const float thr2 = thr2_of(a, thr);
        for (int i = r0 + wave; i < r1; i += EWAVES) {
            const float xi = l.x[i], yi = l.y[i], zi = l.z[i];
            const int fi = l.fl[i];
            int off = a.row_ptr[(long)b * (a.N + 1) + i];
            // tool senders are not confined to surviving chunks: sweep every chunk when the tool rule is on
            unsigned long long todo = (a.cta && flag) ? ~0ull : survivors(l, a.N, xi, yi, zi, thr);
            const int nch = (a.N + 63) >> 6;
            if (nch < 64) todo &= (1ull << nch) - 1ull;
            while (todo) {
                const int c = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const bool m = member_radius(a, l, i, 64 * c + lane, xi, yi, zi, fi, thr2, flag);
                const unsigned long long bm = __ballot(m);
                if (m) {
                    const int p = off + __popcll(bm & lanes_below(lane));
                    recv[p] = i; send[p] = 64 * c + lane;
                }
                off += __popcll(bm);
            }
        }
    }
}

// ---- rollout fast path: index the slot-indexed graph left by k_edge_count (see EdgeArgs::ell_full).  One workgroup
// per candidate: max_nR rule, receiver of every slot, list of the slots that are not self-loops (only those go
// through the relation encoder), edge counts.  Integer scan, slot order = (receiver, position in row) = CSR order.
// r06: slot-parallel.  The r02 version gave every thread whole rows and walked their slots one load at a time (deg -> ell -> ...:
// a chain of ~12 dependent cache-miss latencies, 23 us whether the launch held one graph or 128).  Now pass 1 takes the slots
// e = tid, tid + 1024, ... - consecutive lanes on consecutive slots, every load independent of every other - and leaves one
// "goes through the relation encoder" bit per slot in LDS (a wave ballot: 64 consecutive slots = one 64-bit word, no atomics);
// pass 2 gives every thread a contiguous run of those words: popcount, one block scan, and the set bits leave in slot order.
// Same outputs bit for bit (recv per slot, ns list in slot order, pk, n_ns, n_edges, the share counters).
constexpr int ELL_BITMAP_MAX_WORDS64 = 16384;              // 128 KB of dynamic LDS: 1,048,576 slots (N = 4096 rows of up to 256 slots; the builder's limit is topk <= 128, M <= 8)
__global__ __launch_bounds__(EW) void k_ell_index(EdgeDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(smem);
    __shared__ int wave_tot[EWAVES];
    __shared__ int n_shared;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.live && b >= *a.live) {                           // no forward left: nothing for the relation encoder to do
        if (tid == 0) { a.n_ns[b] = 0; a.n_edges[b] = 0; }
        return;
    }
    if (tid == 0) n_shared = 0;
    // edges of the candidate = sum of its slice totals: one load per thread and a reduction (a loop over the slices in every
    // thread is a chain of dependent scalar loads - 127 of them for one cloth-sized graph: most of the kernel's 24 us)
    int total = 0;
    {
        int part = 0;
        for (int s = tid; s < a.slices; s += EW) part += a.slice_tot[b * a.slices + s];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) wave_tot[wave] = part;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < EWAVES; ++w) total += wave_tot[w];
        __syncthreads();                                    // wave_tot is used again by the scan below
    }
    const bool hide = total > a.max_nR && a.zero_on_overflow;   // downstream kernels then see an empty graph
    int* deg = a.deg + (long)b * a.N;
    const int* ell = a.ell + (long)b * a.ell_bstride;
    int* recv = a.recv + (long)b * a.edge_cap;
    int* ns = a.ns_edge + (long)b * a.edge_cap;
    // First forward of a dynamics() call (EdgeArgs::send_pk): an object-object slot whose sender is in the receiver's row of
    // the base graph takes its C row from the shared table and is left out of the relation encoder's list.  (With the same
    // positions every object sender a candidate keeps IS in the base row - a tool can only push senders out of a row's top-k -
    // but nothing relies on it: a sender that is not found is simply encoded by the candidate itself.)
    int* pk = a.send_pk ? a.send_pk + (long)b * a.edge_cap : nullptr;
    // (prefix sharing: a slot that starts from a later base state is not at the start state's forward - it shares nothing)
    const bool eligible = !a.share_start || a.share_start[a.share_cand ? a.share_cand[b] : a.share_b0 + b] == 0;
    const int stride = a.ell_stride;
    const int nslots = a.N * stride;
    const int nw = (nslots + 63) >> 6;                      // 64-slot words
    // ---- pass 1: one slot per lane
    int shared = 0;
    constexpr int UN = 8;                                   // words per wavefront and trip: the loads of all eight are issued together
    for (int eb = wave * 64; eb < nw * 64; eb += UN * EW) { // wave-uniform trip count: the ballots below see whole words
      int iu[UN], du[UN], ju[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int e = eb + u * EW + lane;
        const bool in = e < nslots;
        iu[u] = in ? e / stride : 0;
        du[u] = (in && !hide) ? deg[iu[u]] : 0;
        ju[u] = in ? ell[e] : 0;                            // (allocated whatever the degree is: slot-indexed rows)
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int e0 = eb + u * EW;
        if (e0 >= nw * 64) break;                           // wave-uniform
        const int e = e0 + lane;
        const bool in = e < nslots;
        const int i = iu[u], d = du[u], j = ju[u];
        const int t = e - i * stride;
        const bool valid = in && t < d;
        bool own = valid && j != i;
        if (pk && valid) {
            int v = j;
            if (own && eligible && i < a.share_No && j < a.share_No) {
                const int bd = a.base_deg[i];
                const int* brow = a.base_send + (long)i * a.base_stride;
                for (int u = 0; u < bd; ++u)
                    if (brow[u] == j) { v = j | ((u + 1) << 12); own = false; ++shared; break; }
            }
            pk[e] = v;
        }
        if (valid) recv[e] = i;
        const unsigned long long bal = __ballot(own);
        if (lane == 0) bits[e0 >> 6] = bal;
      }
    }
    if (hide) for (int i = tid; i < a.N; i += EW) deg[i] = 0;
    __syncthreads();
    // ---- pass 2: contiguous words per thread, block scan of the popcounts, set bits out in slot order
    const int per = (nw + EW - 1) / EW;
    const int w0 = min(nw, tid * per), w1 = min(nw, w0 + per);
    int mine = 0;
    for (int w = w0; w < w1; ++w) mine += __popcll(bits[w]);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < EWAVES; ++w) { const int v = wave_tot[w]; before += w < wave ? v : 0; all += v; }
    int pos = before + incl - mine;
    for (int w = w0; w < w1; ++w) {
        unsigned long long m = bits[w];
        while (m) {
            ns[pos++] = (w << 6) + __ffsll((long long)m) - 1;
            m &= m - 1;
        }
    }
    if (pk && a.share_stats) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) shared += __shfl_xor(shared, o);
        if (lane == 0 && shared) atomicAdd(&n_shared, shared);
        __syncthreads();
    }
    if (tid == EW - 1) {
        a.n_ns[b] = all;
        a.n_edges[b] = hide ? 0 : total;
        if (a.overflow && total > a.max_nR) atomicMax(a.overflow, total);
        if (pk && a.share_stats) {                           // integer counters: order-free
            atomicAdd(a.share_stats + 0, (unsigned long long)n_shared);
            atomicAdd(a.share_stats + 1, (unsigned long long)all);
        }
    }
}

hipError_t launch_edge_build(const EdgeArgs& h, hipStream_t st, void (*mark)(void*, int, int), void* mark_ctx) {
    EdgeDev a;
    a.pos = h.pos; a.pos_bstride = h.pos_bstride; a.mask = h.mask; a.tool = h.tool; a.thr_vec = h.thr_vec; a.thr = h.thr;
    a.thr2_override = h.thr2_override; a.use_thr2 = h.use_thr2;
    a.B = h.B; a.N = h.N; a.k = min(h.N, h.topk); a.topk_active = a.k < h.N; a.cta = h.cta; a.edge_cap = h.edge_cap;
    a.slices = h.slices; a.rows_per_slice = (h.N + h.slices - 1) / h.slices;
    a.ell = h.ell; a.deg = h.deg; a.slice_tot = h.slice_tot; a.cta_flag = h.cta_flag;
    a.ell_full = h.ell_full && a.topk_active;
    a.ell_stride = a.ell_full ? h.ell_stride : a.k;
    a.ell_bstride = a.ell_full ? h.ell_bstride : (long)h.N * a.k;
    a.ns_edge = h.ns_edge; a.n_ns = h.n_ns;
    a.recv = h.recv; a.send = h.send; a.row_ptr = h.row_ptr; a.n_edges = h.n_edges; a.overflow = h.overflow;
    a.max_nR = h.max_nR; a.zero_on_overflow = h.zero_on_overflow; a.live = h.live;
    a.send_pk = a.ell_full ? h.send_pk : nullptr; a.base_send = h.base_send; a.base_deg = h.base_deg; a.base_stride = h.base_stride;
    a.share_No = h.share_No; a.share_stats = h.share_stats;
    a.share_start = h.share_start; a.share_cand = h.share_cand; a.share_b0 = h.share_b0;
    a.block_min_rows = h.block_min_rows >= 0 ? h.block_min_rows : BLOCK_MIN_ROWS;   // A/B switch (Options::edge_block_min); results identical
    const size_t lds = edge_lds_bytes(h.N);
    // the > 64 KB dynamic-LDS opt-in is a per-DEVICE function attribute: track it per device ordinal
    static unsigned long long attr_devices = 0;
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess) dev_id = 0;
    if (dev_id >= 64 || !(attr_devices >> dev_id & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_edge_count),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_edge_emit), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 256);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ell_index), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 256);
        if (e != hipSuccess) return e;
        if (dev_id < 64) attr_devices |= 1ull << dev_id;
    }
    if (mark) mark(mark_ctx, FAM_EDGE_COUNT, 0);
    hipLaunchKernelGGL(k_edge_count, dim3(h.B * h.slices), dim3(EW), lds, st, a);
    if (mark) mark(mark_ctx, FAM_EDGE_COUNT, 1);
    if (mark) mark(mark_ctx, FAM_EDGE_EMIT, 0);
    if (a.ell_full) {
        const long nw = ((long)h.N * a.ell_stride + 63) >> 6;
        if (nw > ELL_BITMAP_MAX_WORDS64) return hipErrorInvalidValue;     // (N <= 4096 and topk + M <= 112: never on this path)
        hipLaunchKernelGGL(k_ell_index, dim3(h.B), dim3(EW), (size_t)nw * 8, st, a);
    }
    else hipLaunchKernelGGL(k_edge_emit, dim3(h.B * h.slices), dim3(EW), lds, st, a);
    if (mark) mark(mark_ctx, FAM_EDGE_EMIT, 1);
    return hipGetLastError();
}

// ---- non-self-loop edge list (self-loop dedupe).  One workgroup per candidate.  Row i has at most one self-loop;
// S(i) = number of self-loops in rows < i (integer scan); the t-th non-self edge keeps the CSR order.
__global__ __launch_bounds__(EW) void k_edge_nonself(const int* __restrict__ send_all, const int* __restrict__ row_ptr_all,
                                                      int N, int edge_cap, int* __restrict__ ns_all, int* __restrict__ n_ns,
                                                      const int* __restrict__ live) {
    __shared__ int scan[EW];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (live && b >= *live) { if (tid == 0) n_ns[b] = 0; return; }
    const int* send = send_all + (long)b * edge_cap;
    const int* rp = row_ptr_all + (long)b * (N + 1);
    int* ns = ns_all + (long)b * edge_cap;
    const int per = (N + EW - 1) / EW;
    const int i0 = min(N, tid * per), i1 = min(N, i0 + per);
    int mine = 0;
    for (int i = i0; i < i1; ++i) {
        const int e0 = rp[i], e1 = rp[i + 1];
        for (int e = e0; e < e1; ++e) mine += send[e] == i ? 1 : 0;
    }
    scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < EW; off <<= 1) {
        int v = 0;
        if (tid >= off) v = scan[tid - off];
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int S = scan[tid] - mine;                              // self-loops before my first row
    for (int i = i0; i < i1; ++i) {
        const int e0 = rp[i], e1 = rp[i + 1];
        for (int e = e0; e < e1; ++e) {
            if (send[e] == i) ++S;
            else ns[e - S] = e;
        }
    }
    if (tid == EW - 1) n_ns[b] = rp[N] - scan[EW - 1];
}
hipError_t launch_edge_nonself(const int* recv, const int* send, const int* row_ptr, int B, int N, int edge_cap,
                               int* ns_edge, int* n_ns, const int* live, hipStream_t st) {
    (void)recv;
    hipLaunchKernelGGL(k_edge_nonself, dim3(B), dim3(EW), 0, st, send, row_ptr, N, edge_cap, ns_edge, n_ns, live);
    return hipGetLastError();
}

size_t edge_build_max_particles() { return MAXCH * 64; }
// ints of ELL scratch per (candidate, particle)
int edge_ell_stride(int N, int topk) { return topk < N ? topk : 0; }

}  // namespace ag
