// Latency-mode MLP chains for small batches (gfx950 only): the same chains as ag_mlp.hip - same mathematics, same memory
// layout, and BIT-IDENTICAL results - restructured so that a short grid finishes sooner.
//
// Why.  The throughput kernels give a wavefront 32 rows and all 160 output features of every layer: 380 MFMAs of 64
// cycles per layer, ~10 us per layer, 35-45 us per chain - fine when thousands of workgroups keep the chip busy, but the
// floor of every launch when the whole batch is a handful of workgroups (one rope graph: 3; the planner's B = 1
// best-candidate rollout, reference src/planning/real_world/planner.py:268-271).  Here a workgroup owns 32 rows and its
// four wavefronts split every layer between them: wave (rg, par) computes, for row group rg (16 rows), the five
// 16-feature output tiles T = par, par+2, .. with v_mfma_f32_16x16x4_f32 - 190 MFMAs of 32 cycles per layer, a quarter of
// the time - and four times as many workgroups share the rows.
//
// Between layers the activations pass through an LDS image in plain row-major order (rows x 160 features): everybody
// writes the tiles it computed, a barrier, everybody reads the whole row set back as its B operand.  The same image is
// what the gather writes, what row stores read (coalesced 16-B pieces) and what makes this file independent of the
// register layout of any other kernel.
//
// Bit-identity with ag_mlp.hip.  An fp32 MFMA is a k-ordered fmaf chain, so a layer's result depends only on the ORDER in
// which the k's are fed.  The 32-row chains feed k-step s, lane half h = feature 32t + (r&3) + 8(r>>2) + 4h with
// s = 16t + r.  Here k-step (T, r) of lane group g is made to take the feature at position 4(4T + r) + g of that very
// sequence: feature 16T + 8(r>>1) + 4(g&1) + 2(r&1) + (g>>1) - a bit permutation inside each 16-feature tile, applied to
// the weight image on both its output-row and its k side (ag_api.hip: pack_layer16).  So a batch gives the same bits
// whichever kernel family its size selects, the self-loop constant rows are shared, and sharded == unsharded holds
// across the switch (tests: test_latency_kernels_equal_throughput_kernels_bitwise).
#include "ag_common.h"
#include <cstdint>
#include <cstdlib>

namespace ag {
namespace lat {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WGL = 256;                    // 4 wavefronts: (row group rg = wave>>1, tile parity par = wave&1)
constexpr int ROWS = 32;
constexpr int NT = 10;                      // 16-feature tiles
constexpr int KS = 38;                      // k-steps of 4 (tile 9 has two: features 144..151)
constexpr int CHUNK_FLOATS = NT * 64 * 4;   // weight image: [chunk of 4 k-steps][tile][lane][4]
constexpr int NCHUNK = 10;
constexpr int LAYER_FLOATS = NCHUNK * CHUNK_FLOATS;
constexpr int HEAD_FLOATS = NCHUNK * 256;   // 3-output head: one tile
constexpr int IMG_PITCH = 164;              // floats per image row (656 B: conflict-free 16-B column reads)
constexpr int IMG_FLOATS = ROWS * IMG_PITCH;             // one image: 32 rows (20,992 B); two images alternate

// offset inside a 16-feature tile of the feature that register r of lane group g holds (see header)
__device__ __forceinline__ constexpr int foff(int g, int r) { return 8 * (r >> 1) + 4 * (g & 1) + 2 * (r & 1) + (g >> 1); }

struct Act { f32x4 t[NT]; };                // B operand: all 160 features of the lane's row
struct Out { f32x4 t[5]; };                 // this wave's five output tiles T = par + 2m

__device__ __forceinline__ void zero(Out& o) {
#pragma unroll
    for (int m = 0; m < 5; ++m) o.t[m] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// ReLU, then force the bias slot (feature 150 = 144 + foff(1, 1): tile 9 = par 1, m 4; lane group 1, register 1) to 1
__device__ __forceinline__ void relu_one(Out& o, int par, int g) {
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) o.t[m][r] = __builtin_fmaxf(o.t[m][r], 0.0f);
    if (par == 1 && g == 1) o.t[4][1] = 1.0f;
}
static_assert(16 * 9 + foff(1, 1) == 150, "bias slot");

// own tiles -> image (row j of this row group), 4-byte pieces
__device__ __forceinline__ void put_tiles(const Out& o, float* img_rg, int par, int j, int g) {
    float* p = img_rg + j * IMG_PITCH;
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[16 * (par + 2 * m) + foff(g, r)] = o.t[m][r];
}
// image -> B operand (all ten tiles of row j)
__device__ __forceinline__ void get_act(Act& x, const float* img_rg, int j, int g) {
    const float* p = img_rg + j * IMG_PITCH;
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) x.t[T][r] = (T < 9 || r < 2) ? p[16 * T + foff(g, r)] : 0.0f;
}
// image -> this wave's accumulator tiles (seed)
__device__ __forceinline__ void get_tiles(Out& o, const float* img_rg, int par, int j, int g) {
    const float* p = img_rg + j * IMG_PITCH;
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) o.t[m][r] = p[16 * (par + 2 * m) + foff(g, r)];
}

// one 160-wide layer for this wave's five tiles: out (+)= W[tiles] * x.  Weight fragments come straight from global memory
// (L2-resident, 1 KB per wave-instruction), three chunks ahead of their use.
#ifndef AG_LAT_PF
#define AG_LAT_PF 2                          // chunks of weight fragments in flight ahead of the one being multiplied
#endif
__device__ __forceinline__ void layer(const float* __restrict__ w, const Act& x, Out& out, int par, int lane) {
    const float* wp = w + (par * 64 + lane) * 4;                     // tile par of chunk 0; tiles step by 2 * 256 floats
    constexpr int PF = AG_LAT_PF, RING = PF + 1;
    f32x4 a[RING][5];
#pragma unroll
    for (int c = 0; c < PF; ++c)
#pragma unroll
        for (int m = 0; m < 5; ++m) a[c][m] = *reinterpret_cast<const f32x4*>(wp + c * CHUNK_FLOATS + m * 512);
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        if (c + PF < NCHUNK) {
#pragma unroll
            for (int m = 0; m < 5; ++m) a[(c + PF) % RING][m] = *reinterpret_cast<const f32x4*>(wp + (c + PF) * CHUNK_FLOATS + m * 512);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = 4 * c + e;
            if (s < KS) {
                const int T = s < 36 ? s / 4 : 9, r = s < 36 ? s % 4 : s - 36;
                const float b = x.t[T][r];
#pragma unroll
                for (int m = 0; m < 5; ++m)
                    out.t[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c % RING][m][e], b, out.t[m], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);                           // keep the prefetch distance: no hoisting of all 50 reads
    }
}
// first layer from NS k-steps: bs[s] = input feature 4s + g of the lane's row
template <int NS>
__device__ __forceinline__ void layer_first(const float* __restrict__ w, const float* bs, Out& out, int par, int lane) {
    const float* wp = w + (par * 64 + lane) * 4;
#pragma unroll
    for (int c = 0; c < (NS + 3) / 4; ++c) {
        f32x4 a[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) a[m] = *reinterpret_cast<const f32x4*>(wp + c * CHUNK_FLOATS + m * 512);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = 4 * c + e;
            if (s < NS) {
#pragma unroll
                for (int m = 0; m < 5; ++m) out.t[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][e], bs[s], out.t[m], 0, 0, 0);
            }
        }
    }
}
__device__ __forceinline__ float sel4(int g, float a, float b, float c, float d) {
    const float lo = (g & 1) ? b : a, hi = (g & 1) ? d : c;
    return (g & 2) ? hi : lo;
}

// image (32 rows) -> global rows, coalesced: the workgroup's 256 lanes move 40 pieces of 16 B per row; rows[] (LDS) holds
// the destination row of every local row, -1 = none
__device__ __forceinline__ void store_image(const float* img, float* __restrict__ base, const int* rows, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 40 / WGL; ++i) {
        const int q = tid + WGL * i, row = q / 40, piece = q - row * 40;
        const int dst = rows[row];
        const f32x4 v = *reinterpret_cast<const f32x4*>(img + row * IMG_PITCH + 4 * piece);
        if (dst >= 0) *reinterpret_cast<f32x4*>(base + (long)dst * NFP + 4 * piece) = v;
    }
}

// ------------------------------------------------------------------------------------------------ edge chain
struct EDev {
    const float* w; const float* node_in; const float* feat12; const float* group; float* C;
    const int* recv; const int* send; const int* n_edges; const int* ns_edge; const int* n_ns;
    int B, N, n_inst, edge_cap, c_cap;
};
struct WLat {                              // offsets (floats) into the latency weight image
    static constexpr int E_L1 = 0;                                            // 2 chunks (5 k-steps: 17 inputs + bias)
    static constexpr int E_L2 = E_L1 + 2 * CHUNK_FLOATS;
    static constexpr int E_L3 = E_L2 + LAYER_FLOATS;
    static constexpr int E_W1 = E_L3 + LAYER_FLOATS;
    static constexpr int P_WB = E_W1 + LAYER_FLOATS;
    static constexpr int N_W2 = P_WB + LAYER_FLOATS;
    static constexpr int N_W3 = N_W2 + LAYER_FLOATS;
    static constexpr int P_P0 = N_W3 + LAYER_FLOATS;
    static constexpr int P_P1 = P_P0 + LAYER_FLOATS;
    static constexpr int P_P2 = P_P1 + LAYER_FLOATS;                          // head
    static constexpr int N_L1 = P_P2 + HEAD_FLOATS;                           // r06, node encode chain: 1 chunk (2 k-steps: 6 inputs + bias)
    static constexpr int N_L2 = N_L1 + CHUNK_FLOATS;
    static constexpr int N_L3 = N_L2 + LAYER_FLOATS;
    static constexpr int N_WA = N_L3 + LAYER_FLOATS;                          // Wa + b_pp
    static constexpr int TOTAL = N_WA + LAYER_FLOATS;
};

// rel_inputs (17) -> Encoder(17,150,150) -> W1*enc + b_rp  => C      (model.py:249-282, 303, 317-318 first block)
__global__ __launch_bounds__(WGL, 2) void k_edge_enc_lat(EDev g) {
    __shared__ __attribute__((aligned(16))) float img[2 * IMG_FLOATS];
    __shared__ int rows[ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rg = wave >> 1, par = wave & 1, j = lane & 15, lg = lane >> 4;
    const int b = (int)(blockIdx.x % (unsigned)g.B);
    const int e0 = (int)(blockIdx.x / (unsigned)g.B) * ROWS;
    const int ne = g.n_ns ? g.n_ns[b] : g.n_edges[b];
    if (e0 >= ne) return;
    const int t = e0 + 16 * rg + j;
    const bool valid = t < ne;
    const int el = g.ns_edge ? g.ns_edge[(long)b * g.edge_cap + (valid ? t : 0)] : t;
    const int elc = valid ? el : (g.ns_edge ? el : 0);
    if (par == 0 && lg == 0) rows[16 * rg + j] = valid ? (int)((long)b * g.c_cap + el) : -1;
    const int r = g.recv[(long)b * g.edge_cap + elc], s = g.send[(long)b * g.edge_cap + elc];
    const long pr = (long)b * g.N + r, ps = (long)b * g.N + s;
    float bs[5];                                             // B operand of the 5 first-layer k-steps: feature 4s + g
    {
        const float* nr = g.node_in + pr * NODE_IN; const float* nsnd = g.node_in + ps * NODE_IN;
        float gd = 0.0f;
        for (int k = 0; k < g.n_inst; ++k) gd += fabsf(g.group[pr * g.n_inst + k] - g.group[ps * g.n_inst + k]);   // model.py:264-267
        const f32x4* fr = reinterpret_cast<const f32x4*>(g.feat12 + pr * F12);
        const f32x4* fs = reinterpret_cast<const f32x4*>(g.feat12 + ps * F12);
        const f32x4 a0 = fr[0], a1 = fr[1], a2 = fr[2], c0 = fs[0], c1 = fs[1], c2 = fs[2];
        // [attrs_r(2), attrs_s(2) | group_diff, d0..d2 | d3..d6 | d7..d10 | d11, 1, 0, 0]   (model.py:253-279)
        bs[0] = sel4(lg, nr[0], nr[1], nsnd[0], nsnd[1]);
        bs[1] = sel4(lg, gd, a0[0] - c0[0], a0[1] - c0[1], a0[2] - c0[2]);
        bs[2] = sel4(lg, a0[3] - c0[3], a1[0] - c1[0], a1[1] - c1[1], a1[2] - c1[2]);
        bs[3] = sel4(lg, a1[3] - c1[3], a2[0] - c2[0], a2[1] - c2[1], a2[2] - c2[2]);
        bs[4] = sel4(lg, a2[3] - c2[3], 1.0f, 0.0f, 0.0f);
    }
    float* imgA = img + rg * 16 * IMG_PITCH;                 // this row group's rows of image 0
    float* imgB = imgA + IMG_FLOATS;                         // ... of image 1
    Act x; Out y;
    zero(y);
    layer_first<5>(g.w + WLat::E_L1, bs, y, par, lane);
    relu_one(y, par, lg);
    put_tiles(y, imgA, par, j, lg);
    __syncthreads();
    get_act(x, imgA, j, lg);
    zero(y);
    layer(g.w + WLat::E_L2, x, y, par, lane);
    relu_one(y, par, lg);
    put_tiles(y, imgB, par, j, lg);
    __syncthreads();
    get_act(x, imgB, j, lg);
    zero(y);
    layer(g.w + WLat::E_L3, x, y, par, lane);
    relu_one(y, par, lg);
    put_tiles(y, imgA, par, j, lg);
    __syncthreads();
    get_act(x, imgA, j, lg);
    zero(y);
    layer(g.w + WLat::E_W1, x, y, par, lane);
    put_tiles(y, imgB, par, j, lg);
    __syncthreads();
    store_image(img + IMG_FLOATS, g.C, rows, tid);
}

// ------------------------------------------------------------------------------------------------ node encode chain (r06)
// p_inputs (6) -> Encoder(6,150,150) = p_enc => eff;  P = Wa*p_enc + b_pp;  U = W2*p_enc;  V = W3*p_enc   (model.py:297-298, 317-318,
// 328-330) - ag_mlp.hip: k_node_enc with 32-row workgroups whose four wavefronts split every layer.  The class table of a rollout
// (2 N_o + B M rows, once per look-ahead step) is a handful of 128-row workgroups there: six layers of ~10 us each, 66 us whatever
// the batch; here 29.  Same k order per layer, same bits (test_latency_kernels_equal_throughput_kernels_bitwise).
struct NEDev { const float* w; const float* node_in; float* eff; float* P; float* U; float* V; long row0, nrows; };

__global__ __launch_bounds__(WGL, 2) void k_node_enc_lat(NEDev g) {
    __shared__ __attribute__((aligned(16))) float img[2 * IMG_FLOATS];
    __shared__ int rows[ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rg = wave >> 1, par = wave & 1, j = lane & 15, lg = lane >> 4;
    const long rend = g.row0 + g.nrows;
    const long row = g.row0 + (long)blockIdx.x * ROWS + 16 * rg + j;
    const bool valid = row < rend;
    const long rowc = valid ? row : rend - 1;
    if (par == 0 && lg == 0) rows[16 * rg + j] = valid ? (int)row : -1;
    float bs[2];                                             // B operand of the 2 first-layer k-steps: feature 4s + g
    {
        const float* p = g.node_in + rowc * NODE_IN;         // [attr_obj, attr_tool, phys, act_xyz, 1, 0]
        bs[0] = p[lg];
        bs[1] = p[4 + lg];
    }
    float* imgA = img + rg * 16 * IMG_PITCH;
    float* imgB = imgA + IMG_FLOATS;
    Act x; Out y;
    zero(y);
    layer_first<2>(g.w + WLat::N_L1, bs, y, par, lane);
    relu_one(y, par, lg);
    put_tiles(y, imgA, par, j, lg);
    __syncthreads();
    get_act(x, imgA, j, lg);
    zero(y);
    layer(g.w + WLat::N_L2, x, y, par, lane);
    relu_one(y, par, lg);
    put_tiles(y, imgB, par, j, lg);
    __syncthreads();
    get_act(x, imgB, j, lg);
    zero(y);
    layer(g.w + WLat::N_L3, x, y, par, lane);
    relu_one(y, par, lg);                                    // p_enc (slot 150 = 1 for the bias of Wa)
    put_tiles(y, imgA, par, j, lg);
    __syncthreads();
    store_image(img, g.eff, rows, tid);
    get_act(x, imgA, j, lg);
    zero(y);
    layer(g.w + WLat::N_WA, x, y, par, lane);
    put_tiles(y, imgB, par, j, lg);
    __syncthreads();                                         // (also: everybody has read image 0 - it may be rewritten)
    store_image(img + IMG_FLOATS, g.P, rows, tid);
    zero(y);
    layer(g.w + WLat::N_W2, x, y, par, lane);
    put_tiles(y, imgA, par, j, lg);
    __syncthreads();                                         // (also: the P rows have left image 1)
    store_image(img, g.U, rows, tid);
    zero(y);
    layer(g.w + WLat::N_W3, x, y, par, lane);
    put_tiles(y, imgB, par, j, lg);
    __syncthreads();
    store_image(img + IMG_FLOATS, g.V, rows, tid);
}

// ------------------------------------------------------------------------------------------------ propagate chain
struct NDev {
    const float* w;
    const float* feat12; float* eff; const float* P; float* U; float* V; const float* C;
    const float* Uin; const float* Vin; const float* c_eff; const float* c_P;
    const int* send; const int* row_ptr; const int* deg; const int* n_guard; const uint8_t* vmask;
    const int* rowlist; const int* n_rows;
    float* pred_pos; float* pred_motion;
    int B, N, n_p, edge_cap, c_cap, ell_stride, dedupe, cls_on, first_round, N_o, M;
    unsigned self_row; float clamp;
};
__device__ __forceinline__ long dense_row(const NDev& g, long slot, long nslots) {
    const long s = slot < nslots ? slot : nslots - 1;
    return g.rowlist ? (long)g.rowlist[s] : s;
}
__device__ __forceinline__ long cls_row(const NDev& g, int b, int i) {
    if (i >= g.N_o) return 2L * g.N_o + (long)b * g.M + (i - g.N_o);
    return g.vmask[(long)b * g.N + i] ? i : g.N_o + i;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
struct PassRow { int i, deg, b, e0; };
// message passing for 8 rows (8 lanes per row), the 128-B tiles [T0, T0+NTL) of every row: acc = sum over the row's edges,
// in CSR order, of ReLU((C + U) + V) - same order, same bits as ag_mlp.hip: gather_agg.  Result -> image rows.
template <int T0, int NTL>
__device__ __forceinline__ void gather_sweep(const NDev& g, const PassRow& r, bool cls, int kmax, int idx0, int idx1,
                                             const int* __restrict__ snd, float* img_rows8, int lane) {
    const int rr = lane >> 3, c = lane & 7;
    const unsigned N_o = g.N_o;
    const unsigned tool0 = N_o + (unsigned)r.b * g.M;                        // + particle index (>= N_o) = class row
    const unsigned vb = cls ? 0u : (unsigned)r.b * (unsigned)g.N;
    const unsigned urow = cls ? (r.i >= (int)N_o ? tool0 + r.i : (unsigned)r.i) : vb + (unsigned)r.i;
    const unsigned selfrow = g.self_row + (r.i >= (int)N_o ? 1u : 0u);
    const unsigned cb = (unsigned)r.b * (unsigned)g.c_cap + (unsigned)r.e0;
    f32x4 u[NTL], acc[NTL];
    const float* up = g.Uin + urow * (unsigned)NFP + 32 * T0 + 4 * c;
#pragma unroll
    for (int t = 0; t < NTL; ++t) { u[t] = *reinterpret_cast<const f32x4*>(up + 32 * t); acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // four edges per trip: their 8 * NTL loads are issued together (a single small graph is latency-bound: one edge per
    // trip made the row's edges a chain of ~0.7-us round trips), then summed in CSR order - same additions, same bits.
    // Edges past the row's (or the wave's) last one read the self-loop constant row / the row's own V row: valid
    // addresses, values unused.
    constexpr int EB = 4;
    for (int k0 = 0; k0 < kmax; k0 += EB) {
        f32x4 cv[EB][NTL], vv[EB][NTL];
#pragma unroll
        for (int q = 0; q < EB; ++q) {
            const int k = k0 + q;                                            // wave-uniform
            const bool on = k < r.deg;
            int sj;
            if (k < 16) sj = __shfl(k < 8 ? idx0 : idx1, (lane & 56) + (k & 7), 64);
            else sj = on ? snd[k] : r.i;
            const unsigned crow = (!on || (g.dedupe && sj == r.i)) ? selfrow : cb + (unsigned)k;
            const unsigned vrow = cls ? (sj >= (int)N_o ? tool0 + (unsigned)sj : (unsigned)sj) : vb + (unsigned)sj;
            const float* cp = g.C + crow * (unsigned)NFP + 32 * T0 + 4 * c;
            const float* vp = g.Vin + vrow * (unsigned)NFP + 32 * T0 + 4 * c;
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                cv[q][t] = *reinterpret_cast<const f32x4*>(cp + 32 * t);
                vv[q][t] = *reinterpret_cast<const f32x4*>(vp + 32 * t);
            }
        }
#pragma unroll
        for (int q = 0; q < EB; ++q) {
            const bool on = k0 + q < r.deg;
#pragma unroll
            for (int t = 0; t < NTL; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float m = fmaxf((cv[q][t][e] + u[t][e]) + vv[q][t][e], 0.0f);
                    acc[t][e] += on ? m : 0.0f;
                }
        }
    }
    float* wp = img_rows8 + rr * IMG_PITCH + 32 * T0 + 4 * c;
#pragma unroll
    for (int t = 0; t < NTL; ++t) *reinterpret_cast<f32x4*>(wp + 32 * t) = acc[t];
}

// eff <- ReLU(Wb*agg + P + eff); not last: U = W2*eff, V = W3*eff; last: predictor, clamp, integrate (model.py:307-338)
template <bool LAST>
__global__ __launch_bounds__(WGL, 2) void k_node_prop_lat(NDev g) {
    __shared__ __attribute__((aligned(16))) float img[2 * IMG_FLOATS];
    __shared__ int rows[ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rg = wave >> 1, par = wave & 1, j = lane & 15, lg = lane >> 4;
    const long nrows = g.n_rows ? (long)*g.n_rows : (long)g.B * g.N;
    const long slot0 = (long)blockIdx.x * ROWS;
    if (slot0 >= nrows) return;
    float* imgA = img + rg * 16 * IMG_PITCH;
    float* imgB = imgA + IMG_FLOATS;
    const bool cls = g.cls_on && g.first_round;
    const bool ell = g.ell_stride != 0;
    // ---- message passing: wave (rg, par) aggregates its row group's 16 rows in two passes of 8; par 0 takes the 128-B
    // tiles 0..2 of every row, par 1 the tiles 3..4; the image then holds agg for the 32 rows
    {
        const int rr = lane >> 3, c = lane & 7;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const long slot = slot0 + 16 * rg + 8 * p + rr;
            const bool rv = slot < nrows;
            const long row = dense_row(g, slot, nrows);
            PassRow r;
            r.b = (int)(row / g.N); r.i = (int)(row - (long)r.b * g.N);
            const int* dp = ell ? g.deg + row : g.row_ptr + (long)r.b * (g.N + 1) + r.i;
            const int d0 = dp[0], d1 = ell ? 0 : dp[1];
            r.e0 = ell ? r.i * g.ell_stride : d0;
            int deg = ell ? d0 : d1 - d0;
            const int guard = g.n_guard ? g.n_guard[r.b] : 1;                // overflowed caller graph: not to be followed
            if (!rv || guard == 0) deg = 0;
            if (guard == 0) r.e0 = 0;
            r.deg = deg;
            const int* snd = g.send + (long)r.b * g.edge_cap + r.e0;
            const int lastk = max(deg - 1, 0), room = g.edge_cap - 1 - r.e0;  // (a degree-0 row at e0 == edge_cap reads the slot before)
            int idx0 = snd[min(min(c, lastk), room)], idx1 = snd[min(min(8 + c, lastk), room)];
            if (c >= deg) idx0 = r.i;
            if (8 + c >= deg) idx1 = r.i;
            const int kmax = wave_max(deg);
            if (par == 0) gather_sweep<0, 3>(g, r, cls, kmax, idx0, idx1, snd, imgA + 8 * p * IMG_PITCH, lane);
            else gather_sweep<3, 2>(g, r, cls, kmax, idx0, idx1, snd, imgA + 8 * p * IMG_PITCH, lane);
        }
    }
    // ---- this lane's row for the chain
    const long slot = slot0 + 16 * rg + j;
    const bool valid = slot < nrows;
    const long row = dense_row(g, slot, nrows);
    const int pb = (int)(row / g.N), pi = (int)(row - (long)pb * g.N);
    const long crow = g.cls_on ? cls_row(g, pb, pi) : row;
    const bool ceff = g.cls_on && g.first_round;                             // round 1: the previous effect is p_enc itself
    if (par == 0 && lg == 0) rows[16 * rg + j] = valid ? (int)row : -1;
    // the residual terms seed the accumulator: y = P + eff (own tiles), then y += Wb*agg
    Out y;
    {
        const float* pp = (g.cls_on ? g.c_P : g.P) + crow * NFP;
        const float* ep = (ceff ? g.c_eff : g.eff) + (ceff ? crow : row) * NFP;
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int T = par + 2 * m, f = 16 * T + foff(lg, r);
                y.t[m][r] = (T < 9 || r < 2) ? pp[f] + ep[f] : 0.0f;
            }
    }
    __syncthreads();                                                         // agg image complete
    Act x;
    get_act(x, imgA, j, lg);
    layer(g.w + WLat::P_WB, x, y, par, lane);
    relu_one(y, par, lg);                                                    // y = new particle effect (bias slot = 1)
    put_tiles(y, imgB, par, j, lg);
    __syncthreads();
    get_act(x, imgB, j, lg);                                                 // x = eff: input of W2, W3 / the predictor
    if (!LAST) {
        store_image(img + IMG_FLOATS, g.eff, rows, tid);
        zero(y);
        layer(g.w + WLat::N_W2, x, y, par, lane);
        put_tiles(y, imgA, par, j, lg);                                      // image 0: everybody read agg before the last barrier
        __syncthreads();
        store_image(img, g.U, rows, tid);
        zero(y);
        layer(g.w + WLat::N_W3, x, y, par, lane);
        put_tiles(y, imgB, par, j, lg);                                      // image 1: eff was read and stored before the last barrier
        __syncthreads();
        store_image(img + IMG_FLOATS, g.V, rows, tid);
    } else {
        zero(y);
        layer(g.w + WLat::P_P0, x, y, par, lane);
        relu_one(y, par, lg);
        put_tiles(y, imgA, par, j, lg);
        __syncthreads();
        get_act(x, imgA, j, lg);
        zero(y);
        layer(g.w + WLat::P_P1, x, y, par, lane);
        relu_one(y, par, lg);
        put_tiles(y, imgB, par, j, lg);
        __syncthreads();
        get_act(x, imgB, j, lg);
        if (par == 0) {                                                      // head: 3 outputs = rows 0..2 of one tile
            f32x4 m = {0.f, 0.f, 0.f, 0.f};
            const float* wp = g.w + WLat::P_P2 + lane * 4;
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(wp + c * 256);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int s = 4 * c + e;
                    if (s < KS) {
                        const int T = s < 36 ? s / 4 : 9, r = s < 36 ? s % 4 : s - 36;
                        m = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], x.t[T][r], m, 0, 0, 0);
                    }
                }
            }
            // motion xyz = output rows 0,1,2 = registers 0,1,2 of lane group 0
            if (valid && lg == 0 && pi < g.n_p) {
                const float* cur = g.feat12 + row * F12 + 9;                 // state[:, -1]  (model.py:338)
                const long o = ((long)pb * g.n_p + pi) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float mo = m[c];
                    g.pred_motion[o + c] = mo;
                    g.pred_pos[o + c] = cur[c] + fminf(fmaxf(mo, -g.clamp), g.clamp);
                }
            }
        }
    }
}

}  // namespace lat

size_t lat_weights_floats() { return lat::WLat::TOTAL; }
size_t lat_weights_offset(int which) {   // 0 E_L1, 1 E_L2, 2 E_L3, 3 E_W1, 4 P_WB, 5 N_W2, 6 N_W3, 7 P_P0, 8 P_P1, 9 P_P2, 10 N_L1, 11 N_L2, 12 N_L3, 13 N_WA
    using W = lat::WLat;
    const int o[14] = {W::E_L1, W::E_L2, W::E_L3, W::E_W1, W::P_WB, W::N_W2, W::N_W3, W::P_P0, W::P_P1, W::P_P2, W::N_L1, W::N_L2, W::N_L3, W::N_WA};
    return o[which];
}
hipError_t launch_node_enc_lat(const float* wl, const GraphBufs& g, long row0, long nrows, hipStream_t st) {
    lat::NEDev d{wl, g.node_in, g.eff, g.P, g.UV[1][0], g.UV[1][1], 0, (long)g.B * g.N};   // (what to_dev gives k_node_enc)
    if (g.cls_on) {   // encode a slice of the class table instead of all B*N rows
        d.node_in = g.c_node_in; d.eff = g.c_eff; d.P = g.c_P; d.U = g.c_U; d.V = g.c_V;
        d.row0 = row0; d.nrows = nrows;
    }
    if (d.nrows <= 0) return hipSuccess;
    hipLaunchKernelGGL(lat::k_node_enc_lat, dim3((unsigned)((d.nrows + lat::ROWS - 1) / lat::ROWS)), dim3(lat::WGL), 0, st, d);
    return hipGetLastError();
}
hipError_t launch_edge_enc_lat(const float* wl, const GraphBufs& g, hipStream_t st) {
    lat::EDev d{wl, g.node_in, g.feat12, g.group, g.C, g.recv, g.send, g.n_edges, g.ns_edge, g.n_ns,
                g.B, g.N, g.n_inst, g.edge_cap, g.c_cap};
    const long rows = (long)g.B * g.c_cap;
    hipLaunchKernelGGL(lat::k_edge_enc_lat, dim3((unsigned)(rows / lat::ROWS)), dim3(lat::WGL), 0, st, d);
    return hipGetLastError();
}
hipError_t launch_node_prop_lat(const float* wl, const GraphBufs& g, int round, bool last, float clamp, float* pred_pos,
                                float* pred_motion, hipStream_t st) {
    lat::NDev d{};
    const bool cls = g.cls_on && round == 0;
    d.w = wl; d.feat12 = g.feat12; d.eff = g.eff; d.P = g.P; d.C = g.C;
    d.Uin = cls ? g.c_U : g.UV[(round - 1) & 1][0]; d.Vin = cls ? g.c_V : g.UV[(round - 1) & 1][1];
    d.U = g.UV[round & 1][0]; d.V = g.UV[round & 1][1];
    d.c_eff = g.c_eff; d.c_P = g.c_P; d.send = g.send; d.row_ptr = g.row_ptr; d.deg = g.deg; d.n_guard = g.n_guard;
    d.vmask = g.vmask; d.rowlist = g.rowlist; d.n_rows = g.n_rows; d.pred_pos = pred_pos; d.pred_motion = pred_motion;
    d.B = g.B; d.N = g.N; d.n_p = g.n_p; d.edge_cap = g.edge_cap; d.c_cap = g.c_cap; d.ell_stride = g.ell_stride;
    d.dedupe = g.c_self ? 1 : 0; d.cls_on = g.cls_on; d.first_round = round == 0; d.N_o = g.N_o; d.M = g.M;
    d.self_row = (unsigned)g.self_row; d.clamp = clamp;
    const long slots = (long)g.B * g.N + (g.rowlist ? g.N_o : 0);
    const dim3 grid((unsigned)((slots + lat::ROWS - 1) / lat::ROWS));
    if (last) hipLaunchKernelGGL(lat::k_node_prop_lat<true>, grid, dim3(lat::WGL), 0, st, d);
    else hipLaunchKernelGGL(lat::k_node_prop_lat<false>, grid, dim3(lat::WGL), 0, st, d);
    return hipGetLastError();
}

}  // namespace ag
