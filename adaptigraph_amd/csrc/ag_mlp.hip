// Fused MLP chains on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950 only.
//
// Replaces Encoder / Propagator / ParticlePredictor forwards (reference src/dynamics/gnn/model.py:4-61) and the
// feature assembly + dense one-hot bmm gathers around them (model.py:152-294, 312-330, 335-338).
//
// Design (MI355X-first, not a translation of nn.Linear + bmm):
//   * The transposed problem Y^T = W * X^T is computed: the MFMA A operand is a 32(out-feature) x 2(k) weight
//     sliver, the B operand is 2(k) x 32(rows).  A wavefront owns 32 rows (edges or particles); after a layer the
//     accumulator holds, for row = lane&31, output features spread over its 80 registers (5 tiles x 16) and the
//     two lane halves.  That IS the B-operand layout of the next layer (k on lane-half, row on lane&31) - so a
//     whole Linear-ReLU-Linear-... chain runs with activations never leaving registers: no LDS round trip, no
//     cross-lane movement.  The K order this implies (feature 32t + (r&3) + 8(r>>2) + 4h for register (t,r),
//     half h) is baked into the host-side weight packing (ag_api.hip: pack_layer).
//   * Bias rides in the contraction: activation slot 150 is forced to 1.0, weight column 150 holds the bias.
//   * The 160x152 weight panel of a layer (95 KB) is shared by the 4 wavefronts of a 256-thread workgroup through
//     LDS, staged in four K-quarters so the next quarter is DMA'd (global_load_lds) into the other buffer underneath
//     the MFMAs of the current one (two 25 KB buffers, no staging registers).  One ds_read_b128 feeds 4 MFMAs per m-block.  A workgroup is one
//     wavefront per SIMD and 50 KB of LDS, so TWO workgroups share a CU and run unsynchronised: the prologue
//     gathers, mid-chain row loads, epilogue stores and barrier waits of one hide under the MFMAs of the other
//     (measured: one 8-wave workgroup per CU left the MFMA pipe idle 18-41 % of the time).
//   * Roofline: fp32 MFMA (157.3 TFLOP/s).  Per 32 rows a 160-wide layer is 380 MFMAs = 2*32*160*152 FLOP.
//   * Rows are independent columns of the MFMA, so results do not depend on which lane / workgroup / chunk / GPU a
//     row lands in: sharded == unsharded bit-for-bit.
#include "ag_common.h"
#include <cstdint>

namespace ag {
#ifdef AG_DIAG   // in-kernel clock / phase probes of the diagnostic build (ag_diag.hip); the product library has none of it
unsigned long long* diag_edge_begin(void* diag, unsigned nwg, hipStream_t st);
void diag_edge_end(void* diag, unsigned nwg, hipStream_t st);
unsigned long long* diag_node_begin(void* diag, unsigned nwg, hipStream_t st);
void diag_node_end(void* diag, unsigned nwg, int round, hipStream_t st);
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WG = 256;                     // 4 wavefronts, one per SIMD; two workgroups share a CU
constexpr int WG_ROWS = 128;
constexpr int QCH = 5;                      // chunks (of 4 k-steps) per staged quarter: 5,5,5,4
constexpr int BUF_FLOATS = QCH * CHUNK_FLOATS_MB5;   // one LDS staging buffer (25,600 B)
constexpr int Q_FLOATS = BUF_FLOATS;        // quarters 0..2
constexpr int Q3_FLOATS = (KCH - 3 * QCH) * CHUNK_FLOATS_MB5;
// Per-wavefront LDS staging area behind the two weight buffers (5,248 B each; a workgroup is 51,200 + 20,992 = 72,192 B,
// two workgroups per CU = 144 KB of the 160 KB).  Used wave-locally - no barrier, LDS operations of one wavefront execute
// in order - for the two layout changes between "row-major in memory" and "accumulator layout in registers":
//   * store: one 32-feature tile of the wave's 32 rows at a time (32 x 128 B at a 144-B pitch), written in accumulator
//     layout, read back so that 8 lanes hold one row's 128 B: every global store instruction writes 8 whole cache lines
//     instead of 64 16-B fragments in 32 lines;
//   * message passing: 8 aggregated rows at a time (8 x 640 B at a 656-B pitch), written feature-on-lane by the
//     gather, read back in B-operand layout.
constexpr int STG_TILE_PITCH = 36;          // floats; 36 j mod 32 = 4 j: the 8 lanes of a ds_write_b128 group hit 32 banks once
constexpr int STG_ROW_PITCH = 164;          // floats; 8 rows x 656 B
constexpr int STG_ROWS = 8;
constexpr int STG_FLOATS = STG_ROWS * STG_ROW_PITCH;       // 1,312 >= 32 * STG_TILE_PITCH
static_assert(32 * STG_TILE_PITCH <= STG_FLOATS, "tile staging fits the per-wave area");
constexpr int CHAIN_LDS_FLOATS = 2 * BUF_FLOATS + (WG / 64) * STG_FLOATS;

struct Act { f32x16 t[5]; };

// ------------------------------------------------------------------------------------------------ staging
// global -> LDS DMA (global_load_lds, 16 B per lane): NFLOATS (a multiple of 256) in 1-KB pieces, each wave-instruction
// writing 64 x 16 B at a wave-uniform LDS base + lane*16.  Asynchronous; awaited by the vmcnt(0) that __syncthreads()
// emits while a DMA is in flight.  Uses no staging registers and no ds_write.
template <int NFLOATS>
__device__ __forceinline__ void dma_copy(float* lds_dst, const float* __restrict__ src, int tid) {
    static_assert(NFLOATS % 256 == 0, "whole 1-KB pieces");
    constexpr int NP = NFLOATS / 256, NW = WG / 64;
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int i = 0; i < (NP + NW - 1) / NW; ++i) {
        const int piece = wave + NW * i;                    // wave-uniform
        if (piece < NP) {
            const float* gp = src + piece * 256 + lane * 4;
            float* lp = lds_dst + piece * 256;
            __builtin_amdgcn_global_load_lds(
                reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(gp)),
                reinterpret_cast<__attribute__((address_space(3))) void*>(static_cast<unsigned>(reinterpret_cast<uintptr_t>(lp))),
                16, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ MFMA sweeps
// chunks [Q0,Q1) of a layer whose LDS image starts at chunk Q0; input = previous accumulator tiles.
// (Reading the fragments of chunk q+1 under the MFMAs of chunk q was measured twice - with two MFMA-issuing wavefronts per
// SIMD, and in the fused propagate chains where the partner wavefront is gathering half of the time: no gain, +20 VGPRs.)
template <int Q0, int Q1, int MB>
__device__ __forceinline__ void mma_act(const float* wl, const Act& in, f32x16* acc, int lane) {
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
        f32x4 a[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            a[mb] = *reinterpret_cast<const f32x4*>(wl + ((q - Q0) * MB + mb) * 256 + lane * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = 4 * q + e;
            const int t = s < 64 ? s / 16 : 4;
            const int r = s < 64 ? s % 16 : s - 64;
            const float b = in.t[t][r];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][e], b, acc[mb], 0, 0, 0);
        }
        // one m-block: nothing else stops the scheduler from hoisting all 19 weight reads (76 VGPRs)
        if (MB == 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// first layer: input features f[0 .. 8*NCH) held per lane; step s consumes (f[2s], f[2s+1]) on the two lane halves
// NST: k-steps that carry an input or the bias slot; later steps of the last chunk multiply zeros and are skipped
template <int NCH, int NST = 4 * NCH>
__device__ __forceinline__ void mma_feat(const float* wl, const float* f, f32x16* acc, int lane) {
    const bool hi = lane >= 32;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        f32x4 a[5];
#pragma unroll
        for (int mb = 0; mb < 5; ++mb) a[mb] = *reinterpret_cast<const f32x4*>(wl + (q * 5 + mb) * 256 + lane * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = 4 * q + e;
            if (s >= NST) continue;
            const float b = hi ? f[2 * s + 1] : f[2 * s];
#pragma unroll
            for (int mb = 0; mb < 5; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][e], b, acc[mb], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void zero(Act& a) {
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a.t[t][r] = 0.0f;
}
// Forces the 80 values to exist in registers HERE.  Without it the compiler sinks elementwise work (the residual
// adds + ReLU) into the next layer's MFMA sweep, next to each first use, and keeps the loaded operands alive by
// spilling ~100 VGPRs to scratch (seen in k_node_prop<true>, which has no store that would pin the values).
__device__ __forceinline__ void materialize(Act& a) {
#pragma unroll
    for (int t = 0; t < 5; ++t) asm volatile("" : "+v"(a.t[t]));
}
__device__ __forceinline__ void relu_one(Act& a, int lane) {   // ReLU, then force slot 150 (tile 4, reg 10, upper half) to 1
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a.t[t][r] = __builtin_fmaxf(a.t[t][r], 0.0f);
    a.t[4][10] = lane >= 32 ? 1.0f : a.t[4][10];
    materialize(a);
}

// row-major (pitch NFP) <-> accumulator layout.  Lane (j = lane&31, h = lane>>5) owns, of row j, the 16-byte
// groups at feature 32t + 8q + 4h (registers 4q..4q+3 of tile t).
__device__ __forceinline__ void load_rows(Act& a, const float* __restrict__ base, long row, int lane) {
    const float* p = base + row * NFP + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + 32 * t + 8 * q);
            a.t[t][4 * q + 0] = v[0]; a.t[t][4 * q + 1] = v[1]; a.t[t][4 * q + 2] = v[2]; a.t[t][4 * q + 3] = v[3];
        }
}
// returns 0, but only after `v` has been computed, and the compiler cannot see that it is 0
__device__ __forceinline__ long pin_after(Act& a) {
    int z = 0;
    float v = a.t[0][0];
    asm volatile("" : "+v"(z), "+v"(v));
    a.t[0][0] = v;
    return z;
}
template <int T0, int T1>
__device__ __forceinline__ void add_rows_part(Act& a, const float* __restrict__ base, long row, int lane) {
    const float* p = base + row * NFP + 4 * (lane >> 5);
#pragma unroll
    for (int t = T0; t < T1; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + 32 * t + 8 * q);
            a.t[t][4 * q + 0] += v[0]; a.t[t][4 * q + 1] += v[1]; a.t[t][4 * q + 2] += v[2]; a.t[t][4 * q + 3] += v[3];
        }
}
__device__ __forceinline__ void store_rows(const Act& a, float* __restrict__ base, long row, int lane, bool valid) {
    if (!valid) return;
    float* p = base + row * NFP + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
            v[0] = a.t[t][4 * q + 0]; v[1] = a.t[t][4 * q + 1]; v[2] = a.t[t][4 * q + 2]; v[3] = a.t[t][4 * q + 3];
            *reinterpret_cast<f32x4*>(p + 32 * t + 8 * q) = v;
        }
    // keep the next layer's accumulator writes behind these stores (else the 80 source registers stay live under a
    // renamed accumulator and the kernel spills)
    __builtin_amdgcn_sched_barrier(0);
}

// Row-major store of the accumulator layout through the wave's LDS staging area, tile by tile.  `myrow`: the row
// (of `base`, pitch NFP) that lane&31 of this wave owns; `valid`: whether that row exists.  Lane L stores, for every tile,
// the 16-B piece (L&7) of the rows owned by lanes (L>>3) + 8i, i = 0..3: one instruction = 8 rows x 128 B, whole lines.
__device__ __forceinline__ void store_rows_t(const Act& a, float* __restrict__ base, int myrow, bool valid, float* stg, int lane) {
    const int j = lane & 31, h = lane >> 5, sr = lane >> 3, sc = lane & 7;
    int srow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = __shfl(valid ? myrow : -1, sr + 8 * i, 64);
        srow[i] = r;
    }
    float* wp = stg + j * STG_TILE_PITCH + 4 * h;
    const float* rp = stg + sr * STG_TILE_PITCH + 4 * sc;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
            v[0] = a.t[t][4 * q + 0]; v[1] = a.t[t][4 * q + 1]; v[2] = a.t[t][4 * q + 2]; v[3] = a.t[t][4 * q + 3];
            *reinterpret_cast<f32x4*>(wp + 8 * q) = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rp + 8 * i * STG_TILE_PITCH);
            if (srow[i] >= 0) *reinterpret_cast<f32x4*>(base + (long)srow[i] * NFP + 32 * t + 4 * sc) = v;
        }
    }
    // keep the next layer's accumulator writes behind these stores (else the 80 source registers stay live under a
    // renamed accumulator and the kernel spills)
    __builtin_amdgcn_sched_barrier(0);
}

// One full 160-wide layer, weights streamed in four K-quarters through two LDS buffers.
// Precondition: quarter 0 is in buffer 0 and a barrier has passed.  Every sweep runs with the NEXT quarter's
// global->register loads in flight; the registers are written to the other buffer after the sweep.
// Postcondition: the first NEXT floats of `next` (the following phase) are in buffer 0 and a barrier has passed.
// PRE: the caller has already issued the DMA of quarter 1 into buffer 1 (it did so BEFORE a burst of row stores: waits
// count vector-memory operations in issue order, so a DMA issued behind the stores could not be awaited without them).
template <int NEXT, bool ZERO = true, bool PRE = false>
__device__ __forceinline__ void layer160(float* lds, const float* __restrict__ w, const float* __restrict__ next,
                                         const Act& in, Act& out, int tid, int lane) {
    float* b0 = lds;
    float* b1 = lds + BUF_FLOATS;
    if (ZERO) zero(out);                                     // else: accumulate on top of what `out` holds
    // every sweep runs with the NEXT quarter's DMA in flight into the other buffer (free since the last barrier)
    if (!PRE) dma_copy<Q_FLOATS>(b1, w + Q_FLOATS, tid);
    mma_act<0, QCH, 5>(b0, in, out.t, lane);
    __syncthreads();
    dma_copy<Q_FLOATS>(b0, w + 2 * Q_FLOATS, tid);
    mma_act<QCH, 2 * QCH, 5>(b1, in, out.t, lane);
    __syncthreads();
    dma_copy<Q3_FLOATS>(b1, w + 3 * Q_FLOATS, tid);
    mma_act<2 * QCH, 3 * QCH, 5>(b0, in, out.t, lane);
    __syncthreads();
    if (NEXT > 0) dma_copy<(NEXT > 0 ? NEXT : 256)>(b0, next, tid);
    mma_act<3 * QCH, KCH, 5>(b1, in, out.t, lane);
    if (NEXT > 0) __syncthreads();
}

template <int NFLOATS>
__device__ __forceinline__ void stage_now(float* dst, const float* __restrict__ src, int tid) {
    dma_copy<NFLOATS>(dst, src, tid);
    __syncthreads();
}

struct GDev {
    const float* w;
    const float* node_in; const float* feat12; const float* group;
    float* eff; float* P; float* U; float* V; float* C;
    const int* recv; const int* send; const int* row_ptr; const int* n_edges;
    int B, N, n_p, n_inst, edge_cap, c_cap;
    float clamp; float* pred_pos; float* pred_motion;
    int cls_on, N_o, M, first_round; const uint8_t* vmask;
    const float* c_eff; const float* c_P;
    long row0, nrows;          // k_node_enc: slice of rows to encode
    const int* ns_edge; const int* n_ns;   // k_edge_enc: non-self-loop edge list (null = every edge)
    const float* wb3;                      // bf16x3 weight image (null = exact fp32 mode)
    // message passing fused into the propagate chains (gather_agg): inputs of THIS round (Uin/Vin: the class table in the
    // first round of a rollout step, else what the previous round's chain wrote) - never the buffers this launch writes
    const float* Uin; const float* Vin; const int* deg; int ell_stride; int dedupe; unsigned self_row; const int* n_guard;
    int stagger_ticks; unsigned first_wave;   // see stagger_second_workgroup
    int reverse;                              // k_node_prop: walk the row tiles from the last to the first (Options::zigzag)
    unsigned n_tiles;                         // k_edge_enc: > 0 = persistent workgroups (AG_ENC_PERSIST) over this many tiles
    // ragged batches (masked rollouts): the propagate chains walk a compact list of the rows that exist - valid object
    // particles and tools, plus one phantom candidate that stands for every masked-out particle (GraphBufs) - instead of
    // all B*N rows; rowlist[slot] = dense row b*N + i, *n_rows = number of slots.  Null: every dense row, in order.
    const int* rowlist; const int* n_rows;
    // first forward of a dynamics() call (GraphBufs::send_pk): `send` then holds sender | (position + 1) << 12, and a slot with
    // non-zero upper bits takes its C row from the shared base table (row = receiver * share_kb + position)
    const float* C_share; unsigned share_kb;
    int f_pitch, cur_off;                     // feature-row pitch and offset of the current position in it (n_his 4: 12, 9)
#ifdef AG_DIAG
    unsigned long long* dbg;   // diagnostic build (ag_diag.hip) only: stamps per workgroup, never read by kernels
#endif
};

using WL = WeightLayout;

// ------------------------------------------------------------------------------------------------ message passing
// agg[i] = sum over edges e with recv(e) = i of ReLU(C[e] + U[i] + V[send(e)])
//   = Rr^T.bmm(relation_propagator([rel_enc | eff_r | eff_s]))                       (reference model.py:312-324)
// with W_rp factored as [W1|W2|W3]: C = W1*rel_enc + b, U = W2*eff, V = W3*eff.
// Fused into the propagate chain: every wavefront aggregates the 32 rows it is about to push through Wb, so `agg` never
// exists in HBM and the gather - HBM / L2 bound, no matrix work - of one workgroup runs under the MFMAs of the other
// workgroup on the same CU (the two are kept half a period apart, see k_node_prop).
// Shape: 8 rows at a time; the 8 lanes (c = lane&7) of a row read the eight 16-B pieces of one 128-B line, so a wave
// instruction fetches 8 whole lines (C row of one edge of 8 different receivers, tile t); five tiles cover the 640-B
// row.  Each lane sums its 4 features edge after edge in CSR (= reference nonzero) order: deterministic, no atomics,
// bit-identical to a sequential segmented sum.  The 8 aggregated rows go to the wave's LDS staging area and come back
// in B-operand layout (row on lane&31, features on registers).
// Sender indices: the 8 lanes of a row load 8 consecutive indices with one instruction and hand them round with
// ds_bpermute, so the index fetch is off the critical path of every edge but the first of a block.
// Algorithmic bytes per receiver: deg*(640 C + 640 V + 4 idx) + 640 U.
// work-list slot -> dense row (b*N + i); slots past the end repeat the last one (their results are never stored)
__device__ __forceinline__ long dense_row(const GDev& g, long slot, long nslots) {
    const long s = slot < nslots ? slot : nslots - 1;
    return g.rowlist ? (long)g.rowlist[s] : s;
}

struct EdgeBuf { f32x4 c[5], v[5]; };
// per-pass state of gather_agg that the load issue needs (all per lane; the row of this lane's 8-lane group)
struct PassRow {
    int i, deg, b, e0;       // particle, in-degree, candidate, first edge slot; everything else is re-derived per use
};
// issue the 10 loads of edge k of the lane's row (C row + V row, five 128-B tiles each, this lane's 16-B piece);
// sjp = the edge's sender; SHARE (first forward of a dynamics() call): sender in the low 12 bits, above them position + 1
// in the receiver's row of the base graph when the C row is the shared table's (GraphBufs::send_pk)
template <bool SHARE>
__device__ __forceinline__ void gather_addr(const GDev& g, const PassRow& r, bool cls, int k, int sjp, const float* __restrict__ C,
                                            const float* __restrict__ V, int lane, const float*& cp, const float*& vp) {
    const unsigned c4 = 4u * (lane & 7);
    const bool on = k < r.deg;
    const int sj = SHARE ? (sjp & 0xfff) : sjp;
    // lanes whose row has no edge k read the self-loop constant row and their own V row: valid addresses, values unused.
    // Class-table rows (first round of a rollout step): a particle that takes part in an edge is valid by construction
    // (masked pairs never pass the radius test, graph.py:253-256), so its row is a pure function of its index.
    const unsigned selfrow = g.self_row + (r.i >= g.N_o ? 1u : 0u);          // C row of the self-loop constant
    const unsigned crow = (!on || (g.dedupe && sj == r.i)) ? selfrow : (unsigned)r.b * (unsigned)g.c_cap + (unsigned)(r.e0 + k);
    const unsigned tool0 = (unsigned)g.N_o + (unsigned)r.b * g.M;            // + particle index (>= N_o) = class row
    const unsigned vrow = cls ? (sj >= g.N_o ? tool0 + (unsigned)sj : (unsigned)sj) : (unsigned)r.b * (unsigned)g.N + (unsigned)sj;
    cp = C + crow * (unsigned)NFP + c4;
    if (SHARE) {   // same bits as the candidate's own row would hold: a row's chain does not depend on where it is computed
        const unsigned sp = (unsigned)sjp >> 12;
        const float* sh = g.C_share + ((unsigned)r.i * g.share_kb + (sp - 1u)) * (unsigned)NFP + c4;
        cp = (on && sp != 0u) ? sh : cp;
    }
    vp = V + vrow * (unsigned)NFP + c4;
}
template <bool SHARE>
__device__ __forceinline__ void gather_issue(const GDev& g, const PassRow& r, bool cls, int k, int sjp, EdgeBuf& buf,
                                             const float* __restrict__ C, const float* __restrict__ V, int lane) {
    const float *cp, *vp;
    gather_addr<SHARE>(g, r, cls, k, sjp, C, V, lane, cp, vp);
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        buf.c[t] = *reinterpret_cast<const f32x4*>(cp + 32 * t);
        buf.v[t] = *reinterpret_cast<const f32x4*>(vp + 32 * t);
    }
}
// acc += ReLU((c + u) + v) for the lanes whose row has this edge; a select, not a branch (adding +0 leaves a sum of
// non-negative terms unchanged, and straight-line code lets the compiler count outstanding loads exactly)
__device__ __forceinline__ void gather_consume(const EdgeBuf& buf, const f32x4* u, f32x4* acc, bool on) {
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float m = fmaxf((buf.c[t][e] + u[t][e]) + buf.v[t][e], 0.0f);
            acc[t][e] += on ? m : 0.0f;
        }
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

template <bool SHARE>
__device__ __forceinline__ void gather_agg(const GDev& g, float* stg, long wave_row0, long nrows, Act& x, int lane) {
    const int rr = lane >> 3, c = lane & 7, j = lane & 31, h = lane >> 5;
    const bool cls = g.cls_on && g.first_round;
    const float* __restrict__ C = g.C;
    const float* __restrict__ U = g.Uin;
    const float* __restrict__ V = g.Vin;
    // per pass: this lane's receiver (candidate, first slot, degree) and its first block of sender indices (8 per block:
    // lane c of a row holds index 8*block + c) - fetched up front so that only the C / V rows themselves are on the
    // critical path of a pass
    int pb[4], pe0[4], pdeg[4], pidx0[4], pkmax[4];
    // Branch-free and batched: every load below has a clamped, always-valid address and is issued unconditionally, its
    // value selected afterwards - two dependent round trips for the whole wave (slot-indexed rollout graphs: one,
    // the sender indices do not depend on the degree) instead of two per pass behind divergent branches.
    long prow[4]; bool prv[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long slot = wave_row0 + 8 * p + rr;
        prv[p] = slot < nrows;
        prow[p] = dense_row(g, slot, nrows);
        pb[p] = (int)(prow[p] / g.N);
    }
    const bool ell = g.ell_stride != 0;                        // wave-uniform
    int d0[4], d1[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = (int)(prow[p] - (long)pb[p] * g.N);
        const int* dp = ell ? g.deg + prow[p] : g.row_ptr + (long)pb[p] * (g.N + 1) + i;
        d0[p] = dp[0];
        d1[p] = ell ? 0 : dp[1];                               // (deg is followed by other workspace arrays: in bounds)
        if (ell) pe0[p] = i * g.ell_stride;
    }
    if (ell) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {                          // issued together with the degree loads
            const long o = (long)pb[p] * g.edge_cap + min(pe0[p] + c, g.edge_cap - 1);
            pidx0[p] = g.send[o];
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int deg = ell ? d0[p] : d1[p] - d0[p];
        if (!ell) pe0[p] = d0[p];
        const int guard = g.n_guard ? g.n_guard[pb[p]] : 1;     // overflowed caller graph: row_ptr is not to be trusted
        if (!prv[p] || guard == 0) deg = 0;
        if (guard == 0) pe0[p] = 0;                             // its offsets are not to be followed either
        pdeg[p] = deg;
    }
    if (!ell) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int e = min(max(pe0[p], 0) + c, g.edge_cap - 1);
            pidx0[p] = g.send[(long)pb[p] * g.edge_cap + e];
        }
    }
    auto row_of = [&](int p) {
        PassRow r;
        r.b = pb[p]; r.e0 = pe0[p]; r.deg = pdeg[p];
        r.i = (int)(prow[p] - (long)pb[p] * g.N);
        return r;
    };
    auto senders_of = [&](const PassRow& r) { return g.send + (long)r.b * g.edge_cap + r.e0; };
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (c >= pdeg[p]) pidx0[p] = (int)(prow[p] - (long)pb[p] * g.N);
        pkmax[p] = wave_max(pdeg[p]);
    }
    const int N_o = g.N_o;
    auto issue_u = [&](const PassRow& r, f32x4* u) {
        const unsigned urow = cls ? (r.i >= N_o ? (unsigned)N_o + (unsigned)r.b * g.M + r.i : (unsigned)r.i)
                                  : (unsigned)r.b * (unsigned)g.N + (unsigned)r.i;
        const float* up = U + urow * (unsigned)NFP + 4u * c;
#pragma unroll
        for (int t = 0; t < 5; ++t) u[t] = *reinterpret_cast<const f32x4*>(up + 32 * t);
    };
    // Rolling pipeline, one edge ahead: while edge k is summed, the 10 loads of edge k+1 are in flight in the other
    // buffer; the U row and the first edge of the NEXT pass are issued before this pass's rows take their trip through
    // LDS, so a pass boundary does not drain the memory pipe either.  The steady-state loop body is straight-line (no
    // conditional issue, no other vector load), which is what lets the compiler wait with exact vmcnt(10)s.
    constexpr int KFAST = 32;                                                  // edges per row served by the pipeline
    f32x4 u[5];
    EdgeBuf A, B;
    PassRow r = row_of(0);
    int idx0 = pidx0[0];
    issue_u(r, u);
    if (pkmax[0] > 0) gather_issue<SHARE>(g, r, cls, 0, __shfl(idx0, lane & 56, 64), A, C, V, lane);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int kmax = pkmax[p], kfast = min(kmax, KFAST);
        // blocks 1..3 of the sender indices (rows with more than 8 edges: rope, the dense granular graphs); in flight
        // from here, first used at edge 8
        const int* snd = senders_of(r);
        const int lastk = max(r.deg - 1, 0);
        // (offsets clamped to the candidate's slots like pidx0: a degree-0 row at e0 == edge_cap must not read past the array)
        const int room = g.edge_cap - 1 - r.e0;
        int idx1 = snd[min(min(8 + c, lastk), room)], idx2 = snd[min(min(16 + c, lastk), room)], idx3 = snd[min(min(24 + c, lastk), room)];
        if (8 + c >= r.deg) idx1 = r.i;
        if (16 + c >= r.deg) idx2 = r.i;
        if (24 + c >= r.deg) idx3 = r.i;
        auto sender = [&](int k) {
            const int blk = k >> 3;                                            // wave-uniform
            const int v = blk == 0 ? idx0 : blk == 1 ? idx1 : blk == 2 ? idx2 : idx3;
            return __shfl(v, (lane & 56) + (k & 7), 64);
        };
        f32x4 acc[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 2 < kfast; k += 2) {
            gather_issue<SHARE>(g, r, cls, k + 1, sender(k + 1), B, C, V, lane);
            gather_consume(A, u, acc, k < r.deg);
            gather_issue<SHARE>(g, r, cls, k + 2, sender(k + 2), A, C, V, lane);
            gather_consume(B, u, acc, k + 1 < r.deg);
        }
        const int left = kfast - k;                                            // 0 (no edge at all), 1 or 2
        if (left == 2) gather_issue<SHARE>(g, r, cls, k + 1, sender(k + 1), B, C, V, lane);
        if (left >= 1) gather_consume(A, u, acc, k < r.deg);
        if (left == 2) gather_consume(B, u, acc, k + 1 < r.deg);
        for (k = KFAST; k < kmax; ++k) {                                       // rows beyond 32 edges: plain loop
            const int sj = k < r.deg ? snd[k] : r.i;
            gather_issue<SHARE>(g, r, cls, k, sj, A, C, V, lane);
            gather_consume(A, u, acc, k < r.deg);
        }
        if (p + 1 < 4) {                                                       // next pass: U row and first edge
            r = row_of(p + 1);
            idx0 = pidx0[p + 1];
            issue_u(r, u);
            if (pkmax[p + 1] > 0) gather_issue<SHARE>(g, r, cls, 0, __shfl(idx0, lane & 56, 64), A, C, V, lane);
        }
        float* wp = stg + rr * STG_ROW_PITCH + 4 * c;
#pragma unroll
        for (int t = 0; t < 5; ++t) *reinterpret_cast<f32x4*>(wp + 32 * t) = acc[t];
        if ((j >> 3) == p) {
            const float* rp = stg + (j & 7) * STG_ROW_PITCH + 4 * h;
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(rp + 32 * t + 8 * q);
                    x.t[t][4 * q + 0] = v[0]; x.t[t][4 * q + 1] = v[1]; x.t[t][4 * q + 2] = v[2]; x.t[t][4 * q + 3] = v[3];
                }
        }
    }
}

// class-table row of particle i of candidate b (see GraphBufs)
__device__ __forceinline__ long cls_row(const GDev& g, int b, int i) {
    if (i >= g.N_o) return 2L * g.N_o + (long)b * g.M + (i - g.N_o);
    return g.vmask[(long)b * g.N + i] ? i : g.N_o + i;
}

// Two workgroups share a CU.  Each alternates a memory-bound phase (the gather) with a matrix-bound phase (the chain);
// started together they stay in lockstep - both gathering, then both computing - and nothing overlaps.  Started half a
// period apart, one gathers while the other has the matrix pipe to itself, and since both take equally long the offset
// persists for every later workgroup that inherits their slots.  The offset is created once per launch: of the
// workgroups resident from the start (blockIdx < first_wave: placement is breadth-first, measured with
// tools/probes/wg_placement.hip) the one that was given the SECOND LDS allocation of its CU (LDS_BASE != 0 in
// HW_REG_LDS_ALLOC) sleeps for `stagger_ticks` x 10 ns.  Speed only: any placement gives the same results.
__device__ __forceinline__ void stagger_second_workgroup(const GDev& g) {
    if (g.stagger_ticks > 0 && blockIdx.x < g.first_wave) {
        const unsigned lds_base = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (7 << 11));   // HW_REG_LDS_ALLOC[7:0]
        if (lds_base != 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)g.stagger_ticks) __builtin_amdgcn_s_sleep(64);
        }
    }
}

// ------------------------------------------------------------------------------------------------ edge chain
// rel_inputs (17) -> Encoder(17,150,150) -> W1*enc + b_rp  => C      (model.py:249-282, 303, 317-318 first block)
// NH = history frames: 4 (rel_inputs 17) or, on the forward path only, 5 (rel_inputs 20: softbody.yaml)
template <int NH>
__global__ __launch_bounds__(WG, 2) void k_edge_enc(GDev g) {
    constexpr int FP = NH == 5 ? F15_PITCH : F12, NQ = FP / 4, RD = 5 + 3 * NH;
    __shared__ __attribute__((aligned(16))) float lds[CHAIN_LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* stg = lds + 2 * BUF_FLOATS + wave * STG_FLOATS;
    // tile-major: block = tile * B + candidate.  Workgroups are dealt to XCDs / shader engines in a fixed rotation;
    // with candidate-major order every candidate's early-exit tail (tiles past its edge count) lands on the same
    // engines and the others carry all the work (measured: 16.7 % fewer rows, same kernel time).  Tile-major puts
    // every early-exit workgroup at the end of the grid.
    // Tile loop.  By default the grid has one workgroup per tile and the loop runs once; AG_ENC_PERSIST=n launches n
    // persistent workgroups that walk the tiles blk, blk + n, ... instead (measured: no faster alone - 1.525 vs 1.50 ms per
    // launch - and worse beside other streams, whose kernels can then only enter at this kernel's end).  The loop FORM is
    // kept for what it does to hipcc's code: with the tile body inside a loop whose trip count it cannot see, and the
    // weight base made opaque per tile (or the 40 DMA piece addresses are hoisted into the preheader and spill: 106
    // VGPRs), the kernel comes out 1.4 % faster than the straight-line version (A/B on the same box, twice on three
    // boxes: 1.526 -> 1.504 ms per launch; 212 instead of 206 VGPRs, no scratch).
    const unsigned n_tiles = g.n_tiles ? g.n_tiles : gridDim.x;
    for (unsigned blk = blockIdx.x; blk < n_tiles; blk += gridDim.x) {
    int zi = 0;
    asm volatile("" : "+s"(zi));                           // per-tile opaque zero (see above)
    const float* const wts = g.w + zi;
    const int b = (int)(blk % (unsigned)g.B);
    const int e0 = (int)(blk / (unsigned)g.B) * WG_ROWS;
    const int ne = g.n_ns ? g.n_ns[b] : g.n_edges[b];       // rows to encode: all edges, or the non-self-loop ones
    if (e0 >= ne) continue;                                  // whole workgroup past this candidate's edges
#ifdef AG_DIAG
    if (g.dbg && tid == 0) {
        g.dbg[blk * 4 + 0] = __builtin_amdgcn_s_memtime();
        g.dbg[blk * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    // small first-layer panel -> buffer 1; under its MFMAs, L2 half 0 -> buffer 0
    stage_now<EDGE_L1_CHUNKS * CHUNK_FLOATS_MB5>(lds + BUF_FLOATS, wts + WL::E_L1, tid);
    dma_copy<Q_FLOATS>(lds, wts + WL::E_L2, tid);            // lands under the feature gather and the L1 sweep

    const int t = e0 + wave * 32 + (lane & 31);
    const bool valid = t < ne;
    const int el = g.ns_edge ? g.ns_edge[(long)b * g.edge_cap + (valid ? t : 0)] : t;   // edge id (C row) of this lane
    const int elc = valid ? el : (g.ns_edge ? el : 0);
    const int r = g.recv[(long)b * g.edge_cap + elc], s = g.send[(long)b * g.edge_cap + elc];
    const long pr = (long)b * g.N + r, ps = (long)b * g.N + s;
    float f[8 * EDGE_L1_CHUNKS];
    {
        const float* nr = g.node_in + pr * NODE_IN; const float* nsnd = g.node_in + ps * NODE_IN;
        f[0] = nr[0]; f[1] = nr[1]; f[2] = nsnd[0]; f[3] = nsnd[1];            // attrs_r, attrs_s   model.py:253-254
        float gd = 0.0f;
        for (int k = 0; k < g.n_inst; ++k) gd += fabsf(g.group[pr * g.n_inst + k] - g.group[ps * g.n_inst + k]);
        f[4] = gd;                                                               // model.py:264-267
        const f32x4* fr = reinterpret_cast<const f32x4*>(g.feat12 + pr * FP);
        const f32x4* fs = reinterpret_cast<const f32x4*>(g.feat12 + ps * FP);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4 a = fr[q], c = fs[q];
            f[5 + 4 * q + 0] = a[0] - c[0]; f[5 + 4 * q + 1] = a[1] - c[1];     // pos_r - pos_s       model.py:277-279
            f[5 + 4 * q + 2] = a[2] - c[2]; f[5 + 4 * q + 3] = a[3] - c[3];
        }
        f[RD] = 1.0f;                                                            // bias slot behind the RD relation inputs
#pragma unroll
        for (int k = RD + 1; k < 8 * EDGE_L1_CHUNKS; ++k) f[k] = 0.0f;
    }
    Act x, y;
    zero(y);
    mma_feat<EDGE_L1_CHUNKS, (RD + 2) / 2>(lds + BUF_FLOATS, f, y.t, lane);     // RD inputs + bias: 9 (n_his 4) / 11 steps of 12
    __syncthreads();
    relu_one(y, lane);
    layer160<Q_FLOATS>(lds, wts + WL::E_L2, wts + WL::E_L3, y, x, tid, lane);
    relu_one(x, lane);
    layer160<Q_FLOATS>(lds, wts + WL::E_L3, wts + WL::E_W1, x, y, tid, lane);
    relu_one(y, lane);
    layer160<0>(lds, wts + WL::E_W1, nullptr, y, x, tid, lane);
    store_rows_t(x, g.C, (int)((long)b * g.c_cap + el), valid, stg, lane);
#ifdef AG_DIAG
    if (g.dbg && tid == 0) {
        g.dbg[blk * 4 + 2] = __builtin_amdgcn_s_memtime();
        g.dbg[blk * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    __syncthreads();                                         // the weight buffers are restaged by the next tile
    }
}

// ------------------------------------------------------------------------------------------------ node encode chain
// p_inputs (6) -> Encoder(6,150,150) = p_enc => eff;  P = Wa*p_enc + b_pp;  U = W2*p_enc;  V = W3*p_enc
// (model.py:297-298 and the particle_effect-dependent blocks of :317-318 / the particle_encode block of :328-330)
__global__ __launch_bounds__(WG, 2) void k_node_enc(GDev g) {
    __shared__ __attribute__((aligned(16))) float lds[CHAIN_LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* stg = lds + 2 * BUF_FLOATS + wave * STG_FLOATS;
    const long rend = g.row0 + g.nrows;
    const long row = g.row0 + (long)blockIdx.x * WG_ROWS + wave * 32 + (lane & 31);
    const bool valid = row < rend;
    const long rowc = valid ? row : rend - 1;

    stage_now<NODE_L1_CHUNKS * CHUNK_FLOATS_MB5>(lds + BUF_FLOATS, g.w + WL::N_L1, tid);
    dma_copy<Q_FLOATS>(lds, g.w + WL::N_L2, tid);
    float f[8];
    {
        const f32x4* p = reinterpret_cast<const f32x4*>(g.node_in + rowc * NODE_IN);
        const f32x4 a = p[0], c = p[1];
        f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = c[0]; f[5] = c[1]; f[6] = c[2]; f[7] = c[3];
    }
    Act x, y;
    zero(y);
    mma_feat<NODE_L1_CHUNKS>(lds + BUF_FLOATS, f, y.t, lane);
    __syncthreads();
    relu_one(y, lane);
    layer160<Q_FLOATS>(lds, g.w + WL::N_L2, g.w + WL::N_L3, y, x, tid, lane);
    relu_one(x, lane);
    layer160<Q_FLOATS>(lds, g.w + WL::N_L3, g.w + WL::N_WA, x, y, tid, lane);
    relu_one(y, lane);                                       // y = p_enc (slot 150 = 1 for the bias of Wa)
    store_rows_t(y, g.eff, (int)row, valid, stg, lane);
    layer160<Q_FLOATS>(lds, g.w + WL::N_WA, g.w + WL::N_W2, y, x, tid, lane);
    store_rows_t(x, g.P, (int)row, valid, stg, lane);
    layer160<Q_FLOATS>(lds, g.w + WL::N_W2, g.w + WL::N_W3, y, x, tid, lane);
    store_rows_t(x, g.U, (int)row, valid, stg, lane);
    layer160<0>(lds, g.w + WL::N_W3, nullptr, y, x, tid, lane);
    store_rows_t(x, g.V, (int)row, valid, stg, lane);
}

// ------------------------------------------------------------------------------------------------ propagate chain
// eff <- ReLU(Wb*agg + P + eff)   (model.py:328-330; P carries Wa*p_enc + b_pp)
//   not last: U = W2*eff, V = W3*eff for the next round (model.py:312-318)
//   last:     motion = ParticlePredictor(eff) (model.py:44-61, 335); pred = cur + clamp(motion) (model.py:338)
template <bool LAST, bool SHARE>
__global__ __launch_bounds__(WG, 2) void k_node_prop(GDev g) {
    __shared__ __attribute__((aligned(16))) float lds[CHAIN_LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* stg = lds + 2 * BUF_FLOATS + wave * STG_FLOATS;
    const long nrows = g.n_rows ? (long)*g.n_rows : (long)g.B * g.N;     // work-list slots (== dense rows without a list)
    const unsigned bid = g.reverse ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
    if ((long)bid * WG_ROWS >= nrows) return;                             // whole workgroup past the end of a ragged batch
    const long slot = (long)bid * WG_ROWS + wave * 32 + (lane & 31);
    const bool valid = slot < nrows;
    const long rowc = dense_row(g, slot, nrows);
    const long row = rowc;

    stagger_second_workgroup(g);
#ifdef AG_DIAG
    if (g.dbg && tid == 0) {
        g.dbg[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));        // HW_REG_HW_ID: cu [11:8], se [15:13]
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));       // HW_REG_XCC_ID[3:0]
        g.dbg[(unsigned long)gridDim.x * 4 + blockIdx.x] = (xcc << 8) | (((hw >> 13) & 7) << 4) | ((hw >> 8) & 15);
        g.dbg[(unsigned long)gridDim.x * 5 + 2 * blockIdx.x] = __builtin_amdgcn_s_memtime();
    }
#endif
    dma_copy<Q_FLOATS>(lds, g.w + WL::P_WB, tid);          // first quarter of Wb lands under the gather / row loads
    Act x, y;
    // The residual terms seed the accumulator: y = P + eff, then y += Wb*agg.  All three row loads are issued here,
    // together, instead of two of them stalling the chain after the Wb layer.
    gather_agg<SHARE>(g, stg, (long)bid * WG_ROWS + wave * 32, nrows, x, lane);
    __builtin_amdgcn_sched_barrier(0);                       // nothing of what follows is worth a register during the gather
#ifdef AG_DIAG
    if (g.dbg && tid == 0) g.dbg[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    const int pb = (int)(rowc / g.N), pi = (int)(rowc - (long)pb * g.N);
    const long crow = g.cls_on ? cls_row(g, pb, pi) : rowc;
    const bool ceff = g.cls_on && g.first_round;             // round 1: the previous effect is p_enc itself
    load_rows(y, g.cls_on ? g.c_P : g.P, crow, lane);
    // the third operand lands in temporaries: two batches keep it inside the register budget
    add_rows_part<0, 2>(y, ceff ? g.c_eff : g.eff, ceff ? crow : rowc, lane);
    materialize(y);
    add_rows_part<2, 5>(y, ceff ? g.c_eff : g.eff, (ceff ? crow : rowc) + pin_after(y), lane);
    __syncthreads();                                         // Wb quarter 0 is in buffer 0
    layer160<Q_FLOATS, false>(lds, g.w + WL::P_WB, g.w + (LAST ? WL::P_P0 : WL::N_W2), x, y, tid, lane);
    relu_one(y, lane);                                       // y = new particle effect (slot 150 forced to 1)
#ifdef AG_DIAG
    if (g.dbg && tid == 0) g.dbg[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
    if (!LAST) {
        // quarter 1 of the next layer is requested BEFORE each burst of row stores, so that the wait for it at the next
        // barrier leaves the stores in flight: they then have two quarter sweeps to drain instead of one
        dma_copy<Q_FLOATS>(lds + BUF_FLOATS, g.w + WL::N_W2 + Q_FLOATS, tid);
        store_rows_t(y, g.eff, (int)row, valid, stg, lane);
        layer160<Q_FLOATS, true, true>(lds, g.w + WL::N_W2, g.w + WL::N_W3, y, x, tid, lane);
        dma_copy<Q_FLOATS>(lds + BUF_FLOATS, g.w + WL::N_W3 + Q_FLOATS, tid);
        store_rows_t(x, g.U, (int)row, valid, stg, lane);
        layer160<0, true, true>(lds, g.w + WL::N_W3, nullptr, y, x, tid, lane);
        store_rows_t(x, g.V, (int)row, valid, stg, lane);
#ifdef AG_DIAG
        if (g.dbg && tid == 0) {
            g.dbg[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
            g.dbg[(unsigned long)gridDim.x * 5 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
        }
#endif
    } else {
        layer160<Q_FLOATS>(lds, g.w + WL::P_P0, g.w + WL::P_P1, y, x, tid, lane);
        relu_one(x, lane);
        layer160<OUT3_FLOATS>(lds, g.w + WL::P_P1, g.w + WL::P_P2, x, y, tid, lane);
        relu_one(y, lane);
        f32x16 m[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) m[0][r] = 0.0f;
        mma_act<0, KCH, 1>(lds, y, m, lane);
        // motion xyz = output features 0,1,2 = registers 0,1,2 of lanes 0..31
        const int b = pb, i = pi;
        if (valid && lane < 32 && i < g.n_p) {
            const float* cur = g.feat12 + rowc * g.f_pitch + g.cur_off;    // state[:, -1]  (model.py:338)
            const long o = ((long)b * g.n_p + i) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float mo = m[0][c];
                g.pred_motion[o + c] = mo;
                g.pred_pos[o + c] = cur[c] + fminf(fmaxf(mo, -g.clamp), g.clamp);
            }
        }
    }
}


// =====================================================================================================================
// "bf16x3" mode (opt-in, ag_ctx_set_precision): the same chains on the bf16 matrix pipe with fp32-grade accuracy.
// Every fp32 operand is split exactly into three bf16 pieces (x = xh + xm + xl, 8+8+8 mantissa bits); a product is
// rebuilt from the six partial products whose weight is >= 2^-24 of it (xl*wh, xh*wl, xm*wm, xm*wh, xh*wm, xh*wh -
// bf16 x bf16 is exact in fp32), accumulated in the fp32 MFMA accumulator, smallest first.  Dropped terms are
// <= 2^-24 relative, the size of an fp32 rounding.  v_mfma_f32_32x32x16_bf16 runs 16x the fp32-MFMA FLOP rate, so six
// of them per fp32-equivalent step is ~2.7x faster.  The accumulator layout is the same as for the fp32 MFMA (C/D
// maps are dtype-independent on gfx950), so the register-chaining scheme carries over: K-step (tile t, half u) takes
// accumulator registers 8u..8u+7 of tile t, i.e. features 32t + 16u + (j&3) + 8(j>>2) + 4h for element j of lane-half
// h - again folded into the host-side weight image.
// Weights: [phase][k-step in phase (2)][m-block][part h/m/l][lane][8 bf16]; one phase = one 32-feature tile of K
// = 30,720 B; a 160-wide layer = 5 phases; the 3-output head = 1 phase of 10 k-steps; first layers = 1 phase.
// =====================================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int PH_FLOATS = 2 * 5 * 3 * 64 * 4;         // one staged phase, in floats (30,720 B)
struct WLB {                                          // phase index of every layer in the bf16x3 weight image
    static constexpr int E_L1 = 0, E_L2 = 1, E_L3 = 6, E_W1 = 11;
    static constexpr int N_L1 = 16, N_L2 = 17, N_L3 = 22, N_WA = 27, N_W2 = 32, N_W3 = 37;
    static constexpr int P_WB = 42, P_P0 = 47, P_P1 = 52, P_P2 = 57, TOTAL = 58;
};

// exact 3-way split of 8 fp32 values into bf16 pieces (round-to-nearest at every level; remainders are exact)
__device__ __forceinline__ void split8(const float* x, bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 hi = (__bf16)x[i];
        const float r = x[i] - (float)hi;
        const __bf16 mi = (__bf16)r;
        const float q = r - (float)mi;
        h[i] = hi; m[i] = mi; l[i] = (__bf16)q;
    }
}
// NK k-steps of one phase: B pieces in bh/bm/bl[NK]; A pieces read from the LDS image [ks][mb][part][lane]
template <int NK, int MB>
__device__ __forceinline__ void mma_b3(const float* wl, const bf16x8* bh, const bf16x8* bm, const bf16x8* bl,
                                       f32x16* acc, int lane) {
    const bf16x8* w = reinterpret_cast<const bf16x8*>(wl);
    // weight pieces double-buffered one m-block ahead and pinned there: left alone the scheduler hoists the reads
    // of several m-blocks (12 VGPRs each) over the live accumulators and spills
    bf16x8 ah = w[0 * 64 + lane], am = w[1 * 64 + lane], al = w[2 * 64 + lane];
#pragma unroll
    for (int it = 0; it < NK * MB; ++it) {
        const int ks = it / MB, mb = it % MB;
        bf16x8 nh = ah, nm = am, nl = al;
#ifndef AG_B3_NOLDS
        if (it + 1 < NK * MB) {
            nh = w[((it + 1) * 3 + 0) * 64 + lane];
            nm = w[((it + 1) * 3 + 1) * 64 + lane];
            nl = w[((it + 1) * 3 + 2) * 64 + lane];
        }
#else   // timing-only experiment (wrong results): one weight read per k-step; the other m-blocks' pieces are made from it by
        // one opaque register move each, so that the MFMAs stay distinct but LDS delivers a fifth of the bytes
        asm volatile("" : "+v"(nh), "+v"(nm), "+v"(nl));
#endif
        __builtin_amdgcn_sched_barrier(0);
        f32x16 c = acc[mb];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ks], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ks], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[ks], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[ks], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[ks], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ks], c, 0, 0, 0);
        acc[mb] = c;
        __builtin_amdgcn_sched_barrier(0);
        ah = nh; am = nm; al = nl;
    }
}
// ---- weight-unit stream (staged by LDS-DMA).  Every chain is a fixed sequence of 30,720-B weight UNITS (a first layer,
// one 32-feature K tile of a 160-wide layer, or the 3-output head).  A 256-thread workgroup (4 wavefronts, 128 rows)
// walks the stream with a 2-slot LDS ring: while unit g is consumed from slot g&1, unit g+1 is DMA'd (global_load_lds)
// straight into the other slot, which held unit g-1 - finished by every wavefront before the barrier that ended it.
// One barrier per unit; the DMA has the whole unit (60 MFMAs per wavefront) to land and uses no staging registers.
// Activations never cross wavefronts, so the units of consecutive layers simply follow each other.  Two workgroups
// share a CU and cover each other's prologue and barrier waits.  (Measured alternatives: register staging per half
// unit - prefetch window too short, waves parked 24 %; 8-wave workgroups with a 4-slot ring - no faster, every
// prologue exposed.)
#ifdef AG_B3_WG512   // A/B experiment (r06, correct results): ONE 8-wavefront workgroup per CU, its eight waves share one weight ring
constexpr int WGB = 512, WGB_ROWS = 256, WGB_PER_CU = 1;
#else
constexpr int WGB = 256;                    // 4 wavefronts; two workgroups per CU cover each other's prologue / barriers
constexpr int WGB_ROWS = 128;
constexpr int WGB_PER_CU = 2;
#endif
constexpr int NSLOT = 2;                    // LDS ring slots of one unit each (2 x 30,720 B per workgroup)
constexpr int UNIT_FLOATS = PH_FLOATS;

// global -> LDS DMA of one unit (30 pieces of 1 KB: each wave-instruction moves 64 lanes x 16 B to a wave-uniform LDS
// base + lane*16).  Asynchronous: completion is awaited by the vmcnt(0) that __syncthreads() emits while a DMA is in
// flight.  No staging registers, no ds_write.
__device__ __forceinline__ void dma_unit(float* lds_slot, const float* __restrict__ src, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int i = 0; i < (UNIT_FLOATS / 256 + WGB / 64 - 1) / (WGB / 64); ++i) {
        const int piece = wave + (WGB / 64) * i;            // wave-uniform
        if (piece < UNIT_FLOATS / 256) {
            const float* gp = src + piece * 256 + lane * 4;
            float* lp = lds_slot + piece * 256;
            __builtin_amdgcn_global_load_lds(
                reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(gp)),
                reinterpret_cast<__attribute__((address_space(3))) void*>(static_cast<unsigned>(reinterpret_cast<uintptr_t>(lp))),
                16, 0, 0);
        }
    }
}
// unit tables (indices into the bf16x3 image, WLB): kind 0 edge chain, 1 node encode, 2 propagate, 3 propagate + head
template <int KIND> __device__ __forceinline__ constexpr int unit_of(int g) {
    return KIND == 0 ? WLB::E_L1 + g
         : KIND == 1 ? WLB::N_L1 + g
         : KIND == 2 ? (g < 5 ? WLB::P_WB + g : g < 10 ? WLB::N_W2 + (g - 5) : WLB::N_W3 + (g - 10))
                     : WLB::P_WB + g;
}
template <int KIND> struct UnitCount { static constexpr int N = KIND == 0 ? 16 : KIND == 1 ? 26 : KIND == 2 ? 15 : 16; };

// consume unit G: prefetch G+2, run `body(wl)` on slot G&3, publish, barrier after odd units
template <int KIND, int G, class Body>
__device__ __forceinline__ void unit(float* lds, const float* __restrict__ W, int tid, Body body) {
    constexpr int NU = UnitCount<KIND>::N;
    // the other slot held unit G-1: every wavefront finished it before the barrier that ended that unit
#ifdef AG_B3_NODMA   // timing-only experiment (wrong results): only the first two units are streamed, the ring is then re-used as it is
    if (G + 1 < 2)
#else
    if (G + 1 < NU)
#endif
        dma_unit(lds + ((G + 1) % NSLOT) * UNIT_FLOATS, W + unit_of<KIND>(G + 1) * UNIT_FLOATS, tid);
    body(lds + (G % NSLOT) * UNIT_FLOATS);
    if (G + 1 < NU) __syncthreads();                           // waits for this unit's DMA, then publishes it
}
template <int KIND>
__device__ __forceinline__ void stream_begin(float* lds, const float* __restrict__ W, int tid) {
    dma_unit(lds, W + unit_of<KIND>(0) * UNIT_FLOATS, tid);
    __syncthreads();
}
// K tile T of a 160-wide layer: out += W[:, tile T] * in.t[T]   (two k-steps, 60 MFMAs)
template <int T>
__device__ __forceinline__ void tile_b3(const float* wl, const Act& in, Act& out, int lane) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = in.t[T][8 * u + j];
        bf16x8 bh, bm, bl;
        split8(x, bh, bm, bl);
        mma_b3<1, 5>(wl + u * (5 * 3 * 64 * 4), &bh, &bm, &bl, out.t, lane);
    }
}
// five consecutive units G0..G0+4 = one 160-wide layer
template <int KIND, int G0>
__device__ __forceinline__ void layer_b3(float* lds, const float* __restrict__ W, const Act& in, Act& out, int tid, int lane) {
    unit<KIND, G0 + 0>(lds, W, tid, [&](const float* wl) { tile_b3<0>(wl, in, out, lane); });
    unit<KIND, G0 + 1>(lds, W, tid, [&](const float* wl) { tile_b3<1>(wl, in, out, lane); });
    unit<KIND, G0 + 2>(lds, W, tid, [&](const float* wl) { tile_b3<2>(wl, in, out, lane); });
    unit<KIND, G0 + 3>(lds, W, tid, [&](const float* wl) { tile_b3<3>(wl, in, out, lane); });
    unit<KIND, G0 + 4>(lds, W, tid, [&](const float* wl) { tile_b3<4>(wl, in, out, lane); });
}
// first layer from NF per-lane input features (zero beyond NF): k-step u, lane-half h, element j <-> feature 16u+8h+j
template <int NK, int NF>
__device__ __forceinline__ void first_b3(const float* wl, const float* f, f32x16* acc, int lane) {
    const bool hi = lane >= 32;
#pragma unroll
    for (int u = 0; u < NK; ++u) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float lo_v = 16 * u + j < NF ? f[(16 * u + j) < NF ? (16 * u + j) : 0] : 0.0f;
            const float hi_v = 16 * u + 8 + j < NF ? f[(16 * u + 8 + j) < NF ? (16 * u + 8 + j) : 0] : 0.0f;
            x[j] = hi ? hi_v : lo_v;
        }
        bf16x8 bh, bm, bl;
        split8(x, bh, bm, bl);
        mma_b3<1, 5>(wl + u * (5 * 3 * 64 * 4), &bh, &bm, &bl, acc, lane);
    }
}

__global__ __launch_bounds__(WGB, WGB_PER_CU) void k_edge_enc_b3(GDev g) {
    __shared__ __attribute__((aligned(16))) float lds[NSLOT * UNIT_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (int)(blockIdx.x % (unsigned)g.B);
    const int e0 = (int)(blockIdx.x / (unsigned)g.B) * WGB_ROWS;
    const int ne = g.n_ns ? g.n_ns[b] : g.n_edges[b];
    if (e0 >= ne) return;
    const float* W = g.wb3;
    stream_begin<0>(lds, W, tid);
    const int t = e0 + wave * 32 + (lane & 31);
    const bool valid = t < ne;
    const int el = g.ns_edge ? g.ns_edge[(long)b * g.edge_cap + (valid ? t : 0)] : t;
    const int elc = valid ? el : (g.ns_edge ? el : 0);
    const int r = g.recv[(long)b * g.edge_cap + elc], s = g.send[(long)b * g.edge_cap + elc];
    const long pr = (long)b * g.N + r, ps = (long)b * g.N + s;
    float f[18];
    {
        const float* nr = g.node_in + pr * NODE_IN; const float* nsnd = g.node_in + ps * NODE_IN;
        f[0] = nr[0]; f[1] = nr[1]; f[2] = nsnd[0]; f[3] = nsnd[1];
        float gd = 0.0f;
        for (int k = 0; k < g.n_inst; ++k) gd += fabsf(g.group[pr * g.n_inst + k] - g.group[ps * g.n_inst + k]);
        f[4] = gd;
        const f32x4* fr = reinterpret_cast<const f32x4*>(g.feat12 + pr * F12);
        const f32x4* fs = reinterpret_cast<const f32x4*>(g.feat12 + ps * F12);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const f32x4 a = fr[q], c = fs[q];
            f[5 + 4 * q + 0] = a[0] - c[0]; f[5 + 4 * q + 1] = a[1] - c[1];
            f[5 + 4 * q + 2] = a[2] - c[2]; f[5 + 4 * q + 3] = a[3] - c[3];
        }
        f[17] = 1.0f;
    }
    Act x, y;
    zero(y);
    unit<0, 0>(lds, W, tid, [&](const float* wl) { first_b3<2, 18>(wl, f, y.t, lane); });
    relu_one(y, lane);
    zero(x);
    layer_b3<0, 1>(lds, W, y, x, tid, lane);
    relu_one(x, lane);
    zero(y);
    layer_b3<0, 6>(lds, W, x, y, tid, lane);
    relu_one(y, lane);
    zero(x);
    layer_b3<0, 11>(lds, W, y, x, tid, lane);
    store_rows(x, g.C, (long)b * g.c_cap + el, lane, valid);
}

__global__ __launch_bounds__(WGB, WGB_PER_CU) void k_node_enc_b3(GDev g) {
    __shared__ __attribute__((aligned(16))) float lds[NSLOT * UNIT_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long rend = g.row0 + g.nrows;
    const long row = g.row0 + (long)blockIdx.x * WGB_ROWS + wave * 32 + (lane & 31);
    const bool valid = row < rend;
    const long rowc = valid ? row : rend - 1;
    const float* W = g.wb3;
    stream_begin<1>(lds, W, tid);
    float f[8];
    {
        const f32x4* p = reinterpret_cast<const f32x4*>(g.node_in + rowc * NODE_IN);
        const f32x4 a = p[0], c = p[1];
        f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = c[0]; f[5] = c[1]; f[6] = c[2]; f[7] = c[3];
    }
    Act x, y;
    zero(y);
    unit<1, 0>(lds, W, tid, [&](const float* wl) { first_b3<1, 8>(wl, f, y.t, lane); });
    relu_one(y, lane);
    zero(x);
    layer_b3<1, 1>(lds, W, y, x, tid, lane);
    relu_one(x, lane);
    zero(y);
    layer_b3<1, 6>(lds, W, x, y, tid, lane);
    relu_one(y, lane);
    store_rows(y, g.eff, row, lane, valid);
    zero(x);
    layer_b3<1, 11>(lds, W, y, x, tid, lane);
    store_rows(x, g.P, row, lane, valid);
    zero(x);
    layer_b3<1, 16>(lds, W, y, x, tid, lane);
    store_rows(x, g.U, row, lane, valid);
    zero(x);
    layer_b3<1, 21>(lds, W, y, x, tid, lane);
    store_rows(x, g.V, row, lane, valid);
}

template <bool LAST, bool SHARE>
__global__ __launch_bounds__(WGB, WGB_PER_CU) void k_node_prop_b3(GDev g) {
    __shared__ __attribute__((aligned(16))) float lds[NSLOT * UNIT_FLOATS];
    constexpr int KIND = LAST ? 3 : 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long nrows = g.n_rows ? (long)*g.n_rows : (long)g.B * g.N;
    if ((long)blockIdx.x * WGB_ROWS >= nrows) return;
    const long slot = (long)blockIdx.x * WGB_ROWS + wave * 32 + (lane & 31);
    const bool valid = slot < nrows;
    const long rowc = dense_row(g, slot, nrows);
    const long row = rowc;
    const float* W = g.wb3;
    Act x, y;
    // fused message passing: the weight ring is not live yet, its first bytes serve as the per-wave staging areas
    gather_agg<SHARE>(g, lds + wave * STG_FLOATS, (long)blockIdx.x * WGB_ROWS + wave * 32, nrows, x, lane);
    __syncthreads();
    stream_begin<KIND>(lds, W, tid);
    const int pb = (int)(rowc / g.N), pi = (int)(rowc - (long)pb * g.N);
    const long crow = g.cls_on ? cls_row(g, pb, pi) : rowc;
    const bool ceff = g.cls_on && g.first_round;
    load_rows(y, g.cls_on ? g.c_P : g.P, crow, lane);
    add_rows_part<0, 2>(y, ceff ? g.c_eff : g.eff, ceff ? crow : rowc, lane);
    materialize(y);
    add_rows_part<2, 5>(y, ceff ? g.c_eff : g.eff, (ceff ? crow : rowc) + pin_after(y), lane);
    layer_b3<KIND, 0>(lds, W, x, y, tid, lane);            // y = P + eff + Wb*agg
    relu_one(y, lane);
    if (!LAST) {
        store_rows(y, g.eff, row, lane, valid);
        zero(x);
        layer_b3<KIND, 5>(lds, W, y, x, tid, lane);
        store_rows(x, g.U, row, lane, valid);
        zero(x);
        layer_b3<KIND, 10>(lds, W, y, x, tid, lane);
        store_rows(x, g.V, row, lane, valid);
    } else {
        zero(x);
        layer_b3<KIND, 5>(lds, W, y, x, tid, lane);
        relu_one(x, lane);
        zero(y);
        layer_b3<KIND, 10>(lds, W, x, y, tid, lane);
        relu_one(y, lane);
        f32x16 m[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) m[0][r] = 0.0f;
        unit<KIND, 15>(lds, W, tid, [&](const float* wl) {   // head: 10 k-steps, one m-block
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float xx[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xx[j] = y.t[t][8 * u + j];
                    bf16x8 bh, bm, bl;
                    split8(xx, bh, bm, bl);
                    mma_b3<1, 1>(wl + (2 * t + u) * (3 * 64 * 4), &bh, &bm, &bl, m, lane);
                }
        });
        const int b = pb, i = pi;
        if (valid && lane < 32 && i < g.n_p) {
            const float* curp = g.feat12 + rowc * F12 + 9;
            const long o = ((long)b * g.n_p + i) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float mo = m[0][c];
                g.pred_motion[o + c] = mo;
                g.pred_pos[o + c] = curp[c] + fminf(fmaxf(mo, -g.clamp), g.clamp);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
static GDev to_dev(const float* w, const GraphBufs& g) {
    GDev d;
    d.w = w; d.node_in = g.node_in; d.feat12 = g.feat12; d.group = g.group; d.eff = g.eff; d.P = g.P; d.U = g.UV[1][0];
    d.V = g.UV[1][1];; d.C = g.C; d.recv = g.recv; d.send = g.send; d.row_ptr = g.row_ptr;
    d.n_edges = g.n_edges; d.B = g.B; d.N = g.N; d.n_p = g.n_p; d.n_inst = g.n_inst; d.edge_cap = g.edge_cap;
    d.c_cap = g.c_cap; d.clamp = 0; d.pred_pos = nullptr; d.pred_motion = nullptr; d.n_tiles = 0;
    d.cls_on = g.cls_on; d.N_o = g.N_o; d.M = g.M; d.first_round = 0; d.vmask = g.vmask; d.c_eff = g.c_eff; d.c_P = g.c_P;
    d.row0 = 0; d.nrows = (long)g.B * g.N;
    d.ns_edge = g.ns_edge; d.n_ns = g.n_ns;
    d.wb3 = g.wb3;
    d.Uin = nullptr; d.Vin = nullptr; d.deg = g.deg; d.ell_stride = g.ell_stride; d.dedupe = g.c_self ? 1 : 0;
    d.self_row = (unsigned)g.self_row; d.n_guard = g.n_guard;
    d.rowlist = g.rowlist; d.n_rows = g.n_rows;
    d.C_share = nullptr; d.share_kb = 0;
    d.f_pitch = feat_pitch(g.n_his); d.cur_off = g.n_his == 5 ? 12 : 9;
    // Options::stagger_us: offset between the two workgroups of a CU in the fused propagate chains.  Off by default: it removes
    // the "both computing / both gathering" states (probe: 8 % -> 0 % of CU time) but the kernel time moves by <= 1 %
    // either way (two streams: 169 vs 171 ms per rollout with 30 us; four streams: 484.9 vs 482.9 ms per rollout without)
    d.stagger_ticks = g.stagger_us * 100; d.first_wave = 512; d.reverse = 0;
#ifdef AG_DIAG
    d.dbg = nullptr;
#endif
    return d;
}
// upper bound of the work-list length: every dense row, plus the phantom candidate's object rows of a ragged batch
static long node_slots(const GraphBufs& g) { return (long)g.B * g.N + (g.rowlist ? g.N_o : 0); }
static int node_grid(const GraphBufs& g) { return (int)((node_slots(g) + WG_ROWS - 1) / WG_ROWS); }
static int node_grid_b3(const GraphBufs& g) { return (int)((node_slots(g) + WGB_ROWS - 1) / WGB_ROWS); }

hipError_t launch_edge_enc(const float* w, const GraphBufs& g, hipStream_t st) {
    const long rows = (long)g.B * g.c_cap;
    const unsigned nwg = (unsigned)(rows / WG_ROWS);
    GDev d = to_dev(w, g);
    bool probing = false;
#ifdef AG_DIAG
    d.dbg = diag_edge_begin(g.diag, nwg, st);
    probing = d.dbg != nullptr;
#endif
    if (d.wb3) hipLaunchKernelGGL(k_edge_enc_b3, dim3((unsigned)g.B * (unsigned)((g.c_cap + WGB_ROWS - 1) / WGB_ROWS)), dim3(WGB), 0, st, d);
    else if (g.n_his == 5) hipLaunchKernelGGL(k_edge_enc<5>, dim3(nwg), dim3(WG), 0, st, d);
    else {
        unsigned grid = nwg;
        if (g.enc_persist > 0 && !probing && nwg > (unsigned)g.enc_persist) { d.n_tiles = nwg; grid = (unsigned)g.enc_persist; }
        hipLaunchKernelGGL(k_edge_enc<4>, dim3(grid), dim3(WG), 0, st, d);
    }
#ifdef AG_DIAG
    if (probing) diag_edge_end(g.diag, nwg, st);
#endif
    return hipGetLastError();
}
hipError_t launch_node_enc(const float* w, const GraphBufs& g, long row0, long nrows, hipStream_t st) {
    GDev d = to_dev(w, g);
    if (g.cls_on) {   // encode a slice of the class table instead of all B*N rows
        d.node_in = g.c_node_in; d.eff = g.c_eff; d.P = g.c_P; d.U = g.c_U; d.V = g.c_V;
        d.row0 = row0; d.nrows = nrows;
    }
    if (d.nrows <= 0) return hipSuccess;
    const dim3 grid((unsigned)((d.nrows + WG_ROWS - 1) / WG_ROWS));
    if (d.wb3) hipLaunchKernelGGL(k_node_enc_b3, dim3((unsigned)((d.nrows + WGB_ROWS - 1) / WGB_ROWS)), dim3(WGB), 0, st, d);
    else hipLaunchKernelGGL(k_node_enc, grid, dim3(WG), 0, st, d);
    return hipGetLastError();
}
// Round r of the message passing reads U/V of parity (r-1)&1 (the class table in the first round of a rollout step)
// and writes parity r&1: a workgroup's gather must never see rows another workgroup of the same launch has rewritten.
static void set_round(GDev& d, const GraphBufs& g, int round) {
    const int first = round == 0;
    d.first_round = first;
    const bool cls = g.cls_on && first;
    d.Uin = cls ? g.c_U : g.UV[(round - 1) & 1][0];         // round 0 without a class table: k_node_enc wrote parity 1
    d.Vin = cls ? g.c_V : g.UV[(round - 1) & 1][1];
    d.U = g.UV[round & 1][0]; d.V = g.UV[round & 1][1];
    d.reverse = g.zigzag ? (round & 1) : 0;
    if (g.send_pk) { d.send = g.send_pk; d.C_share = g.C_share; d.share_kb = (unsigned)g.share_kb; }   // the message passing only
}
hipError_t launch_node_prop(const float* w, const GraphBufs& g, int round, hipStream_t st) {
    GDev d = to_dev(w, g);
    set_round(d, g, round);
    const unsigned nwg = (unsigned)node_grid(g);
#ifdef AG_DIAG
    if (!d.wb3) d.dbg = diag_node_begin(g.diag, nwg, st);
#endif
    if (d.wb3) {
        if (d.C_share) hipLaunchKernelGGL((k_node_prop_b3<false, true>), dim3(node_grid_b3(g)), dim3(WGB), 0, st, d);
        else hipLaunchKernelGGL((k_node_prop_b3<false, false>), dim3(node_grid_b3(g)), dim3(WGB), 0, st, d);
    } else if (d.C_share) hipLaunchKernelGGL((k_node_prop<false, true>), dim3(nwg), dim3(WG), 0, st, d);
    else hipLaunchKernelGGL((k_node_prop<false, false>), dim3(nwg), dim3(WG), 0, st, d);
#ifdef AG_DIAG
    if (d.dbg) diag_node_end(g.diag, nwg, round, st);
#endif
    return hipGetLastError();
}
hipError_t launch_node_final(const float* w, const GraphBufs& g, int round, float clamp, float* pred_pos,
                             float* pred_motion, hipStream_t st) {
    GDev d = to_dev(w, g);
    set_round(d, g, round);
    d.clamp = clamp; d.pred_pos = pred_pos; d.pred_motion = pred_motion;
    if (d.wb3) {
        if (d.C_share) hipLaunchKernelGGL((k_node_prop_b3<true, true>), dim3(node_grid_b3(g)), dim3(WGB), 0, st, d);
        else hipLaunchKernelGGL((k_node_prop_b3<true, false>), dim3(node_grid_b3(g)), dim3(WGB), 0, st, d);
    } else if (d.C_share) hipLaunchKernelGGL((k_node_prop<true, true>), dim3(node_grid(g)), dim3(WG), 0, st, d);
    else hipLaunchKernelGGL((k_node_prop<true, false>), dim3(node_grid(g)), dim3(WG), 0, st, d);
    return hipGetLastError();
}

}  // namespace ag
