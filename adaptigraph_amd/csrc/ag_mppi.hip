// MPPI action sampling and softmax-weighted update on the device, gfx950 only.
//
// What the reference computes (src/planning/plan_utils.py): an action is (x, z, theta, length); its push runs from the
// start point (x, z) to the end point (x, z) - length*push_length*(cos theta, sin theta).  Perturbation (:42-77) and
// averaging (:80-101) both happen on the (start, end) point pairs, which are then re-encoded as (theta, length) and
// limited by wrapping theta into [-pi, pi) and clamping every component (:31-39).
//
// Here: k_mppi_update is the B-long reduction that belongs right behind the reward gather - one workgroup per
// look-ahead step, three passes over the B rewards (max, sum of exponentials, weighted point sums) with fixed-order
// LDS trees, so the result does not depend on B's sharding history; k_mppi_sample is one thread per (sample, step).
// Both are tiny and latency-bound (B <= tens of thousands, 16 B per action); no roofline applies.
// fp32 throughout, each product/sum spelled in the reference's order (-ffp-contract=off: no FMA contraction).
#include "ag_common.h"

namespace ag {

constexpr int MT = 1024;
constexpr float PI_F = 3.14159265358979323846f;          // fp32(math.pi), the scalar torch adds
constexpr float TWO_PI_F = 6.28318530717958647692f;      // fp32(2*math.pi)

// torch.remainder semantics for a positive divisor (the result takes the divisor's sign)
__device__ __forceinline__ float floor_mod(float x, float d) {
    float r = fmodf(x, d);
    if (r != 0.0f && r < 0.0f) r += d;
    return r;
}
// (theta, length) of the push start -> end, then the limits: plan_utils.py:31-39 (clip_actions).  A NaN stays a NaN, as in
// the reference (clamp_ propagates it): a NaN reward makes every softmax weight, hence the whole updated action, NaN - the
// fmaxf of the max pass drops it, but exp(NaN - m) brings it back into the normaliser - and must not come out as a
// valid-looking action at the lower limit.
__device__ __forceinline__ void limit4(float v[4], const float* lo, const float* hi) {
    v[2] = floor_mod(v[2] + PI_F, TWO_PI_F) - PI_F;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = v[k] != v[k] ? v[k] : fminf(fmaxf(v[k], lo[k]), hi[k]);   // torch.clamp propagates NaN
}
__device__ __forceinline__ void encode_limit(float xs, float ys, float xe, float ye, float pl, const float* lo,
                                             const float* hi, float* out) {
    const float dx = xe - xs, dy = ye - ys;
    float v[4] = {xs, ys, atan2f(ys - ye, xs - xe), sqrtf(dx * dx + dy * dy) / pl};
    limit4(v, lo, hi);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = v[k];
}
__device__ __forceinline__ void end_point(const float* a, float pl, float& xe, float& ye) {
    const float reach = a[3] * pl;                        // lengths * push_length, then * cos / sin   (:57-58, :87-88)
    xe = a[0] - reach * cosf(a[2]);
    ye = a[1] - reach * sinf(a[2]);
}

// ---- sampling -------------------------------------------------------------------------------------------------------
// mode 0 (iter_index == 0, :48-50): out = u * (hi - lo) + lo with u (S,H,4) uniform draws.
// mode 1 (:51-77): rnd (H,S,4) = the N(0, noise_level) draws of look-ahead step i in the order the reference draws
// them; scale[i] = fp32(0.1 * 10^i); start and end point of the nominal action move by scale*rnd; sample 0 keeps the
// nominal action untouched (:75).
struct SampleDev {
    const float* act_seq; const float* lo; const float* hi; const float* rnd; const float* scale;
    float* out; int S, H, mode; float pl;
};
__global__ void k_mppi_sample(SampleDev d) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)d.S * d.H) return;
    const int s = (int)(t / d.H), i = (int)(t - (long)s * d.H);
    float* o = d.out + t * 4;
    if (d.mode == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = d.rnd[t * 4 + k] * (d.hi[k] - d.lo[k]) + d.lo[k];
        return;
    }
    const float* a = d.act_seq + i * 4;
    if (s == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = a[k];
        return;
    }
    float xe, ye;
    end_point(a, d.pl, xe, ye);
    const float* n = d.rnd + ((long)i * d.S + s) * 4;
    const float sc = d.scale[i];
    encode_limit(a[0] + sc * n[0], a[1] + sc * n[1], xe + sc * n[2], ye + sc * n[3], d.pl, d.lo, d.hi, o);
}

// ---- update ---------------------------------------------------------------------------------------------------------
// weights = softmax(reward * reward_weight) over the B candidates (:83); per look-ahead step the weighted means of the
// start and end points (:90-93), re-encoded and limited (:95-101).
struct UpdateDev {
    const float* acts; const float* reward; const float* lo; const float* hi; float* out;
    int B, H; float rw, pl;
};
__device__ float tree_max(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = MT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}
__device__ float tree_sum(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = MT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(MT) void k_mppi_update(UpdateDev d) {
    __shared__ float red[MT];
    const int h = blockIdx.x, tid = threadIdx.x;
    float m = -3.4e38f;
    for (int b = tid; b < d.B; b += MT) m = fmaxf(m, d.reward[b] * d.rw);
    m = tree_max(m, red);
    float z = 0.0f;
    for (int b = tid; b < d.B; b += MT) z += expf(d.reward[b] * d.rw - m);
    z = tree_sum(z, red);
    float sx = 0.f, sy = 0.f, sxe = 0.f, sye = 0.f;
    for (int b = tid; b < d.B; b += MT) {
        const float w = expf(d.reward[b] * d.rw - m) / z;
        const float* a = d.acts + ((long)b * d.H + h) * 4;
        float xe, ye;
        end_point(a, d.pl, xe, ye);
        sx += w * a[0]; sy += w * a[1]; sxe += w * xe; sye += w * ye;
    }
    sx = tree_sum(sx, red); sy = tree_sum(sy, red); sxe = tree_sum(sxe, red); sye = tree_sum(sye, red);
    if (tid == 0) encode_limit(sx, sy, sxe, sye, d.pl, d.lo, d.hi, d.out + h * 4);
}

// ---- limits alone (clip_actions, :35-39) on n actions ---------------------------------------------------------------
__global__ void k_mppi_clip(const float* in, const float* lo, const float* hi, float* out, long n) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float v[4] = {in[t * 4], in[t * 4 + 1], in[t * 4 + 2], in[t * 4 + 3]};
    limit4(v, lo, hi);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[t * 4 + k] = v[k];
}

hipError_t launch_mppi_sample(const float* act_seq, const float* lo, const float* hi, const float* rnd,
                              const float* scale, int S, int H, int mode, float pl, float* out, hipStream_t st) {
    SampleDev d{act_seq, lo, hi, rnd, scale, out, S, H, mode, pl};
    const long n = (long)S * H;
    hipLaunchKernelGGL(k_mppi_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d);
    return hipGetLastError();
}
hipError_t launch_mppi_update(const float* acts, const float* reward, const float* lo, const float* hi, int B, int H,
                              float rw, float pl, float* out, hipStream_t st) {
    UpdateDev d{acts, reward, lo, hi, out, B, H, rw, pl};
    hipLaunchKernelGGL(k_mppi_update, dim3(H), dim3(MT), 0, st, d);
    return hipGetLastError();
}
hipError_t launch_mppi_clip(const float* in, const float* lo, const float* hi, float* out, long n, hipStream_t st) {
    hipLaunchKernelGGL(k_mppi_clip, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, lo, hi, out, n);
    return hipGetLastError();
}

}  // namespace ag
