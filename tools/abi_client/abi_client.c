/* A plain-C99 host of the C-ABI (include/adaptigraph_amd.h): no Python, no C++, no torch - what a maintainer binding the
 * engine from another language would write.  Reads a small binary case file, runs one rollout through ag_rollout (host-decoded
 * actions) and - if the case carries raw actions - through ag_rollout_actions (device-planned), writes the outputs.
 *
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tools/abi_client/abi_client.c \
 *       -L adaptigraph_amd/csrc -ladaptigraph_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,... -o abi_client
 *   ./abi_client case.bin out.bin            (tests/test_gpu_abi_client.py writes case.bin and checks out.bin)
 *
 * case.bin: int32 header {magic 0x41474331, B, H, N_o, M, topk, connect_tools_all, max_nR, gripper_enable, pstep, n_his,
 * max_repeat}, float32 {adj_thresh, gripper_offset, physics_param, push_length, tool_off[8]}, then float32 arrays: the 22
 * state_dict tensors (header order of ag_ctx_load_weights), state0 (N_o,3), eef_xz (B,H,M,2), eef_delta (B,H,M,3), actions
 * (B,H,4), and int32 repeat (B,H).
 * out.bin: float32 state_seqs of ag_rollout (B,H,N_o,3), state_seqs of ag_rollout_actions, its action_seqs (B,H,4); int64
 * executed, needed candidate-forwards of the last call. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "adaptigraph_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_AG(x) do { int rc_ = (x); if (rc_ != AG_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, ag_last_error(ctx)); return 3; } } while (0)

static void* read_n(FILE* f, size_t n, size_t sz) {
    void* p = malloc(n * sz > 0 ? n * sz : 1);
    if (!p || fread(p, sz, n, f) != n) { fprintf(stderr, "short read\n"); exit(4); }
    return p;
}
static void* to_dev(const void* h, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess || hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) exit(5);
    return d;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s case.bin out.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t* hd = (int32_t*)read_n(f, 12, 4);
    float* fl = (float*)read_n(f, 12, 4);
    if (hd[0] != 0x41474331) { fprintf(stderr, "bad magic\n"); return 1; }
    const int B = hd[1], H = hd[2], N_o = hd[3], M = hd[4], pstep = hd[9], n_his = hd[10], max_repeat = hd[11];
    const int rel_dim = 5 + 3 * n_his;
    /* the 22 tensors of DynamicsPredictor.state_dict(), in the order the header lists */
    const size_t wshape[22] = {150 * 6, 150, 150 * 150, 150, 150 * 150, 150, (size_t)150 * rel_dim, 150, 150 * 150, 150, 150 * 150, 150,
                               150 * 300, 150, 150 * 450, 150, 150 * 150, 150, 150 * 150, 150, 3 * 150, 3};
    const float* w[22];
    for (int i = 0; i < 22; ++i) w[i] = (const float*)read_n(f, wshape[i], 4);
    float* state0 = (float*)read_n(f, (size_t)N_o * 3, 4);
    float* eef_xz = (float*)read_n(f, (size_t)B * H * M * 2, 4);
    float* eef_delta = (float*)read_n(f, (size_t)B * H * M * 3, 4);
    float* actions = (float*)read_n(f, (size_t)B * H * 4, 4);
    int32_t* repeat = (int32_t*)read_n(f, (size_t)B * H, 4);
    fclose(f);

    if (ag_abi_version() != AG_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    ag_dims dims = {150, n_his, pstep, 6, rel_dim, 100.0f};
    ag_ctx* ctx = NULL;
    CHECK_AG(ag_ctx_create(0, &dims, &ctx));
    CHECK_AG(ag_ctx_load_weights(ctx, w, AG_NUM_WEIGHT_TENSORS));

    ag_rollout_params p;
    memset(&p, 0, sizeof p);
    p.B = B; p.H = H; p.N_o = N_o; p.M = M; p.topk = hd[5]; p.connect_tools_all = hd[6]; p.max_nR = hd[7]; p.y_mode = 0;
    p.adj_thresh = fl[0]; p.gripper_offset = fl[1]; p.gripper_enable = hd[8]; p.physics_param = fl[2];

    const size_t out_n = (size_t)B * H * N_o * 3;
    float *d_out1 = NULL, *d_out2 = NULL, *d_dec = NULL;
    int32_t* d_flags = NULL;
    CHECK_HIP(hipMalloc((void**)&d_out1, out_n * 4));
    CHECK_HIP(hipMalloc((void**)&d_out2, out_n * 4));
    CHECK_HIP(hipMalloc((void**)&d_dec, (size_t)B * H * 4 * 4));
    CHECK_HIP(hipMalloc((void**)&d_flags, 16));
    CHECK_HIP(hipMemset(d_flags, 0, 16));
    float* d_state0 = (float*)to_dev(state0, (size_t)N_o * 3 * 4);
    float* d_xz = (float*)to_dev(eef_xz, (size_t)B * H * M * 2 * 4);
    float* d_delta = (float*)to_dev(eef_delta, (size_t)B * H * M * 3 * 4);
    float* d_act = (float*)to_dev(actions, (size_t)B * H * 4 * 4);

    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    /* 1. host-decoded actions: the caller did decode_action + the tool layout (forward_dynamics.py:23,42-75) */
    CHECK_AG(ag_rollout(ctx, st, &p, d_state0, NULL, d_xz, d_delta, repeat, NULL, d_out1));
    /* 2. raw actions resident on the device: decode + launch plan on the GPU */
    CHECK_AG(ag_ctx_set_option(ctx, "streams", 2));
    CHECK_AG(ag_rollout_actions(ctx, st, &p, d_state0, d_act, fl[3], fl + 4, max_repeat, NULL, d_out2, d_dec, d_flags));
    CHECK_HIP(hipStreamSynchronize(st));
    int32_t flags[2];
    CHECK_HIP(hipMemcpy(flags, d_flags, 8, hipMemcpyDeviceToHost));
    if (flags[0] > p.max_nR || flags[1] > max_repeat) { fprintf(stderr, "overflow flags %d %d\n", flags[0], flags[1]); return 6; }
    int64_t counts[2];
    CHECK_AG(ag_ctx_rollout_counts(ctx, &counts[0], &counts[1]));

    float* h = (float*)malloc(out_n * 4);
    FILE* o = fopen(argv[2], "wb");
    if (!o || !h) return 1;
    CHECK_HIP(hipMemcpy(h, d_out1, out_n * 4, hipMemcpyDeviceToHost));
    fwrite(h, 4, out_n, o);
    CHECK_HIP(hipMemcpy(h, d_out2, out_n * 4, hipMemcpyDeviceToHost));
    fwrite(h, 4, out_n, o);
    CHECK_HIP(hipMemcpy(h, d_dec, (size_t)B * H * 4 * 4, hipMemcpyDeviceToHost));
    fwrite(h, 4, (size_t)B * H * 4, o);
    fwrite(counts, 8, 2, o);
    fclose(o);
    CHECK_AG(ag_ctx_destroy(ctx));
    printf("abi_client ok: B=%d H=%d N_o=%d, candidate-forwards executed %lld needed %lld\n", B, H, N_o, (long long)counts[0], (long long)counts[1]);
    return 0;
}
