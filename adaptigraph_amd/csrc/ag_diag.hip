// Diagnostic build only (-DAG_DIAG -> libadaptigraph_hip_diag.so, used by tools/ and one test): in-kernel clock / phase
// probes of the MLP chains and the injected-failure hook of the rollout's chunk loop.  The product library
// (libadaptigraph_hip.so) is compiled without AG_DIAG and contains none of this: no probe state, no stamps in the kernels,
// no environment reads in the launchers.  All state lives in a Diag object owned by ONE context (created by ag_ctx_create,
// reported and freed by ag_ctx_destroy); nothing is process-static.
//   AG_CLOCK_PROBE=n  stamp the first n k_edge_enc launches (synchronises!) and print the in-kernel clock
//   AG_NODE_PROBE=n   stamp the phases of the first n k_node_prop<false> launches (synchronises!); n < 0: stamp every launch
//                     without synchronising and report the clock of the last one when the context is destroyed
//   AG_TEST_FAIL_AT_CHUNK=n   ag_rollout_async fails before enqueuing chunk n, as a failed launch would
//   AG_TIMING_SKIP=mask       TIMING ONLY, WRONG RESULTS: the rollout's step loop leaves out the edge builder (bit 0; the graph of
//                             a look-ahead step's first forward is kept) and / or k_roll_update (bit 1; only the capturing step
//                             runs it) - the upper bound of what fusing those launches into a neighbour could save (r05)
#ifdef AG_DIAG
#include "ag_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace ag {

struct Diag {
    int clock_left = 0, node_left = 0, fail_at_chunk = -1, timing_skip = 0;
    bool node_tail = false;
    unsigned long long* dbg_e = nullptr; unsigned cap_e = 0;
    unsigned long long* dbg_n = nullptr; unsigned cap_n = 0; unsigned tail_nwg = 0;
};

void* diag_create() {
    Diag* d = new Diag();
    if (const char* e = getenv("AG_CLOCK_PROBE")) d->clock_left = atoi(e);
    if (const char* e = getenv("AG_NODE_PROBE")) { const int v = atoi(e); d->node_left = v > 0 ? v : 0; d->node_tail = v < 0; }
    if (const char* e = getenv("AG_TEST_FAIL_AT_CHUNK")) d->fail_at_chunk = atoi(e);
    if (const char* e = getenv("AG_TIMING_SKIP")) d->timing_skip = atoi(e);
    return d;
}
int diag_fail_at_chunk(void* v) { return v ? static_cast<Diag*>(v)->fail_at_chunk : -1; }
int diag_timing_skip(void* v) { return v ? static_cast<Diag*>(v)->timing_skip : 0; }

static double node_clock(const std::vector<unsigned long long>& h, unsigned nwg, int* n_out) {
    double ghz = 0; int n = 0;
    for (unsigned i = 0; i < nwg; ++i) {
        const unsigned long long c0 = h[(size_t)nwg * 5 + 2 * i], c1 = h[(size_t)nwg * 5 + 2 * i + 1];
        if (c1 > c0 && h[4 * i + 3] > h[4 * i]) { ghz += (double)(c1 - c0) / (double)(h[4 * i + 3] - h[4 * i]) * 0.1; ++n; }
    }
    *n_out = n;
    return n ? ghz / n : 0.0;
}

void diag_destroy(void* v) {
    Diag* d = static_cast<Diag*>(v);
    if (!d) return;
    if (d->node_tail && d->dbg_n && d->tail_nwg) {
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)d->tail_nwg * 7);
        (void)hipMemcpy(h.data(), d->dbg_n, (size_t)d->tail_nwg * 56, hipMemcpyDeviceToHost);
        int n = 0;
        const double ghz = node_clock(h, d->tail_nwg, &n);
        if (n) fprintf(stderr, "[ag node probe] last k_node_prop<false> launch of the context: in-kernel clock %.3f GHz over %d workgroups\n", ghz, n);
    }
    if (d->dbg_e) (void)hipFree(d->dbg_e);
    if (d->dbg_n) (void)hipFree(d->dbg_n);
    delete d;
}

unsigned long long* diag_edge_begin(void* v, unsigned nwg, hipStream_t st) {
    Diag* d = static_cast<Diag*>(v);
    if (!d || d->clock_left <= 0) return nullptr;
    if (d->cap_e < nwg) { if (d->dbg_e) (void)hipFree(d->dbg_e); (void)hipMalloc((void**)&d->dbg_e, (size_t)nwg * 32); d->cap_e = nwg; }
    (void)hipMemsetAsync(d->dbg_e, 0, (size_t)nwg * 32, st);
    return d->dbg_e;
}
void diag_edge_end(void* v, unsigned nwg, hipStream_t st) {
    Diag* d = static_cast<Diag*>(v);
    --d->clock_left;
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)nwg * 4);
    (void)hipMemcpy(h.data(), d->dbg_e, (size_t)nwg * 32, hipMemcpyDeviceToHost);
    double sum = 0, sumc = 0; int n = 0; double mn = 1e9, mx = 0;
    for (unsigned i = 0; i < nwg; ++i) {
        if (!h[4 * i + 3] || h[4 * i + 3] == h[4 * i + 1]) continue;
        const double cyc = (double)(h[4 * i + 2] - h[4 * i + 0]), rt = (double)(h[4 * i + 3] - h[4 * i + 1]);
        const double ghz = cyc / rt * 0.1;   // s_memrealtime ticks at 100 MHz
        sum += ghz; sumc += cyc; ++n; mn = ghz < mn ? ghz : mn; mx = ghz > mx ? ghz : mx;
    }
    if (n) fprintf(stderr, "[ag clock probe] k_edge_enc: %d workgroups, in-kernel clock mean %.3f GHz (min %.3f max %.3f), mean WG lifetime %.0f cycles\n", n, sum / n, mn, mx, sumc / n);
}

unsigned long long* diag_node_begin(void* v, unsigned nwg, hipStream_t st) {
    Diag* d = static_cast<Diag*>(v);
    if (!d || (!d->node_tail && d->node_left <= 0)) return nullptr;
    if (d->cap_n < nwg) {
        if (d->dbg_n) (void)hipFree(d->dbg_n);
        (void)hipMalloc((void**)&d->dbg_n, (size_t)nwg * 56); d->cap_n = nwg;
        (void)hipMemset(d->dbg_n, 0, (size_t)nwg * 56);
    }
    if (d->node_tail) d->tail_nwg = nwg;
    else (void)hipMemsetAsync(d->dbg_n, 0, (size_t)nwg * 56, st);
    return d->dbg_n;
}
void diag_node_end(void* v, unsigned nwg, int round, hipStream_t st) {
    Diag* d = static_cast<Diag*>(v);
    if (d->node_tail) return;                                // stamped without synchronising; reported at destroy
    --d->node_left;
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)nwg * 7);
    (void)hipMemcpy(h.data(), d->dbg_n, (size_t)nwg * 56, hipMemcpyDeviceToHost);
    int nghz = 0;
    const double ghz = node_clock(h, nwg, &nghz);
    unsigned long long t0 = ~0ull, t1 = 0;
    double a = 0, b = 0, c2 = 0; int n = 0;
    for (unsigned i = 0; i < nwg; ++i) {
        if (!h[4 * i + 3]) continue;
        t0 = std::min(t0, h[4 * i]); t1 = std::max(t1, h[4 * i + 3]);
        a += (double)(h[4 * i + 1] - h[4 * i]); b += (double)(h[4 * i + 2] - h[4 * i + 1]); c2 += (double)(h[4 * i + 3] - h[4 * i + 2]); ++n;
    }
    // per CU: how much of the launch had 0 / 1 / 2 workgroups in their chain phase (stamps 1..3), and in their gather
    std::vector<std::vector<unsigned>> by_cu(2048);
    for (unsigned i = 0; i < nwg; ++i) if (h[4 * i + 3]) by_cu[h[(size_t)nwg * 4 + i] & 0x7ff].push_back(i);
    double chain[3] = {0, 0, 0}, gath[3] = {0, 0, 0}; int ncu = 0;
    for (auto& vv : by_cu) {
        if (vv.empty()) continue;
        ++ncu;
        std::vector<std::pair<unsigned long long, int>> ev, eg;
        for (unsigned i : vv) {
            ev.push_back({h[4 * i + 1], +1}); ev.push_back({h[4 * i + 3], -1});
            eg.push_back({h[4 * i + 0], +1}); eg.push_back({h[4 * i + 1], -1});
        }
        auto sweep = [&](std::vector<std::pair<unsigned long long, int>>& e, double* acc) {
            std::sort(e.begin(), e.end());
            int lvl = 0; unsigned long long prev = t0;
            for (auto& x : e) { acc[std::min(lvl, 2)] += (double)(x.first - prev); prev = x.first; lvl += x.second; }
            acc[0] += (double)(t1 - prev);
        };
        sweep(ev, chain); sweep(eg, gath);
    }
    const double span = (double)(t1 - t0) * ncu;
    if (n) fprintf(stderr, "[ag node probe] round %d: %d workgroups on %d CUs, span %.1f us; mean per workgroup: gather+loads %.1f us, Wb layer %.1f us, "
                           "W2+W3 layers+stores %.1f us, in-kernel clock %.3f GHz | CU time with 0/1/2 workgroups in chain: %.0f%% %.0f%% %.0f%%; in gather: %.0f%% %.0f%% %.0f%%\n",
                   round, n, ncu, (t1 - t0) * 0.01, a / n * 0.01, b / n * 0.01, c2 / n * 0.01, ghz,
                   100 * chain[0] / span, 100 * chain[1] / span, 100 * chain[2] / span, 100 * gath[0] / span, 100 * gath[1] / span, 100 * gath[2] / span);
}

}  // namespace ag
#endif  // AG_DIAG
