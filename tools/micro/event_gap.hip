// What a hipEventRecord between two dependent kernels costs on the stream (the library brackets every solver launch with a pair when the
// caller asks for kernel times).  Chain of N short kernels, timed with the host clock around a stream sync:
//   none: no events; default: hipEventCreate; nofence: hipEventDisableSystemFence; notiming: hipEventDisableTiming
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/event_gap.hip -o tools/micro/event_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_spin(unsigned *out, int n) {
    unsigned v = threadIdx.x;
    for (int i = 0; i < n; ++i) v = v * 1664525u + 1013904223u;
    if (v == 12345u) out[0] = v;
}
static double run(hipStream_t st, unsigned *buf, int N, int spin, std::vector<hipEvent_t> *ev, int per) {
    for (int w = 0; w < 2; ++w) {
        hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            if (ev) for (int e = 0; e < per; ++e) hipEventRecord((*ev)[(size_t)(i * per + e)], st);
            hipLaunchKernelGGL(k_spin, dim3(1024), dim3(256), 0, st, buf, spin);
        }
        hipStreamSynchronize(st);
        if (w == 1) return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    }
    return 0;
}
int main() {
    hipStream_t st; hipStreamCreate(&st);
    unsigned *buf; hipMalloc(&buf, 4096);
    const int N = 400;
    for (int spin : {2000, 20000}) {
        const double base = run(st, buf, N, spin, nullptr, 0);
        printf("kernel of %d steps: %.2f us per launch without events\n", spin, base);
        const unsigned flags[] = {hipEventDefault, hipEventDisableSystemFence, hipEventDisableTiming, hipEventDisableTiming | hipEventDisableSystemFence};
        const char *names[] = {"default", "nofence", "notiming", "notiming+nofence"};
        for (int f = 0; f < 4; ++f)
            for (int per : {1, 2}) {
                std::vector<hipEvent_t> ev((size_t)N * per);
                for (auto &e : ev) if (hipEventCreateWithFlags(&e, flags[f]) != hipSuccess) { printf("create failed\n"); return 1; }
                const double t = run(st, buf, N, spin, &ev, per);
                float ms = 0; if (!(flags[f] & hipEventDisableTiming)) hipEventElapsedTime(&ms, ev[0], ev[(size_t)per]);
                printf("  %-18s %d per launch: %.2f us per launch (+%.2f); event-to-event %.2f us\n", names[f], per, t, t - base, ms * 1e3);
                for (auto &e : ev) hipEventDestroy(e);
            }
    }
    return 0;
}
