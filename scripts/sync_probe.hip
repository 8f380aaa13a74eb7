// Probe (round 6): what does the host's wait for a stream cost?  An empty kernel, then (a) hipStreamSynchronize, (b) a spin on
// hipStreamQuery, (c) a spin on a pinned word the kernel writes: round trips per iteration, and the same behind a 200 us kernel
// (the host arrives long before the GPU is done: the case of a generation's end).
//   hipcc --offload-arch=gfx950 -O2 scripts/sync_probe.hip -o scripts/sync_probe && ./scripts/sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(volatile int* flag, int v) { if (flag && threadIdx.x == 0) { __threadfence_system(); *flag = v; } }
__global__ void k_busy(long long cycles) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int* pin; hipHostMalloc((void**)&pin, 64, hipHostMallocDefault);
    for (int busy = 0; busy < 2; busy++) {
        const long long cyc = busy ? 20000 : 0;          // wall_clock64 ticks at 100 MHz: 200 us
        for (int mode = 0; mode < 3; mode++) {
            double tot = 0; const int R = 200;
            for (int it = 0; it < R + 20; it++) {
                *pin = 0;
                const double t0 = now();
                if (cyc) hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s, cyc);
                hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (volatile int*)pin, it + 1);
                if (mode == 0) hipStreamSynchronize(s);
                else if (mode == 1) { while (hipStreamQuery(s) == hipErrorNotReady) {} }
                else { while (*(volatile int*)pin != it + 1) {} }
                if (it >= 20) tot += now() - t0;
            }
            printf("%s  %-28s %.2f us per round trip\n", busy ? "behind a 200 us kernel:" : "empty stream:          ",
                   mode == 0 ? "hipStreamSynchronize" : (mode == 1 ? "spin on hipStreamQuery" : "spin on a pinned word"), tot / R);
        }
    }
    return 0;
}
