// Diagnostic: what the host pays per step of K short dependent launches, issued (a) one hipLaunchKernel per launch from a
// C loop and (b) as one hipGraphLaunch of the recorded chain -- steps issued back to back without waiting, so the rate is
// whichever of host and GPU is slower.  Kernels: `small` (256 workgroups, a load, an add, a store: ~2 us on the GPU).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/host_rate_probe.hip -o tools/probes/host_rate_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void small(const float* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0f;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const int n = 256 * 256;
    float *a, *b;
    hipMalloc(&a, n * sizeof(float)), hipMalloc(&b, n * sizeof(float));
    hipMemset(a, 0, n * sizeof(float));
    hipStream_t s;
    hipStreamCreate(&s);
    const int steps = 2000;
    for (int K : {2, 3, 4, 7}) {
        auto issue = [&]() {
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(small, dim3(256), dim3(256), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, n);
        };
        for (int w = 0; w < 50; ++w) issue();
        hipStreamSynchronize(s);
        double t0 = now_us();
        for (int i = 0; i < steps; ++i) issue();
        const double host_direct = (now_us() - t0) / steps;
        hipStreamSynchronize(s);
        const double all_direct = (now_us() - t0) / steps;

        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        issue();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int w = 0; w < 50; ++w) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        t0 = now_us();
        for (int i = 0; i < steps; ++i) hipGraphLaunch(ge, s);
        const double host_graph = (now_us() - t0) / steps;
        hipStreamSynchronize(s);
        const double all_graph = (now_us() - t0) / steps;
        printf("%d launches per step: direct %.2f us per step (host alone %.2f), graph %.2f (host alone %.2f)\n", K, all_direct, host_direct,
               all_graph, host_graph);
        hipGraphExecDestroy(ge), hipGraphDestroy(g);
    }
    return 0;
}
