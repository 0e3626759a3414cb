// Diagnostic: what a kernel pays before its first vector load can leave.  A chain of dependent launches of a small kernel
// (every thread loads one float through a pointer argument, adds, stores); per-launch time of the chain with the arguments
// read the usual way and -- built with -mllvm -amdgpu-kernarg-preload-count=16 -- handed over in registers at wavefront
// start.  Also: in-kernel clock stamps of the argument read and of a dependent scalar read.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/entry_probe.hip -o tools/probes/entry_probe.bin
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/probes/entry_probe.hip -o tools/probes/entry_probe_preload.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct Big {
    const float* in;
    float* out;
    unsigned long long* stamps;
    const unsigned long long* word;
    int n;
    int pad[240];
};

__global__ void by_pointers(const float* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0f;
}

__global__ void nothing(int n) {}

__global__ void by_struct(Big a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) a.out[i] = a.in[i] + 1.0f;
}

__global__ void stamped(Big a) {
    const unsigned long long t0 = wall_clock64();
    const int n = a.n;                      // argument read
    asm volatile("" ::"s"(n));
    const unsigned long long t1 = wall_clock64();
    const unsigned long long w = *reinterpret_cast<const __attribute__((address_space(4))) unsigned long long*>(
        (const __attribute__((address_space(4))) char*)a.word);   // dependent scalar read from device memory
    asm volatile("" ::"s"(w));
    const unsigned long long t2 = wall_clock64();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float v = i < n ? a.in[i] : 0.0f;
    asm volatile("" ::"v"(v));
    const unsigned long long t3 = wall_clock64();
    if (i < n) a.out[i] = v + 1.0f;
    if (threadIdx.x == 0) {
        a.stamps[4 * blockIdx.x] = t1 - t0, a.stamps[4 * blockIdx.x + 1] = t2 - t1, a.stamps[4 * blockIdx.x + 2] = t3 - t2;
        a.stamps[4 * blockIdx.x + 3] = t0;
    }
}

int main() {
    const int n = 256 * 256 * 4;
    float *a, *b;
    unsigned long long *stamps, *word;
    hipMalloc(&a, n * sizeof(float)), hipMalloc(&b, n * sizeof(float));
    hipMalloc(&stamps, 4096 * 4 * sizeof(unsigned long long)), hipMalloc(&word, 64);
    hipMemset(a, 0, n * sizeof(float)), hipMemset(word, 0, 64);
    hipStream_t s;
    hipStreamCreate(&s);
    Big big{};
    big.in = a, big.out = b, big.stamps = stamps, big.word = word, big.n = n;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int blocks : {256, 1024}) {
            hipGraph_t g;
            hipGraphExec_t ge;
            hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            for (int k = 0; k < 100; ++k) {
                float* src = (k & 1) ? b : a;
                float* dst = (k & 1) ? a : b;
                if (mode == 0) hipLaunchKernelGGL(by_pointers, dim3(blocks), dim3(256), 0, s, src, dst, blocks * 256);
                else {
                    big.in = src, big.out = dst, big.n = blocks * 256;
                    hipLaunchKernelGGL(by_struct, dim3(blocks), dim3(256), 0, s, big);
                }
            }
            hipStreamEndCapture(s, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            float best = 1e9f;
            for (int rep = 0; rep < 10; ++rep) {
                hipEventRecord(e0, s);
                hipGraphLaunch(ge, s);
                hipEventRecord(e1, s);
                hipStreamSynchronize(s);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms);
            }
            printf("%s, %4d workgroups: %.2f us per launch in a chain of 100\n", mode ? "2 KB struct argument" : "three scalar arguments", blocks,
                   best * 1e3 / 100);
            hipGraphExecDestroy(ge), hipGraphDestroy(g);
        }
    }
    // the dispatcher's rate: empty workgroups
    for (int threads : {64, 256, 320}) {
        for (int blocks : {256, 1024, 4096, 16384, 65536}) {
            hipGraph_t g;
            hipGraphExec_t ge;
            hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            for (int k = 0; k < 100; ++k) hipLaunchKernelGGL(nothing, dim3(blocks), dim3(threads), 0, s, k);
            hipStreamEndCapture(s, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0, s);
                hipGraphLaunch(ge, s);
                hipEventRecord(e1, s);
                hipStreamSynchronize(s);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms);
            }
            printf("empty kernel, %5d workgroups of %3d threads: %.2f us per launch\n", blocks, threads, best * 1e3 / 100);
            hipGraphExecDestroy(ge), hipGraphDestroy(g);
        }
    }
    // what a replay costs beyond its kernels: graphs of k empty kernels, 200 replays back to back
    for (int k : {1, 2, 4, 7, 14}) {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < k; ++i) hipLaunchKernelGGL(nothing, dim3(256), dim3(256), 0, s, i);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, s);
            for (int r = 0; r < 200; ++r) hipGraphLaunch(ge, s);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms);
        }
        printf("graph of %2d empty kernels: %.2f us per replay (200 replays back to back)\n", k, best * 1e3 / 200);
        hipGraphExecDestroy(ge), hipGraphDestroy(g);
    }
    big.in = a, big.out = b, big.n = 256 * 256;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stamped, dim3(256), dim3(256), 0, s, big);
        hipStreamSynchronize(s);
    }
    std::vector<unsigned long long> h(256 * 4);
    hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    for (int k = 0; k < 3; ++k) {
        std::vector<unsigned long long> v;
        for (int blk = 0; blk < 256; ++blk) v.push_back(h[4 * blk + k]);
        std::sort(v.begin(), v.end());
        const char* what[] = {"argument read", "dependent scalar read (device memory)", "vector load"};
        printf("%-40s min %.2f  median %.2f  max %.2f us\n", what[k], v.front() / 100.0, v[128] / 100.0, v.back() / 100.0);
    }
    unsigned long long first = ~0ull, last = 0;
    for (int blk = 0; blk < 256; ++blk) first = std::min(first, h[4 * blk + 3]), last = std::max(last, h[4 * blk + 3]);
    printf("workgroup entries spread over %.2f us\n", (last - first) / 100.0);
    return 0;
}
