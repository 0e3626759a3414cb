// Diagnostic: shader clock against the 100 MHz wall clock inside a kernel that keeps every SIMD busy with vector work.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/clock_probe.hip -o gpurun_out/clock_probe && gpurun_out/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void spin(unsigned long long* out, int iters, float seed) {
    float a = seed + threadIdx.x, b = 1.0001f;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) a = a * b + 0.5f;
    }
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = w1 - w0, out[3 * blockIdx.x + 1] = c1 - c0;
        out[3 * blockIdx.x + 2] = static_cast<unsigned long long>(a);
    }
}

int main() {
    const int max_blocks = 256 * 4 * 16;
    unsigned long long* d;
    hipMalloc(&d, max_blocks * 3 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(max_blocks * 3);
    const int iters = 2000;
    // one wavefront per workgroup, k wavefronts per SIMD if the dispatcher spreads them evenly
    for (int k : {1, 2, 3, 4, 6, 8, 12}) {
        const int blocks = 256 * 4 * k;
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        float ms = 0.0f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0f);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(h.data(), d, blocks * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double w = 0, c = 0;
        for (int b = 0; b < blocks; ++b) w += h[3 * b], c += h[3 * b + 1];
        const double instrs = 128.0 * iters;   // dependent v_mul + v_add pairs
        printf("%2d wavefronts per SIMD: kernel %.1f us, per wavefront %.1f us (rounds %.2f), %.2f shader cycles and %.2f ns per "
               "instruction per wavefront, %.3f ns per instruction per SIMD\n", k, ms * 1e3, w / blocks / 100.0, ms * 1e3 / (w / blocks / 100.0),
               c / blocks / instrs, w / blocks * 10.0 / instrs, w / blocks * 10.0 / instrs / k);
    }
    return 0;
}
