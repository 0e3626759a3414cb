// Shader clock actually delivered during short kernels: s_memtime ticks per 10 ns tick of s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long* out, int iters) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = w1 - w0, out[1] = c1 - c0, out[2] = (unsigned long long)x;
}
int main() {
    unsigned long long* d; unsigned long long h[3];
    hipMalloc(&d, 24);
    for (int rep = 0; rep < 6; ++rep) {
        const int iters = rep < 3 ? 2000 : 200000;
        spin<<<1024, 256>>>(d, iters);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("iters %d: %llu wall ticks (10 ns), %llu clock ticks -> %.1f MHz counter; %.2f ns per dependent FMA\n", iters, h[0], h[1],
               h[1] / (h[0] * 0.01), h[0] * 10.0 / iters);
    }
    return 0;
}
