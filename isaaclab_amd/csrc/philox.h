// Counter-based random numbers shared by the kernels that draw on the device (observation noise, event terms).
#pragma once

#include "math_dev.h"

namespace ilx {

// Philox4x32-10 (counter based): reproducible per (seed, step, env, element block) whatever the launch geometry.
struct u4 {
    uint32_t x, y, z, w;
};
ILX_DEV u4 philox4(uint64_t seed, uint64_t step, uint64_t env, uint32_t elem) {
    uint32_t c0 = elem, c1 = static_cast<uint32_t>(env), c2 = static_cast<uint32_t>(env >> 32) ^ static_cast<uint32_t>(step),
             c3 = static_cast<uint32_t>(step >> 32);
    uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32); integer multiplies issue at a quarter of the plain rate,
        // so separate high / low multiplies would double the cost of a call
        const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0, p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
        const uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
        const uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
        // three-way exclusive or in one instruction (v_bitop3_b32, truth table 0x96)
        c0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
        c1 = lo1;
        c2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u4{c0, c1, c2, c3};
}
ILX_DEV float u01(uint32_t bits) { return static_cast<float>(bits >> 8) * (1.0f / 16777216.0f); }  // [0, 1), 24 bits
ILX_DEV float philox_uniform(uint64_t seed, uint64_t step, uint64_t env, uint32_t elem) {
    return u01(philox4(seed, step, env, elem).x);
}

}  // namespace ilx
