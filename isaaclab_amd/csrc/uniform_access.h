// Device helpers for values and addresses that are the same in every lane of a wavefront.
#pragma once

#include <cstdint>

#include "math_dev.h"

namespace ilx {

// A value that is the same in every lane, moved into scalar registers.
ILX_DEV int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
ILX_DEV uint32_t uni(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int32_t>(v))); }
ILX_DEV float uni(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int32_t, v))); }
ILX_DEV uint64_t uni(uint64_t v) {
    return (static_cast<uint64_t>(uni(static_cast<uint32_t>(v >> 32))) << 32) | uni(static_cast<uint32_t>(v));
}
ILX_DEV int64_t uni(int64_t v) { return static_cast<int64_t>(uni(static_cast<uint64_t>(v))); }
template <typename T>
ILX_DEV __attribute__((address_space(1))) T* uni(T* p) {  // ... and a pointer, as a global-memory one
    return (__attribute__((address_space(1))) T*)uni(reinterpret_cast<uint64_t>(p));
}
template <typename T>
ILX_DEV __attribute__((address_space(1))) T* as_global(T* p) {  // a generic pointer known to point to global memory
    return (__attribute__((address_space(1))) T*)p;
}

// Element access as a uniform 64-bit base (scalar registers; advance it to a block's first element with scalar
// arithmetic) plus a 32-bit per-lane byte offset: the address costs no vector instruction beyond forming the offset.
template <typename T>
ILX_DEV __attribute__((address_space(1))) char* scalar_base(T* base) {
    uint64_t a = reinterpret_cast<uint64_t>(base);
    asm volatile("" : "+s"(a));  // no instruction: pins the base to scalar registers, so that the compiler does not fold it into the lane's offset
    return (__attribute__((address_space(1))) char*)a;
}
template <typename T>
ILX_DEV T ld_at(const __attribute__((address_space(1))) T* base, uint32_t byte_off) {
    return *reinterpret_cast<const __attribute__((address_space(1))) T*>(scalar_base(base) + byte_off);
}
template <typename T, typename V>
ILX_DEV void st_at(__attribute__((address_space(1))) T* base, uint32_t byte_off, V v) {
    *reinterpret_cast<__attribute__((address_space(1))) T*>(scalar_base(base) + byte_off) = static_cast<T>(v);
}

// The same for a workgroup that owns a contiguous run of rows: the run's first element as a scalar base (computed once, with
// scalar arithmetic), and 32-bit byte offsets from it per access.
template <typename T>
ILX_DEV __attribute__((address_space(1))) char* block_base(T* p, int64_t first_elem) {
    uint64_t a = reinterpret_cast<uint64_t>(p) + static_cast<uint64_t>(sizeof(T)) * static_cast<uint64_t>(first_elem);
    asm volatile("" : "+s"(a));   // no instruction: pins the base to scalar registers
    return (__attribute__((address_space(1))) char*)a;
}
template <typename T>
ILX_DEV T ld_b(const __attribute__((address_space(1))) char* base, uint32_t byte_off) {
    return *reinterpret_cast<const __attribute__((address_space(1))) T*>(base + byte_off);
}
template <typename T, typename V>
ILX_DEV void st_b(__attribute__((address_space(1))) char* base, uint32_t byte_off, V v) {
    *reinterpret_cast<__attribute__((address_space(1))) T*>(base + byte_off) = static_cast<T>(v);
}

// A kernel whose arguments fill many cache lines reads them with scalar loads as its code gets to them; where the scalar
// registers do not hold all of them at once the compiler loads a few, waits, parks them in lanes of a vector register
// and loads the next few -- a chain of trips to memory, one behind the other, before the kernel's first vector load
// leaves.  Called first thing in a kernel, this asks for one word of every 64-byte line of the arguments (kBytes: their
// total size, the kernel's kernarg segment) at once and waits for them: one trip, after which the loads that follow are
// served by the scalar cache.  The words go to one register that nothing reads; the wait inside the block is what makes
// that register safe to hand back to the compiler.
// The size of a kernel's explicit arguments: each at the next multiple of its alignment.
template <class... Args>
constexpr int kernel_argument_bytes() {
    int at = 0;
    ((at = (at + static_cast<int>(alignof(Args)) - 1) / static_cast<int>(alignof(Args)) * static_cast<int>(alignof(Args)) + static_cast<int>(sizeof(Args))), ...);
    return (at + 3) / 4 * 4;
}
template <int kBytes>
ILX_DEV void request_kernel_arguments() {
    static_assert(kBytes >= 4 && kBytes % 4 == 0, "the arguments' size in bytes");
    uint32_t sink;
    asm volatile(
        ".set ilx_ka_off, 0\n"
        ".rept %2\n"
        "s_load_dword %0, %1, ilx_ka_off\n"
        ".set ilx_ka_off, ilx_ka_off + 64\n"
        ".endr\n"
        "s_load_dword %0, %1, %3\n"   // (the arguments need not start on a line's boundary)
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(sink)
        : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"((kBytes + 63) / 64), "n"(kBytes - 4)
        : "memory");
}

}  // namespace ilx
