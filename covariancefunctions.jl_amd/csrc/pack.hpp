// pack.hpp — the column-packing abstraction shared by the kernels: fp32 carries two columns per stream element
// (v_pk_add/mul/fma_f32), fp64 one.
#pragma once
#include <hip/hip_runtime.h>

namespace covgram {

typedef float v2f __attribute__((ext_vector_type(2)));

// columns per stream element: 2 for float (packed math), 1 for double
template <typename T> struct Pk;
template <> struct Pk<float> {
    using V = v2f;
    static constexpr int N = 2;
    static __device__ __forceinline__ V splat(float x) { return (V){x, x}; }
    static __device__ __forceinline__ V fma(V a, V b, V c) { return __builtin_elementwise_fma(a, b, c); }
    static __device__ __forceinline__ float hsum(V a) { return a.x + a.y; }
    template <class F> static __device__ __forceinline__ V map(V s, F f) { return (V){f(s.x), f(s.y)}; }
    // (phi, phi', phi'') per component: f(s, v, d1, d2)
    template <class F> static __device__ __forceinline__ void map3(V s, F f, V& v, V& d1, V& d2) {
        float a0, a1, a2, b0, b1, b2;
        f(s.x, a0, a1, a2); f(s.y, b0, b1, b2);
        v = (V){a0, b0}; d1 = (V){a1, b1}; d2 = (V){a2, b2};
    }
};
template <> struct Pk<double> {
    using V = double;
    static constexpr int N = 1;
    static __device__ __forceinline__ V splat(double x) { return x; }
    static __device__ __forceinline__ V fma(V a, V b, V c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ double hsum(V a) { return a; }
    template <class F> static __device__ __forceinline__ V map(V s, F f) { return f(s); }
    template <class F> static __device__ __forceinline__ void map3(V s, F f, V& v, V& d1, V& d2) { f(s, v, d1, d2); }
};

// ---- split-J partials summed by the LAST workgroup to arrive (round 4) ------------------------------------------------------
// The dense kernels leave their column-split partials in a slab [split][...][npad]; a second launch used to add them in fixed order.
// Instead every workgroup of a row block takes a ticket after its slab stores; the one that draws the last ticket reads all partials of
// the block — its own included — and sums them in the SAME fixed order the reduce kernel used, so the result is bit-identical and
// deterministic whatever the arrival order; it also puts the ticket back to 0 for the next launch (kernel boundaries order that).
// No float atomics.  One launch and one dependent-launch gap less per MVM.
// Coherence across the eight XCDs (each has its own L2): NOT by __threadfence() — at device scope that is a write-back plus an
// invalidate of the whole L2 per workgroup, measured at +200 us on a 230 us row shard and 29 -> 90 us on C1
// (profiles/r04_inkernel_reduce_ab.txt, first build).  The slab traffic of a ticketed launch goes AROUND the non-coherent lines
// instead: partials are stored and re-read with device-scope relaxed atomic accesses (global_store / global_load ... sc1: written
// through to, and read from, the memory side), the stores are waited for (s_waitcnt vmcnt(0)) before the ticket is taken, and the
// ticket is a device-scope atomic add.
template <typename T>
__device__ __forceinline__ void slab_store(T* p, T v, bool ticketed) {
    if (ticketed) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <typename T>
__device__ __forceinline__ T slab_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool last_arrival(unsigned* __restrict__ ticket, unsigned expected) {
    __shared__ unsigned flag;
    // EVERY wave drains its own write-through (sc1) slab stores before the barrier the ticket is taken behind.  Written as inline asm:
    // a workgroup-scope release fence lowers to an lgkmcnt wait only on gfx950 (no vmcnt), and the compiler may drop a builtin wait it
    // believes redundant (MI355X guide, "Compiler hazard").  tools/check_isa.py asserts the wait sits in front of the barrier.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = (old == expected - 1u) ? 1u : 0u;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag = last;
    }
    __syncthreads();
    return flag != 0u;
}
// sum_s partial[s * stride] over s < count in dense_reduce_kernel's order: four chains s = q (mod 4), then (c0 + c1) + (c2 + c3)
template <typename T>
__device__ __forceinline__ T ordered_split_sum(const T* __restrict__ partial, int64_t stride, int count) {
    T c[4] = {(T)0, (T)0, (T)0, (T)0};
    int s = 0;
    for (; s + 4 <= count; s += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] += slab_load(partial + (int64_t)(s + q) * stride);
    }
    for (int q = 0; s + q < count; ++q) c[q] += slab_load(partial + (int64_t)(s + q) * stride);
    return (c[0] + c[1]) + (c[2] + c[3]);
}

}  // namespace covgram
