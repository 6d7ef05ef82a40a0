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

}  // namespace covgram
