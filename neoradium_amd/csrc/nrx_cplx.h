// Minimal complex arithmetic for device code (interleaved re,im pairs, float or double).
#pragma once
#include <hip/hip_runtime.h>

namespace nrx {

template <typename T> struct alignas(2 * sizeof(T)) cx {
  T re, im;
  __host__ __device__ cx() = default;
  __host__ __device__ constexpr cx(T r, T i) : re(r), im(i) {}
  template <typename U> __host__ __device__ explicit cx(const cx<U>& o) : re((T)o.re), im((T)o.im) {}
};

template <typename T> __device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) { return {a.re + b.re, a.im + b.im}; }
template <typename T> __device__ __forceinline__ cx<T> operator-(cx<T> a, cx<T> b) { return {a.re - b.re, a.im - b.im}; }
template <typename T> __device__ __forceinline__ cx<T> operator*(cx<T> a, cx<T> b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <typename T> __device__ __forceinline__ cx<T> operator*(cx<T> a, T s) { return {a.re * s, a.im * s}; }
template <typename T> __device__ __forceinline__ cx<T> conj(cx<T> a) { return {a.re, -a.im}; }
template <typename T> __device__ __forceinline__ T norm2(cx<T> a) { return a.re * a.re + a.im * a.im; }
// a += b*c
template <typename T> __device__ __forceinline__ void cmac(cx<T>& a, cx<T> b, cx<T> c) {
  a.re += b.re * c.re - b.im * c.im;
  a.im += b.re * c.im + b.im * c.re;
}
// a += conj(b)*c
template <typename T> __device__ __forceinline__ void cmacc(cx<T>& a, cx<T> b, cx<T> c) {
  a.re += b.re * c.re + b.im * c.im;
  a.im += b.re * c.im - b.im * c.re;
}
// a / b (Smith-free: |b|^2 denominator; operands here are O(1) pilots / pivots)
template <typename T> __device__ __forceinline__ cx<T> cdiv(cx<T> a, cx<T> b) {
  const T d = (T)1 / norm2(b);
  return {(a.re * b.re + a.im * b.im) * d, (a.im * b.re - a.re * b.im) * d};
}

}  // namespace nrx
