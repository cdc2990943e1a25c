// Stand-in for the HIP runtime header, for running the builder's kernels thread by thread on the host
// (tools/reinsert_host_check.cpp): the kernel qualifiers vanish, the thread coordinates are plain globals that the
// harness sets, atomics are plain read-modify-writes (one thread runs at a time).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
struct ShimDim { unsigned int x, y, z; };
inline ShimDim threadIdx{0, 0, 0}, blockIdx{0, 0, 0}, blockDim{1, 1, 1}, gridDim{1, 1, 1};
inline unsigned int __float_as_uint(float f) { unsigned int u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned int u) { float f; std::memcpy(&f, &u, 4); return f; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
inline void __threadfence() {}
template <class T> inline T __shfl_down(T v, int, int) { return v; }
inline int __clz(unsigned int v) { return v ? __builtin_clz(v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
#define __builtin_nontemporal_load(p) (*(p))
