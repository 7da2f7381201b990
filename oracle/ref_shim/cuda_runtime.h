/*
 * oracle/ref_shim/cuda_runtime.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A host stand-in for the handful of CUDA language / runtime names that the reference's two plain-C++ kernels use
 *   /root/reference/extensions/sgutils/sg.cu            evaluate_gaussian_{fwd,bwd}_kernel
 *   /root/reference/extensions/utils/utils_kernel.cu    compute_raydirs_forward_kernel
 * and that extensions/include/helper_math.h expects from <cuda_runtime.h> in its non-NVCC mode: the built-in
 * vector types with their make_* constructors, the __host__/__device__/__global__ qualifiers, the launch indices
 * (set by oracle/ref_shim/ref_host.cpp's serial grid loops), __expf and atomicAdd.  With it the reference's kernel
 * text compiles with g++ as ordinary C++ (those kernels contain no warp intrinsics), giving oracle/_ref/libref.so --
 * the reference's OWN arithmetic on the host.  Nothing of the reference is copied into the repository: the build
 * recipe (oracle/Makefile, target _ref) reads the kernels from /root/reference at build time.
 */
#pragma once
#include <cmath>
#include <cstdint>

#define __host__
#define __device__
#define __global__
#ifndef __forceinline__
#define __forceinline__ inline
#endif

#define ORC_VEC(T, N)                                                                  \
  struct N##1 { T x; };                                                                \
  struct N##2 { T x, y; };                                                             \
  struct N##3 { T x, y, z; };                                                          \
  struct N##4 { T x, y, z, w; };                                                       \
  constexpr inline N##1 make_##N##1(T x) { return N##1{x}; }                           \
  constexpr inline N##2 make_##N##2(T x, T y) { return N##2{x, y}; }                   \
  constexpr inline N##3 make_##N##3(T x, T y, T z) { return N##3{x, y, z}; }           \
  constexpr inline N##4 make_##N##4(T x, T y, T z, T w) { return N##4{x, y, z, w}; }

typedef unsigned int uint;
typedef unsigned char uchar;
typedef unsigned short ushort;
ORC_VEC(float, float)
ORC_VEC(double, double)
ORC_VEC(char, char)
ORC_VEC(unsigned char, uchar)
ORC_VEC(short, short)
ORC_VEC(unsigned short, ushort)
ORC_VEC(int, int)
ORC_VEC(unsigned int, uint)
#undef ORC_VEC

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
/* launch indices of the "thread" being executed: one host thread runs one CUDA block at a time */
extern thread_local dim3 blockIdx, blockDim, threadIdx, gridDim;

/* -use_fast_math intrinsics of the reference build (extensions/sgutils/setup.py:30): the host has no fast
 * variants, so the accurate ones stand in (the difference is below fp32 rounding of the result) */
#define __expf(x) expf(x) /* (glibc declares a private __expf of its own: a macro, not a function) */
static inline float __saturatef(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

/* float atomics: blocks may run on different OpenMP threads */
static inline float atomicAdd(float* addr, float v) {
  float old;
#pragma omp atomic capture
  { old = *addr; *addr += v; }
  return old;
}
