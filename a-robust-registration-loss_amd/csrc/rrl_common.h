// rrl_common.h -- shared device helpers for librrl_hip (gfx950 only).
//
// Everything that decides a label is written op-for-op in fp32 in the reference's
// evaluation order (code/loss.py:84-88, 94-110) and the library is compiled with
// -ffp-contract=off: a fused multiply-add in dist_sq() flips labels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rrl.h"

#define RRL_EPS 2e-4f    // code/loss.py:88
#define RRL_CTHR 1.731f  // code/loss.py:109
#define PTRI_STRIDE 12   // floats per prepared triangle

typedef float v2f __attribute__((ext_vector_type(2)));

#define RRL_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// (dAC - proj) + eps of code/loss.py:84-88 for scalar (T = float) or two lines at
// once (T = v2f -> v_pk_mul_f32 / v_pk_add_f32, each half rounded like the scalar op).
template <typename T>
__device__ __forceinline__ T dist_sq(float px, float py, float pz, T ux, T uy, T uz, T ox, T oy,
                                     T oz) {
    T ax = px - ox, ay = py - oy, az = pz - oz;
    T dot = ax * ux;
    dot = dot + ay * uy;
    dot = dot + az * uz;
    T proj = dot * dot;
    T dac = ax * ax;
    dac = dac + ay * ay;
    dac = dac + az * az;
    T x = dac - proj;
    return x + RRL_EPS;
}

__device__ __forceinline__ float norm3(float x, float y, float z) {
    float s = x * x;
    s = s + y * y;
    s = s + z * z;
    return sqrtf(s);  // correctly rounded (v_sqrt_f32 + fix-up)
}

__device__ __forceinline__ uint32_t f2u(float x) { return __float_as_uint(x); }
