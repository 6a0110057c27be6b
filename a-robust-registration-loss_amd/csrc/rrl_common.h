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

// roctx ranges around the stages of a step (host-side markers for rocprofv3 --marker-trace): active
// only with RRL_ROCTX=1 in the environment, in which case librocprofiler-sdk-roctx.so (or libroctx64.so)
// is dlopen'ed on first use -- no link-time dependency, no cost otherwise.  Defined in rrl_scan.hip.
void rrl_range_push(const char *name);
void rrl_range_pop(void);
struct RrlRange {
    explicit RrlRange(const char *name) { rrl_range_push(name); }
    ~RrlRange() { rrl_range_pop(); }
    RrlRange(const RrlRange &) = delete;
    RrlRange &operator=(const RrlRange &) = delete;
};

// Fill / copy as KERNELS.  The library never issues hipMemsetAsync / hipMemcpyAsync on device buffers (the one
// device-to-host read-back of rrl_loss_forward_info waits for the stream and cannot be captured anyway): inside a
// captured hipGraph (the bench and the demo replay their step as one) memset nodes were observed
// to race with the kernels that follow them on this stack -- the captured demo step
// intermittently read half-initialised Chamfer keys until the memsets became a kernel.
static __global__ void rrl_fill_words_kernel(uint32_t *__restrict__ p, uint32_t v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
static __global__ void rrl_copy_words_kernel(uint32_t *__restrict__ d, const uint32_t *__restrict__ s, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
// nbytes must be a multiple of 4 (every buffer of this library is)
static inline int rrl_fill(void *p, uint32_t word, size_t nbytes, hipStream_t s) {
    const size_t n = nbytes / 4;
    if (n == 0) return 0;
    size_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(rrl_fill_words_kernel, dim3((unsigned)nb), dim3(256), 0, s, (uint32_t *)p, word, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
static inline int rrl_copy(void *d, const void *src, size_t nbytes, hipStream_t s) {
    const size_t n = nbytes / 4;
    if (n == 0) return 0;
    size_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(rrl_copy_words_kernel, dim3((unsigned)nb), dim3(256), 0, s, (uint32_t *)d,
                       (const uint32_t *)src, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

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

// ---- agent-coherent (sc1) accesses for data handed from one workgroup to another INSIDE a launch (the chained step's build +
//      scan launch, rrl_cull_scan.inc; cdna_hip_programming.md Guideline 16 R1): the producer's 16-byte stores are
//      WRITE-THROUGH (no release fence: every storing wave drains, barrier, one relaxed ticket), the consumer's loads bypass
//      its CU's L1 (no acquire fence).  16-byte accesses go through a buffer descriptor built from wave-uniform values
//      (base: kernel argument + blockIdx-derived offset; bytes < 2^31), 4-byte ones are relaxed agent-scope atomics.
typedef unsigned int rrl_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rrl_rsrc(const void *base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(bytes > 0x7ffffff0u ? 0x7ffffff0u : bytes), 0x00020000);
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
    const rrl_v4u u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)byte_off, 0, 16);
}
__device__ __forceinline__ float4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const rrl_v4u u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
__device__ __forceinline__ void st4_sc1(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld4_sc1(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float norm3(float x, float y, float z) {
    float s = x * x;
    s = s + y * y;
    s = s + z * z;
    return sqrtf(s);  // correctly rounded (v_sqrt_f32 + fix-up)
}

__device__ __forceinline__ uint32_t f2u(float x) { return __float_as_uint(x); }

// Per-triangle threshold of code/loss.py:94-110 and its exact squared form:
//   thr  = mean(|P1-P0|, |P2-P0|, |P1-P2|) * 1.731 / 2
//   thr2 = min { x >= 0 : sqrtf(x) >= thr }  (sqrtf correctly rounded and monotone), so that
//          sqrtf(x) < thr  <=>  x < thr2 exactly.  Start at fl(thr*thr) and walk a few ulps.
// e01_out (optional): max(|P1-P0|, |P2-P0|) as evaluated here -- how far points 1, 2 can sit from point 0
// (the culled scan's NaN reach, rrl_cull.hip).
__device__ __forceinline__ void tri_thresholds(const float *c, float *thr_out, float *thr2_out, float *e01_out = nullptr) {
    float e0 = norm3(c[3] - c[0], c[4] - c[1], c[5] - c[2]);
    float e1 = norm3(c[6] - c[0], c[7] - c[1], c[8] - c[2]);
    float e2 = norm3(c[3] - c[6], c[4] - c[7], c[5] - c[8]);
    if (e01_out) *e01_out = fmaxf(e0, e1);
    float delta = ((e0 + e1) + e2) / 3.0f;
    float t = delta * RRL_CTHR;
    float thr = t / 2.0f;
    float x = thr * thr;
    if (thr > 0.0f && x < INFINITY) {
        for (int it = 0; it < 8 && x > 0.0f && sqrtf(x) >= thr; ++it)
            x = __uint_as_float(__float_as_uint(x) - 1u);
        for (int it = 0; it < 16 && sqrtf(x) < thr; ++it)
            x = __uint_as_float(__float_as_uint(x) + 1u);
    } else if (!(thr > 0.0f)) {
        x = 0.0f;  // thr == 0 (degenerate triangle) or NaN: nothing is strictly closer
    }
    *thr_out = thr;
    *thr2_out = x;
}

// NaN-impossibility bound of a line v = (dir, x0) against a cloud with max |P|^2 = pm
// (DESIGN.md "NaN bound"): |dir|^2 <= 1 + 1e-6 and (|x0| + max|P|)^2 <= 100.
__device__ __forceinline__ bool rrl_line_safe(const float *v, float pm) {
    float s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float o2 = v[3] * v[3] + v[4] * v[4] + v[5] * v[5];
    float a2 = o2 + pm + 2.0f * sqrtf(o2 * pm);
    return (s <= 1.000001f) && (a2 <= 100.0f);
}


// ---- wave64 cross-lane helpers on DPP (no LDS crossbar: ds_bpermute-based __shfl trees cost
//      hundreds of cycles per step under load, DPP steps cost one VALU issue each) ----------
// inclusive prefix sum over the 64 lanes (the sequence of LLVM's AMDGPU atomic optimizer):
// row_shr 1,2,4,8 inside each row of 16, then row_bcast:15 / row_bcast:31 across rows
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}
// sum over the 64 lanes, returned in every lane (fixed association: deterministic)
__device__ __forceinline__ float wave_sum(float v) {
#define RRL_DPP_ADD(ctrl, rm)                                                                   \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rm, 0xf, false))
    RRL_DPP_ADD(0x111, 0xf); RRL_DPP_ADD(0x112, 0xf); RRL_DPP_ADD(0x114, 0xf); RRL_DPP_ADD(0x118, 0xf);
    RRL_DPP_ADD(0x142, 0xa); RRL_DPP_ADD(0x143, 0xc);
#undef RRL_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_sum_i(int v) {
    return __builtin_amdgcn_readlane(wave_incl_scan(v), 63);
}

// the value of lane A of every quad of four lanes (DPP quad_perm [A, A, A, A]; the source lanes must be active)
template <int A>
__device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), A * 0x55, 0xf, 0xf, true));
}

// all-reduce inside each row of 16 lanes (butterfly on DPP: quad_perm [1,0,3,2], [2,3,0,1],
// row_half_mirror, row_mirror); every lane of the row receives the result
#define RRL_DPP_F(v, ctrl) \
    __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xf, 0xf, false))
__device__ __forceinline__ float row16_min(float v) {
    v = fminf(v, RRL_DPP_F(v, 0xB1)); v = fminf(v, RRL_DPP_F(v, 0x4E));
    v = fminf(v, RRL_DPP_F(v, 0x141)); v = fminf(v, RRL_DPP_F(v, 0x140));
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, RRL_DPP_F(v, 0xB1)); v = fmaxf(v, RRL_DPP_F(v, 0x4E));
    v = fmaxf(v, RRL_DPP_F(v, 0x141)); v = fmaxf(v, RRL_DPP_F(v, 0x140));
    return v;
}
// wave-wide: row results of lanes 0, 16, 32, 48 combined through SGPRs
__device__ __forceinline__ float wave_min(float v) {
    v = row16_min(v);
    const int b = __float_as_int(v);
    return fminf(fminf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                 fminf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    const int b = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}
