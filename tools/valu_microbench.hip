// valu_microbench.hip -- what fp32 VALU issue rates does gfx950 sustain for the scan kernel's
// instruction mix?  Inline-asm chains (hipcc SLP-packs plain C++ scalar code, so the opcode is
// pinned by hand): 8 independent accumulators per lane, each iteration = 8 mul + 8 add, as
//   kind 0  v_mul_f32 / v_add_f32          all-VGPR operands
//   kind 1  v_pk_mul_f32 / v_pk_add_f32    all-VGPR operands
//   kind 2  v_mul_f32 / v_add_f32          one SGPR operand
//   kind 3  v_pk_mul_f32 / v_pk_add_f32    one SGPR-pair operand (op_sel_hi broadcast)
// at 1..8 waves per SIMD.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o gpurun_out/valu_mb && gpurun_out/valu_mb
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 2048

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float s0, float s1) {
    if constexpr (KIND == 0 || KIND == 2) {
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
              a6 = a0 + 6, a7 = a0 + 7;
        float m = 1.0f + threadIdx.x * 1e-9f, c = threadIdx.x * 1e-9f;
        for (int it = 0; it < ITERS; ++it) {
            if constexpr (KIND == 0) {
#define MUL(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a##i) : "v"(m));
#define ADD(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a##i) : "v"(c));
                REP8(MUL) REP8(ADD) REP8(MUL) REP8(ADD)
#undef MUL
#undef ADD
            } else {
#define MUL(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a##i) : "s"(s0));
#define ADD(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a##i) : "s"(s1));
                REP8(MUL) REP8(ADD) REP8(MUL) REP8(ADD)
#undef MUL
#undef ADD
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else {
        v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f,
            a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
        v2f m = {1.0f + threadIdx.x * 1e-9f, 1.0f - threadIdx.x * 1e-9f};
        v2f c = {threadIdx.x * 1e-9f, threadIdx.x * 2e-9f};
        v2f sm = {s0, s0}, sc = {s1, s1};
        for (int it = 0; it < ITERS; ++it) {
            if constexpr (KIND == 1) {
#define MUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(m));
#define ADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
                REP8(MUL) REP8(ADD) REP8(MUL) REP8(ADD)
#undef MUL
#undef ADD
            } else {
#define MUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a##i) : "s"(sm));
#define ADD(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a##i) : "s"(sc));
                REP8(MUL) REP8(ADD) REP8(MUL) REP8(ADD)
#undef MUL
#undef ADD
            }
        }
        v2f r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
    }
}

template <int KIND>
double run(float *d, int blocks) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001f, 1e-9f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001f, 1e-9f);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float *d;
    (void)hipMalloc(&d, sizeof(float) * 256 * 256 * 64);
    const char *names[4] = {"v_mul/add_f32 vgpr", "v_pk_mul/add_f32 vgpr", "v_mul/add_f32 sgpr-op",
                            "v_pk_mul/add_f32 sgpr-op"};
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256 * wps;  // 4 waves per block -> wps waves per SIMD on 256 CUs
        double t[4] = {run<0>(d, blocks), run<1>(d, blocks), run<2>(d, blocks), run<3>(d, blocks)};
        for (int kd = 0; kd < 4; ++kd) {
            double insts = (double)blocks * 4 * ITERS * 32;  // wave-instructions
            double laneops = insts * 64 * ((kd & 1) ? 2 : 1);
            printf("waves/SIMD %d  %-26s %.3f ms  %6.2f T lane-ops/s  %.2f cycles/wave-inst @2.4GHz\n",
                   wps, names[kd], t[kd], laneops / t[kd] / 1e9,
                   t[kd] * 1e-3 * 2.4e9 / (insts / (256.0 * 4)));
        }
    }
    return 0;
}
