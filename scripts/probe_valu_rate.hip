// Dev probe: what the SIMD's VALU pipe sustains for the instruction classes the game kernels are made of, with 1, 2
// and 4 waves per SIMD.  (MI355X_MICROARCH.md: a wave64 VALU instruction takes 2 cycles on the SIMD-32, one wave
// alone issues one every 4.)  Each wave runs `iters` iterations of 64 instructions of one class on 8 independent
// register chains; cycles per instruction per SIMD = kernel cycles / (iters * 64 * waves_per_simd).
//   hipcc --offload-arch=gfx950 -O3 scripts/probe_valu_rate.hip -o build_probe/probe_valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int CLASS>
__global__ __launch_bounds__(64) void rate_kernel(float* out, int iters, float seed) {
  float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7;
  float a = seed * 0.5f, b = seed * 0.25f;
  unsigned long long m = 0x5555555555555555ull;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (CLASS == 0) {  // v_add_f32 (VOP2)
#define X(k) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r##k) : "v"(a));
        REP8(X)
#undef X
      } else if constexpr (CLASS == 1) {  // v_max3_f32 (VOP3, three VGPR sources)
#define X(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r##k) : "v"(a), "v"(b));
        REP8(X)
#undef X
      } else if constexpr (CLASS == 2) {  // v_cndmask_b32 with an SGPR-pair mask (VOP3)
#define X(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r##k) : "v"(a), "s"(m));
        REP8(X)
#undef X
      } else if constexpr (CLASS == 3) {  // v_mov_b32 DPP quad_perm
#define X(k) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r##k));
        REP8(X)
#undef X
      } else if constexpr (CLASS == 4) {  // v_sub_f32 with a DPP operand
#define X(k) asm volatile("v_sub_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r##k) : "v"(a));
        REP8(X)
#undef X
      } else if constexpr (CLASS == 5) {  // v_cmp writing an SGPR pair (VOP3) -- result unused apart from the clobber
#define X(k) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(r##k), "v"(a) : "s20", "s21");
        REP8(X)
#undef X
      } else if constexpr (CLASS == 6) {  // v_cmp to vcc + v_cndmask from vcc (a dependent pair through vcc)
#define X(k) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r##k) : "v"(a) : "vcc");
        REP8(X)
#undef X
      } else if constexpr (CLASS == 7) {  // v_min_u32 (VOP2 integer)
#define X(k) asm volatile("v_min_u32 %0, %1, %0" : "+v"(r##k) : "v"(a));
        REP8(X)
#undef X
      } else if constexpr (CLASS == 9) {  // v_max_f32 (VOP2)
#define X(k) asm volatile("v_max_f32 %0, %1, %0" : "+v"(r##k) : "v"(a));
        REP8(X)
#undef X
      } else if constexpr (CLASS == 10) {  // v_med3_f32 (what hk_fmin / hk_fmax compile to)
#define X(k) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r##k) : "v"(a), "v"(b));
        REP8(X)
#undef X
      } else if constexpr (CLASS == 11) {  // the pair test with vcc forms: 3 sub, max3, min3, min, cmp vcc, cndmask vcc, max
#define X(k) asm volatile("v_sub_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %2\n\tv_sub_f32 %0, %0, %1\n\tv_max3_f32 %0, %0, %1, %2\n\tv_min3_f32 %0, %0, %1, %2\n\tv_min_f32 %0, %1, %0\n\tv_cmp_lt_f32_e32 vcc, 0, %0\n\tv_cndmask_b32_e32 %0, %1, %0, vcc\n\tv_max_f32 %0, %2, %0" : "+v"(r##k) : "v"(a), "v"(b) : "vcc");
        REP8(X)
#undef X
      } else if constexpr (CLASS == 12) {  // ... and as the compiler emits it: cmp -> SGPR pair, cndmask e64, med3 accumulators
#define X(k) asm volatile("v_sub_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %2\n\tv_sub_f32 %0, %0, %1\n\tv_max3_f32 %0, %0, %1, %2\n\tv_min3_f32 %0, %0, %1, %2\n\tv_med3_f32 %0, %0, %1, %2\n\tv_cmp_lt_f32_e64 s[20:21], 0, %0\n\tv_cndmask_b32_e64 %0, %1, -%0, s[20:21]\n\tv_med3_f32 %0, %0, %2, %1" : "+v"(r##k) : "v"(a), "v"(b) : "s20", "s21");
        REP8(X)
#undef X
      } else if constexpr (CLASS == 8) {  // the pair test's shape: 3 sub, max3, min3, cmp->sgpr, cndmask, min (per chain pair)
#define X(k) asm volatile("v_sub_f32 %0, %0, %1\n\tv_max3_f32 %0, %0, %1, %2\n\tv_cmp_lt_f32_e64 s[20:21], 0, %0\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r##k) : "v"(a), "v"(b) : "s20", "s21");
        REP8(X)
#undef X
      }
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

template <int CLASS>
double run(float* out, int waves_per_simd, int iters, int per_iter) {
  const int grid = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<CLASS>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(rate_kernel<CLASS>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_inst_simd = (double)ms * 1e6 / 5 / ((double)iters * per_iter * waves_per_simd);
  return ns_per_inst_simd;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
  const int iters = 20000;
  const char* names[] = {"v_add_f32 (VOP2)", "v_max3_f32 (VOP3)", "v_cndmask sgpr mask (VOP3)", "v_mov_b32 dpp quad_perm",
                         "v_sub_f32 dpp", "v_cmp -> sgpr pair (VOP3)", "v_cmp vcc + v_cndmask vcc", "v_min_u32 (VOP2)",
                         "sub, max3, cmp->sgpr, cndmask", "v_max_f32 (VOP2)", "v_med3_f32 (VOP3)",
                         "pair test, vcc forms (9 instr)", "pair test, as compiled (9 instr)"};
  const int per_iter[] = {64, 64, 64, 64, 64, 64, 128, 64, 256, 64, 64, 576, 576};
  printf("ns per wave64 instruction per SIMD (2.4 GHz: 2 cycles = 0.83 ns, 4 cycles = 1.67 ns)\n");
  for (int c = 0; c < 13; ++c) {
    double v[3];
    int w[3] = {1, 2, 4};
    for (int k = 0; k < 3; ++k) {
      switch (c) {
        case 0: v[k] = run<0>(out, w[k], iters, per_iter[c]); break;
        case 1: v[k] = run<1>(out, w[k], iters, per_iter[c]); break;
        case 2: v[k] = run<2>(out, w[k], iters, per_iter[c]); break;
        case 3: v[k] = run<3>(out, w[k], iters, per_iter[c]); break;
        case 4: v[k] = run<4>(out, w[k], iters, per_iter[c]); break;
        case 5: v[k] = run<5>(out, w[k], iters, per_iter[c]); break;
        case 6: v[k] = run<6>(out, w[k], iters, per_iter[c]); break;
        case 7: v[k] = run<7>(out, w[k], iters, per_iter[c]); break;
        case 8: v[k] = run<8>(out, w[k], iters, per_iter[c]); break;
        case 9: v[k] = run<9>(out, w[k], iters, per_iter[c]); break;
        case 10: v[k] = run<10>(out, w[k], iters, per_iter[c]); break;
        case 11: v[k] = run<11>(out, w[k], iters / 4, per_iter[c]); break;
        default: v[k] = run<12>(out, w[k], iters / 4, per_iter[c]); break;
      }
    }
    printf("%-34s  1 wave/SIMD %.2f   2 waves %.2f   4 waves %.2f\n", names[c], v[0], v[1], v[2]);
  }
  return 0;
}
