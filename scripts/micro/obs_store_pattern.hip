// Dev micro-benchmark: the store pattern of a recording rollout without the game -- every wave writes T slabs of
// `slab` bytes (16 games x 240 B at (20,3)) at stride batch x 240 B, as the four-lane recording kernel does.
// variant 0: stores only; 1: every slab goes through LDS first (write, fence, read back) like qr_build_image + slab store;
// 2: stores only, two waves share a slab (twice the waves, half the bytes each).  Where the 315 MB of observations per
// episode lose against a plain fill (scripts/probe_obs_pattern.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float vf4 __attribute__((ext_vector_type(4)));

// variants 3..: as 1 plus `spin` dependent VALU instructions per step (the game's stages); 4: the four stores of a step
// spread over the spin of the next; 5: odd waves spin before the first step (half a step out of phase)
template <int VARIANT>
__global__ __launch_bounds__(256) void pattern_kernel(float* out, int64_t batch, int steps, float value, int spin) {
  __shared__ __align__(16) float lds[4 * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + wave;
  if (VARIANT == 2) {
    const int64_t g0 = (w >> 1) * 16;
    if (g0 >= batch) return;
    const int half = (int)(w & 1);
    for (int t = 0; t < steps; ++t) {
      float* dst = out + ((int64_t)t * batch + g0) * 60 + half * 480 + lane * 4;  // 1920 B per wave: 2 x 64 x 16 B (last: 56 lanes)
      const vf4 v = {value, value, value, (float)t};
      *reinterpret_cast<vf4*>(dst) = v;
      if (lane < 56) *reinterpret_cast<vf4*>(dst + 256) = v;
    }
    return;
  }
  const int64_t g0 = w * 16;
  if (g0 >= batch) return;
  float* region = lds + wave * 1024;
  for (int t = 0; t < steps; ++t) {
    float* dst = out + ((int64_t)t * batch + g0) * 60 + lane * 4;
    vf4 v[4];
    for (int i = 0; i < 4; ++i) v[i] = vf4{value, value, value, (float)(t + i)};
    if (VARIANT == 5 && t == 0 && (wave & 1)) {
      float x = value;
      for (int k = 0; k < spin / 2; ++k) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
      v[0].x = x;
    }
    if (VARIANT >= 1) {
      for (int i = 0; i < 4; ++i)
        if (i < 3 || lane < 48) *reinterpret_cast<vf4*>(region + (lane + 64 * i) * 4) = v[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const vf4*>(region + ((lane + 64 * i) & 255) * 4);
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(v[i]));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (VARIANT == 4) {
      float x = value;
      for (int i = 0; i < 4; ++i) {
        if (i < 3 || lane < 48) *reinterpret_cast<vf4*>(dst + 256 * i) = v[i];
        for (int k = 0; k < spin / 4; ++k) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
      }
      value = x;
      continue;
    }
    for (int i = 0; i < 4; ++i)
      if (i < 3 || lane < 48) {
        if (VARIANT == 6) { if (value == 123.0f) *reinterpret_cast<vf4*>(dst + 256 * i) = v[i]; }  // (never: compute only)
        else if (VARIANT == 7) __builtin_nontemporal_store(v[i], reinterpret_cast<vf4*>(dst + 256 * i));
        else *reinterpret_cast<vf4*>(dst + 256 * i) = v[i];
      }
    if (VARIANT >= 3) {
      float x = value;
      for (int k = 0; k < spin; ++k) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
      value = x;
    }
  }
}

extern "C" int obs_pattern(int variant, float* out, long long batch, int steps, void* stream, int spin) {
  const int64_t waves = (batch + 15) / 16 * (variant == 2 ? 2 : 1);
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (variant == 0) pattern_kernel<0><<<grid, block, 0, s>>>(out, batch, steps, 1.0f, spin);
  else if (variant == 1) pattern_kernel<1><<<grid, block, 0, s>>>(out, batch, steps, 1.0f, spin);
  else if (variant == 2) pattern_kernel<2><<<grid, block, 0, s>>>(out, batch, steps, 1.0f, spin);
  else if (variant == 3) pattern_kernel<3><<<grid, block, 0, s>>>(out, batch, steps, 1.0f, spin);
  else if (variant == 4) pattern_kernel<4><<<grid, block, 0, s>>>(out, batch, steps, 1.0f, spin);
  else if (variant == 5) pattern_kernel<5><<<grid, block, 0, s>>>(out, batch, steps, 1.0f, spin);
  else if (variant == 6) pattern_kernel<6><<<grid, block, 0, s>>>(out, batch, steps, 1.0f, spin);
  else pattern_kernel<7><<<grid, block, 0, s>>>(out, batch, steps, 1.0f, spin);
  return (int)hipGetLastError();
}
