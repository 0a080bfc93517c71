// Shared device/host helpers for the LC-GAN gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LCGAN_OK 0
#define LCGAN_EINVAL (-1)
#define LCGAN_ELAUNCH (-2)

#define DT_F32 0
#define DT_BF16 1

#define ACT_NONE 0
#define ACT_LRELU 1
#define ACT_TANH 2
#define LRELU_SLOPE 0.2f

static inline __host__ __device__ int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- 8-wide feature vectors: the unit every NHWC kernel moves (16 B of bf16, 32 B of f32) -------------
struct F8 { float v[8]; };

template <typename T> struct Feat;
template <> struct Feat<float> {
  static __device__ __forceinline__ F8 load(const float* p) {
    F8 r; f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    r.v[0]=a[0]; r.v[1]=a[1]; r.v[2]=a[2]; r.v[3]=a[3]; r.v[4]=b[0]; r.v[5]=b[1]; r.v[6]=b[2]; r.v[7]=b[3];
    return r;
  }
  static __device__ __forceinline__ void store(float* p, const F8& r) {
    f32x4 a = {r.v[0], r.v[1], r.v[2], r.v[3]}, b = {r.v[4], r.v[5], r.v[6], r.v[7]};
    *(f32x4*)p = a; *(f32x4*)(p + 4) = b;
  }
  static __device__ __forceinline__ float ld1(const float* p) { return *p; }
  static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }          // the value a stored element reads back as
};
template <> struct Feat<__bf16> {
  static __device__ __forceinline__ F8 load(const __bf16* p) {
    F8 r; bf16x8 a = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (float)a[i];
    return r;
  }
  static __device__ __forceinline__ void store(__bf16* p, const F8& r) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)r.v[i];
    *(bf16x8*)p = a;
  }
  static __device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }
  static __device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }
  static __device__ __forceinline__ float rnd(float v) { return (float)(__bf16)v; }
};

__device__ __forceinline__ F8 f8_zero() { F8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0.f;
  return r; }

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == ACT_LRELU) return v > 0.f ? v : v * LRELU_SLOPE;
  if (act == ACT_TANH) return tanhf(v);
  return v;
}
// derivative of act expressed through the SAVED OUTPUT y = act(v) * gain (gain > 0)
__device__ __forceinline__ float act_grad_from_out(float y, int act, float gain) {
  if (act == ACT_LRELU) return (y > 0.f ? 1.f : LRELU_SLOPE) * gain;
  if (act == ACT_TANH) { float t = y / gain; return (1.f - t * t) * gain; }
  return gain;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- optional per-launch profiling (HIP events on the launch stream), see prof.cpp --------------------
#define KID_CONV_IGEMM 0
#define KID_CONV_WGRAD 1
#define KID_WEIGHT_PREP 2
#define KID_STENCIL 3
#define KID_ACT_BWD 4
#define KID_WARP_FWD 5
#define KID_WARP_BWD 6
#define KID_RGB 7
#define KID_LINEAR 8
#define KID_SMALL 9
#define KID_OPTIM 10
#define KID_LAYOUT 11
#define KID_SCALE_REDUCE 12
#define KID_COUNT 13

extern "C" int lcgan_prof_active();
int lcgan_prof_begin(int kid, double flops, double bytes, hipStream_t s, const char* tag);
void lcgan_prof_end(int idx, hipStream_t s);

struct ProfScope {
  hipStream_t s; bool on; int idx;
  ProfScope(int kid, double flops, double bytes, hipStream_t st, const char* tag = nullptr) : s(st), on(lcgan_prof_active() != 0), idx(-1) {
    if (on) idx = lcgan_prof_begin(kid, flops, bytes, s, tag);
  }
  ~ProfScope() { if (on) lcgan_prof_end(idx, s); }
};

static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? LCGAN_OK : LCGAN_ELAUNCH;
}
