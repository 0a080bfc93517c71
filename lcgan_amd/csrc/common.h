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

// ---- raw buffer loads: resource in scalar registers + ONE 32-bit byte offset per lane (no 64-bit address arithmetic in vector
// registers; tensors addressed this way are < 4 GB, checked by the launchers) ---------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
template <typename T> __device__ __forceinline__ F8 buf_load8(__amdgpu_buffer_rsrc_t r, unsigned byte_off);
template <> __device__ __forceinline__ F8 buf_load8<__bf16>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
  F8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.v[i] = (float)a[i];
  return o;
}
template <> __device__ __forceinline__ F8 buf_load8<float>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
  const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off + 16, 0, 0));
  F8 o; o.v[0]=a[0]; o.v[1]=a[1]; o.v[2]=a[2]; o.v[3]=a[3]; o.v[4]=b[0]; o.v[5]=b[1]; o.v[6]=b[2]; o.v[7]=b[3];
  return o;
}
// 8-channel dot product of two vectors as loaded (no unpacking for bf16: four v_dot2c_f32_bf16, exact products, f32 accumulation)
template <typename T> struct Vec8;
template <> struct Vec8<__bf16> {
  typedef bf16x8 Raw;
  static __device__ __forceinline__ Raw load(const __bf16* p) { return *(const bf16x8*)p; }
  static __device__ __forceinline__ Raw buf(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); }
  static __device__ __forceinline__ float dot(const Raw& a, const Raw& b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) d = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{a[2 * i], a[2 * i + 1]}, bf16x2{b[2 * i], b[2 * i + 1]}, d, false);
    return d;
  }
};
template <> struct Vec8<float> {
  typedef F8 Raw;
  static __device__ __forceinline__ Raw load(const float* p) { f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    F8 r; r.v[0]=a[0]; r.v[1]=a[1]; r.v[2]=a[2]; r.v[3]=a[3]; r.v[4]=b[0]; r.v[5]=b[1]; r.v[6]=b[2]; r.v[7]=b[3]; return r; }
  static __device__ __forceinline__ Raw buf(__amdgpu_buffer_rsrc_t r, unsigned off) { return buf_load8<float>(r, off); }
  static __device__ __forceinline__ float dot(const Raw& a, const Raw& b) {
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) d += a.v[i] * b.v[i];
    return d;
  }
};

// n / d and n % d for the index decompositions of the elementwise kernels: d is a power of two in every LC-GAN shape (shift >= 0:
// one shift / mask instead of a 32-bit multiply-high sequence); other sizes take the division
struct FastDiv {
  unsigned d; int shift;
  __host__ __device__ FastDiv() : d(1), shift(0) {}
  __host__ explicit FastDiv(unsigned dd) : d(dd), shift(-1) { for (int s = 0; s < 31; ++s) if ((1u << s) == dd) shift = s; }
  __device__ __forceinline__ unsigned div(unsigned n) const { return shift >= 0 ? n >> shift : n / d; }
  __device__ __forceinline__ unsigned mod(unsigned n) const { return shift >= 0 ? n & (d - 1) : n % d; }
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
