// Implicit-GEMM convolution family for gfx950 (CDNA4): forward / data-gradient / weight-gradient of the
// k in {1,3}, stride in {1,2} convolutions of LC-GAN's generator and discriminator, NHWC activations,
// bf16 MFMA (v_mfma_f32_32x32x16_bf16) with fp32 accumulation.
//
// Replaces the ATen calls of the reference:  F.conv2d custom_layers.py:41,43,83 ; F.conv_transpose2d :78 ;
// and their autograd derivatives (convolution_backward, incl. the double-backward used by loss.py:28-33).
//
//  * One "NT" kernel (conv_igemm_kernel) serves conv forward (stride 1/2), the data gradient of a stride-1
//    conv (flipped taps, transposed weights) and -- through four sub-pixel phases with 1/2/2/4 taps --
//    the data gradient of a stride-2 conv == the x2 transposed convolution of ModulatedConv2d(up=2).
//    Geometry is a tap table (dy, dx, weight-tap) per phase; the M index walks (b, i, j) of an Hm x Wm grid.
//  * Style modulation is folded in as a per-(sample, in-channel) pre-scale applied while the A tile is staged
//    and a per-(sample, out-channel) demodulation post-scale in the epilogue, so the per-sample weights the
//    reference materialises ([B,Co,Ci,3,3], custom_layers.py:62-72) never exist.
//  * Epilogue fuses demod, bias, leaky-ReLU, gain and a residual add.
//  * fp32 feature maps (parity mode) run the same MFMA path with a 3-way bf16 split x = p0 + p1 + p2 of both operands
//    (24 mantissa bits) and the 6 products with i + j <= 2 -- fp32-grade (~2^-24) products, template parameter P == 3.
//    (A 2-way split, ~2^-17, is not enough: the backward chain D -> G amplifies relative error ~200x.)
//  * conv_wgrad_kernel reduces over positions: operands are staged row-major ([position][channel]) and read
//    with ds_read_b64_tr_b16 so no transpose pass is needed; split-K partials are combined with fp32 atomics.
#include "common.h"
#include <type_traits>

#include <algorithm>
#include <cstdio>

extern "C" int lcgan_scale_reduce(void* u, const void* x, const float* sc, float* gs, int B, int HW, int C, int dtype, void* stream);
extern "C" int lcgan_scale_reduce_res(void* u, const void* x, const float* sc, float* gs, const void* res, int B, int HW, int C, int dtype, void* stream);
extern "C" int lcgan_avgpool2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
extern "C" int lcgan_conv_fwd_m(const void* x, const void* wp, void* y, int B, int Hin, int Win, int Cin, int Cout, int N, int k, int stride,
                                const float* pre, const float* post, const float* bias, float bias_scale, int act, float gain, const void* residual,
                                int residual_half, const void* xs, float* gs, void* pool_out, void* mask_out, int* mask_written, int dtype, void* stream);

namespace {

bool g_mask_written = false;              // set by the launch path that wrote ConvArgs::mask_out in its epilogue
bool g_pool_written = false;              // set by the launch path that wrote ConvArgs::pool_out in its epilogue (else lcgan_conv_fwd runs the pooling kernel)
int g_use_halo = 1;                       // lcgan_set_option(0, ...): bf16 halo-tile fast path on/off (A/B testing)
int g_use_splitk = 1;                     // lcgan_set_option(1, ...): split-K for small-M convolutions
int g_mfma16 = 0;                         // lcgan_set_option(4, ...): halo kernel uses v_mfma_f32_16x16x32_bf16 (1) or 32x32x16 (0, default:
                                          // measured 8-10 % faster here -- the 16x16 form needs 140 VGPRs and loses the second workgroup per CU)
int g_wgrad3_small = 0;                   // lcgan_set_option(5, ...): row-segment wgrad kernel for 16/8-wide layers and 1x1 kernels: 0 = only without per-sample scales (multi-sample splits), 1 = always, 2 = never
                                          // (default off: measured 0.5 ms/iteration SLOWER than the generic kernel on those shapes)
int g_wgrad3_wgs = 0;                     // lcgan_set_option(2, ...): 0 = cost-model split of the row-segment wgrad kernel, > 0 = explicit workgroup target
int g_halo_min_wgs = 192;                 // lcgan_set_option(6, ...): halo launches with fewer workgroups go to the split-K implicit GEMM
int g_wgrad_slab_min = 4;                 // lcgan_set_option(8, ...): row-segment wgrad launches with at least this many splits reduce through a slab instead of atomics (0 = never)
int g_halo_narrow_min_wgs = 256;          // lcgan_set_option(7, ...): narrow-layer halo kernel (Cout <= 64, 16 x 32 tiles) from this many workgroups; 0 = never
int g_wgrad3_pack = 1;                    // lcgan_set_option(9, ...): packed channel groups in the row-segment wgrad kernel for layers with <= 64 channels
int g_halo_dma = 2;                       // lcgan_set_option(10, ...): LDS-DMA staging in the halo kernel (stride-1 geometries without input scales)
int g_halo_dma_mod = 2;                   // lcgan_set_option(11, ...): the same structure for convolutions with per-sample input scales (halo by DMA, scaled in place in LDS): 0 = off, 1 / 2 = taps per step
int g_wgrad_dma = 3;                      // lcgan_set_option(12, ...): LDS-DMA staging in the row-segment weight-gradient kernel (3x3): 0 = off, 1 = stride 1 with the one-workgroup-per-CU split, 2 = stride 1, split for two workgroups per CU, 3 = also stride 2 (32-position chunks)
int g_halo_s2dma = 4;                     // lcgan_set_option(13, ...): stride-2 forward 3x3 on the parity-plane LDS-DMA structure: 0 off, 1 = layers without per-sample input scales, 2 = all (two stages per workgroup, one workgroup per CU), 4 = as 2 but unscaled layers with ONE stage per workgroup and two workgroups per CU
int g_halo_nb_group_kb = 8192;               // lcgan_set_option(14, ...): KB of weights (all taps x 128 rows x Cin) that concurrent channel blocks of one tile may hold in an XCD's L2; 0 = channel blocks slowest (one pass over the input per block)
int g_wgrad_xcd = 1;                      // lcgan_set_option(15, ...): row-segment wgrad workgroups of one split share an XCD (1-D grid; see WG3_INDEX)
int g_wgrad_low_direct = 1;               // lcgan_set_option(20, ...): launch plan of the small-grid (8 x 8, 16 x 16) weight gradients without per-sample scales: splits chosen by the measured
                                          // cost model in conv_wgrad_impl, one split = the epilogue writes the finished gradient in weight layout, XCD order only from 8 splits
                                          // (0 = the round-2 plan: >= 1024 positions per split, XCD order always, atomics below 4 splits)
int g_wgrad_low_parts = 0;                // lcgan_set_option(21, ...): force the number of splits of the small-grid weight gradients (tuning experiments; 0 = automatic)
int g_s2duo = 1;                           // lcgan_set_option(26, ...): stride-2 forward convolutions without per-sample scales / residual / fused reductions on the two-team kernel (conv_s2duo_kernel)
int g_wgrad_slab_bf16 = 1;                 // lcgan_set_option(25, ...): bf16 launches store their split partial tiles in bf16 (fp32 accumulation inside a split and across the splits)
int g_halo_phase_x = 0;                    // lcgan_set_option(24, ...): the 4 sub-pixel phases of a transposed convolution tile run side by side on one XCD (1-D grid) instead of as grid.z planes (measured: fabric reads -3.6x, time 0 ... +25 %: off)
int g_flow_wgrad = 1;                     // lcgan_set_option(23, ...): one-pass weight gradient of the flow layer's 1x1 GEMM (flow_wgrad_kernel); 0 = the row-segment kernel
int g_halo_split = 0;                     // lcgan_set_option(22, ...): halo launches below option 6's workgroup count split their input-channel range so that about this many workgroups run
                                          // (stride-1 LDS-DMA structure).  0 = off, the default: such launches go to the generic split-K kernel.  Measured with 256: the conv launches
                                          // of an iteration +0.35 ms at batch 4, +0.5 at batch 8, neutral at batch 32 -- the last split re-reads nsplit x 128 KB of partials per
                                          // tile through device-scope loads (16 dependent round trips to memory), which costs more than the generic kernel's extra L2 traffic
int g_splitk_slabs = 8;                  // lcgan_set_option(19, ...): split-K launches of the 8-wave generic kernel with up to this many splits exchange partials through per-split slabs and
                                          // the last split to arrive finishes the tile; more splits (or 0) = atomics + the finalize launch (the last split's serial sum grows with the count:
                                          // measured -5..-7 us per launch at 2-4 splits, -1.6 at 8, +5 at 16, +14 at 32)
int g_igemm_dma = 3;                      // lcgan_set_option(16, ...): LDS-DMA staging in the generic implicit-GEMM kernel (bf16, no input scales, Cin % 32 == 0)
int g_halo_wmod_mb = 40;                  // (measured at batch 32: 9 MB of copies (128 x 128 layers) -56 us per launch, 38 MB (256 x 256) -25 us, 75 MB (512 x 256) +89 us: the copies stop fitting the L2s)
                                          // lcgan_set_option(18, ...): convolutions with per-sample INPUT scales (modulated convs and their data gradients) whose per-sample
                                          // weight copies (B x taps x N x Kpad bf16) fit this many MB fold the scales into the weights once and run the unscaled kernels; 0 = never
int g_wgrad_prescale_mb = 180;            // lcgan_set_option(17, ...): weight gradients with per-sample operand scales whose two operands together are at most this many MB
                                          // get the scales applied ONCE by an elementwise pass (bf16) and then run as ONE batch-wide reduction; 0 = never
int g_dbg_no_atomics = 0;                 // lcgan_set_option(3, ...): experiments, bit mask (the wgrad3 no-atomics switch is gone: it sat in the epilogue);
                                          // halo kernel: 8 = linear tile order (the store / emit / main-loop skipping switches used for the
                                          // fixed-cost analysis in DESIGN.md were removed again: they sat in the hot epilogue)

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_ROW = 40;               // bf16 per staged row: 32 + 8 pad (80 B stride: conflict-free ds_read_b128)
constexpr int TILE = BM * LDS_ROW;        // elements per staged operand tile

struct TapTable { int n; int dy[9]; int dx[9]; int wt[9]; };

// ---- split-K slab exchange: what orders it ---------------------------------------------------------------------------------------
// Partial tiles cross XCDs (non-coherent L2s) as RELAXED agent-scope atomic stores / loads (`sc1`: written through, read past the local
// L2) around a RELAXED agent-scope arrival counter.  Two different things make that correct, and neither is "the compiler happens to keep
// relaxed accesses in program order":
//   * language level: a release fence sits between a thread's slab stores and the counter add its workgroup makes on its behalf, and an
//     acquire fence between that add (whose returned value names the last arriver) and the slab loads.  Fences order ALL atomic accesses
//     of the thread across them whatever their own ordering (C++ [atomics.fences]), and the workgroup barriers carry the happens-before
//     from every storing thread to the adding lane and from it to every loading thread.  Workgroup scope is enough for that (the
//     cross-workgroup edge is the counter's own modification order) and costs only waitcnts here; an AGENT-scope release / acquire would
//     emit `buffer_wbl2 sc1` / `buffer_inv sc1` per wave: a write-back / invalidate of the whole L2, measured at +55 us per launch with
//     __threadfence(), for data that never sits dirty in L2.
//   * hardware level: `s_waitcnt vmcnt(0)` retires the write-through stores at the memory side before the barrier releases the adding
//     lane (MI355X_MICROARCH.md, inter-workgroup visibility: "16-B sc1 stores + asm vmcnt(0) + agent-scope atomic add; the workgroup
//     whose add came last loads after a workgroup barrier that wave then joins").  The asm carries a "memory" clobber as well.
// tests: test_conv_igemm_dma_variants asserts run-to-run bit identity of split launches (a lost partial would break it).
#define SLAB_PUBLISH_FENCE()                                                                                          \
  do {                                                                                                                \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                                                            \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                  \
  } while (0)
#define SLAB_CONSUME_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup")

inline int log2_exact(long long v) { int s = 0; while ((1ll << s) < v) ++s; return (1ll << s) == v ? s : -1; }   // log2 of a power of two, else -1

// quotient and remainder by a launch constant: a shift where the host found a power of two (sh = its log2), else the division
__device__ __forceinline__ void divmod_sh(unsigned n, int d, int sh, int& q, int& r) {
  if (sh >= 0) { q = (int)(n >> sh); r = (int)(n & ((1u << sh) - 1u)); }
  else { q = (int)(n / (unsigned)d); r = (int)(n - (unsigned)q * (unsigned)d); }
}

struct ConvArgs {
  const void* x; const __bf16* w; size_t w_part; void* y;   // w: [P][taps][N][Kpad], w_part = elements per part
  const float* pre; const float* post; const float* bias; const void* residual;
  int B, Hin, Win, Cin;
  int Hout, Wout, Cout;
  int Hm, Wm, M;
  int N, Kpad, kc_per_tap;
  int in_mul, out_mul;
  int pre_stride, post_stride;
  float bias_scale, gain; int act;
  const void* xs; float* gs;                 // fused style-gradient reduction: gs[b,n] += sum_pixels xs[b,p,n] * acc  (xs: [B,Hout,Wout,Cout]; y = post * acc)
  int res_half;                              // residual is [B,Hout/2,Wout/2,Cout]: add 0.25 * residual[oy/2][ox/2] (avg_pool2d adjoint)
  int lw, lh;                                // log2 of Wout / Hout where they are powers of two, else -1 (the half-resolution residual's index without two divisions per ELEMENT)
  int l_hwm, l_wm, l_kc;                     // the same for Hm * Wm, Wm and kc_per_tap (conv_igemm8_kernel: row decode, and the tap of a chunk once per main-loop stage)
  void* pool_out;                            // optional by-product [B,Hout/2,Wout/2,Cout] = avg_pool2d(y, 2) (the next DiscriminatorBlock's skip input)
  unsigned* mask_out;                        // optional by-product [B*Hout*Wout][Cout/32] words: bit c%32 of word c/32 = (pre-activation > 0), what the activation backward needs of y
  int nsplit; float* ws;                     // split-K: blockIdx.z = phase * nsplit + split; raw fp32 partials are atomically added to ws [M_out pixels][Cout]
  float* slab; int* cnt;                     // split-K of conv_igemm8_kernel: per-(tile, split) partial tiles and per-tile arrival counters (the last split to arrive sums and finishes)
  TapTable taps[4];
};

__device__ __forceinline__ bf16x8 to_bf16x8(const F8& f) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (__bf16)f.v[i];
  return r;
}
// peel one bf16 part off f (f <- f - part): repeated P times this yields the P-way split
__device__ __forceinline__ bf16x8 peel_bf16x8(F8& f) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) { r[i] = (__bf16)f.v[i]; f.v[i] -= (float)r[i]; }
  return r;
}
__device__ __forceinline__ bf16x8 zero_bf16x8() {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (__bf16)0.f;
  return r;
}

// DMAQ (bf16, P = 1, no per-sample input scales, Cin % 32 == 0): both operand tiles are staged by LDS-DMA as unpadded 64-byte
// records (row x 32 channels) with the four 16-byte slots XOR-swizzled through the source address (slot = chunk ^ ((row >> 2) & 3):
// conflict-free ds_read_b128), as in conv_halo_kernel: a step then issues 4 DMA pieces per wave instead of 4 loads, 2 fp32 round
// trips and 4 LDS stores per thread, and the im2col address work is two adds and a range check per piece.  This is the kernel of
// the low-resolution layers (4 x 4 ... 16 x 16 grids, and every layer whose halo grid would leave CUs idle: split-K partials).
typedef __attribute__((address_space(3))) void lds_void_g;
template <typename T, int P, bool DMAQ = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  static_assert(!DMAQ || P == 1, "LDS-DMA staging: bf16 operands");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 2 * P;                            // operand tiles per stage: A parts 0..P-1, then B parts 0..P-1
  __bf16* lds = (__bf16*)smem;
  int* row_off = (int*)(smem + (DMAQ ? 12 * 128 * 64 : 2 * NT * TILE * sizeof(__bf16)));   // (DMAQ: three stages of four 8-KB tiles)
  int* row_b = row_off + BM;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, phase = blockIdx.z / a.nsplit, split = blockIdx.z - phase * a.nsplit;
  const TapTable& tt = a.taps[phase];
  const int HWm = a.Hm * a.Wm;
  const T* __restrict__ x = (const T*)a.x;

  // ---- loader bookkeeping: each thread stages rows (lrow, lrow+64), 8 channels at lvec*8 ---------------
  const int lrow = tid >> 2, lvec = tid & 3;
  int lb[2], liy[2], lix[2];
  bool lvalid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + lrow + 64 * i;
    lvalid[i] = m < a.M;
    const int mm = lvalid[i] ? m : 0;
    const int b = mm / HWm, rem = mm - b * HWm;
    const int iy = rem / a.Wm, ix = rem - iy * a.Wm;
    lb[i] = b; liy[i] = iy * a.in_mul; lix[i] = ix * a.in_mul;
    if (lvec == 0) {
      const int oy = iy * a.out_mul + (phase >> 1), ox = ix * a.out_mul + (phase & 1);
      row_off[lrow + 64 * i] = lvalid[i] ? ((b * a.Hout + oy) * a.Wout + ox) : -1;
      row_b[lrow + 64 * i] = b;
    }
  }

  F8 ra[2];
  bf16x8 rb[P][2];

  auto gload = [&](int q) {
    const int tap = q / a.kc_per_tap;
    const int c0 = (q - tap * a.kc_per_tap) * BK + lvec * 8;
    const int dy = tt.dy[tap], dx = tt.dx[tap], wt = tt.wt[tap];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int iy = liy[i] + dy, ix = lix[i] + dx;
      const bool ok = lvalid[i] && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win && c0 < a.Cin;
      if (ok) {
        ra[i] = Feat<T>::load(x + ((size_t)(lb[i] * a.Hin + iy) * a.Win + ix) * a.Cin + c0);
        if (a.pre) {
          const float* ps = a.pre + (size_t)lb[i] * a.pre_stride + c0;
          const f32x4 p0 = *(const f32x4*)ps, p1 = *(const f32x4*)(ps + 4);
          ra[i].v[0] *= p0[0]; ra[i].v[1] *= p0[1]; ra[i].v[2] *= p0[2]; ra[i].v[3] *= p0[3];
          ra[i].v[4] *= p1[0]; ra[i].v[5] *= p1[1]; ra[i].v[6] *= p1[2]; ra[i].v[7] *= p1[3];
        }
      } else {
        ra[i] = f8_zero();
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = n0 + lrow + 64 * i;
      const size_t off = ((size_t)wt * a.N + n) * a.Kpad + c0;
#pragma unroll
      for (int pp = 0; pp < P; ++pp) rb[pp][i] = (n < a.N) ? *(const bf16x8*)(a.w + pp * a.w_part + off) : zero_bf16x8();
    }
  };

  auto sstore = [&](int buf) {
    __bf16* base = lds + buf * NT * TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = (lrow + 64 * i) * LDS_ROW + lvec * 8;
      F8 f = ra[i];
#pragma unroll
      for (int pp = 0; pp < P; ++pp) {
        *(bf16x8*)(base + pp * TILE + o) = peel_bf16x8(f);
        *(bf16x8*)(base + (P + pp) * TILE + o) = rb[pp][i];
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int buf) {
    const __bf16* A = lds + buf * NT * TILE;
    const __bf16* Bt = A + P * TILE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int koff = ks * 16 + (lane >> 5) * 8;
      bf16x8 af[P][2], bf[P][2];
#pragma unroll
      for (int pp = 0; pp < P; ++pp) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          af[pp][mi] = *(const bf16x8*)(A + pp * TILE + (wm * 64 + mi * 32 + (lane & 31)) * LDS_ROW + koff);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          bf[pp][ni] = *(const bf16x8*)(Bt + pp * TILE + (wn * 64 + ni * 32 + (lane & 31)) * LDS_ROW + koff);
      }
      // products a_i * b_j with i + j < P, smallest terms first
#pragma unroll
      for (int sum = P - 1; sum >= 0; --sum)
#pragma unroll
        for (int i = 0; i <= sum; ++i)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][mi], bf[sum - i][ni], acc[mi][ni], 0, 0, 0);
    }
  };

  // ---- main loop: register-staged double buffering, one barrier per K chunk ---------------------------
  const int nq_all = tt.n * a.kc_per_tap;
  const int per = (nq_all + a.nsplit - 1) / a.nsplit;
  const int q0 = split * per, nq = min(q0 + per, nq_all);
  if constexpr (DMAQ) {
    if constexpr (sizeof(T) == 2) {
    constexpr int QT = 128 * 64;                                 // bytes per operand tile; a stage = A then B
    const int widu = __builtin_amdgcn_readfirstlane(wid);
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(2u * (unsigned)(a.B * a.Hin * a.Win * a.Cin)), 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);
    // pieces i = widu, widu + 4 of each tile: rows 16 i .. 16 i + 15, four slots per row
    int abase[2], aiy[2], aix[2];
    unsigned wvo[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = 16 * (widu + 4 * k) + (lane >> 2), ch = ((lane & 3) ^ ((row >> 2) & 3)) * 8;
      const int m = m0 + row;
      const bool ok = m < a.M;
      const int mm = ok ? m : 0;
      const int b = mm / HWm, rem = mm - b * HWm;
      const int iy = (rem / a.Wm) * a.in_mul, ix = (rem % a.Wm) * a.in_mul;
      aiy[k] = ok ? iy : -0x40000000; aix[k] = ix;
      abase[k] = 2 * (((b * a.Hin + iy) * a.Win + ix) * a.Cin + ch);
      const int n = n0 + row;
      wvo[k] = n < a.N ? 2u * (unsigned)(n * a.Kpad + ch) : 0xffffffffu;
    }
    // A stage = TWO K chunks (A, B, A, B: 32 KB); three stages, loads two stages ahead.  These launches run one workgroup of 4 waves
    // per CU (few tiles x split-K), so nothing hides a wave's own LDS / barrier latency: with one chunk per barrier and one step of
    // prefetch a step took ~1 us for 8 MFMAs per wave (a layer worth 12 us took 36-46).  Now 16 MFMAs per barrier, and the loads of the
    // next two stages stay in flight across it: a stage is 8 DMA instructions per wave, `s_waitcnt vmcnt(8)` leaves the youngest
    // stage pending, and a raw s_barrier (no vmcnt(0), which __syncthreads would add while a DMA is pending) publishes the stage
    // that landed and retires the one just read.  A chunk beyond the split's range is issued with out-of-range offsets (zeros).
    constexpr int NST = 3, SB = 4 * QT;                          // stages, bytes per stage
    auto dma = [&](int q, int buf) {                             // chunks q, q + 1 -> stage buf
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool live = q + h < nq;
        const int qq = live ? q + h : q;
        const int tap = qq / a.kc_per_tap, c0 = (qq - tap * a.kc_per_tap) * BK;
        const int dy = tt.dy[tap], dx = tt.dx[tap];
        const int tofs = 2 * ((dy * a.Win + dx) * a.Cin + c0);
        char* S = smem + buf * SB + h * 2 * QT;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const bool ok = live && (unsigned)(aiy[k] + dy) < (unsigned)a.Hin && (unsigned)(aix[k] + dx) < (unsigned)a.Win;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void_g*)(S + (widu + 4 * k) * 1024), 16, ok ? (unsigned)(abase[k] + tofs) : 0xffffffffu, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void_g*)(S + QT + (widu + 4 * k) * 1024), 16, live ? wvo[k] : 0xffffffffu,
                                                   __builtin_amdgcn_readfirstlane(2 * (tt.wt[tap] * a.N * a.Kpad + c0)), 0, 0);
      }
    };
    const int half = lane >> 5, sw = (lane >> 2) & 3;
    const int fa0 = (wm * 64 + (lane & 31)) * 64 + ((half ^ sw) << 4);          // mi = 1: + 32 rows; k-step 1: ^ 32
    const int fb0 = QT + (wn * 64 + (lane & 31)) * 64 + ((half ^ sw) << 4);
    auto computeq = [&](int buf) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const char* S = smem + buf * SB + h * 2 * QT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 af[2], bf[2];
          af[0] = *(const bf16x8*)(S + (fa0 ^ (ks * 32)));
          af[1] = *(const bf16x8*)(S + (fa0 ^ (ks * 32)) + 32 * 64);
          bf[0] = *(const bf16x8*)(S + (fb0 ^ (ks * 32)));
          bf[1] = *(const bf16x8*)(S + (fb0 ^ (ks * 32)) + 32 * 64);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
      }
    };
    const int nstep = (nq - q0 + 1) >> 1;                        // stages of this split
    if (nstep > 0) dma(q0, 0);
    if (nstep > 1) dma(q0 + 2, 1);
    if (nstep > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int st = 0; st < nstep; ++st) {
      const int nxt2 = cur + 2 >= NST ? cur + 2 - NST : cur + 2;
      if (st + 2 < nstep) dma(q0 + 2 * (st + 2), nxt2);
      computeq(cur);
      // stage st + 1 must have landed: everything but the stage issued after it (st + 2, where it exists)
      if (st + 2 < nstep) asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = cur + 1 == NST ? 0 : cur + 1;
    }
    }
  } else {
  if (q0 < nq) {
    gload(q0);
    sstore(0);
  }
  __syncthreads();
  for (int q = q0; q < nq; ++q) {
    const int cur = (q - q0) & 1;
    if (q + 1 < nq) gload(q + 1);
    compute(cur);
    if (q + 1 < nq) sstore(cur ^ 1);
    __syncthreads();
  }
  }

  if (a.nsplit > 1) {                       // split-K: raw partial sums; lcgan finalize kernel applies the epilogue
    if (q0 >= nq) return;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int ro = row_off[row];
          if (ro >= 0 && n < a.Cout) atomicAdd(a.ws + (size_t)ro * a.Cout + n, acc[mi][ni][r]);
        }
      }
    return;
  }

  // ---- epilogue: demod * acc + bias -> act * gain (+ residual) -----------------------------------------
  T* __restrict__ y = (T*)a.y;
  const T* __restrict__ res = (const T*)a.residual;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + (lane & 31);
      const bool nalloc = n < a.Cout, nlog = n < a.N;
      const float bv = (a.bias && nlog) ? a.bias[n] * a.bias_scale : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int ro = row_off[row];
        if (ro < 0 || !nalloc) continue;
        float v = acc[mi][ni][r];
        if (a.post) v *= a.post[(size_t)row_b[row] * a.post_stride + n];   // post is [B][Cout] (alloc width)
        v += bv;
        v = act_fwd(v, a.act) * a.gain;
        const size_t off = (size_t)ro * a.Cout + n;
        if (res) {
          if (a.res_half) {
            int ox, oy, bb;
            if (a.lw >= 0 && a.lh >= 0) { ox = ro & (a.Wout - 1); const int t = ro >> a.lw; oy = t & (a.Hout - 1); bb = t >> a.lh; }
            else { ox = ro % a.Wout; const int t = ro / a.Wout; oy = t % a.Hout; bb = t / a.Hout; }
            v += 0.25f * Feat<T>::ld1(res + ((size_t)(bb * (a.Hout >> 1) + (oy >> 1)) * (a.Wout >> 1) + (ox >> 1)) * a.Cout + n);
          } else {
            v += Feat<T>::ld1(res + off);
          }
        }
        Feat<T>::st1(y + off, v);
      }
    }
}

// Eight-wave form of the LDS-DMA main loop above (bf16, same ConvArgs, same tiles and LDS records): 512 threads, waves 4 (M) x 2 (N),
// each 32 x 64 outputs.  The launches that take this kernel run ONE workgroup per CU (few tiles x split-K), so with 4 waves every SIMD
// held a single wave and nothing overlapped its DMA issue (8 instructions per stage, ~100 cycles each) or its ds_read latency with
// MFMA work: a stage of 16 MFMAs per wave (512 cycles) took ~3000.  With two waves per SIMD one wave's MFMAs run under the other's
// address work, each wave issues half the DMA pieces (one A and one B piece per chunk), and NST stages keep NST - 1 in flight.
template <int NST>
__global__ __launch_bounds__(512) void conv_igemm8_kernel(ConvArgs a) {
  typedef __bf16 T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int QT = 128 * 64, SB = 4 * QT;                    // bytes per operand tile; per stage (two chunks: A, B, A, B)
  int* row_off = (int*)(smem + NST * SB);
  int* row_b = row_off + BM;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, phase = blockIdx.z / a.nsplit, split = blockIdx.z - phase * a.nsplit;
  const TapTable& tt = a.taps[phase];
  const int HWm = a.Hm * a.Wm;
  if (tid < BM) {
    const int m = m0 + tid;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    int b, rem, iy, ix;
    divmod_sh((unsigned)mm, HWm, a.l_hwm, b, rem);
    divmod_sh((unsigned)rem, a.Wm, a.l_wm, iy, ix);
    const int oy = iy * a.out_mul + (phase >> 1), ox = ix * a.out_mul + (phase & 1);
    row_off[tid] = ok ? ((b * a.Hout + oy) * a.Wout + ox) : -1;
    row_b[tid] = b;
  }
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int nq_all = tt.n * a.kc_per_tap;
  const int per = (nq_all + a.nsplit - 1) / a.nsplit;
  const int q0 = split * per, nq = min(q0 + per, nq_all);
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(2u * (unsigned)(a.B * a.Hin * a.Win * a.Cin)), 0x00020000);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);
  // piece widu of each tile: rows 16 widu .. 16 widu + 15, four 16-byte slots per row
  int abase, aiy, aix;
  unsigned wvo;
  {
    const int row = 16 * widu + (lane >> 2), ch = ((lane & 3) ^ ((row >> 2) & 3)) * 8;
    const int m = m0 + row;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    int b, rem, iy, ix;
    divmod_sh((unsigned)mm, HWm, a.l_hwm, b, rem);
    divmod_sh((unsigned)rem, a.Wm, a.l_wm, iy, ix);
    iy *= a.in_mul; ix *= a.in_mul;
    aiy = ok ? iy : -0x40000000; aix = ix;
    abase = 2 * (((b * a.Hin + iy) * a.Win + ix) * a.Cin + ch);
    const int n = n0 + row;
    wvo = n < a.N ? 2u * (unsigned)(n * a.Kpad + ch) : 0xffffffffu;
  }
  auto dma = [&](int q, int buf) {                             // chunks q, q + 1 -> stage buf: 4 DMA instructions per wave
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool live = q + h < nq;
      const int qq = live ? q + h : q;
      int tap, c0;
      divmod_sh((unsigned)qq, a.kc_per_tap, a.l_kc, tap, c0);        // (was a scalar division -- ~25 dependent instructions -- twice per main-loop stage)
      c0 *= BK;
      const int dy = tt.dy[tap], dx = tt.dx[tap];
      const int tofs = 2 * ((dy * a.Win + dx) * a.Cin + c0);
      char* S = smem + buf * SB + h * 2 * QT;
      const bool ok = live && (unsigned)(aiy + dy) < (unsigned)a.Hin && (unsigned)(aix + dx) < (unsigned)a.Win;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void_g*)(S + widu * 1024), 16, ok ? (unsigned)(abase + tofs) : 0xffffffffu, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void_g*)(S + QT + widu * 1024), 16, live ? wvo : 0xffffffffu,
                                               __builtin_amdgcn_readfirstlane(2 * (tt.wt[tap] * a.N * a.Kpad + c0)), 0, 0);
    }
  };
  const int half = lane >> 5, sw = (lane >> 2) & 3;
  const int fa0 = (wm * 32 + (lane & 31)) * 64 + ((half ^ sw) << 4);            // k-step 1: ^ 32
  const int fb0 = QT + (wn * 64 + (lane & 31)) * 64 + ((half ^ sw) << 4);       // ni = 1: + 32 rows
  auto compute = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const char* S = smem + buf * SB + h * 2 * QT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 af = *(const bf16x8*)(S + (fa0 ^ (ks * 32)));
        const bf16x8 b0 = *(const bf16x8*)(S + (fb0 ^ (ks * 32)));
        const bf16x8 b1 = *(const bf16x8*)(S + (fb0 ^ (ks * 32)) + 32 * 64);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc[1], 0, 0, 0);
      }
    }
  };
  // NST stages, NST - 1 in flight: `s_waitcnt vmcnt(4 (NST - 2))` leaves the youngest NST - 2 stages pending (4 instructions each);
  // towards the end fewer are outstanding and the wait is for everything (a stricter wait is always safe)
  const int nstep = (nq - q0 + 1) >> 1;                        // stages of this split
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < nstep) dma(q0 + 2 * i, i);
  if (nstep >= NST - 1) {
    if constexpr (NST == 3) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();                                // (also publishes row_off / row_b: LDS writes are in order per wave, lgkmcnt below)
  int cur = 0;
  for (int st = 0; st < nstep; ++st) {
    const int nxt = cur + NST - 1 >= NST ? cur - 1 : cur + NST - 1;
    if (st + NST - 1 < nstep) dma(q0 + 2 * (st + NST - 1), nxt);
    compute(cur);
    if (st + NST - 1 < nstep) {
      if constexpr (NST == 3) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    cur = cur + 1 == NST ? 0 : cur + 1;
  }

  if (a.nsplit > 1 && !a.slab) {            // split-K through atomics: raw partial sums; the finalize kernel applies the epilogue
    if (q0 >= nq) return;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int ro = row_off[row];
        if (ro >= 0 && n < a.Cout) atomicAdd(a.ws + (size_t)ro * a.Cout + n, acc[ni][r]);
      }
    }
    return;
  }
  if (a.nsplit > 1) {
    // split-K without float atomics or a second launch: every split stores its partial tile (registers as they are, one dword per
    // thread and store: coalesced) in its own slab and bumps the tile's counter; the split that arrives LAST sums the slabs in split
    // order (its own from registers: the result does not depend on the arrival order, unlike atomic accumulation) and runs the
    // epilogue below.  The splits of a tile run on different XCDs, whose L2s are not coherent with each other: the partials move as
    // device-scope relaxed atomic stores / loads (sc1: written through, read past the local L2) ordered around the counter by
    // vmcnt(0) + the workgroup barrier; device-scope FENCES (__threadfence) would write back and invalidate the whole L2 per wave,
    // measured at +55 us per launch.  An empty split (q0 >= nq) contributes zeros.  The counter is left at 0 for the next launch.
    int& s_last = row_b[BM];                                   // (one more int of the dynamic allocation)
    const int tile = (phase * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* slabs = a.slab + (size_t)tile * a.nsplit * (BM * BN);
    float* mine = slabs + (size_t)split * (BM * BN);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __hip_atomic_store(mine + (ni * 16 + r) * 512 + tid, acc[ni][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SLAB_PUBLISH_FENCE();
    __syncthreads();
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(a.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == a.nsplit - 1;
      if (last) __hip_atomic_store(a.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = last;
    }
    __syncthreads();
    SLAB_CONSUME_FENCE();
    if (!s_last) return;
    f32x16 tot[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[j][r] = 0.f;
    for (int sp = 0; sp < a.nsplit; ++sp) {
      if (sp == split) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) tot[j][r] += acc[j][r];
      } else {
        const float* other = slabs + (size_t)sp * (BM * BN);
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = __hip_atomic_load(other + j * 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 32; ++j) tot[j >> 4][j & 15] += v[j];
      }
    }
    acc[0] = tot[0]; acc[1] = tot[1];
  }
  T* __restrict__ y = (T*)a.y;
  const T* __restrict__ res = (const T*)a.residual;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wn * 64 + ni * 32 + (lane & 31);
    const bool nalloc = n < a.Cout, nlog = n < a.N;
    const float bv = (a.bias && nlog) ? a.bias[n] * a.bias_scale : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int ro = row_off[row];
      if (ro < 0 || !nalloc) continue;
      float v = acc[ni][r];
      if (a.post) v *= a.post[(size_t)row_b[row] * a.post_stride + n];
      v += bv;
      v = act_fwd(v, a.act) * a.gain;
      const size_t off = (size_t)ro * a.Cout + n;
      if (res) {
        if (a.res_half) {
          int ox, oy, bb;
          if (a.lw >= 0 && a.lh >= 0) { ox = ro & (a.Wout - 1); const int t = ro >> a.lw; oy = t & (a.Hout - 1); bb = t >> a.lh; }
          else { ox = ro % a.Wout; const int t = ro / a.Wout; oy = t % a.Hout; bb = t / a.Hout; }
          v += 0.25f * Feat<T>::ld1(res + ((size_t)(bb * (a.Hout >> 1) + (oy >> 1)) * (a.Wout >> 1) + (ox >> 1)) * a.Cout + n);
        } else {
          v += Feat<T>::ld1(res + off);
        }
      }
      Feat<T>::st1(y + off, v);
    }
  }
}

// =========================================================================================================
// halo-tile kernel (bf16 fast path): the input patch of a 16x16 output tile (plus its tap halo) is staged in LDS ONCE per
// 32-channel chunk and reused by every tap, so the A-side staging cost (global loads, style multiply, conversions,
// ds_write) is amortised over up to 9 taps instead of being repeated per tap as in conv_igemm_kernel.  Per tap only the
// 128 x 32 weight tile moves.  512 threads = 8 waves (4 x 2), each wave 64 x 64 outputs; one barrier per tap.
// Same tap-table geometry as conv_igemm_kernel: forward stride 1/2 (IN_MUL), data gradient stride 1, 4-phase transposed conv.
// =========================================================================================================
constexpr int HT = 16;                        // tile edge: 16 x 16 = 256 output positions of ONE sample
constexpr int HROW = 40;                      // bf16 per staged pixel row (32 channels + 8 pad = 80 B)

struct HaloArgs {
  const __bf16* x; const __bf16* w; __bf16* y;
  const float* pre; const float* post; const float* bias; const __bf16* residual; int res_half;
  __bf16* pool_out;                          // see ConvArgs (written by the EPI == 1 epilogue)
  unsigned* mask_out;                        // see ConvArgs (written by the transposed-accumulator epilogue)
  long long w_bstride;                       // elements between the weights of consecutive samples (0 = shared): per-sample modulated weight copies
  const __bf16* xs; float* gs;               // see ConvArgs
  int B, Hin, Win, Cin, Hout, Wout, Cout, Hm, Wm, N, Kpad, kc_per_tap;
  int out_mul, tiles_x, tiles_y;
  float bias_scale, gain; int act;
  TapTable taps[4];
  int hy0[4], hx0[4], hh[4], hw[4];          // per phase: halo origin (min dy, min dx) and extent in input pixels
  int nblocks, nb_group;                     // channel blocks of 128; how many of them run together per tile (see the kernel's workgroup order)
  int nph_x;                                 // 4: the sub-pixel phases of a transposed convolution are folded into grid.x (grid.z = 1); else 1
  int halo_elems;                            // LDS elements reserved for the halo (max over phases)
  int dbg;                                   // option 3, bit 8: linear instead of XCD-contiguous tile order
  // split of the input-channel range over blockIdx.y (stride-1 LDS-DMA structure only; launches that would leave most CUs idle): every
  // split runs its share of the 32-channel chunks; partial tiles and the finish as in conv_igemm8_kernel (slab + arrival counter)
  int nsplit; float* slab; int* cnt;
  // workgroup -> tile decode without integer divisions: every wave of every workgroup did ten of them (each ~25 dependent scalar / v_rcp
  // instructions) before its first load -- 2.4 us of a workgroup's 4.2-us prologue (scripts/halo_life.py).  Host-side: the tile count and,
  // per divisor of the decode, its log2 when it is a power of two (else -1: that step divides)
  // (one contiguous block, copies of nblocks / nb_group / nph_x / tiles_x / tiles_y / dbg included, read by ONE wide scalar load at kernel
  //  entry: as scattered fields the decode took a dozen dependent s_load_dword round trips)
  struct Decode { int ntile, sh_per, sh_grp, sh_nph, sh_tx, sh_ty, nblocks, nb_group, nph_x, tiles_x, tiles_y, dbg; } dec;
};

#ifdef HALO_STAMPS
// diagnostic build (scripts/halo_stamps.py; never the shipped library): per wave of the first 2048 workgroups, cycle sums of the four
// segments of a main-loop step -- [barrier exit -> fragments landed] [MFMA issue] [tile store -> barrier arrival] [barrier wait]
__device__ unsigned long long g_halo_stamps[2048 * 8 * 5];
__device__ unsigned long long g_halo_life[16384 * 9];        // per workgroup (wave 0): s_memrealtime at kernel entry, main-loop start, main-loop end, stores acknowledged; HW_ID; first DMA issued, first stage landed, output tile in LDS, stores issued
__device__ unsigned long long g_halo_clock[2048 * 8 * 2];     // per wave: (s_memtime, s_memrealtime [100 MHz]) deltas over the main loop -> the clock the chip held
#define STAMP_RT(var)                                                                                               \
  __builtin_amdgcn_sched_barrier(0);                                                                                \
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");                                   \
  __builtin_amdgcn_sched_barrier(0);
#define STAMP(var)                                                                                                  \
  __builtin_amdgcn_sched_barrier(0);                                                                                \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");                                       \
  __builtin_amdgcn_sched_barrier(0);
#else
#define STAMP(var)
#endif

// EPI selects the epilogue at compile time: 0 = plain, 1 = + residual, 2 = + 0.25 * half-resolution residual, 3 = style-gradient
// reduction (gs += sum xs * acc).  As run-time branches the last two cost every launch 5-7 % on the short-K top layer (measured
// with scripts/ab_raw.py against the builds that preceded them).
// DMA (stride-1 geometries without per-sample input scales, Cin % 32 == 0): both operand tiles go global -> LDS by
// `buffer_load ... lds` (no staging registers, no ds_write, no per-step vector-memory wait before an LDS store).  An LDS-DMA
// instruction writes lane l's 16 bytes at base + 16 l, so the images are unpadded 64-byte records (pixel / weight row x 32
// channels) whose four 16-byte slots are XOR-swizzled through the SOURCE address: slot = chunk ^ ((x >> 2) & 3), which makes the
// fixed 16-lane groups of ds_read_b128 conflict-free for every tap shift (halo row pitch 20 pixels, a multiple of 4).
constexpr int DMA_HP = 20, DMA_HROWS = 18;                       // halo image: <= 18 rows of 20 pixels
constexpr int DMA_HBUF = (DMA_HROWS * DMA_HP * 64 + 1023) & ~1023; // bytes per halo image, whole 1-KB DMA pieces (two images: the next chunk lands during this one's taps)
constexpr int DMA_BBUF = 128 * 64;                               // bytes per weight tile (two)
typedef __attribute__((address_space(3))) void lds_void;

template <int IN_MUL, bool M16, int EPI_, int DMA = 0, bool MOD = false>
__global__ __launch_bounds__(512, (IN_MUL == 1 || DMA == 4) ? 4 : 2) void conv_halo_kernel(HaloArgs a) {   // (4 waves per SIMD = two workgroups per CU: at most 128 VGPRs)
  // EPI_ == 4: the plain epilogue (0) that also leaves the activation's sign mask (a.mask_out).  Its own instantiation: as a run-time
  // branch of the plain epilogue the four mask words cost EVERY plain launch 6 VGPRs and ~40 spilled SGPRs, and the modulated
  // instantiations (123 -> 129 VGPRs) their second workgroup per CU (+30 % on the 512-channel generator layers, measured).
  constexpr int EPI = EPI_ == 4 ? 0 : EPI_;
  constexpr bool MK = EPI_ == 4;
  static_assert(DMA == 0 || (IN_MUL == 1 && DMA <= 2 && (!M16 || DMA == 2)) || (IN_MUL == 2 && (DMA == 3 || DMA == 4) && !M16),
                "LDS-DMA staging: stride-1 geometries (1 / 2 taps per barrier) or the stride-2 forward structure (DMA == 3)");
  static_assert(!MOD || DMA != 0, "MOD: the swizzled-record structure with the halo staged through registers (per-sample input scales)");
  constexpr bool SR = EPI == 3;
  // TR: the 32x32 MFMAs run with their operands swapped (weights as the row operand), so a lane's 16 accumulator registers are
  // 4 groups of 4 CONSECUTIVE CHANNELS of one pixel instead of 16 pixels of one channel: the epilogue moves 8-byte groups
  // (16 ds_write_b64 per lane) instead of 64 two-byte elements.  The style-gradient epilogue keeps channel-per-lane (its column sums).
  constexpr bool TR = !M16 && !SR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NI = IN_MUL == 1 ? 3 : 9;     // halo (pixel, 8-channel vector) items per thread: ceil(hh*hw*4 / 512)
  __bf16* halo = (__bf16*)smem;
  __bf16* Bt = halo + a.halo_elems;           // 2 x [128][HROW]
  constexpr bool PAIR = IN_MUL == 2 && !M16;   // two taps per step (below): where one workgroup fits per CU
  float* psc = (float*)(Bt + (PAIR ? 4 : 2) * TILE);       // [Kpad] style scales of this workgroup's sample (modulated convs; zero beyond Cin)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
#ifdef HALO_STAMPS
  unsigned long long life0 = 0, life1 = 0, life2 = 0, life3 = 0, lifeA = 0, lifeB = 0, lifeC = 0, lifeD = 0;
  STAMP_RT(life0)
#endif
  // the 4 sub-pixel phases of a transposed conv have 1/2/2/4 taps: dispatch the long ones first (shorter tail)
  int phase = gridDim.z - 1 - blockIdx.z;
  // Workgroup order (grid.x = tiles x channel blocks).  Workgroups are dealt round-robin over the 8 XCDs, so every XCD gets a
  // contiguous run of tiles (neighbouring tiles share halo columns and rows in that XCD's L2: -1.3 % on the conv launches of an
  // iteration).  Inside an XCD the channel blocks of one tile run TOGETHER in groups of a.nb_group blocks (as many as keep their
  // weights resident in the XCD's 4 MB L2), so the tile's input is fetched from HBM once per group instead of once per block.
  // The 4 phases of a transposed convolution read the SAME 17 x 17 input halo.  As grid.z planes (all tiles of one phase, then the next)
  // every phase fetches the input over the fabric again: 4.1-5.5x the input bytes per launch (rocprofv3 FETCH_SIZE per dispatch,
  // scripts/micro_s2_pmc.py).  Option 24 folds the phases into the 1-D order (a.nph_x = 4: the phases of a tile neighbours in one XCD's
  // queue, longest first).  MEASURED: the reads fall 3.6x (1 093 -> 299 MB at 128 x 128 -> 256 x 256) and the launch takes the same time
  // there, 25 % LONGER on the 512-channel layers: these launches are not bound by what crosses the fabric but by the per-workgroup fixed
  // cost paid four times per tile (a 1-tap phase is 8 steps of main loop); eight workgroups asking for the same lines at the same moment
  // costs more than the Infinity Cache hits it saves.  Off by default.
  const HaloArgs::Decode dc = a.dec;
  {
    // the scalars the prologue needs, requested together at kernel entry: left to the compiler they were loaded one by one where each is
    // first used -- eight dependent scalar-load round trips between the decode and the first DMA
    const void *p0 = a.x, *p1 = a.w, *p2 = a.bias, *p3 = a.post;
    const long long q0 = a.w_bstride;
    const int i0 = a.B, i1 = a.Hin, i2 = a.Win, i3 = a.Cin, i4 = a.Cout, i5 = a.N, i6 = a.Kpad, i7 = a.kc_per_tap, i8 = a.nsplit, i9 = a.halo_elems;
    asm volatile("" ::"s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(q0), "s"(i0), "s"(i1), "s"(i2), "s"(i3), "s"(i4), "s"(i5), "s"(i6), "s"(i7), "s"(i8), "s"(i9));
  }
  const int nph = dc.nph_x, ntile = dc.ntile;
  int tile, nb;
  if (!(dc.dbg & 8) && (ntile & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, grp = dc.nb_group * nph, per = (ntile >> 3) * grp;
    int nbo, rem, tq, r2, bq, ph;
    divmod_sh((unsigned)slot, per, dc.sh_per, nbo, rem);
    divmod_sh((unsigned)rem, grp, dc.sh_grp, tq, r2);
    tile = xcd * (ntile >> 3) + tq;
    divmod_sh((unsigned)r2, nph, dc.sh_nph, bq, ph);
    nb = nbo * dc.nb_group + bq;
    if (nph > 1) phase = nph - 1 - ph;
  } else {
    tile = blockIdx.x % ntile;
    const int t2 = blockIdx.x / ntile;
    nb = t2 % dc.nblocks;
    if (nph > 1) phase = nph - 1 - t2 / dc.nblocks;
  }
  const int n0 = nb * BN;
  int tx, ty, b, trow;
  divmod_sh((unsigned)tile, dc.tiles_x, dc.sh_tx, trow, tx);
  divmod_sh((unsigned)trow, dc.tiles_y, dc.sh_ty, b, ty);
  // channel-range split: this workgroup runs chunks [cb, cb + nchunks) -- as a shift of both operand bases, so the main loop below
  // counts from 0 as ever
  constexpr bool CAN_SPLIT = IN_MUL == 1 && (DMA == 1 || DMA == 2) && !M16 && EPI != 3 && !MOD;
  int coff = 0, nchunks_l = a.kc_per_tap;
  if (CAN_SPLIT && a.nsplit > 1) {
    const int per = (a.kc_per_tap + a.nsplit - 1) / a.nsplit, cb = blockIdx.y * per;
    nchunks_l = max(0, min(per, a.kc_per_tap - cb));
    coff = cb * BK;
  }
  const TapTable& tt = a.taps[phase];
  const int hy0 = a.hy0[phase], hx0 = a.hx0[phase], hh = a.hh[phase], hw = a.hw[phase];
  // the epilogue's per-channel constants, fetched now (thread t < 128: channel n0 + t) instead of in front of the epilogue's barrier
  float ep_bias = 0.f, ep_post = 1.f;
  if (TR && tid < BN) {
    const int n = n0 + tid;
    if (a.bias && n < a.N) ep_bias = a.bias[n];                   // (the RAW value: multiplied by bias_scale here, the load was WAITED for here -- a memory round trip in every workgroup's prologue; the scale is applied where the value is staged for the epilogue)
    if (a.post && n < a.Cout) ep_post = a.post[(size_t)b * a.Cout + n];
  }
  // LDS pitch of a halo ROW in elements: a multiple of 256 bytes.  A wave's 32 fragment rows are 2 image rows of 16 pixels; with the
  // second row a whole number of bank rows below the first, the fixed lane groups of ds_read_b128 see the same conflict-free
  // pattern as 32 consecutive pixels (at the natural pitch hw * 80 B every A read was a 2-way bank conflict)
  const int rp = (hw * HROW + 127) & ~127;
  const int gy0 = ty * HT * IN_MUL + hy0, gx0 = tx * HT * IN_MUL + hx0;     // input pixel of halo (0,0)

  // ---- halo items of this thread (fixed for the whole K loop; only the channel chunk moves) -------------------------
  const int hvec = tid & 3;
  // (global loads are raw buffer loads: resource in scalar registers, a 32-bit BYTE offset per lane, the moving part -- channel
  //  chunk, tap -- as the scalar offset; no 64-bit address lives in vector registers and a lane outside the image / the weight
  //  rows reads zeros by the range check instead of a branch.  Tensors are < 2^31 elements, checked by the caller)
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + coff), 0, (int)(2u * (unsigned)(a.B * a.Hin * a.Win * a.Cin - coff)), 0x00020000);
  // (w_bstride != 0: this sample's own copy of the weights, the per-sample input scales already folded in)
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.w + (size_t)b * (size_t)a.w_bstride + coff), 0, 0x7fffffff, 0x00020000);
  unsigned goff[NI];
  int loff[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int hp = (tid >> 2) + k * 128;
    loff[k] = -1; goff[k] = 0xffffffffu;
    if (hp < hh * hw) {
      const int hy = hp / hw, hx = hp - hy * hw;
      const int gy = gy0 + hy, gx = gx0 + hx;
      loff[k] = hy * rp + hx * HROW + hvec * 8;
      if ((unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win) goff[k] = 2u * (unsigned)(((b * a.Hin + gy) * a.Win + gx) * a.Cin + hvec * 8);
    }
  }
  bf16x8 hreg[NI];
  // The style multiply happens when the halo goes to LDS, not when it is loaded: multiplying right behind the load made the
  // prefetch of the next chunk synchronous (a wait for HBM at every chunk start; modulated layers ran 10 % behind plain ones).
  // The sample's scales sit in LDS behind the weight buffers (`psc`, filled below) and are read back at store time: holding them
  // in registers across the taps costs the second workgroup per CU (148 VGPRs).
  int h_c0 = 0;
  auto halo_load = [&](int c0) {
    h_c0 = c0;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const unsigned go = c0 + hvec * 8 < a.Cin ? goff[k] : 0xffffffffu;       // out of range -> the buffer load returns zeros
      hreg[k] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xres, go, __builtin_amdgcn_readfirstlane(c0 * 2), 0));
    }
  };
  auto halo_store = [&]() {
    if (a.pre) {
      const f32x4 s0 = *(const f32x4*)(psc + h_c0 + hvec * 8), s1 = *(const f32x4*)(psc + h_c0 + hvec * 8 + 4);
#pragma unroll
      for (int k = 0; k < NI; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) hreg[k][j] = (__bf16)((float)hreg[k][j] * (j < 4 ? s0[j] : s1[j - 4]));
    }
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (loff[k] >= 0) *(bf16x8*)(halo + loff[k]) = hreg[k];
  };

  // ---- weight tiles: 128 rows x 4 vectors = 512 items per tap, one per thread ------------------------------------------------
  const int brow = tid >> 2;
  const int ntaps = tt.n, nchunks = nchunks_l;
  const unsigned wrow = 2u * (unsigned)((n0 + brow) * a.Kpad + hvec * 8);
  const bool bvalid = n0 + brow < a.N;

  // ---- per-lane fragment bases ---------------------------------------------------------------------------------------
  // M16 == false: v_mfma_f32_32x32x16_bf16, wave tile 2 x 2 (lane row = lane & 31, k half = lane >> 5, two k-steps per chunk)
  // M16 == true : v_mfma_f32_16x16x32_bf16, wave tile 4 x 4 (lane row = lane & 15, k quarter = lane >> 4, one k-step per chunk);
  //               same LDS traffic and cycles per FLOP, but the chip holds a higher clock on this shape (guide: DVFS item 7)
  constexpr int NM = M16 ? 4 : 2, RS = M16 ? 16 : 32;
  const int lrow = M16 ? (lane & 15) : (lane & 31), lk = (M16 ? (lane >> 4) : (lane >> 5)) * 8;
  int abase[NM];
#pragma unroll
  for (int mi = 0; mi < NM; ++mi) {
    const int r = wm * 64 + mi * RS + lrow;
    abase[mi] = ((r >> 4) * IN_MUL) * rp + (r & 15) * IN_MUL * HROW + lk;
  }
  const int bbase = (wn * 64 + lrow) * HROW + lk;

  f32x16 acc[2][2];
  f32x4 acc16[4][4];
  if (M16) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  if constexpr (IN_MUL == 2 && (DMA == 3 || DMA == 4)) {
    // ---- stride-2 forward, 3 x 3: everything by LDS-DMA, ONE barrier per 16-channel half-chunk (36 MFMAs per wave) --------------
    // The 33 x 33 input patch of a 16 x 16 output tile is kept as its four (row parity, column parity) PLANES, so the stride-2
    // gather of a tap is a unit-stride walk in one plane: tap (ky, kx) reads plane (ky & 1, kx & 1) at (y + (ky >> 1), x + (kx >> 1)).
    // Records are 32 bytes (16 channels), plane rows 20 records; the two 16-byte slots of record (r, c) hold channel half
    // slot ^ ((c >> 2) & 1): with the 640-byte row pitch the fixed lane groups of ds_read_b128 are conflict-free for every tap.
    // A half-chunk stage = the four planes (42 KB) + the weight tiles of ALL nine taps (9 x 128 rows x 32 B = 36 KB); two stages
    // (156 KB: one workgroup per CU, as the 33 x 33 patch always forced) let the next half-chunk land while this one's nine taps
    // run without a barrier between them.
    constexpr int PP = 20;                                       // plane row pitch in records
    constexpr int P_OFF[4] = {0, 17 * PP, 2 * 17 * PP, 2 * 17 * PP + 16 * PP};   // record index of planes (0,0) (0,1) (1,0) (1,1)
    constexpr int NREC = 2 * 17 * PP + 2 * 16 * PP;              // 1 320 records
    constexpr int HPIECES = (NREC + 31) / 32;                    // 42 one-KB pieces
    constexpr int S2_H = HPIECES * 1024, S2_B = 9 * 4096;        // bytes of the planes / of the nine weight tiles
    constexpr int S2_STAGE = S2_H + S2_B;
    const int widu = __builtin_amdgcn_readfirstlane(wid);
    const int nh = 2 * nchunks;                                  // 16-channel half-chunks
    unsigned hvo[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int L = 32 * (widu + 8 * k) + (lane >> 1);           // record of this lane in piece widu + 8 k
      const int pl = L < P_OFF[1] ? 0 : L < P_OFF[2] ? 1 : L < P_OFF[3] ? 2 : 3;
      const int rc = L - (pl == 0 ? P_OFF[0] : pl == 1 ? P_OFF[1] : pl == 2 ? P_OFF[2] : P_OFF[3]);
      const int r = rc / PP, cc = rc - r * PP, pr = pl >> 1, pc = pl & 1;
      const int ch = (lane & 1) ^ ((cc >> 2) & 1);
      const int gy = gy0 + 2 * r + pr, gx = gx0 + 2 * cc + pc;
      const bool ok = L < NREC && r < 17 - pr && cc < 17 - pc && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
      hvo[k] = ok ? 2u * (unsigned)(((b * a.Hin + gy) * a.Win + gx) * a.Cin + ch * 8) : 0xffffffffu;
    }
    const int wrow = 32 * (widu & 3) + (lane >> 1);              // weight row of this lane: piece i = widu + 8 k is (tap i / 4, rows 32 (i % 4) ..)
    const unsigned wvo = n0 + wrow < a.N ? 2u * (unsigned)((n0 + wrow) * a.Kpad + (((lane & 1) ^ ((wrow >> 3) & 1)) * 8)) : 0xffffffffu;
    // MOD: per-sample input scales are applied in place by the lanes that issued the pieces (see the stride-1 structure)
    float* dpsc = (float*)(smem + (DMA == 4 ? 1 : 2) * S2_STAGE);   // [Cin]
    int hch6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int L = 32 * (widu + 8 * k) + (lane >> 1);
      const int pl = L < P_OFF[1] ? 0 : L < P_OFF[2] ? 1 : L < P_OFF[3] ? 2 : 3;
      const int rc = L - (pl == 0 ? P_OFF[0] : pl == 1 ? P_OFF[1] : pl == 2 ? P_OFF[2] : P_OFF[3]);
      hch6[k] = ((lane & 1) ^ (((rc % PP) >> 2) & 1)) * 8;
    }
    auto scale_stage = [&](int h, int buf) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      char* S = smem + buf * S2_STAGE;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        if (widu + 8 * k >= HPIECES) continue;
        bf16x8* ptr = (bf16x8*)(S + (widu + 8 * k) * 1024 + lane * 16);
        const f32x4 s0 = *(const f32x4*)(dpsc + h * 16 + hch6[k]), s1 = *(const f32x4*)(dpsc + h * 16 + hch6[k] + 4);
        bf16x8 v = *ptr;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)((float)v[j] * (j < 4 ? s0[j] : s1[j - 4]));
        *ptr = v;
      }
    };
    if constexpr (MOD) {
      for (int i = tid; i < a.Cin; i += 512) dpsc[i] = a.pre[(size_t)b * a.Cin + i];
      __syncthreads();
    }
    auto dma_stage = [&](int h, int buf) {
      char* S = smem + buf * S2_STAGE;
      const int cofs = __builtin_amdgcn_readfirstlane(h * 32);   // byte offset of the half-chunk's first channel
#if defined(HALO_EXP) && (HALO_EXP == 4 || HALO_EXP == 5)
      if (h == 0)
#endif
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (widu + 8 * k < HPIECES)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(S + (widu + 8 * k) * 1024), 16, hvo[k], cofs, 0, 0);
#if defined(HALO_EXP) && (HALO_EXP == 3 || HALO_EXP == 5)
      if (h == 0)
#endif
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int i = widu + 8 * k;
        if (i < 36)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void*)(S + S2_H + i * 1024), 16, wvo,
                                                   __builtin_amdgcn_readfirstlane(2 * (tt.wt[i >> 2] * a.N * a.Kpad) + h * 32), 0, 0);
      }
    };
    // fragment byte addresses: A = plane (immediate) + row part + one of two column parts ; B = tap (immediate) + row part
    const int half = lane >> 5, xl = lane & 15, yl = wm * 4 + ((lane & 31) >> 4);
    int acol[2];
#pragma unroll
    for (int sft = 0; sft < 2; ++sft) acol[sft] = yl * (PP * 32) + (xl + sft) * 32 + ((half ^ (((xl + sft) >> 2) & 1)) << 4);
    const int brl = wn * 64 + (lane & 31);
    const int baddr = S2_H + brl * 32 + ((half ^ ((brl >> 3) & 1)) << 4);
    dma_stage(0, 0);
    if constexpr (MOD) scale_stage(0, 0);
    __syncthreads();
    auto half_chunk = [&](auto bufc) {
      constexpr int buf = decltype(bufc)::value;
      const char* S = smem + buf * S2_STAGE;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        // the stride-2 forward tap table is row-major over (ky, kx) (see conv_tap_tables): tap t = 3 ky + kx
        constexpr int dummy = 0; (void)dummy;
        const int ky = t / 3, kx = t - 3 * (t / 3);
        const int pbase = P_OFF[(ky & 1) * 2 + (kx & 1)] * 32 + (ky >> 1) * (PP * 32);
        const int aa = acol[kx >> 1] + pbase;
        bf16x8 af[2], bf[2];
        af[0] = *(const bf16x8*)(S + aa);
        af[1] = *(const bf16x8*)(S + aa + 2 * PP * 32);
        bf[0] = *(const bf16x8*)(S + baddr + t * 4096);
        bf[1] = *(const bf16x8*)(S + baddr + t * 4096 + 32 * 32);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
      }
    };
    if constexpr (DMA == 4) {
      // ONE stage per workgroup (78 KB): two workgroups share a CU and overlap each other's load and compute phases instead of a
      // workgroup double-buffering its own
      for (int h = 0; h < nh; ++h) {
        if (h > 0) {
          dma_stage(h, 0);
          if constexpr (MOD) scale_stage(h, 0);
          __syncthreads();
        }
        half_chunk(std::integral_constant<int, 0>{});
        __syncthreads();
      }
    } else {
    for (int h = 0; h < nh; h += 2) {
      if (h + 1 < nh) dma_stage(h + 1, 1);
      half_chunk(std::integral_constant<int, 0>{});
      if constexpr (MOD) { if (h + 1 < nh) scale_stage(h + 1, 1); }
      __syncthreads();
      if (h + 1 < nh) {
        if (h + 2 < nh) dma_stage(h + 2, 0);
        half_chunk(std::integral_constant<int, 1>{});
        if constexpr (MOD) { if (h + 2 < nh) scale_stage(h + 2, 0); }
        __syncthreads();
      }
    }
    }
  } else if constexpr (DMA != 0) {
    char* Hb = smem;                                             // 2 halo images
    char* Bb = smem + 2 * DMA_HBUF;                              // 2 (x TP) weight tiles
    const int dslot = lane & 3, widu = __builtin_amdgcn_readfirstlane(wid);
    // halo: wave-instruction i = wid + 8 k moves pixels 16 i .. 16 i + 15 (linear over [hh][20]) x 4 slots
    unsigned hvo[3];
    bool hdo[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = widu + 8 * k, pi = 16 * i + (lane >> 2);
      const int hy = pi / DMA_HP, hx = pi - hy * DMA_HP;
      const int ch = dslot ^ ((hx >> 2) & 3);
      const int gy = gy0 + hy, gx = gx0 + hx;
      hdo[k] = 16 * i < hh * DMA_HP;
      const bool ok = hy < hh && hx < hw && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
      hvo[k] = ok ? 2u * (unsigned)(((b * a.Hin + gy) * a.Win + gx) * a.Cin + ch * 8) : 0xffffffffu;
    }
    // MOD (per-sample input scales: modulated convolutions and their data gradients): the halo goes by LDS-DMA like every other
    // one and each lane then multiplies ITS OWN 16-byte pieces in place (ds_read -> 8 multiplies -> ds_write) in the last step of the
    // previous chunk, right before that step's barrier: a DMA'd piece is visible to the wave that issued it after its vmcnt(0),
    // which the barrier's wait would have taken anyway, so no staging registers live across the taps (the register-staged version
    // of this held 12 VGPRs per chunk: 133 VGPRs at two taps per barrier).
    float* dpsc = (float*)(smem + 2 * DMA_HBUF + 2 * DMA * DMA_BBUF);        // [Cin] scales of this workgroup's sample
    int hch[3];                                                  // this lane's channel offset within a chunk, per piece
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int pi = 16 * (widu + 8 * k) + (lane >> 2);
      hch[k] = (dslot ^ (((pi % DMA_HP) >> 2) & 3)) * 8;
    }
    int h_c0 = 0;
    auto dma_halo_piece = [&](int c0, int buf, int k) {
      h_c0 = c0;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(Hb + buf * DMA_HBUF + (widu + 8 * k) * 1024), 16, hvo[k],
                                               __builtin_amdgcn_readfirstlane(c0 * 2), 0, 0);
    };
    auto dma_halo = [&](int c0, int buf) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (hdo[k]) dma_halo_piece(c0, buf, k);
    };
    auto scale_inplace = [&](int buf, bool wait) {
      if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's pieces have landed (no wait needed one barrier after their issue)
      // all nine LDS reads first (a piece that does not exist reads inside the allocation and is not written back): one LDS round
      // trip per chunk instead of three serialised ones
      bf16x8 v[3];
      f32x4 s0[3], s1[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        v[k] = *(const bf16x8*)(Hb + buf * DMA_HBUF + (hdo[k] ? (widu + 8 * k) * 1024 : 0) + lane * 16);
        s0[k] = *(const f32x4*)(dpsc + h_c0 + hch[k]); s1[k] = *(const f32x4*)(dpsc + h_c0 + hch[k] + 4);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[k][j] = (__bf16)((float)v[k][j] * (j < 4 ? s0[k][j] : s1[k][j - 4]));
      }
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (hdo[k]) *(bf16x8*)(Hb + buf * DMA_HBUF + (widu + 8 * k) * 1024 + lane * 16) = v[k];
    };
    // weights: wave `wid` moves rows 16 wid .. 16 wid + 15 x 4 slots
    const int drow = 16 * widu + (lane >> 2);
    const unsigned wvo = n0 + drow < a.N ? 2u * (unsigned)((n0 + drow) * a.Kpad + (dslot ^ ((drow >> 2) & 3)) * 8) : 0xffffffffu;
    // TP taps per step (one barrier per step): DMA == 2 moves two weight tiles per step and runs the second tap's fragment reads
    // under the first tap's MFMAs (16 MFMAs per wave between barriers)
    constexpr int TP = DMA;
    const int ngroups = (ntaps + TP - 1) / TP, total = ngroups * nchunks;
    // SPREAD (option 3, bit 16; OFF: measured 1.5-3 % SLOWER than draining everything at every barrier, ab_multi "3=16" vs ""): the halo
    // image of the next chunk issued as pieces BEHIND the step's weight tiles -- two in step 0, the third in step 1 -- with the step
    // barrier waiting on a COUNTED vmcnt that leaves exactly those newest pieces in flight.  The idea came from builds that skip the halo
    // fetch (-DHALO_EXP=4) running 16-25 % faster; the in-loop stamps (scripts/halo_stamps.py) then showed why it does not pay: a step is
    // ~2 300 cycles for 2 048 cycles of MFMA work per SIMD, the vmcnt wait is ~180 of them and the barrier wait ~450 with or without the
    // spread issue -- the loop is matrix-pipe-bound at the clock the chip holds under this load, and the EXP builds ran faster because
    // stale LDS operands draw less power, not because the waits went away.
    const bool spread = ngroups >= 4 && (a.dbg & 16);
    auto dma_b = [&](int cc, int g, int buf) {
#pragma unroll
      for (int j = 0; j < TP; ++j) {
        const int t = g * TP + j;
        if (t < ntaps)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void*)(Bb + (buf * TP + j) * DMA_BBUF + widu * 1024), 16, wvo,
                                                   __builtin_amdgcn_readfirstlane(2 * (tt.wt[t] * a.N * a.Kpad + cc * BK)), 0, 0);
      }
    };
    // fragment byte addresses: A = row part + column part of the tap's x shift (0..2) ; k-step 1 is the same address ^ 32
    const int half = lane >> 5, txl = lane & 15, halfs = half << 4;
    const int rowofs = (wm * 4 + ((lane & 31) >> 4)) * (DMA_HP * 64);          // mi = 1: + 2 image rows
    const int brow_l = wn * 64 + (lane & 31);
    const int baddr0 = 2 * DMA_HBUF + brow_l * 64 + ((half ^ ((brow_l >> 2) & 3)) << 4);   // ni = 1: + 32 rows (same swizzle)
    const int baddr1 = baddr0 ^ 32;

    int lc = 0, lg = 0;
    auto advance = [&]() { if (++lg == ngroups) { lg = 0; ++lc; } };
    dma_halo(0, 0);
    dma_b(0, 0, 0);
    advance();
#ifdef HALO_STAMPS
    STAMP_RT(lifeA)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP_RT(lifeB)
#endif
    if constexpr (MOD) {                                         // (the sample's scales are fetched while the first tiles are in flight)
      for (int i = tid; i < a.Cin; i += 512) dpsc[i] = a.pre[(size_t)b * a.Cin + i];
      __syncthreads();
      scale_inplace(0, true);
    }
    __syncthreads();
    int c = 0, g = 0;
    bf16x8 af[2][2], bf[2][2];
    auto frag_reads = [&](int t, int tilebase, int ks) {         // k-step ks of tap t: 2 A + 2 B fragments
      const int hxl = txl + tt.dx[t] - hx0;                      // this lane's halo column under the tap: record hxl, slot (chunk ^ swizzle)
      const int a0 = (rowofs + (c & 1) * DMA_HBUF + (tt.dy[t] - hy0) * (DMA_HP * 64) + ((hxl << 6) | (halfs ^ ((hxl << 2) & 0x30)))) ^ (ks * 32);
      af[ks][0] = *(const bf16x8*)(smem + a0);
#if defined(HALO_EXP) && HALO_EXP >= 1                           // timing experiment only (wrong results): how much of a step is LDS read traffic
      af[ks][1] = af[ks][0];
#else
      af[ks][1] = *(const bf16x8*)(smem + a0 + 2 * DMA_HP * 64);
#endif
      const int b0 = (ks ? baddr1 : baddr0) + tilebase;
      bf[ks][0] = *(const bf16x8*)(smem + b0);
#if defined(HALO_EXP) && HALO_EXP >= 2
      bf[ks][1] = bf[ks][0];
#else
      bf[ks][1] = *(const bf16x8*)(smem + b0 + 32 * 64);
#endif
    };
    auto mfmas = [&](int ks) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks][ni], af[ks][mi], acc[mi][ni], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][mi], bf[ks][ni], acc[mi][ni], 0, 0, 0);
    };
    // M16: v_mfma_f32_16x16x32_bf16 (lane row = lane & 15, the lane's 16-byte chunk = lane >> 4; 4 x 4 tiles per wave, one k-step per
    // tap): same LDS traffic and MFMA cycles, but the chip holds a higher clock on this shape (guide, DVFS item 7)
    const int kq = lane >> 4, l16 = lane & 15;
    const int rowofs16 = (wm * 4) * (DMA_HP * 64);                               // mi: + one image row
    const int bn16 = wn * 64 + l16;
    const int baddr16 = 2 * DMA_HBUF + bn16 * 64 + ((kq ^ ((bn16 >> 2) & 3)) << 4);   // ni: + 16 rows (same swizzle)
    bf16x8 af16[4], bf16[4];
    auto readsA16 = [&](int t, int p) {
      const int hxl = l16 + tt.dx[t] - hx0;
      const int a0 = rowofs16 + (c & 1) * DMA_HBUF + (tt.dy[t] - hy0) * (DMA_HP * 64) + ((hxl << 6) | ((kq << 4) ^ ((hxl << 2) & 0x30)));
      af16[2 * p] = *(const bf16x8*)(smem + a0 + (2 * p) * DMA_HP * 64);
      af16[2 * p + 1] = *(const bf16x8*)(smem + a0 + (2 * p + 1) * DMA_HP * 64);
    };
    auto readsB16 = [&](int tilebase) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bf16[ni] = *(const bf16x8*)(smem + baddr16 + tilebase + ni * 16 * 64);
    };
    auto mfmas16 = [&](int p) {
#pragma unroll
      for (int mi = 2 * p; mi < 2 * p + 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc16[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af16[mi], bf16[ni], acc16[mi][ni], 0, 0, 0);
    };
#ifdef HALO_STAMPS
    unsigned long long dsA = 0, dsB = 0, dsC = 0, dsD = 0, dst0 = 0, dst1 = 0, dst2 = 0, dst3 = 0, dst4 = 0;
#endif
    auto step = [&](int q, auto bufc) {
      constexpr int buf = decltype(bufc)::value;
#ifdef HALO_STAMPS
      STAMP(dst0)
#endif
      if constexpr (MOD) {
        // the next chunk's halo was issued in step g == 0 and that step's barrier waited for it: it is scaled at the top of step 1,
        // under this step's MFMAs, not in the chunk's last step where every wave would do it right in front of the barrier
        // (spread issue: the last piece has landed at the barrier of step 2 -> scaled at the top of step 3)
        if (ngroups > 1 && g == (spread ? 3 : 1) && c + 1 < nchunks) scale_inplace((c + 1) & 1, false);
      }
      // HALO_EXP 3 / 4 / 5 (timing experiments, wrong results): the weight tiles / the halo images / both are not fetched after the
      // first stage -- how much of a step is the L2 -> LDS fill stream
#if defined(HALO_EXP) && (HALO_EXP == 3 || HALO_EXP == 5)
      if (q + 1 < total) advance();
#else
      if (q + 1 < total) { dma_b(lc, lg, buf ^ 1); advance(); }
#endif
      int pend = 0;                                              // halo pieces issued in this step (wave-uniform)
#if !(defined(HALO_EXP) && (HALO_EXP == 4 || HALO_EXP == 5))
      if (spread) {
        if (c + 1 < nchunks) {
          if (g == 0) {
            if (hdo[0]) { dma_halo_piece((c + 1) * BK, (c + 1) & 1, 0); ++pend; }
            if (hdo[1]) { dma_halo_piece((c + 1) * BK, (c + 1) & 1, 1); ++pend; }
          } else if (g == 1 && hdo[2]) { dma_halo_piece((c + 1) * BK, (c + 1) & 1, 2); ++pend; }
        }
      } else if (g == 0 && c + 1 < nchunks) dma_halo((c + 1) * BK, (c + 1) & 1);
#endif
      auto step_barrier = [&]() {
#ifdef HALO_STAMPS
        STAMP(dst2)
        if (spread && pend == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (spread && pend == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(dst3)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        STAMP(dst4)
        dsA += dst1 - dst0; dsB += dst2 - dst1; dsC += dst3 - dst2; dsD += dst4 - dst3;
#else
        if (spread) {
          if (pend == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          else if (pend == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        } else {
          __syncthreads();                                       // (waits for this step's DMA: vmcnt(0), then the barrier)
        }
#endif
      };
      const int t0 = g * TP;
      if constexpr (M16) {
        readsA16(t0, 0); readsA16(t0, 1); readsB16(buf * TP * DMA_BBUF);
        if (TP == 2 && t0 + 1 < ntaps) {
          mfmas16(0);
          __builtin_amdgcn_sched_barrier(0);
          readsA16(t0 + 1, 0);
          __builtin_amdgcn_sched_barrier(0);
          mfmas16(1);
          __builtin_amdgcn_sched_barrier(0);
          readsA16(t0 + 1, 1); readsB16((buf * TP + 1) * DMA_BBUF);
          __builtin_amdgcn_sched_barrier(0);
        }
        mfmas16(0);
        mfmas16(1);
        if constexpr (MOD) {
          if (ngroups == 1 && c + 1 < nchunks) scale_inplace((c + 1) & 1, true);
        }
        step_barrier();
        if (++g == ngroups) { g = 0; ++c; }
        return;
      }
      frag_reads(t0, buf * TP * DMA_BBUF, 0);
      frag_reads(t0, buf * TP * DMA_BBUF, 1);
#ifdef HALO_STAMPS
      STAMP(dst1)
#endif
#pragma unroll
      for (int j = 1; j < TP; ++j)
        if (t0 + j < ntaps) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            mfmas(ks);
            __builtin_amdgcn_sched_barrier(0);                   // (pins the register reuse: the reads below overwrite the operands above)
            frag_reads(t0 + j, (buf * TP + j) * DMA_BBUF, ks);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      mfmas(0);
      mfmas(1);
      if constexpr (MOD) {
        if (ngroups == 1 && c + 1 < nchunks) scale_inplace((c + 1) & 1, true);
      }
      step_barrier();
      if (++g == ngroups) { g = 0; ++c; }
    };
#ifdef HALO_STAMPS
    unsigned long long ck0 = 0, ck1 = 0, rt0 = 0, rt1 = 0;
    STAMP(ck0) STAMP_RT(rt0)
    life1 = rt0;
#endif
    for (int q = 0; q < total; q += 2) {
      step(q, std::integral_constant<int, 0>{});
      if (q + 1 < total) step(q + 1, std::integral_constant<int, 1>{});
    }
#ifdef HALO_STAMPS
    STAMP(ck1) STAMP_RT(rt1)
    life2 = rt1;
    if (!M16 && lane == 0 && blockIdx.x < 2048 && blockIdx.z == 0 && blockIdx.y == 0) {
      unsigned long long* o = g_halo_stamps + ((size_t)blockIdx.x * 8 + wid) * 5;
      o[0] = dsA; o[1] = dsB; o[2] = dsC; o[3] = dsD; o[4] = total;
      g_halo_clock[((size_t)blockIdx.x * 8 + wid) * 2] = ck1 - ck0; g_halo_clock[((size_t)blockIdx.x * 8 + wid) * 2 + 1] = rt1 - rt0;
    }
#endif
  } else if constexpr (PAIR) {
    // ---- weight tiles of TWO taps per step (one barrier per tap pair): 2 x 128 rows x 4 vectors, two items per thread ----
    const int ngroups = (ntaps + 1) >> 1, total = ngroups * nchunks;
    struct BR { bf16x8 v[2]; };
    auto b_load = [&](int c, int g) -> BR {
      BR r;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = 2 * g + j;
        const bool ok = t < ntaps;
        r.v[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wres, (bvalid && ok) ? wrow : 0xffffffffu,
                                                                                 __builtin_amdgcn_readfirstlane(2 * (tt.wt[ok ? t : 0] * a.N * a.Kpad + c * BK)), 0));
      }
      return r;
    };
    auto b_store = [&](int buf, const BR& r) {
#pragma unroll
      for (int j = 0; j < 2; ++j) *(bf16x8*)(Bt + (buf * 2 + j) * TILE + brow * HROW + hvec * 8) = r.v[j];
    };
    int lc = 0, lg = 0;
    auto advance = [&]() { if (++lg == ngroups) { lg = 0; ++lc; } };
    if (a.pre) {
      for (int i = tid; i < a.Kpad; i += 512) psc[i] = i < a.Cin ? a.pre[(size_t)b * a.Cin + i] : 0.f;
      __syncthreads();
    }
    halo_load(0);
    halo_store();
    b_store(0, b_load(0, 0));
    advance();
    BR r0;                                                       // weight tiles ONE step ahead (a step is 16 MFMAs per wave deep)
    r0.v[0] = zero_bf16x8(); r0.v[1] = zero_bf16x8();
    __syncthreads();

    int c = 0, g = 0;
    // one step: the (up to) two taps of group q from buffer pair (q & 1); the second tap's fragments are fetched into the registers
    // the first tap's MFMAs release, so its LDS reads run under those MFMAs
    auto step = [&](int q, BR& rs) {
      if (g == 0 && c + 1 < nchunks) halo_load((c + 1) * BK);
      if (q + 1 < total) { rs = b_load(lc, lg); advance(); }
      const int t0 = 2 * g;
      const bool two = t0 + 1 < ntaps;
      const int toff0 = (tt.dy[t0] - hy0) * rp + (tt.dx[t0] - hx0) * HROW;
      const int toff1 = two ? (tt.dy[t0 + 1] - hy0) * rp + (tt.dx[t0 + 1] - hx0) * HROW : 0;
      const __bf16* Bc = Bt + (q & 1) * 2 * TILE;
      bf16x8 af[2][2], bf[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[ks][mi] = *(const bf16x8*)(halo + abase[mi] + toff0 + ks * 16);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bf[ks][ni] = *(const bf16x8*)(Bc + bbase + ni * 32 * HROW + ks * 16);
      }
      if (two) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks][ni], af[ks][mi], acc[mi][ni], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][mi], bf[ks][ni], acc[mi][ni], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);                     // (pins the register reuse: the reads below overwrite the operands above)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) af[ks][mi] = *(const bf16x8*)(halo + abase[mi] + toff1 + ks * 16);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) bf[ks][ni] = *(const bf16x8*)(Bc + TILE + bbase + ni * 32 * HROW + ks * 16);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks][ni], af[ks][mi], acc[mi][ni], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][mi], bf[ks][ni], acc[mi][ni], 0, 0, 0);
      if (g == ngroups - 1 && c + 1 < nchunks) {
        __syncthreads();
        halo_store();
      }
      if (q + 1 < total) b_store((q + 1) & 1, rs);
      __syncthreads();
      if (++g == ngroups) { g = 0; ++c; }
    };
    for (int q = 0; q < total; ++q) step(q, r0);
  } else {
    // ---- weight tile: 128 rows x 4 vectors = 512 items, one per thread; prefetched TWO taps ahead in two named registers ----
    const int total = ntaps * nchunks;
    auto b_load = [&](int c, int t) -> bf16x8 {
      return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wres, bvalid ? wrow : 0xffffffffu, __builtin_amdgcn_readfirstlane(2 * (tt.wt[t] * a.N * a.Kpad + c * BK)), 0));
    };
    auto b_store = [&](int buf, const bf16x8& r) { *(bf16x8*)(Bt + buf * TILE + brow * HROW + hvec * 8) = r; };

    // (c, t) of the tile two steps ahead of the one being computed
    int lc = 0, lt = 0;
    auto advance = [&]() { if (++lt == ntaps) { lt = 0; ++lc; } };

    if (a.pre) {
      for (int i = tid; i < a.Kpad; i += 512) psc[i] = i < a.Cin ? a.pre[(size_t)b * a.Cin + i] : 0.f;
      __syncthreads();
    }
    halo_load(0);
    halo_store();
    b_store(0, b_load(0, 0));
    advance();                                                   // -> tile 1
    bf16x8 r0 = (1 < total) ? b_load(lc, lt) : zero_bf16x8();    // tile 1 in flight
    advance();                                                   // -> tile 2
    bf16x8 r1 = zero_bf16x8();
    __syncthreads();

    int c = 0, t = 0;
#ifdef HALO_STAMPS
    unsigned long long sA = 0, sB = 0, sC = 0, sD = 0, st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0;
    STAMP(st0)
#endif
    // one step: compute tile q from buffer (q & 1); `rs` holds tile q+1 (loaded one step ago), `rl` receives tile q+2
    auto step = [&](int q, bf16x8& rs, bf16x8& rl) {
      if (t == 0 && c + 1 < nchunks) halo_load((c + 1) * BK);    // next chunk's halo: in flight during this chunk's taps
      if (q + 2 < total) { rl = b_load(lc, lt); advance(); }
      const int toff = (tt.dy[t] - hy0) * rp + (tt.dx[t] - hx0) * HROW;
      const __bf16* Bc = Bt + (q & 1) * TILE;
      if (M16) {
        bf16x8 af[4], bf[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi] = *(const bf16x8*)(halo + abase[mi] + toff);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bf[ni] = *(const bf16x8*)(Bc + bbase + ni * 16 * HROW);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc16[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bf[ni], acc16[mi][ni], 0, 0, 0);
      } else {
        bf16x8 af[2][2], bf[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) af[ks][mi] = *(const bf16x8*)(halo + abase[mi] + toff + ks * 16);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) bf[ks][ni] = *(const bf16x8*)(Bc + bbase + ni * 32 * HROW + ks * 16);
        }
        STAMP(st1)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks][ni], af[ks][mi], acc[mi][ni], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][mi], bf[ks][ni], acc[mi][ni], 0, 0, 0);
        STAMP(st2)
      }
      if (t == ntaps - 1 && c + 1 < nchunks) {
        __syncthreads();                                         // every wave is done reading this chunk's halo
        halo_store();
      }
      if (q + 1 < total) b_store((q + 1) & 1, rs);
      STAMP(st3)
      __syncthreads();
      STAMP(st4)
#ifdef HALO_STAMPS
      sA += st1 - st0; sB += st2 - st1; sC += st3 - st2; sD += st4 - st3; st0 = st4;
#endif
      if (++t == ntaps) { t = 0; ++c; }
    };
    for (int q = 0; q < total; q += 2) {
      step(q, r0, r1);
      if (q + 1 < total) step(q + 1, r1, r0);
    }

#ifdef HALO_STAMPS
  if (!M16 && lane == 0 && blockIdx.x < 2048 && nb == 0 && blockIdx.z == 0) {
    unsigned long long* o = g_halo_stamps + ((size_t)blockIdx.x * 8 + wid) * 5;
    o[0] = sA; o[1] = sB; o[2] = sC; o[3] = sD; o[4] = total;
  }
#endif
  }
  if constexpr (CAN_SPLIT) {
    if (a.nsplit > 1) {
      // partial tiles meet as in conv_igemm8_kernel: relaxed device-scope stores to this split's slab, vmcnt(0), the tile's arrival
      // counter; the last split to arrive re-reads ALL the slabs in split order (its own included: the sum does not depend on who was
      // last, and no second set of 64 accumulator registers is needed) and goes on to the epilogue
      int* flag = (int*)smem;
      const int tileid = blockIdx.z * gridDim.x + blockIdx.x, split = blockIdx.y;
      float* slabs = a.slab + (size_t)tileid * a.nsplit * (256 * BN);
      float* mine = slabs + (size_t)split * (256 * BN);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __hip_atomic_store(mine + ((mi * 2 + ni) * 16 + r) * 512 + tid, acc[mi][ni][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      SLAB_PUBLISH_FENCE();
      __syncthreads();
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(a.cnt + tileid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == a.nsplit - 1;
        if (last) __hip_atomic_store(a.cnt + tileid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
      }
      __syncthreads();
      SLAB_CONSUME_FENCE();
      const int last = *flag;
      __syncthreads();                                           // (the epilogue reuses this LDS)
      if (!last) return;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
      for (int sp = 0; sp < a.nsplit; ++sp) {
        const float* other = slabs + (size_t)sp * (256 * BN);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = __hip_atomic_load(other + (q * 16 + r) * 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[q >> 1][q & 1][r] += v[r];
        }
      }
    }
  }
  // ---- epilogue: demod/bias/act in registers -> bf16 tile in LDS -> 16-byte coalesced stores (+ residual) ----------------
  // (the main loop ended with a barrier, so the staging buffers are free)
  constexpr int OROW = BN + 8;                                     // bf16 per output row in LDS (272 B: conflict-light)
  __bf16* ot = (__bf16*)smem;                                      // [256][OROW]
  float* colbuf = (float*)(smem + 256 * OROW * sizeof(__bf16));    // [128] column sums of xs * acc (fused style-gradient reduction)
  float* cbias = colbuf + BN;                                      // [128] bias * bias_scale, [128] demodulation scale of this sample
  float* cpost = cbias + BN;
  const __bf16* side = SR ? a.xs : a.residual;                     // the tile that meets the accumulators: residual, or xs
  if (SR && tid < BN) colbuf[tid] = 0.f;
  if (TR && tid < BN) { cbias[tid] = ep_bias * a.bias_scale; cpost[tid] = ep_post; }
  const float ep_slope = a.act == ACT_LRELU ? LRELU_SLOPE : 1.f;
  // half-resolution residual under a stride-1 geometry (TR layout): the 4 lanes that share a source pixel read its 8-byte channel
  // groups straight from global memory (L1 hits) in the loop below -- no staging pass, no 4x-redundant 16-byte loads
  const bool quarter = TR && EPI == 2 && a.out_mul == 1;
  if (EPI != 0 && !quarter) {                                      // stage it with coalesced 16-byte loads
    // (unconditional buffer loads, rows outside the image / channels beyond Cout read zeros by the range check: the eight loads of
    //  a thread are in flight together; as conditional loads each one waited for the one before)
    const int sh = EPI == 2 ? 1 : 0;
    const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc((void*)side, 0, (int)(2u * (unsigned)(a.B * (a.Hout >> sh) * (a.Wout >> sh) * a.Cout)), 0x00020000);
    bf16x8 rr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = tid + k * 512;
      const int row = idx >> 4, vv = idx & 15;
      const int py = ty * HT + (row >> 4), px = tx * HT + (row & 15);
      const int n = n0 + vv * 8;
      const int oy = (py * a.out_mul + (phase >> 1)) >> sh, ox = (px * a.out_mul + (phase & 1)) >> sh;
      const bool ok = py < a.Hm && px < a.Wm && n < a.Cout;
      rr[k] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(sres, ok ? 2u * (unsigned)(((b * (a.Hout >> sh) + oy) * (a.Wout >> sh) + ox) * a.Cout + n) : 0xffffffffu, 0, 0));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = tid + k * 512;
      *(bf16x8*)(ot + (idx >> 4) * OROW + (idx & 15) * 8) = rr[k];
    }
  }
  if (TR || EPI != 0) __syncthreads();
  constexpr float res_scale = EPI == 2 ? 0.25f : 1.f;
  if constexpr (TR) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    // half-resolution residual: the 16 eight-byte groups this lane needs (2 pixels x 2 x 4 channel groups) come by unconditional
    // buffer loads issued together (an invalid source pixel / channel group gets offset 0xffffffff = zeros): as conditional loads
    // inside the loop below they were 16 serialised L1 round trips per workgroup (+16 % on the 256 x 256 data gradient)
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 rq[2][2][4];
    if (quarter) {
      const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc((void*)side, 0, (int)(2u * (unsigned)(a.B * (a.Hout >> 1) * (a.Wout >> 1) * a.Cout)), 0x00020000);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int row = wm * 64 + mi * 32 + (lane & 31);
        const int qy = (ty * HT + (row >> 4)) >> 1, qx = (tx * HT + (row & 15)) >> 1;
        const bool pok = qy < (a.Hout >> 1) && qx < (a.Wout >> 1);
        const unsigned pofs = 2u * (unsigned)(((b * (a.Hout >> 1) + qy) * (a.Wout >> 1) + qx) * a.Cout + n0);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int ch = wn * 64 + ni * 32 + 8 * g + 4 * (lane >> 5);
            rq[mi][ni][g] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(qres, (pok && n0 + ch < a.Cout) ? pofs + 2u * ch : 0xffffffffu, 0, 0));
          }
      }
    }
    unsigned mw[2][2] = {{0u, 0u}, {0u, 0u}};                                   // sign bits of this lane's pre-activations (MK: a.mask_out)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = wn * 64 + ni * 32 + 8 * g + 4 * (lane >> 5);            // this lane's 4 channels of accumulator group g
        const f32x4 bv = *(const f32x4*)(cbias + ch), pv = *(const f32x4*)(cpost + ch);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int row = wm * 64 + mi * 32 + (lane & 31);                     // this lane's pixel
          // The epilogue is VALU work -- 64 elements per lane, the two waves of a SIMD at once: 3.4-3.9 us of a workgroup's life as ~10
          // instructions per element (scripts/halo_life.py) -- so it is written on 4-vectors (packed fma / mul, two-element bf16 converts) and
          // leaky ReLU is max(t, 0.2 t) (slope 1 without activation): the same values as compare + select, bit for bit
          const f32x4 av = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
          const f32x4 t = av * pv + bv;
          const f32x4 u = t * ep_slope;
          f32x4 v = {fmaxf(t[0], u[0]), fmaxf(t[1], u[1]), fmaxf(t[2], u[2]), fmaxf(t[3], u[3])};
          v = v * a.gain;
          if (MK) {
            // (the elements' bits through ONE cast of the whole vector: __builtin_bit_cast of t[j] inside the unrolled loop was compiled to the
            //  bits of t[0] for every j -- caught by test_activation_sign_masks)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 tb = __builtin_bit_cast(u32x4, t);
            // "not positive" bits by integer arithmetic on the float's bits -- bits(t) - 1 has its top bit set exactly for t <= +0 and
            //  for negative t other than -0, which fp32 accumulation from +0 never produces -- instead of 64 compares, each holding an SGPR pair
#pragma unroll
            for (int j = 0; j < 4; ++j) mw[mi][ni] |= ((tb[j] - 1u) >> 31) << (8 * g + j);           // (the lane's half -- 4 more -- is applied once, below)
          }
          if (EPI != 0) {
            bf16x4 rr = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            if (!quarter) rr = *(const bf16x4*)(ot + row * OROW + ch);
            else rr = __builtin_bit_cast(bf16x4, rq[mi][ni][g]);
            v = v + res_scale * __builtin_convertvector(rr, f32x4);
          }
          *(bf16x4*)(ot + row * OROW + ch) = __builtin_convertvector(v, bf16x4);
        }
      }
    if (MK) {
      // activation sign mask: one 32-bit word per (pixel, 32 channels), bit c % 32 = (pre-activation of channel c > 0).  A lane holds the
      // nibbles 8 g + 4 (lane >> 5) of its pixel's two words (ni); its partner 32 lanes away holds the other nibbles.
      const int words = a.Cout >> 5;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const unsigned mh = mw[mi][ni] << (4 * (lane >> 5));
          const unsigned wv = ~(mh | (unsigned)__shfl_xor((int)mh, 32, 64));      // (the lanes collected the COMPLEMENT)
          const int row = wm * 64 + mi * 32 + (lane & 31);
          const int py = ty * HT + (row >> 4), px = tx * HT + (row & 15);
          const int wi = (n0 >> 5) + wn * 2 + ni;
          if (lane < 32 && py < a.Hm && px < a.Wm && wi < words)
            a.mask_out[((size_t)(b * a.Hout + py) * a.Wout + px) * words + wi] = wv;
        }
    }
  } else {
    auto colconst = [&](int nl, float& bv, float& pv) {              // per output column: bias and demodulation scale
      const int n = n0 + nl;
      bv = (a.bias && n < a.N) ? a.bias[n] * a.bias_scale : 0.f;
      pv = (a.post && n < a.Cout) ? a.post[(size_t)b * a.Cout + n] : 1.f;
    };
    auto emit = [&](int row, int nl, float accv, float bv, float pv, float& cs) {
      float v = accv * pv + bv;
      v = fmaxf(v, v * ep_slope) * a.gain;
      if (SR) cs += accv * (float)ot[row * OROW + nl];               // style-gradient partial: x * (unscaled data gradient)
      else if (EPI != 0) v += res_scale * (float)ot[row * OROW + nl];     // same thread reads and rewrites this element: one rounding
      ot[row * OROW + nl] = (__bf16)v;
    };
    auto colflush = [&](int nl, float cs, int width) {               // lanes sharing a column -> one LDS add per wave and column
      if (!SR) return;
      for (int o = width; o < 64; o <<= 1) cs += __shfl_xor(cs, o, 64);
      if (lane < width) atomicAdd(&colbuf[nl], cs);
    };
    if (M16) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int nl = wn * 64 + ni * 16 + (lane & 15);
        float bv, pv, cs = 0.f;
        colconst(nl, bv, pv);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) emit(wm * 64 + mi * 16 + (lane >> 4) * 4 + r, nl, acc16[mi][ni][r], bv, pv, cs);
        colflush(nl, cs, 16);
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int nl = wn * 64 + ni * 32 + (lane & 31);
        float bv, pv, cs = 0.f;
        colconst(nl, bv, pv);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) emit(wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), nl, acc[mi][ni][r], bv, pv, cs);
        colflush(nl, cs, 32);
      }
    }
  }
  __syncthreads();
#ifdef HALO_STAMPS
  STAMP_RT(lifeC)
#endif
  if (SR && tid < BN && n0 + tid < a.Cout) atomicAdd(a.gs + (size_t)b * a.Cout + n0 + tid, colbuf[tid]);
  if (SR && a.residual) {
    // style-gradient epilogue + a full-resolution residual: the gradient another consumer of the same tensor has already produced
    // (ops.SynthForkFn chains the data gradients of a generator block's three consumers) is added as the finished tile goes out --
    // bf16(bf16(s * u) + r), the value the separate `add_` pass it replaces would have stored
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)a.residual, 0, (int)(2u * (unsigned)(a.B * a.Hout * a.Wout * a.Cout)), 0x00020000);
    bf16x8 rr[8];
    unsigned ro[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = tid + k * 512;
      const int row = idx >> 4, vv = idx & 15;
      const int py = ty * HT + (row >> 4), px = tx * HT + (row & 15);
      const int n = n0 + vv * 8;
      const int oy = py * a.out_mul + (phase >> 1), ox = px * a.out_mul + (phase & 1);
      ro[k] = (py < a.Hm && px < a.Wm && n < a.Cout) ? 2u * (unsigned)(((b * a.Hout + oy) * a.Wout + ox) * a.Cout + n) : 0xffffffffu;
      rr[k] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rres, ro[k], 0, 0));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = tid + k * 512;
      if (ro[k] == 0xffffffffu) continue;
      const bf16x8 t = *(const bf16x8*)(ot + (idx >> 4) * OROW + (idx & 15) * 8);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)((float)t[j] + (float)rr[k][j]);
      *(bf16x8*)((char*)a.y + ro[k]) = o;
    }
  } else {
    // 256 rows x 16 vectors, thread t: vector t & 15 of the rows (t >> 4) + 32 k -- the same tile column, two tile rows further per k:
    // ONE byte offset per thread, the k-th store adds k strides; a row / column / channel outside the output gets offset 0xffffffff and is
    // dropped by the buffer's range check.  (As eight 64-bit addresses the loop was ~150 vector instructions, 1.4 us per workgroup.  The
    // stride goes into the VECTOR offset: passed as the instruction's scalar offset -- an SGPR the compiler also used for the lane mask of the
    // row test just before -- a few stores per launch went astray; test_conv_fwd_pooled_byproduct caught it.)
    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)(2u * (unsigned)(a.B * a.Hout * a.Wout * a.Cout)), 0x00020000);
    const int r0 = tid >> 4, vv = tid & 15;
    const int py0 = ty * HT + (r0 >> 4), px = tx * HT + (r0 & 15), n = n0 + vv * 8;
    const unsigned off0 = 2u * (unsigned)(((b * a.Hout + py0 * a.out_mul + (phase >> 1)) * a.Wout + px * a.out_mul + (phase & 1)) * a.Cout + n);
    const unsigned kstride = 4u * (unsigned)(a.out_mul * a.Wout * a.Cout);          // bytes between the rows of k and k + 1
    const bool colok = px < a.Wm && n < a.Cout;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool ok = colok && py0 + 2 * k < a.Hm;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, *(const bf16x8*)(ot + (r0 + 32 * k) * OROW + vv * 8)), yres, ok ? off0 + k * kstride : 0xffffffffu, 0, 0);
    }
  }
#ifdef HALO_STAMPS
  STAMP_RT(lifeD)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the tile's stores have been acknowledged
  STAMP_RT(life3)
  if (tid == 0) {
    const size_t wgid = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
    if (wgid < 16384) {
      unsigned hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      unsigned long long* o = g_halo_life + wgid * 9;
      o[0] = life0; o[1] = life1; o[2] = life2; o[3] = life3; o[4] = hwid; o[5] = lifeA; o[6] = lifeB; o[7] = lifeC; o[8] = lifeD;
    }
  }
#endif
  if constexpr (EPI == 1) {
    // by-product of a DiscriminatorBlock's closing 1x1 convolution (custom_layers.py:203,209): avg_pool2d(out, 2), which the NEXT
    // block's skip branch reads (custom_layers.py:202), from the finished bf16 tile in LDS -- the same values, summation and
    // rounding as the avgpool2 kernel that otherwise re-reads the whole output from HBM.  (launcher: even Hout / Wout, out_mul 1)
    if (a.pool_out) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = tid + k * 512;                             // 64 pooled pixels x 16 vectors
        const int q = idx >> 4, vv = idx & 15;
        const int qy = q >> 3, qx = q & 7;
        const int py = ty * HT + 2 * qy, px = tx * HT + 2 * qx, n = n0 + vv * 8;
        if (py >= a.Hm || px >= a.Wm || n >= a.Cout) continue;
        const __bf16* r0 = ot + ((2 * qy) * 16 + 2 * qx) * OROW + vv * 8;
        const bf16x8 t00 = *(const bf16x8*)r0, t01 = *(const bf16x8*)(r0 + OROW), t10 = *(const bf16x8*)(r0 + 16 * OROW), t11 = *(const bf16x8*)(r0 + 17 * OROW);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)(((((float)t00[j] + (float)t01[j]) + (float)t10[j]) + (float)t11[j]) * 0.25f);   // (the pooling kernel's order)
        *(bf16x8*)(a.pool_out + ((size_t)(b * (a.Hout >> 1) + (py >> 1)) * (a.Wout >> 1) + (px >> 1)) * a.Cout + n) = o;
      }
    }
  }
}


// =========================================================================================================
// conv_s2duo_kernel -- stride-2 3 x 3 forward convolution, two ANTI-PHASED teams in one 1024-thread workgroup.
//
// Why (round 4, measured): conv_halo_kernel<2, ., ., 4> runs ONE stage per workgroup (the planes of a 16-channel half-chunk + the weight
// tiles of all nine taps: 78 KB; load, barrier, 36 MFMAs per wave, barrier) and relies on the second workgroup of the CU to compute while
// this one loads.  It does not happen: with one workgroup per CU (option 3, bit 32) a 256 x 256 launch takes 537 us against 501 with two --
// the two co-resident workgroups start together, stay in lock step, load together (sharing the CU's fill path: each load twice as long) and
// compute together (sharing the matrix pipe): 2 L + 2 C per pair of stages where max(L, C) ~ (L + C) / 2 would do (L = 1.56 us, C = 1.44 us).
// Nothing synchronises two workgroups, so here the pair IS one workgroup: team A (waves 0-7) and team B (waves 8-15) own one output
// tile each (neighbours in x, the SAME 128-channel block, hence the same weights) and one barrier per phase keeps them half a period apart:
//     phase 2 h     : A computes half-chunk h            | B's planes of half-chunk h and 20 KB of the weight tiles of h + 1 land
//     phase 2 h + 1 : B computes half-chunk h            | A's planes of h + 1 and the other 16 KB of the weight tiles of h + 1 land
// LDS: planes A, planes B (42 KB each), two weight stages (36 KB each) = 156 KB; the weight bytes per MFMA halve on the way.
// Same operand images, fragment addresses, tap order and accumulation order as the one-stage kernel: results are bit-identical to it.
// Epilogue: the plain transposed-accumulator form (bias, leaky ReLU, gain; EPI_ == 4: + the activation sign mask), per team.
// =========================================================================================================
template <int EPI_>
__global__ __launch_bounds__(1024) void conv_s2duo_kernel(HaloArgs a) {
  constexpr bool MK = EPI_ == 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PP = 20;                                       // plane row pitch in 32-byte records (see conv_halo_kernel, DMA == 3 / 4)
  constexpr int P_OFF[4] = {0, 17 * PP, 2 * 17 * PP, 2 * 17 * PP + 16 * PP};
  constexpr int NREC = 2 * 17 * PP + 2 * 16 * PP;
  constexpr int HPIECES = (NREC + 31) / 32;                    // 42 one-KB pieces
  constexpr int S2_H = HPIECES * 1024, S2_B = 9 * 4096;
  const int tid = threadIdx.x, team = tid >> 9, ltid = tid & 511, lane = tid & 63, lwid = ltid >> 6;
  const int wm = lwid >> 1, wn = lwid & 1;
  const int widu = __builtin_amdgcn_readfirstlane(lwid), teamu = __builtin_amdgcn_readfirstlane(team);
  char* planes = smem + teamu * S2_H;                           // this team's planes
  char* wst = smem + 2 * S2_H;                                  // two weight stages

  // workgroup -> (pair of tiles, channel block), XCD-contiguous like conv_halo_kernel; team t owns tile 2 pair + t
  const HaloArgs::Decode dc = a.dec;
  {                                                             // (the prologue's scalars in one batch of loads: see conv_halo_kernel)
    const void *p0 = a.x, *p1 = a.w, *p2 = a.bias, *p3 = a.post;
    const int i0 = a.B, i1 = a.Hin, i2 = a.Win, i3 = a.Cin, i4 = a.Cout, i5 = a.N, i6 = a.Kpad, i7 = a.Hm, i8 = a.Wm;
    asm volatile("" ::"s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(i0), "s"(i1), "s"(i2), "s"(i3), "s"(i4), "s"(i5), "s"(i6), "s"(i7), "s"(i8));
  }
  const int npair = dc.ntile;                                   // (this kernel's launch passes its PAIR count and the matching shifts)
  int pair, nb;
  if ((npair & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = (npair >> 3) * dc.nb_group;
    int nbo, rem, pq, br;
    divmod_sh((unsigned)slot, per, dc.sh_per, nbo, rem);
    divmod_sh((unsigned)rem, dc.nb_group, dc.sh_grp, pq, br);
    pair = xcd * (npair >> 3) + pq;
    nb = nbo * dc.nb_group + br;
  } else {
    pair = blockIdx.x % npair;
    nb = blockIdx.x / npair;
  }
  const int tile = 2 * pair + team;
  const int n0 = nb * BN;
  int tx, ty, b, trow;
  divmod_sh((unsigned)tile, dc.tiles_x, dc.sh_tx, trow, tx);
  divmod_sh((unsigned)trow, dc.tiles_y, dc.sh_ty, b, ty);
  const int gy0 = ty * HT * 2 - 1, gx0 = tx * HT * 2 - 1;       // input pixel of plane record (0, 0): halo origin (-1, -1)
  const TapTable& tt = a.taps[0];
  float ep_bias = 0.f, ep_post = 1.f;
  if (ltid < BN) {
    const int n = n0 + ltid;
    if (a.bias && n < a.N) ep_bias = a.bias[n];                   // (the RAW value: multiplied by bias_scale here, the load was WAITED for here -- a memory round trip in every workgroup's prologue; the scale is applied where the value is staged for the epilogue)
    if (a.post && n < a.Cout) ep_post = a.post[(size_t)b * a.Cout + n];
  }
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(2u * (unsigned)(a.B * a.Hin * a.Win * a.Cin)), 0x00020000);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);
  unsigned hvo[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int L = 32 * (widu + 8 * k) + (lane >> 1);           // record of this lane in piece widu + 8 k
    const int pl = L < P_OFF[1] ? 0 : L < P_OFF[2] ? 1 : L < P_OFF[3] ? 2 : 3;
    const int rc = L - (pl == 0 ? P_OFF[0] : pl == 1 ? P_OFF[1] : pl == 2 ? P_OFF[2] : P_OFF[3]);
    const int r = rc / PP, cc = rc - r * PP, pr = pl >> 1, pc = pl & 1;
    const int ch = (lane & 1) ^ ((cc >> 2) & 1);
    const int gy = gy0 + 2 * r + pr, gx = gx0 + 2 * cc + pc;
    const bool ok = L < NREC && r < 17 - pr && cc < 17 - pc && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
    hvo[k] = ok ? 2u * (unsigned)(((b * a.Hin + gy) * a.Win + gx) * a.Cin + ch * 8) : 0xffffffffu;
  }
  const int wrow = 32 * (widu & 3) + (lane >> 1);              // weight row of this lane: piece i = widu + 8 k is (tap i / 4, rows 32 (i % 4) ..)
  const unsigned wvo = n0 + wrow < a.N ? 2u * (unsigned)((n0 + wrow) * a.Kpad + (((lane & 1) ^ ((wrow >> 3) & 1)) * 8)) : 0xffffffffu;
  auto dma_planes = [&](int h) {
    const int cofs = __builtin_amdgcn_readfirstlane(h * 32);
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (widu + 8 * k < HPIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(planes + (widu + 8 * k) * 1024), 16, hvo[k], cofs, 0, 0);
  };
  // weight stage of half-chunk h: pieces [8 k0, 8 k1) + this wave (the nine taps are 36 one-KB pieces; the two load phases of a period
  // take 20 and 16 of them so that both move ~60 KB)
  auto dma_weights = [&](int h, int k0, int k1) {
    char* W = wst + (h & 1) * S2_B;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int i = widu + 8 * k;
      if (k >= k0 && k < k1 && i < 36)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void*)(W + i * 1024), 16, wvo,
                                                 __builtin_amdgcn_readfirstlane(2 * (tt.wt[i >> 2] * a.N * a.Kpad) + h * 32), 0, 0);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int half = lane >> 5, xl = lane & 15, yl = wm * 4 + ((lane & 31) >> 4);
  int acol[2];
#pragma unroll
  for (int sft = 0; sft < 2; ++sft) acol[sft] = yl * (PP * 32) + (xl + sft) * 32 + ((half ^ (((xl + sft) >> 2) & 1)) << 4);
  const int brl = wn * 64 + (lane & 31);
  const int baddr = brl * 32 + ((half ^ ((brl >> 3) & 1)) << 4);
  // a computing team has two waves per SIMD: the fragments of tap t + 1 are requested BEFORE the MFMAs of tap t are issued (two named
  // register sets, the loop fully unrolled), so that a wave's LDS round trip runs under its own matrix work, not only under the other wave's
  auto half_chunk = [&](int h) {
    const char* S = planes;
    const char* W = wst + (h & 1) * S2_B;
    bf16x8 af[2][2], bf[2][2];
    auto frags = [&](int t, int set) {                           // stride-2 forward tap table is row-major over (ky, kx): tap t = 3 ky + kx
      const int ky = t / 3, kx = t - 3 * (t / 3);
      const int pbase = P_OFF[(ky & 1) * 2 + (kx & 1)] * 32 + (ky >> 1) * (PP * 32);
      const int aa = acol[kx >> 1] + pbase;
      af[set][0] = *(const bf16x8*)(S + aa);
      af[set][1] = *(const bf16x8*)(S + aa + 2 * PP * 32);
      bf[set][0] = *(const bf16x8*)(W + baddr + t * 4096);
      bf[set][1] = *(const bf16x8*)(W + baddr + t * 4096 + 32 * 32);
    };
    frags(0, 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 1 < 9) frags(t + 1, (t + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[t & 1][ni], af[t & 1][mi], acc[mi][ni], 0, 0, 0);     // (operands swapped: see TR in conv_halo_kernel)
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  const int nh = 2 * a.kc_per_tap;                              // 16-channel half-chunks
  // prologue: A's first planes + the first weight stage
  if (teamu == 0) { dma_planes(0); dma_weights(0, 0, 5); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int h = 0; h < nh; ++h) {
    // phase 2 h: A computes h | B's planes of h and the first 20 pieces of the weights of h + 1 land (that stage was last read in phase 2 h - 1)
    if (teamu == 0) half_chunk(h);
    else { dma_planes(h); if (h + 1 < nh) dma_weights(h + 1, 0, 3); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // phase 2 h + 1: B computes h | A's planes of h + 1 and the other 16 weight pieces of h + 1 land
    if (teamu == 1) half_chunk(h);
    else if (h + 1 < nh) { dma_planes(h + 1); dma_weights(h + 1, 3, 5); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue (per team): bias / activation / gain in registers -> bf16 tile in LDS -> 16-byte coalesced stores ---------------------
  constexpr int OROW = BN + 8;
  constexpr int TEAM_EPI = 256 * OROW * (int)sizeof(__bf16) + 2 * BN * (int)sizeof(float);
  __bf16* ot = (__bf16*)(smem + teamu * TEAM_EPI);
  float* cbias = (float*)((char*)ot + 256 * OROW * sizeof(__bf16));
  float* cpost = cbias + BN;
  if (ltid < BN) { cbias[ltid] = ep_bias * a.bias_scale; cpost[ltid] = ep_post; }
  const float ep_slope = a.act == ACT_LRELU ? LRELU_SLOPE : 1.f;
  __syncthreads();
  {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    unsigned mw[2][2] = {{0u, 0u}, {0u, 0u}};
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = wn * 64 + ni * 32 + 8 * g + 4 * (lane >> 5);
        const f32x4 bv = *(const f32x4*)(cbias + ch), pv = *(const f32x4*)(cpost + ch);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int row = wm * 64 + mi * 32 + (lane & 31);
          const f32x4 av = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
          const f32x4 t = av * pv + bv;                            // (4-vectors, max(t, slope t): see conv_halo_kernel's epilogue)
          const f32x4 u = t * ep_slope;
          f32x4 v = {fmaxf(t[0], u[0]), fmaxf(t[1], u[1]), fmaxf(t[2], u[2]), fmaxf(t[3], u[3])};
          v = v * a.gain;
          if (MK) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 tb = __builtin_bit_cast(u32x4, t);
#pragma unroll
            for (int j = 0; j < 4; ++j) mw[mi][ni] |= ((tb[j] - 1u) >> 31) << (8 * g + j);           // (the lane's half -- 4 more -- is applied once, below)
          }
          const bf16x4 o = __builtin_convertvector(v, bf16x4);
#ifdef DUO_DIRECT                                                 // timing experiment (measured +3.5 ... +4.5 % per launch, not shipped): 8-byte stores straight from the accumulator layout, no LDS tile
          {
            const int py = ty * HT + (row >> 4), px = tx * HT + (row & 15);
            *(bf16x4*)(a.y + ((size_t)(b * a.Hout + py) * a.Wout + px) * a.Cout + n0 + ch) = o;
          }
#else
          *(bf16x4*)(ot + row * OROW + ch) = o;
#endif
        }
      }
    if (MK) {
      const int words = a.Cout >> 5;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const unsigned mh = mw[mi][ni] << (4 * (lane >> 5));
          const unsigned wv = ~(mh | (unsigned)__shfl_xor((int)mh, 32, 64));
          const int row = wm * 64 + mi * 32 + (lane & 31);
          const int py = ty * HT + (row >> 4), px = tx * HT + (row & 15);
          const int wi = (n0 >> 5) + wn * 2 + ni;
          if (lane < 32 && py < a.Hm && px < a.Wm && wi < words)
            a.mask_out[((size_t)(b * a.Hout + py) * a.Wout + px) * words + wi] = wv;
        }
    }
  }
#ifndef DUO_DIRECT
  __syncthreads();
  {                                                               // (one offset per thread + k strides: see conv_halo_kernel)
    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)(2u * (unsigned)(a.B * a.Hout * a.Wout * a.Cout)), 0x00020000);
    const int r0 = ltid >> 4, vv = ltid & 15;
    const int py0 = ty * HT + (r0 >> 4), px = tx * HT + (r0 & 15), n = n0 + vv * 8;
    const unsigned off0 = 2u * (unsigned)(((b * a.Hout + py0) * a.Wout + px) * a.Cout + n);
    const unsigned kstride = 4u * (unsigned)(a.Wout * a.Cout);
    const bool colok = px < a.Wm && n < a.Cout;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool ok = colok && py0 + 2 * k < a.Hm;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, *(const bf16x8*)(ot + (r0 + 32 * k) * OROW + vv * 8)), yres, ok ? off0 + k * kstride : 0xffffffffu, 0, 0);
    }
  }
#endif
}

// =========================================================================================================
// halo-tile kernel for NARROW layers (Cout <= 64: the C = 32 / 64 octaves of the 512 x 512 and 1024 x 1024 networks), stride-1
// geometries.  conv_halo_kernel's 128-channel N tile wastes 2-4x of its MFMA work there and its 256-position tiles pay the
// per-workgroup fixed cost 131 072 times per 1024 x 1024 launch (3.2 ms for a layer whose HBM traffic is worth 0.9 ms).  Here a
// workgroup owns (8 RW) x 32 positions x all (<= 64) channels: each of the 8 waves RW image rows (RW x NT accumulators of 32 x 32).
// RW = 2 (16 x 32 tiles) keeps two workgroups per CU; RW = 4 (32 x 32, one workgroup per CU) measured slower: with a 9-step
// main loop the exposed first-load / store-drain latency of a lone workgroup dominates.  Same staging / tap / epilogue scheme
// as conv_halo_kernel.
// =========================================================================================================
constexpr int NTW = 32;                       // tile width (positions); height = 8 waves x RW rows
template <int NT, int RW, int EPI>
__global__ __launch_bounds__(512, RW == 2 ? 4 : 2) void conv_halo_narrow_kernel(HaloArgs a) {
  constexpr bool SR = EPI == 3;
  constexpr int BNn = 32 * NT;
  constexpr int NTH = 8 * RW;                 // RW = 2: 16 x 32 tiles, <= 128 VGPRs and < 80 KB of LDS: two workgroups per CU
  constexpr int NI = ((NTH + 2) * (NTW + 2) * 4 + 511) / 512;   // halo items per thread
  constexpr int BTILE = BNn * HROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* halo = (__bf16*)smem;
  __bf16* Bt = halo + a.halo_elems;           // 2 x [BNn][HROW]
  float* psc = (float*)(Bt + 2 * BTILE);      // [Kpad] style scales of this workgroup's sample (modulated convs; zero beyond Cin)

  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int phase = gridDim.z - 1 - blockIdx.z, n0 = blockIdx.y * BNn;
  int tile = blockIdx.x;
  if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous tiles
  const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
  const TapTable& tt = a.taps[phase];
  const int hy0 = a.hy0[phase], hx0 = a.hx0[phase], hh = a.hh[phase], hw = a.hw[phase];
  const int gy0 = ty * NTH + hy0, gx0 = tx * NTW + hx0;

  const int hvec = tid & 3;
  // (raw buffer loads as in conv_halo_kernel: scalar resource + one 32-bit byte offset per lane, zeros by the range check)
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(2u * (unsigned)(a.B * a.Hin * a.Win * a.Cin)), 0x00020000);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);
  unsigned goff[NI];
  int loff[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int hp = (tid >> 2) + k * 128;
    loff[k] = -1; goff[k] = 0xffffffffu;
    if (hp < hh * hw) {
      const int hy = hp / hw, hx = hp - hy * hw;
      const int gy = gy0 + hy, gx = gx0 + hx;
      loff[k] = hp * HROW + hvec * 8;
      if ((unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win) goff[k] = 2u * (unsigned)(((b * a.Hin + gy) * a.Win + gx) * a.Cin + hvec * 8);
    }
  }
  bf16x8 hreg[NI];
  // The style multiply happens when the halo goes to LDS, not when it is loaded: multiplying right behind the load made the
  // prefetch of the next chunk synchronous (a wait for HBM at every chunk start; modulated layers ran 10 % behind plain ones).
  // The sample's scales sit in LDS behind the weight buffers (`psc`, filled below) and are read back at store time: holding them
  // in registers across the taps costs the second workgroup per CU (148 VGPRs).
  int h_c0 = 0;
  auto halo_load = [&](int c0) {
    h_c0 = c0;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const unsigned go = c0 + hvec * 8 < a.Cin ? goff[k] : 0xffffffffu;
      hreg[k] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xres, go, __builtin_amdgcn_readfirstlane(c0 * 2), 0));
    }
  };
  auto halo_store = [&]() {
    if (a.pre) {
      const f32x4 s0 = *(const f32x4*)(psc + h_c0 + hvec * 8), s1 = *(const f32x4*)(psc + h_c0 + hvec * 8 + 4);
#pragma unroll
      for (int k = 0; k < NI; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) hreg[k][j] = (__bf16)((float)hreg[k][j] * (j < 4 ? s0[j] : s1[j - 4]));
    }
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (loff[k] >= 0) *(bf16x8*)(halo + loff[k]) = hreg[k];
  };

  // weight tile: BNn rows x 4 vectors, one item per thread (threads beyond it idle); prefetched two taps ahead
  const int brow = tid >> 2;
  const int ntaps = tt.n, nchunks = a.kc_per_tap, total = ntaps * nchunks;
  const unsigned wrow = 2u * (unsigned)((n0 + brow) * a.Kpad + hvec * 8);
  const bool bthread = brow < BNn, bvalid = bthread && n0 + brow < a.N;
  auto b_load = [&](int c, int t) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wres, bvalid ? wrow : 0xffffffffu, __builtin_amdgcn_readfirstlane(2 * (tt.wt[t] * a.N * a.Kpad + c * BK)), 0));
  };
  auto b_store = [&](int buf, const bf16x8& r) { if (bthread) *(bf16x8*)(Bt + buf * BTILE + brow * HROW + hvec * 8) = r; };

  const int lrow = lane & 31, lk = (lane >> 5) * 8;
  const int abase0 = ((wm * RW) * hw + lrow) * HROW + lk;         // + mi * hw * HROW: tile pixel (row wm*RW + mi, column lane & 31)
  const int bbase = lrow * HROW + lk;                             // + ni * 32 * HROW

  f32x16 acc[RW][NT];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int lc = 0, lt = 0;
  auto advance = [&]() { if (++lt == ntaps) { lt = 0; ++lc; } };

  if (a.pre) {
    for (int i = tid; i < a.Kpad; i += 512) psc[i] = i < a.Cin ? a.pre[(size_t)b * a.Cin + i] : 0.f;
    __syncthreads();
  }
  halo_load(0);
  halo_store();
  b_store(0, b_load(0, 0));
  advance();
  bf16x8 r0 = (1 < total) ? b_load(lc, lt) : zero_bf16x8();
  advance();
  bf16x8 r1 = zero_bf16x8();
  __syncthreads();

  int c = 0, t = 0;
  auto step = [&](int q, bf16x8& rs, bf16x8& rl) {
    if (t == 0 && c + 1 < nchunks) halo_load((c + 1) * BK);
    if (q + 2 < total) { rl = b_load(lc, lt); advance(); }
    const int toff = ((tt.dy[t] - hy0) * hw + (tt.dx[t] - hx0)) * HROW;
    const __bf16* Bc = Bt + (q & 1) * BTILE + bbase;
    const __bf16* Ab = halo + abase0 + toff;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[RW], bf[NT];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) bf[ni] = *(const bf16x8*)(Bc + ni * 32 * HROW + ks * 16);
#pragma unroll
      for (int mi = 0; mi < RW; ++mi) af[mi] = *(const bf16x8*)(Ab + mi * hw * HROW + ks * 16);
#pragma unroll
      for (int mi = 0; mi < RW; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
    if (t == ntaps - 1 && c + 1 < nchunks) {
      __syncthreads();
      halo_store();
    }
    if (q + 1 < total) b_store((q + 1) & 1, rs);
    __syncthreads();
    if (++t == ntaps) { t = 0; ++c; }
  };
  for (int q = 0; q < total; q += 2) {
    step(q, r0, r1);
    if (q + 1 < total) step(q + 1, r1, r0);
  }

  // ---- epilogue: registers -> bf16 tile [NTH*32][OROW] in LDS -> 16-byte coalesced stores -------------------------------------
  constexpr int OROW = BNn + 8, NV = BNn / 8;                       // NV vectors per output row
  constexpr int NIT = NTH * NTW * NV / 512;                         // (row, vector) items per thread
  __bf16* ot = (__bf16*)smem;
  float* colbuf = (float*)(smem + (size_t)NTH * NTW * OROW * sizeof(__bf16));
  const __bf16* side = SR ? a.xs : a.residual;
  if (SR && tid < BNn) colbuf[tid] = 0.f;
  if (EPI != 0) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = tid + k * 512;
      const int row = idx / NV, vv = idx - row * NV;
      const int py = ty * NTH + (row >> 5), px = tx * NTW + (row & 31);
      const int n = n0 + vv * 8;
      bf16x8 rr = zero_bf16x8();
      if (py < a.Hm && px < a.Wm && n < a.Cout) {
        const int oy = py * a.out_mul + (phase >> 1), ox = px * a.out_mul + (phase & 1);
        rr = (EPI == 2) ? *(const bf16x8*)(side + ((size_t)(b * (a.Hout >> 1) + (oy >> 1)) * (a.Wout >> 1) + (ox >> 1)) * a.Cout + n)
                        : *(const bf16x8*)(side + ((size_t)(b * a.Hout + oy) * a.Wout + ox) * a.Cout + n);
      }
      *(bf16x8*)(ot + row * OROW + vv * 8) = rr;
    }
    __syncthreads();
  }
  constexpr float res_scale = EPI == 2 ? 0.25f : 1.f;
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    const int nl = ni * 32 + (lane & 31);
    const int n = n0 + nl;
    const float bv = (a.bias && n < a.N) ? a.bias[n] * a.bias_scale : 0.f;
    const float pv = (a.post && n < a.Cout) ? a.post[(size_t)b * a.Cout + n] : 1.f;
    float cs = 0.f;
#pragma unroll
    for (int mi = 0; mi < RW; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * RW + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float accv = acc[mi][ni][r];
        float v = accv * pv + bv;
        v = (a.act == ACT_LRELU ? (v > 0.f ? v : v * LRELU_SLOPE) : v) * a.gain;
        if (SR) cs += accv * (float)ot[row * OROW + nl];
        else if (EPI != 0) v += res_scale * (float)ot[row * OROW + nl];
        ot[row * OROW + nl] = (__bf16)v;
      }
    if (SR) {
      cs += __shfl_xor(cs, 32, 64);
      if (lane < 32) atomicAdd(&colbuf[nl], cs);
    }
  }
  __syncthreads();
  if (SR && tid < BNn && n0 + tid < a.Cout) atomicAdd(a.gs + (size_t)b * a.Cout + n0 + tid, colbuf[tid]);
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int idx = tid + k * 512;
    const int row = idx / NV, vv = idx - row * NV;
    const int py = ty * NTH + (row >> 5), px = tx * NTW + (row & 31);
    const int n = n0 + vv * 8;
    if (py >= a.Hm || px >= a.Wm || n >= a.Cout) continue;
    const int oy = py * a.out_mul + (phase >> 1), ox = px * a.out_mul + (phase & 1);
    const size_t off = ((size_t)(b * a.Hout + oy) * a.Wout + ox) * a.Cout + n;
    bf16x8 o = *(const bf16x8*)(ot + row * OROW + vv * 8);
    if (SR && a.residual) {                                        // (see conv_halo_kernel: residual of the style-gradient epilogue)
      const bf16x8 rr = *(const bf16x8*)(a.residual + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)((float)o[j] + (float)rr[j]);
    }
    *(bf16x8*)(a.y + off) = o;
  }
}

// fp32 scratch of the split-K paths ([output pixel][Cout] partial sums met by atomics).  Grow-only, owned by the library,
// zeroed once when allocated: conv_finalize_kernel re-zeroes what it consumes, so launches need no memset.
// One scratch per DEVICE (a process drives one GPU in this package, but nothing here assumes it); the split-K protocol -- the
// finalize kernel re-zeroes what it consumed -- additionally assumes that the convolutions of one device are issued on ONE stream
// (the launch stream of the training step), which lcgan_amd guarantees.
constexpr int MAX_DEV = 16;
// The library-owned scratch buffers (split-K partials, weight-gradient slabs, prescaled operands) are reused launch after launch with
// no event tracking: correct because the launches of a device are ordered on ONE stream.  A caller that switches streams (a side
// stream, a capture stream) is ordered behind the work of the stream used before: the previous stream is drained once at the switch.
hipStream_t g_scratch_stream[MAX_DEV] = {};
bool g_scratch_stream_set[MAX_DEV] = {};
void scratch_order(int dev, hipStream_t s) {
  if (g_scratch_stream_set[dev] && g_scratch_stream[dev] != s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st == hipStreamCaptureStatusNone) hipStreamSynchronize(g_scratch_stream[dev]);
    (void)hipGetLastError();
  }
  g_scratch_stream[dev] = s; g_scratch_stream_set[dev] = true;
}
float* g_splitk_ws[MAX_DEV] = {};
size_t g_splitk_ws_bytes[MAX_DEV] = {};
float* splitk_scratch(size_t bytes, hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  scratch_order(dev, s);
  if (bytes > g_splitk_ws_bytes[dev]) {
    if (g_splitk_ws[dev]) hipFree(g_splitk_ws[dev]);
    g_splitk_ws_bytes[dev] = std::max(bytes, (size_t)8 << 20);
    if (hipMalloc((void**)&g_splitk_ws[dev], g_splitk_ws_bytes[dev]) != hipSuccess) { g_splitk_ws[dev] = nullptr; g_splitk_ws_bytes[dev] = 0; return nullptr; }
    hipMemsetAsync(g_splitk_ws[dev], 0, g_splitk_ws_bytes[dev], s);
  }
  return g_splitk_ws[dev];
}
// partial-tile slabs + arrival counters of conv_igemm8_kernel's split-K (counters: zero at allocation, left at zero by every launch)
float* g_splitk_slab[MAX_DEV] = {};
size_t g_splitk_slab_bytes[MAX_DEV] = {};
int* g_splitk_cnt[MAX_DEV] = {};
constexpr int SPLITK_MAX_TILES = 4096;
float* splitk_slab_scratch(size_t bytes, int** cnt, hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  scratch_order(dev, s);
  if (!g_splitk_cnt[dev]) {
    if (hipMalloc((void**)&g_splitk_cnt[dev], SPLITK_MAX_TILES * sizeof(int)) != hipSuccess) { g_splitk_cnt[dev] = nullptr; return nullptr; }
    hipMemsetAsync(g_splitk_cnt[dev], 0, SPLITK_MAX_TILES * sizeof(int), s);
  }
  if (bytes > g_splitk_slab_bytes[dev]) {
    if (g_splitk_slab[dev]) hipFree(g_splitk_slab[dev]);
    g_splitk_slab_bytes[dev] = std::max(bytes, (size_t)32 << 20);
    if (hipMalloc((void**)&g_splitk_slab[dev], g_splitk_slab_bytes[dev]) != hipSuccess) { g_splitk_slab[dev] = nullptr; g_splitk_slab_bytes[dev] = 0; return nullptr; }
  }
  *cnt = g_splitk_cnt[dev];
  return g_splitk_slab[dev];
}
template <typename T>
__global__ void conv_finalize_kernel(float* __restrict__ ws, T* __restrict__ y, const float* __restrict__ post,
                                     const float* __restrict__ bias, const T* __restrict__ residual,
                                     long long npix, int pix_per_sample, int Cout, int N, float bias_scale, float gain, int act,
                                     int res_half, int Wout);
template <typename T>
void launch_finalize(const ConvArgs& a, float* ws, hipStream_t s) {
  const long long npix = (long long)a.B * a.Hout * a.Wout, nthr = npix * (a.Cout / 8);
  hipLaunchKernelGGL((conv_finalize_kernel<T>), dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, ws, (T*)a.y, a.post, a.bias,
                     (const T*)a.residual, npix, a.Hout * a.Wout, a.Cout, a.N, a.bias_scale, a.gain, a.act, a.res_half, a.Wout);
}

// narrow layers (Cout <= 64), stride-1 geometries whose grid holds whole 32 x 32 tiles and fills the chip
bool try_launch_halo_narrow(const ConvArgs& c, int nphase, hipStream_t s) {
  constexpr int RW = 2, NTH = 8 * RW;
  if (c.Cout > 64 || (c.Hm % NTH) || (c.Wm % NTW)) return false;
  HaloArgs a = {};
  a.x = (const __bf16*)c.x; a.w = c.w; a.y = (__bf16*)c.y; a.pre = c.pre; a.post = c.post; a.bias = c.bias;
  a.residual = (const __bf16*)c.residual; a.res_half = c.res_half; a.dbg = g_dbg_no_atomics;
  a.xs = (const __bf16*)c.xs; a.gs = c.gs;
  a.B = c.B; a.Hin = c.Hin; a.Win = c.Win; a.Cin = c.Cin; a.Hout = c.Hout; a.Wout = c.Wout; a.Cout = c.Cout;
  a.Hm = c.Hm; a.Wm = c.Wm; a.N = c.N; a.Kpad = c.Kpad; a.kc_per_tap = c.kc_per_tap; a.out_mul = c.out_mul;
  a.tiles_x = c.Wm / NTW; a.tiles_y = c.Hm / NTH;
  a.bias_scale = c.bias_scale; a.gain = c.gain; a.act = c.act;
  const int nt = c.Cout <= 32 ? 1 : 2, bnn = 32 * nt;
  const long long wgs = (long long)c.B * a.tiles_x * a.tiles_y * cdiv(c.Cout, bnn) * nphase;
  if (wgs < g_halo_narrow_min_wgs) return false;
  int max_halo = 0;
  for (int p = 0; p < nphase; ++p) {
    a.taps[p] = c.taps[p];
    int ymin = 99, ymax = -99, xmin = 99, xmax = -99;
    for (int t = 0; t < c.taps[p].n; ++t) {
      ymin = std::min(ymin, c.taps[p].dy[t]); ymax = std::max(ymax, c.taps[p].dy[t]);
      xmin = std::min(xmin, c.taps[p].dx[t]); xmax = std::max(xmax, c.taps[p].dx[t]);
    }
    a.hy0[p] = ymin; a.hx0[p] = xmin;
    a.hh[p] = (NTH - 1) + (ymax - ymin) + 1; a.hw[p] = (NTW - 1) + (xmax - xmin) + 1;
    max_halo = std::max(max_halo, a.hh[p] * a.hw[p]);
  }
  if (max_halo > (NTH + 2) * (NTW + 2)) return false;
  a.halo_elems = max_halo * HROW;
  const size_t smem = std::max(((size_t)a.halo_elems + 2 * bnn * HROW) * sizeof(__bf16) + (size_t)c.Kpad * sizeof(float),
                               (size_t)NTH * NTW * (bnn + 8) * sizeof(__bf16) + bnn * sizeof(float));
  dim3 grid((unsigned)(c.B * a.tiles_x * a.tiles_y), cdiv(c.Cout, bnn), nphase);
#define LAUNCH_NARROW(NTT, EP)                                                                                          \
  {                                                                                                                     \
    static bool set = false;                                                                                            \
    if (!set) { hipFuncSetAttribute((const void*)conv_halo_narrow_kernel<NTT, RW, EP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
    hipLaunchKernelGGL((conv_halo_narrow_kernel<NTT, RW, EP>), grid, dim3(512), smem, s, a);                            \
  }
#define LAUNCH_NARROW_EPI(NTT)                                                                                          \
  {                                                                                                                     \
    if (a.xs) LAUNCH_NARROW(NTT, 3) else if (a.residual && a.res_half) LAUNCH_NARROW(NTT, 2)                            \
    else if (a.residual) LAUNCH_NARROW(NTT, 1) else LAUNCH_NARROW(NTT, 0)                                               \
  }
  if (nt == 1) LAUNCH_NARROW_EPI(1) else LAUNCH_NARROW_EPI(2)
#undef LAUNCH_NARROW_EPI
#undef LAUNCH_NARROW
  return true;
}

// epilogue staging of conv_halo_kernel: output tile [256][BN + 8] bf16 and three float rows (column sums, bias, demodulation)
constexpr size_t HALO_EPI_SMEM = (size_t)256 * (BN + 8) * sizeof(__bf16) + 3 * BN * sizeof(float);

// Per-sample weights with the input scales folded in: out[b][t][n][k] = bf16(w[t][n][k] * pre[b][k]).  A modulated convolution is
// y = post * conv(pre * x, W); scaling the staged INPUT costs every chunk of every workgroup an in-place LDS pass (ds_read, 8 multiplies,
// ds_write per 16-byte piece: +8 ... +16 % per launch, 1.5 ms per iteration), scaling the WEIGHTS once per launch costs one pass over
// B x |W| bytes (38 MB for a 256 x 256 x 9 layer at batch 32) and the launch then runs the plain LDS-DMA kernel.  The reference forms
// the same per-sample weights (custom_layers.py:62-64).  Rounding: one bf16 rounding of w * pre here instead of one of pre * x there.
__global__ void modulate_weights_kernel(const __bf16* __restrict__ w, const float* __restrict__ pre, __bf16* __restrict__ out,
                                        long long nvec, int Kpad, int Cin, int pre_stride) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;        // one 8-element vector of ONE sample's copy
  if (i >= nvec) return;
  const int b = blockIdx.y, kv = Kpad >> 3;
  const int k0 = (int)(i % kv) * 8;
  const bf16x8 t = *(const bf16x8*)(w + i * 8);
  const float* ps = pre + (size_t)b * pre_stride + k0;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (__bf16)((float)t[j] * (k0 + j < Cin ? ps[j] : 0.f));
  *(bf16x8*)(out + ((size_t)b * nvec + i) * 8) = o;
}
__bf16* g_wmod[MAX_DEV] = {};
size_t g_wmod_bytes[MAX_DEV] = {};
__bf16* wmod_scratch(size_t bytes, hipStream_t s) {              // grow-only, per device; written in full before every use
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  scratch_order(dev, s);
  if (bytes > g_wmod_bytes[dev]) {
    if (g_wmod[dev]) hipFree(g_wmod[dev]);
    g_wmod_bytes[dev] = std::max(bytes, (size_t)96 << 20);
    if (hipMalloc((void**)&g_wmod[dev], g_wmod_bytes[dev]) != hipSuccess) { g_wmod[dev] = nullptr; g_wmod_bytes[dev] = 0; return nullptr; }
  }
  return g_wmod[dev];
}

// host side: returns true when the halo kernel was launched for this geometry
bool try_launch_halo(const ConvArgs& c, int nphase, int in_mul, hipStream_t s) {
  if (c.Hm < HT || c.Wm < HT || c.act == ACT_TANH) return false;
  if ((long long)c.B * c.Hin * c.Win * c.Cin >= (1ll << 31) || (long long)c.B * c.Hout * c.Wout * c.Cout >= (1ll << 31)) return false;
  if (in_mul == 1 && g_halo_narrow_min_wgs > 0 && try_launch_halo_narrow(c, nphase, s)) return true;   // (writes no pooled by-product: g_pool_written stays false)
  HaloArgs a = {};
  a.x = (const __bf16*)c.x; a.w = c.w; a.y = (__bf16*)c.y; a.pre = c.pre; a.post = c.post; a.bias = c.bias;
  a.residual = (const __bf16*)c.residual; a.res_half = c.res_half; a.dbg = g_dbg_no_atomics;
  a.xs = (const __bf16*)c.xs; a.gs = c.gs;
  a.B = c.B; a.Hin = c.Hin; a.Win = c.Win; a.Cin = c.Cin; a.Hout = c.Hout; a.Wout = c.Wout; a.Cout = c.Cout;
  a.Hm = c.Hm; a.Wm = c.Wm; a.N = c.N; a.Kpad = c.Kpad; a.kc_per_tap = c.kc_per_tap; a.out_mul = c.out_mul;
  a.tiles_x = cdiv(c.Wm, HT); a.tiles_y = cdiv(c.Hm, HT);
  // pooled by-product: the full-resolution-residual epilogue (EPI == 1) of this kernel writes it (every variant below shares that epilogue)
  a.pool_out = (c.pool_out && c.residual && !c.res_half && !c.xs && nphase == 1 && c.out_mul == 1 && !((c.Hout | c.Wout) & 1)) ? (__bf16*)c.pool_out : nullptr;
  a.bias_scale = c.bias_scale; a.gain = c.gain; a.act = c.act;
  // activation sign mask: written by the transposed-accumulator epilogue of the plain (EPI == 0) variants
  // (the unmodulated LDS-DMA structures below carry the EPI 4 instantiation; every other path clears a.mask_out again)
  a.mask_out = (c.mask_out && c.act == ACT_LRELU && !c.xs && !c.residual && !c.pre && nphase == 1 && c.out_mul == 1 && (c.Cout & 31) == 0 && g_mfma16 == 0) ? c.mask_out : nullptr;
  int max_halo = 0, max_halo_elems = 0;
  for (int p = 0; p < nphase; ++p) {
    a.taps[p] = c.taps[p];
    int ymin = 99, ymax = -99, xmin = 99, xmax = -99;
    for (int t = 0; t < c.taps[p].n; ++t) {
      ymin = std::min(ymin, c.taps[p].dy[t]); ymax = std::max(ymax, c.taps[p].dy[t]);
      xmin = std::min(xmin, c.taps[p].dx[t]); xmax = std::max(xmax, c.taps[p].dx[t]);
    }
    a.hy0[p] = ymin; a.hx0[p] = xmin;
    a.hh[p] = (HT - 1) * in_mul + (ymax - ymin) + 1; a.hw[p] = (HT - 1) * in_mul + (xmax - xmin) + 1;
    max_halo = std::max(max_halo, a.hh[p] * a.hw[p]);
    max_halo_elems = std::max(max_halo_elems, a.hh[p] * ((a.hw[p] * HROW + 127) & ~127));    // rows padded to 256 B (see the kernel)
  }
  if (max_halo * 4 > (in_mul == 1 ? 3 : 9) * 512) return false;
  // a launch that cannot cover half the CUs (small local batch x low resolution: 16..64 tiles, each walking the full K =
  // 9*Cin reduction) goes to the split-K implicit GEMM instead, which spreads the reduction over ~256 workgroups
  const int halo_wgs = c.B * a.tiles_x * a.tiles_y * cdiv(c.Cout, BN) * nphase;
  a.nsplit = 1;
  if (g_use_splitk && halo_wgs < g_halo_min_wgs && c.taps[0].n * c.kc_per_tap >= 8) {
    // ... unless the launch takes the stride-1 LDS-DMA structure, which can split its input-channel range (the tile's input patch is
    // still fetched once per chunk for all taps: the generic kernel re-reads it per tap and runs into the L2 -> LDS bandwidth)
    bool can = g_halo_split > 0 && in_mul == 1 && g_halo_dma == 2 && g_mfma16 == 0 && !c.xs && c.Cin % 32 == 0 && c.Kpad == c.Cin && halo_wgs <= SPLITK_MAX_TILES;
    for (int p = 0; p < nphase; ++p) can = can && a.hh[p] <= DMA_HROWS && a.hw[p] <= DMA_HP && a.hw[p] - HT <= 2;
    if (can && c.pre) {                                           // per-sample input scales: only where they will be folded into weight copies (below)
      int wt_max = 0;
      for (int p = 0; p < nphase; ++p) for (int t = 0; t < c.taps[p].n; ++t) wt_max = std::max(wt_max, c.taps[p].wt[t]);
      can = g_halo_wmod_mb > 0 && c.pre_stride >= c.Cin && (size_t)c.B * (wt_max + 1) * c.N * c.Kpad * sizeof(__bf16) <= ((size_t)g_halo_wmod_mb << 20);
    }
    const int ns = std::min({c.kc_per_tap / 2, cdiv(g_halo_split, halo_wgs), 16});
    if (!can || ns < 2) return false;
    int* cnt = nullptr;
    float* slab = splitk_slab_scratch((size_t)halo_wgs * ns * 256 * BN * sizeof(float), &cnt, s);
    if (!slab) return false;
    a.nsplit = ns; a.slab = slab; a.cnt = cnt;
  }
  a.halo_elems = max_halo_elems;
  g_pool_written = a.pool_out != nullptr;                         // (no `return false` below this line)
  g_mask_written = false;                                         // (set where an EPI 4 instantiation is launched)
  if (a.pre && g_halo_wmod_mb > 0 && c.Cin % 32 == 0 && c.Kpad == c.Cin && c.pre_stride >= c.Cin && g_mfma16 != 1) {
    // taps of the whole prepared weight (9 for a 3x3 kernel whatever the phase structure, 1 for 1x1)
    int wt_max = 0;
    for (int p = 0; p < nphase; ++p) for (int t = 0; t < c.taps[p].n; ++t) wt_max = std::max(wt_max, c.taps[p].wt[t]);
    const size_t per = (size_t)(wt_max + 1) * c.N * c.Kpad;       // elements of one sample's copy
    const size_t bytes = (size_t)c.B * per * sizeof(__bf16);
    const bool fast = in_mul == 2 ? (nphase == 1 && c.taps[0].n == 9) : true;      // geometries whose unscaled launch takes an LDS-DMA structure below
    if (fast && bytes <= ((size_t)g_halo_wmod_mb << 20)) {
      __bf16* wm = wmod_scratch(bytes, s);
      if (wm) {
        const long long nvec = (long long)(per / 8);
        hipLaunchKernelGGL(modulate_weights_kernel, dim3((unsigned)((nvec + 255) / 256), c.B), dim3(256), 0, s, c.w, c.pre, wm, nvec, c.Kpad, c.Cin, c.pre_stride);
        a.w = wm; a.w_bstride = (long long)per; a.pre = nullptr;
      }
    }
  }
  const int NBT = (in_mul == 2 && g_mfma16 != 1) ? 4 : 2;             // stride-2 forward stages the weight tiles of two taps per step
  const size_t smem = std::max(((size_t)a.halo_elems + NBT * TILE) * sizeof(__bf16) + (size_t)c.Kpad * sizeof(float),
                               HALO_EPI_SMEM);
  a.nblocks = cdiv(c.Cout, BN);
  a.nb_group = 1;
  if (g_halo_nb_group_kb > 0) {                                   // largest divisor of nblocks whose weights fit the budget
    int ntaps_all = 0;
    for (int p = 0; p < nphase; ++p) ntaps_all += c.taps[p].n;
    const size_t wbytes = (size_t)ntaps_all * BN * c.Kpad * sizeof(__bf16);          // weights one channel block streams per tile
    for (int gsz = a.nblocks; gsz >= 1; --gsz)
      if (a.nblocks % gsz == 0 && gsz * wbytes <= (size_t)g_halo_nb_group_kb * 1024) { a.nb_group = gsz; break; }
  }
  dim3 grid(c.B * a.tiles_x * a.tiles_y * a.nblocks, a.nsplit, nphase);
  a.nph_x = 1;
  if (nphase == 4 && g_halo_phase_x) { a.nph_x = 4; grid = dim3(grid.x * 4, a.nsplit, 1); }
  {
    const int ntile = c.B * a.tiles_x * a.tiles_y;
    a.dec = {ntile, log2_exact((long long)(ntile >> 3) * a.nb_group * a.nph_x), log2_exact(a.nb_group * a.nph_x), log2_exact(a.nph_x),
             log2_exact(a.tiles_x), log2_exact(a.tiles_y), a.nblocks, a.nb_group, a.nph_x, a.tiles_x, a.tiles_y, a.dbg};
  }
#define LAUNCH_HALO(IM, MM, EP)                                                                                         \
  {                                                                                                                     \
    static bool set = false;                                                                                            \
    if (!set) { hipFuncSetAttribute((const void*)conv_halo_kernel<IM, MM, EP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
    hipLaunchKernelGGL((conv_halo_kernel<IM, MM, EP>), grid, dim3(512), smem, s, a);                                    \
  }
#define LAUNCH_HALO_EPI(IM, MM)                                                                                         \
  {                                                                                                                     \
    if (a.xs) LAUNCH_HALO(IM, MM, 3) else if (a.residual && a.res_half) LAUNCH_HALO(IM, MM, 2)                          \
    else if (a.residual) LAUNCH_HALO(IM, MM, 1) else LAUNCH_HALO(IM, MM, 0)                                             \
  }
  if (g_halo_s2dma && in_mul == 2 && nphase == 1 && g_mfma16 != 1 && (a.pre ? g_halo_s2dma >= 2 : true) && c.taps[0].n == 9 && c.Cin % 32 == 0 && c.Kpad == c.Cin &&
      (!a.pre || c.Cin <= 1024)) {
    constexpr size_t S2_SMEM = 2 * (size_t)(42 * 1024 + 9 * 4096);      // two half-chunk stages (see the kernel)
    // one stage per workgroup only pays when two workgroups share a CU: a grid that cannot give every CU two keeps the two-stage form
    const bool single = g_halo_s2dma == 4 && (long long)grid.x * grid.z >= 2 * 256;
    if (single && a.pre && c.Cin <= 512) {                     // one stage per workgroup + the sample's scales: exactly 80 KB at Cin = 512
      const size_t dsm1 = std::max(S2_SMEM / 2 + (size_t)c.Cin * sizeof(float), HALO_EPI_SMEM);
#define LAUNCH_S2SM(EP)                                                                                                 \
  {                                                                                                                     \
    static bool set = false;                                                                                            \
    if (!set) { hipFuncSetAttribute((const void*)conv_halo_kernel<2, false, EP, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
    hipLaunchKernelGGL((conv_halo_kernel<2, false, EP, 4, true>), grid, dim3(512), dsm1, s, a);                         \
  }
      if (a.xs) LAUNCH_S2SM(3) else if (a.residual && a.res_half) LAUNCH_S2SM(2) else if (a.residual) LAUNCH_S2SM(1) else LAUNCH_S2SM(0)
#undef LAUNCH_S2SM
      return true;
    }
    // two anti-phased teams per 1024-thread workgroup (see conv_s2duo_kernel): plain epilogue only, whole tiles, an even number of them
    if (g_s2duo && !a.pre && !a.xs && !a.residual && c.w_part != 0 && a.w_bstride == 0 && (c.Hm % HT) == 0 && (c.Wm % HT) == 0 && ((c.B * a.tiles_x * a.tiles_y) & 1) == 0 &&
        (g_s2duo >= 2 || (long long)c.B * a.tiles_x * a.tiles_y * a.nblocks >= 2 * 256) && a.nsplit == 1) {     // (option 26 = 2: small grids too -- tests)
      constexpr size_t DUO_SMEM = 2 * (size_t)42 * 1024 + 2 * (size_t)9 * 4096;        // 156 KB (the epilogue's 2 x 70 KB fit inside)
      const dim3 dgrid((unsigned)(c.B * a.tiles_x * a.tiles_y / 2) * a.nblocks, 1, 1);
      a.dec.ntile = c.B * a.tiles_x * a.tiles_y / 2;                          // pairs of tiles
      a.dec.sh_per = log2_exact((long long)(a.dec.ntile >> 3) * a.nb_group); a.dec.sh_grp = log2_exact(a.nb_group);
#define LAUNCH_DUO(EP)                                                                                                  \
  {                                                                                                                     \
    static bool set = false;                                                                                            \
    if (!set) { hipFuncSetAttribute((const void*)conv_s2duo_kernel<EP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
    hipLaunchKernelGGL((conv_s2duo_kernel<EP>), dgrid, dim3(1024), DUO_SMEM, s, a);                                     \
  }
      if (a.mask_out) { g_mask_written = true; LAUNCH_DUO(4) } else LAUNCH_DUO(0)
#undef LAUNCH_DUO
      return true;
    }
    if (single && !a.pre) {                                               // one stage per workgroup, two workgroups per CU
      const size_t dsm1 = (g_dbg_no_atomics & 32) ? (size_t)100 * 1024 : std::max(S2_SMEM / 2, HALO_EPI_SMEM);   // (option 3, bit 32: ONE workgroup per CU -- load and compute phases of a stage then do not overlap at all: the timing experiment behind DESIGN section 5's per-CU fill-rate model)
#define LAUNCH_S2S(EP)                                                                                                  \
  {                                                                                                                     \
    static bool set = false;                                                                                            \
    if (!set) { hipFuncSetAttribute((const void*)conv_halo_kernel<2, false, EP, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
    hipLaunchKernelGGL((conv_halo_kernel<2, false, EP, 4>), grid, dim3(512), dsm1, s, a);                               \
  }
      if (a.xs) LAUNCH_S2S(3) else if (a.residual && a.res_half) LAUNCH_S2S(2) else if (a.residual) LAUNCH_S2S(1)
      else if (a.mask_out) { g_mask_written = true; LAUNCH_S2S(4) } else LAUNCH_S2S(0)
#undef LAUNCH_S2S
      return true;
    }
    const size_t dsmem = std::max(S2_SMEM + (a.pre ? (size_t)c.Cin * sizeof(float) : 0), HALO_EPI_SMEM);
#define LAUNCH_S2M(EP)                                                                                                  \
  {                                                                                                                     \
    static bool set = false;                                                                                            \
    if (!set) { hipFuncSetAttribute((const void*)conv_halo_kernel<2, false, EP, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
    hipLaunchKernelGGL((conv_halo_kernel<2, false, EP, 3, true>), grid, dim3(512), dsmem, s, a);                        \
  }
    if (a.pre) {
      if (a.xs) LAUNCH_S2M(3) else if (a.residual && a.res_half) LAUNCH_S2M(2) else if (a.residual) LAUNCH_S2M(1) else LAUNCH_S2M(0)
      return true;
    }
#undef LAUNCH_S2M
#define LAUNCH_S2(EP)                                                                                                   \
  {                                                                                                                     \
    static bool set = false;                                                                                            \
    if (!set) { hipFuncSetAttribute((const void*)conv_halo_kernel<2, false, EP, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
    hipLaunchKernelGGL((conv_halo_kernel<2, false, EP, 3>), grid, dim3(512), dsmem, s, a);                              \
  }
    if (a.xs) LAUNCH_S2(3) else if (a.residual && a.res_half) LAUNCH_S2(2) else if (a.residual) LAUNCH_S2(1)
    else if (a.mask_out) { g_mask_written = true; LAUNCH_S2(4) } else LAUNCH_S2(0)
#undef LAUNCH_S2
    return true;
  }
  const bool mod = a.pre != nullptr;
  bool dma_ok = g_halo_dma && in_mul == 1 && g_mfma16 != 1 && (mod ? g_halo_dma_mod != 0 : true) && c.Cin % 32 == 0 && c.Kpad == c.Cin;
  for (int p = 0; p < nphase; ++p) dma_ok = dma_ok && a.hh[p] <= DMA_HROWS && a.hw[p] <= DMA_HP && a.hw[p] - HT <= 2;
  if (dma_ok) {
    const int tp = mod ? (g_halo_dma_mod == 2 ? 2 : 1) : (g_halo_dma == 2 ? 2 : 1);
    // (option 3, bit 64: ONE workgroup per CU -- the timing experiment behind DESIGN section 5's lone-workgroup figure)
    const size_t dsmem = (g_dbg_no_atomics & 64) ? (size_t)100 * 1024
                                                 : std::max((size_t)(2 * DMA_HBUF + 2 * tp * DMA_BBUF) + (mod ? (size_t)c.Cin * sizeof(float) : 0), HALO_EPI_SMEM);
#define LAUNCH_DMA(EP, TPV, MD)                                                                                         \
  {                                                                                                                     \
    static bool set = false;                                                                                            \
    if (!set) { hipFuncSetAttribute((const void*)conv_halo_kernel<1, false, EP, TPV, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
    hipLaunchKernelGGL((conv_halo_kernel<1, false, EP, TPV, MD>), grid, dim3(512), dsmem, s, a);                        \
  }
#define LAUNCH_DMA_EPI(TPV)                                                                                             \
  { if (a.xs) LAUNCH_DMA(3, TPV, false) else if (a.residual && a.res_half) LAUNCH_DMA(2, TPV, false) else if (a.residual) LAUNCH_DMA(1, TPV, false) \
    else if (a.mask_out) { g_mask_written = true; LAUNCH_DMA(4, TPV, false) } else LAUNCH_DMA(0, TPV, false) }
#define LAUNCH_DMA_MOD(TPV)                                                                                             \
  { if (a.xs) LAUNCH_DMA(3, TPV, true) else if (a.residual && a.res_half) LAUNCH_DMA(2, TPV, true)                      \
    else if (a.residual) LAUNCH_DMA(1, TPV, true) else LAUNCH_DMA(0, TPV, true) }
#define LAUNCH_DMA16(EP)                                                                                               \
  {                                                                                                                     \
    static bool set = false;                                                                                            \
    if (!set) { hipFuncSetAttribute((const void*)conv_halo_kernel<1, true, EP, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
    hipLaunchKernelGGL((conv_halo_kernel<1, true, EP, 2, false>), grid, dim3(512), dsmem, s, a);                        \
  }
    if (g_mfma16 == 2 && !mod && !a.xs && tp == 2) { a.mask_out = nullptr; if (a.residual && a.res_half) LAUNCH_DMA16(2) else if (a.residual) LAUNCH_DMA16(1) else LAUNCH_DMA16(0) }
    else if (mod) { if (tp == 2) LAUNCH_DMA_MOD(2) else LAUNCH_DMA_MOD(1) }
    else if (tp == 2) LAUNCH_DMA_EPI(2) else LAUNCH_DMA_EPI(1)
#undef LAUNCH_DMA16
#undef LAUNCH_DMA_MOD
#undef LAUNCH_DMA_EPI
#undef LAUNCH_DMA
    return true;
  }
  a.mask_out = nullptr;                                          // (register-staged structures: no mask instantiation)
  if (in_mul == 1 && g_mfma16 == 1 && !a.xs && !a.residual) LAUNCH_HALO(1, true, 0)
  else if (in_mul == 1) LAUNCH_HALO_EPI(1, false)
  else if (g_mfma16 == 1 && !a.xs && !a.residual) LAUNCH_HALO(2, true, 0)
  else LAUNCH_HALO_EPI(2, false)
#undef LAUNCH_HALO_EPI
#undef LAUNCH_HALO
  return true;
}

// =========================================================================================================
// weight gradient
// =========================================================================================================
constexpr int WG_ROW = 160;                 // bf16 per staged position row: 128 channels + 32 pad (320 B stride)
constexpr int WG_TILE = 32 * WG_ROW;

struct WgradArgs {
  const void* x; const void* g; float* gwp;
  const float* pre_x; const float* pre_g;
  int B, Hx, Wx, Cx, Hm, Wm, Cg, A, Bc, M;
  int stride, k, pad;
  int chunks_per_split, nsplit, nchunks;
  int parts;                                 // wgrad3: split = group * parts + part
  int cps_group;                             // wgrad3: chunks per group (group = one sample when per-sample scales exist, else the whole batch)
  float* slab;                               // wgrad3: non-null = every split stores its partial tile to slab[split][tap][A][Bc] (plain stores) instead of atomics
  int slab_bf16;                             // ... as bf16 elements (the slab pointer is then a __bf16*): half the partial-tile traffic of a bf16 launch
  int xcd_order, na, nc;                     // wgrad3: 1-D XCD-aware workgroup order (see WG3_INDEX); a / c blocks of 128  // wgrad3, single split, fused un-prep: the epilogue writes the gradient in WEIGHT layout itself (no gwp clear, no atomics, no second launch):
  // dgw[a][b][t] = dscale * acc + 2 dscale^2 dw[a][b][t] dgwsq[a][b]  (see unprep_wgrad_kernel; dtransposed: the weight is [Bc][A])
  float* dgw; const float* dw; const float* dgwsq; float dscale; int dtransposed;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int row0, int col) {
  // 8 reduction rows (row0 .. row0+7) x this lane's column, via two transposed 4x16 block reads.
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + row0 * WG_ROW + col));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + (row0 + 4) * WG_ROW + col));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <typename T, int P>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 2 * P;                            // G parts 0..P-1, then X parts 0..P-1
  __bf16* lds = (__bf16*)smem;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int a0 = blockIdx.x * 128, c0 = blockIdx.y * 128;
  const int tap = blockIdx.z / a.nsplit, split = blockIdx.z - tap * a.nsplit;
  const int ky = tap / a.k, kx = tap - ky * a.k;
  const int q_begin = split * a.chunks_per_split;
  const int q_end = min(q_begin + a.chunks_per_split, a.nchunks);
  if (q_begin >= q_end) return;                                  // uniform per block
  const int HWm = a.Hm * a.Wm;
  const T* __restrict__ x = (const T*)a.x;
  const T* __restrict__ g = (const T*)a.g;

  const int lpos = tid >> 4, lvec = tid & 15;                    // positions lpos, lpos+16; channels lvec*8
  F8 rg[2], rx[2];

  auto gload = [&](int q) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = q * 32 + lpos + 16 * i;
      rg[i] = f8_zero(); rx[i] = f8_zero();
      if (m < a.M) {
        const int b = m / HWm, rem = m - b * HWm;
        const int iy = rem / a.Wm, ix = rem - iy * a.Wm;
        const int ca = a0 + lvec * 8;
        if (ca < a.Cg) {
          rg[i] = Feat<T>::load(g + ((size_t)(b * a.Hm + iy) * a.Wm + ix) * a.Cg + ca);
          if (a.pre_g) {
            const float* ps = a.pre_g + (size_t)b * a.Cg + ca;
#pragma unroll
            for (int j = 0; j < 8; ++j) rg[i].v[j] *= ps[j];
          }
        }
        const int yy = iy * a.stride + ky - a.pad, xx = ix * a.stride + kx - a.pad;
        const int cc = c0 + lvec * 8;
        if ((unsigned)yy < (unsigned)a.Hx && (unsigned)xx < (unsigned)a.Wx && cc < a.Cx) {
          rx[i] = Feat<T>::load(x + ((size_t)(b * a.Hx + yy) * a.Wx + xx) * a.Cx + cc);
          if (a.pre_x) {
            const float* ps = a.pre_x + (size_t)b * a.Cx + cc;
#pragma unroll
            for (int j = 0; j < 8; ++j) rx[i].v[j] *= ps[j];
          }
        }
      }
    }
  };
  auto sstore = [&](int buf) {
    __bf16* base = lds + buf * NT * WG_TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = (lpos + 16 * i) * WG_ROW + lvec * 8;
      F8 fg = rg[i], fx = rx[i];
#pragma unroll
      for (int pp = 0; pp < P; ++pp) {
        *(bf16x8*)(base + pp * WG_TILE + o) = peel_bf16x8(fg);
        *(bf16x8*)(base + (P + pp) * WG_TILE + o) = peel_bf16x8(fx);
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read lane geometry (ds_read_b64_tr_b16): within each 16-lane group lane 4q+p addresses
  // row q, columns 4p..4p+3 of a 4x16 block and lane i receives column i.
  const int g16 = lane >> 4, i16 = lane & 15;
  const int trow = 8 * (g16 >> 1) + (i16 >> 2);                  // + 16*ks (+4 for the second read)
  const int tcol = 16 * (g16 & 1) + 4 * (i16 & 3);               // + sub-tile column base

  auto compute = [&](int buf) {
    const __bf16* G = lds + buf * NT * WG_TILE;
    const __bf16* X = G + P * WG_TILE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[P][2], bf[P][2];
#pragma unroll
      for (int pp = 0; pp < P; ++pp) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[pp][mi] = tr_frag(G + pp * WG_TILE, ks * 16 + trow, wm * 64 + mi * 32 + tcol);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bf[pp][ni] = tr_frag(X + pp * WG_TILE, ks * 16 + trow, wn * 64 + ni * 32 + tcol);
      }
#pragma unroll
      for (int sum = P - 1; sum >= 0; --sum)
#pragma unroll
        for (int i = 0; i <= sum; ++i)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][mi], bf[sum - i][ni], acc[mi][ni], 0, 0, 0);
    }
  };

  gload(q_begin);
  sstore(0);
  __syncthreads();
  for (int q = q_begin; q < q_end; ++q) {
    const int cur = (q - q_begin) & 1;
    if (q + 1 < q_end) gload(q + 1);
    compute(cur);
    if (q + 1 < q_end) sstore(cur ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int cc = c0 + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int aa = a0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (aa >= a.A || cc >= a.Bc) continue;
        if (a.dgw) {                                             // single split + fused un-prep: the finished gradient in weight layout (see WgradArgs)
          const size_t ab = a.dtransposed ? (size_t)cc * a.A + aa : (size_t)aa * a.Bc + cc;
          const size_t idx = ab * (a.k * a.k) + tap;
          float v = acc[mi][ni][r] * a.dscale;
          if (a.dgwsq) v += 2.f * a.dscale * a.dscale * a.dw[idx] * a.dgwsq[ab];
          a.dgw[idx] = v;
        } else {
          atomicAdd(a.gwp + ((size_t)tap * a.A + aa) * a.Bc + cc, acc[mi][ni][r]);
        }
      }
    }
}

// =========================================================================================================
// weight gradient, bf16 fast path for 3x3 kernels: one workgroup owns a whole kernel ROW (ky; kx = 0,1,2).
// A reduction chunk is a 32-position row segment of the output-side grid, so
//   * the output-gradient tile G [32 pos][128 a] is staged once for 3 taps,
//   * the three shifted input windows come from ONE halo row segment X [(32*STRIDE + 2) px][128 c] in LDS (tap kx reads
//     rows p*STRIDE + kx through the transposed LDS read),
//   * (sample, row, segment) decoding is scalar, and chunks whose input row is padding are skipped.
// =========================================================================================================
// Workgroup -> (a block, c block, kernel row ky, split).  Every workgroup of one split reads the same chunk range (G rows for its a
// block, X rows for its c block and ky), so they should share an L2.  With the 3-D grid (blockIdx.z = split x kernel row, dispatched
// round-robin over the 8 XCDs) the three kernel-row siblings of a position range sat on three different XCDs and every operand byte
// crossed the fabric 3-5 times (rocprofv3 FETCH_SIZE per dispatch, scripts/micro_wgrad_xcd.py: 3.2 GB for the 1.07 GB of operands of the
// 256 x 256 layer).  XCD order (1-D grid): XCD i takes splits i, i + 8, ... with all (a, c, ky) workgroups of a split consecutive on it
// -- 1.0-1.7x the operand bytes, -10 ... -15 % on the 256 x 256 launches.  Splits beyond the last whole group of 8 are dealt tile by
// tile over all XCDs (their siblings do not share an L2, but no XCD is left with a longer queue than another: the first version put
// whole splits only, and the plan had to pick split counts that are multiples of 8 x rounds -- 3x the partial-tile traffic on the
// 512-channel layers, where it lost 10 %).
#define WG3_INDEX(NKXV)                                                                                               \
  int a0, c0, split, ky;                                                                                              \
  if (a.xcd_order) {                                                                                                  \
    const int tiles = a.na * a.nc * (NKXV), xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;                             \
    const int nfull8 = a.nsplit >> 3, grouped = nfull8 * tiles;                                                       \
    int t;                                                                                                            \
    if (slot < grouped) {                                                                                             \
      const int sl = slot / tiles;                                                                                    \
      t = slot - sl * tiles;                                                                                          \
      split = sl * 8 + xcd;                                                                                           \
    } else {                                                                                                          \
      const int r = (slot - grouped) * 8 + xcd, sr = r / tiles;                                                       \
      t = r - sr * tiles;                                                                                             \
      split = nfull8 * 8 + sr;                                                                                        \
      if (split >= a.nsplit) return;                                                                                  \
    }                                                                                                                 \
    a0 = (t % a.na) * 128; c0 = ((t / a.na) % a.nc) * 128; ky = t / (a.na * a.nc);                                    \
  } else {                                                                                                            \
    a0 = blockIdx.x * 128; c0 = blockIdx.y * 128;                                                                     \
    split = blockIdx.z / (NKXV); ky = blockIdx.z - split * (NKXV);                                                    \
  }

template <int STRIDE>
__device__ __forceinline__ bf16x8 tr_frag_rows(const __bf16* tile, int row_a, int row_b, int col) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + row_a * WG_ROW + col));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + row_b * WG_ROW + col));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// SEG   = positions per chunk (64 or 32) = ROWS image rows x SEGW columns (SEGW = min(SEG, grid width): narrow layers take
//         several rows per chunk);   NKX = 3 (3x3 kernel: one kernel row per workgroup) or 1 (1x1 kernel)
// PK > 1 (narrow layers, <= 128 / PK channels on both sides: the C = 32 / 64 octaves of the high-resolution networks): the 128
// staged columns hold PK GROUPS of channels, group g carrying image row row0 + g, so one chunk moves PK x 64 positions through
// the same staging / barrier / fragment pipeline (its cost per chunk does not depend on how many columns are real: the 32 -> 32
// layer at 1024 x 1024 took 4.9 ms for 0.9 ms worth of HBM traffic).  The 128 x 128 product then holds PK x PK blocks of which
// only the diagonal ones (same rows on both sides) are weight-gradient terms; they are summed by the epilogue's atomics.
template <int STRIDE, int SEG, int SEGW, int NKX, int PK>
__global__ __launch_bounds__(512) void conv_wgrad3_kernel(WgradArgs a) {
  static_assert(PK == 1 || (SEG == 64 && SEGW == 64), "packed groups: one image row per group");
  constexpr int CG = 128 / PK, VG = 16 / PK;                 // channels / 8-channel vectors per group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PAD = NKX == 3 ? 1 : 0;
  constexpr int ROWS = SEG / SEGW;                           // image rows per chunk
  constexpr int XW = SEGW * STRIDE + 2 * PAD;                // halo pixels per image row
  constexpr int XR = ROWS * XW;                              // halo pixels per chunk
  constexpr int NG = SEG / 32;                               // G items (position, vec) per thread
  constexpr int NX = (XR * 16 + 511) / 512;                  // X items (pixel, vec) per thread (512 threads)
  constexpr int STAGE = (SEG + XR) * WG_ROW;                 // elements per stage: G then X
  __bf16* lds = (__bf16*)smem;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;                    // 8 waves: 2 (a) x 4 (c), wave tile 64 x 32
  WG3_INDEX(NKX)
  const int segs = a.Wm / SEGW;                              // segments per image row (1 for narrow layers)
  const int rgroups = a.Hm / (ROWS * PK);                    // row groups per sample
  // split = (group, part).  With per-sample style / demod scales a group is ONE sample, so the scales can be applied once to
  // the fp32 accumulator in the epilogue instead of to every staged operand vector; without scales (discriminator convs) the
  // whole batch is one group and a workgroup's chunk range may span samples (low-resolution layers: few chunks per sample).
  const int bsmp = split / a.parts, part = split - bsmp * a.parts;
  const int cps = a.cps_group;
  const int q_begin = bsmp * cps + part * a.chunks_per_split;
  const int q_end = min(q_begin + a.chunks_per_split, (bsmp + 1) * cps);
  if (q_begin >= q_end) return;
  const __bf16* __restrict__ x = (const __bf16*)a.x;
  const __bf16* __restrict__ g = (const __bf16*)a.g;
  const int lpos = tid >> 4, lvec = tid & 15;
  const int lgrp = PK == 1 ? 0 : lvec / VG, cvec = PK == 1 ? lvec : lvec - lgrp * VG;   // this thread's group (image row) and vector in it
  const bool ga_ok = a0 + cvec * 8 < a.Cg, xc_ok = c0 + cvec * 8 < a.Cx;

  struct Stage { bf16x8 g[NG]; bf16x8 x[NX]; };

  // Global loads are raw buffer loads (resource in scalar registers, one 32-bit byte offset per lane): the part of an item's
  // offset that belongs to the thread is computed once, the chunk adds one uniform base, and a lane outside the image (padding
  // rows / columns) or beyond the channels gets offset 0xffffffff, which the range check turns into zeros -- no branch and no
  // 64-bit address per load.  Tensors are < 2^31 elements (checked by the caller).
  const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (int)(2u * (unsigned)(a.B * a.Hm * a.Wm * a.Cg)), 0x00020000);
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(2u * (unsigned)(a.B * a.Hx * a.Wx * a.Cx)), 0x00020000);
  int goffs[NG], xoffs[NX], xhr[NX], xhc[NX];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int p = lpos + 32 * i, pr = p / SEGW, pc = p - pr * SEGW;
    goffs[i] = 2 * (((lgrp + pr) * a.Wm + pc) * a.Cg + a0 + cvec * 8);
  }
#pragma unroll
  for (int k = 0; k < NX; ++k) {
    const int r = lpos + 32 * k, hr = r / XW, hc = r - hr * XW;
    xhr[k] = r < XR && xc_ok ? (lgrp + hr) * STRIDE + ky - PAD : -0x40000000;      // row / column relative to the chunk origin
    xhc[k] = hc - PAD;
    xoffs[k] = 2 * ((xhr[k] * a.Wx + xhc[k]) * a.Cx + c0 + cvec * 8);
  }

  auto gload = [&](int q, Stage& st) {
    const int seg = q % segs, t = q / segs;
    const int row0 = (t % rgroups) * ROWS * PK, b = t / rgroups, j0 = seg * SEGW;
    const int gbase = 2 * (((b * a.Hm + row0) * a.Wm + j0) * a.Cg);
#pragma unroll
    for (int i = 0; i < NG; ++i)
      st.g[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(gres, ga_ok ? (unsigned)(gbase + goffs[i]) : 0xffffffffu, 0, 0));
    const int y0 = row0 * STRIDE, x0 = j0 * STRIDE;
    const int xbase = 2 * (((b * a.Hx + y0) * a.Wx + x0) * a.Cx);
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const bool ok = (unsigned)(y0 + xhr[k]) < (unsigned)a.Hx && (unsigned)(x0 + xhc[k]) < (unsigned)a.Wx;
      st.x[k] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xres, ok ? (unsigned)(xbase + xoffs[k]) : 0xffffffffu, 0, 0));
    }
  };
  auto sstore = [&](int buf, const Stage& st) {
    __bf16* G = lds + buf * STAGE;
    __bf16* X = G + SEG * WG_ROW;
#pragma unroll
    for (int i = 0; i < NG; ++i) *(bf16x8*)(G + (lpos + 32 * i) * WG_ROW + lvec * 8) = st.g[i];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int r = lpos + 32 * k;
      if (r < XR) *(bf16x8*)(X + r * WG_ROW + lvec * 8) = st.x[k];
    }
  };

  f32x16 acc[NKX][2];
#pragma unroll
  for (int t = 0; t < NKX; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;

  const int g16 = lane >> 4, i16 = lane & 15;
  const int trow = 8 * (g16 >> 1) + (i16 >> 2);
  const int tcol = 16 * (g16 & 1) + 4 * (i16 & 3);
  // halo row of position p for tap kx = xrow(p) + kx
  auto xrow = [&](int p) { const int pr = p / SEGW; return pr * XW + (p - pr * SEGW) * STRIDE; };

  auto compute = [&](int buf) {
    const __bf16* G = lds + buf * STAGE;
    const __bf16* X = G + SEG * WG_ROW;
#pragma unroll
    for (int ks = 0; ks < SEG / 16; ++ks) {
      const int p = ks * 16 + trow;                          // position (within the chunk) of this lane's first row block
      bf16x8 af[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) af[mi] = tr_frag_rows<1>(G, p, p + 4, wm * 64 + mi * 32 + tcol);
      const int xa = xrow(p), xb = xrow(p + 4);
#pragma unroll
      for (int kx = 0; kx < NKX; ++kx) {
        const bf16x8 bf = tr_frag_rows<STRIDE>(X, xa + kx, xb + kx, wn * 32 + tcol);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[kx][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf, acc[kx][mi], 0, 0, 0);
      }
    }
  };

  // software pipeline: chunk q computes from LDS buffer (q - q_begin) & 1 while chunk q+1 sits in one register stage and the
  // loads of chunk q+2 are in flight in the other (two named stages -> static register indexing)
  Stage s0, s1;
  gload(q_begin, s0);
  sstore(0, s0);
  if (q_begin + 1 < q_end) gload(q_begin + 1, s0);
  __syncthreads();
  auto step = [&](int q, Stage& rs, Stage& rl) {
    if (q + 2 < q_end) gload(q + 2, rl);
    compute((q - q_begin) & 1);
    if (q + 1 < q_end) sstore(((q - q_begin) & 1) ^ 1, rs);
    __syncthreads();
  };
  for (int q = q_begin; q < q_end; q += 2) {
    step(q, s0, s1);
    if (q + 1 < q_end) step(q + 1, s1, s0);
  }

  if (PK == 1 && a.dgw) {                                    // single split: finished gradient straight to weight layout
    const int cc = c0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int aa = a0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (aa >= a.A || cc >= a.Bc) continue;
        const size_t ab = a.dtransposed ? (size_t)cc * a.A + aa : (size_t)aa * a.Bc + cc;
        const size_t base = ab * (NKX * NKX) + ky * NKX;
        float v[NKX];
#pragma unroll
        for (int kx = 0; kx < NKX; ++kx) v[kx] = acc[kx][mi][r] * a.dscale;
        if (a.dgwsq) {
          const float f = 2.f * a.dscale * a.dscale * a.dgwsq[ab];
#pragma unroll
          for (int kx = 0; kx < NKX; ++kx) v[kx] += f * a.dw[base + kx];
        }
#pragma unroll
        for (int kx = 0; kx < NKX; ++kx) a.dgw[base + kx] = v[kx];
      }
    return;
  }
#pragma unroll
  for (int kx = 0; kx < NKX; ++kx)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      // packed groups: only the diagonal (group, group) blocks are weight-gradient terms; columns fold back to channel indices
      if (PK > 1 && (wm * 64 + mi * 32) / CG != (wn * 32) / CG) continue;
      const int cc = PK == 1 ? c0 + wn * 32 + (lane & 31) : (wn * 32) % CG + (lane & 31);
      const float sxv = (a.pre_x && cc < a.Cx) ? a.pre_x[(size_t)bsmp * a.Cx + cc] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int arow = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int aa = PK == 1 ? a0 + wm * 64 + arow : (wm * 64 + arow) % CG;
        if (aa < a.A && cc < a.Bc) {
          const float sgv = a.pre_g ? a.pre_g[(size_t)bsmp * a.Cg + aa] : 1.f;
          const size_t off = ((size_t)(ky * NKX + kx) * a.A + aa) * a.Bc + cc;
          if (a.slab) {
            const size_t si = (size_t)split * (NKX * NKX) * a.A * a.Bc + off;
            if (a.slab_bf16) ((__bf16*)a.slab)[si] = (__bf16)(acc[kx][mi][r] * sxv * sgv); else a.slab[si] = acc[kx][mi][r] * sxv * sgv;
          }
          else atomicAdd(a.gwp + off, acc[kx][mi][r] * sxv * sgv);
        }
      }
    }
}

// =========================================================================================================
// conv_wgrad3_dma_kernel: the row-segment weight-gradient kernel for its main case (3x3, 64-position row segments, no packed
// groups) with both operand tiles staged by LDS-DMA (`buffer_load ... lds`): no staging registers (the register-staged kernel
// holds two stages = 40 VGPRs and sits at 200+), so at stride 1 TWO workgroups share a CU (<= 128 VGPRs, 2 x 33 KB of LDS each).
// LDS image: one 256-byte record per position / halo pixel (128 channels), unpadded because a DMA piece is 64 lanes x 16 B =
// 4 records; the sixteen 16-byte slots of record r hold channel chunk  slot ^ (4 * f(r))  (swizzle through the SOURCE address),
// f(r) = r & 3 for G and (r >> (STRIDE - 1)) & 3 for X, so the four rows a transposed 4 x 16 block read touches sit in four
// different bank groups, exactly as the 320-byte padded pitch of the register-staged kernel arranged.
// =========================================================================================================
// positions per chunk: 64 at stride 1; 32 at stride 2, where a 64-position chunk (16 + 33 KB per stage) would leave one workgroup per CU
template <int STRIDE, bool HALF = false>                     // HALF: 32-position chunks at stride 1 too (32-wide grids)
__global__ __launch_bounds__(512, 4) void conv_wgrad3_dma_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SEG = (STRIDE == 1 && !HALF) ? 64 : 32, NKX = 3;
  constexpr int XW = SEG * STRIDE + 2;                       // halo pixels of a row segment
  constexpr int GP = SEG / 4, XP = (XW + 3) / 4;             // 1-KB DMA pieces (4 records each) of G and X
  constexpr int NXI = (XP + 7) / 8;                          // X pieces per wave
  constexpr int NGI = (GP + 7) / 8;                          // G pieces per wave
  constexpr int STAGE_B = (GP + XP) * 1024;                  // bytes per stage
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, widu = __builtin_amdgcn_readfirstlane(wid);
  const int wm = wid >> 2, wn = wid & 3;                    // 8 waves: 2 (a) x 4 (c), wave tile 64 x 32
  WG3_INDEX(NKX)
  const int segs = a.Wm / SEG, rgroups = a.Hm;
  const int bsmp = split / a.parts, part = split - bsmp * a.parts;
  const int cps = a.cps_group;
  const int q_begin = bsmp * cps + part * a.chunks_per_split;
  const int q_end = min(q_begin + a.chunks_per_split, (bsmp + 1) * cps);
  if (q_begin >= q_end) return;
  const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc((void*)a.g, 0, (int)(2u * (unsigned)(a.B * a.Hm * a.Wm * a.Cg)), 0x00020000);
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(2u * (unsigned)(a.B * a.Hx * a.Wx * a.Cx)), 0x00020000);

  // ---- DMA items of this lane: record lrec = lane >> 4 of a piece, slot = lane & 15 ----------------------------------------
  const int lrec = lane >> 4, slot = lane & 15;
  unsigned goffs[2];                                         // G piece i = widu + 8 k : positions 4 i .. 4 i + 3
#pragma unroll
  for (int k = 0; k < NGI; ++k) {
    const int pos = 4 * (widu + 8 * k) + lrec;
    const int ch = (slot ^ ((pos & 3) << 2)) * 8;
    goffs[k] = a0 + ch < a.Cg ? 2u * (unsigned)(pos * a.Cg + a0 + ch) : 0xffffffffu;
  }
  int xoffs[NXI], xhc[NXI];                                  // X piece i = widu + 8 k : halo pixels 4 i .. 4 i + 3
#pragma unroll
  for (int k = 0; k < NXI; ++k) {
    const int hc = 4 * (widu + 8 * k) + lrec;
    const int ch = (slot ^ (((hc >> (STRIDE - 1)) & 3) << 2)) * 8;
    xhc[k] = (hc < XW && c0 + ch < a.Cx) ? hc - 1 : -0x40000000;     // column relative to the segment origin (padding: -1)
    xoffs[k] = 2 * ((hc - 1) * a.Cx + c0 + ch);
  }
  // chunk q -> (sample b, image row row0, segment seg): decoded ONCE; the loop below walks the chunks in order and steps the three
  // counters (the four integer divisions per chunk were most of the ~570 cycles a wave spent issuing a chunk's DMA: scripts/wgrad_stamps.py)
  int d_seg = q_begin % segs, d_row = (q_begin / segs) % rgroups, d_b = (q_begin / segs) / rgroups;
  auto dma = [&](int q, int buf) {
    (void)q;
    const int seg = d_seg, row0 = d_row, b = d_b, j0 = seg * SEG;
    if (++d_seg == segs) { d_seg = 0; if (++d_row == rgroups) { d_row = 0; ++d_b; } }
    char* G = smem + buf * STAGE_B;
    const int gbase = 2 * (((b * a.Hm + row0) * a.Wm + j0) * a.Cg);
#pragma unroll
    for (int k = 0; k < NGI; ++k)
      if (widu + 8 * k < GP)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(gres, (lds_void*)(G + (widu + 8 * k) * 1024), 16, goffs[k], __builtin_amdgcn_readfirstlane(gbase), 0, 0);
    const int yy = row0 * STRIDE + ky - 1, x0 = j0 * STRIDE;
    const bool yok = (unsigned)yy < (unsigned)a.Hx;
    const int xbase = 2 * (((b * a.Hx + yy) * a.Wx + x0) * a.Cx);
#pragma unroll
    for (int k = 0; k < NXI; ++k) {
      if (widu + 8 * k >= XP) continue;
      const bool ok = yok && (unsigned)(x0 + xhc[k]) < (unsigned)a.Wx;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(G + (GP + widu + 8 * k) * 1024), 16, ok ? (unsigned)(xbase + xoffs[k]) : 0xffffffffu, 0, 0, 0);
    }
  };

  f32x16 acc[NKX][2];
#pragma unroll
  for (int t = 0; t < NKX; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;

  // ---- fragment byte addresses (transposed 4 x 16 block reads; see tr_frag_rows) ------------------------------------------
  const int g16 = lane >> 4, i16 = lane & 15;
  const int trow = 8 * (g16 >> 1) + (i16 >> 2);
  const int tcol = 16 * (g16 & 1) + 4 * (i16 & 3);
  const int rsw = i16 >> 2;                                  // (row & 3) of this lane's G rows: trow + 16 ks (+ 4)
  int gaddr[2], xaddr[NKX];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) gaddr[mi] = trow * 256 + 2 * ((wm * 64 + mi * 32 + tcol) ^ (rsw << 5));
#pragma unroll
  for (int kx = 0; kx < NKX; ++kx) {
    const int r = trow * STRIDE + kx;                        // + 16 STRIDE ks (+ 4 STRIDE): the swizzle term does not change
    xaddr[kx] = GP * 1024 + r * 256 + 2 * ((wn * 32 + tcol) ^ (((r >> (STRIDE - 1)) & 3) << 5));
  }
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto tr_pair = [&](const char* base, int addr, int rowb_bytes) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr + rowb_bytes));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto compute = [&](int buf) {
    const char* S = smem + buf * STAGE_B;
#pragma unroll
    for (int ks = 0; ks < SEG / 16; ++ks) {
      bf16x8 af[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) af[mi] = tr_pair(S, gaddr[mi] + ks * 16 * 256, 4 * 256);
#pragma unroll
      for (int kx = 0; kx < NKX; ++kx) {
        const bf16x8 bf = tr_pair(S, xaddr[kx] + ks * 16 * STRIDE * 256, 4 * STRIDE * 256);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[kx][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf, acc[kx][mi], 0, 0, 0);
      }
    }
  };

  dma(q_begin, 0);
  __syncthreads();
#ifdef HALO_STAMPS
  // diagnostic build: per wave, cycle sums of a chunk's segments -- [DMA issue] [fragment reads + 24 MFMAs issued] [vmcnt wait] [barrier wait]
  unsigned long long wsA = 0, wsB = 0, wsC = 0, wsD = 0, wt0 = 0, wt1 = 0, wt2 = 0, wt3 = 0, wt4 = 0;
  auto chunk = [&](int q, int nb, int cb) {
    STAMP(wt0)
    if (q + 1 < q_end) dma(q + 1, nb);
    STAMP(wt1)
    compute(cb);
    STAMP(wt2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(wt3)
    __syncthreads();
    STAMP(wt4)
    wsA += wt1 - wt0; wsB += wt2 - wt1; wsC += wt3 - wt2; wsD += wt4 - wt3;
  };
  for (int q = q_begin; q < q_end; q += 2) {
    chunk(q, 1, 0);
    if (q + 1 < q_end) chunk(q + 1, 0, 1);
  }
  if (lane == 0 && blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z) < 2048) {
    unsigned long long* o = g_halo_stamps + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + wid) * 5;
    o[0] = wsA; o[1] = wsB; o[2] = wsC; o[3] = wsD; o[4] = q_end - q_begin;
  }
#else
  for (int q = q_begin; q < q_end; q += 2) {
    if (q + 1 < q_end) dma(q + 1, 1);
    compute(0);
    __syncthreads();                                         // (vmcnt(0): the next chunk has landed; then the barrier)
    if (q + 1 < q_end) {
      if (q + 2 < q_end) dma(q + 2, 0);
      compute(1);
      __syncthreads();
    }
  }
#endif

#pragma unroll
  for (int kx = 0; kx < NKX; ++kx)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int cc = c0 + wn * 32 + (lane & 31);
      const float sxv = (a.pre_x && cc < a.Cx) ? a.pre_x[(size_t)bsmp * a.Cx + cc] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int aa = a0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (aa < a.A && cc < a.Bc) {
          const float sgv = a.pre_g ? a.pre_g[(size_t)bsmp * a.Cg + aa] : 1.f;
          const size_t off = ((size_t)(ky * NKX + kx) * a.A + aa) * a.Bc + cc;
          if (a.slab) {
            const size_t si = (size_t)split * (NKX * NKX) * a.A * a.Bc + off;
            if (a.slab_bf16) ((__bf16*)a.slab)[si] = (__bf16)(acc[kx][mi][r] * sxv * sgv); else a.slab[si] = acc[kx][mi][r] * sxv * sgv;
          }
          else atomicAdd(a.gwp + off, acc[kx][mi][r] * sxv * sgv);
        }
      }
    }
}

// gwp[i] = sum_s slab[s][i]: the partial weight-gradient tiles of the row-segment kernel meet here instead of through fp32
// atomics (12.5 M atomic adds onto 147 K addresses for the 128x128 top layer: ~20 % of that kernel)
template <typename ST>                                       // ST: element type of the partial tiles (float, or __bf16: WgradArgs::slab_bf16)
__global__ void slab_reduce_kernel(const ST* __restrict__ slab, float* __restrict__ gwp, int total, int nsplit) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float acc = 0.f;
#pragma unroll 4
  for (int sI = 0; sI < nsplit; ++sI) acc += (float)slab[(size_t)sI * total + i];
  gwp[i] = acc;
}

// slab reduction fused with the un-prep of the weight gradient (lcgan_conv_wgrad_unprep): gw[a][b][t] = scale * sum_s slab[s][src] (+ demod
// term), src = the element's place in the prepared layout.  One launch instead of slab_reduce + unprep and no gwp round trip.
template <typename ST>
__global__ void slab_reduce_unprep_kernel(const ST* __restrict__ slab, int nsplit, int A, int Bc, int kk, float scale, int transposed,
                                          const float* __restrict__ w, const float* __restrict__ gwsq, float* __restrict__ gw) {
  const int total = A * Bc * kk;
  const int i = blockIdx.x * 256 + threadIdx.x;              // index in the prepared layout ([t][A][Bc] or [t][Bc][A]): coalesced slab reads
  if (i >= total) return;
  float acc = 0.f;
#pragma unroll 4
  for (int sI = 0; sI < nsplit; ++sI) acc += (float)slab[(size_t)sI * total + i];
  const int t = i / (A * Bc), r = i - t * (A * Bc);
  const int aa = transposed ? r % A : r / Bc, b = transposed ? r / A : r % Bc;
  const size_t ab = (size_t)aa * Bc + b, idx = ab * kk + t;
  float v = acc * scale;
  if (gwsq) v += 2.f * scale * scale * w[idx] * gwsq[ab];
  gw[idx] = v;
}

float* g_slab[MAX_DEV] = {};
size_t g_slab_bytes[MAX_DEV] = {};
float* wgrad_slab_scratch(size_t bytes, hipStream_t s) {          // grow-only, per device, owned by the library; every element is written before it is read
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  scratch_order(dev, s);
  if (bytes > g_slab_bytes[dev]) {
    if (g_slab[dev]) hipFree(g_slab[dev]);
    g_slab_bytes[dev] = std::max(bytes, (size_t)64 << 20);
    if (hipMalloc((void**)&g_slab[dev], g_slab_bytes[dev]) != hipSuccess) { g_slab[dev] = nullptr; g_slab_bytes[dev] = 0; return nullptr; }
  }
  return g_slab[dev];
}

// =========================================================================================================
// weight layout kernels
// =========================================================================================================
// w [A][Bc][kk] fp32 (reference layout, custom_layers.py:32,55)  ->  wp [parts][kk][N][Kpad] bf16 split, scaled.
//   transpose == 0: N = A,  reduction channel = Bc   (forward conv)
//   transpose == 1: N = Bc, reduction channel = A    (data gradient / transposed conv)
__global__ void prep_weight_kernel(const float* __restrict__ w, int A, int Bc, int kk, float scale, int transpose,
                                   __bf16* __restrict__ out, int parts, int N, int Kc, int Kpad) {
  const size_t total = (size_t)kk * N * Kpad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Kpad);
    const int n = (int)((idx / Kpad) % N);
    const int t = (int)(idx / ((size_t)Kpad * N));
    float v = 0.f;
    if (c < Kc) {
      const int aa = transpose ? c : n, bb = transpose ? n : c;
      v = w[((size_t)aa * Bc + bb) * kk + t] * scale;
    }
    for (int pp = 0; pp < parts; ++pp) {           // bf16 split v = p0 + p1 + ... (8 mantissa bits per part)
      const __bf16 h = (__bf16)v;
      out[(size_t)pp * total + idx] = h;
      v -= (float)h;
    }
  }
}

// All weight preparations of a network in ONE launch (after its Adam step every conv weight needs its forward layout, its
// transposed layout and, for modulated convs, the demodulation statistic: ~120 launches per iteration otherwise).  A device
// table describes the jobs; a block handles one chunk of one job: chunk_index >= 0 is a tile of 16 output rows n x 64 input
// channels c with ALL taps (tile = index, row-major over ceil(N/16) x ceil(Kpad/64)), chunk_index < 0 is the 16384-element
// chunk -index-1 of the wsq statistic.  Outputs live in two flat buffers at the offsets recorded in the table.
// The parameter keeps its taps innermost ([A][Bc][k][k]) and the prepared layouts keep them outermost, so a tile is read as
// whole runs of the parameter (64*kk floats per row, or 16*kk for the transposed layout), transposed through LDS and written
// as 128-byte runs of every tap plane; an element-per-thread gather fetched each parameter line once per tap (9 x).
struct PrepDesc { const float* w; long long out_off; long long wsq_off; int A, Bc, kk, transpose, parts, N, Kc, Kpad; float scale; int pad; };
constexpr int PREP_CHUNK = 16384;
constexpr int PREP_TN = 16, PREP_TC = 64, PREP_TT = 9;                 // tile rows, tile channels, taps staged per pass
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
template <int TT>                                                      // TT = taps per pass when known (1, 9); 0 = min(kk, 9) at run time
__device__ __forceinline__ void prep_tile(const PrepDesc& d, int tile, __bf16* __restrict__ out, float* __restrict__ wsq_base, float* lds) {
  const int tiles_c = (d.Kpad + PREP_TC - 1) / PREP_TC;
  const int n0 = (tile / tiles_c) * PREP_TN, c0 = (tile % tiles_c) * PREP_TC;
  const size_t total = (size_t)d.kk * d.N * d.Kpad;
  const int tid = threadIdx.x;
  for (int t0 = 0; t0 < d.kk; t0 += PREP_TT) {
    const int tt_n = TT ? TT : min(d.kk - t0, PREP_TT);
    const int pitch = PREP_TC * tt_n + 1;                              // lds[nl][cl][tt], odd row pitch
    if (t0) __syncthreads();
    if constexpr (TT != 0) {
      // Known tap count: every thread owns E / 256 elements; it issues them in batches of unconditional range-checked buffer
      // loads (an element outside the weight reads 0 through offset -1) so the batch's latencies overlap, then parks them in LDS.
      constexpr int E = PREP_TN * PREP_TC * TT, PER = E / 256, BATCH = PER % 12 == 0 ? 12 : 4;
      static_assert(E % 256 == 0 && PER % BATCH == 0, "tile elements per thread");
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, (int)((size_t)d.A * d.Bc * d.kk * 4), 0x00020000);
      for (int b0 = 0; b0 < PER; b0 += BATCH) {
        float v[BATCH];
        int dst[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int e = tid + (b0 + j) * 256;
          int nl, cl, tt;
          if (!d.transpose) { nl = e / (PREP_TC * TT); const int r = e % (PREP_TC * TT); cl = r / TT; tt = r % TT; }   // w[n][c][t]: 64*TT-float runs
          else              { cl = e / (PREP_TN * TT); const int r = e % (PREP_TN * TT); nl = r / TT; tt = r % TT; }   // w[c][n][t]: 16*TT-float runs
          const int n = n0 + nl, c = c0 + cl;
          const unsigned row = d.transpose ? (unsigned)c * d.Bc + n : (unsigned)n * d.Bc + c;
          const unsigned off = (n < d.N && c < d.Kc) ? (row * d.kk + t0 + tt) * 4u : 0xffffffffu;
          v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
          dst[j] = nl * pitch + cl * TT + tt;
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) lds[dst[j]] = v[j];
      }
    } else if (!d.transpose) {                                         // w[n][c][t]: a row n holds 64 * kk contiguous floats of the tile
      const int run = PREP_TC * tt_n;
      for (int e = tid; e < PREP_TN * run; e += 256) {
        const int nl = e / run, r = e - nl * run, cl = r / tt_n, tt = r - cl * tt_n;
        const int n = n0 + nl, c = c0 + cl;
        lds[nl * pitch + r] = (n < d.N && c < d.Kc) ? d.w[((size_t)n * d.Bc + c) * d.kk + t0 + tt] : 0.f;
      }
    } else {                                                           // w[c][n][t]: a row c holds 16 * kk contiguous floats of the tile
      const int run = PREP_TN * tt_n;
      for (int e = tid; e < PREP_TC * run; e += 256) {
        const int cl = e / run, r = e - cl * run, nl = r / tt_n, tt = r - nl * tt_n;
        const int n = n0 + nl, c = c0 + cl;
        lds[nl * pitch + cl * tt_n + tt] = (n < d.N && c < d.Kc) ? d.w[((size_t)c * d.Bc + n) * d.kk + t0 + tt] : 0.f;
      }
    }
    __syncthreads();
    if (d.pad && d.kk <= PREP_TT) {                                    // the tile holds every tap of its weights: wsq comes for free
      float* wsq = wsq_base + d.wsq_off;
      for (int e = tid; e < PREP_TN * PREP_TC; e += 256) {
        const int nl = e / PREP_TC, cl = e % PREP_TC, n = n0 + nl, c = c0 + cl;
        if (n >= d.N || c >= d.Kc) continue;
        float acc = 0.f;
        for (int tt = 0; tt < tt_n; ++tt) { const float v = lds[nl * pitch + cl * tt_n + tt] * d.scale; acc += v * v; }
        wsq[d.transpose ? (size_t)c * d.Bc + n : (size_t)n * d.Bc + c] = acc;
      }
    }
    for (int e = tid; e < tt_n * PREP_TN * (PREP_TC / 2); e += 256) {  // two channels per thread: 32 lanes write one 128-byte run
      const int tt = e / (PREP_TN * PREP_TC / 2), r = e % (PREP_TN * PREP_TC / 2), nl = r / (PREP_TC / 2), cp = r % (PREP_TC / 2);
      const int n = n0 + nl, c = c0 + 2 * cp;
      if (n >= d.N || c >= d.Kpad) continue;                           // Kpad is a multiple of 32: a pair is in or out as a whole
      float v0 = lds[nl * pitch + (2 * cp) * tt_n + tt] * d.scale, v1 = lds[nl * pitch + (2 * cp + 1) * tt_n + tt] * d.scale;
      const size_t idx = ((size_t)(t0 + tt) * d.N + n) * d.Kpad + c;
      for (int pp = 0; pp < d.parts; ++pp) {
        bf16x2 h;
        h[0] = (__bf16)v0; h[1] = (__bf16)v1;
        *(bf16x2*)(out + (size_t)pp * total + idx) = h;
        v0 -= (float)h[0]; v1 -= (float)h[1];
      }
    }
  }
}
__global__ __launch_bounds__(256) void prep_group_kernel(const PrepDesc* __restrict__ descs, const int* __restrict__ chunk_entry,
                                                         const int* __restrict__ chunk_index, __bf16* __restrict__ out_base,
                                                         float* __restrict__ wsq_base) {
  __shared__ float lds[PREP_TN * (PREP_TC * PREP_TT + 1)];
  const PrepDesc d = descs[chunk_entry[blockIdx.x]];
  const int ci = chunk_index[blockIdx.x];
  if (ci >= 0) {
    __bf16* out = out_base + d.out_off;
    if (d.kk == 9) prep_tile<9>(d, ci, out, wsq_base, lds);
    else if (d.kk == 1) prep_tile<1>(d, ci, out, wsq_base, lds);
    else prep_tile<0>(d, ci, out, wsq_base, lds);
  } else {
    const int AB = d.A * d.Bc;
    const int beg = (-ci - 1) * PREP_CHUNK, end = min(beg + PREP_CHUNK, AB);
    float* wsq = wsq_base + d.wsq_off;
    for (int i = beg + threadIdx.x; i < end; i += 256) {
      float acc = 0.f;
      for (int t = 0; t < d.kk; ++t) { const float v = d.w[(size_t)i * d.kk + t] * d.scale; acc += v * v; }
      wsq[i] = acc;
    }
  }
}

// wsq[a][b] = sum_t (scale * w[a][b][t])^2      (demodulation statistic, custom_layers.py:67)
__global__ void wsq_kernel(const float* __restrict__ w, int AB, int kk, float scale, float* __restrict__ wsq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= AB) return;
  float s = 0.f;
  for (int t = 0; t < kk; ++t) { const float v = w[(size_t)i * kk + t] * scale; s += v * v; }
  wsq[i] = s;
}

// gw[a][b][t] = scale * gwp[t][a][b] + 2 * scale^2 * w[a][b][t] * gwsq[a][b]
// transposed != 0: gwp is [t][Bc][A] (weight gradient of the transposed convolution, whose low-res operand carries the
// parameter's SECOND axis): gw[a][b][t] = scale * gwp[t][b][a] + ...
__global__ void unprep_wgrad_kernel(const float* __restrict__ gwp, int A, int Bc, int kk, float scale, int transposed,
                                    const float* __restrict__ w, const float* __restrict__ gwsq, float* __restrict__ gw) {
  const size_t total = (size_t)A * Bc * kk;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(idx % kk);
    const size_t ab = idx / kk;
    const int b = (int)(ab % Bc), aa = (int)(ab / Bc);
    const size_t src = transposed ? ((size_t)t * Bc + b) * A + aa : (size_t)t * A * Bc + ab;
    float v = gwp[src] * scale;
    if (gwsq) v += 2.f * scale * scale * w[idx] * gwsq[ab];
    gw[idx] = v;
  }
}

// epilogue of a split-K convolution: y = act(post * ws + bias) * gain + residual   over [B*Hout*Wout][Cout]
template <typename T>
__global__ void conv_finalize_kernel(float* __restrict__ ws, T* __restrict__ y, const float* __restrict__ post,
                                     const float* __restrict__ bias, const T* __restrict__ residual,
                                     long long npix, int pix_per_sample, int Cout, int N, float bias_scale, float gain, int act,
                                     int res_half, int Wout) {
  const int nvec = Cout >> 3;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= npix * nvec) return;
  const int v = (int)(gid % nvec);
  const long long pix = gid / nvec;
  const int b = (int)(pix / pix_per_sample);
  const size_t off = (size_t)pix * Cout + v * 8;
  F8 s = Feat<float>::load(ws + off);
  Feat<float>::store(ws + off, f8_zero());                         // leave the scratch zeroed for the next split-K launch
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = v * 8 + j;
    float t = s.v[j];
    if (post) t *= post[(size_t)b * Cout + n];
    if (bias && n < N) t += bias[n] * bias_scale;
    s.v[j] = act_fwd(t, act) * gain;
  }
  if (residual) {
    size_t roff = off;
    float rs = 1.f;
    if (res_half) {                                               // [B][Hout/2][Wout/2][Cout] residual, nearest x2, * 1/4
      const int Hout = pix_per_sample / Wout, p = (int)(pix - (long long)b * pix_per_sample), oy = p / Wout, ox = p - oy * Wout;
      roff = ((size_t)(b * (Hout >> 1) + (oy >> 1)) * (Wout >> 1) + (ox >> 1)) * Cout + v * 8;
      rs = 0.25f;
    }
    const F8 r = Feat<T>::load(residual + roff);
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] += rs * r.v[j];
  }
  Feat<T>::store(y + off, s);
}

template <typename T, int NS>
int launch_igemm(const ConvArgs& a, int nphase, hipStream_t s) {
  constexpr int NT = 2 * NS;
  const size_t smem = 2 * NT * TILE * sizeof(__bf16) + 2 * BM * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)conv_igemm_kernel<T, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, BM), cdiv(a.Cout, BN), nphase * a.nsplit);
  bool dmaq = false;
  if constexpr (NS == 1 && sizeof(T) == 2) {
    dmaq = g_igemm_dma && !a.pre && a.Cin % 32 == 0 && a.Kpad == a.Cin && (long long)a.B * a.Hin * a.Win * a.Cin < (1ll << 31);
    if (dmaq) {
      static bool dset = false;
      if (!dset) { hipFuncSetAttribute((const void*)conv_igemm_kernel<T, NS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); dset = true; }
      if (g_igemm_dma >= 3) {
        static bool d8 = false;
        if (!d8) {
          hipFuncSetAttribute((const void*)conv_igemm8_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 4 * 128 * 64 + (2 * BM + 4) * (int)sizeof(int));
          hipFuncSetAttribute((const void*)conv_igemm8_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4 * 128 * 64 + (2 * BM + 4) * (int)sizeof(int));
          d8 = true;
        }
        if (g_igemm_dma == 3) hipLaunchKernelGGL((conv_igemm8_kernel<3>), grid, dim3(512), (size_t)(3 * 4 * 128 * 64 + (2 * BM + 4) * sizeof(int)), s, a);
        else hipLaunchKernelGGL((conv_igemm8_kernel<4>), grid, dim3(512), (size_t)(4 * 4 * 128 * 64 + (2 * BM + 4) * sizeof(int)), s, a);
      } else
      hipLaunchKernelGGL((conv_igemm_kernel<T, NS, true>), grid, dim3(256), (size_t)(12 * 128 * 64 + 2 * BM * sizeof(int)), s, a);
    }
  }
  if (!dmaq) hipLaunchKernelGGL((conv_igemm_kernel<T, NS>), grid, dim3(256), smem, s, a);
  if (a.nsplit > 1 && !a.slab) launch_finalize<T>(a, a.ws, s);
  return launch_status();
}

// x_scaled[b,p,c] = bf16(x[b,p,c] * pre[b,c]): the per-sample input scale of a LOW-RESOLUTION modulated convolution applied once (a few
// MB, ~5 us) so that the launch can take the LDS-DMA main loop of the generic kernel (which has no register stage to scale in); same
// rounding as the register-staged loop (fp32 product, one bf16 rounding).
__global__ void prescale_kernel(const __bf16* __restrict__ x, const float* __restrict__ pre, __bf16* __restrict__ out,
                                long long nvec, int HW, int C, int pre_stride) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const int cv = C >> 3, v = (int)(i % cv);
  const int b = (int)(i / ((long long)cv * HW));
  const bf16x8 t = *(const bf16x8*)(x + i * 8);
  const float* ps = pre + (size_t)b * pre_stride + v * 8;
  const f32x4 p0 = *(const f32x4*)ps, p1 = *(const f32x4*)(ps + 4);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (__bf16)((float)t[j] * (j < 4 ? p0[j] : p1[j - 4]));
  *(bf16x8*)(out + i * 8) = o;
}
__bf16* g_prescale[MAX_DEV] = {};
size_t g_prescale_bytes[MAX_DEV] = {};
__bf16* prescale_scratch(size_t bytes, hipStream_t s) {               // grow-only, per device; written in full before every use
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  scratch_order(dev, s);
  if (bytes > g_prescale_bytes[dev]) {
    if (g_prescale[dev]) hipFree(g_prescale[dev]);
    g_prescale_bytes[dev] = std::max(bytes, (size_t)16 << 20);
    if (hipMalloc((void**)&g_prescale[dev], g_prescale_bytes[dev]) != hipSuccess) { g_prescale[dev] = nullptr; g_prescale_bytes[dev] = 0; return nullptr; }
  }
  return g_prescale[dev];
}

// Weight gradient of the flow layer's 1x1 GEMM (ops.FlowConvFn: A = 18 = 9 taps x 2 flow channels against Cin input channels):
//   gwp[a][c] += pre_x[b][c] * sum_p g[b][p][a] * x[b][p][c]
// One pass over x, which is all the work there is (9.6 GFLOP against 268 MB at 128 x 128 x 256, batch 32): the row-segment MFMA kernel ran it
// as a 128-wide tile for 18 useful columns, one register-staged workgroup per CU, two chunks of 16 KB in flight -- 229 us, 1.3 TB/s.  Here a
// workgroup takes FW_POS positions of one sample: g goes to LDS once (48 B per position), thread (pl, cv) walks positions pl, pl + npl, ...
// with 8 channels of x per 16-byte load and 18 x 8 accumulators, the position lanes meet through LDS and the sample's scale is applied
// to the sum (as the row-segment kernel applies it to its accumulator).
constexpr int FW_POS = 1024, FW_A = 18, FW_CG = 24;
__global__ __launch_bounds__(256) void flow_wgrad_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ g, float* __restrict__ gwp,
                                                         const float* __restrict__ pre_x, int HW, int Cx, int Bc, int npos) {
  __shared__ __attribute__((aligned(16))) __bf16 gsh[FW_POS * FW_CG];       // 48 KB
  __shared__ float red[256 * 4];                                            // 4 KB: [position lane][channel]
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const int tid = threadIdx.x, b = blockIdx.y, p0 = blockIdx.x * npos, np = min(npos, HW - p0);        // npos <= FW_POS
  // thread (pl, cq): FOUR channels of x per 8-byte load (eight would need 144 accumulators: one wave per SIMD), positions pl, pl + npl, ...
  const int ncq = Cx >> 2, npl = 256 / ncq, cq = tid % ncq, pl = tid / ncq;
  {
    const u32x4* src = (const u32x4*)(g + ((size_t)b * HW + p0) * FW_CG);
    u32x4* dst = (u32x4*)gsh;
    for (int i = tid; i < np * 3; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  f32x2 acc[FW_A][2];
#pragma unroll
  for (int a = 0; a < FW_A; ++a) { acc[a][0] = f32x2{0.f, 0.f}; acc[a][1] = f32x2{0.f, 0.f}; }
  const __bf16* xb = x + ((size_t)b * HW + p0) * Cx + cq * 4;
  auto one = [&](const u32x2 u, int p) {
    const f32x2 x0 = {__builtin_bit_cast(float, u[0] << 16), __builtin_bit_cast(float, u[0] & 0xffff0000u)};
    const f32x2 x1 = {__builtin_bit_cast(float, u[1] << 16), __builtin_bit_cast(float, u[1] & 0xffff0000u)};
    const u32x4* gp = (const u32x4*)(gsh + p * FW_CG);
    const u32x4 g0 = gp[0], g1 = gp[1];
    const unsigned g2 = ((const unsigned*)(gp + 2))[0];
    const unsigned gw[9] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3], g2};
#pragma unroll
    for (int a = 0; a < FW_A; ++a) {
      const float gv = __builtin_bit_cast(float, (a & 1) ? (gw[a >> 1] & 0xffff0000u) : (gw[a >> 1] << 16));
      const f32x2 g2v = {gv, gv};
      acc[a][0] = __builtin_elementwise_fma(x0, g2v, acc[a][0]);
      acc[a][1] = __builtin_elementwise_fma(x1, g2v, acc[a][1]);
    }
  };
  int p = pl;
  constexpr int UNR = 4;                                         // loads of x in flight per thread (eight: more registers, one workgroup less per CU, 111 -> 130 us at 128 x 128 x 256)
  for (; p + (UNR - 1) * npl < np; p += UNR * npl) {
    u32x2 u[UNR];
#pragma unroll
    for (int k = 0; k < UNR; ++k) u[k] = *(const u32x2*)(xb + (size_t)(p + k * npl) * Cx);
#pragma unroll
    for (int k = 0; k < UNR; ++k) one(u[k], p + k * npl);
  }
  for (; p < np; p += npl) one(*(const u32x2*)(xb + (size_t)p * Cx), p);
  // position lanes meet through LDS, one output row (a) at a time
  for (int a = 0; a < FW_A; ++a) {
    __syncthreads();
    float* r = red + pl * Cx + cq * 4;
    r[0] = acc[a][0][0]; r[1] = acc[a][0][1]; r[2] = acc[a][1][0]; r[3] = acc[a][1][1];
    __syncthreads();
    for (int c = tid; c < Cx; c += 256) {
      float sum = 0.f;
      for (int q = 0; q < npl; ++q) sum += red[q * Cx + c];
      if (pre_x) sum *= pre_x[(size_t)b * Cx + c];
      if (c < Bc) atomicAdd(gwp + (size_t)a * Bc + c, sum);
    }
  }
}

// Small-M layers (4x4 ... 16x16 grids) are weight-streaming bound and would occupy a handful of CUs: split the (tap, chunk)
// loop over blockIdx.z so >= ~256 workgroups stream disjoint weight slices; partials meet in an fp32 workspace.
int dispatch_igemm(const ConvArgs& a_in, int nphase, int dtype, hipStream_t s) {
  ConvArgs a = a_in;
  a.nsplit = 1; a.ws = nullptr; a.slab = nullptr; a.cnt = nullptr;
  g_pool_written = false; g_mask_written = false;
  if (dtype == DT_BF16 && g_use_halo && try_launch_halo(a, nphase, a.in_mul, s)) return launch_status();
  if (g_igemm_dma >= 2 && dtype == DT_BF16 && a.pre && a.Cin % 32 == 0 && a.Kpad == a.Cin && a.M >= 256) {   // (with the eight-wave kernel and its slab split-K behind it the pass pays from M = 256: conv family 8.96 -> 8.83 ms at local batch 4, 13.39 -> 13.31 at 8; it was 2048 with the four-wave loop)
    const size_t elems = (size_t)a.B * a.Hin * a.Win * a.Cin;
    if (elems * sizeof(__bf16) <= ((size_t)32 << 20)) {
      __bf16* xs_ = prescale_scratch(elems * sizeof(__bf16), s);
      if (xs_) {
        const long long nvec = (long long)(elems / 8);
        hipLaunchKernelGGL(prescale_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, (const __bf16*)a.x, a.pre, xs_, nvec,
                           a.Hin * a.Win, a.Cin, a.pre_stride);
        a.x = xs_; a.pre = nullptr;
      }
    }
  }
  const int wgs = cdiv(a.M, BM) * cdiv(a.Cout, BN) * nphase;
  int nq_min = 1 << 30;
  for (int p = 0; p < nphase; ++p) nq_min = std::min(nq_min, a.taps[p].n * a.kc_per_tap);
  if (g_use_splitk && wgs <= 192 && nq_min >= 8) {
    int ns = std::min(nq_min / 4, (256 + wgs - 1) / wgs);
    if (ns > 1) {
      // (... and only launches of >= 2048 output positions: below that the atomics are few and the slab's chain of device-scope round trips -- store,
      //  counter, reload -- costs a 10-20 us launch 5-10 us more than the finalize launch it saves; measured per shape in the batch-4 iteration)
      const bool k8 = dtype == DT_BF16 && g_igemm_dma >= 3 && g_splitk_slabs > 0 && ns <= g_splitk_slabs && (long long)a.M * nphase >= 2048 && !a.pre && a.Cin % 32 == 0 && a.Kpad == a.Cin &&
                      (long long)a.B * a.Hin * a.Win * a.Cin < (1ll << 31) && wgs <= SPLITK_MAX_TILES;      // (launch_igemm's conditions for conv_igemm8_kernel)
      if (k8) {
        int* cnt = nullptr;
        float* slab = splitk_slab_scratch((size_t)wgs * ns * BM * BN * sizeof(float), &cnt, s);
        if (slab) { a.nsplit = ns; a.slab = slab; a.cnt = cnt; }
      } else {
        float* ws = splitk_scratch((size_t)a.B * a.Hout * a.Wout * a.Cout * sizeof(float), s);
        if (ws) { a.nsplit = ns; a.ws = ws; }
      }
    }
  }
  // generic path: the fused style-gradient reduction (xs, gs) runs as its own pass over the unscaled output
  const void* xs = a.xs; float* gs = a.gs; const float* sr_scale = a.post; const void* sr_res = a.residual;
  if (xs) { a.post = nullptr; a.xs = nullptr; a.gs = nullptr; a.residual = nullptr; }      // (a residual of the style-gradient form joins AFTER the scale)
  int rc = LCGAN_EINVAL;
  if (dtype == DT_BF16) rc = launch_igemm<__bf16, 1>(a, nphase, s);
  else if (dtype == DT_F32) rc = launch_igemm<float, 3>(a, nphase, s);
  if (xs && rc == LCGAN_OK) rc = lcgan_scale_reduce_res(a.y, xs, sr_scale, gs, sr_res, a.B, a.Hout * a.Wout, a.Cout, dtype, s);
  return rc;
}

}  // namespace

// =========================================================================================================
// C ABI (declared in include/lcgan_hip.h)
// =========================================================================================================
extern "C" {

// option 0: use the halo-tile conv kernel for bf16 (1 = default) ; returns the previous value
int lcgan_set_option(int option, int value) {
  if (option == 0) { const int old = g_use_halo; g_use_halo = value; return old; }
  if (option == 1) { const int old = g_use_splitk; g_use_splitk = value; return old; }
  if (option == 2) { const int old = g_wgrad3_wgs; g_wgrad3_wgs = value; return old; }
  if (option == 3) { const int old = g_dbg_no_atomics; g_dbg_no_atomics = value; return old; }
  if (option == 4) { const int old = g_mfma16; g_mfma16 = value; return old; }
  if (option == 5) { const int old = g_wgrad3_small; g_wgrad3_small = value; return old; }
  if (option == 6) { const int old = g_halo_min_wgs; g_halo_min_wgs = value; return old; }
  if (option == 7) { const int old = g_halo_narrow_min_wgs; g_halo_narrow_min_wgs = value; return old; }
  if (option == 8) { const int old = g_wgrad_slab_min; g_wgrad_slab_min = value; return old; }
  if (option == 9) { const int old = g_wgrad3_pack; g_wgrad3_pack = value; return old; }
  if (option == 10) { const int old = g_halo_dma; g_halo_dma = value; return old; }
  if (option == 11) { const int old = g_halo_dma_mod; g_halo_dma_mod = value; return old; }
  if (option == 12) { const int old = g_wgrad_dma; g_wgrad_dma = value; return old; }
  if (option == 13) { const int old = g_halo_s2dma; g_halo_s2dma = value; return old; }
  if (option == 14) { const int old = g_halo_nb_group_kb; g_halo_nb_group_kb = value; return old; }
  if (option == 15) { const int old = g_wgrad_xcd; g_wgrad_xcd = value; return old; }
  if (option == 16) { const int old = g_igemm_dma; g_igemm_dma = value; return old; }
  if (option == 19) { const int old = g_splitk_slabs; g_splitk_slabs = value; return old; }
  if (option == 20) { const int old = g_wgrad_low_direct; g_wgrad_low_direct = value; return old; }
  if (option == 21) { const int old = g_wgrad_low_parts; g_wgrad_low_parts = value; return old; }
  if (option == 22) { const int old = g_halo_split; g_halo_split = value; return old; }
  if (option == 23) { const int old = g_flow_wgrad; g_flow_wgrad = value; return old; }
  if (option == 24) { const int old = g_halo_phase_x; g_halo_phase_x = value; return old; }
  if (option == 25) { const int old = g_wgrad_slab_bf16; g_wgrad_slab_bf16 = value; return old; }
  if (option == 26) { const int old = g_s2duo; g_s2duo = value; return old; }
  if (option == 17) { const int old = g_wgrad_prescale_mb; g_wgrad_prescale_mb = value; return old; }
  if (option == 18) { const int old = g_halo_wmod_mb; g_halo_wmod_mb = value; return old; }
  return LCGAN_EINVAL;
}

#ifdef HALO_STAMPS
int lcgan_halo_life(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_halo_life), sizeof(unsigned long long) * 16384 * 9) == hipSuccess ? LCGAN_OK : LCGAN_ELAUNCH;
}
int lcgan_halo_clock(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_halo_clock), sizeof(unsigned long long) * 2048 * 8 * 2) == hipSuccess ? LCGAN_OK : LCGAN_ELAUNCH;
}
int lcgan_halo_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_halo_stamps), sizeof(unsigned long long) * 2048 * 8 * 5) == hipSuccess ? LCGAN_OK : LCGAN_ELAUNCH;
}
#endif

int lcgan_conv_weight_prep(const float* w, int A, int Bc, int k, float scale, int transpose,
                           void* wp, int parts, float* wsq, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((k != 1 && k != 3) || parts < 1 || parts > 3) return LCGAN_EINVAL;
  const int kk = k * k, N = transpose ? Bc : A, Kc = transpose ? A : Bc, Kpad = (Kc + 31) / 32 * 32;
  ProfScope p(KID_WEIGHT_PREP, 0, (double)A * Bc * kk * 8, s);
  const size_t total = (size_t)kk * N * Kpad;
  hipLaunchKernelGGL(prep_weight_kernel, dim3((unsigned)min((size_t)4096, (total + 255) / 256)), dim3(256), 0, s,
                     w, A, Bc, kk, scale, transpose, (__bf16*)wp, parts, N, Kc, Kpad);
  if (wsq) hipLaunchKernelGGL(wsq_kernel, dim3(cdiv((long long)A * Bc, 256)), dim3(256), 0, s, w, A * Bc, kk, scale, wsq);
  return launch_status();
}

// descs: device array of 64-byte job descriptors {w (8 B), out_off (8 B, bf16 elements), wsq_off (8 B, floats, unused when the
// job has no wsq chunks), A, Bc, kk, transpose, parts, N, Kc, Kpad (8 ints), scale (float), pad (!= 0 with kk <= 9: the
// job's tiles also write wsq)}; chunk_entry / chunk_index:
// device int arrays, one entry per chunk (chunk_index >= 0: tile of 16 rows x 64 channels x all taps of the prepared weight,
// row-major over ceil(N/16) x ceil(Kpad/64); < 0: 16384-element wsq chunk -index-1).
int lcgan_conv_weight_prep_group(const void* descs, const int* chunk_entry, const int* chunk_index, int n_chunks,
                                 void* out_base, float* wsq_base, double total_elems, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_chunks <= 0) return LCGAN_OK;
  ProfScope p(KID_WEIGHT_PREP, 0, total_elems * 6.0, s);
  hipLaunchKernelGGL(prep_group_kernel, dim3(n_chunks), dim3(256), 0, s, (const PrepDesc*)descs, chunk_entry, chunk_index,
                     (__bf16*)out_base, wsq_base);
  return launch_status();
}

int lcgan_conv_wgrad_unprep(const float* gwp, int A, int Bc, int k, float scale, int transposed, const float* w,
                            const float* gwsq, float* gw, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int kk = k * k;
  ProfScope p(KID_WEIGHT_PREP, 0, (double)A * Bc * kk * 8, s);
  const size_t total = (size_t)A * Bc * kk;
  hipLaunchKernelGGL(unprep_wgrad_kernel, dim3((unsigned)min((size_t)4096, (total + 255) / 256)), dim3(256), 0, s,
                     gwp, A, Bc, kk, scale, transposed, w, gwsq, gw);
  return launch_status();
}

// (parity mode, dtype f32, expects weights prepared with parts = 3; bf16 with parts = 1)
// Forward convolution  y[b,ho,wo,n] = act(post[b,n] * sum_{t,c} pre[b,c] x[b, ho*stride+ky-pad, wo*stride+kx-pad, c] wp[t][n][c]
//                                        + bias[n]*bias_scale) * gain + residual
// residual: [B,Hout,Wout,Cout], or with residual_half = 1 [B,Hout/2,Wout/2,Cout] added as 0.25 * residual[ho/2][wo/2]
// (the adjoint of avg_pool2d(2), custom_layers.py:202: the gradient of a block's pooled skip branch)
// x: [B,Hin,Win,Cin]  wp: lcgan_conv_weight_prep(transpose=0)  y: [B,Hout,Wout,Cout], Hout = ceil(Hin/stride)
int lcgan_conv_fwd(const void* x, const void* wp, void* y,
                   int B, int Hin, int Win, int Cin, int Cout, int N, int k, int stride,
                   const float* pre, const float* post, const float* bias, float bias_scale,
                   int act, float gain, const void* residual, int residual_half, const void* xs, float* gs, void* pool_out,
                   int dtype, void* stream) {
  return lcgan_conv_fwd_m(x, wp, y, B, Hin, Win, Cin, Cout, N, k, stride, pre, post, bias, bias_scale, act, gain, residual, residual_half, xs, gs,
                          pool_out, nullptr, nullptr, dtype, stream);
}

// lcgan_conv_fwd + an optional second by-product: mask_out [B*Hout*Wout][Cout/32] 32-bit words, bit c % 32 of word c / 32 = (pre-activation of
// channel c > 0) -- all the activation backward needs of y (lcgan_act_bwd_reduce_m / lcgan_box3_actbwd_reduce_m read 1/16 of the bytes).
// *mask_written = 1 when the launch path wrote it (bf16 halo kernels, leaky ReLU, no residual, Cout % 32 == 0), else 0: use y.
int lcgan_conv_fwd_m(const void* x, const void* wp, void* y,
                     int B, int Hin, int Win, int Cin, int Cout, int N, int k, int stride,
                     const float* pre, const float* post, const float* bias, float bias_scale,
                     int act, float gain, const void* residual, int residual_half, const void* xs, float* gs, void* pool_out,
                     void* mask_out, int* mask_written, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mask_written) *mask_written = 0;
  if ((k != 1 && k != 3) || (stride != 1 && stride != 2) || (Cin & 7) || (Cout & 7) || N > Cout) return LCGAN_EINVAL;
  if (xs && (!gs || !post || (residual && residual_half) || bias || act != ACT_NONE || gain != 1.f)) return LCGAN_EINVAL;
  ConvArgs a = {};
  a.xs = xs; a.gs = xs ? gs : nullptr;
  a.pool_out = pool_out;
  a.mask_out = (unsigned*)mask_out;
  a.x = x; a.w = (const __bf16*)wp; a.y = y;
  a.pre = pre; a.post = post; a.bias = bias; a.residual = residual; a.res_half = residual ? residual_half : 0;
  a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = Cin;
  a.Hout = (Hin + stride - 1) / stride; a.Wout = (Win + stride - 1) / stride; a.Cout = Cout;
  a.Hm = a.Hout; a.Wm = a.Wout;
  if (a.res_half && ((a.Hout | a.Wout) & 1)) return LCGAN_EINVAL;
  a.lw = log2_exact(a.Wout); a.lh = log2_exact(a.Hout);
  const long long M = (long long)B * a.Hm * a.Wm;
  if (M <= 0 || M >= (1ll << 31) || (long long)B * Hin * Win >= (1ll << 31)) return LCGAN_EINVAL;
  a.M = (int)M; a.N = N; a.Kpad = (Cin + 31) / 32 * 32; a.kc_per_tap = a.Kpad / BK;
  a.l_hwm = log2_exact((long long)a.Hm * a.Wm); a.l_wm = log2_exact(a.Wm); a.l_kc = log2_exact(a.kc_per_tap);
  a.w_part = (size_t)k * k * N * a.Kpad;
  a.in_mul = stride; a.out_mul = 1; a.pre_stride = Cin; a.post_stride = Cout;
  a.bias_scale = bias_scale; a.gain = gain; a.act = act;
  const int pad = k / 2;
  TapTable& t = a.taps[0];
  t.n = k * k;
  for (int ky = 0; ky < k; ++ky)
    for (int kx = 0; kx < k; ++kx) { const int i = ky * k + kx; t.dy[i] = ky - pad; t.dx[i] = kx - pad; t.wt[i] = i; }
  if (pool_out && ((a.Hout | a.Wout) & 1)) return LCGAN_EINVAL;
  char tag[96] = "";
  if (lcgan_prof_active()) snprintf(tag, sizeof(tag), "fwd B%d %dx%d C%d->%d k%d s%d%s%s%s", B, Hin, Win, Cin, N, k, stride, pre ? " mod" : "", xs ? " +gs" : residual ? " +res" : "", pool_out ? " +pool" : "");
  int rc;
  {
    ProfScope p(KID_CONV_IGEMM, 2.0 * M * N * Cin * k * k, 0, s, tag);
    rc = dispatch_igemm(a, 1, dtype, s);
  }
  if (mask_written) *mask_written = (rc == LCGAN_OK && mask_out && g_mask_written) ? 1 : 0;
  // the by-product avg_pool2d(y, 2): written by the epilogue where the launch path has one for it, by the pooling kernel otherwise
  if (rc == LCGAN_OK && pool_out && !g_pool_written) rc = lcgan_avgpool2(y, pool_out, B, a.Hout, a.Wout, Cout, dtype, stream);
  return rc;
}

// Data gradient of the convolution above == transposed convolution:
//   gx[b,h,w,n] = act(post[b,n] * sum_{t,c} pre[b,c] g[b,(h+pad-ky)/stride,(w+pad-kx)/stride,c] wpT[t][n][c] + bias) * gain + residual
// g: [B,Hg,Wg,Cg] (the conv's output grid)  wpT: lcgan_conv_weight_prep(transpose=1)  gx: [B,Hg*stride,Wg*stride,Cout]
// stride 2 runs as 4 sub-pixel phases with 1/2/2/4 taps (== F.conv_transpose2d(stride=2, padding=1, output_padding=1)).
int lcgan_conv_bwd_data(const void* g, const void* wpT, void* gx,
                        int B, int Hg, int Wg, int Cg, int Cout, int N, int k, int stride,
                        const float* pre, const float* post, const float* bias, float bias_scale,
                        int act, float gain, const void* residual, int residual_half, const void* xs, float* gs, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((k != 1 && k != 3) || (stride != 1 && stride != 2) || (Cg & 7) || (Cout & 7) || N > Cout) return LCGAN_EINVAL;
  if (stride == 2 && k != 3) return LCGAN_EINVAL;
  if (xs && (!gs || !post || (residual && residual_half) || bias || act != ACT_NONE || gain != 1.f)) return LCGAN_EINVAL;
  ConvArgs a = {};
  a.xs = xs; a.gs = xs ? gs : nullptr;
  a.x = g; a.w = (const __bf16*)wpT; a.y = gx;
  a.pre = pre; a.post = post; a.bias = bias; a.residual = residual; a.res_half = residual ? residual_half : 0;
  a.B = B; a.Hin = Hg; a.Win = Wg; a.Cin = Cg;
  a.Hout = Hg * stride; a.Wout = Wg * stride; a.Cout = Cout;
  a.Hm = Hg; a.Wm = Wg;
  if (a.res_half && ((a.Hout | a.Wout) & 1)) return LCGAN_EINVAL;
  a.lw = log2_exact(a.Wout); a.lh = log2_exact(a.Hout);
  const long long M = (long long)B * a.Hm * a.Wm;
  if (M <= 0 || (long long)B * a.Hout * a.Wout >= (1ll << 31)) return LCGAN_EINVAL;
  a.M = (int)M; a.N = N; a.Kpad = (Cg + 31) / 32 * 32; a.kc_per_tap = a.Kpad / BK;
  a.l_hwm = log2_exact((long long)a.Hm * a.Wm); a.l_wm = log2_exact(a.Wm); a.l_kc = log2_exact(a.kc_per_tap);
  a.w_part = (size_t)k * k * N * a.Kpad;
  a.in_mul = 1; a.out_mul = stride; a.pre_stride = Cg; a.post_stride = Cout;
  a.bias_scale = bias_scale; a.gain = gain; a.act = act;
  const int pad = k / 2;
  int nphase = 1;
  double taps_total = k * k;
  if (stride == 1) {
    TapTable& t = a.taps[0];
    t.n = k * k;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) { const int i = ky * k + kx; t.dy[i] = pad - ky; t.dx[i] = pad - kx; t.wt[i] = i; }
  } else {
    // output row 2i+ph receives tap ky iff (2i+ph+1-ky) is even; source row (2i+ph+1-ky)/2 = i + (ph+1-ky)/2
    nphase = 4;
    for (int ph = 0; ph < 2; ++ph)
      for (int pw = 0; pw < 2; ++pw) {
        TapTable& t = a.taps[ph * 2 + pw];
        t.n = 0;
        for (int ky = 0; ky < 3; ++ky) {
          if ((ph + 1 - ky) & 1) continue;
          for (int kx = 0; kx < 3; ++kx) {
            if ((pw + 1 - kx) & 1) continue;
            t.dy[t.n] = (ph + 1 - ky) / 2; t.dx[t.n] = (pw + 1 - kx) / 2; t.wt[t.n] = ky * 3 + kx; ++t.n;
          }
        }
      }
    taps_total = 9.0 / 4.0;   // average taps per output pixel
  }
  char tag[96] = "";
  if (lcgan_prof_active()) snprintf(tag, sizeof(tag), "dgrad B%d %dx%d C%d->%d k%d s%d%s%s", B, Hg, Wg, Cg, N, k, stride, pre ? " mod" : "", xs ? " +gs" : residual ? (residual_half ? " +res/2" : " +res") : "");
  ProfScope p(KID_CONV_IGEMM, 2.0 * (double)B * a.Hout * a.Wout * N * Cg * taps_total, 0, s, tag);
  return dispatch_igemm(a, nphase, dtype, s);
}

// Weight gradient: gwp[t][a][c] += sum_{b,i,j} (pre_g[b,a] g[b,i,j,a]) * (pre_x[b,c] x[b, i*stride+ky-pad, j*stride+kx-pad, c])
// x: [B,Hx,Wx,Cx] (the conv's input side), g: [B,Hg,Wg,Cg] (the conv's output side); gwp fp32 [k*k][A][Bc], must be zeroed.
struct UnprepArgs { float scale; int transposed; const float* w; const float* gwsq; float* gw; };   // optional fused un-prep (gw != NULL)
static int conv_wgrad_impl(const void* x, const void* g, float* gwp,
                     int B, int Hx, int Wx, int Cx, int Hg, int Wg, int Cg, int A, int Bc, int k, int stride,
                     const float* pre_x, const float* pre_g, int dtype, void* stream, const UnprepArgs* up) {
  hipStream_t s = (hipStream_t)stream;
  if ((k != 1 && k != 3) || (stride != 1 && stride != 2) || (Cx & 7) || (Cg & 7) || A > Cg || Bc > Cx) return LCGAN_EINVAL;
  WgradArgs a = {};
  a.x = x; a.g = g; a.gwp = gwp; a.pre_x = pre_x; a.pre_g = pre_g;
  a.B = B; a.Hx = Hx; a.Wx = Wx; a.Cx = Cx; a.Hm = Hg; a.Wm = Wg; a.Cg = Cg; a.A = A; a.Bc = Bc;
  const long long M = (long long)B * Hg * Wg;
  if (M <= 0 || M >= (1ll << 31)) return LCGAN_EINVAL;
  a.M = (int)M; a.stride = stride; a.k = k; a.pad = k / 2;
  a.nchunks = cdiv(M, 32);
  const int tiles = cdiv(A, 128) * cdiv(Bc, 128) * k * k;
  int nsplit = (2048 + tiles - 1) / tiles;                       // aim at ~2048 workgroups
  nsplit = max(1, min(nsplit, a.nchunks / 8 > 0 ? a.nchunks / 8 : 1));
  a.chunks_per_split = cdiv(a.nchunks, nsplit);
  a.nsplit = cdiv(a.nchunks, a.chunks_per_split);
  dim3 grid(cdiv(A, 128), cdiv(Bc, 128), k * k * a.nsplit);
  char tag[96] = "";
  if (lcgan_prof_active()) snprintf(tag, sizeof(tag), "wgrad B%d %dx%d A%d Bc%d k%d s%d%s", B, Hg, Wg, A, Bc, k, stride, (pre_x || pre_g) ? " mod" : "");
  ProfScope p(KID_CONV_WGRAD, 2.0 * M * A * Bc * k * k, 0, s, tag);
  if (dtype == DT_BF16 && g_flow_wgrad && k == 1 && stride == 1 && A == FW_A && Cg == FW_CG && !pre_g && Hx == Hg && Wx == Wg &&
      Cx >= 32 && Cx <= 1024 && (Cx & (Cx - 1)) == 0 &&           // (256 threads = (Cx / 4) channel quads x position lanes)
      (long long)B * Hg * Wg >= 32768) {                          // (smaller launches are latency either way: measured equal or behind the row-segment plan at local batch 4)
    if (up) hipMemsetAsync(gwp, 0, (size_t)A * Bc * sizeof(float), s);
    // positions per workgroup: ~512 workgroups (two per CU) where the grid allows, at least 64 positions each (18 x Cx atomics per workgroup)
    const int npos = std::min(FW_POS, std::max(64, cdiv((long long)B * Hg * Wg, 512)));
    hipLaunchKernelGGL(flow_wgrad_kernel, dim3(cdiv(Hg * Wg, npos), B), dim3(256), 0, s, (const __bf16*)x, (const __bf16*)g, gwp, pre_x, Hg * Wg, Cx, Bc, npos);
    return launch_status();
  }
  // Per-sample operand scales (modulated convolutions: style on the input side, demodulation on the output side) pin every split of
  // the row-segment kernel to ONE sample -- the scales are applied to the fp32 accumulator, once per sample -- which costs the
  // low-resolution layers 115-170 us per launch at batch 32 (B x more, B x smaller workgroups; slabs of B x parts partial tiles).
  // Where both operands are small the scales are applied once by an elementwise pass instead (fp32 product, one bf16 rounding, as
  // the generic kernel's staging does) and the whole batch is one reduction range.
  if (dtype == DT_BF16 && (pre_x || pre_g) && g_wgrad_prescale_mb > 0) {
    const size_t bx = (size_t)B * Hx * Wx * Cx * sizeof(__bf16), bg = (size_t)B * Hg * Wg * Cg * sizeof(__bf16);
    const size_t ox = (bx + 255) & ~(size_t)255;
    // (the threshold scales with the batch: what pinning splits to samples costs grows with B, what the pass costs with the bytes.
    // Measured at batch 4: 256 x 256 x 128 layers 157 us prescaled against 116 per sample, 128 x 128 x 256 132 against 114)
    if (bx + bg <= ((size_t)g_wgrad_prescale_mb << 20) * (size_t)min(B, 32) / 32) {
      __bf16* buf = prescale_scratch(ox + bg, s);
      if (buf) {
        if (pre_x) {
          const long long nvec = (long long)(bx / 16);
          hipLaunchKernelGGL(prescale_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, (const __bf16*)x, pre_x, buf, nvec, Hx * Wx, Cx, Cx);
          a.x = buf; a.pre_x = nullptr; pre_x = nullptr;
        }
        if (pre_g) {
          __bf16* gb = (__bf16*)((char*)buf + ox);
          const long long nvec = (long long)(bg / 16);
          hipLaunchKernelGGL(prescale_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, (const __bf16*)g, pre_g, gb, nvec, Hg * Wg, Cg, Cg);
          a.g = gb; a.pre_g = nullptr; pre_g = nullptr;
        }
      }
    }
  }
  int segw = (Wg & 63) == 0 ? 64 : (Wg & 31) == 0 ? 32 : (Wg == 16 ? 16 : (Wg == 8 ? 8 : 0));
  // LDS-DMA kernel: 64-position chunks at stride 1; 32-position chunks at stride 2 (two workgroups per CU either way)
  const bool dma_geom = dtype == DT_BF16 && k == 3 && Cg % 8 == 0 && Cx % 8 == 0 && !(g_wgrad3_pack && Cg <= 64 && Cx <= 64);
  const bool dma_s1 = dma_geom && g_wgrad_dma && stride == 1 && segw == 64;
  const bool dma_s1h = dma_geom && g_wgrad_dma == 3 && stride == 1 && segw == 32;      // 32-wide grids, stride 1
  const bool dma_s2 = dma_geom && g_wgrad_dma == 3 && stride == 2 && segw >= 32;
  if (dma_s2) segw = 32;
  if (dtype == DT_BF16 && g_use_halo && segw != 0 && Hg * Wg >= 64 && (Hg & 3) == 0 && (g_wgrad3_small == 1 || (k == 3 && segw >= 32) || (g_wgrad3_small == 0 && !(pre_x || pre_g))) &&
      (long long)B * Hx * Wx * Cx < (1ll << 31) && (long long)B * Hg * Wg * Cg < (1ll << 31)) {
    // row-segment kernel: chunk = (sample, row group, SEGW-column segment) of `seg` positions; grid.z = split x kernel row
    const int seg = segw == 64 ? 64 : 32, rows = seg / segw, nkx = k;
    const int tiles3 = cdiv(A, 128) * cdiv(Bc, 128) * nkx;
    const bool scaled = pre_x || pre_g;
    const int groups = scaled ? B : 1;
    // packed groups for narrow layers (see the kernel): PK image rows share one chunk when both sides have <= 128 / PK channels
    const int pk = (g_wgrad3_pack && segw == 64 && k == 3) ? ((Cg <= 32 && Cx <= 32 && (Hg & 3) == 0) ? 4 : (Cg <= 64 && Cx <= 64 && (Hg & 1) == 0) ? 2 : 1) : 1;
    const int cps = (Hg / (rows * pk)) * (Wg / segw) * (scaled ? 1 : B);   // chunks per group
    a.cps_group = cps;
    a.nchunks = groups * cps;
    const int xw = segw * stride + (k == 3 ? 2 : 0);
    const size_t smem3 = 2 * (size_t)(seg + rows * xw) * WG_ROW * sizeof(__bf16);
    // parts per sample.  A workgroup costs a fixed ~32 us (prologue + the 49K-element atomic epilogue) plus ~31 us per 1024
    // positions, and workgroups run in rounds of (256 CUs x occupancy): pick the split that minimises rounds x per-workgroup
    // cost, which lands the grid on whole rounds (1536 workgroups of one-per-CU occupancy were 6 rounds; 768 are 3 longer ones).
    int parts = 1;
    if (g_wgrad3_wgs > 0) {                                       // explicit target (tuning experiments)
      parts = (g_wgrad3_wgs + tiles3 * groups - 1) / (tiles3 * groups);
      const int min_chunks = 2048 / seg;
      parts = max(1, min(parts, cps / min_chunks > 0 ? cps / min_chunks : 1));
    } else {
      // register-staged kernel: 166-236 VGPRs, one 512-thread workgroup per CU whatever the LDS size; the LDS-DMA kernel at
      // stride 1 keeps two (128 VGPRs, 66 KB of LDS each)
      const int occ = (g_wgrad_dma >= 2 && pk == 1 && (dma_s1 || dma_s2 || dma_s1h)) ? 2 : 1;
      const int max_parts = max(1, cps / (1024 / seg));
      double best = 1e30;
      for (int pt = 1; pt <= max_parts; ++pt) {
        const int cpsplit = cdiv(cps, pt), real_parts = cdiv(cps, cpsplit);
        // (XCD order: XCD i runs splits i, i + 8, ...: the fullest XCD, 32 CUs, sets the number of rounds)
        const long long rounds = ((long long)tiles3 * groups * real_parts + 256 * occ - 1) / (256 * occ);   // (XCD order deals the splits beyond whole groups of 8 tile by tile: every XCD gets the same share)
        const double cost = (double)rounds * (1024.0 + (double)cpsplit * seg);   // (a packed chunk costs what a plain one does)
        if (cost < best) { best = cost; parts = real_parts; }
      }
    }
    // Small grids (8 x 8 and 16 x 16; the register-staged kernels): a launch is a handful of chunks per workgroup and tiles3 = 16-48
    // workgroups per split, so what it costs is everything AROUND the chunk loop: clearing gwp, 2.4 M atomics (or a slab of nsplit
    // copies of the 9 MB gradient and its reduction pass) and the un-prep launch.  Unscaled launches therefore take ONE split where
    // the chunk loop is short (the epilogue then writes the finished gradient in weight layout: one launch, 9 MB of traffic), and
    // otherwise as many as the cost model below likes, no longer held to >= 1024 positions per split.
    const bool lowres = (segw <= 16 || k == 1) && pk == 1 && !scaled && g_wgrad_low_direct > 0;   // (1x1 kernels on any grid: tiles3 is 2-16 workgroups per split, the old plan's >= 1024 positions per split left most CUs idle at small batch)
    if (lowres && g_wgrad3_wgs <= 0) {
      if (g_wgrad_low_parts > 0) parts = min(g_wgrad_low_parts, cps);
      else {
        // measured (scripts/micro_wgrad_low.py, 512 x 512 layers, batch 4 and 32): a launch takes ~c0 + cc x chunks per workgroup
        // (cc = 0.69 us for 3x3, 0.5 us for 1x1) while the grid fits one round of 256 workgroups, and S > 1 splits add
        // ~2 + MB x (1.2 + 0.45 S) us for the slab of S gradient copies (MB each) and its reduction launch
        const double mb = (double)k * k * A * Bc * 4e-6, cc = (k == 3 ? 0.69 : 0.5) * (seg == 64 ? 1.6 : 1.0);
        double best = 1e30;
        for (int pt = 1; pt <= max(1, cps / 2); ++pt) {
          const int cpsplit = cdiv(cps, pt), S = cdiv(cps, cpsplit);
          const int slots = k == 1 ? 512 : 256;                  // (the 1x1 instantiations fit two workgroups per CU)
          const int rounds = (tiles3 * S + slots - 1) / slots;
          const double cost = cc * cpsplit * rounds + (S > 1 ? 2.0 + mb * (1.2 + 0.45 * S) : 0.0);
          if (cost < best) { best = cost; parts = S; }
        }
      }
    }
    a.chunks_per_split = cdiv(cps, parts);
    a.parts = cdiv(cps, a.chunks_per_split);
    a.nsplit = groups * a.parts;
    const bool direct = a.nsplit == 1 && !scaled && up && up->gw && lowres;
    if (direct) { a.dgw = up->gw; a.dw = up->w; a.dgwsq = up->gwsq; a.dscale = up->scale; a.dtransposed = up->transposed; }
    // partial tiles meet in a slab + one reduction pass instead of atomics when there are enough splits to make atomics hurt
    a.slab = nullptr;
    const size_t slab_bytes = (size_t)a.nsplit * k * k * A * Bc * sizeof(float);
    // (measured per layer shape with scripts/ab_conv.py: -17 % at 16x16, -6 % at 32x32, -3..-7 % on the stride-2 layers, neutral at
    // 64x64 and 256x256, +3 % at 128x128 stride 1: the atomics of the big stride-1 layers hide under other workgroups' compute)
    // With the LDS-DMA kernel (shorter chunk loops) the big stride-1 layers are neutral at local batch 32 and -5 % of the whole
    // iteration at local batch 4 (170 splits x 49 K atomics per layer were 77 us at the chip's 1.3 TB/s atomic rate), so every
    // launch with enough splits takes the slab.
    if (pk == 1 && g_wgrad_slab_min > 0 && (a.nsplit >= g_wgrad_slab_min || (lowres && a.nsplit > 1)) && slab_bytes <= ((size_t)1 << 30))
      a.slab = wgrad_slab_scratch(slab_bytes, s);                    // (packed groups: several waves add into one element -> atomics only)
    // Partial tiles of a bf16 launch cross memory as bf16: a split accumulates its ~10^4 positions in fp32, rounds the tile once (2^-9
    // relative), and the reduction pass sums the splits in fp32 again -- the rounding errors of S splits add up to ~2^-9 of the total,
    // against the 3-5 % the bf16 operands already cost a weight gradient; the slabs were 12 GB of the 38 GB the family moved.
    a.slab_bf16 = (a.slab && g_wgrad_slab_bf16 && a.nsplit >= 8) ? 1 : 0;
    if (up && !a.slab && !direct) hipMemsetAsync(gwp, 0, (size_t)k * k * A * Bc * sizeof(float), s);   // fused entry: gwp arrives uncleared
    // (XCD order puts split i on XCD i % 8: with fewer than 8 splits it would leave whole XCDs idle)
    a.na = cdiv(A, 128); a.nc = cdiv(Bc, 128); a.xcd_order = g_wgrad_xcd && (a.nsplit >= 8 || g_wgrad_low_direct <= 0);
    dim3 grid3(a.na, a.nc, nkx * a.nsplit);
    if (a.xcd_order) {
      const int tiles = a.na * a.nc * nkx, nfull8 = a.nsplit >> 3, rem = a.nsplit - 8 * nfull8;
      grid3 = dim3(8 * (nfull8 * tiles + cdiv(rem * tiles, 8)), 1, 1);
    }
#define LAUNCH_WG3(ST, SG, SW, NK)                                                                                      \
    {                                                                                                                   \
      static bool set = false;                                                                                          \
      if (!set) { hipFuncSetAttribute((const void*)conv_wgrad3_kernel<ST, SG, SW, NK, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
      hipLaunchKernelGGL((conv_wgrad3_kernel<ST, SG, SW, NK, 1>), grid3, dim3(512), smem3, s, a);                       \
    }
#define LAUNCH_WG3_PK(ST, PKK)                                                                                          \
    {                                                                                                                   \
      static bool set = false;                                                                                          \
      if (!set) { hipFuncSetAttribute((const void*)conv_wgrad3_kernel<ST, 64, 64, 3, PKK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
      hipLaunchKernelGGL((conv_wgrad3_kernel<ST, 64, 64, 3, PKK>), grid3, dim3(512), smem3, s, a);                      \
    }
#define LAUNCH_WG3_K(ST, SG, SW) { if (k == 3) LAUNCH_WG3(ST, SG, SW, 3) else LAUNCH_WG3(ST, SG, SW, 1) }
    if (pk == 4) { if (stride == 1) LAUNCH_WG3_PK(1, 4) else LAUNCH_WG3_PK(2, 4) }
    else if (pk == 2) { if (stride == 1) LAUNCH_WG3_PK(1, 2) else LAUNCH_WG3_PK(2, 2) }
    else if (dma_s1h) {
      const size_t dsm = 2 * (size_t)(8 + 9) * 1024;              // 32 positions + 34 halo pixels per stage
      static bool set = false;
      if (!set) { hipFuncSetAttribute((const void*)conv_wgrad3_dma_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
      hipLaunchKernelGGL((conv_wgrad3_dma_kernel<1, true>), grid3, dim3(512), dsm, s, a);
    }
    else if (dma_s1 || dma_s2) {
      if (dma_s1) {
        const size_t dsm = 2 * (size_t)(16 + 17) * 1024;
        static bool set = false;
        if (!set) { hipFuncSetAttribute((const void*)conv_wgrad3_dma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
        hipLaunchKernelGGL((conv_wgrad3_dma_kernel<1>), grid3, dim3(512), dsm, s, a);
      } else {
        const size_t dsm = 2 * (size_t)(8 + 17) * 1024;
        static bool set = false;
        if (!set) { hipFuncSetAttribute((const void*)conv_wgrad3_dma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
        hipLaunchKernelGGL((conv_wgrad3_dma_kernel<2>), grid3, dim3(512), dsm, s, a);
      }
    }
    else if (stride == 1) {
      if (segw == 64) LAUNCH_WG3_K(1, 64, 64) else if (segw == 32) LAUNCH_WG3_K(1, 32, 32)
      else if (segw == 16) LAUNCH_WG3_K(1, 32, 16) else LAUNCH_WG3_K(1, 32, 8)
    } else {
      if (segw == 64) LAUNCH_WG3_K(2, 64, 64) else if (segw == 32) LAUNCH_WG3_K(2, 32, 32)
      else if (segw == 16) LAUNCH_WG3_K(2, 32, 16) else LAUNCH_WG3_K(2, 32, 8)
    }
#undef LAUNCH_WG3_K
#undef LAUNCH_WG3
#undef LAUNCH_WG3_PK
    if (direct) return launch_status() ? launch_status() : 1;    // 1 = un-prep done
    if (a.slab) {
      const int total = k * k * A * Bc;
      if (up && up->gw) {
        // (the un-prep's A / Bc are the WEIGHT's [A][Bc] = this call's [A][Bc] or its transpose; prepared layout [t][A][Bc] of this call)
        if (a.slab_bf16)
          hipLaunchKernelGGL(slab_reduce_unprep_kernel<__bf16>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const __bf16*)a.slab, a.nsplit,
                             up->transposed ? Bc : A, up->transposed ? A : Bc, k * k, up->scale, up->transposed, up->w, up->gwsq, up->gw);
        else
          hipLaunchKernelGGL(slab_reduce_unprep_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float*)a.slab, a.nsplit,
                             up->transposed ? Bc : A, up->transposed ? A : Bc, k * k, up->scale, up->transposed, up->w, up->gwsq, up->gw);
        return launch_status() ? launch_status() : 1;          // 1 = un-prep done
      }
      if (a.slab_bf16) hipLaunchKernelGGL(slab_reduce_kernel<__bf16>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const __bf16*)a.slab, gwp, total, a.nsplit);
      else hipLaunchKernelGGL(slab_reduce_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float*)a.slab, gwp, total, a.nsplit);
    }
  } else if (dtype == DT_BF16) {
    const size_t smem = 2 * 2 * WG_TILE * sizeof(__bf16);
    // 4 x 4 grids (and whatever else lands here with a short reduction): one split whose epilogue writes the finished gradient -- no
    // clear, no atomics, no un-prep launch (same reasoning as the small-grid plan of the row-segment kernel)
    if (up && up->gw && g_wgrad_low_direct > 0 && a.nchunks <= 64) {
      a.nsplit = 1; a.chunks_per_split = a.nchunks;
      a.dgw = up->gw; a.dw = up->w; a.dgwsq = up->gwsq; a.dscale = up->scale; a.dtransposed = up->transposed;
      hipLaunchKernelGGL((conv_wgrad_kernel<__bf16, 1>), dim3(cdiv(A, 128), cdiv(Bc, 128), k * k), dim3(256), smem, s, a);
      return launch_status() ? launch_status() : 1;              // 1 = un-prep done
    }
    if (up) hipMemsetAsync(gwp, 0, (size_t)k * k * A * Bc * sizeof(float), s);
    hipLaunchKernelGGL((conv_wgrad_kernel<__bf16, 1>), grid, dim3(256), smem, s, a);
  } else if (dtype == DT_F32) {
    const size_t smem = 2 * 6 * WG_TILE * sizeof(__bf16);
    if (up) hipMemsetAsync(gwp, 0, (size_t)k * k * A * Bc * sizeof(float), s);
    static bool attr_set = false;
    if (!attr_set) {
      hipFuncSetAttribute((const void*)conv_wgrad_kernel<float, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_set = true;
    }
    hipLaunchKernelGGL((conv_wgrad_kernel<float, 3>), grid, dim3(256), smem, s, a);
  } else {
    return LCGAN_EINVAL;
  }
  return launch_status();
}

int lcgan_conv_wgrad(const void* x, const void* g, float* gwp,
                     int B, int Hx, int Wx, int Cx, int Hg, int Wg, int Cg, int A, int Bc, int k, int stride,
                     const float* pre_x, const float* pre_g, int dtype, void* stream) {
  return conv_wgrad_impl(x, g, gwp, B, Hx, Wx, Cx, Hg, Wg, Cg, A, Bc, k, stride, pre_x, pre_g, dtype, stream, nullptr);
}

// lcgan_conv_wgrad followed by lcgan_conv_wgrad_unprep(gwp, ..., gw) as one call: when the partial tiles meet through the slab, the
// reduction pass writes gw directly (one launch less per convolution backward, no gwp round trip); gwp is scratch either way
// and need NOT be cleared by the caller: the call clears it itself on the paths that accumulate into it (the big layers, which
// all take the slab, never touch it: 0.5 GB of fills per iteration at 256x256).
// wA / wBc: the WEIGHT's [wA][wBc][k][k] (= [A][Bc] of the gradient call, or its transpose with transposed = 1).
int lcgan_conv_wgrad_fused(const void* x, const void* g, float* gwp,
                           int B, int Hx, int Wx, int Cx, int Hg, int Wg, int Cg, int A, int Bc, int k, int stride,
                           const float* pre_x, const float* pre_g, int dtype,
                           float scale, int transposed, const float* w, const float* gwsq, float* gw, void* stream) {
  if (!gw) return LCGAN_EINVAL;
  UnprepArgs up = {scale, transposed, w, gwsq, gw};
  const int rc = conv_wgrad_impl(x, g, gwp, B, Hx, Wx, Cx, Hg, Wg, Cg, A, Bc, k, stride, pre_x, pre_g, dtype, stream, &up);
  if (rc == 1) return 0;
  if (rc != 0) return rc;
  return lcgan_conv_wgrad_unprep(gwp, transposed ? Bc : A, transposed ? A : Bc, k, scale, transposed, w, gwsq, gw, stream);
}

}  // extern "C"
