// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands: which (row, k) does byte j of lane l hold, and how do the scales apply?
// hipcc --offload-arch=gfx950 -O2 scripts/probes/mx_probe.hip -o /tmp/mx_probe && /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(const unsigned char* A, const unsigned char* B, const unsigned* sa, const unsigned* sb, float* D) {
  const int lane = threadIdx.x;
  // hypothesis: lane l holds A[row = l & 31][k = 32 (l >> 5) + j] in byte j (j = 0..31), B[k = 32 (l >> 5) + j][col = l & 31] likewise
  i32x8 a, b;
  const unsigned char* ap = A + (lane & 31) * 64 + 32 * (lane >> 5);       // A row-major [32][64]
  const unsigned char* bp = B + (lane & 31) * 64 + 32 * (lane >> 5);       // B stored [col][k] (i.e. B^T row-major)
  for (int i = 0; i < 8; ++i) { a[i] = ((const int*)ap)[i]; b[i] = ((const int*)bp)[i]; }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, (int)sa[lane], 0, (int)sb[lane]);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
    D[row * 32 + col] = c[r];
  }
}

static unsigned char enc(int v) {            // small integers as e4m3
  static const unsigned char t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};
  return v >= 0 ? t[v] : (unsigned char)(0x80 | t[-v]);
}
int run(int mode);
int main() { int rc = 0; for (int m = 0; m < 4; ++m) rc |= run(m); return rc; }
int run(int mode) {
  unsigned char hA[32 * 64], hB[32 * 64];
  int iA[32 * 64], iB[32 * 64];
  srand(1);
  for (int i = 0; i < 32 * 64; ++i) { iA[i] = rand() % 9 - 4; iB[i] = rand() % 7 - 3; hA[i] = enc(iA[i]); hB[i] = enc(iB[i]); }
  unsigned hsa[64], hsb[64];
  // mode 0: unit scales; 1: A scale depends on the k block only; 2: on the row only; 3: per (row, block) on A and (col, block) on B
  for (int l = 0; l < 64; ++l) {
    hsa[l] = mode == 0 ? 127 : mode == 1 ? 127 + (l >> 5) : mode == 2 ? 127 + ((l & 31) % 3) : 127 + ((l & 31) % 3) + (l >> 5);
    hsb[l] = mode == 3 ? 127 - ((l & 31) % 2) : 127;
  }
  unsigned char *dA, *dB; unsigned *dsa, *dsb; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
  float hD[32 * 32];
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int r = 0; r < 32; ++r)
    for (int c = 0; c < 32; ++c) {
      double ref = 0;
      for (int kb = 0; kb < 2; ++kb) {
        double s = 0;
        for (int j = 0; j < 32; ++j) s += (double)iA[r * 64 + kb * 32 + j] * iB[c * 64 + kb * 32 + j];
        ref += s * std::ldexp(1.0, (int)hsa[kb * 32 + r] - 127) * std::ldexp(1.0, (int)hsb[kb * 32 + c] - 127);
      }
      worst = std::fmax(worst, std::fabs(ref - hD[r * 32 + c]));
    }
  if (mode == 1) {      // which 32 of a row's 64 k values share the scale held by lanes 0-31 / 32-63?  group (h, q) = lane half h, byte half q
    const char* names[4] = {"by lane half (h)", "by byte half (q)", "h xor q", "all from lanes 0-31"};
    for (int hyp = 0; hyp < 4; ++hyp) {
      double w = 0;
      for (int r = 0; r < 32; ++r)
        for (int c = 0; c < 32; ++c) {
          double ref = 0;
          for (int h = 0; h < 2; ++h)
            for (int q = 0; q < 2; ++q) {
              double g = 0;
              for (int j = 0; j < 16; ++j) g += (double)iA[r * 64 + h * 32 + q * 16 + j] * iB[c * 64 + h * 32 + q * 16 + j];
              const int sel = hyp == 0 ? h : hyp == 1 ? q : hyp == 2 ? (h ^ q) : 0;
              ref += g * std::ldexp(1.0, (int)hsa[sel * 32 + r] - 127);
            }
          w = std::fmax(w, std::fabs(ref - hD[r * 32 + c]));
        }
      printf("   hypothesis '%s': max err %g\n", names[hyp], w);
    }
  }
  {
    double s0 = 0, s1 = 0;
    for (int j = 0; j < 32; ++j) { s0 += (double)iA[j] * iB[j]; s1 += (double)iA[32 + j] * iB[32 + j]; }
    double t0 = 0, t1 = 0;
    for (int j = 0; j < 32; ++j) { t0 += (double)iA[5 * 64 + j] * iB[7 * 64 + j]; t1 += (double)iA[5 * 64 + 32 + j] * iB[7 * 64 + 32 + j]; }
    printf("   partial sums (0,0): block0 %g block1 %g ; (5,7): block0 %g block1 %g ; scales a[0]=%u a[32]=%u a[5]=%u a[37]=%u b[0]=%u b[7]=%u\n", s0, s1, t0, t1, hsa[0], hsa[32], hsa[5], hsa[37], hsb[0], hsb[7]);
  }
  printf("mx probe mode %d: max |D - ref| = %g  (D[0][0] = %g, D[5][7] = %g)\n", mode, worst, hD[0], hD[5 * 32 + 7]);
  return worst == 0 ? 0 : 1;
}
