// Layout probe for v_mfma_f32_32x32x16_bf16 on gfx950: which (row, k) does each lane's 8-element operand hold, and which
// (row, col) does each accumulator register hold?  Compares the builtin against a float reference for the assumed layout
//   A: lane l → row l%32, k = 8*(l/32) .. +7      B: lane l → col l%32, k = 8*(l/32) .. +7
//   D: lane l, reg r → row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31
// and measures the error of the 3-way bf16 split (6 products) against float64.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float *A, const float *B, float *D, float *D3) {
  const int l = threadIdx.x, r32 = l & 31, h = l >> 5;
  bf16x8 a, b, a2, b2, a3, b3;
  for (int i = 0; i < 8; i++) {
    float av = A[r32 * 16 + 8 * h + i], bv = B[(8 * h + i) * 32 + r32];
    __bf16 a1 = (__bf16)av; float ar = av - (float)a1; __bf16 a2s = (__bf16)ar; float ar2 = ar - (float)a2s;
    __bf16 b1 = (__bf16)bv; float br = bv - (float)b1; __bf16 b2s = (__bf16)br; float br2 = br - (float)b2s;
    a[i] = a1; a2[i] = a2s; a3[i] = (__bf16)ar2;
    b[i] = b1; b2[i] = b2s; b3[i] = (__bf16)br2;
  }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + r32] = acc[r];
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b3, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) D3[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + r32] = c[r];
}

static float bf16_round(float x) { return (float)(__bf16)x; }

int main() {
  std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), D3(32 * 32);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f * 4.0f - 2.0f; };
  for (auto &v : A) v = rnd() * 10.0f;
  for (auto &v : B) v = rnd();
  float *dA, *dB, *dD, *dD3;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4); hipMalloc(&dD3, D3.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dD3);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(D3.data(), dD3, D3.size() * 4, hipMemcpyDeviceToHost);
  double e1 = 0, e3 = 0, mag = 0;
  for (int i = 0; i < 32; i++)
    for (int j = 0; j < 32; j++) {
      double ref1 = 0, ref = 0, mabs = 0;
      for (int k = 0; k < 16; k++) {
        ref1 += (double)bf16_round(A[i * 16 + k]) * (double)bf16_round(B[k * 32 + j]);
        ref += (double)A[i * 16 + k] * (double)B[k * 32 + j];
        mabs += fabs((double)A[i * 16 + k] * (double)B[k * 32 + j]);
      }
      e1 = fmax(e1, fabs(D[i * 32 + j] - ref1));
      e3 = fmax(e3, fabs(D3[i * 32 + j] - ref) / mabs);
      mag = fmax(mag, mabs);
    }
  printf("single bf16 MFMA vs bf16-rounded reference: max abs err %.3e (layout %s)\n", e1, e1 < 1e-3 ? "OK" : "WRONG");
  printf("3-way split (6 products) vs float64 of the float inputs: max err / sum|terms| = %.3e (2^-24 = 5.96e-8), sum|terms| up to %.1f\n", e3, mag);
  return e1 < 1e-3 ? 0 : 1;
}
