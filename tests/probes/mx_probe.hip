// Probe (not product): operand / scale lane maps of v_mfma_scale_f32_32x32x64_f8f6f4 and _16x16x128_ on gfx950, checked with
// exact integer data (cdna_hip_programming.md §3: "check the map with exact integer data before relying on it").
// Findings (MI355X, ROCm 7.2): accumulator = [first operand's row: (v&3)+8(v>>2)+4h][second operand's row: lane&31]; op_sel picks
// the byte of the scale register; a sum over k is invariant under any byte->k map shared by both operands, so the k map only
// shows with UNEQUAL block scales (test 3b): lane (r, h) bytes 0-15 are k = 16h.., bytes 16-31 are k = 32+16h.., and the scale of
// lane (r, h) applies to k-block h = bytes 16h..16h+15 of both lanes of the row.  v_cvt_pk_fp8_f32 is OCP e4m3 without saturation
// (480 -> NaN): clamp to +-448 first.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mx_probe tests/probes/mx_probe.hip && /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// e4m3 (OCP) encodings of small integers
static uint8_t enc(int v) {
  static const uint8_t tab[5] = {0x00, 0x38, 0x40, 0x44, 0x48};  // 0, 1, 2, 3, 4
  const int a = v < 0 ? -v : v;
  return tab[a] | (v < 0 ? 0x80 : 0);
}

// layout hypothesis: byte j of lane (r, h) holds k = kmap(h, j)
__global__ void k32(const uint8_t* A, const uint8_t* B, float* D, const int* sa, const int* sb, int hyp, int opsel) {
  // A: [32 rows][64 k] bytes, B: [32 cols][64 k] bytes (both K-contiguous), D: [32][32]
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  union { v8i v; uint8_t b[32]; } a, b;
  for (int j = 0; j < 32; ++j) {
    const int k = hyp == 0 ? 32 * h + j : (j < 16 ? 16 * h + j : 32 + 16 * h + (j - 16));
    a.b[j] = A[r * 64 + k];
    b.b[j] = B[r * 64 + k];
  }
  v16f c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, 0, sa[l], 0, sb[l]);
  else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, 1, sa[l], 2, sb[l]);
  for (int v = 0; v < 16; ++v) D[((v & 3) + 8 * (v >> 2) + 4 * h) * 32 + r] = c[v];
}
__global__ void k16(const uint8_t* A, const uint8_t* B, float* D, const int* sa, const int* sb, int hyp) {
  // A: [16 rows][128 k], B: [16 cols][128 k], D: [16][16]
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  union { v8i v; uint8_t b[32]; } a, b;
  for (int j = 0; j < 32; ++j) {
    const int k = hyp == 0 ? 32 * g + j : (j < 16 ? 16 * g + j : 64 + 16 * g + (j - 16));
    a.b[j] = A[r * 128 + k];
    b.b[j] = B[r * 128 + k];
  }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, b.v, c, 0, 0, 0, sa[l], 0, sb[l]);
  for (int v = 0; v < 4; ++v) D[(4 * g + v) * 16 + r] = c[v];
}
// fp8 conversion: what does v_cvt_pk_fp8_f32 produce for a few values (OCP e4m3 expected), and does it saturate?
__global__ void kcvt(const float* x, uint32_t* out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.f, 0, false) & 0xffffu;
}

int main() {
  const int M = 32, K = 64;
  std::vector<uint8_t> A(M * K), B(M * K);
  std::vector<int> Ai(M * K), Bi(M * K);
  srand(1);
  for (int i = 0; i < M * K; ++i) { Ai[i] = rand() % 9 - 4; Bi[i] = rand() % 9 - 4; A[i] = enc(Ai[i]); B[i] = enc(Bi[i]); }
  uint8_t *dA, *dB; float* dD; int *dsa, *dsb;
  hipMalloc(&dA, 16 * 128); hipMalloc(&dB, 16 * 128 > M * K ? 16 * 128 : M * K); hipMalloc(&dD, 32 * 32 * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
  hipFree(dA); hipMalloc(&dA, 4096); hipFree(dB); hipMalloc(&dB, 4096);
  std::vector<int> sa(64, 0x7f7f7f7f), sb(64, 0x7f7f7f7f);
  std::vector<float> D(32 * 32);
  auto run32 = [&](int hyp, int opsel) {
    hipMemcpy(dA, A.data(), M * K, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), M * K, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    k32<<<1, 64>>>(dA, dB, dD, dsa, dsb, hyp, opsel);
    hipMemcpy(D.data(), dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
  };
  // (1) K layout, unit scales: D[m][n] = sum_k A[m][k] B[n][k]
  for (int hyp = 0; hyp < 2; ++hyp) {
    run32(hyp, 0);
    int bad = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
      int s = 0; for (int k = 0; k < 64; ++k) s += Ai[m * 64 + k] * Bi[n * 64 + k];
      // kernel: A operand rows -> ? we stored D[row(v,h)][r]; with A = first operand the result is D[i = A row][j = B row]?  test both
      if ((int)D[m * 32 + n] != s) ++bad;
    }
    int badT = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
      int s = 0; for (int k = 0; k < 64; ++k) s += Ai[m * 64 + k] * Bi[n * 64 + k];
      if ((int)D[n * 32 + m] != s) ++badT;
    }
    printf("32x32x64 k-layout hyp %d (0: k=32h+j, 1: split halves): mismatches %d (D[Arow][Brow] at [(v..)+4h][lane&31]) / %d (transposed)\n", hyp, bad, badT);
  }
  // (2) scale map: all data ones; double ONE lane's A scale (exponent 128) and see which outputs move
  for (int i = 0; i < M * K; ++i) { A[i] = 0x38; B[i] = 0x38; }
  for (int lane : {0, 5, 37, 63}) {
    std::fill(sa.begin(), sa.end(), 0x7f7f7f7f); std::fill(sb.begin(), sb.end(), 0x7f7f7f7f);
    sa[lane] = 0x7f7f7f80;  // byte 0 = 128
    run32(0, 0);
    printf("scale_a lane %d byte0 x2: ", lane);
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) if (D[m * 32 + n] != 64.f) { printf("first changed D[%d][%d] = %g; ", m, n, D[m * 32 + n]); m = 99; break; }
    int cnt = 0; for (float v : D) cnt += v != 64.f;
    printf("%d outputs changed\n", cnt);
    std::fill(sa.begin(), sa.end(), 0x7f7f7f7f);
    sb[lane] = 0x7f7f7f80;
    run32(0, 0);
    printf("scale_b lane %d byte0 x2: ", lane);
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) if (D[m * 32 + n] != 64.f) { printf("first changed D[%d][%d] = %g; ", m, n, D[m * 32 + n]); m = 99; break; }
    cnt = 0; for (float v : D) cnt += v != 64.f;
    printf("%d outputs changed\n", cnt);
  }
  // (3) opsel: byte 1 of scale_a, byte 2 of scale_b
  std::fill(sa.begin(), sa.end(), 0x7f7f807f); std::fill(sb.begin(), sb.end(), 0x7f817f7f);
  run32(0, 1);
  printf("opsel a=1 (byte1 = 128), b=2 (byte2 = 129): D[0][0] = %g (expect 64 * 2 * 4 = 512)\n", D[0]);
  // (3b) one-hot A in the h = 1 half, zero blocks carrying scale byte 0 (2^-127): what vx_op_gemm_mx feeds for a one-hot row
  {
    for (int i = 0; i < M * K; ++i) { A[i] = 0x00; B[i] = enc(1 + (i % 64) % 4); }
    for (int m = 0; m < 32; ++m) A[m * 64 + 37] = 0x78;  // 256
    for (int l = 0; l < 64; ++l) { sa[l] = l < 32 ? 0x00000000 : 0x77777777; sb[l] = 0x7f7f7f7f; }
    run32(0, 0);
    printf("one-hot k=37 (value 256 x 2^-8), zero half with scale byte 0: D[0][0] = %g (expect %g)\n", D[0], (double)(1 + 37 % 4));
    for (int l = 0; l < 64; ++l) sa[l] = l < 32 ? 0x7f7f7f7f : 0x77777777;
    run32(0, 0);
    printf("same with scale byte 127 on the zero half: D[0][0] = %g\n", D[0]);
    for (int l = 0; l < 64; ++l) sa[l] = l < 32 ? 0x01010101 : 0x77777777;
    run32(0, 0);
    printf("same with scale byte 1 on the zero half: D[0][0] = %g\n", D[0]);
  }
  // (4) 16x16x128
  {
    std::vector<uint8_t> A2(16 * 128), B2(16 * 128); std::vector<int> A2i(16 * 128), B2i(16 * 128);
    for (int i = 0; i < 16 * 128; ++i) { A2i[i] = rand() % 9 - 4; B2i[i] = rand() % 9 - 4; A2[i] = enc(A2i[i]); B2[i] = enc(B2i[i]); }
    std::fill(sa.begin(), sa.end(), 0x7f7f7f7f); std::fill(sb.begin(), sb.end(), 0x7f7f7f7f);
    hipMemcpy(dA, A2.data(), 16 * 128, hipMemcpyHostToDevice); hipMemcpy(dB, B2.data(), 16 * 128, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 2; ++hyp) {
      k16<<<1, 64>>>(dA, dB, dD, dsa, dsb, hyp);
      hipMemcpy(D.data(), dD, 16 * 16 * 4, hipMemcpyDeviceToHost);
      int bad = 0, badT = 0;
      for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        int s = 0; for (int k = 0; k < 128; ++k) s += A2i[m * 128 + k] * B2i[n * 128 + k];
        bad += (int)D[m * 16 + n] != s; badT += (int)D[n * 16 + m] != s;
      }
      printf("16x16x128 k-layout hyp %d: mismatches %d / %d (transposed)\n", hyp, bad, badT);
    }
    for (int i = 0; i < 16 * 128; ++i) { A2[i] = 0x38; B2[i] = 0x38; }
    hipMemcpy(dA, A2.data(), 16 * 128, hipMemcpyHostToDevice); hipMemcpy(dB, B2.data(), 16 * 128, hipMemcpyHostToDevice);
    for (int lane : {3, 19, 50}) {
      std::fill(sa.begin(), sa.end(), 0x7f7f7f7f); sa[lane] = 0x7f7f7f80;
      hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice);
      k16<<<1, 64>>>(dA, dB, dD, dsa, dsb, 0);
      hipMemcpy(D.data(), dD, 16 * 16 * 4, hipMemcpyDeviceToHost);
      int cnt = 0, fm = -1, fn = -1; for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) if (D[m * 16 + n] != 128.f) { if (!cnt) { fm = m; fn = n; } ++cnt; }
      printf("16x16x128 scale_a lane %d x2: %d outputs changed, first D[%d][%d] = %g\n", lane, cnt, fm, fn, cnt ? D[fm * 16 + fn] : 0.f);
    }
  }
  // (5) conversion
  {
    float xs[12] = {0.f, 1.f, -1.f, 1.5f, 448.f, 449.f, 480.f, 1000.f, 0.0019f, 0.001f, 3.3f, -0.3f};
    float* dx; uint32_t* dout; hipMalloc(&dx, 64); hipMalloc(&dout, 64);
    hipMemcpy(dx, xs, 48, hipMemcpyHostToDevice);
    kcvt<<<1, 64>>>(dx, dout, 12);
    uint32_t o[12]; hipMemcpy(o, dout, 48, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; ++i) printf("cvt_pk_fp8_f32(%g) = 0x%02x\n", xs[i], o[i] & 0xff);
  }
  hipError_t e = hipDeviceSynchronize();
  printf("done: %s\n", hipGetErrorString(e));
  return 0;
}
