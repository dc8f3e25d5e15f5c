// bf16 MFMA kernels for the compute-bound row paths (AR prefill, NAR stages):
//   C[M,N] = A[M,K] . W[N,K]^T with fused bias / ReLU / residual epilogues.
// Both operands are K-contiguous, which is exactly the v_mfma_f32_32x32x16_bf16 fragment shape:
// lane (r = l&31, h = l>>5) holds 8 consecutive k (one 16-byte LDS read) of row r of A and of
// row r of W (cdna_hip_programming.md §3 "A/B operand lane maps").
#pragma once
#include "common.hpp"
#include "rows_kernels.hpp"

namespace vx {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// 128x128x64 tile, 4 waves as 2(M) x 2(N), each wave 64x64 = 2x2 MFMA tiles of 32x32.
// LDS: double-buffered [128 rows][64 bf16] images of A and W (128-byte rows), 16-byte chunks
// XOR-swizzled by (row & 7) so the 16 rows a ds_read_b128 lane group touches fall on 16
// distinct 16-byte slots of the 256-byte bank row (T2).  Global->register->LDS staging with
// the next tile's global loads issued before the current tile's MFMAs (T14 split).
template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(256) void mfma_gemm_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                        const float* __restrict__ bias, void* __restrict__ Cv, int M,
                                                        int N, int K) {
  constexpr int BM = 128, BN = 128, BK = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // layout: [buf][A|W][128 rows * 128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  // staging: 1024 16-byte chunks per operand tile, 4 per thread
  uint4 ra[4], rw[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + i * 256, row = q >> 3, c = q & 7;
      const int gm = m0 + row, gn = n0 + row;
      ra[i] = (gm < M) ? ld16(A + (size_t)gm * K + k0 + c * 8) : make_uint4(0u, 0u, 0u, 0u);
      rw[i] = (gn < N) ? ld16(W + (size_t)gn * K + k0 + c * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* ba = lds + buf * 32768;
    unsigned char* bw = ba + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + i * 256, row = q >> 3, c = q & 7;
      const int off = row * 128 + ((c ^ (row & 7)) << 4);
      *reinterpret_cast<uint4*>(ba + off) = ra[i];
      *reinterpret_cast<uint4*>(bw + off) = rw[i];
    }
  };

  const int nk = K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const unsigned char* ba = lds + cur * 32768;
    const unsigned char* bw = ba + 16384;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + r;
        fa[i] = *reinterpret_cast<const bf16x8_t*>(ba + row * 128 + (((ks * 2 + h) ^ (row & 7)) << 4));
        const int col = wn * 64 + i * 32 + r;
        fb[i] = *reinterpret_cast<const bf16x8_t*>(bw + col * 128 + (((ks * 2 + h) ^ (col & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }

  // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + r;
    const float bv = (EPI != GE_PLAIN && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int m = m0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (m < M && n < N) {
          float x = acc[i][j][v] + bv;
          if (EPI == GE_RELU) x = fmaxf(x, 0.f);
          if (OUT_F32) {
            float* c = reinterpret_cast<float*>(Cv) + (size_t)m * N + n;
            *c = (EPI == GE_RESID) ? (*c + x) : x;
          } else {
            reinterpret_cast<bf16*>(Cv)[(size_t)m * N + n] = (bf16)x;
          }
        }
      }
    }
  }
}

static inline int mfma_gemm_dispatch(const bf16* A, const bf16* W, const float* bias, void* C, int M, int N, int K,
                                     int epi, bool out_f32, hipStream_t s) {
  if (K % 64 != 0 || N % 8 != 0) {
    // shapes outside the tiling: scalar-FMA fallback
    dim3 g((N + 63) / 64, (M + 63) / 64);
    if (epi == GE_RESID) gemm_simple_kernel<bf16, float, GE_RESID><<<g, 256, 0, s>>>(A, W, bias, (float*)C, M, N, K);
    else if (epi == GE_PLAIN) gemm_simple_kernel<bf16, float, GE_PLAIN><<<g, 256, 0, s>>>(A, W, bias, (float*)C, M, N, K);
    else if (epi == GE_BIAS && out_f32) gemm_simple_kernel<bf16, float, GE_BIAS><<<g, 256, 0, s>>>(A, W, bias, (float*)C, M, N, K);
    else if (epi == GE_RELU && out_f32) gemm_simple_kernel<bf16, float, GE_RELU><<<g, 256, 0, s>>>(A, W, bias, (float*)C, M, N, K);
    else if (epi == GE_BIAS) gemm_simple_kernel<bf16, bf16, GE_BIAS><<<g, 256, 0, s>>>(A, W, bias, (bf16*)C, M, N, K);
    else gemm_simple_kernel<bf16, bf16, GE_RELU><<<g, 256, 0, s>>>(A, W, bias, (bf16*)C, M, N, K);
    return 0;
  }
  dim3 grid((N + 127) / 128, (M + 127) / 128);
  const size_t lds = 65536;
#define MG(E, F)                                                                                             \
  do {                                                                                                       \
    static bool attr_done = false;                                                                           \
    if (!attr_done) {                                                                                        \
      (void)hipFuncSetAttribute((const void*)mfma_gemm_kernel<E, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_done = true;                                                                                      \
    }                                                                                                        \
    mfma_gemm_kernel<E, F><<<grid, 256, lds, s>>>(A, W, bias, C, M, N, K);                                   \
  } while (0)
  if (epi == GE_RESID) MG(GE_RESID, true);
  else if (epi == GE_PLAIN) MG(GE_PLAIN, true);
  else if (epi == GE_BIAS && out_f32) MG(GE_BIAS, true);
  else if (epi == GE_RELU && out_f32) MG(GE_RELU, true);
  else if (epi == GE_BIAS) MG(GE_BIAS, false);
  else MG(GE_RELU, false);
#undef MG
  return 0;
}

// MFMA flash attention is not built yet: bf16 rows use the tiled scalar-FMA kernel.
static inline int mfma_attn_dispatch(const bf16* qkv, bf16* out, int M, int d, int H, int text_len, float scale,
                                     hipStream_t s) {
  dim3 grid((M + 63) / 64, H);
  attn_rows_simple_kernel<bf16, 64><<<grid, 256, 0, s>>>(qkv, out, M, d, text_len, scale);
  return 0;
}

}  // namespace vx
