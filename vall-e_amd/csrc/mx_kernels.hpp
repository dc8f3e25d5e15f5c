// MXFP8 row GEMMs for the NAR stages (BASELINE configs[4]: "fp8 MFMA QKV/FFN path for NAR stages"):
//   C[M,N] = A[M,K] . W[N,K]^T with both operands in OCP e4m3 (gfx950's fp8: e4m3fn, not MI300's fnuz) and one E8M0
//   scale per 32 consecutive k of every row (OCP MX block scaling), multiplied by v_mfma_scale_f32_32x32x64_f8f6f4:
//   the matrix core applies the two block scales itself, at twice the bf16 rate per clock, fp32 accumulate.
// Why block scales and not per-row / per-channel ones: the A operand of FFN2 is FFN1's output, whose row maximum spans
// sixteen 256-wide tiles of sixteen workgroups; a 32-wide block lives inside one accumulator fragment, so FFN1's epilogue
// can quantise its own output (and the LayerNorm kernel its own row) with no extra pass over the activations.
//
// Operand storage:  bytes [rows][K] e4m3, K-contiguous;  scales [K/32][ld_s] E8M0 bytes (k-block major, ld_s >= rows rounded
// up to 256 and zero-padded; inside a k-block's run row r sits at mx_spos(r)), so that the scales one 256-row tile needs for one
// 64-deep K stage are two runs of 256 bytes.
// Quantisation (mx_block_scale / mx_pack4, mirrored by tests/mx_ref.py): with E the biased exponent of the block's largest
// magnitude, scale byte = max(E, 8) - 8 (i.e. 2^(floor(log2 amax) - 8): e4m3's largest binade), values multiplied by the
// exact inverse power of two, clamped to +-448 and rounded to nearest even by v_cvt_pk_fp8_f32.
#pragma once
#include "common.hpp"
#include "mfma_kernels.hpp"

namespace vx {

typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float group8_max_dpp(float v) {  // max over aligned groups of 8 lanes, in every lane of the group
  v = fmaxf(v, dpp_f<0xB1>(v, v));
  v = fmaxf(v, dpp_f<0x4E>(v, v));
  v = fmaxf(v, dpp_f<0x141>(v, v));
  return v;
}
// Position of a row's scale inside a k-block's run of the scale array: rows r, r + 32, r + 64, r + 96 of every 128-row group sit
// in one dword, so a lane of the GEMM reads the scales of its four A fragments (two W fragments) with one LDS read and picks
// the byte with the MFMA's op_sel.
__host__ __device__ __forceinline__ int mx_spos(int row) { return (row & ~127) | ((row & 31) << 2) | ((row >> 5) & 3); }

__device__ __forceinline__ uint32_t mx_block_scale(float amax, float& inv) {
  const uint32_t E = (__float_as_uint(amax) >> 23) & 0xffu;
  const uint32_t byte = max(E, 8u) - 8u;
  inv = __uint_as_float((254u - byte) << 23);  // 2^(127 - byte), exact
  return byte;
}
__device__ __forceinline__ uint32_t mx_pack4(float a, float b, float c, float d, float inv) {
  a = fminf(fmaxf(a * inv, -448.f), 448.f); b = fminf(fmaxf(b * inv, -448.f), 448.f);
  c = fminf(fmaxf(c * inv, -448.f), 448.f); d = fminf(fmaxf(d * inv, -448.f), 448.f);
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
  return (uint32_t)p;
}

// rows of fp32 -> MXFP8 (weights at load time, operands of the op-level test).  One wave per row, lane l owns k = 4 (64 i + l) .. +3.
__global__ __launch_bounds__(256) void mx_quant_rows_kernel(const float* __restrict__ x, uint8_t* __restrict__ q,
                                                            uint8_t* __restrict__ sc, int rows, int K, int ld_s) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  for (int k0 = 0; k0 < K; k0 += 256) {
    const int k = k0 + lane * 4;
    const bool ok = k < K;  // K % 32 == 0: a block is all in or all out
    const float4 v = ok ? *reinterpret_cast<const float4*>(x + (size_t)r * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float am = group8_max_dpp(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    float inv;
    const uint32_t sb = mx_block_scale(am, inv);
    if (ok) {
      *reinterpret_cast<uint32_t*>(q + (size_t)r * K + k) = mx_pack4(v.x, v.y, v.z, v.w, inv);
      if ((lane & 7) == 0) sc[(size_t)(k >> 5) * ld_s + mx_spos(r)] = (uint8_t)sb;
    }
  }
}

// (Adaptive)LayerNorm with an MXFP8 result: layernorm_rows_kernel's arithmetic (rows_kernels.hpp; same fold prologue for the
// split-K slabs of the preceding GEMM), the normalised row quantised in registers: the 32-wide blocks are aligned groups of 8 lanes.
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_rows_mx_kernel(const float* x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ ada_w,
                                                                const float* __restrict__ ada_b, uint8_t* __restrict__ q,
                                                                uint8_t* __restrict__ sc, int rows, int d, int ld_s) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + (size_t)r * d;
  const bool ada = ada_w != nullptr;
  float4 v[MAXV], g[MAXV], b[MAXV], w[MAXV], c[MAXV];
  int kk[MAXV];
  bool ok[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int k = (i * 64 + lane) * 4;
    ok[i] = k < d;
    kk[i] = ok[i] ? k : 0;
    v[i] = *reinterpret_cast<const float4*>(xr + kk[i]);
    g[i] = *reinterpret_cast<const float4*>(gamma + kk[i]);
    b[i] = *reinterpret_cast<const float4*>(beta + kk[i]);
    if (ada) {
      w[i] = *reinterpret_cast<const float4*>(ada_w + kk[i]);
      c[i] = *reinterpret_cast<const float4*>(ada_b + kk[i]);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) s += ok[i] ? (v[i].x + v[i].y) + (v[i].z + v[i].w) : 0.f;
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
    ss += ok[i] ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)d + LN_EPS);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    float o[4] = {(v[i].x - mean) * rstd * g[i].x + b[i].x, (v[i].y - mean) * rstd * g[i].y + b[i].y,
                  (v[i].z - mean) * rstd * g[i].z + b[i].z, (v[i].w - mean) * rstd * g[i].w + b[i].w};
    if (ada) {
      o[0] = __fadd_rn(__fmul_rn(w[i].x, o[0]), c[i].x); o[1] = __fadd_rn(__fmul_rn(w[i].y, o[1]), c[i].y);
      o[2] = __fadd_rn(__fmul_rn(w[i].z, o[2]), c[i].z); o[3] = __fadd_rn(__fmul_rn(w[i].w, o[3]), c[i].w);
    }
    if (!ok[i]) { o[0] = o[1] = o[2] = o[3] = 0.f; }
    const float am = group8_max_dpp(fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    float inv;
    const uint32_t sb = mx_block_scale(am, inv);
    if (ok[i]) {
      *reinterpret_cast<uint32_t*>(q + (size_t)r * d + kk[i]) = mx_pack4(o[0], o[1], o[2], o[3], inv);
      if ((lane & 7) == 0) sc[(size_t)(kk[i] >> 5) * ld_s + mx_spos(r)] = (uint8_t)sb;
    }
  }
}

// ---- 256x256 tile MXFP8 GEMM: mfma256_kernel's structure (persistent workgroups, XCD-aware tile walk, global -> LDS by
// LDS-DMA into a 4-deep ring of 64-byte-row stages, counted vmcnt + raw barrier, fragments double-buffered in registers) with
// 64 k per stage instead of 32 and ONE v_mfma_scale_f32_32x32x64_f8f6f4 per (A fragment, W fragment) per stage:
// 8 waves as 2 (M) x 4 (N), each 128 x 64 = 4 x 2 fragments of 32 x 32, 8 MFMAs of 64 cycles per stage = the matrix time of the
// bf16 kernel's 32 MFMAs of 16 cycles at twice the K.
// Lane maps (tests/probes/mx_probe.hip, exact data with unequal block scales): lane (r = l & 31, h = l >> 5) holds, of row r,
// k = 16 h .. 16 h + 15 in operand bytes 0-15 and k = 32 + 16 h .. 32 + 16 h + 15 in bytes 16-31; the scale byte lane (r, h)
// supplies applies to k-block h of the row (k = 32 h .. 32 h + 31), i.e. to bytes 16 h .. 16 h + 15 of BOTH lanes of the row -
// not to the lane's own 32 bytes (with equal scales per row the two readings cannot be told apart).
// The MFMA is issued as (W fragment, A fragment), so the accumulator has m on the lane and n = (v&3) + 8 (v>>2) + 4 h in its
// registers: 4 consecutive n per register quad.
// Per stage the ring slot also carries the stage's 2 x 256 scale bytes of each operand (every wave brings 128 of the 1024 bytes
// with one 2-byte-per-lane LDS-DMA).
enum MxOut { MX_OUT_F32 = 0, MX_OUT_BF16 = 1, MX_OUT_MX = 2 };

template <int EPI, int OUT>
__global__ __launch_bounds__(512) void mx256_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ SA, int lda_s,
                                                    const uint8_t* __restrict__ W, const uint8_t* __restrict__ SW, int ldw_s,
                                                    const float* __restrict__ bias, void* __restrict__ Cv,
                                                    uint8_t* __restrict__ SC, int ldc_s, int M, int N, int K,
                                                    bf16* __restrict__ vt, int vt_n0, int vt_ld, int ntn, int ntiles) {
  constexpr int STAGE = 32768 + 1024;  // A 16 KB | W 16 KB | A scales 2 x 256 | W scales 2 x 256
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // `wave` in an SGPR: the branches on it below enclose scalar instructions (s_waitcnt / s_barrier / the LDS-DMA's M0 setup),
  // which ignore EXEC - under a branch the compiler takes for divergent a wave would run BOTH sides' waits and barriers
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int G = gridDim.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  auto tile_of = [&](int v) {
    const int xcd = v & 7, loc = v >> 3;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  };
  const int cnt = (ntiles - (int)blockIdx.x + G - 1) / G;

  // one stage = 2048 16-byte slots (A: 1024, W: 1024), 4 per thread; slot q -> row q>>2, position q&3 holds source chunk
  // (q&3) ^ ((-(row>>2))&3)  (rows are 64 bytes: the swizzle of mfma256_kernel, conflict-free for these fragment reads too)
  const int q0 = tid, q1 = tid + 512;
  const int rowa0 = q0 >> 2, rowa1 = q1 >> 2;
  const int ca0 = ((q0 & 3) ^ ((0 - (rowa0 >> 2)) & 3)) * 16, ca1 = ((q1 & 3) ^ ((0 - (rowa1 >> 2)) & 3)) * 16;
  const int d0 = (q0 - lane) * 16, d1 = (q1 - lane) * 16;
  const int nk = K / 64;

  const uint8_t *srcA0, *srcA1, *srcW0, *srcW1, *srcS;
  int l_ord = 0, l_k = 0, l_slot = 0;
  size_t s_step = 0;  // bytes between two stages in the scale array this wave streams
  auto set_load_tile = [&](int ord) {
    const int tile = tile_of((int)blockIdx.x + min(ord, cnt - 1) * G);
    const int mt = tile / ntn, nt = tile - mt * ntn;
    srcA0 = A + (size_t)min(mt * 256 + rowa0, M - 1) * K + ca0;
    srcA1 = A + (size_t)min(mt * 256 + rowa1, M - 1) * K + ca1;
    srcW0 = W + (size_t)min(nt * 256 + rowa0, N - 1) * K + ca0;
    srcW1 = W + (size_t)min(nt * 256 + rowa1, N - 1) * K + ca1;
    // scales: the stage's block is [A kb0 | A kb1 | W kb0 | W kb1] x 256 bytes = one 16-byte LDS-DMA load of one wave; every
    // wave issues it (same bytes to the same place), so that all waves run the same five loads per stage: one counted wait, no
    // branch in the loop, and only 16-byte LDS-DMA in the kernel (a narrower one next to the ds_reads made hipcc drain vmcnt(0))
    {
      const int part = lane >> 4, off = (lane & 15) * 16;
      if (part < 2) { srcS = SA + (size_t)part * lda_s + mt * 256 + off; s_step = (size_t)2 * lda_s; }
      else { srcS = SW + (size_t)(part - 2) * ldw_s + nt * 256 + off; s_step = (size_t)2 * ldw_s; }
    }
  };
  auto stage = [&]() {
    unsigned char* base = lds + (l_slot & 3) * STAGE;
    const int k0 = l_k * 64;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA0 + k0),
                                     (__attribute__((address_space(3))) void*)(base + d0), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA1 + k0),
                                     (__attribute__((address_space(3))) void*)(base + d1), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW0 + k0),
                                     (__attribute__((address_space(3))) void*)(base + 16384 + d0), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW1 + k0),
                                     (__attribute__((address_space(3))) void*)(base + 16384 + d1), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcS + (size_t)l_k * s_step),
                                     (__attribute__((address_space(3))) void*)(base + 32768), 16, 0, 0);
    ++l_slot;
    if (++l_k == nk) { l_k = 0; set_load_tile(++l_ord); }
  };

  // Registers: 128 accumulators leave room for ONE and a half stages of fragments, not two: the W fragments (and the scale
  // dwords) of the current and of the next stage live in two named sets, the A fragments in two halves (rows i = 0,1 / 2,3) that
  // are refilled one half-stage (4 MFMAs = 256 matrix cycles) ahead of their use.
  f32x16_t acc[4][2];
  u32x4 w0[2][2], w1[2][2], aX[2][2], aY[2][2];
  unsigned sw0 = 0, sw1 = 0, sa0 = 0, sa1 = 0;
  auto frag = [](const u32x4 (&p)[2]) {
    i32x8_t v;
    v[0] = (int)p[0].x; v[1] = (int)p[0].y; v[2] = (int)p[0].z; v[3] = (int)p[0].w;
    v[4] = (int)p[1].x; v[5] = (int)p[1].y; v[6] = (int)p[1].z; v[7] = (int)p[1].w;
    return v;
  };
  auto read_w = [&](int slot, u32x4 (&w)[2][2], unsigned& sw, unsigned& sa) {
    const unsigned char* ba = lds + (slot & 3) * STAGE;
    const unsigned char* bw = ba + 16384;
    const unsigned char* bs = ba + 32768 + h * 256;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wn * 64 + j * 32 + r, sz = (0 - (row >> 2)) & 3;
      w[j][0] = *reinterpret_cast<const u32x4*>(bw + row * 64 + ((h ^ sz) << 4));        // k = 16 h .. +15   (block 0)
      w[j][1] = *reinterpret_cast<const u32x4*>(bw + row * 64 + (((2 + h) ^ sz) << 4));  // k = 32 + 16 h .. (block 1)
    }
    // the scale dwords of this lane: bytes i = 0..3 are rows r + 32 i of the wave's 128-row group (mx_spos)
    sa = *reinterpret_cast<const unsigned*>(bs + wm * 128 + r * 4);
    sw = *reinterpret_cast<const unsigned*>(bs + 512 + (wn >> 1) * 128 + r * 4) >> ((wn & 1) * 16);  // bytes 0,1 = fragments j = 0,1
  };
  auto read_a = [&](int slot, int half, u32x4 (&a)[2][2]) {
    const unsigned char* ba = lds + (slot & 3) * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm * 128 + (2 * half + i) * 32 + r, sz = (0 - (row >> 2)) & 3;
      a[i][0] = *reinterpret_cast<const u32x4*>(ba + row * 64 + ((h ^ sz) << 4));
      a[i][1] = *reinterpret_cast<const u32x4*>(ba + row * 64 + (((2 + h) ^ sz) << 4));
    }
  };
  // 4 MFMAs: A fragments 2 half .. 2 half + 1 against both W fragments (op_sel = byte of the scale dword)
#define MX_MM(HALF, W, SW, A, SA)                                                                                                   \
  do {                                                                                                                             \
    acc[2 * HALF][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(W[0]), frag(A[0]), acc[2 * HALF][0], 0, 0, 0, (int)SW, 2 * HALF, (int)SA);         \
    acc[2 * HALF][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(W[1]), frag(A[0]), acc[2 * HALF][1], 0, 0, 1, (int)SW, 2 * HALF, (int)SA);         \
    acc[2 * HALF + 1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(W[0]), frag(A[1]), acc[2 * HALF + 1][0], 0, 0, 0, (int)SW, 2 * HALF + 1, (int)SA); \
    acc[2 * HALF + 1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(W[1]), frag(A[1]), acc[2 * HALF + 1][1], 0, 0, 1, (int)SW, 2 * HALF + 1, (int)SA); \
  } while (0)
  // counted wait: the youngest stage (5 loads per wave) may stay in flight across the barrier
  auto wait_barrier = [&]() { asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory"); };

  // Stage s of the stream lives in ring slot s & 3.  On stage s: [issue stage s+3] [read A rows 2,3 of s] [MFMAs rows 0,1]
  // [read W + scales + A rows 0,1 of stage s+1] [MFMAs rows 2,3] [vmcnt: stage s+2 landed] [barrier].  Slot (s+3)&3 = (s-1)&3
  // was last read (A rows 2,3) during stage s-1, whose MFMAs consumed those reads before that stage's barrier.
  set_load_tile(0);
  stage();
  stage();
  stage();
  wait_barrier();  // stages 0 and 1 landed everywhere
  read_w(0, w0, sw0, sa0);
  read_a(0, 0, aX);
  int c_slot = 0;
  for (int ord = 0; ord < cnt; ++ord) {
    const int tile = tile_of((int)blockIdx.x + ord * G);
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int m0 = mt * 256, n0 = nt * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    for (int t = 0; t < nk; t += 2) {  // nk is even (K % 128 == 0)
      stage();
      read_a(c_slot, 1, aY);
      MX_MM(0, w0, sw0, aX, sa0);
      read_w(c_slot + 1, w1, sw1, sa1);
      read_a(c_slot + 1, 0, aX);
      MX_MM(1, w0, sw0, aY, sa0);
      wait_barrier();
      stage();
      read_a(c_slot + 1, 1, aY);
      MX_MM(0, w1, sw1, aX, sa1);
      read_w(c_slot + 2, w0, sw0, sa0);
      read_a(c_slot + 2, 0, aX);
      MX_MM(1, w1, sw1, aY, sa1);
      wait_barrier();
      c_slot += 2;
    }
#undef MX_MM

    // epilogue: acc[i][j][v] = C[m = m0 + wm*128 + i*32 + r][n = n0 + wn*64 + j*32 + (v&3) + 8*(v>>2) + 4*h]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nf = n0 + wn * 64 + j * 32;  // N % 256 == 0: always in range
      float4 bq[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) bq[qd] = EPI != GE_PLAIN ? *reinterpret_cast<const float4*>(bias + nf + 8 * qd + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + r;
        float x[16];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          x[4 * qd] = acc[i][j][4 * qd] + bq[qd].x; x[4 * qd + 1] = acc[i][j][4 * qd + 1] + bq[qd].y;
          x[4 * qd + 2] = acc[i][j][4 * qd + 2] + bq[qd].z; x[4 * qd + 3] = acc[i][j][4 * qd + 3] + bq[qd].w;
        }
        if (EPI == GE_RELU) {
#pragma unroll
          for (int v = 0; v < 16; ++v) x[v] = fmaxf(x[v], 0.f);
        }
        if (OUT == MX_OUT_MX) {  // the row's 32 columns of this fragment are one MX block: 16 here, 16 in the other half-wave
          float am = 0.f;
#pragma unroll
          for (int v = 0; v < 16; ++v) am = fmaxf(am, fabsf(x[v]));
          am = fmaxf(am, xor32_f(am));
          float inv;
          const uint32_t sb = mx_block_scale(am, inv);
          if (m < M) {
            uint8_t* cp = reinterpret_cast<uint8_t*>(Cv) + (size_t)m * N + nf + 4 * h;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
              *reinterpret_cast<uint32_t*>(cp + 8 * qd) = mx_pack4(x[4 * qd], x[4 * qd + 1], x[4 * qd + 2], x[4 * qd + 3], inv);
            if (h == 0) SC[(size_t)(nf >> 5) * ldc_s + mx_spos(m)] = (uint8_t)sb;
          }
        } else if (m < M) {
          if (OUT == MX_OUT_F32) {
            float* cp = reinterpret_cast<float*>(Cv) + (size_t)m * N + nf + 4 * h;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              float4* p4 = reinterpret_cast<float4*>(cp + 8 * qd);
              if (EPI == GE_RESID) {
                const float4 o = *p4;
                *p4 = make_float4(o.x + x[4 * qd], o.y + x[4 * qd + 1], o.z + x[4 * qd + 2], o.w + x[4 * qd + 3]);
              } else {
                *p4 = make_float4(x[4 * qd], x[4 * qd + 1], x[4 * qd + 2], x[4 * qd + 3]);
              }
            }
          } else {
            bf16* cp = reinterpret_cast<bf16*>(Cv) + (size_t)m * N + nf + 4 * h;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              union { bf16 e[4]; uint2 u; } pk;
#pragma unroll
              for (int v = 0; v < 4; ++v) pk.e[v] = (bf16)x[4 * qd + v];
              *reinterpret_cast<uint2*>(cp + 8 * qd) = pk.u;
              if (vt != nullptr && nf >= vt_n0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) vt[(size_t)(nf + 8 * qd + 4 * h + v - vt_n0) * vt_ld + m] = pk.e[v];
              }
            }
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stages issued past the last tile
}

// ---- the same GEMM with FULL-LINE staging: a stage is 128 k (128-byte rows, two MFMA k-steps), two LDS buffers.  One
// LDS-DMA wave-instruction then covers 8 rows x one whole 128-byte line each instead of 16 rows x half a line: half the texture /
// L1 requests for the same bytes (cdna_hip_programming.md §5: operands "through LDS in full 128-B lines"; the 64-byte-row ring above
// ran at ~40 GB/s per CU, the matrix cores need ~66).  Chunk swizzle for 128-byte rows: position = chunk ^ ((row >> 1) & 7).
// Per stage and wave: 9 LDS-DMA loads (4 A, 4 W, 1 KB of scales), 24 fragment reads, 16 MFMAs, ONE barrier.
template <int EPI, int OUT>
__global__ __launch_bounds__(512) void mx256w_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ SA, int lda_s,
                                                     const uint8_t* __restrict__ W, const uint8_t* __restrict__ SW, int ldw_s,
                                                     const float* __restrict__ bias, void* __restrict__ Cv,
                                                     uint8_t* __restrict__ SC, int ldc_s, int M, int N, int K,
                                                     bf16* __restrict__ vt, int vt_n0, int vt_ld, int ntn, int ntiles) {
  constexpr int STAGE = 65536 + 2048;  // A 32 KB | W 32 KB | A scales 4 x 256 | W scales 4 x 256
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int G = gridDim.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  auto tile_of = [&](int v) {
    const int xcd = v & 7, loc = v >> 3;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  };
  const int cnt = (ntiles - (int)blockIdx.x + G - 1) / G;

  // one operand stage = 2048 16-byte slots, 4 per thread: slot q = tid + 512 i -> row (tid >> 3) + 64 i, position tid & 7, which
  // holds source chunk (tid & 7) ^ ((row >> 1) & 7) = the same chunk for all four (64 i does not reach bits 1-3 of the row)
  const int row0 = tid >> 3;
  const int csrc = ((tid & 7) ^ ((row0 >> 1) & 7)) * 16;
  const int dbase = (tid - lane) * 16;  // wave-uniform LDS offset of lane 0's slot (slot i: + 8192 i)
  const int nk = K / 128;
  int ra[4], rw[4];                     // clamped global rows of this thread's four slots (current load tile)
  size_t s_off = 0, s_step = 0;         // scale bytes: even waves stream the A scales, odd waves the W scales (4 k-blocks x 256)
  const uint8_t* s_base = (wave & 1) ? SW : SA;
  int l_ord = 0, l_k = 0, l_buf = 0;
  auto set_load_tile = [&](int ord) {
    const int tile = tile_of((int)blockIdx.x + min(ord, cnt - 1) * G);
    const int mt = tile / ntn, nt = tile - mt * ntn;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = min(mt * 256 + row0 + 64 * i, M - 1);
      rw[i] = min(nt * 256 + row0 + 64 * i, N - 1);
    }
    const int ld = (wave & 1) ? ldw_s : lda_s;
    s_off = (size_t)(lane >> 4) * ld + ((wave & 1) ? nt : mt) * 256 + (lane & 15) * 16;
    s_step = (size_t)4 * ld;
  };
  auto stage = [&]() {
    unsigned char* base = lds + l_buf * STAGE;
    const size_t k0 = (size_t)l_k * 128 + csrc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (size_t)ra[i] * K + k0),
                                       (__attribute__((address_space(3))) void*)(base + dbase + 8192 * i), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + (size_t)rw[i] * K + k0),
                                       (__attribute__((address_space(3))) void*)(base + 32768 + dbase + 8192 * i), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s_base + s_off + (size_t)l_k * s_step),
                                     (__attribute__((address_space(3))) void*)(base + 65536 + (wave & 1) * 1024), 16, 0, 0);
    l_buf ^= 1;
    if (++l_k == nk) { l_k = 0; set_load_tile(++l_ord); }
  };

  f32x16_t acc[4][2];
  u32x4 w0[2][2], w1[2][2], aX[2][2], aY[2][2];
  unsigned sw0 = 0, sw1 = 0, sa0 = 0, sa1 = 0;
  auto frag = [](const u32x4 (&p)[2]) {
    i32x8_t v;
    v[0] = (int)p[0].x; v[1] = (int)p[0].y; v[2] = (int)p[0].z; v[3] = (int)p[0].w;
    v[4] = (int)p[1].x; v[5] = (int)p[1].y; v[6] = (int)p[1].z; v[7] = (int)p[1].w;
    return v;
  };
  // k-step ks of the stage in buffer `buf`: lane (r, h) takes chunks 4 ks + h (k-block 2 ks, bytes 16 h..) and 4 ks + 2 + h
  // (k-block 2 ks + 1) of its rows, and supplies the scale of k-block 2 ks + h
  auto read_w = [&](int buf, int ks, u32x4 (&w)[2][2], unsigned& sw, unsigned& sa) {
    const unsigned char* ba = lds + buf * STAGE;
    const unsigned char* bw = ba + 32768;
    const unsigned char* bs = ba + 65536 + (2 * ks + h) * 256;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wn * 64 + j * 32 + r, sz = (row >> 1) & 7;
      w[j][0] = *reinterpret_cast<const u32x4*>(bw + row * 128 + (((4 * ks + h) ^ sz) << 4));
      w[j][1] = *reinterpret_cast<const u32x4*>(bw + row * 128 + (((4 * ks + 2 + h) ^ sz) << 4));
    }
    sa = *reinterpret_cast<const unsigned*>(bs + wm * 128 + r * 4);
    sw = *reinterpret_cast<const unsigned*>(bs + 1024 + (wn >> 1) * 128 + r * 4) >> ((wn & 1) * 16);
  };
  auto read_a = [&](int buf, int ks, int half, u32x4 (&a)[2][2]) {
    const unsigned char* ba = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm * 128 + (2 * half + i) * 32 + r, sz = (row >> 1) & 7;
      a[i][0] = *reinterpret_cast<const u32x4*>(ba + row * 128 + (((4 * ks + h) ^ sz) << 4));
      a[i][1] = *reinterpret_cast<const u32x4*>(ba + row * 128 + (((4 * ks + 2 + h) ^ sz) << 4));
    }
  };
#define MX_MM(HALF, W, SW, A, SA)                                                                                                   \
  do {                                                                                                                             \
    acc[2 * HALF][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(W[0]), frag(A[0]), acc[2 * HALF][0], 0, 0, 0, (int)SW, 2 * HALF, (int)SA);         \
    acc[2 * HALF][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(W[1]), frag(A[0]), acc[2 * HALF][1], 0, 0, 1, (int)SW, 2 * HALF, (int)SA);         \
    acc[2 * HALF + 1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(W[0]), frag(A[1]), acc[2 * HALF + 1][0], 0, 0, 0, (int)SW, 2 * HALF + 1, (int)SA); \
    acc[2 * HALF + 1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(W[1]), frag(A[1]), acc[2 * HALF + 1][1], 0, 0, 1, (int)SW, 2 * HALF + 1, (int)SA); \
  } while (0)

  // Stage s lives in buffer s & 1.  On stage s: [issue stage s+1 into the other buffer] [16 MFMAs of stage s, fragments refilled a
  // quarter-stage ahead] - before the last quarter: [this wave's loads of stage s+1 landed and its reads of stage s returned]
  // [barrier] [first fragments of stage s+1].  After the barrier every wave's reads of buffer s & 1 are complete, so the next
  // iteration's loads may overwrite it.
  set_load_tile(0);
  stage();
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  read_w(0, 0, w0, sw0, sa0);
  read_a(0, 0, 0, aX);
  int cur = 0;
  for (int ord = 0; ord < cnt; ++ord) {
    const int tile = tile_of((int)blockIdx.x + ord * G);
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int m0 = mt * 256, n0 = nt * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    for (int t = 0; t < nk; ++t) {
      stage();
      read_a(cur, 0, 1, aY);
      MX_MM(0, w0, sw0, aX, sa0);
      read_w(cur, 1, w1, sw1, sa1);
      read_a(cur, 1, 0, aX);
      MX_MM(1, w0, sw0, aY, sa0);
      read_a(cur, 1, 1, aY);
      MX_MM(0, w1, sw1, aX, sa1);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      cur ^= 1;
      read_w(cur, 0, w0, sw0, sa0);
      read_a(cur, 0, 0, aX);
      MX_MM(1, w1, sw1, aY, sa1);
    }
#undef MX_MM

    // epilogue: acc[i][j][v] = C[m = m0 + wm*128 + i*32 + r][n = n0 + wn*64 + j*32 + (v&3) + 8*(v>>2) + 4*h]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nf = n0 + wn * 64 + j * 32;  // N % 256 == 0: always in range
      float4 bq[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) bq[qd] = EPI != GE_PLAIN ? *reinterpret_cast<const float4*>(bias + nf + 8 * qd + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + r;
        float x[16];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          x[4 * qd] = acc[i][j][4 * qd] + bq[qd].x; x[4 * qd + 1] = acc[i][j][4 * qd + 1] + bq[qd].y;
          x[4 * qd + 2] = acc[i][j][4 * qd + 2] + bq[qd].z; x[4 * qd + 3] = acc[i][j][4 * qd + 3] + bq[qd].w;
        }
        if (EPI == GE_RELU) {
#pragma unroll
          for (int v = 0; v < 16; ++v) x[v] = fmaxf(x[v], 0.f);
        }
        if (OUT == MX_OUT_MX) {  // the row's 32 columns of this fragment are one MX block: 16 here, 16 in the other half-wave
          float am = 0.f;
#pragma unroll
          for (int v = 0; v < 16; ++v) am = fmaxf(am, fabsf(x[v]));
          am = fmaxf(am, xor32_f(am));
          float inv;
          const uint32_t sb = mx_block_scale(am, inv);
          if (m < M) {
            uint8_t* cp = reinterpret_cast<uint8_t*>(Cv) + (size_t)m * N + nf + 4 * h;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
              *reinterpret_cast<uint32_t*>(cp + 8 * qd) = mx_pack4(x[4 * qd], x[4 * qd + 1], x[4 * qd + 2], x[4 * qd + 3], inv);
            if (h == 0) SC[(size_t)(nf >> 5) * ldc_s + mx_spos(m)] = (uint8_t)sb;
          }
        } else if (m < M) {
          if (OUT == MX_OUT_F32) {
            float* cp = reinterpret_cast<float*>(Cv) + (size_t)m * N + nf + 4 * h;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              float4* p4 = reinterpret_cast<float4*>(cp + 8 * qd);
              if (EPI == GE_RESID) {
                const float4 o = *p4;
                *p4 = make_float4(o.x + x[4 * qd], o.y + x[4 * qd + 1], o.z + x[4 * qd + 2], o.w + x[4 * qd + 3]);
              } else {
                *p4 = make_float4(x[4 * qd], x[4 * qd + 1], x[4 * qd + 2], x[4 * qd + 3]);
              }
            }
          } else {
            bf16* cp = reinterpret_cast<bf16*>(Cv) + (size_t)m * N + nf + 4 * h;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              union { bf16 e[4]; uint2 u; } pk;
#pragma unroll
              for (int v = 0; v < 4; ++v) pk.e[v] = (bf16)x[4 * qd + v];
              *reinterpret_cast<uint2*>(cp + 8 * qd) = pk.u;
              if (vt != nullptr && nf >= vt_n0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) vt[(size_t)(nf + 8 * qd + 4 * h + v - vt_n0) * vt_ld + m] = pk.e[v];
              }
            }
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stages issued past the last tile
}

// C = A . W^T on MXFP8 operands.  M any (rows clamped, scale arrays padded to 256 rows), N % 256 == 0, K % 128 == 0.
static inline int mx_gemm_dispatch(const uint8_t* A, const uint8_t* SA, int lda_s, const uint8_t* W, const uint8_t* SW, int ldw_s,
                                   const float* bias, void* C, uint8_t* SC, int ldc_s, int M, int N, int K, int epi, int out,
                                   hipStream_t s, bf16* vt = nullptr, int vt_n0 = 0, int vt_ld = 0) {
  if (N % 256 != 0 || K % 128 != 0 || M < 1) return 1;
  const int ntn = N / 256, ntm = (M + 255) / 256;
  static const int ncu = [] {
    int dev = 0, cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
    return cu > 0 ? cu : 256;
  }();
  const int grid = ntn * ntm < ncu ? ntn * ntm : ncu;
  // default: the 64-byte-row ring kernel (mx256_kernel); VX_MX_ALG=0 selects the full-line kernel (mx256w_kernel).  A/B at
  // 117 k rows (profiles/r02_notes.md): 1335 vs 1350 TF/s - full 128-byte lines per LDS-DMA instruction buy nothing here
  static const int alg = [] { const char* v = getenv("VX_MX_ALG"); return v ? atoi(v) : 1; }();
#define MX(E, O)                                                                                                         \
  do {                                                                                                                  \
    static bool attr_done = false;                                                                                      \
    if (!attr_done) {                                                                                                   \
      (void)hipFuncSetAttribute((const void*)mx256_kernel<E, O>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (32768 + 1024));  \
      (void)hipFuncSetAttribute((const void*)mx256w_kernel<E, O>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (65536 + 2048)); \
      attr_done = true;                                                                                                 \
    }                                                                                                                   \
    if (alg == 1)                                                                                                       \
      mx256_kernel<E, O><<<grid, 512, 4 * (32768 + 1024), s>>>(A, SA, lda_s, W, SW, ldw_s, bias, C, SC, ldc_s, M, N, K, vt, vt_n0, vt_ld, ntn, ntn * ntm);  \
    else                                                                                                                \
      mx256w_kernel<E, O><<<grid, 512, 2 * (65536 + 2048), s>>>(A, SA, lda_s, W, SW, ldw_s, bias, C, SC, ldc_s, M, N, K, vt, vt_n0, vt_ld, ntn, ntn * ntm); \
  } while (0)
  if (out == MX_OUT_MX && epi == GE_RELU) MX(GE_RELU, MX_OUT_MX);
  else if (out == MX_OUT_BF16 && epi == GE_BIAS) MX(GE_BIAS, MX_OUT_BF16);
  else if (out == MX_OUT_F32 && epi == GE_RESID) MX(GE_RESID, MX_OUT_F32);
  else if (out == MX_OUT_F32 && epi == GE_BIAS) MX(GE_BIAS, MX_OUT_F32);
  else if (out == MX_OUT_F32 && epi == GE_RELU) MX(GE_RELU, MX_OUT_F32);
  else if (out == MX_OUT_F32 && epi == GE_PLAIN) MX(GE_PLAIN, MX_OUT_F32);
  else return 2;
#undef MX
  return 0;
}

}  // namespace vx
