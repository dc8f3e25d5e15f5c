// bf16 MFMA kernels for the compute-bound row paths (AR prefill, NAR stages):
//   C[M,N] = A[M,K] . W[N,K]^T with fused bias / ReLU / residual epilogues.
// Both operands are K-contiguous, which is exactly the v_mfma_f32_32x32x16_bf16 fragment shape:
// lane (r = l&31, h = l>>5) holds 8 consecutive k (one 16-byte LDS read) of row r of A and of
// row r of W (cdna_hip_programming.md §3 "A/B operand lane maps").
#pragma once
#include <cstdlib>
#include <type_traits>
#include <unordered_map>

#include "common.hpp"
#include "rows_kernels.hpp"

namespace vx {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// 128x128x64 tile, 4 waves as 2(M) x 2(N), each wave 64x64 = 2x2 MFMA tiles of 32x32.
// LDS: double-buffered [128 rows][64 bf16] images of A and W (128-byte rows), 16-byte chunks
// XOR-swizzled by ((row >> 1) & 7) so the 16 rows a ds_read_b128 lane group touches fall on 16
// distinct 16-byte slots of the 256-byte bank row (T2).  The lane groups of ds_read_b128 are NOT 16 consecutive
// lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...): `row & 7` looks right on paper and measures 2-way
// (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE); tests/probes/lds_conflicts.py checks a swizzle against the groups.  Global->register->LDS staging with
// the next tile's global loads issued before the current tile's MFMAs (T14 split).
// RING = 64 / 32: the same tile and epilogues on the staging scheme of the 256^2 kernel below - operands go global -> LDS with
// global_load_lds_dwordx4 (no staging registers, no ds_write) into a ring of FOUR stages of RING k (128 KB / 64 KB of LDS), the
// fragments of stage s+1 are read while stage s is multiplied (two register sets), one counted-vmcnt wait + raw s_barrier per
// stage.  At M ~ 1k rows a CU holds one workgroup = one wave per SIMD, so nothing but the wave's own instruction stream can overlap
// the LDS phase with the matrix phase: the register-staged loop spends 0.65 us per 64 of k where the MFMAs alone need 0.21.
// RING = 32 keeps two workgroups per CU possible: grids of 288 workgroups (FFN1 and the split-K slices at 1025 rows) otherwise run
// as two rounds on 256 CUs.  Its rows are 64 bytes: chunk position = chunk ^ ((0 - (row >> 2)) & 3) (as in the 256^2 kernel).
template <int EPI, bool OUT_F32, int RING = 0>
__global__ __launch_bounds__(256) void mfma_gemm_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                        const float* __restrict__ bias, void* __restrict__ Cv, int M,
                                                        int N, int K, bf16* __restrict__ vt, int vt_n0, int vt_ld, int ld) {
  // K = the K range this workgroup multiplies, ld = row stride of A and W.  gridDim.z > 1: split K across
  // workgroups, slice z covers columns [z K, (z+1) K) and writes its own (M, N) fp32 slab (GE_PLAIN, OUT_F32): the
  // LayerNorm that follows the GEMM anyway adds bias + slabs in a fixed order (deterministic, no atomics).
  // Which tile: launch slot lin = x + gx (y + gy z) runs on XCD lin % 8.  With gx % 8 == 0 (QKV, FFN1) an N tile's M tiles already
  // share an XCD.  The split-K launches of the N = d GEMMs (gx = 8, gz slices) put an N tile's slices AND all M tiles on one
  // XCD: per 64 of k it then pulls gy gz A tiles + gz W tiles over the fabric (FFN2 at 1025 rows: 20 tiles = 320 KB per XCD and
  // step, 5.9 TB/s chip-wide at the measured 0.43 us per step).  VX_GEMM_XCD_Z (A/B): XCD x takes slice x % gz of the N tiles
  // [(x / gz) gz, +gz) instead - gy A tiles + gz W tiles (11 tiles = 176 KB).  Speed only: a bijection of the grid.
  int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
#ifdef VX_GEMM_XCD_Z
  if (gridDim.z > 1 && gridDim.x == 8) {
    const int gy = gridDim.y, gz = gridDim.z;
    const int lin = blockIdx.x + 8 * (blockIdx.y + gy * blockIdx.z), xcd = lin & 7, idx = lin >> 3;  // idx < gy gz
    bzi = xcd % gz;
    bxi = (xcd / gz) * gz + idx / gy;
    byi = idx % gy;
  }
#endif
  A += (size_t)bzi * K;
  W += (size_t)bzi * K;
  if (OUT_F32) Cv = reinterpret_cast<float*>(Cv) + (size_t)bzi * M * N;
  constexpr int BM = 128, BN = 128, BK = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // layout: [buf][A|W][128 rows * 128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = byi * BM, n0 = bxi * BN;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  // the epilogue's bias values, loaded now: fetched after the K loop they are one more exposed memory round trip
  float bias_r[2] = {0.f, 0.f};
  if (EPI != GE_PLAIN) {
#pragma unroll
    for (int j = 0; j < 2; ++j) bias_r[j] = bias[min(n0 + wn * 64 + j * 32 + r, N - 1)];
  }

  if constexpr (RING != 0) {
    constexpr int RK = RING;              // k per stage
    constexpr int RB = RK * 2;            // bytes per row of a stage
    constexpr int CPR = RB / 16;          // 16-byte chunks per row: 8 / 4
    constexpr int KS = RK / 16;           // MFMA k-steps per stage: 4 / 2
    constexpr int OPB = BM * RB;          // bytes per operand stage: 16 KB / 8 KB
    constexpr int NI = OPB / 4096;        // LDS-DMA instructions per operand stage: 4 / 2
    constexpr int ND = 2 * NI;            // ... per stage
    auto swz = [](int row) { return RB == 128 ? ((row >> 1) & 7) : ((0 - (row >> 2)) & 3); };
    const int nk = K / RK;
    // slot q = tid + 256 i of an operand stage -> row q / CPR = row0 + (256 / CPR) i, position tid % CPR, which holds source chunk
    // position ^ swz(row); swz has period 16 / 16 rows resp. and 256 / CPR is a multiple of it: the same chunk for every i
    const int row0 = tid / CPR;
    const int csrc = ((tid % CPR) ^ swz(row0)) * 8;  // elements
    const bf16 *sA[NI], *sW[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      sA[i] = A + (size_t)min(m0 + row0 + (256 / CPR) * i, M - 1) * ld + csrc;
      sW[i] = W + (size_t)min(n0 + row0 + (256 / CPR) * i, N - 1) * ld + csrc;
    }
    const int dbase = (tid - lane) * 16;  // wave-uniform LDS byte offset of lane 0's slot (the DMA adds lane * 16)
    auto stage = [&](int st) {
      unsigned char* base = lds + (st & 3) * (2 * OPB);
      const int k0 = st * RK;
#pragma unroll
      for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sA[i] + k0),
                                         (__attribute__((address_space(3))) void*)(base + dbase + 4096 * i), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sW[i] + k0),
                                         (__attribute__((address_space(3))) void*)(base + OPB + dbase + 4096 * i), 16, 0, 0);
    };
    bf16x8_t fa0[KS][2], fb0[KS][2], fa1[KS][2], fb1[KS][2];
    auto lread = [&](int st, bf16x8_t (&fa)[KS][2], bf16x8_t (&fb)[KS][2]) {
      const unsigned char* ba = lds + (st & 3) * (2 * OPB);
      const unsigned char* bw = ba + OPB;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wm * 64 + i * 32 + r;
          fa[ks][i] = *reinterpret_cast<const bf16x8_t*>(ba + row * RB + (((ks * 2 + h) ^ swz(row)) << 4));
          const int col = wn * 64 + i * 32 + r;
          fb[ks][i] = *reinterpret_cast<const bf16x8_t*>(bw + col * RB + (((ks * 2 + h) ^ swz(col)) << 4));
        }
    };
    auto mm = [&](const bf16x8_t (&fa)[KS][2], const bf16x8_t (&fb)[KS][2]) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
    };
    // Step on stage st (st + 3 < nk): [issue stage st+3 into the slot stage st-1 left two barriers ago] [read the fragments of
    // stage st+1] [MFMAs of stage st] [counted vmcnt: stage st+2 landed, only stage st+3 may be in flight] [barrier]
    auto steady = [&](int st, bf16x8_t (&fa)[KS][2], bf16x8_t (&fb)[KS][2], bf16x8_t (&na)[KS][2], bf16x8_t (&nb)[KS][2]) {
      stage(st + 3);
      lread(st + 1, na, nb);
      mm(fa, fb);
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // VMEM (LDS-DMA)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
      }
#pragma unroll
      for (int i = 0; i < 4 * KS - ND; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * KS / (4 * KS - ND), 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      if (ND == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    };
    // the last three stages (and short K): nothing, or less, left to issue
    auto tail = [&](int st, bf16x8_t (&fa)[KS][2], bf16x8_t (&fb)[KS][2], bf16x8_t (&na)[KS][2], bf16x8_t (&nb)[KS][2]) {
      const bool more = st + 3 < nk;  // uniform
      if (more) stage(st + 3);
      if (st + 1 < nk) lread(st + 1, na, nb);
      mm(fa, fb);
      if (more && ND == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
      else if (more) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    };
    VX_STAMP(0);
    const int pre = nk < 3 ? nk : 3;
    for (int st = 0; st < pre; ++st) stage(st);
    VX_STAMP(1);
    // stages 0 and 1 landed everywhere (older loads too)
    if (pre == 3 && ND == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else if (pre == 3) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    lread(0, fa0, fb0);
    VX_STAMP(2);
    int st = 0;
    for (; st + 4 < nk; st += 2) {
      steady(st, fa0, fb0, fa1, fb1);
      steady(st + 1, fa1, fb1, fa0, fb0);
    }
    VX_STAMP(3);
    for (; st < nk; st += 2) {
      tail(st, fa0, fb0, fa1, fb1);
      if (st + 1 < nk) tail(st + 1, fa1, fb1, fa0, fb0);
    }
    VX_STAMP(4);
    // the bias values were loaded at kernel entry; with the hand-counted waits above hipcc no longer knows that they have landed
    // and put a vmcnt(0) in front of EVERY guarded store of the fp32 epilogue (64 serialised stores, 7 us): one use here, in the
    // block that dominates the epilogue, settles it
    asm volatile("" : "+v"(bias_r[0]), "+v"(bias_r[1]));
  } else {
  // staging: 1024 16-byte chunks per operand tile, 4 per thread; THREE register sets so that three
  // K tiles of global loads are in flight while one is being multiplied (at M ~ 1k rows there is
  // about one workgroup per CU, so nothing else hides the load latency: one tile ahead ran at
  // ~20 GB/s per CU).  Tile t always lives in set t % 3; vmcnt retires in order, so the wait in
  // front of each LDS store only covers that tile's loads.
  uint4 ra[3][4], rw[3][4];
  auto gload = [&](auto SET, int k0) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + i * 256, row = q >> 3, c = q & 7;
      ra[S][i] = ld16(A + (size_t)min(m0 + row, M - 1) * ld + k0 + c * 8);
      rw[S][i] = ld16(W + (size_t)min(n0 + row, N - 1) * ld + k0 + c * 8);
    }
  };
  auto lstore = [&](auto SET, int buf) {
    constexpr int S = decltype(SET)::value;
    unsigned char* ba = lds + buf * 32768;
    unsigned char* bw = ba + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + i * 256, row = q >> 3, c = q & 7;
      const int off = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
      *reinterpret_cast<uint4*>(ba + off) = ra[S][i];
      *reinterpret_cast<uint4*>(bw + off) = rw[S][i];
    }
  };
  auto compute = [&](int cur) {
    const unsigned char* ba = lds + cur * 32768;
    const unsigned char* bw = ba + 16384;
    // all 16 fragment reads of the K tile first, then the 16 MFMAs behind counted lgkmcnt waits: with one wave per SIMD
    // nothing else hides LDS latency, and read-4 / wait / multiply-4 per 16 of K exposed it four times per tile
    bf16x8_t fa[4][2], fb[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + r;
        fa[ks][i] = *reinterpret_cast<const bf16x8_t*>(ba + row * 128 + (((ks * 2 + h) ^ ((row >> 1) & 7)) << 4));
        const int col = wn * 64 + i * 32 + r;
        fb[ks][i] = *reinterpret_cast<const bf16x8_t*>(bw + col * 128 + (((ks * 2 + h) ^ ((col >> 1) & 7)) << 4));
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  const int nk = K / BK;
  // one pipeline step for tile t (set S = t % 3, next tile's set N = (t + 1) % 3)
  auto step = [&](auto SET, auto NEXT, int t) {
    // set S was freed by the LDS store at the end of step t-1.  UNCONDITIONAL (k clamped): a load under a
    // branch makes hipcc's vmcnt bookkeeping conservative and the wait before the LDS store below
    // degenerates to vmcnt(0), i.e. a one-tile pipeline.
    gload(SET, min((t + 3) * BK, K - BK));
    compute(t & 1);
    if (t + 1 < nk) lstore(NEXT, (t + 1) & 1);
    __syncthreads();
  };
  VX_STAMP(0);
  gload(I0{}, 0);
  gload(I1{}, min(BK, K - BK));
  gload(I2{}, min(2 * BK, K - BK));
  VX_STAMP(1);
  lstore(I0{}, 0);
  __syncthreads();
  VX_STAMP(2);
  int kt = 0;
  for (; kt + 3 <= nk; kt += 3) {
    step(I0{}, I1{}, kt);
    step(I1{}, I2{}, kt + 1);
    step(I2{}, I0{}, kt + 2);
    if (kt == 0) VX_STAMP(3);  // after the first three K tiles
  }
  if (kt < nk) step(I0{}, I1{}, kt);
  if (kt + 1 < nk) step(I1{}, I2{}, kt + 1);
  VX_STAMP(4);
  }

  // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5)
  if constexpr (!OUT_F32) {
    // bf16 outputs (QKV, FFN1) leave through LDS: the accumulator layout has n on the lane and scattered m in its
    // registers, so direct stores are 64 two-byte stores per lane (and fully scattered ones for the transposed V
    // copy).  Staged as a bf16 [128][136] tile, every global store is 16 bytes and contiguous along n (C) or along
    // m (V^T).  N % 128 == 0 and K loop done: the operand buffers are free.
    constexpr int LDT = 136;  // row stride in elements (272 B): column reads for V^T hit distinct banks
    bf16* tile = reinterpret_cast<bf16*>(lds);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nl = wn * 64 + j * 32 + r;
      const float bv = bias_r[j];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int ml = wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
          float x = acc[i][j][v] + bv;
          if (EPI == GE_RELU) x = fmaxf(x, 0.f);
          tile[ml * LDT + nl] = (bf16)x;
        }
    }
    __syncthreads();
    bf16* C = reinterpret_cast<bf16*>(Cv);
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // 128 rows x 16 chunks of 8 elements
      const int id = tid + 256 * q, row = id >> 4, c = id & 15;
      if (m0 + row < M)
        *reinterpret_cast<uint4*>(C + (size_t)(m0 + row) * N + n0 + c * 8) = *reinterpret_cast<const uint4*>(tile + row * LDT + c * 8);
    }
    if (vt != nullptr && n0 >= vt_n0) {  // a tile is 128 wide and vt_n0 a multiple of 128: all of it or none
#pragma unroll
      for (int q = 0; q < 8; ++q) {  // 128 channels x 16 chunks of 8 rows; rows past M are finite padding keys
        const int id = tid + 256 * q, nl = id >> 4, mc = id & 15;
        union { bf16 e[8]; uint4 u; } pk;
#pragma unroll
        for (int k = 0; k < 8; ++k) pk.e[k] = tile[(mc * 8 + k) * LDT + nl];
        *reinterpret_cast<uint4*>(vt + (size_t)(n0 + nl - vt_n0) * vt_ld + m0 + mc * 8) = pk.u;
      }
    }
  } else if (m0 + BM <= M && n0 + BN <= N) {
    // full tile (workgroup-uniform): no per-element guards, so the 64 stores of a lane go out back to back, and the residual
    // form reads all its old values first instead of one dependent load -> add -> store round trip per element
    float* cbase = reinterpret_cast<float*>(Cv) + (size_t)(m0 + wm * 64 + 4 * h) * N + n0 + wn * 64 + r;
    float old[2][2][16];
    if (EPI == GE_RESID) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int v = 0; v < 16; ++v) old[i][j][v] = cbase[(size_t)(i * 32 + (v & 3) + 8 * (v >> 2)) * N + j * 32];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float bv = bias_r[j];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          float x = acc[i][j][v] + bv;
          if (EPI == GE_RELU) x = fmaxf(x, 0.f);
          if (EPI == GE_RESID) x += old[i][j][v];
          cbase[(size_t)(i * 32 + (v & 3) + 8 * (v >> 2)) * N + j * 32] = x;
        }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + r;
      const float bv = bias_r[j];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int m = m0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
          if (m < M && n < N) {
            float x = acc[i][j][v] + bv;
            if (EPI == GE_RELU) x = fmaxf(x, 0.f);
            float* c = reinterpret_cast<float*>(Cv) + (size_t)m * N + n;
            *c = (EPI == GE_RESID) ? (*c + x) : x;
          }
        }
      }
    }
  }
  VX_STAMP(5);
}

// ---- wave-tile GEMM: every wave owns one 64x64 output tile and stages its own A / W tiles through a
// wave-private, double-buffered LDS region, so the K loop has NO workgroup barrier: the four waves of a
// CU drift freely and hide each other's global-load latency (at M ~ 1k rows there is about one
// workgroup per CU, so nothing else would).  SPLITK = 1: the 4 waves of a workgroup take 4 neighbouring
// N tiles of one M tile (the A rows hit L1).  SPLITK = 4: the 4 waves take the 4 quarters of K of ONE
// tile and are summed through LDS in a fixed order (deterministic) — used when N/64 x M/64 tiles alone
// would leave the chip idle (the N = 1024 out-projection / FFN2 / predict GEMMs).
// Block ids are laid out so that all M tiles of one N group share blockIdx % 8, i.e. an XCD and its L2
// (speed only; dispatch placement is not a contract).
template <int EPI, bool OUT_F32, int SPLITK>
__global__ __launch_bounds__(256) void wgemm_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                    const float* __restrict__ bias, void* __restrict__ Cv, int M, int N,
                                                    int K, bf16* __restrict__ vt, int vt_n0, int vt_ld, int ngroups,
                                                    int ngroups_pad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [wave][buf][A 8 KB | W 8 KB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ng = blockIdx.x % ngroups_pad, mt = blockIdx.x / ngroups_pad;
  if (ng >= ngroups) return;
  const int m0 = mt * 64;
  const int n0 = (SPLITK == 1) ? (ng * 4 + wave) * 64 : ng * 64;
  const int kbeg = (SPLITK == 1) ? 0 : wave * (K / SPLITK);
  const int nk = (K / SPLITK) / 64;
  unsigned char* my = lds + wave * 32768;
  const bool active = n0 < N;  // SPLITK == 1: the last group may be partial

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  if (active) {
    uint4 ra[3][8], rw[3][8];  // three K tiles of loads in flight per wave (tile t in set t % 3)
    auto gload = [&](auto SET, int k0) {
      constexpr int S = decltype(SET)::value;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = lane + i * 64, row = q >> 3, c = q & 7;
        ra[S][i] = ld16(A + (size_t)min(m0 + row, M - 1) * K + k0 + c * 8);
        rw[S][i] = ld16(W + (size_t)min(n0 + row, N - 1) * K + k0 + c * 8);
      }
    };
    auto lstore = [&](auto SET, int buf) {
      constexpr int S = decltype(SET)::value;
      unsigned char* ba = my + buf * 16384;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = lane + i * 64, row = q >> 3, c = q & 7;
        const int off = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
        *reinterpret_cast<uint4*>(ba + off) = ra[S][i];
        *reinterpret_cast<uint4*>(ba + 8192 + off) = rw[S][i];
      }
    };
    auto compute = [&](int cur) {
      const unsigned char* ba = my + cur * 16384;
      const unsigned char* bw = ba + 8192;
      // all 16 fragment reads first (one wave per SIMD: a read / multiply chain exposes the LDS latency per k-step)
      bf16x8_t fa[4][2], fb[4][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = i * 32 + r;
          const int off = row * 128 + (((ks * 2 + h) ^ ((row >> 1) & 7)) << 4);
          fa[ks][i] = *reinterpret_cast<const bf16x8_t*>(ba + off);
          fb[ks][i] = *reinterpret_cast<const bf16x8_t*>(bw + off);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    auto step = [&](auto SET, auto NEXT, int t) {  // no workgroup barrier: the LDS region is wave-private
      gload(SET, kbeg + min((t + 3) * 64, (nk - 1) * 64));  // unconditional: keeps the vmcnt waits counted
      compute(t & 1);
      if (t + 1 < nk) lstore(NEXT, (t + 1) & 1);
    };
    gload(I0{}, kbeg);
    gload(I1{}, kbeg + min(64, (nk - 1) * 64));
    gload(I2{}, kbeg + min(128, (nk - 1) * 64));
    lstore(I0{}, 0);
    int kt = 0;
    for (; kt + 3 <= nk; kt += 3) {
      step(I0{}, I1{}, kt);
      step(I1{}, I2{}, kt + 1);
      step(I2{}, I0{}, kt + 2);
    }
    if (kt < nk) step(I0{}, I1{}, kt);
    if (kt + 1 < nk) step(I1{}, I2{}, kt + 1);
  }

  auto emit = [&](int m, int n, float x) {
    if (m < M && n < N) {
      if (EPI != GE_PLAIN) x += bias[n];
      if (EPI == GE_RELU) x = fmaxf(x, 0.f);
      if (OUT_F32) {
        float* c = reinterpret_cast<float*>(Cv) + (size_t)m * N + n;
        *c = (EPI == GE_RESID) ? (*c + x) : x;
      } else {
        reinterpret_cast<bf16*>(Cv)[(size_t)m * N + n] = (bf16)x;
        if (vt != nullptr && n >= vt_n0) vt[(size_t)(n - vt_n0) * vt_ld + m] = (bf16)x;
      }
    }
  };

  if (SPLITK == 1) {
    if (!active) return;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v)
          emit(m0 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h, n0 + j * 32 + r, acc[i][j][v]);
  } else {
    // fixed-order sum of the 4 K-quarters through LDS: slab w = wave w's 64x64 fp32 tile (16 KB)
    __syncthreads();  // every wave is done reading its staging buffers
    float* slab = reinterpret_cast<float*>(lds) + wave * 4096;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v)
          slab[(i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h) * 64 + j * 32 + r] = acc[i][j][v];
    __syncthreads();
    const float* all = reinterpret_cast<const float*>(lds);
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {  // wave w finishes rows [16w, 16w+16), lane = column
      const int ml = wave * 16 + rr;
      const float x = ((all[ml * 64 + lane] + all[4096 + ml * 64 + lane]) + all[8192 + ml * 64 + lane]) + all[12288 + ml * 64 + lane];
      emit(m0 + ml, n0 + lane, x);
    }
  }
}

// ---- 256x256 tile, direct-to-LDS operand staging, 4-stage pipeline: the large-M GEMM (batched NAR, M ~ 33 k) ----
// 8 waves as 2(M) x 4(N), each 128x64 of output on v_mfma_f32_16x16x32_bf16 (8 x 4 fragments, 128
// accumulator registers).  Per 64 of K a CU multiplies 8.4 MFLOP (2048 MFMA cycles per SIMD) and moves 64 KB
// through the texture path: half the bytes per FLOP of the 128^2 tile.
//  * Operands go global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write).  The LDS image is
//    lane-linear, so the bank swizzle is applied to the SOURCE address and again on the read (guide rule 21).
//  * K is consumed in stages of 32 (rows of 64 B), FOUR LDS buffers of 32 KB: while stage t is multiplied, stages
//    t+1..t+3 are in flight.  hipcc would drain LDS-DMA with vmcnt(0) at every __syncthreads(), so the loop uses a
//    counted `s_waitcnt vmcnt(4)` (= the youngest stage stays in flight) + raw s_barrier.
//  * Persistent: one workgroup per CU walks its tiles, and the stage stream runs on across tile boundaries, so the
//    next tile's first stages land while the current tile's accumulators are written out (256 KB per tile: at
//    K = 1024 the output write is a third of the tile's time if nothing overlaps it).
//  * Chunk swizzle for 64-byte rows: slot = chunk ^ ((0 - (row >> 2)) & 3): the 16 rows of a ds_read_b128 lane group
//    ({0-3,12-15,20-27}, ...: rows 0-3 / 12-15 with one chunk, rows 4-11 with the next) land on 16 distinct
//    16-byte slots of the 256-byte bank row; the obvious (row >> 2) & 3 is 2-way on these groups (measured).
//  * The MFMA is issued as W-fragment x A-fragment, so the accumulator has m on the lane and 4 consecutive n in
//    its registers: the epilogue stores 8 / 16 bytes per lane instead of 4 x 2-byte pieces.
typedef float f32x4v_t __attribute__((ext_vector_type(4)));

template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(512) void mfma256_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                      const float* __restrict__ bias, void* __restrict__ Cv, int M, int N,
                                                      int K, bf16* __restrict__ vt, int vt_n0, int vt_ld, int ntn,
                                                      int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [4 stages][A 16 KB | W 16 KB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, g = lane >> 4;
  // Persistent workgroups: workgroup b multiplies tiles v = b, b + G, b + 2G, ... (G = gridDim.x, a multiple of 8
  // unless there are fewer tiles than CUs).  XCD-aware order: virtual blocks with equal v % 8 (one XCD under
  // round-robin placement; speed only) walk a contiguous run of tiles, n fastest, so concurrent neighbours share
  // A rows / W rows in that XCD's L2.
  const int G = gridDim.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  auto tile_of = [&](int v) {
    const int xcd = v & 7, loc = v >> 3;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  };
  const int cnt = (ntiles - (int)blockIdx.x + G - 1) / G;  // tiles of this workgroup (>= 1)

  // one stage = 2048 16-byte slots (A: 1024, W: 1024), 4 per thread; slot q -> row q>>2, position q&3 holds
  // source chunk (q&3) ^ ((-(row>>2))&3)
  const int q0 = tid, q1 = tid + 512;
  const int rowa0 = q0 >> 2, rowa1 = q1 >> 2;
  const int ca0 = ((q0 & 3) ^ ((0 - (rowa0 >> 2)) & 3)) * 8, ca1 = ((q1 & 3) ^ ((0 - (rowa1 >> 2)) & 3)) * 8;
  const int d0 = (q0 - lane) * 16, d1 = (q1 - lane) * 16;  // wave-uniform LDS slot of lane 0
  const int nk = K / 32;

  // ---- load side: ONE continuous stream of stages over all tiles of this workgroup, 3 stages ahead of the
  // multiply side, so the first stages of tile i+1 arrive while tile i finishes and writes its output
  const bf16 *srcA0, *srcA1, *srcW0, *srcW1;
  int l_ord = 0, l_k = 0, l_slot = 0;
  auto set_load_tile = [&](int ord) {
    const int tile = tile_of((int)blockIdx.x + min(ord, cnt - 1) * G);  // past the end: re-read the last tile (never used)
    const int mt = tile / ntn, nt = tile - mt * ntn;
    srcA0 = A + (size_t)min(mt * 256 + rowa0, M - 1) * K + ca0;
    srcA1 = A + (size_t)min(mt * 256 + rowa1, M - 1) * K + ca1;
    srcW0 = W + (size_t)min(nt * 256 + rowa0, N - 1) * K + ca0;
    srcW1 = W + (size_t)min(nt * 256 + rowa1, N - 1) * K + ca1;
  };
  auto stage = [&]() {
    unsigned char* base = lds + (l_slot & 3) * 32768;
    const int k0 = l_k * 32;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA0 + k0),
                                     (__attribute__((address_space(3))) void*)(base + d0), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA1 + k0),
                                     (__attribute__((address_space(3))) void*)(base + d1), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW0 + k0),
                                     (__attribute__((address_space(3))) void*)(base + 16384 + d0), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW1 + k0),
                                     (__attribute__((address_space(3))) void*)(base + 16384 + d1), 16, 0, 0);
    ++l_slot;
    if (++l_k == nk) { l_k = 0; set_load_tile(++l_ord); }
  };

  // fragments of one stage: 4 W + 8 A ds_read_b128 per lane.  Two register sets: the reads of stage t+1 are issued
  // among the MFMAs of stage t, so the LDS phase of one stage overlaps the matrix phase of the previous one.
  f32x4v_t acc[8][4];
  bf16x8_t fbA[4], faA[8], fbB[4], faB[8];
  auto lread = [&](int slot, bf16x8_t (&fb)[4], bf16x8_t (&fa)[8]) {
    const unsigned char* ba = lds + (slot & 3) * 32768;
    const unsigned char* bw = ba + 16384;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wn * 64 + j * 16 + r;
      fb[j] = *reinterpret_cast<const bf16x8_t*>(bw + row * 64 + ((g ^ ((0 - (row >> 2)) & 3)) << 4));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = wm * 128 + i * 16 + r;
      fa[i] = *reinterpret_cast<const bf16x8_t*>(ba + row * 64 + ((g ^ ((0 - (row >> 2)) & 3)) << 4));
    }
  };
  auto mm = [&](const bf16x8_t (&fb)[4], const bf16x8_t (&fa)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
  };
  // issue order inside one half-iteration: the 4 LDS-DMA loads and the 12 fragment reads of the NEXT stage are
  // spread between the 32 MFMAs of the current one (1 memory instruction per 2 MFMAs) instead of being bunched in
  // front of them, so the LDS and matrix pipes run concurrently within a wave too
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // VMEM
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
  };

  // Stage s of the stream lives in ring slot s & 3.  Half-iteration on stage s: [issue stage s+3] [read fragments of
  // stage s+1] [MFMAs of stage s] [vmcnt(4): stage s+2 landed; only stage s+3 may still be in flight] [barrier].
  // Slot (s+3)&3 was last read (as stage s-1) in the half-iteration of stage s-2: two barriers ago.
  // hipcc would drain LDS-DMA with vmcnt(0) at every __syncthreads(), hence the counted wait + raw s_barrier.  The
  // epilogue's stores retire in order ahead of younger loads, so the first wait of the next tile also covers them.
  set_load_tile(0);
  stage();
  stage();
  stage();
  asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");  // stages 0 and 1 landed everywhere
  lread(0, fbA, faA);
  int c_slot = 0;
  for (int ord = 0; ord < cnt; ++ord) {
    const int tile = tile_of((int)blockIdx.x + ord * G);
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int m0 = mt * 256, n0 = nt * 256;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nk; t += 2) {  // nk is even (K % 64 == 0)
      stage();
      lread(c_slot + 1, fbB, faB);
      mm(fbA, faA);
      interleave();
      asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      stage();
      lread(c_slot + 2, fbA, faA);
      mm(fbB, faB);
      interleave();
      asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      c_slot += 2;
    }

    // epilogue: acc[i][j][v] = C[m = m0 + wm*128 + i*16 + r][n = n0 + wn*64 + j*16 + 4g + v]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nb = n0 + wn * 64 + j * 16 + 4 * g;  // N % 256 == 0: always in range
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (EPI != GE_PLAIN) bv = *reinterpret_cast<const float4*>(bias + nb);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = m0 + wm * 128 + i * 16 + r;
        float x[4] = {acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w};
        if (EPI == GE_RELU) {
#pragma unroll
          for (int v = 0; v < 4; ++v) x[v] = fmaxf(x[v], 0.f);
        }
        if (m < M) {
          if (OUT_F32) {
            float4* cp = reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (size_t)m * N + nb);
            if (EPI == GE_RESID) {
              const float4 o = *cp;
              *cp = make_float4(o.x + x[0], o.y + x[1], o.z + x[2], o.w + x[3]);
            } else {
              *cp = make_float4(x[0], x[1], x[2], x[3]);
            }
          } else {
            union { bf16 e[4]; uint2 u; } pk;
#pragma unroll
            for (int v = 0; v < 4; ++v) pk.e[v] = (bf16)x[v];
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(Cv) + (size_t)m * N + nb) = pk.u;
            if (vt != nullptr && nb >= vt_n0) {
#pragma unroll
              for (int v = 0; v < 4; ++v) vt[(size_t)(nb + v - vt_n0) * vt_ld + m] = pk.e[v];
            }
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stages issued past the last tile
}

// ---- the 256x256 GEMM on an 8-phase ping-pong schedule (cdna_hip_programming.md, "The 256^2 8-phase template") ----
// Same tile, wave grid (2 x 4, 128 x 64 per wave), fragment order and epilogue as mfma256_kernel; what changes is WHEN things happen.
//  * K is consumed in tiles of 64 (128-byte rows, chunk position = chunk ^ ((row >> 1) & 7)).  A K-tile lives in one of two 64 KB
//    buffers as FOUR 16 KB half-tiles, cut along the waves' register sub-tiles rather than along the tile: A-h0 / A-h1 hold the
//    first / second 64 rows of BOTH row groups, B-h0 / B-h1 the first / second 32 columns of all four column groups.
//  * A K-tile is four PHASES; phase p multiplies one quadrant of the wave's 128 x 64 output (16 MFMAs):
//        p0: read A-sub0 (8 ds_read_b128), multiply A0 x B0       p1: read B-sub1 (4), A0 x B1
//        p2: read A-sub1 (8), A1 x B0                               p3: read B-sub0 of the NEXT K-tile (4), A1 x B1
//    (SCHED 0, kept for A/B: B-sub0 read with A-sub0 in p0, quadrants A0B0 A0B1 A1B1 A1B0, reads 12 / 4 / 8 / 0)
//    Each phase = [fragment reads + ONE half-tile staged by LDS-DMA (2 loads per lane) + counted vmcnt] barrier [16 MFMAs at raised
//    priority] barrier.  The two row groups (the two waves of every SIMD) run ONE BARRIER APART: while one multiplies, the other
//    reads fragments and issues loads, so the matrix pipe and the LDS pipe of a SIMD are both busy all the time.
//  * The load side is one continuous stream of half-tile "events" in the order A-h0, B-h0, B-h1, A-h1 of K-tile 0, 1, 2, ... running
//    on across output tiles (B-h0, A-h0, B-h1, A-h1 = the order of first use); phase P issues event P + 6.  Event e is first read
//    in phase e - 1 and its slot was last read in phase e - 9, three phases before it is overwritten at e - 6 (the guide's WAR rule
//    for staggered groups asks for two).
//    `s_waitcnt vmcnt(8)` in phase P, after that phase's two loads, leaves events P+3 .. P+6 in flight and retires event P+2: read
//    one phase AFTER the wait that retires it, behind a barrier both groups have passed.  Four half-tiles in flight = four phases
//    (about a microsecond) for a load to land.
//  * Output: full tiles store unguarded, so the number of store instructions per lane is a constant (32) and the first four waits
//    after an epilogue can count them in (`vmcnt(40)`: stores and loads retire in issue order on gfx9) instead of waiting for the
//    256 KB tile to be acknowledged; ragged tiles, V^T tiles and residual tiles take the plain count (= wait for the stores).
//    Bias values are loaded before the K loop (the epilogue would otherwise wait for its bias load and, in order, for every
//    operand load in flight).
// Tail split of the persistent 256^2 GEMM (host side: p8_tail_plan): workspace of partial tiles + the split factor (0: none).
struct P8Tail {
  float* ws;
  int split;
};

// Second launch of a tail-split GEMM: tile a's S partial tiles ([item a S + s][256][256] fp32 in the workspace) summed in item
// order (fixed, so the result does not depend on timing) + the GEMM's epilogue.  Grid (tail tiles, 16), 256 threads: a block
// finishes 16 rows x 256 columns, a thread 4 columns of 4 rows.
template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(256) void p8_tail_reduce_kernel(const float* __restrict__ ws, int S, int tile0, int ntn, int ntiles,
                                                             const float* __restrict__ bias, void* __restrict__ Cv, int M, int N,
                                                             bf16* __restrict__ vt, int vt_n0, int vt_ld) {
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int v = tile0 + (int)blockIdx.x, xcd = v & 7, loc = v >> 3;  // the GEMM's tile order (mfma256p_kernel::tile_of)
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const int mt = tile / ntn, nt = tile - mt * ntn;
  const int c4 = (threadIdx.x & 63) * 4, n = nt * 256 + c4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (EPI != GE_PLAIN) bv = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = (int)blockIdx.y * 16 + it * 4 + ((int)threadIdx.x >> 6), m = mt * 256 + row;
    const float* p = ws + ((size_t)blockIdx.x * S * 256 + row) * 256 + c4;
    float4 t = *reinterpret_cast<const float4*>(p);
    for (int sp = 1; sp < S; ++sp) {
      const float4 u = *reinterpret_cast<const float4*>(p + (size_t)sp * 65536);
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    float x[4] = {t.x + bv.x, t.y + bv.y, t.z + bv.z, t.w + bv.w};
    if (EPI == GE_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
    }
    if (m < M) {
      if (OUT_F32) {
        float4* cp = reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (size_t)m * N + n);
        if (EPI == GE_RESID) {
          const float4 o = *cp;
          *cp = make_float4(o.x + x[0], o.y + x[1], o.z + x[2], o.w + x[3]);
        } else {
          *cp = make_float4(x[0], x[1], x[2], x[3]);
        }
      } else {
        union { bf16 e[4]; uint2 u; } pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk.e[e] = (bf16)x[e];
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(Cv) + (size_t)m * N + n) = pk.u;
        if (vt != nullptr && n >= vt_n0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vt[(size_t)(n + e - vt_n0) * vt_ld + m] = pk.e[e];
        }
      }
    }
  }
}

#ifndef VX_P8_SCHED
#define VX_P8_SCHED 1
#endif
template <int EPI, bool OUT_F32, int SCHED = VX_P8_SCHED, bool TAIL = false>
__global__ __launch_bounds__(512) void mfma256p_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                       const float* __restrict__ bias, void* __restrict__ Cv, int M, int N,
                                                       int K, bf16* __restrict__ vt, int vt_n0, int vt_ld, int ntn,
                                                       int ntiles, P8Tail tl) {
  // [2 buffers][A-h0 | B-h0 | B-h1 | A-h1][16 KB], then 8 x 256 B: each wave's 64 bias values of the current tile
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the stagger barrier below must sit under a scalar branch
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 15, g = lane >> 4;
  const int G = gridDim.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  auto tile_of = [&](int v) {
    const int xcd = v & 7, loc = v >> 3;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  };
  const int nk = K >> 6;  // K % 128 == 0: an even number of K-tiles, so a K-tile's buffer is its index & 1 in every output tile
  // Work items of this workgroup.  Without a tail split: tiles b, b + G, b + 2G, ... over all of K.  With one (tl.split = S in
  // {2, 4, 8}; the host sets it when the last round would leave most CUs idle): R = ntiles / G whole tiles each, then the
  // remaining tiles S ways along K: workgroup b < rem * S multiplies K-tiles [s nk / S, (s + 1) nk / S) of tail tile b / S,
  // s = b % S, into a partial tile of the workspace; a second launch sums the partial tiles and applies the epilogue.
  const int S = TAIL ? tl.split : 0;  // TAIL = false: everything below folds to the plain tile walk
  const int R = S > 1 ? ntiles / G : 0;
  const bool has_tail = S > 1 && (int)blockIdx.x < (ntiles - R * G) * S;
  const int cnt = S > 1 ? R + (has_tail ? 1 : 0) : (ntiles - (int)blockIdx.x + G - 1) / G;
  const int tail_a = S > 1 ? (int)blockIdx.x / S : 0, tail_s = S > 1 ? (int)blockIdx.x - tail_a * S : 0;
  auto item = [&](int ord, int& tile, int& kt0, int& kte) {
    if (S > 1 && ord >= R) {
      tile = tile_of(R * G + tail_a);
      kt0 = tail_s * (nk / S);
      kte = kt0 + nk / S;
    } else {
      tile = tile_of((int)blockIdx.x + ord * G);
      kt0 = 0;
      kte = nk;
    }
  };

  // ---- load side.  A half-tile is 1024 16-byte slots, two per thread: slot tid + 512 i -> local row (tid >> 3) + 64 i, position
  // tid & 7 holds source chunk (tid & 7) ^ ((row >> 1) & 7) (the same chunk for both, 64 >> 1 being a multiple of 8).
  // local row -> tile row: A-h(s): (lr >> 6) * 128 + 64 s + (lr & 63); B-h(s): (lr >> 5) * 64 + 32 s + (lr & 31).
  // Source offsets are rebuilt from two lane constants and scalars at every load (the matrix pipe is the busy one; eight
  // offsets held in registers were eight registers too many next to 128 accumulators and 64 fragment registers).
  const int lrow = tid >> 3;
  const unsigned csrc = (unsigned)(((tid & 7) ^ ((lrow >> 1) & 7)) * 16);
  const unsigned ldst = (unsigned)(wave * 1024);  // scalar; lane l lands at +16 l
  const unsigned tb = (unsigned)((lrow >> 5) * 64 + (lrow & 31));
  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(A);
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(W);
  const unsigned K2 = (unsigned)K * 2u;
  int l_ord = 0, l_kt = 0, l_kend = 0, l_m0 = 0, l_n0 = 0;  // the stream's item (ordinal, first row, first column), K-tile, last K-tile + 1: scalars
  auto set_load_tile = [&](int ord) {
    int tile, kt0;
    item(min(ord, cnt - 1), tile, kt0, l_kend);  // past the end: re-read the last item (never used)
    const int mt = tile / ntn;
    l_m0 = mt * 256;
    l_n0 = (tile - mt * ntn) * 256;
    l_kt = kt0;
  };
  // SCHED 0: kind 0: A-h0, 1: B-h0, 2: B-h1, 3: A-h1; SCHED 1: kind 0: B-h0, 1: A-h0, 2: B-h1, 3: A-h1 of the stream's current
  // K-tile (the order of first use); kind 0 opens the next K-tile
  constexpr int KA0 = SCHED == 0 ? 0 : 1, KB0 = SCHED == 0 ? 1 : 0, KB1 = 2, KA1 = 3;
  auto stage = [&](auto kindc, auto bufc) {
    constexpr int kind = decltype(kindc)::value, buf = decltype(bufc)::value;
    constexpr bool isA = kind == KA0 || kind == KA1;
    constexpr int sub = (kind == KA0 || kind == KB0) ? 0 : 1;
    if (kind == 0) {
      if (++l_kt == l_kend) set_load_tile(++l_ord);
    }
    unsigned char* base = lds + buf * 65536 + kind * 16384 + ldst;
    const unsigned kb = (unsigned)l_kt * 128u + csrc;
    unsigned o0, o1;
    if (isA) {
      const int row = l_m0 + lrow + 64 * sub;
      o0 = (unsigned)min(row, M - 1) * K2 + kb;
      o1 = (unsigned)min(row + 128, M - 1) * K2 + kb;
    } else {
      o0 = ((unsigned)(l_n0 + 32 * sub) + tb) * K2 + kb;  // N % 256 == 0: always in range
      o1 = o0 + 128u * K2;
    }
    const unsigned char* src = isA ? Ab : Wb;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)o0),
                                     (__attribute__((address_space(3))) void*)(base), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)o1),
                                     (__attribute__((address_space(3))) void*)(base + 8192), 16, 0, 0);
  };

  // ---- read side: lane (r, g) reads row r of a 16-row fragment, chunk 4 kk + g, at position chunk ^ ((row >> 1) & 7); fragment
  // rows start at multiples of 16, so the XOR term is (r >> 1) for all of them
  const unsigned pos0 = (unsigned)((g ^ (r >> 1)) << 4);
  const unsigned aoff = (unsigned)((wr * 64 + r) * 128), boff = (unsigned)((wc * 32 + r) * 128);
  f32x4v_t acc[8][4];
  bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];
  auto read_a = [&](auto bufc, auto subc) {
    constexpr int buf = decltype(bufc)::value, sub = decltype(subc)::value;
    const unsigned char* b = lds + buf * 65536 + (sub == 0 ? KA0 : KA1) * 16384 + aoff;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fa[i][kk] = *reinterpret_cast<const bf16x8_t*>(b + i * 2048 + (pos0 ^ (kk * 64)));
  };
  auto read_b = [&](auto bufc, auto subc, bf16x8_t (&fb)[2][2]) {
    constexpr int buf = decltype(bufc)::value, sub = decltype(subc)::value;
    const unsigned char* b = lds + buf * 65536 + (sub == 0 ? KB0 : KB1) * 16384 + boff;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fb[j][kk] = *reinterpret_cast<const bf16x8_t*>(b + j * 2048 + (pos0 ^ (kk * 64)));
  };
  auto mm = [&](auto ic, auto jc, const bf16x8_t (&fb)[2][2]) {  // quadrant (i0.., j0..) += A-sub x B-sub over the K-tile's two k-steps
    constexpr int i0 = decltype(ic)::value, j0 = decltype(jc)::value;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa[i][kk], acc[i0 + i][j0 + j], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  // phase x of an 8-phase iteration: K-tile buffer (x >> 2) & 1, quadrant x & 3, stages event kind (x + 2) & 3 into buffer
  // ((x + 6) >> 2) & 1.  laxc: the counted wait lets 32 store instructions through (the four phases behind a full tile's epilogue).
  auto phase = [&](auto xc, auto laxc) {
    constexpr int x = decltype(xc)::value;
    constexpr bool lax = decltype(laxc)::value;
    constexpr int p = x & 3;
    using CB = std::integral_constant<int, (x >> 2) & 1>;
    using SK = std::integral_constant<int, (x + 2) & 3>;
    using SB = std::integral_constant<int, ((x + 6) >> 2) & 1>;
    using NB = std::integral_constant<int, ((x >> 2) & 1) ^ 1>;
    if (SCHED == 0) {  // fragment reads per phase 12 / 4 / 8 / 0; quadrants A0B0, A0B1, A1B1, A1B0
      if (p == 0) {
        read_b(CB{}, I0{}, fb0);
        read_a(CB{}, I0{});
      } else if (p == 1) {
        read_b(CB{}, I1{}, fb1);
      } else if (p == 2) {
        read_a(CB{}, I1{});
      }
    } else {  // 8 / 4 / 8 / 4: quadrants A0B0, A0B1, A1B0, A1B1, the last phase reads the NEXT K-tile's B-sub0 (B0 is dead by then)
      if (p == 0) read_a(CB{}, I0{});
      else if (p == 1) read_b(CB{}, I1{}, fb1);
      else if (p == 2) read_a(CB{}, I1{});
      else read_b(NB{}, I0{}, fb0);
    }
    stage(SK{}, SB{});
    if (lax) {  // 8 operand loads + 32 stores (+ the bias load)
      if (EPI != GE_PLAIN) asm volatile("s_waitcnt vmcnt(41)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(40)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (p == 0) mm(I0{}, I0{}, fb0);
    else if (p == 1) mm(I0{}, I2{}, fb1);
    else if (p == 2) { if (SCHED == 0) mm(I4{}, I2{}, fb1); else mm(I4{}, I0{}, fb0); }
    else { if (SCHED == 0) mm(I4{}, I0{}, fb0); else mm(I4{}, I2{}, fb1); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;
  using P4 = std::integral_constant<int, 4>;
  using P5 = std::integral_constant<int, 5>;
  using P6 = std::integral_constant<int, 6>;
  using P7 = std::integral_constant<int, 7>;

  // prologue: events 0..5 (K-tile 0 whole, A-h0 / B-h0 of K-tile 1); events 0 and 1 landed everywhere behind the barrier
  set_load_tile(0);
  --l_kt;
  stage(I0{}, I0{});
  stage(I1{}, I0{});
  stage(I2{}, I0{});
  stage(I3{}, I0{});
  stage(I0{}, I1{});
  stage(I1{}, I1{});
  asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  if (SCHED == 1) read_b(I0{}, I0{}, fb0);  // event 0: B-sub0 of the first K-tile (every later one is read in the phase before its K-tile)
  if (wr == 1) asm volatile("s_barrier" ::: "memory");  // the second row group runs one barrier behind the first from here on
  __builtin_amdgcn_sched_barrier(0);

  int lax = 0;  // scalar flag: the previous tile left exactly 32 store instructions per lane behind its operand loads
  unsigned char* blds = lds + 131072 + wave * 256;
  for (int ord = 0; ord < cnt; ++ord) {
    int tile, kt0, kte;
    item(ord, tile, kt0, kte);
    const bool is_tail = S > 1 && ord >= R;
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int m0 = mt * 256, n0 = nt * 256;
    // the wave's 64 bias values go to its own LDS line by LDS-DMA: older than every operand load of this tile's K loop, so the
    // counted waits retire it long before the epilogue reads it (a register load would make hipcc drain everything in flight there)
    if (EPI != GE_PLAIN) {
      unsigned l4 = (unsigned)lane * 4u;
      asm volatile("" : "+v"(l4));  // rebuilt per tile: hoisted out of the tile loop the 64-bit address was spilled, and its reload drained vmcnt
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const unsigned char*>(bias + n0 + wc * 64) + l4),
                                       (__attribute__((address_space(3))) void*)(blds), 4, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
    if (lax) {
      phase(P0{}, std::true_type{});
      phase(P1{}, std::true_type{});
      phase(P2{}, std::true_type{});
      phase(P3{}, std::true_type{});
    } else {
      phase(P0{}, std::false_type{});
      phase(P1{}, std::false_type{});
      phase(P2{}, std::false_type{});
      phase(P3{}, std::false_type{});
    }
    phase(P4{}, std::false_type{});
    phase(P5{}, std::false_type{});
    phase(P6{}, std::false_type{});
    phase(P7{}, std::false_type{});
    for (int t = kt0 + 2; t < kte; t += 2) {
      phase(P0{}, std::false_type{});
      phase(P1{}, std::false_type{});
      phase(P2{}, std::false_type{});
      phase(P3{}, std::false_type{});
      phase(P4{}, std::false_type{});
      phase(P5{}, std::false_type{});
      phase(P6{}, std::false_type{});
      phase(P7{}, std::false_type{});
    }

    // epilogue: acc[i][j][v] = C[m = m0 + wr*128 + i*16 + r][n = n0 + wc*64 + j*16 + 4g + v]
    const bool has_vt = !OUT_F32 && vt != nullptr && n0 + 256 > vt_n0;
    const bool full = m0 + 256 <= M;
    unsigned gq = (unsigned)(g * 16);  // lane part of the bias reads, rebuilt per tile (hoisted, one read address per j was spilled: its reload drained vmcnt)
    asm volatile("" : "+v"(gq));
    auto emit = [&](auto guardc) {
      constexpr bool guard = decltype(guardc)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nb = n0 + wc * 64 + j * 16 + 4 * g;
        f32x4v_t bv = f32x4v_t{0.f, 0.f, 0.f, 0.f};
        if (EPI != GE_PLAIN) bv = *reinterpret_cast<const f32x4v_t*>(blds + gq + j * 64);
        // residual form: the eight old values of this column group are requested together (the fragment registers are free
        // now) by asm loads behind ONE wait that hands the values on: written as plain loads hipcc sinks each one next to its add,
        // 32 dependent round trips per lane and tile
        f32x4v_t old[8];
        if (OUT_F32 && EPI == GE_RESID) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int m = m0 + wr * 128 + i * 16 + r;
            const float* op = reinterpret_cast<const float*>(Cv) + (size_t)(guard ? min(m, M - 1) : m) * N + nb;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(old[i]) : "v"(op) : "memory");
          }
          asm volatile("s_waitcnt vmcnt(0)"
                       : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(old[4]), "+v"(old[5]), "+v"(old[6]), "+v"(old[7])
                       :
                       : "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int m = m0 + wr * 128 + i * 16 + r;
          float x[4] = {acc[i][j][0] + bv[0], acc[i][j][1] + bv[1], acc[i][j][2] + bv[2], acc[i][j][3] + bv[3]};
          if (EPI == GE_RELU) {
#pragma unroll
            for (int v = 0; v < 4; ++v) x[v] = fmaxf(x[v], 0.f);
          }
          if (!guard || m < M) {
            if (OUT_F32) {
              float4* cp = reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (size_t)m * N + nb);
              if (EPI == GE_RESID) {
                const f32x4v_t o = old[i];
                *cp = make_float4(o[0] + x[0], o[1] + x[1], o[2] + x[2], o[3] + x[3]);
              } else {
                *cp = make_float4(x[0], x[1], x[2], x[3]);
              }
            } else {
              union { bf16 e[4]; uint2 u; } pk;
#pragma unroll
              for (int v = 0; v < 4; ++v) pk.e[v] = (bf16)x[v];
              *reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(Cv) + (size_t)m * N + nb) = pk.u;
              if (vt != nullptr && nb >= vt_n0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) vt[(size_t)(nb + v - vt_n0) * vt_ld + m] = pk.e[v];
              }
            }
          }
        }
      }
    };
    if (is_tail) {
      // tail item: the partial tile goes to the workspace as it is ([item][256][256] fp32, same 16-byte pieces as the fp32
      // epilogue); p8_tail_reduce_kernel, launched behind this kernel, adds the S partial tiles of a tile in a fixed order and
      // applies the epilogue
      unsigned lo = (unsigned)((r * 256 + 4 * g) * 4);  // lane part of the address, rebuilt here: hoisted, the 32 store addresses were spilled
      asm volatile("" : "+v"(lo));
      unsigned char* slab = reinterpret_cast<unsigned char*>(tl.ws + (size_t)(tail_a * S + tail_s) * 65536 + wr * 32768 + wc * 64) + lo;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4v_t*>(slab + i * 16384 + j * 64) = acc[i][j];
      lax = 0;
    } else if (full) {
      emit(std::false_type{});
      // exactly 32 store instructions per lane behind the operand loads in flight; V^T and residual tiles wait for their stores
      lax = __builtin_amdgcn_readfirstlane((has_vt || EPI == GE_RESID) ? 0 : 1);
    } else {
      emit(std::true_type{});
      lax = 0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the events issued past the last tile
  if (wr == 0) asm volatile("s_barrier" ::: "memory");  // every wave has passed the same number of barriers
}

// Main loop of the 128^2 kernel for a grid of `wgs` workgroups: the LDS-DMA ring with 64-k stages (128 KB of LDS, one workgroup
// per CU) when the grid fits one round, the 32-k ring (64 KB, two per CU) otherwise.  VX_GEMM_RING = 0 / 32 / 64 forces the
// register-staged loop / one ring for A/B runs.
static inline int gemm_ring128(long long wgs) {
  static const int forced = [] { const char* v = getenv("VX_GEMM_RING"); return (v && *v >= '0' && *v <= '9') ? atoi(v) : -1; }();
  if (forced == 0 || forced == 32 || forced == 64) return forced;
  const int ncu = vx_cu_count();
  return wgs <= ncu ? 64 : 32;
}

// Tail split of the persistent 8-phase GEMM (VX_GEMM_TAIL=1; OFF by default, see below): when the last round of 256^2 tiles would
// occupy few of the workgroups (544 tiles on 256 CUs: two full rounds, then 32 tiles at the price of a third), those tiles are split
// S ways along K, every workgroup gets one more, short item that leaves a partial tile in a workspace, and a second, small launch
// sums the partial tiles in a fixed order and applies the epilogue.  S = the largest of 8 / 4 / 2 with items <= workgroups and >= 8
// K-tiles per item, and only if the estimate (K loop / S + partial-tile traffic + one launch) is well under the full round it
// replaces: true at K = 4096, not at K = 1024.  One workspace per stream (<= 64 MB), allocated at first use.
// Measured (profiles/r02_notes.md): 34816 x 1024 x 4096 322-332 -> 303-308 us, 134144 x 1024 x 4096 943-950 -> 922-925 us: the 128 MB
// of partial tiles written and read back eat most of the idle round.  Why it is off: a split tile's fp32 sum is S chunk sums added
// afterwards, a whole tile's is one running sum - deterministic, but a row's rounding then depends on WHICH tile it falls in, and
// the batched NAR pass loses the property that equal utterances give equal codes whatever their slot (tests/test_gpu_batch.py).
static inline P8Tail p8_tail_plan(int ntiles, int grid, int K, hipStream_t s) {
  static const bool on = [] { const char* v = getenv("VX_GEMM_TAIL"); return v && atoi(v) != 0; }();
  P8Tail tl{nullptr, 0};
  if (!on || ntiles <= grid || grid > 256) return tl;
  const int rem = ntiles % grid, nk = K / 64;
  if (rem == 0 || rem * 2 > grid) return tl;
  int S = 0;
  for (int c = 8; c >= 2; c >>= 1)
    if (rem * c <= grid && nk % (2 * c) == 0 && nk / c >= 8) { S = c; break; }
  if (S == 0) return tl;
  // microseconds: 1.5 per K-tile of a 256^2 tile; a partial tile written and read back at ~4 TB/s chip-wide; ~7 for the store
  // phase and the second launch
  const float full = 1.5f * nk, split = 1.5f * nk / S + (float)rem * S * 0.262144f * 2.f / 4.f + 7.f;
  if (split > 0.7f * full) return tl;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return tl;  // no allocation inside a capture
  static std::unordered_map<hipStream_t, float*> wss;
  float*& w = wss[s];
  if (w == nullptr && hipMalloc((void**)&w, (size_t)256 * 262144) != hipSuccess) { w = nullptr; return tl; }
  tl.ws = w;
  tl.split = S;
  return tl;
}

static inline int mfma_gemm_dispatch(const bf16* A, const bf16* W, const float* bias, void* C, int M, int N, int K,
                                     int epi, bool out_f32, hipStream_t s, bf16* vt = nullptr, int vt_n0 = 0,
                                     int vt_ld = 0) {
  if (K % 64 != 0 || N % 64 != 0) {
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
  // alg 0 (default): 128x128 shared tiles when there are enough of them, wave tiles + in-kernel split-K otherwise;
  // 1 / 2 force one or the other (A/B runs)
  static const int alg = [] { const char* v = getenv("VX_GEMM_ALG"); return v ? atoi(v) : 0; }();
  const long long tiles64 = (long long)((M + 63) / 64) * (N / 64);
  if ((alg == 3 || (alg == 0 && M >= 4096)) && N % 256 == 0 && K >= 128) {  // K % 64 == 0 checked above  // enough 256^2 tiles for several per CU
    const int ntn = N / 256, ntm = (M + 255) / 256;
    const int ncu = vx_cu_count();
    const int grid256 = ntn * ntm < ncu ? ntn * ntm : ncu;  // one persistent workgroup per CU (128 KB of LDS each)
    // the 8-phase schedule needs whole pairs of 64-k tiles and 32-bit byte offsets into A and W; any other K or size runs the
    // 32-k ring (mfma256_kernel)
    const bool p8 = K % 128 == 0 && (size_t)M * K * 2 < 0xFFFF0000ull && (size_t)N * K * 2 < 0xFFFF0000ull;
    const P8Tail tl = p8 ? p8_tail_plan(ntn * ntm, grid256, K, s) : P8Tail{nullptr, 0};  // all forms; in this model only the K = 4096 one (FFN2) ever splits
#define M2(E, F)                                                                                                         \
  do {                                                                                                                  \
    static bool attr_dev[16] = {}; bool& attr_done = attr_dev[vx_cur_device()];                                                                                      \
    if (!attr_done) {                                                                                                   \
      (void)hipFuncSetAttribute((const void*)mfma256_kernel<E, F>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);  \
      (void)hipFuncSetAttribute((const void*)mfma256p_kernel<E, F>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 2048); \
      (void)hipFuncSetAttribute((const void*)mfma256p_kernel<E, F, VX_P8_SCHED, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 2048); \
      attr_done = true;                                                                                                 \
    }                                                                                                                   \
    if (p8 && tl.split > 1) {                                                                                     \
      const int nt_all = ntn * ntm, rem_t = nt_all % grid256;                                                           \
      mfma256p_kernel<E, F, VX_P8_SCHED, true><<<grid256, 512, 131072 + 2048, s>>>(A, W, bias, C, M, N, K, vt, vt_n0, vt_ld, ntn, nt_all, tl); \
      p8_tail_reduce_kernel<E, F><<<dim3(rem_t, 16), 256, 0, s>>>(tl.ws, tl.split, nt_all - rem_t, ntn, nt_all, bias, C, M, N, vt, vt_n0, vt_ld); \
    } else if (p8) {                                                                                                     \
      mfma256p_kernel<E, F><<<grid256, 512, 131072 + 2048, s>>>(A, W, bias, C, M, N, K, vt, vt_n0, vt_ld, ntn, ntn * ntm, tl); \
    } else {                                                                                                             \
      mfma256_kernel<E, F><<<grid256, 512, 131072, s>>>(A, W, bias, C, M, N, K, vt, vt_n0, vt_ld, ntn, ntn * ntm);        \
    }                                                                                                                   \
  } while (0)
    if (N % 256 == 0) {
      if (epi == GE_RESID) M2(GE_RESID, true);
      else if (epi == GE_PLAIN) M2(GE_PLAIN, true);
      else if (epi == GE_BIAS && out_f32) M2(GE_BIAS, true);
      else if (epi == GE_RELU && out_f32) M2(GE_RELU, true);
      else if (epi == GE_BIAS) M2(GE_BIAS, false);
      else M2(GE_RELU, false);
      return 0;
    }
#undef M2
  }
  if (alg == 1 || (alg == 0 && N % 128 == 0 && tiles64 >= 512)) {
    dim3 grid((N + 127) / 128, (M + 127) / 128);
#define MG(E, F)                                                                                                        \
  do {                                                                                                                  \
    static bool attr_dev[16] = {}; bool& attr_done = attr_dev[vx_cur_device()];                                                                                      \
    if (!attr_done) {                                                                                                   \
      (void)hipFuncSetAttribute((const void*)mfma_gemm_kernel<E, F>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
      (void)hipFuncSetAttribute((const void*)mfma_gemm_kernel<E, F, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
      (void)hipFuncSetAttribute((const void*)mfma_gemm_kernel<E, F, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
      attr_done = true;                                                                                                 \
    }                                                                                                                   \
    const int ring = gemm_ring128((long long)grid.x * grid.y);                                                          \
    if (ring == 64) mfma_gemm_kernel<E, F, 64><<<grid, 256, 131072, s>>>(A, W, bias, C, M, N, K, vt, vt_n0, vt_ld, K);   \
    else if (ring == 32) mfma_gemm_kernel<E, F, 32><<<grid, 256, 65536, s>>>(A, W, bias, C, M, N, K, vt, vt_n0, vt_ld, K); \
    else mfma_gemm_kernel<E, F><<<grid, 256, 65536, s>>>(A, W, bias, C, M, N, K, vt, vt_n0, vt_ld, K);                  \
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
  const int mtiles = (M + 63) / 64;
  const size_t lds = 131072;
#define WG(E, F, S)                                                                                                     \
  do {                                                                                                                  \
    static bool attr_dev[16] = {}; bool& attr_done = attr_dev[vx_cur_device()];                                                                                      \
    if (!attr_done) {                                                                                                   \
      (void)hipFuncSetAttribute((const void*)wgemm_kernel<E, F, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_done = true;                                                                                                 \
    }                                                                                                                   \
    const int ngr = (S == 1) ? (N + 255) / 256 : N / 64;                                                                \
    const int ngp = (ngr + 7) / 8 * 8;                                                                                  \
    wgemm_kernel<E, F, S><<<mtiles * ngp, 256, lds, s>>>(A, W, bias, C, M, N, K, vt, vt_n0, vt_ld, ngr, ngp);             \
  } while (0)
  // split K inside the workgroup when the tile grid alone cannot fill 4 waves on each of the 256 CUs
  const bool split = (K % 256 == 0) && tiles64 < 512;
  if (split) {
    if (epi == GE_RESID) WG(GE_RESID, true, 4);
    else if (epi == GE_PLAIN) WG(GE_PLAIN, true, 4);
    else if (epi == GE_BIAS && out_f32) WG(GE_BIAS, true, 4);
    else if (epi == GE_RELU && out_f32) WG(GE_RELU, true, 4);
    else if (epi == GE_BIAS) WG(GE_BIAS, false, 4);
    else WG(GE_RELU, false, 4);
  } else {
    if (epi == GE_RESID) WG(GE_RESID, true, 1);
    else if (epi == GE_PLAIN) WG(GE_PLAIN, true, 1);
    else if (epi == GE_BIAS && out_f32) WG(GE_BIAS, true, 1);
    else if (epi == GE_RELU && out_f32) WG(GE_RELU, true, 1);
    else if (epi == GE_BIAS) WG(GE_BIAS, false, 1);
    else WG(GE_RELU, false, 1);
  }
#undef WG
  return 0;
}

// C_z = A[:, zK/s:(z+1)K/s] . W[:, same]^T for z < splits, each into its own fp32 (M, N) slab: the N = d GEMMs
// (out-projection, FFN2) at M ~ 1k rows, where 128^2 tiles alone give 72 workgroups.  splits in {1,2,4}.
static inline int mfma_gemm_partial(const bf16* A, const bf16* W, float* slabs, int M, int N, int K, int splits, hipStream_t s) {
  static bool attr_dev[16] = {}; bool& attr_done = attr_dev[vx_cur_device()];
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)mfma_gemm_kernel<GE_PLAIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)mfma_gemm_kernel<GE_PLAIN, true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)mfma_gemm_kernel<GE_PLAIN, true, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr_done = true;
  }
  dim3 grid(N / 128, (M + 127) / 128, splits);
  const int ring = gemm_ring128((long long)grid.x * grid.y * grid.z);
  if (ring == 64) mfma_gemm_kernel<GE_PLAIN, true, 64><<<grid, 256, 131072, s>>>(A, W, nullptr, slabs, M, N, K / splits, nullptr, 0, 0, K);
  else if (ring == 32) mfma_gemm_kernel<GE_PLAIN, true, 32><<<grid, 256, 65536, s>>>(A, W, nullptr, slabs, M, N, K / splits, nullptr, 0, 0, K);
  else mfma_gemm_kernel<GE_PLAIN, true><<<grid, 256, 65536, s>>>(A, W, nullptr, slabs, M, N, K / splits, nullptr, 0, 0, K);
  return 0;
}

// ---- flash attention over rows (NAR stages: no mask; AR prefill: the reference's prefix mask) ----
// seg_start != nullptr: the rows are a concatenation of utterances (batched NAR / batched prefill), blockIdx.z
// picks the segment; seg_text (optional) holds each segment's text length for the prefix mask.
// One wave = 32 queries, workgroup = NW waves sharing 64-key K / V^T tiles in LDS (double-buffered).
// Orientation (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand"):
//   S^T[key][query] = K . Q^T     A = K rows from LDS, B = Q rows (registers, pre-scaled by 1/8, exact)
//   O^T[dim][query] = V^T . P^T   A = V^T rows from LDS, B = the S^T accumulator itself, cast to bf16
// so the query index stays on the LANE in both products: the online-softmax statistics and the
// rescale of O are per-lane scalars, the key reduction is over the 16 accumulator registers plus one
// half-wave exchange, and P never leaves registers.
__device__ __forceinline__ float xor32_f(float v) {
  // v_permlane32_swap: lanes 32-63 of the 1st operand swap with lanes 0-31 of the 2nd
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

// KG key groups: with KG = 2 the workgroup has 2 NW waves; wave (qw, kg) multiplies the 32 queries of slice qw
// against key tiles kg, kg + 2, ... and the two partial softmaxes of a slice are merged through LDS at the end
// (flash-decoding inside the workgroup).  Used when the (query block, head) grid alone gives each CU about one
// workgroup (batch-1 NAR: 272 workgroups of 2 waves left two of the four SIMDs of every CU idle).
template <int NW, int KG>
__global__ __launch_bounds__(NW * KG * 64, (KG == 2 ? 2 : 1)) void mfma_attn_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ vt,
                                                                 bf16* __restrict__ out, int M, int vt_ld, int d,
                                                                 int text_len, const int* __restrict__ seg_start,
                                                                 const int* __restrict__ seg_len,
                                                                 const int* __restrict__ seg_text) {
  constexpr int HD = 64, NT = NW * KG * 64;
  // XCD-aware placement: workgroups are handed to the 8 XCDs round-robin in launch order, so the query blocks of one
  // (head, segment) - which all stream the SAME K / V^T rows - would land on 8 different L2s and each fetch them over the fabric
  // (32 x 1088 rows: 1.28 GB per layer-stage for 142 MB of distinct K / V; 6 TB/s of fabric reads at the measured 211 us).
  // Launch slot `lin` therefore works on logical block (lin % 8) * (total / 8) + lin / 8: every XCD gets a CONTIGUOUS run of
  // logical blocks (query block fastest), i.e. all query blocks of a (head, segment) share one L2 and run back to back.
  // Speed only - any mapping is a bijection of the grid.
  int bx, by, bz;
  {
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned n8 = total >> 3;
    const unsigned logical = lin < (n8 << 3) ? (lin & 7u) * n8 + (lin >> 3) : lin;
    bx = (int)(logical % gx);
    by = (int)((logical / gx) % gy);
    bz = (int)(logical / (gx * gy));
  }
  // batched NAR / prefill: segment z of a concatenated row buffer (starts are multiples of 64 rows, so the 16-byte
  // K / V^T tile loads stay aligned); single sequence: seg_start == nullptr
  if (seg_start != nullptr) {
    const int r0 = seg_start[bz];
    M = seg_len[bz];
    if (seg_text != nullptr) text_len = seg_text[bz];  // batched prefill: every utterance has its own text length
    if (bx * 32 * NW >= M) return;
    qkv += (size_t)r0 * 3 * d;
    out += (size_t)r0 * d;
    vt += r0;
  }
  constexpr int CPT = 512 * KG / NT;  // 16-byte chunks per thread per operand per iteration (KG tiles of 64 rows x 8 chunks)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];  // [buf][key group][K | V^T][64 rows * 128 B]
  auto ldsp = [&](int buf, int g, int kv) { return lds_raw + (size_t)(((buf * KG + g) * 2 + kv)) * 8192; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qw = wave % NW, kg = wave / NW;
  const int r = lane & 31, hh = lane >> 5;
  const int head = by;
  const int q0 = bx * (32 * NW) + qw * 32;
  const int ld3 = 3 * d;
  const int qrow = q0 + r;
  const bool qvalid = qrow < M;

  // Q fragments for the 4 k-steps of 16 dims (scaled by 1/sqrt(64) = 2^-3, exact in bf16, once the first tile is requested: the
  // scaling in place made the compiler wait for Q before it issued three of the first tile's four loads)
  bf16x8_t qf[4];
  {
    const bf16* qp = qkv + (size_t)min(qrow, M - 1) * ld3 + head * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16);
  }
  // keys visible to this lane's query: [0, limit)
  const int limit = !qvalid ? 0 : (text_len < 0 ? M : (qrow < text_len ? text_len : qrow + 1));
  int blk_limit = M;  // highest key any query of the workgroup may see
  if (text_len >= 0) {
    const int last = min(M, (bx + 1) * 32 * NW) - 1;
    blk_limit = last < text_len ? text_len : last + 1;
  }
  const int ntiles = (blk_limit + 63) / 64;
  const int niter = (ntiles + KG - 1) / KG;  // iteration `it` stages tiles KG*it .. KG*it + KG-1 (past the end: masked)

  // KG == 2 (batch-1 NAR / prefill: about one workgroup per CU, one wave per SIMD): TWO register sets - the tiles of iteration
  // it+2 are requested while iteration it is multiplied and stored to LDS at the end of iteration it+1, so a load has two
  // iterations to land.  With one set an iteration took 1.56 us of which ~0.4 us is arithmetic (tests/probes/attn_stamps.py): the
  // rest was the wait for loads issued one short iteration earlier.  The launch bound keeps the kernel at two workgroups per CU
  // (the 272-workgroup grid at 1025 rows must stay one round).  Loads are unconditional on a clamped tile index (a load under a
  // branch makes hipcc's wait in front of the LDS store a vmcnt(0)).
  constexpr bool TWO = KG == 2;
  struct TileRegs { uint4 k[CPT], v[CPT]; };
  TileRegs R0, R1;
  // Tile loads: uniform base pointers + 32-bit per-thread BYTE offsets that advance by a scalar per tile (the dispatcher checks
  // that both operands stay under 4 GB).  Rows past the sequence are clamped by clamping the offset (it grows with the row), so a
  // load costs an add and a min instead of a 64-bit multiply-add per address (14 -> 6 VALU instructions per tile).
  const char* __restrict__ const kbase = reinterpret_cast<const char*>(qkv + d + head * HD);
  const char* __restrict__ const vbase = reinterpret_cast<const char*>(vt);
  unsigned k_off[CPT], k_max[CPT], v_off[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int q = tid + i * NT, g = q >> 9, row = (q >> 3) & 63, c = q & 7;
    k_off[i] = 2u * ((unsigned)(g * 64 + row) * (unsigned)ld3 + c * 8);  // key row of tile 0 (group g), 8 dims
    k_max[i] = 2u * ((unsigned)(M - 1) * (unsigned)ld3 + c * 8);         // the same chunk of the last row
    v_off[i] = 2u * ((unsigned)(head * HD + row) * (unsigned)vt_ld + c * 8);  // channel row, 8 keys of tile column 0
  }
  const int v_kt_max = (vt_ld - 64) & ~63;
  auto gload = [&](TileRegs& R, int it) {
    const unsigned kstep = 2u * (unsigned)(it * KG * 64) * (unsigned)ld3;  // uniform
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int g = (i * NT) >> 9;  // = (tid + i NT) >> 9
      const unsigned vkt = 2u * (unsigned)min((it * KG + g) * 64, v_kt_max);  // uniform
      const unsigned ko = min(k_off[i] + kstep, k_max[i]), vo = v_off[i] + vkt;
      R.k[i] = ld16(reinterpret_cast<const bf16*>(kbase + ko));
      R.v[i] = ld16(reinterpret_cast<const bf16*>(vbase + vo));
    }
  };
  auto lstore = [&](const TileRegs& R, int buf) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int q = tid + i * NT, g = q >> 9, row = (q >> 3) & 63, c = q & 7;
      const int sw = (row >> 1) & 7;
      *reinterpret_cast<uint4*>(ldsp(buf, g, 0) + row * 128 + ((c ^ sw) << 4)) = R.k[i];
      // V^T: the second product's A fragment of half-wave hh is keys {4hh..4hh+3} and {8+4hh..8+4hh+3} of a 16-key group
      // (the accumulator order of S^T, see below), so the group's two 8-key chunks are regrouped on the way in - chunk 2g holds
      // keys {0-3, 8-11}, chunk 2g+1 keys {4-7, 12-15} - and a fragment is ONE 16-byte read (two 8-byte reads plus three register
      // moves per fragment before: 24 of the loop's ~190 VALU instructions)
      unsigned char* vrow_p = ldsp(buf, g, 1) + row * 128 + 8 * (c & 1);
      *reinterpret_cast<uint2*>(vrow_p + (((c & 6) ^ sw) << 4)) = make_uint2(R.v[i].x, R.v[i].y);
      *reinterpret_cast<uint2*>(vrow_p + (((c | 1) ^ sw) << 4)) = make_uint2(R.v[i].z, R.v[i].w);
    }
  };

  f32x16_t accO[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) accO[t][v] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // l_run: this half-wave's share of the row sum

  VX_STAMP(8);
  gload(R0, 0);
  __builtin_amdgcn_sched_barrier(0);  // all eight loads of the prologue are out before anything waits
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[ks][j] = (bf16)((float)qf[ks][j] * 0.125f);
  lstore(R0, 0);
  if (TWO) gload(R1, min(1, niter - 1));
  __syncthreads();
  VX_STAMP(9);
  // iteration `it`: tile set it+2 -> RL (the set tile it left), multiply LDS buffer it & 1, tile set it+1 (in RS) -> the other buffer
  auto step = [&](auto masked_c, int it, TileRegs& RL, const TileRegs& RS) {
    constexpr bool MASKED = decltype(masked_c)::value;
    const int cur = it & 1, kt = (it * KG + kg) * 64;
#ifdef VX_STAMPS
    if (it < 12) VX_STAMP(10 + it);
    if (it == 3) VX_STAMP(24);  // 24..28: phases of iteration 3 (tests/probes/nar_batch_driver.py prints them)
#endif
    if (TWO) gload(RL, min(it + 2, niter - 1));
    else if (it + 1 < niter) gload(RL, it + 1);
    const unsigned char* kb = ldsp(cur, kg, 0);
    const unsigned char* vb = ldsp(cur, kg, 1);
    // S^T for the two 32-key sub-tiles.  All eight K fragments are requested before the first product (left alone the compiler
    // read every fragment into the same four registers: read, wait, multiply, eight times over)
    const int swz = (r >> 1) & 7;  // rows r and r + 32 swizzle alike
    bf16x8_t kf[2][4];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        kf[sub][ks] = *reinterpret_cast<const bf16x8_t*>(kb + (sub * 32 + r) * 128 + (((ks * 2 + hh) ^ swz) << 4));
    f32x16_t accS[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int v = 0; v < 16; ++v) accS[sub][v] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) accS[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][ks], qf[ks], accS[sub], 0, 0, 0);
    }
    if (!TWO) {
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // 8 LDS reads
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);  // 8 MFMAs
    } else {  // two register sets of tile loads are live here: four fragments ahead instead of eight (eight spilled)
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    // Per-element masking exists only in the MASKED instance of this body (register v of half hh holds key
    // (v&3) + 8(v>>2) + 4hh of the sub-tile): the tile loop below runs the plain instance over the tiles EVERY query of the
    // workgroup sees in full - all but the last tile of an unmasked (NAR) stage, everything left of the diagonal under the prefix
    // mask - and the masked one over the rest.  One body with a wave-uniform choice per tile cost the plain path 16 64-bit
    // register moves per tile (S^T or O^T copied to where the other path keeps it).
    {
      float mloc = -INFINITY;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          if (MASKED) {
            const int kgi = kt + sub * 32 + (v & 3) + 8 * (v >> 2) + 4 * hh;
            accS[sub][v] = (kgi < limit) ? accS[sub][v] : -INFINITY;
          }
          mloc = fmaxf(mloc, accS[sub][v]);
        }
      mloc = fmaxf(mloc, xor32_f(mloc));
#ifdef VX_STAMPS
      if (it == 3) { asm volatile("" : "+v"(mloc)); VX_STAMP(25); }  // S^T there, tile maximum known
#endif
      const float m_new = fmaxf(m_run, mloc);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // fully masked so far: exp(-inf - 0) = 0
      // rescale the running sums only when some query's maximum moved (wave-uniform test; the factor is exactly 1
      // otherwise): after the first few key tiles it rarely does
      if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
        const float corr = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m_use) * 1.4426950408889634f);
        l_run *= corr;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int v = 0; v < 16; ++v) accO[t][v] *= corr;
      }
      m_run = m_new;
      // P^T = exp(S^T - m) = exp2(S^T log2(e) - m log2(e)): one (packed) fma + v_exp_f32 per element; row sums on packed adds in two
      // independent chains (one chain was an add + a wait state per pair); cast to bf16 in accumulator order = B fragments of
      // the next product
      constexpr float LOG2E = 1.4426950408889634f;
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      const f32x2_t le2 = {LOG2E, LOG2E}, nm2 = {-m_use * LOG2E, -m_use * LOG2E};
      f32x2_t l2[2] = {{0.f, 0.f}, {0.f, 0.f}};
      bf16x8_t pf[2][2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          const f32x2_t s2 = {accS[sub][v], accS[sub][v + 1]};
          const f32x2_t e2 = __builtin_elementwise_fma(s2, le2, nm2);  // v_pk_fma_f32: two elements per instruction
          f32x2_t p2;
          p2.x = __builtin_amdgcn_exp2f(e2.x);
          p2.y = __builtin_amdgcn_exp2f(e2.y);
          l2[(v >> 1) & 1] += p2;
          pf[sub][v >> 3][v & 7] = (bf16)p2.x;
          pf[sub][v >> 3][(v & 7) + 1] = (bf16)p2.y;
        }
      l_run += (l2[0].x + l2[1].x) + (l2[0].y + l2[1].y);
      // O^T += V^T . P^T : element j of half hh is key 16 s2 + 8(j>>2) + 4hh + (j&3) of the sub-tile = chunk 2(2 sub + s2) + hh
      // of the regrouped V^T row
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(vb + (t * 32 + r) * 128 + (((2 * (2 * sub + s2) + hh) ^ swz) << 4));
            accO[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[sub][s2], accO[t], 0, 0, 0);
          }
    }
#ifdef VX_STAMPS
    if (it == 3) { asm volatile("" : "+v"(accO[0]), "+v"(accO[1])); VX_STAMP(26); }  // exponentials + second product done
#endif
    if (it + 1 < niter) lstore(RS, cur ^ 1);
#ifdef VX_STAMPS
    if (it == 3) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); VX_STAMP(27); }  // next tile's loads landed and stored
#endif
    __syncthreads();
#ifdef VX_STAMPS
    if (it == 3) VX_STAMP(28);
#endif
  };
  // iterations whose tiles every valid query of the workgroup sees in full (a query row past M is never stored: it may see anything)
  const int wg_row0 = bx * 32 * NW;
  const int min_limit = text_len < 0 ? M : (wg_row0 < text_len ? text_len : wg_row0 + 1);
  const int n_plain = min(niter, (min_limit / 64) / KG);
  constexpr std::false_type PLAIN{};
  constexpr std::true_type MASK{};
  int it = 0;
  if (TWO) {
    for (; it + 1 < n_plain; it += 2) {
      step(PLAIN, it, R0, R1);
      step(PLAIN, it + 1, R1, R0);
    }
    for (; it < niter; it += 2) {  // `it` is even here
      step(MASK, it, R0, R1);
      if (it + 1 < niter) step(MASK, it + 1, R1, R0);
    }
  } else {
    for (; it < n_plain; ++it) step(PLAIN, it, R0, R0);
    for (; it < niter; ++it) step(MASK, it, R0, R0);
  }
  VX_STAMP(22);
  float l_tot = l_run + xor32_f(l_run);
  if constexpr (KG > 1) {
    // merge the key groups of every query slice: groups 1.. park (m, l, O^T) in LDS (the operand buffers are free after
    // the loop's last barrier), group 0 rescales everything to the common maximum, in group order, and finishes
    float* park = reinterpret_cast<float*>(lds_raw);  // [group - 1][slice][lane][36]: 34 used floats per lane
    if (kg > 0) {
      float* p = park + ((size_t)((kg - 1) * NW + qw) * 64 + lane) * 36;
      p[0] = m_run; p[1] = l_tot;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; v += 4) *reinterpret_cast<float4*>(p + 4 + t * 16 + v) = make_float4(accO[t][v], accO[t][v + 1], accO[t][v + 2], accO[t][v + 3]);
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int g = 1; g < KG; ++g) {
      const float* p = park + ((size_t)((g - 1) * NW + qw) * 64 + lane) * 36;
      const float m1 = p[0], l1 = p[1];
      const float mm = fmaxf(m_run, m1);
      const float f0 = (m_run == -INFINITY) ? 0.f : __expf(m_run - mm), f1 = (m1 == -INFINITY) ? 0.f : __expf(m1 - mm);
      l_tot = l_tot * f0 + l1 * f1;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) accO[t][v] = accO[t][v] * f0 + p[4 + t * 16 + v] * f1;
      m_run = mm;
    }
  }
  if (qvalid) {
    const float inv = 1.0f / l_tot;
    bf16* op = out + (size_t)qrow * d + head * HD;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {  // registers 4g4..4g4+3 = dims t*32 + 8*g4 + 4*hh + 0..3
        union { bf16 b[4]; uint2 u; } pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk.b[j] = (bf16)(accO[t][4 * g4 + j] * inv);
        *reinterpret_cast<uint2*>(op + t * 32 + 8 * g4 + 4 * hh) = pk.u;
      }
  }
  VX_STAMP(23);
}

// (channel, row) transpose of the V third of a packed qkv buffer (only for the stand-alone op test;
// in the engine the QKV GEMM epilogue writes V^T itself)
__global__ __launch_bounds__(256) void vt_from_qkv_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ vt, int M, int d,
                                                          int vt_ld) {
  const int m = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (m < vt_ld) vt[(size_t)c * vt_ld + m] = (m < M) ? qkv[(size_t)m * 3 * d + 2 * d + c] : (bf16)0.f;
}

static inline int mfma_attn_dispatch(const bf16* qkv, const bf16* vt, int vt_ld, bf16* out, int M, int d, int H,
                                     int text_len, hipStream_t s, const int* seg_start = nullptr,
                                     const int* seg_len = nullptr, int nseg = 1, int max_seg_len = 0,
                                     const int* seg_text = nullptr) {
  const int rows = seg_start ? max_seg_len : M;
  // the kernel addresses both operands with 32-bit byte offsets (from the segment's first row / the V^T buffer's start)
  if ((unsigned long long)rows * 3ull * d * 2ull >= (1ull << 32) || (unsigned long long)d * vt_ld * 2ull >= (1ull << 32)) return 1;
  // fewer than ~2 workgroups per CU: split the keys over two wave groups inside the workgroup (4 waves = all 4 SIMDs)
  static const int kgsel = [] { const char* v = getenv("VX_ATTN_KG"); return v ? atoi(v) : 0; }();  // A/B runs
  static const int nwsel = [] { const char* v = getenv("VX_ATTN_NW"); return v ? atoi(v) : 0; }();  // A/B runs: query waves per workgroup
  static bool attr_dev[16] = {}; bool& attr_done = attr_dev[vx_cur_device()];
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)mfma_attn_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)mfma_attn_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    attr_done = true;
  }
  const long long nwg2 = (long long)((rows + 63) / 64) * H * (seg_start ? nseg : 1);
  const int kg = kgsel ? kgsel : (nwg2 < 512 ? 2 : 1);  // 4 groups measured slower than 2 at batch-1 NAR (10.7 vs 10.3 ms)
  // many workgroups (batched NAR / prefill): FOUR query waves share every staged K / V tile (128 query rows per workgroup)
  const int nw = kg != 1 ? 2 : nwsel ? (nwsel == 4 ? 4 : 2) : (nwg2 >= 2048 ? 4 : 2);  // 8 waves: 105.4 vs 101.8 ms (profiles/r03_notes.md)
  dim3 grid((rows + 32 * nw - 1) / (32 * nw), H, seg_start ? nseg : 1);
  if (kg == 4)
    mfma_attn_kernel<2, 4><<<grid, 2 * 4 * 64, 131072, s>>>(qkv, vt, out, M, vt_ld, d, text_len, seg_start, seg_len, seg_text);
  else if (kg == 2)
    mfma_attn_kernel<2, 2><<<grid, 2 * 2 * 64, 65536, s>>>(qkv, vt, out, M, vt_ld, d, text_len, seg_start, seg_len, seg_text);
  else if (nw == 4)
    mfma_attn_kernel<4, 1><<<grid, 4 * 64, 32768, s>>>(qkv, vt, out, M, vt_ld, d, text_len, seg_start, seg_len, seg_text);
  else
    mfma_attn_kernel<2, 1><<<grid, 2 * 64, 32768, s>>>(qkv, vt, out, M, vt_ld, d, text_len, seg_start, seg_len, seg_text);
  return 0;
}

}  // namespace vx
