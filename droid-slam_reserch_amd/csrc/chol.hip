// chol.hip -- on-device dense fp64 Cholesky solve of the reduced camera system.
//
// Replaces the reference's host-side Eigen::SimplicialLLT<double> (SparseBlock::solve,
// /root/reference/src/droid_kernels.cu:1192-1213): same damping `diag += ep + lm*diag` (:1197),
// same fp64 LL^T factorisation of the same matrix, failure => caller zeroes dx (:1207-1210).
//
// Layout: S is (n+1) x ld row-major, ld = n+1 rounded up to a multiple of 16 (128-byte rows); the
// lower triangle of S[0:n,0:n] is the matrix, row n holds the right-hand side.  Factoring the augmented matrix
// turns row n into y^T = (L^-1 b)^T for free (the forward substitution rides along with the panel TRSM), so only
// the backward substitution L^T x = y remains.
//
// The solve is a latency chain (n = 6P sequential pivots), not a flop problem, so the design
// minimises the dependent path per pivot:
//   * blocked right-looking factorisation, NB = 64.  One 512-thread workgroup handles one 64x64 tile of one
//     block-column step (`chol_tile`): trailing tiles get their rank-64 update on v_mfma_f64_16x16x4; tiles of the
//     next block column additionally re-factor the diagonal tile in LDS (redundantly, cheaper than a hand-off
//     inside the step) as 4 steps of {16x16 in-register wave factorisation with DPP-broadcast multipliers and an
//     rsqrt+Newton pivot reciprocal, triangular solves as GEMMs with the block inverse, MFMA block updates} and
//     solve their own tile against it;
//   * `chol_factor_persistent_kernel`: the whole factorisation in ONE launch of a co-resident grid (static tile
//     ownership, hand-off flags for trailing tiles, data-tagged 16-column strips for the panel chain);
//     `chol_step_kernel` (one launch per block column, same body) is the fallback;
//   * `chol_backsolve_persistent_kernel`: the backward substitution in one launch, x itself is the hand-off flag.
#include <hip/hip_runtime.h>
#include <mutex>

#include <cstdint>
#include <cstdlib>

#include "ba_internal.hpp"

namespace droid {

constexpr int NB = CHOL_NB;

#ifdef CHOL_STAMPS
// Diagnostic build only.  Per-step kernel: s_memtime stamps of wave 0 of the first two panel workgroups.
// Single-launch kernel: wall-clock (100 MHz, chip-wide) stamps of the diagonal workgroup and the one below it;
// slots 11..13 = {wait begins, inputs seen, tile published}.
__device__ unsigned long long g_chol_stamps[64 * 16];
__shared__ unsigned long long g_chol_lstamps[16];  // slots 14, 15: inside wave 0's first solve GEMM (operands loaded, MFMAs done)
#define STAMP(slot)                                                                         \
  do {                                                                                      \
    if (!PERSIST && threadIdx.x == 0 && panel && (blockIdx.x < 2)) {                                     \
      g_chol_stamps[((k + 1) & 31) * 32 + blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    }                                                                                       \
    if (PERSIST && threadIdx.x == 0 && panel && (bi - kp < 2)) {                              \
      /* into LDS (a global store here would sit in front of the next s_waitcnt vmcnt(0)); flushed at slot 8 */ \
      g_chol_lstamps[slot] = wall_clock64();                                                  \
      if ((slot) == 2) { g_chol_lstamps[14] = 0; g_chol_lstamps[15] = 0; g_chol_lstamps[9] = 0; }                    \
      if ((slot) == 8)                                                                        \
        for (int ss = 0; ss < 16; ss++)                                                       \
          if (ss < 11 || ss > 13) g_chol_stamps[((k + 1) & 31) * 32 + (bi - kp) * 16 + ss] = g_chol_lstamps[ss]; \
    }                                                                                       \
  } while (0)
#define PSTAMP(slot)                                                                        \
  do {                                                                                      \
    if (t == 0 && panel && (bi - kp < 2)) g_chol_stamps[(kp & 31) * 32 + (bi - kp) * 16 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define STAMP(slot) do { } while (0)
#define PSTAMP(slot) do { } while (0)
#endif
constexpr int LDP = NB + 2;  // LDS row pitch in doubles: rows stay 16-B aligned, b64 MFMA operand reads conflict-free
constexpr int WLP = 18;   // row pitch of an inverse block: with 16 the 16 rows of a GEMM operand read fell on two banks
constexpr int WLB = 16 * WLP;  // doubles per inverse block

typedef double f64x4 __attribute__((ext_vector_type(4)));


typedef double f64x2 __attribute__((ext_vector_type(2)));
// LDS-qualified pointers: every helper below takes its tiles through these, so non-inlined helpers
// read them with ds_read instead of flat loads and no generic-pointer casts are generated
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) f64x2 lds_f64x2;
#define DROID_LDS(arr) ((lds_f64*)(arr))
// global-memory-qualified pointers for the same reason: inside a non-kernel function a plain double* is a
// generic pointer and every access a flat instruction that also ties up the LDS counter
typedef __attribute__((address_space(1))) double gbl_f64;
typedef __attribute__((address_space(1))) f64x2 gbl_f64x2;
// COH: data handed between workgroups of the single-launch factorisation while the kernel runs.  Agent-scope
// relaxed atomics = sc1 loads/stores: written through to memory and re-fetched past the per-XCD L2s, so the
// hand-off needs no L2 write-back / invalidate (an agent-scope release+acquire fence pair per tile cost more
// than the kernel boundary it replaced).
template <bool COH>
__device__ __forceinline__ double gload(const gbl_f64* p) {
  if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <bool COH>
__device__ __forceinline__ void gstore(gbl_f64* p, double v) {
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// Cooperative load of a 64x64 fp64 tile (rows r0.., cols c0..c0+63 of S) into LDS by 256 threads:
// 8 independent 16-byte loads per thread are issued before the first LDS store, so the tile costs
// one memory round trip instead of sixteen.  Rows >= row_end and columns >= col_end read as
// `fill_diag` on the (tile-local) diagonal and 0 elsewhere; with `lower` only j <= i is kept.
// Requires ld % 2 == 0 and c0 % 2 == 0 (16-byte aligned rows).
template <bool COH = false>
__device__ __forceinline__ void load_tile64(lds_f64* __restrict__ dst, const gbl_f64* __restrict__ S,
                                            int ld, int r0, int c0, int row_begin, int row_end,
                                            int col_end, bool lower, double fill_diag) {
  const int t = threadIdx.x & 255;  // a 256-thread group (workgroups of 512 threads run two)
  f64x2 v[8];
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int idx2 = it * 256 + t;
    const int i = idx2 >> 5, j = (idx2 & 31) * 2;
    const bool ok = (r0 + i >= row_begin) && (r0 + i < row_end) && (c0 + j < col_end);
    const gbl_f64* p = S + (size_t)(ok ? r0 + i : 0) * ld + (ok ? c0 + j : 0);
    if (COH) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[it]) : "v"(p) : "memory");
    else v[it] = *(const gbl_f64x2*)p;
    if (!COH && !ok) v[it] = (f64x2){0.0, 0.0};
  }
  if (COH) {  // the compiler does not count loads issued from inline asm
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 8; it++) {
      asm volatile("" : "+v"(v[it]));
      const int idx2 = it * 256 + t;
      const int i = idx2 >> 5, j = (idx2 & 31) * 2;
      if (!((r0 + i >= row_begin) && (r0 + i < row_end) && (c0 + j < col_end))) v[it] = (f64x2){0.0, 0.0};
    }
  }
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int idx2 = it * 256 + t;
    const int i = idx2 >> 5, j = (idx2 & 31) * 2;
    double a = v[it][0], b = v[it][1];
    const bool rok = (r0 + i >= row_begin) && (r0 + i < row_end);
    if (!(rok && c0 + j + 1 < col_end)) b = 0.0;
    if (lower) {
      if (j > i) a = 0.0;
      if (j + 1 > i) b = 0.0;
    }
    if (fill_diag != 0.0) {
      if (j == i && !(rok && c0 + j < col_end)) a = fill_diag;
      if (j + 1 == i && !(rok && c0 + j + 1 < col_end)) b = fill_diag;
    }
    *(lds_f64x2*)(&dst[i * LDP + j]) = (f64x2){a, b};
  }
}

// Store rows [row_begin,row_end) x cols [0,ncols) of an LDS tile back (optionally lower part only).
template <bool COH = false>
__device__ __forceinline__ void store_tile64(gbl_f64* __restrict__ S, const lds_f64* __restrict__ src,
                                             int ld, int r0, int c0, int row_begin, int row_end,
                                             int ncols, bool lower) {
  const int t = threadIdx.x & 255;
  if (COH) {  // 16-byte write-through stores (8-byte ones take 2.5x as long per tile); never `lower`
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int idx2 = it * 256 + t;
      const int i = idx2 >> 5, j = (idx2 & 31) * 2;
      if (r0 + i >= row_begin && r0 + i < row_end && j < ncols) {
        gbl_f64* p = &S[(size_t)(r0 + i) * ld + c0 + j];
        const f64x2 v = *(const lds_f64x2*)&src[i * LDP + j];
        if (j + 1 < ncols) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        else gstore<true>(p, v[0]);
      }
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const int idx = it * 256 + t;
    const int i = idx >> 6, j = idx & 63;
    if (r0 + i >= row_begin && r0 + i < row_end && j < ncols && (!lower || j <= i))
      S[(size_t)(r0 + i) * ld + c0 + j] = src[i * LDP + j];
  }
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(d) to fp64 accuracy: v_rsq_f64 seed (24 bits, tools/micro/rsq_acc.hip) + ONE third-order (Halley) step
//   e = 1 - d y^2,  y <- y + y e (1/2 + 3/8 e)        (next term 5/16 e^3 ~ 4e-23)
// = 5 instructions instead of the 8 of two Newton steps (-2.5 us per 1530-pivot factorisation, same residuals).
__device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
#ifdef CHOL_TWO_NEWTON
  double e = fma(-d * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  e = fma(-d * y, y, 1.0);
  y = fma(0.5 * y, e, y);
#else
  const double e = fma(-d * y, y, 1.0);
  const double p = fma(0.375, e, 0.5);
  y = fma(y * e, p, y);
#endif
  return y;
}

// MFMA operand fetch from an LDS tile of pitch LDP.  `volatile` on purpose: left alone the compiler fuses neighbouring
// 8-byte reads into ds_read2_b64, which the LDS services in 16-lane groups with banks taken mod 32 -- 8 LDS cycles per
// instruction and, with 16-byte aligned rows, always a 2-way conflict -- where two plain ds_read_b64 (32-lane groups,
// 64 banks, conflict-free at pitch 66) take 4 cycles.  PMC on the factorisation before the change: 172 ds_read2_b64
// sites, SQ_LDS_BANK_CONFLICT = 42 % of SQ_LDS_IDX_ACTIVE, LDS busy ~70 % of a panel workgroup's time.
__device__ __forceinline__ double lds_operand(const lds_f64* p) { return *(const volatile lds_f64*)p; }

// C(16x16) -= A(16x16) * B(16x16)^T on one wave.  A, C: LDS blocks with row pitch LDP; B: LDS
// block with row pitch ldb.  v_mfma_f64_16x16x4_f64: lane l feeds A[l&15][l>>4], B^T[l>>4][l&15];
// result register i of lane l is D[(l>>4) + 4i][l&15].
// NOT inlined on purpose: the panel path calls this from a dozen places, each executed once or
// twice per workgroup; inlined, every call site is cold code and the instruction-cache misses cost
// 4-5x the arithmetic (measured with s_memtime: 2.3-3.1k cycles cold vs 0.5k warm per call).
template <bool ASSIGN>
__device__ __forceinline__ void wave_gemm_nt16_inl(lds_f64* Cl, const lds_f64* Al, const lds_f64* Bl, int ldb) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
#ifdef CHOL_GEMM_FLAT
  // (round 1 read the operands with FLAT loads: its ds_read form was 1.3 us slower per block column -- because the compiler
  // had fused the reads into conflicting ds_read2_b64, see lds_operand() -- kept for A/B builds)
  double* C = (double*)Cl;
  const double* A = (const double*)Al;
  const double* B = (const double*)Bl;
  asm volatile("" : "+v"(C), "+v"(A), "+v"(B));
#define DROID_GEMM_LD(p) (*(p))
#else
  lds_f64* C = Cl;
  const lds_f64* A = Al;
  const lds_f64* B = Bl;
#define DROID_GEMM_LD(p) lds_operand(p)
#endif
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  double av[4], bv[4], cv[4];
#ifdef CHOL_STAMPS
  if (ASSIGN && threadIdx.x == 0 && g_chol_lstamps[9] == 0) g_chol_lstamps[9] = wall_clock64();   // function entered
#endif
#pragma unroll
  for (int k = 0; k < 4; k++) { av[k] = DROID_GEMM_LD(&A[r * LDP + 4 * k + g]); bv[k] = DROID_GEMM_LD(&B[r * ldb + 4 * k + g]); }
  if (!ASSIGN) {  // the block to update travels with the operands instead of after the products
#pragma unroll
    for (int i = 0; i < 4; i++) cv[i] = DROID_GEMM_LD(&C[(g + 4 * i) * LDP + r]);
  }
#ifdef CHOL_STAMPS
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (ASSIGN && threadIdx.x == 0 && g_chol_lstamps[14] == 0) g_chol_lstamps[14] = wall_clock64();
#endif
  {  // two chains of two instead of one of four: a dependent fp64 MFMA waits 18 cycles beyond the 64 of its predecessor
    f64x4 acc1 = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc1, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc1, 0, 0, 0);
    acc += acc1;
  }
#ifdef CHOL_STAMPS
  asm volatile("s_nop 15\n s_nop 15" : "+v"(acc));
  if (ASSIGN && threadIdx.x == 0 && g_chol_lstamps[15] == 0) g_chol_lstamps[15] = wall_clock64();
#endif
#pragma unroll
  for (int i = 0; i < 4; i++) C[(g + 4 * i) * LDP + r] = ASSIGN ? acc[i] : cv[i] - acc[i];
#undef DROID_GEMM_LD
}
template <bool ASSIGN>
__device__ __attribute__((noinline)) void wave_gemm_nt16(lds_f64* Cl, const lds_f64* Al, const lds_f64* Bl, int ldb) {
  wave_gemm_nt16_inl<ASSIGN>(Cl, Al, Bl, ldb);
}

// In-register Cholesky of the 16x16 block at Lb (LDS, pitch LDP) by one wave.  Lane l < 16 holds
// row l of the block; lanes 16..31 hold the rows of the identity.  The factorisation is a chain
// of column operations (scale column j, subtract multiples of it from the later columns) whose
// scalars travel by v_readlane; applied to the identity rows as well they leave L^-T there, so
// the inverse comes for free and every triangular solve against this block becomes a GEMM.
// Outputs: L (lower part, in place) and Wl[j*WLP + k] = (L^-1)[j][k].
// The wave runs alone on its SIMD and issues ~one instruction per 4-6 cycles whatever the type, so
// the instruction COUNT is the cost (measured: 3.1-3.5 cycles per removed instruction).  Hence:
//   * lane r (0..15) keeps row r of the block AND row r of the identity (32 doubles), so that all
//     multipliers L[c][j] come from lanes of the same 16-lane row and travel by DPP
//     row_newbcast inside the FMA itself (v_fmac_f64_dpp: one instruction per updated element
//     instead of two v_readlane + one FMA through an SGPR pair);
//   * the pivot is broadcast the same way (v_mov_b64_dpp) and 1/sqrt is seeded by v_rsq_f64;
//   * no per-pivot positivity test: a pivot <= 0 or NaN turns 1/sqrt into NaN, which reaches the
//     last pivot through the updates, so ONE test at the end sees any failure;
//   * unconditional row stores (the strict upper triangle of a diagonal block is never read).
// DPP reads of a VGPR written by the previous VALU instruction need 2 wait states; inline asm is
// invisible to hipcc's hazard recogniser, hence the explicit s_nop in front of the first DPP use.
template <int J, int C>
__device__ __forceinline__ void fmac_bcast(double& acc, const double& piv, const double& own) {
  // acc -= piv[lane C of this 16-lane row] * own
  asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc) : "v"(piv), "v"(own), "n"(C));
}
template <int J>
__device__ __forceinline__ void potrf16_pivot(double (&a)[16], double (&w)[16], double& ylast) {
  double d;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
               : "=v"(d) : "v"(a[J]), "n"(J));
  const double y = rsqrt_nr(d);
  if (J == 15) ylast = y;
  a[J] *= y;
  w[J] *= y;
  asm volatile("s_nop 1" : "+v"(a[J]));  // a[J] is the DPP operand of everything below
#pragma unroll
  for (int c = J + 1; c < 16; c++) {
    // unrolled with compile-time lane numbers
    switch (c) {
#ifdef CHOL_NO_W  // timing experiment only (wrong inverses): the column operations on the identity rows left out
#define DROID_CASE(CC) case CC: if (CC > J) { fmac_bcast<J, CC>(a[CC], a[J], a[J]); } break;
#else
#define DROID_CASE(CC) case CC: if (CC > J) { fmac_bcast<J, CC>(a[CC], a[J], a[J]); fmac_bcast<J, CC>(w[CC], a[J], w[J]); } break;
#endif
      DROID_CASE(1) DROID_CASE(2) DROID_CASE(3) DROID_CASE(4) DROID_CASE(5) DROID_CASE(6) DROID_CASE(7) DROID_CASE(8)
      DROID_CASE(9) DROID_CASE(10) DROID_CASE(11) DROID_CASE(12) DROID_CASE(13) DROID_CASE(14) DROID_CASE(15)
#undef DROID_CASE
      default: break;
    }
  }
}

__device__ __forceinline__ void wave_potrf16(lds_f64* Lb, lds_f64* Wl, const lds_f64* Idn, int* fail, bool report) {
  const int lane = threadIdx.x & 63, row = lane & 15;
  __builtin_amdgcn_s_setprio(3);  // the pivot chain outranks the MFMA waves sharing this SIMD
  double a[16], w[16];
#pragma unroll
  for (int c = 0; c < 16; c++) {
    a[c] = Lb[row * LDP + c];
    w[c] = Idn[row * WLP + c];
  }
  double ylast = 0.0;
  potrf16_pivot<0>(a, w, ylast);   potrf16_pivot<1>(a, w, ylast);   potrf16_pivot<2>(a, w, ylast);
  potrf16_pivot<3>(a, w, ylast);   potrf16_pivot<4>(a, w, ylast);   potrf16_pivot<5>(a, w, ylast);
  potrf16_pivot<6>(a, w, ylast);   potrf16_pivot<7>(a, w, ylast);   potrf16_pivot<8>(a, w, ylast);
  potrf16_pivot<9>(a, w, ylast);   potrf16_pivot<10>(a, w, ylast);  potrf16_pivot<11>(a, w, ylast);
  potrf16_pivot<12>(a, w, ylast);  potrf16_pivot<13>(a, w, ylast);  potrf16_pivot<14>(a, w, ylast);
  potrf16_pivot<15>(a, w, ylast);
  if (lane < 16) {  // row of L, and row `row` of L^-T: (L^-1)[j][row] = w[j]
#pragma unroll
    for (int c = 0; c < 16; c++) {
      Lb[row * LDP + c] = a[c];
      Wl[c * WLP + row] = w[c];
    }
  }
  const bool bad = !(ylast > 0.0 && ylast < 1.0e300);  // NaN (non-positive pivot somewhere) or overflow
  if (lane == 0 && bad && report) atomicMax(fail, 1);
  __builtin_amdgcn_s_setprio(0);
}


// 16-row strip of a 64x64 tile update: acc[0..NN) (row group rg, column tiles nt0..nt0+NN) -= Lr Lc^T.
template <int NN>
__device__ __forceinline__ void strip_update(f64x4 (&acc)[4], const lds_f64* Lr, const lds_f64* Lc, int rg,
                                             int nt0) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const lds_f64* Ar = &Lr[(16 * rg + r) * LDP + g];
#pragma unroll 4
  for (int kk = 0; kk < NB; kk += 4) {
    const double a = -lds_operand(&Ar[kk]);
#pragma unroll
    for (int nn = 0; nn < NN; nn++)
      acc[nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, lds_operand(&Lc[(16 * (nt0 + nn) + r) * LDP + kk + g]), acc[nn], 0, 0, 0);
  }
}

// accumulator fragment <- global: element i of tile nn is (row 16*rg + (lane>>4) + 4i, col 16*(nt0+nn) + (lane&15))
template <int NN>
__device__ __forceinline__ void frag_load_global(f64x4 (&acc)[4], const gbl_f64* __restrict__ S, int ld,
                                                 int r0, int q0, int rg, int nt0, int row_end, int col_end,
                                                 bool lower, bool coh = false) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int nn = 0; nn < NN; nn++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int li = 16 * rg + g + 4 * i, lj = 16 * (nt0 + nn) + r;
      const int row = r0 + li, col = q0 + lj;
      double v = 0.0;
      if (row < row_end && col < col_end && (!lower || lj <= li))
        v = coh ? gload<true>(&S[(size_t)row * ld + col]) : S[(size_t)row * ld + col];
      acc[nn][i] = v;
    }
}

// single-tile variants: one 16x16 accumulator fragment (row group rg, column tile nt)
__device__ __forceinline__ void strip_update_tile(f64x4& acc, const lds_f64* Lr, const lds_f64* Lc, int rg, int nt) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const lds_f64* Ar = &Lr[(16 * rg + r) * LDP + g];
  const lds_f64* Br = &Lc[(16 * nt + r) * LDP + g];
  // fully unrolled: all 32 operand reads of the tile are in flight before the first MFMA, so the
  // LDS latency is paid once per tile and the 16 MFMAs issue back to back
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-lds_operand(&Ar[kk]), lds_operand(&Br[kk]), acc, 0, 0, 0);
}
template <bool COH = false>
__device__ __forceinline__ void frag_load_tile(f64x4& acc, const gbl_f64* __restrict__ S, int ld, int r0, int q0,
                                               int rg, int nt, int row_end, int col_end, bool lower, bool coh = false) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int li = 16 * rg + g + 4 * i, lj = 16 * nt + r;
    const int row = r0 + li, col = q0 + lj;
    double v = 0.0;
    if (row < row_end && col < col_end && (!lower || lj <= li))
      v = (COH || coh) ? gload<true>(&S[(size_t)row * ld + col]) : gload<false>(&S[(size_t)row * ld + col]);
    acc[i] = v;
  }
}

// acc (16x16 fragment) -= A(16x16 LDS block) * B(16x16 LDS block)^T, both with row pitch LDP
__device__ __forceinline__ void block_update16(f64x4& acc, const lds_f64* A, const lds_f64* B) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  double av[4], bv[4];  // all eight operands in flight before the first product (the volatile reads otherwise wait pair by pair)
#pragma unroll
  for (int k = 0; k < 4; k++) { av[k] = lds_operand(&A[r * LDP + 4 * k + g]); bv[k] = lds_operand(&B[r * LDP + 4 * k + g]); }
#pragma unroll
  for (int k = 0; k < 4; k++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[k], bv[k], acc, 0, 0, 0);
}

// One step of the blocked right-looking factorisation in ONE launch of 512-thread workgroups:
//   every tile (bi >= bj > k) of the trailing matrix:  A[bi,bj] -= L[bi,k] L[bj,k]^T
//   tiles of block column k+1 additionally finish panel k+1: they rebuild and factor the updated
//   diagonal tile A[k+1,k+1] (redundantly per workgroup: cheaper than a grid hand-off) and solve
//   their own tile against it, so panel k+1 is ready when the launch ends.
// k = -1 is the initial panel (no update).  Grid: (row blocks, column blocks) from k+1.
//
// Panel workgroups are organised around the pivot chain (wave 0), which is the critical path:
//   * up front only the diagonal tile and column 0 of the tile to solve get the rank-64 update of
//     the previous panel, spread as 16x16 tiles over the eight waves (wave 0: block (0,0) only,
//     then straight into its factorisation; wave 4, which shares wave 0's SIMD, one tile);
//   * per 16-column step p: {16x16 solves as GEMMs with the block inverse: one block per wave},
//     then wave 0 updates block (p+1,p+1) and factors it while the other waves apply the
//     remaining rank-16 updates of the diagonal tile and finish column p+1 of the tile to solve
//     (its rank-64 update was deferred to this slot, where the matrix pipe is otherwise idle).
// hand-off flags of the single-launch factorisation (agent scope = sc1: other XCDs see them)
constexpr int CFP_SPIN_LIMIT = 1 << 20;

__device__ __forceinline__ int cfp_load(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cfp_store(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16-column strip `sp` of a final panel tile -> LDS, by a 256-thread group, from the tile's hand-over slot
// (64x64, row pitch 64).  The slot is data-tagged: it was preset to 0xFF bytes, the producer writes the strip with
// 16-byte sc1 stores as soon as it is solved, and every thread here re-reads its two 16-byte units until none of
// the four values is the preset pattern.  No flag, no store acknowledgement on the producer side (that hand-off
// took 3 us of the 14 per block column).  Rows outside [row_begin, row_end) are not part of the tile: zeros.
constexpr long long CFP_TAG = -1LL;  // 0xFF bytes; a computed NaN never has this payload
__device__ __forceinline__ void load_strip64(lds_f64* __restrict__ dst, const gbl_f64* __restrict__ slot, int r0,
                                             int sp, int row_begin, int row_end, int* __restrict__ fail,
                                             int* __restrict__ abortf) {
  const int t = threadIdx.x & 255;
  f64x2 v[2];
  bool ok[2];
  const gbl_f64* p[2];
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int u = it * 256 + t;
    const int i = u >> 3, j = 16 * sp + (u & 7) * 2;
    ok[it] = (r0 + i >= row_begin) && (r0 + i < row_end);
    p[it] = slot + (size_t)(ok[it] ? i : 0) * NB + j;
  }
  int spins = 0;
  while (true) {
#pragma unroll
    for (int it = 0; it < 2; it++)
      asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[it]) : "v"(p[it]) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bool there = true;
#pragma unroll
    for (int it = 0; it < 2; it++) {
      asm volatile("" : "+v"(v[it]));
      if (ok[it] && (__double_as_longlong(v[it][0]) == CFP_TAG || __double_as_longlong(v[it][1]) == CFP_TAG)) there = false;
    }
    if (there) break;
    if (++spins > CFP_SPIN_LIMIT || ((spins & 255) == 0 && cfp_load(abortf) == 1)) {
      cfp_store(abortf, 1);  // never hang: the caller sees `abort` before its next tile, the result is discarded
      atomicMax(fail, 2);    // 2 = stalled grid (not a numerical failure): reported as STATUS_CHOL_STALL
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int u = it * 256 + t;
    const int i = u >> 3, j = 16 * sp + (u & 7) * 2;
    if (!ok[it]) v[it] = (f64x2){0.0, 0.0};
    *(lds_f64x2*)(&dst[i * LDP + j]) = v[it];
  }
}

// LDS tiles of a factorisation workgroup (145 KB: one workgroup per CU).  Declared at namespace scope so
// that the step body can be a real function of the single-launch kernel without passing LDS arrays
// around as generic pointers.
__shared__ double g_cholB0[NB * LDP];
__shared__ double g_cholB1[NB * LDP];
__shared__ double g_cholB2[NB * LDP];
__shared__ double g_cholBT[NB * LDP];
__shared__ double g_cholWl[4 * WLB];
__shared__ double g_cholIdn[WLB];  // (same padded pitch)

// The body is shared by the one-launch-per-step kernel and the single-launch kernel below
// (PERSIST: the workgroup is handed tile (bi, bj) of step k by its caller, the factored diagonal
// tile goes to the side buffer Ldiag instead of overwriting the tile other workgroups still read).
// Columns 16p..16p+15 of the solved tile (LDS, BT) are final: one wave writes them through and publishes the strip; the next
// block column's workgroups start their rank-16 updates on it right away.  Scalar base + 32-bit lane offsets: with
// per-lane 64-bit addresses and row predicates per store, ~200 loop-invariant instructions of this block were hoisted in
// front of the 16-column loop of the panel path, i.e. behind the first pivot block on EVERY wave including the pivot
// chain (0.25 us per block column, stamps).  Inlined on purpose: a function of its own has to wait for its write-through
// stores before it returns, which puts their round trip in front of the next barrier (+0.4 us, stamps).
__device__ __forceinline__ void chol_strip_out(gbl_f64* __restrict__ S, int ld, gbl_f64* __restrict__ slot,
                                                         const lds_f64* __restrict__ BT, int r0, int c0, int wk, int nrows,
                                                         int p, int* done_flag, int done_value) {
  const int lane = threadIdx.x & 63;
  const int rb = max(r0, c0 + wk) - r0, re = min(r0 + NB, nrows) - r0;   // tile-local row range to publish
  const int i0 = lane >> 3, j = 16 * p + (lane & 7) * 2;
  // scalar base + 32-bit lane offset (the tile spans < 4 GB): no 64-bit address arithmetic per store
  auto uniform = [](const gbl_f64* q) {  // (wave-uniform by construction; the compiler cannot see it)
    const unsigned long long a = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (const gbl_f64*)(((unsigned long long)hi << 32) | lo);
  };
  const gbl_f64* sbase = uniform(S + (size_t)r0 * ld + c0);
  const gbl_f64* sslot = uniform(slot);
  unsigned off_slot = (unsigned)((i0 * NB + j) * 8), off_mat = (unsigned)((i0 * ld + j) * 8);
  const unsigned step_mat = (unsigned)(8 * ld * 8);
  const lds_f64* src = &BT[i0 * LDP + j];
  if (wk == NB) {
#pragma unroll
    for (int it = 0; it < 8; it++) {  // first the hand-over slot the panel chain polls ...
      const int i = i0 + 8 * it;
      if (i >= rb && i < re) {
        const f64x2 v = *(const lds_f64x2*)&src[8 * it * LDP];
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(off_slot + 8u * it * NB * 8u), "v"(v), "s"(sslot) : "memory");
      }
    }
  }
#pragma unroll
  for (int it = 0; it < 8; it++) {  // ... then the matrix itself (trailing tiles and the back-substitution read it)
    const int i = i0 + 8 * it;
    if (i >= rb && i < re && j < wk) {
      const f64x2 v = *(const lds_f64x2*)&src[8 * it * LDP];
      if (j + 1 < wk) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(off_mat), "v"(v), "s"(sbase) : "memory");
      else asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(off_mat), "v"(v[0]), "s"(sbase) : "memory");
    }
    off_mat += step_mat;
  }
  // trailing tiles and the back-substitution read the matrix copy only when the whole tile is out: one
  // acknowledgement wait and one flag, after the last strip (the panel chain polls the hand-over slot instead)
  if (done_flag != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) cfp_store(done_flag, done_value);
  }
}

// coh_first (overlap mode, see chol_factor_persistent_kernel): the tile's own values were written by ANOTHER kernel
// while this one was already running (no kernel-boundary invalidate in between), so this read of them must go past
// the XCD's L2 (sc1); later reads see the workgroup's own write-through stores as before.
// OV = false compiles the two overlap-mode parameters out (the default path pays nothing for them).
template <bool PERSIST, bool OV = false>
__device__ __forceinline__ void chol_tile(gbl_f64* __restrict__ S, int n, int ld, int k, int bi, int bj,
                                          int* __restrict__ fail, double lm, double ep,
                                          gbl_f64* __restrict__ Ldiag, int* __restrict__ done,
                                          int* __restrict__ abortf, bool coh_first_arg = false, int k_last_arg = -2) {
  const bool coh_first = OV && coh_first_arg;
  const int k_last = OV ? k_last_arg : -2;
  lds_f64* const B0 = DROID_LDS(g_cholB0);      // L[bi,k]          (workgroup-local tiles, see their declaration)
  lds_f64* const B1 = DROID_LDS(g_cholB1);      // L[bj,k]
  lds_f64* const B2 = DROID_LDS(g_cholB2);      // the diagonal tile D -> L
  lds_f64* const BT = DROID_LDS(g_cholBT);      // the tile being solved (T -> X)
  lds_f64* const Wl = DROID_LDS(g_cholWl);      // inverses of the four 16x16 diagonal blocks
  lds_f64* const Idn = DROID_LDS(g_cholIdn);    // 16x16 identity: initial rows of the lanes that build L^-T
  const int nrows = n + 1;          // row n = right-hand side
  const int kp = k + 1;             // block column being finished
  if (bj > bi) return;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int grp = wave >> 2, w4 = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int r0 = bi * NB, q0 = bj * NB;
  const int c0 = kp * NB;                 // panel column
  const int wk = min(NB, n - c0);         // its width
  const bool panel = (bj == kp);
  const bool diag = panel && (bi == kp);
  f64x4 acc[4];

  if (!panel) {
    // ---- plain trailing tile: wave = (row group w4, column tiles 2*grp, 2*grp+1)
    if (k < 0) {
      // initial launch: the spare workgroups of block column 1 apply the damping diag += ep + lm*diag
      // (SparseBlock::solve, dk:1197) to the rows below the first block; block 0 is damped on load
      // (single-launch kernel: done by the owners of the diagonal tiles before the first step)
      const int i = r0 + t;
      if (!PERSIST && bj == kp + 1 && t < NB && i >= NB && i < n) {
        const double d = S[(size_t)i * ld + i];
        S[(size_t)i * ld + i] = d + (ep + lm * d);
      }
      return;
    }
    const int p0 = k * NB;
    frag_load_global<2>(acc, S, ld, r0, q0, w4, 2 * grp, nrows, n, bi == bj, coh_first);
    if (PERSIST && bj == kp + 1) {
      // tiles of the NEXT panel column take the two panel tiles strip by strip out of the hand-over slots, like
      // the panel workgroups do: their update ends right behind the last strip instead of a flag, an
      // acknowledgement and two whole-tile loads later, so the next step's panel workgroups find their tile ready
      const int nrb = (nrows + NB - 1) / NB;
      const gbl_f64* slot0 = Ldiag + chol_lfin_offset(n) + (size_t)chol_tile_index(nrb, bi, k) * NB * NB;
      const gbl_f64* slot1 = Ldiag + chol_lfin_offset(n) + (size_t)chol_tile_index(nrb, bj, k) * NB * NB;
      const lds_f64* Bx = (bi == bj) ? B0 : B1;
      for (int sp = 0; sp < 4; sp++) {
        if (grp == 0) load_strip64(B0, slot0, r0, sp, r0, nrows, fail, abortf);
        else if (bi != bj) load_strip64(B1, slot1, q0, sp, q0, n, fail, abortf);
        __syncthreads();
#pragma unroll
        for (int nn = 0; nn < 2; nn++)
          block_update16(acc[nn], &B0[(16 * w4) * LDP + 16 * sp], &Bx[(16 * (2 * grp + nn)) * LDP + 16 * sp]);
      }
    } else {
    // k_last > k (overlap mode): a tile whose block column arrived late REPLAYS the updates of all block columns k..k_last
    // that are final by now in one go -- tile loaded once, one pair of panel tiles per column, stored once (2-3 us per
    // column instead of the 12 us of a full scheduler step)
    const int klast = (k_last > k) ? k_last : k;
    for (int kk = k; kk <= klast; kk++) {
      const int pk = kk * NB;
      if (kk > k) __syncthreads();   // the previous column's operand reads are over
      if (grp == 0) load_tile64<PERSIST>(B0, S, ld, r0, pk, r0, nrows, pk + NB, false, 0.0);
      else if (bi != bj) load_tile64<PERSIST>(B1, S, ld, q0, pk, q0, n, pk + NB, false, 0.0);
      __syncthreads();
      strip_update<2>(acc, B0, (bi == bj) ? B0 : B1, w4, 2 * grp);
    }
    }
#pragma unroll
    for (int nn = 0; nn < 2; nn++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int li = 16 * w4 + fg + 4 * i, lj = 16 * (2 * grp + nn) + fr;
        const int row = r0 + li, col = q0 + lj;
        if (row < nrows && col < n && (bi != bj || lj <= li)) gstore<PERSIST>(&S[(size_t)row * ld + col], acc[nn][i]);
      }
    return;
  }

  // ---- panel workgroup
  // rows to solve: the own tile, or for the diagonal workgroup the rows of its block below the
  // diagonal tile (only the rhs row, and only when the last block column is narrower than NB)
  const bool solve_rows = !diag || (wk < NB);
  if (t < 256) Idn[(t >> 4) * WLP + (t & 15)] = ((t >> 4) == (t & 15)) ? 1.0 : 0.0;  // visible after the barrier below
  STAMP(0);
  // 16x16 tiles (rg, nt): D = diagonal tile (lower part), T = the tile to solve.  Only what the
  // first pivot block and the first solve need is updated up front (all of D, column 0 of T);
  // columns 1..3 of T stay in registers and are finished one per 16-column step, left-looking,
  // while the pivot wave is busy with the next 16x16 factorisation:
  //   up front   w0: D00 | w1: D10 D11 | w2: D20 D21 | w3: D30 D31 | w4: T00 | w5: T10 D22 | w6: T20 D32 | w7: T30 D33
  //   deferred   T(0,q): wave q (q = 1..3);  T(g,q), g = 1..3: wave 4+g
  // tile code: bit 4 = T, bits 3:2 = rg, bits 1:0 = nt
  int ntile, code[2];
  switch (wave) {
    case 0: ntile = 1; code[0] = 0x00; code[1] = 0; break;
    case 1: ntile = 2; code[0] = 0x04; code[1] = 0x05; break;
    case 2: ntile = 2; code[0] = 0x08; code[1] = 0x09; break;
    case 3: ntile = 2; code[0] = 0x0c; code[1] = 0x0d; break;
    case 4: ntile = 1; code[0] = 0x10; code[1] = 0; break;
    case 5: ntile = 2; code[0] = 0x14; code[1] = 0x0a; break;
    case 6: ntile = 2; code[0] = 0x18; code[1] = 0x0e; break;
    default: ntile = 2; code[0] = 0x1c; code[1] = 0x0f; break;
  }
  const int dg = (wave >= 5) ? wave - 4 : 0;                       // row group of this wave's deferred tiles
  const int ndef = !solve_rows ? 0 : (wave >= 5 ? 3 : ((wave >= 1 && wave <= 3) ? 1 : 0));
  f64x4 tacc[2], dacc[3];
#pragma unroll
  for (int i = 0; i < 3; i++) dacc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 2; i++) {
    if (i < ntile) {
      const int rg = (code[i] >> 2) & 3, nt = code[i] & 3;
      if (code[i] & 0x10) {
        if (solve_rows) frag_load_tile(tacc[i], S, ld, r0, c0, rg, nt, nrows, n, diag, coh_first);   // tile to solve
      } else {
        frag_load_tile<PERSIST>(tacc[i], S, ld, c0, c0, rg, nt, c0 + wk, c0 + wk, true);   // diagonal tile
        if (k < 0 && rg == nt) {  // damping of block 0, applied to every workgroup's private copy
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (fg + 4 * e == fr) tacc[i][e] += ep + lm * tacc[i][e];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 3; i++)
    if (i < ndef) frag_load_tile(dacc[i], S, ld, r0, c0, dg, (wave >= 5) ? i + 1 : wave, nrows, n, diag, coh_first);
  if (!PERSIST && k >= 0) {
    if (grp == 0) load_tile64(B1, S, ld, c0, k * NB, c0, n, k * NB + NB, false, 0.0);
    else load_tile64(B0, S, ld, r0, k * NB, r0, nrows, k * NB + NB, false, 0.0);
  }
  if (PERSIST && k >= 0) {
    // The two panel tiles arrive in 16-column strips (their producers write a strip into the tile's data-tagged
    // hand-over slot as soon as it is solved): strips 0..2 and their rank-16 updates overlap with the producers'
    // remaining work, only the last strip's load and update stay in front of the first pivot block.
    const int nrb = (nrows + NB - 1) / NB;
    const gbl_f64* slot1 = Ldiag + chol_lfin_offset(n) + (size_t)chol_tile_index(nrb, kp, k) * NB * NB;  // L[kp,k]
    const gbl_f64* slot0 = Ldiag + chol_lfin_offset(n) + (size_t)chol_tile_index(nrb, bi, k) * NB * NB;  // L[bi,k]
    for (int sp = 0; sp < 4; sp++) {
#ifdef CHOL_DIRECT00
      // Candidate (i) of DESIGN section 8 (A/B build): the pivot wave takes ITS 16x16 block of the last
      // strip -- rows 0..15 of L[kp,k], the only operand of D00's last rank-16 update -- straight from the hand-over
      // slot into MFMA operands (8-byte sc1 loads, data-tagged like the strip loads) and applies the update in front
      // of the barrier instead of behind it (no LDS stage on the chain for this block).
      if (sp == 3 && wave == 0) {
        const gbl_f64* q = slot1 + (size_t)fr * NB + 48 + fg;
        const bool rowok = c0 + fr < n;
        double dv[4];
        int spins = 0;
        while (true) {
#pragma unroll
          for (int kk = 0; kk < 4; kk++)
            asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(dv[kk]) : "v"(q + 4 * kk) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          bool there = true;
#pragma unroll
          for (int kk = 0; kk < 4; kk++) {
            asm volatile("" : "+v"(dv[kk]));
            if (rowok && __double_as_longlong(dv[kk]) == CFP_TAG) there = false;
          }
          if (__all(there)) break;
          if (++spins > CFP_SPIN_LIMIT || ((spins & 255) == 0 && cfp_load(abortf) == 1)) {
            cfp_store(abortf, 1);
            atomicMax(fail, 2);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const double x = rowok ? dv[kk] : 0.0;
          tacc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x, x, tacc[0], 0, 0, 0);
        }
      }
#endif
      if (grp == 0) load_strip64(B1, slot1, c0, sp, c0, n, fail, abortf);
      else load_strip64(B0, slot0, r0, sp, r0, nrows, fail, abortf);
      __syncthreads();
      if (sp == 3) STAMP(1);
      if (sp == 3 && wave == 0) __builtin_amdgcn_s_setprio(3);
#ifdef CHOL_WARM
      if (sp == 1 && wave == 7) {  // experiment: fetch the GEMM helpers' code long before the chain calls them (B2 is not in use yet)
        wave_gemm_nt16<true>(&B2[48 * LDP + 48], &B2[48 * LDP + 32], &B2[32 * LDP + 48], LDP);
        wave_gemm_nt16<false>(&B2[48 * LDP + 48], &B2[48 * LDP + 32], &B2[32 * LDP + 48], LDP);
      }
#endif
#pragma unroll
      for (int i = 0; i < 2; i++) {
        if (i < ntile) {
          const int rg = (code[i] >> 2) & 3, nt = code[i] & 3;
          if (code[i] & 0x10) {
            if (solve_rows) block_update16(tacc[i], &B0[(16 * rg) * LDP + 16 * sp], &B1[(16 * nt) * LDP + 16 * sp]);
          } else {
#ifdef CHOL_DIRECT00
            if (!(sp == 3 && wave == 0))   // done in front of the barrier, from registers
#endif
            block_update16(tacc[i], &B1[(16 * rg) * LDP + 16 * sp], &B1[(16 * nt) * LDP + 16 * sp]);
          }
        }
      }
      // the deferred columns of the tile to solve take their rank-16 updates here too: during strips 0..2 the waves
      // wait for the producer anyway, and the last strip's share (waves 1-3, 5-7) overlaps with the first pivot
      // block.  Only the left-looking block updates and the solves stay in the pivot-block slots.
      {
#pragma unroll
        for (int i = 0; i < 3; i++)
          if (i < ndef)
            block_update16(dacc[i], &B0[(16 * dg) * LDP + 16 * sp], &B1[(16 * ((wave >= 5) ? i + 1 : wave)) * LDP + 16 * sp]);
      }
    }
  } else {
    __syncthreads();
    STAMP(1);
  }
  auto store_solve = [&](const f64x4& a, int rg, int nt) {
#pragma unroll
    for (int i = 0; i < 4; i++) {  // tile to solve; the diagonal workgroup keeps only the rows below the tile
      const int li = 16 * rg + fg + 4 * i, lj = 16 * nt + fr;
      double val = a[i];
      if (diag && !(li >= wk && r0 + li < nrows && lj < wk)) val = 0.0;
      BT[li * LDP + lj] = val;
    }
  };
  if (wave == 0) __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int i = 0; i < 2; i++) {
    if (i < ntile) {
      const int rg = (code[i] >> 2) & 3, nt = code[i] & 3;
      const bool isT = (code[i] & 0x10) != 0;
      if (isT) {
        if (solve_rows) {
          if (!PERSIST && k >= 0) strip_update_tile(tacc[i], B0, B1, rg, nt);
          store_solve(tacc[i], rg, nt);
        }
      } else {
        if (!PERSIST && k >= 0) strip_update_tile(tacc[i], B1, B1, rg, nt);
#pragma unroll
        for (int e = 0; e < 4; e++) {  // diagonal tile: strict upper part 0, identity beyond wk
          const int li = 16 * rg + fg + 4 * e, lj = 16 * nt + fr;
          double val = tacc[i][e];
          if (lj > li) val = 0.0;
          else if (li >= wk || lj >= wk) val = (li == lj) ? 1.0 : 0.0;
          B2[li * LDP + lj] = val;
        }
      }
    }
  }
  STAMP(2);
  // block (0,0) was updated and written by wave 0 alone: factor it right away
  if (wave == 0) wave_potrf16(&B2[0], &Wl[0], Idn, fail, diag);
  STAMP(3);
  __syncthreads();
  STAMP(4);

#pragma unroll 1  // one copy of the step body: iterations 1..3 then run from a warm instruction cache
  for (int p = 0; p < 4; p++) {
    {  // 16x16 triangular solves as GEMMs with the block inverse: L_qp = D_qp W^T, X_gp = T_gp W^T
      const int nd = 3 - p;  // diagonal-tile blocks below the pivot block
      if (wave < nd) {
        const int q = p + 1 + wave;
#ifdef CHOL_INL_EXP
        wave_gemm_nt16_inl<true>(&B2[(16 * q) * LDP + 16 * p], &B2[(16 * q) * LDP + 16 * p], &Wl[WLB * p], WLP);
#else
        wave_gemm_nt16<true>(&B2[(16 * q) * LDP + 16 * p], &B2[(16 * q) * LDP + 16 * p], &Wl[WLB * p], WLP);
#endif
      } else if (solve_rows && wave < nd + 4) {
        const int g = wave - nd;
        wave_gemm_nt16<true>(&BT[(16 * g) * LDP + 16 * p], &BT[(16 * g) * LDP + 16 * p], &Wl[WLB * p], WLP);
      }
    }
    if (p == 0) STAMP(10);
    __syncthreads();
    if (p == 0) STAMP(5);
    if (PERSIST && solve_rows && wave == 4)
      chol_strip_out(S, ld, Ldiag + chol_lfin_offset(n) + (size_t)chol_tile_index((nrows + NB - 1) / NB, bi, kp) * NB * NB, BT,
                     r0, c0, wk, nrows, p, (p == 3) ? &done[bi] : nullptr, 4 * kp + 3);
    if (p == 3) break;
    const int q = p + 1;
    if (wave == 0) {  // the pivot chain: next diagonal block, then its factorisation
      wave_gemm_nt16<false>(&B2[(16 * q) * LDP + 16 * q], &B2[(16 * q) * LDP + 16 * p],
                            &B2[(16 * q) * LDP + 16 * p], LDP);
      wave_potrf16(&B2[(16 * q) * LDP + 16 * q], &Wl[WLB * q], Idn, fail, diag);
      if (p == 0) STAMP(6);
    } else {
      // the other rank-16 updates of the diagonal tile, round-robin over the less loaded waves (not wave 4: it
      // shares the pivot wave's SIMD and writes the strips out)
      int cnt = 0;
#pragma unroll 1
      for (int r = q; r < 4; r++)
#pragma unroll 1
        for (int s2 = q; s2 <= r; s2++) {
          if (r == q && s2 == q) continue;  // wave 0's block
          {  // waves 1..3 except wave q, which finishes T(0,q) in this slot
            const int slot = cnt % 2;
            const int owner = 1 + slot + ((1 + slot >= q) ? 1 : 0);  // the two of waves 1..3 that are not wave q
            if (owner == wave)
              wave_gemm_nt16<false>(&B2[(16 * r) * LDP + 16 * s2], &B2[(16 * r) * LDP + 16 * p],
                                    &B2[(16 * s2) * LDP + 16 * p], LDP);
          }
          cnt++;
        }
      // column q of the tile to solve, left-looking: the rank-64 update by the previous panel and
      // the rank-16 updates by the columns solved so far, then into LDS for the next solve
      const bool mine = (wave >= 5) ? (ndef > 0) : (ndef > 0 && wave == q);
      if (mine) {
        f64x4 acc = dacc[0];
        if (!PERSIST && k >= 0) strip_update_tile(acc, B0, B1, dg, q);
#pragma unroll
        for (int pp = 0; pp < 3; pp++)  // unrolled: the operand reads of all blocks are in flight together
          if (pp <= p) block_update16(acc, &BT[(16 * dg) * LDP + 16 * pp], &B2[(16 * q) * LDP + 16 * pp]);
        store_solve(acc, dg, q);
        dacc[0] = dacc[1];
        dacc[1] = dacc[2];
      }
    }
    __syncthreads();
    if (p == 0) STAMP(7);
  }
  STAMP(8);
  if (PERSIST && bi == kp + 1 && kp + 1 < (n + NB - 1) / NB) {
    // The tile right below the diagonal also leaves M = X L_kk^-1 (X = the solved tile L[kp+1,kp]) for the
    // back-substitution, which then multiplies by M^T instead of solving with L_kk^T on its critical chain:
    // M_q = (X_q - sum_{s>q} M_s L_sq) L_qq^-1, 16-column blocks right to left, in place; each wave owns 16 rows,
    // so the waves do not have to synchronise.  Off the factorisation's chain (the strips are out already).
    __syncthreads();  // wave 4 has read the last strip out of BT
    if (wave < 4) {
      const int g = wave;
#pragma unroll
      for (int q = 3; q >= 0; q--) {
        lds_f64* Cq = &BT[(16 * g) * LDP + 16 * q];
        f64x4 acc;
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = Cq[(fg + 4 * i) * LDP + fr];
#pragma unroll
        for (int s2 = q + 1; s2 < 4; s2++)  // acc -= M_s L_sq (B operand read down the columns of the L block)
#pragma unroll
          for (int kk = 0; kk < 16; kk += 4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-BT[(16 * g + fr) * LDP + 16 * s2 + kk + fg],
                                                       B2[(16 * s2 + kk + fg) * LDP + 16 * q + fr], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) Cq[(fg + 4 * i) * LDP + fr] = acc[i];  // back through LDS into the A-operand layout
        f64x4 m = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4)  // M_q = acc * L_qq^-1
          m = __builtin_amdgcn_mfma_f64_16x16x4f64(Cq[fr * LDP + kk + fg], Wl[WLB * q + (kk + fg) * WLP + fr], m, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) Cq[(fg + 4 * i) * LDP + fr] = m[i];
      }
      gbl_f64* Mj = Ldiag + chol_mbuf_offset(n) + (size_t)kp * NB * NB;
#pragma unroll
      for (int it = 0; it < 8; it++) {  // 16 rows x 64 columns of this wave; rows at or beyond n (the rhs row) are zero
        const int u = it * 64 + lane;
        const int i = 16 * g + (u >> 5), j2 = (u & 31) * 2;
        f64x2 v = *(const lds_f64x2*)&BT[i * LDP + j2];
        if (r0 + i >= n) v = (f64x2){0.0, 0.0};
        *(gbl_f64x2*)&Mj[i * NB + j2] = v;
      }
    }
  }
  if (PERSIST && diag) {
    // the inverses of the four 16x16 diagonal blocks ride along in the unused upper blocks (0,1) (0,2) (0,3)
    // (1,2) of the side-buffer tile: the backward substitution solves each 16-unknown block with them as one
    // mat-vec instead of sixteen dependent steps
#pragma unroll
    for (int e = t; e < 1024; e += 512) {
      const int pb = e >> 8, j = (e >> 4) & 15, kk = e & 15;
      const int br = pb == 3 ? 1 : 0, bc = pb == 3 ? 2 : pb + 1;
      B2[(16 * br + j) * LDP + 16 * bc + kk] = Wl[WLB * pb + WLP * j + kk];
    }
    __syncthreads();
  }
  if (grp == 0) {
    if (diag) {
      if (PERSIST) store_tile64(Ldiag + (size_t)kp * NB * NB, B2, NB, 0, 0, 0, NB, NB, false);
      else store_tile64(S, B2, ld, c0, c0, c0, c0 + wk, wk, true);
    }
  } else if (solve_rows && !PERSIST) {  // (single launch: the strips went out one by one)
    store_tile64(S, BT, ld, r0, c0, max(r0, c0 + wk), min(r0 + NB, nrows), wk, false);
  }
  if (!PERSIST) STAMP(9);
}

__global__ __launch_bounds__(512) void chol_step_kernel(double* __restrict__ S, int n, int ld, int k,
                                                        int* __restrict__ fail, double lm, double ep) {
  chol_tile<false>((gbl_f64*)S, n, ld, k, k + 1 + (int)blockIdx.x, k + 1 + (int)blockIdx.y, fail, lm, ep, nullptr,
                   nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------
// The whole factorisation in ONE launch.  A kernel boundary per block column costs 3-4.5 us of
// the 17 us step and makes the pivot chain wait for the slowest trailing tile; here a grid of
// co-resident workgroups (one per CU: 145 KB of LDS) walks the steps itself.
//   * static ownership: lower-triangle tiles in column-major order, tile idx -> workgroup
//     idx % grid; a workgroup applies every step to its own tiles, so a tile's intermediate
//     versions never change hands (and panel tiles come first in every workgroup's step);
//   * data flow instead of barriers: done[i] = 4*c + 3 once the final tile L[i, c] of block row i is completely
//     in the matrix, dver[j] = trailing updates applied to the diagonal tile (j, j) by its owner (0 = damped).
//     A trailing tile (bi, bj) at step k waits for done[bi], done[bj] >= 4k+3; a panel tile waits for
//     dver[k+1] >= k and then takes the two panel tiles strip by strip out of their data-tagged hand-over slots,
//     as they are written; so do the trailing tiles of the next panel column (no flag at all);
//     nothing waits for unrelated trailing tiles (look-ahead comes for free);
//   * hand-off without cache maintenance: every tile store is written through (sc1), tiles of other
//     workgroups are read with sc1 loads, the flags likewise: store, s_waitcnt vmcnt(0), barrier, flag |
//     flag poll, barrier, loads.  Rows are 128-byte aligned (ld % 16 == 0) so tiles of different
//     workgroups never share a cache line of the per-XCD L2s;
//   * the factored diagonal tile L_jj goes to Ldiag[j] (64x64): the other panel workgroups of the
//     column read the unfactored tile at their own pace;
//   * every spin is bounded: a stalled grid raises `abort`, reports a failed factorisation and
//     drains (it cannot happen with a resident grid; the GPU is never left hanging).

// One tile of one step.  NOT inlined into the step loops: inlined, the loop-invariant lane offsets of the
// whole body are hoisted in front of the loops and held (or spilled) across them.
template <bool OV>
__device__ __attribute__((noinline)) void chol_tile_persist(double* __restrict__ S, int n, int ld, int k, int bi,
                                                            int bj, int* __restrict__ fail, double lm, double ep,
                                                            double* __restrict__ Ldiag, int* __restrict__ done,
                                                            int* __restrict__ abortf, bool coh_first, int k_last) {
  chol_tile<true, OV>((gbl_f64*)S, n, ld, k, bi, bj, fail, lm, ep, (gbl_f64*)Ldiag, done, abortf, coh_first, k_last);
}

// OVERLAP mode (ready != nullptr; multi-GPU, SURVEY 8e / VERDICT r02 #6): the kernel is launched BEFORE the reduced
// system is there.  The all-reduce of the packed system (block-column major) runs in column chunks on a side stream;
// each chunk is expanded into S (damping applied there: pass lm = ep = 0 here) and then ready[block column] = epoch
// is published for its block columns.  A tile (bi, bj) is first read by its owner at its first step: that read waits
// for ready[bj] and goes past the L2 (sc1); a plain trailing tile then replays all the updates it has missed in one
// go (chol_tile, k_last).  Why columns: row i of L needs every earlier column, so a late block ROW costs a sweep as
// long as the factorisation itself (measured: profiles/r03_overlap_emulation.txt), while a late block COLUMN j is not
// needed before the chain gets there, ~10.8 us x j after the start.
// The grid is capped below the CU count by the launcher, so the collective's and the unpack kernels always find CUs.
#ifdef OV_DEBUG
__device__ unsigned long long g_ov_dbg[64];
extern "C" int droid_debug_overlap_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ov_dbg), sizeof(unsigned long long) * 64);
}
#endif
template <bool OV>
__global__ __launch_bounds__(512) void chol_factor_persistent_kernel(double* __restrict__ S, int n, int ld,
                                                                     int* __restrict__ fail, double lm, double ep,
                                                                     int* __restrict__ flags,
                                                                     double* __restrict__ Ldiag,
                                                                     const int* __restrict__ ready_arg, int epoch) {
  const int* const ready = OV ? ready_arg : nullptr;   // (OV = false: every overlap branch below is compiled out)
  __shared__ int s_abort;
  const int nb = (n + NB - 1) / NB;       // block columns
  const int nrb = (n + 1 + NB - 1) / NB;  // block rows (row n = rhs)
  int* done = flags;
  int* dver = flags + nrb;
  int* abortf = flags + 2 * nrb;
  const int G = (int)gridDim.x, wg = (int)blockIdx.x, t = (int)threadIdx.x;
  int total = 0;
  for (int j = 0; j < nb; j++) total += nrb - j;

  {  // damping diag += ep + lm*diag (dk:1197) of the blocks >= 1 by the owners of their diagonal tiles
    int cs = 0;
    for (int j = 0; j < nb; j++) {
      if (j >= 1 && (cs % G) == wg) {
        const int i = j * NB + t;
        if (ready == nullptr && t < NB && i < n) {   // (overlap mode: the rows are not there yet; damped by the unpack)
          const double d = S[(size_t)i * ld + i];
          gstore<true>((gbl_f64*)&S[(size_t)i * ld + i], d + (ep + lm * d));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have been acknowledged
        __syncthreads();
        if (t == 0) cfp_store(&dver[j], 0);
      }
      cs += nrb - j;
    }
  }

#ifdef OV_DEBUG
  if (blockIdx.x == 0 && threadIdx.x == 0) g_ov_dbg[63] = wall_clock64();
#endif
  if (OV && ready != nullptr) {
    // ---- overlap mode: per-tile progress instead of the step-major walk below.  A workgroup owns up to a handful of
    // tiles; stepping them in lockstep would park it at step 0 on a tile whose block column arrives with the LAST chunk
    // of the collective while its other tile sits on the chain.  Here every owned tile keeps its own next step; wave 0
    // scans them in ownership order (earlier block columns first), takes the first whose inputs are there -- the same
    // flags as below, polled once instead of waited for, plus ready[bi] at the tile's first step -- and spins
    // (bounded) only when none is.  Dependencies only point to earlier block columns, so skipping a waiting tile can
    // never starve the tile that is picked.
    constexpr int MAXT = 12;   // the launcher admits at most 10 tiles per workgroup
    __shared__ int s_ti[MAXT], s_tj[MAXT], s_nx[MAXT], s_nt, s_pick, s_klast;
    if (t == 0) {
      int nt = 0, cj = 0, cs = 0;
      for (int idx = wg; idx < total && nt < MAXT; idx += G) {
        while (idx >= cs + (nrb - cj)) {
          cs += nrb - cj;
          cj++;
        }
        s_ti[nt] = cj + (idx - cs);
        s_tj[nt] = cj;
        s_nx[nt] = (cj == 0) ? -1 : 0;     // column 0 has the initial panel step only
        nt++;
      }
      s_nt = nt;
    }
    __syncthreads();
    while (true) {
      if (t < 64) {
        int pick = -1, spins = 0, klast = -2;
        bool ok = true;
        const int nt = s_nt;
        while (true) {
          bool any_left = false;
          for (int i = 0; i < nt; i++) {
            const int bi = s_ti[i], bj = s_tj[i], k = s_nx[i];
            if (k > bj - 1) continue;      // this tile is final (last step: its panel step k = bj - 1)
            any_left = true;
            const int kp = k + 1;
            const bool panel = (bj == kp);
            const bool first = (k == ((bj == 0) ? -1 : 0));
            const int* p = nullptr;
            int need = 4 * k + 3;
            const bool flagged = k >= 0 && !panel && bj != kp + 1;
            if (t == 0 && flagged) p = &done[bi];
            else if (t == 1 && flagged && bj != bi) p = &done[bj];
            else if (t == 2 && k >= 0 && panel && bi != bj) { p = &dver[bj]; need = k; }
            else if (t == 3 && first) p = &ready[bj], need = epoch;    // the tile's block column has been reduced
            const int fv = (p == nullptr) ? 0x7fffffff : cfp_load(p);
            bool sat = fv >= need;
            // Panel tiles and tiles of the next panel column take their two panel tiles strip by strip out of the
            // hand-over slots INSIDE the body, and wait there: enter only once the producers have started writing
            // (first strip of each slot no longer carries the preset tag) -- a producer in a block row that comes
            // with a late chunk would otherwise park this workgroup, and its other tiles with it.
            if ((t == 4 || t == 5) && k >= 0 && (panel || bj == kp + 1)) {
              const int r = (t == 4) ? bi : (panel ? kp : bj);
              const gbl_f64* sl = (const gbl_f64*)Ldiag + chol_lfin_offset(n) + (size_t)chol_tile_index(nrb, r, k) * NB * NB;
              sat = __double_as_longlong(gload<true>(sl)) != CFP_TAG;
            }
            if (__all(sat)) {
              pick = i;
              klast = k;
              if (flagged) {  // plain trailing tile: how far are BOTH rows final?  done = 4 c + 3 for the last final column c
                const int d0 = __shfl(fv, 0), d1 = (bj != bi) ? __shfl(fv, 1) : d0;
                const int cok = (min(d0, d1) - 3) >> 2;
                klast = max(k, min(cok, bj - 3));   // k = bj - 2 takes the strips from the slots, k = bj - 1 is the panel step
              }
              break;
            }
          }
          if (!any_left) pick = -2;
          if (pick != -1) break;
          if (++spins > (CFP_SPIN_LIMIT << 2) || ((spins & 63) == 0 && __any(cfp_load(abortf) == 1))) {
            ok = false;
            break;
          }
          __builtin_amdgcn_s_sleep(4);
        }
        if (t == 0) {
          s_pick = ok ? pick : -3;
          s_klast = klast;
          if (!ok) {
            cfp_store(abortf, 1);
            atomicMax(fail, 2);
          }
        }
      }
      __syncthreads();  // also: the previous tile's LDS reads are over
      const int pick = s_pick;
      if (pick < 0) return;   // -2: every owned tile is final; -3: stalled (reported)
      const int bi = __builtin_amdgcn_readfirstlane(s_ti[pick]), bj = __builtin_amdgcn_readfirstlane(s_tj[pick]);
      const int k = __builtin_amdgcn_readfirstlane(s_nx[pick]);
      const bool first = (k == ((bj == 0) ? -1 : 0));
      const int klast = __builtin_amdgcn_readfirstlane(s_klast);
      chol_tile_persist<true>(S, n, ld, k, bi, bj, fail, lm, ep, Ldiag, done, abortf, first, klast);
#ifdef OV_DEBUG
      if (t == 0 && bi == bj && bj == k + 1 && bj < 62) g_ov_dbg[bj] = wall_clock64();   // diagonal tile of column bj factored
#endif
      if (bj != k + 1 && bi == bj) {  // publish the next version of a diagonal tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) cfp_store(&dver[bj], max(k, klast) + 1);
      }
      __syncthreads();
      if (t == 0) s_nx[pick] = max(k, klast) + 1;
      __syncthreads();
    }
  }

  int colstart = 0;  // index of tile (kp, kp)
#pragma unroll 1
  for (int k = -1; k + 1 < nb; k++) {
    const int kp = k + 1;
    int idx = colstart + (((wg - colstart) % G) + G) % G;
    int cj = kp, cs = colstart;
#pragma unroll 1
    for (; idx < total; idx += G) {
      while (idx >= cs + (nrb - cj)) {
        cs += nrb - cj;
        cj++;
      }
      const int bj = __builtin_amdgcn_readfirstlane(cj), bi = __builtin_amdgcn_readfirstlane(cj + (idx - cs));
      if (k < 0 && bj != 0) break;  // step -1 is the first panel only
      const bool panel = (bj == kp);
      // ---- wait for the inputs (wave 0: one flag per lane)
      PSTAMP(11);
      const bool first_touch = (ready != nullptr) && (k <= 0);   // step -1: column 0; step 0: every other tile
      if (t < 64) {
        bool ok = true;
        if (k >= 0 || first_touch) {
          // plain tile: both panel tiles complete (all four strips); panel tile: only the diagonal tile's
          // version here, the strips are awaited inside the body
          const int* p = nullptr;
          int need = 4 * k + 3;
          const bool flagged = k >= 0 && !panel && bj != kp + 1;  // tiles of the next panel column poll the hand-over slots
          if (t == 0 && flagged) p = &done[bi];
          else if (t == 1 && flagged && bj != bi) p = &done[bj];
          else if (t == 2 && k >= 0 && panel && bi != bj) { p = &dver[bj]; need = k; }
          else if (t == 3 && first_touch) { p = &ready[bj]; need = epoch; }   // the tile's block column has been reduced
          bool sat = (p == nullptr);
          int spins = 0;
          // overlap mode waits for the collective (and, when ranks share a GPU, for another rank's grid): seconds, not ms
          const int limit = (ready != nullptr) ? (CFP_SPIN_LIMIT << 4) : CFP_SPIN_LIMIT;
          while (true) {
            if (!sat) sat = cfp_load(p) >= need;
            if (__all(sat)) break;
            spins++;
            if (spins > limit || ((spins & 255) == 0 && __any(cfp_load(abortf) == 1))) {
              ok = false;
              break;
            }
            if (ready != nullptr) __builtin_amdgcn_s_sleep(4);
            else __builtin_amdgcn_s_sleep(1);
          }
        } else if (__any(cfp_load(abortf) == 1)) {
          ok = false;
        }
        if (t == 0) {
          s_abort = ok ? 0 : 1;
          if (!ok) {
            cfp_store(abortf, 1);
            atomicMax(fail, 2);
          }
        }
      }
      PSTAMP(12);
      __syncthreads();  // also: the previous tile's LDS reads are over
      if (s_abort) return;
      chol_tile_persist<false>(S, n, ld, k, bi, bj, fail, lm, ep, Ldiag, done, abortf, false, k);
      if (!panel && bi == bj) {  // publish the next version of a diagonal tile (panel tiles publish their strips)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave's write-through stores are acknowledged
        __syncthreads();
        if (t == 0) cfp_store(&dver[bj], k + 1);
      }
      PSTAMP(13);
    }
    colstart += nrb - kp;
  }
}

// One block column of the backward substitution L^T x = y (y = row n of S, consumed in place):
// x_k = L_kk^-T y_k, then y[0:c0] -= L[k rows, 0:c0]^T x_k.
__global__ __launch_bounds__(256) void chol_backsolve_kernel(double* __restrict__ S, int n, int ld,
                                                             int k, double* __restrict__ x) {
  __shared__ double L[NB * LDP];
  __shared__ double xk[NB];
  const int t = threadIdx.x;
  const int c0 = k * NB;
  const int wk = min(NB, n - c0);
  load_tile64(DROID_LDS(L), (const gbl_f64*)S, ld, c0, c0, c0, c0 + wk, c0 + wk, true, 1.0);
  __syncthreads();
  if (t < 64) {
    // lane j owns z_j; column-oriented back substitution with static lane broadcasts
    double z = (t < wk) ? S[(size_t)n * ld + c0 + t] : 0.0;
    const double rinv = 1.0 / L[t * LDP + t];
#pragma unroll
    for (int i = NB - 1; i >= 0; i--) {
      const double xi = readlane_f64(z * rinv, i);
      z = (t == i) ? xi : ((t < i) ? fma(-L[i * LDP + t], xi, z) : z);
    }
    xk[t] = z;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (t < wk) x[c0 + t] = xk[t];
    return;
  }
  const int c = (blockIdx.x - 1) * 256 + t;
  if (c < c0) {
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    const double* col = S + (size_t)c0 * ld + c;
#pragma unroll
    for (int r = 0; r < NB; r += 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = (r + u < wk) ? col[(size_t)(r + u) * ld] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; u++) s4[u & 3] = fma(v[u], xk[r + u], s4[u & 3]);
    }
    S[(size_t)n * ld + c] -= (s4[0] + s4[1]) + (s4[2] + s4[3]);
  }
}

// Backward substitution as ONE launch: workgroup j owns block column j and consumes the
// solutions x_k (k > j) of the workgroups to its right in exactly the order they are produced.
// Wave 0 carries the dependent chain alone -- poll x_k, 64x64 mat-vec from LDS, finally the
// 64-step solve -- while waves 1-3 prefetch the next off-diagonal tile L[k-1,j] into the other
// LDS buffer; one workgroup barrier per block hands the tile over.
// Hand-off: x itself is the flag.  x is pre-filled with an all-ones NaN pattern; the producer
// publishes each x_j element with ONE 8-byte agent-scope (sc1) store and the consumer lanes poll
// their own element with 8-byte agent-scope loads until it changes (single-copy atomic, so no
// separate flag, fence or drain is needed; MI355X guide: data-tagged 8-byte granules).  The grid
// is co-resident (<= BSP_MAX_BLOCKS workgroups, one per CU) and every spin is bounded.
constexpr int BSP_MAX_BLOCKS = 200;
#ifdef CHOL_STAMPS
__device__ unsigned long long g_bs_stamps[64 * 8];
#define BSTAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 64) g_bs_stamps[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define BSTAMP(slot) do { } while (0)
#endif
constexpr long long BSP_SENTINEL = -1LL;

// 64x64 tile -> LDS by the 192 threads of waves 1..3 (11 independent 16-byte loads per thread)
__device__ __forceinline__ void load_tile64_w123(lds_f64* __restrict__ dst, const gbl_f64* __restrict__ S,
                                                 int ld, int r0, int c0, int row_end, int col_end) {
  const int t = threadIdx.x - 64;
  f64x2 v[11];
#pragma unroll
  for (int it = 0; it < 11; it++) {
    const int idx2 = it * 192 + t;
    const int i = idx2 >> 5, j = (idx2 & 31) * 2;
    const bool ok = (idx2 < 2048) && (r0 + i < row_end) && (c0 + j < col_end);
    const gbl_f64* p = S + (size_t)(ok ? r0 + i : 0) * ld + (ok ? c0 + j : 0);
    v[it] = *(const gbl_f64x2*)p;
    if (!ok) v[it] = (f64x2){0.0, 0.0};
    else if (c0 + j + 1 >= col_end) v[it][1] = 0.0;
  }
#pragma unroll
  for (int it = 0; it < 11; it++) {
    const int idx2 = it * 192 + t;
    if (idx2 < 2048) {
      const int i = idx2 >> 5, j = (idx2 & 31) * 2;
      *(lds_f64x2*)(&dst[i * LDP + j]) = v[it];
    }
  }
}

// FAST: the factorisation ran as one launch and left L_jj plus the inverses of its 16x16 diagonal blocks in Ldiag.
template <bool FAST>
__global__ __launch_bounds__(256) void chol_backsolve_persistent_kernel(
    double* __restrict__ S, int n, int ld, double* __restrict__ x, int* __restrict__ err,
    const double* __restrict__ Ldiag) {
  __shared__ double Ld[NB * LDP];
  __shared__ double Tt[2][NB * LDP];
  __shared__ double xk[NB];
  __shared__ int dead;
  const int t = threadIdx.x;
  const int nb = gridDim.x;
  const int j = blockIdx.x;
  const int c0 = j * NB;
  const int wj = min(NB, n - c0);
  BSTAMP(0);
  if (t == 0) dead = 0;
  if (FAST) load_tile64(DROID_LDS(Ld), (const gbl_f64*)Ldiag + (size_t)j * NB * NB, NB, 0, 0, 0, wj, wj, true, 1.0);
  else load_tile64(DROID_LDS(Ld), (const gbl_f64*)S, ld, c0, c0, c0, c0 + wj, c0 + wj, true, 1.0);
  double z = (t < wj) ? S[(size_t)n * ld + c0 + t] : 0.0;  // wave 0 owns z
  int cur = 0;
  // tile of block row k in this block column; FAST: for k = j+1 the factorisation left M = L[j+1,j] L_jj^-1, which
  // is applied AFTER the diagonal solve (x_j = L_jj^-T z' - M^T x_{j+1}): the solve runs while x_{j+1} is still on
  // its way and only one mat-vec follows its arrival
  auto load_k = [&](double* dst, int k) {
    if (FAST && k == j + 1)
      load_tile64_w123(DROID_LDS(dst), (const gbl_f64*)Ldiag + chol_mbuf_offset(n) + (size_t)j * NB * NB, NB, 0, 0, NB, NB);
    else
      load_tile64_w123(DROID_LDS(dst), (const gbl_f64*)S, ld, k * NB, c0, n, c0 + wj);
  };
  if (j < nb - 1 && t >= 64) load_k(Tt[0], nb - 1);
  __syncthreads();
  double Lcol[NB];  // wave 0: column t of the diagonal tile (rows above the diagonal read 0)
  double Wcol[16];  // single-launch factor: column (t & 15) of the inverse of this lane's 16x16 diagonal block
  if (t < 64) {
#pragma unroll
    for (int i = FAST ? 16 : 0; i < NB; i++) Lcol[i] = Ld[i * LDP + t];  // FAST: only the updates of the rows above
    if (FAST) {
      const int pb = t >> 4, br = pb == 3 ? 1 : 0, bc = pb == 3 ? 2 : pb + 1;
      const double* Wg = Ldiag + (size_t)j * NB * NB + (size_t)(16 * br) * NB + 16 * bc + (t & 15);
#pragma unroll
      for (int c = 0; c < 16; c++) Wcol[c] = Wg[c * NB];  // (L_pp^-1)[c][t & 15]
    }
  }
  // poll x_k element-wise (lanes beyond the matrix take 0) and return  sum_r Tc[r] x_k[r]  for this lane's column
  auto poll_matvec = [&](int k, const double (&Tc)[NB]) -> double {
    const int gi = k * NB + t;
    double xv = 0.0;
    if (gi < n) {
      int spins = 0;
      while (true) {
        xv = __hip_atomic_load(&x[gi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__double_as_longlong(xv) != BSP_SENTINEL) break;
        if (++spins > (1 << 22)) {  // cannot happen with a resident grid; never hang the GPU
          dead = 1;
          atomicMax(err, 2);
          xv = 0.0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (k == j + 1) BSTAMP(1);
    xk[t] = xv;  // same wave writes and reads: LDS operations of one wave are ordered
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int r = 0; r < NB; r += 4) {  // x_k in 16-byte broadcast reads against the preloaded column
      const f64x2 x01 = *(const lds_f64x2*)&DROID_LDS(xk)[r], x23 = *(const lds_f64x2*)&DROID_LDS(xk)[r + 2];
      a0 = fma(Tc[r + 0], x01[0], a0);
      a1 = fma(Tc[r + 1], x01[1], a1);
      a2 = fma(Tc[r + 2], x23[0], a2);
      a3 = fma(Tc[r + 3], x23[1], a3);
    }
    return (a0 + a1) + (a2 + a3);
  };
  const int klast = FAST ? j + 2 : j + 1;  // FAST: the tile of block row j+1 (as M) comes after the diagonal solve
  for (int k = nb - 1; k >= klast; k--) {
    if (t >= 64) {
      if (k - 1 > j) load_k(Tt[cur ^ 1], k - 1);
    } else {
      // column t of the tile (it landed before the last barrier) goes to registers BEFORE the poll: once x_k is
      // there, the mat-vec is 32 broadcast reads and 64 FMAs (it was a rolled loop paying the LDS latency 16 times)
      double Tc[NB];
      {
        const double* T = Tt[cur];
#pragma unroll
        for (int r = 0; r < NB; r++) Tc[r] = T[r * LDP + t];
      }
      z -= poll_matvec(k, Tc);
    }
    cur ^= 1;
    if (!FAST && k == j + 1) {
      BSTAMP(2);
      break;  // nothing left to prefetch: wave 0 goes straight into the diagonal solve
    }
    __syncthreads();  // tile k-1 landed, tile k consumed
    if (dead) return;
  }
  BSTAMP(3);
  if (t < 64) {
    // L_jj^T x = z by one wave, lane t = unknown t.  Measured: the plain 64-step lane-broadcast
    // loop cost 150 cycles per step (an LDS read of L[i][t] inside every dependent step).  Here
    // column t of L_jj sits in registers (loaded before the first poll), each 16-unknown block is
    // solved inside its own 16-lane row with DPP broadcasts, and the rows above receive the block
    // through one 16-term update.
    double xsol = 0.0;
    double Mc[NB];  // FAST: column t of M, in registers before the solve so that nothing but the poll is left after it
    if (FAST && j < nb - 1) {
      const double* T = Tt[cur];
#pragma unroll
      for (int r = 0; r < NB; r++) Mc[r] = T[r * LDP + t];
    }
    const double rinv = FAST ? 0.0 : 1.0 / Ld[t * LDP + t];
    const int rr = t & 15, rowb = t >> 4;
#define DROID_BS_STEP(B, II)                                                                            \
    {                                                                                                  \
      double zr = z * rinv, xi;                                                                        \
      asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" #II " row_mask:0xf bank_mask:0xf"   \
                   : "=v"(xi) : "v"(zr));                                                              \
      z = fma(-Lcol[16 * B + II], xi, z);  /* zero for lanes above the diagonal; lane II itself is done */ \
      xsol = (rr == II) ? xi : xsol;                                                                   \
    }
// with the block inverse: x_r = sum_c (L_pp^-1)[c][r] z_c, sixteen independent DPP-broadcast FMAs
#define DROID_BS_WTERM(ACC, C)                                                                          \
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #C " row_mask:0xf bank_mask:0xf"             \
                 : "+v"(ACC) : "v"(z), "v"(Wcol[C]));
#define DROID_BS_BLOCK(B)                                                                               \
    if (rowb == B) {                                                                                   \
      if (FAST) {                                                                                      \
        double xw = 0.0, xv2 = 0.0;                                                                    \
        asm volatile("s_nop 1" : "+v"(z));                                                             \
        DROID_BS_WTERM(xw, 0) DROID_BS_WTERM(xv2, 1) DROID_BS_WTERM(xw, 2) DROID_BS_WTERM(xv2, 3)      \
        DROID_BS_WTERM(xw, 4) DROID_BS_WTERM(xv2, 5) DROID_BS_WTERM(xw, 6) DROID_BS_WTERM(xv2, 7)      \
        DROID_BS_WTERM(xw, 8) DROID_BS_WTERM(xv2, 9) DROID_BS_WTERM(xw, 10) DROID_BS_WTERM(xv2, 11)    \
        DROID_BS_WTERM(xw, 12) DROID_BS_WTERM(xv2, 13) DROID_BS_WTERM(xw, 14) DROID_BS_WTERM(xv2, 15)  \
        xsol = xw + xv2;                                                                               \
      } else {                                                                                         \
        DROID_BS_STEP(B, 15) DROID_BS_STEP(B, 14) DROID_BS_STEP(B, 13) DROID_BS_STEP(B, 12)            \
        DROID_BS_STEP(B, 11) DROID_BS_STEP(B, 10) DROID_BS_STEP(B, 9) DROID_BS_STEP(B, 8)              \
        DROID_BS_STEP(B, 7) DROID_BS_STEP(B, 6) DROID_BS_STEP(B, 5) DROID_BS_STEP(B, 4)                \
        DROID_BS_STEP(B, 3) DROID_BS_STEP(B, 2) DROID_BS_STEP(B, 1) DROID_BS_STEP(B, 0)                \
      }                                                                                                \
      xk[t] = xsol;                                                                                    \
    }                                                                                                  \
    if (B > 0 && rowb < B) {                                                                           \
      double u0 = 0.0, u1 = 0.0;                                                                       \
      _Pragma("unroll") for (int ii = 0; ii < 16; ii += 2) {                                           \
        const f64x2 xx = *(const lds_f64x2*)&DROID_LDS(xk)[16 * B + ii];                                \
        u0 = fma(Lcol[16 * B + ii], xx[0], u0);                                                        \
        u1 = fma(Lcol[16 * B + ii + 1], xx[1], u1);                                                    \
      }                                                                                                \
      z -= u0 + u1;                                                                                    \
    }
    DROID_BS_BLOCK(3)
    DROID_BS_BLOCK(2)
    DROID_BS_BLOCK(1)
    DROID_BS_BLOCK(0)
#undef DROID_BS_BLOCK
#undef DROID_BS_WTERM
#undef DROID_BS_STEP
    BSTAMP(2);
    if (FAST && j < nb - 1) xsol -= poll_matvec(j + 1, Mc);  // x_j = L_jj^-T z' - M^T x_{j+1}
    if (__double_as_longlong(xsol) == BSP_SENTINEL) xsol = __longlong_as_double(0x7ff8000000000000LL);
    BSTAMP(4);
    if (t < wj) __hip_atomic_store(&x[c0 + t], xsol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    BSTAMP(5);
  }
}

// Single-launch path: needs the hand-off flags, the side buffer for the factored diagonal tiles,
// 128-byte aligned rows (no cache line shared between tiles of different workgroups) and a grid
// that is resident as a whole.
static int chol_resident_workgroups() {
  static int cached = -1;
  if (cached < 0) {
    int dev = 0, occ = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, chol_factor_persistent_kernel<false>, 512, 0) != hipSuccess)
      cached = 0;
    else
      cached = occ * prop.multiProcessorCount;
  }
  return cached;
}

static bool chol_single_launch(const double* sys, int n, int ld, const int* flags, const double* ldiag) {
  static const bool off = (getenv("DROID_CHOL_MULTI_LAUNCH") != nullptr);  // diagnostics: the per-step path
  const int nb = (n + NB - 1) / NB;
  // beyond ~10 tiles per resident workgroup (n ~ 4400 on 256 CUs) the trailing updates, serialised inside the
  // persistent workgroups, outweigh the saved kernel boundaries (measured cross-over: n = 4500)
  return !off && flags != nullptr && ldiag != nullptr && nb >= 2 && (ld % 16) == 0 &&
         (reinterpret_cast<uintptr_t>(sys) % 128) == 0 && chol_resident_workgroups() >= 8 &&
         chol_tiles(n) <= (size_t)10 * chol_resident_workgroups();
}

// ---- residency of the single-launch kernels ------------------------------------------------------------------
// Both persistent kernels spin across workgroups, so their whole grid must be resident.  A plain launch gives that
// whenever nothing else holds the CUs for good: ordinary kernels of other streams only delay the dispatch of the
// remaining workgroups.  The one thing that can deadlock is ANOTHER spinning grid:
//   * within a process (two streams / threads calling `ba`): serialised here.  When a persistent launch arrives on
//     a different stream than the previous one, an event recorded on the previous stream is waited for first, so
//     two such grids never overlap.  Single-stream use (the SLAM loop) never records or waits.
//   * across processes sharing one GPU: DROID_CHOL_COOPERATIVE=1 launches through hipLaunchCooperativeKernel,
//     which admits the grid only if it can be resident and runs one cooperative grid at a time (+19 us per launch,
//     tools/micro/coop_launch.hip, which is why it is not the default); if the launch is refused the per-step
//     kernels run instead.  A stall that happens anyway ends in the bounded spins: fail flag = 2, reported as
//     STATUS_CHOL_STALL (an error, unlike the numerical STATUS_CHOL_FAIL whose dx = 0 is the reference's behaviour).
struct PersistState {
  hipStream_t last = nullptr;
  hipEvent_t ev = nullptr;
  bool any = false;
};
static PersistState g_persist[64];
static std::mutex g_persist_mu;

// Returns the lock: the caller keeps it until its spinning grid is ENQUEUED, so that another host thread cannot record
// its "previous stream" event in front of this launch (with the lock released before the launch, thread B
// could order itself behind thread A's stream before A's grid was in it, and the two grids could overlap).
static std::unique_lock<std::mutex> persist_enter(hipStream_t s) {
  std::unique_lock<std::mutex> lock(g_persist_mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return lock;
  PersistState& st = g_persist[dev];
  if (st.any && st.last != s) {
    if (!st.ev && hipEventCreateWithFlags(&st.ev, hipEventDisableTiming) != hipSuccess) st.ev = nullptr;
    if (st.ev && hipEventRecord(st.ev, st.last) == hipSuccess) (void)hipStreamWaitEvent(s, st.ev, 0);
    (void)hipGetLastError();  // a destroyed previous stream must not poison this call
  }
  st.last = s;
  st.any = true;
  return lock;
}

// The single-launch factorisation leaves the factored diagonal tiles L_jj in the side buffer `ldiag`, not in the
// matrix (the other panel workgroups still read the unfactored tile).  The per-step back-substitution reads them from
// the matrix: this copies the lower triangles back when that kernel has to follow a single-launch factorisation
// (cooperative launch of the back-substitution refused).
__global__ __launch_bounds__(256) void chol_ldiag_to_sys_kernel(double* __restrict__ S, int n, int ld,
                                                                const double* __restrict__ Ldiag) {
  const int j = blockIdx.x, c0 = j * NB, wk = min(NB, n - c0);
  for (int idx = threadIdx.x; idx < NB * NB; idx += blockDim.x) {
    const int i = idx / NB, c = idx % NB;
    if (i < wk && c <= i) S[(size_t)(c0 + i) * ld + c0 + c] = Ldiag[(size_t)j * NB * NB + idx];
  }
}

static bool chol_cooperative() {
  static const bool on = (getenv("DROID_CHOL_COOPERATIVE") != nullptr && atoi(getenv("DROID_CHOL_COOPERATIVE")) != 0);
  return on;
}

// returns true when the single-launch kernel ran (the back-substitution may then use what it left in `ldiag`)
// Overlap mode: the single-launch kernel only (returns false when it cannot be used: the caller then unpacks the
// whole system and solves the ordinary way).  `ready` [block columns] / `epoch`: see the kernel.  The grid leaves
// `reserve` CUs free for the collective and the unpack kernels (DROID_OVERLAP_RESERVE_CUS, default 32).
bool launch_chol_factor_overlap(double* sys, int n, int ld, int* fail_flag, int* flags, double* ldiag,
                                const int* ready, int epoch, hipStream_t s) {
  if (n <= 0 || !ready || chol_cooperative() || !chol_single_launch(sys, n, ld, flags, ldiag)) return false;
  const int nb = (n + NB - 1) / NB, nrb = (n + 1 + NB - 1) / NB;
  int total = 0;
  for (int j = 0; j < nb; j++) total += nrb - j;
  static const int reserve = getenv("DROID_OVERLAP_RESERVE_CUS") ? atoi(getenv("DROID_OVERLAP_RESERVE_CUS")) : 32;
  int cap = chol_resident_workgroups() - (reserve > 0 ? reserve : 0);
  if (cap < 8 || (size_t)total > (size_t)10 * cap) return false;
  const int grid = total < cap ? total : cap;
  auto lock = persist_enter(s);
  const double zero = 0.0;
  hipLaunchKernelGGL(chol_factor_persistent_kernel<true>, dim3(grid), dim3(512), 0, s, sys, n, ld, fail_flag, zero, zero,
                     flags, ldiag, ready, epoch);
  return true;
}

bool launch_chol_factor(double* sys, int n, int ld, double lm, double ep, int* fail_flag, int* flags,
                        double* ldiag, hipStream_t s) {
  if (n <= 0) return false;
  const int nb = (n + NB - 1) / NB;        // block columns
  const int nrb = (n + 1 + NB - 1) / NB;   // block rows (row n = rhs)
  if (chol_single_launch(sys, n, ld, flags, ldiag)) {
    int total = 0;
    for (int j = 0; j < nb; j++) total += nrb - j;
    int cap = chol_resident_workgroups();
    static const int env_cap = getenv("DROID_CHOL_GRID") ? atoi(getenv("DROID_CHOL_GRID")) : 0;  // diagnostics
    if (env_cap >= 8 && env_cap < cap) cap = env_cap;
    const int grid = total < cap ? total : cap;
    auto lock = persist_enter(s);
    if (chol_cooperative()) {
      const int* no_ready = nullptr;
      int no_epoch = 0;
      void* args[] = {&sys, &n, &ld, &fail_flag, &lm, &ep, &flags, &ldiag, &no_ready, &no_epoch};
      if (hipLaunchCooperativeKernel((const void*)chol_factor_persistent_kernel<false>, dim3(grid), dim3(512), args, 0, s) == hipSuccess)
        return true;
      (void)hipGetLastError();  // refused (grid cannot be resident now): per-step kernels below
    } else {
      hipLaunchKernelGGL(chol_factor_persistent_kernel<false>, dim3(grid), dim3(512), 0, s, sys, n, ld, fail_flag, lm,
                         ep, flags, ldiag, (const int*)nullptr, 0);
      return true;
    }
  }
  for (int k = -1; k + 1 < nb; k++)         // launch k finishes panel k+1; launch -1 also damps
    hipLaunchKernelGGL(chol_step_kernel, dim3(nrb - k - 1, k < 0 ? 2 : nb - k - 1), dim3(512), 0, s, sys, n,
                       ld, k, fail_flag, lm, ep);
  return false;
}

void launch_chol_backsolve(double* sys, int n, int ld, double* x, int* flags, double* ldiag, int* err,
                           hipStream_t s, bool factor_single) {
  const int nb = (n + NB - 1) / NB;
  if (flags != nullptr && nb >= 2 && nb <= BSP_MAX_BLOCKS) {
    auto lock = persist_enter(s);
    double* ld_arg = factor_single ? ldiag : nullptr;
    const void* fn = factor_single ? (const void*)chol_backsolve_persistent_kernel<true>
                                   : (const void*)chol_backsolve_persistent_kernel<false>;
    if (chol_cooperative()) {
      void* args[] = {&sys, &n, &ld, &x, &err, &ld_arg};
      static const bool force_refusal = getenv("DROID_CHOL_FORCE_BS_REFUSAL") != nullptr;   // test switch
      if (!force_refusal && hipLaunchCooperativeKernel(fn, dim3(nb), dim3(256), args, 0, s) == hipSuccess) return;
      (void)hipGetLastError();
      // refused: the per-step kernels below run (`x` carries the sentinel preset, they overwrite all of it).  After a
      // single-launch factorisation the diagonal tiles must first come back from the side buffer.
      if (factor_single)
        hipLaunchKernelGGL(chol_ldiag_to_sys_kernel, dim3(nb), dim3(256), 0, s, sys, n, ld, ldiag);
    } else {
      if (factor_single)
        hipLaunchKernelGGL(chol_backsolve_persistent_kernel<true>, dim3(nb), dim3(256), 0, s, sys, n, ld, x, err, ldiag);
      else
        hipLaunchKernelGGL(chol_backsolve_persistent_kernel<false>, dim3(nb), dim3(256), 0, s, sys, n, ld, x, err,
                           nullptr);
      return;
    }
  }
  for (int k = nb - 1; k >= 0; k--) {
    const int c0 = k * NB;
    hipLaunchKernelGGL(chol_backsolve_kernel, dim3(1 + (c0 + 255) / 256), dim3(256), 0, s, sys, n,
                       ld, k, x);
  }
}

// x [n] and flags [chol_flag_words(n)] are pre-set to 0xFF bytes here ("nothing published yet"); when the
// flags directly follow x (BA workspace) one fill covers both.
void launch_chol_solve(double* sys, int n, int ld, double lm, double ep, double* x, int* fail_flag,
                       int* flags, double* ldiag, hipStream_t s, bool preset_done) {
  if (n <= 0) return;
  if (preset_done) {
    const bool single = launch_chol_factor(sys, n, ld, lm, ep, fail_flag, flags, ldiag, s);
    launch_chol_backsolve(sys, n, ld, x, flags, ldiag, fail_flag, s, single);
    return;
  }
  if (ldiag)  // hand-over slots of the panel tiles: data-tagged, 0xFF bytes = not there yet
    (void)hipMemsetAsync(ldiag + chol_lfin_offset(n), 0xFF, sizeof(double) * chol_tiles(n) * NB * NB, s);
  const size_t xbytes = sizeof(double) * (size_t)n, fbytes = flags ? sizeof(int) * chol_flag_words(n) : 0;
  char* xb = reinterpret_cast<char*>(x);
  char* fb = reinterpret_cast<char*>(flags);
  if (flags && fb >= xb + xbytes && fb - xb <= (ptrdiff_t)(xbytes + 4096)) {
    (void)hipMemsetAsync(x, 0xFF, (size_t)(fb - xb) + fbytes, s);
  } else {
    (void)hipMemsetAsync(x, 0xFF, xbytes, s);
    if (flags) (void)hipMemsetAsync(flags, 0xFF, fbytes, s);
  }
  const bool single = launch_chol_factor(sys, n, ld, lm, ep, fail_flag, flags, ldiag, s);
  launch_chol_backsolve(sys, n, ld, x, flags, ldiag, fail_flag, s, single);
}

#ifdef CHOL_STAMPS
extern "C" int droid_debug_bs_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bs_stamps), sizeof(unsigned long long) * 64 * 8);
}
extern "C" int droid_debug_chol_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chol_stamps), sizeof(unsigned long long) * 64 * 16);
}
#endif

// helper for droid_chol_solve: pack (A, b) into the augmented layout
__global__ void chol_pack_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                 double* __restrict__ S, int n, int ld) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)(n + 1) * ld) return;
  const int r = (int)(idx / ld), c = (int)(idx % ld);
  double val = 0.0;
  if (r < n && c < n) val = A[(size_t)r * n + c];
  else if (r == n && c < n) val = b[c];
  S[idx] = val;
}

void launch_chol_pack(const double* A, const double* b, double* S, int n, int ld, hipStream_t s) {
  const size_t tot = (size_t)(n + 1) * ld;
  hipLaunchKernelGGL(chol_pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, b,
                     S, n, ld);
}

}  // namespace droid
