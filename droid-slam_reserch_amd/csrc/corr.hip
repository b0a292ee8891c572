// corr.hip -- correlation lookups on gfx950.
//
//  * corr_index_forward/backward: bilinear (2r+1)^2 window lookup in a precomputed all-pairs
//    volume (/root/reference/src/correlation_kernels.cu:19-124).  Pure gather, HBM bound.
//  * altcorr_forward/backward: the same lookup with the correlation computed on the fly from
//    channels-last feature maps (/root/reference/src/altcorr_kernel.cu:27-286).
//
// Arithmetic contract (restated from the reference, checked bit-for-bit against oracle/corr.py
// for f16/f32 volumes): every bilinear weight is formed in fp32 and rounded to the element type,
// each product is rounded to the element type, and the four contributions of one output are
// added in the order taps (a,c), (a,c+1), (a+1,c), (a+1,c+1) with a rounding after every add
// (correlation_kernels.cu:46-66; a = x offset, c = y offset).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/droid_backends_hip.h"

// The rounding points are part of the contract: no fused multiply-add and no mixed-precision
// folding (hipcc's default -ffp-contract=fast turns round_f16(a*b) into v_fma_mixlo_f16, which
// rounds once instead of f32-then-f16 like the reference's `scalar_t(dx * dy)`).
#pragma clang fp contract(off)

namespace droid {

// ---- element-type arithmetic with the reference's rounding points -------------------------
template <typename T> struct Elem;
template <> struct Elem<__half> {
  typedef float work;  // c10::Half operators compute in fp32 and round to half
  static __device__ __forceinline__ float load(const __half* p) { return __half2float(*p); }
  static __device__ __forceinline__ float round(float x) { return __half2float(__float2half_rn(x)); }
  static __device__ __forceinline__ void store(__half* p, float x) { *p = __float2half_rn(x); }
  static __device__ __forceinline__ float mul(float a, float b) { return round(a * b); }
  static __device__ __forceinline__ float add(float a, float b) { return round(a + b); }
};
template <> struct Elem<float> {
  typedef float work;
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ float round(float x) { return x; }
  static __device__ __forceinline__ void store(float* p, float x) { *p = x; }
  static __device__ __forceinline__ float mul(float a, float b) { return a * b; }
  static __device__ __forceinline__ float add(float a, float b) { return a + b; }
};
template <> struct Elem<double> {
  typedef double work;
  static __device__ __forceinline__ double load(const double* p) { return *p; }
  static __device__ __forceinline__ double round(double x) { return x; }
  static __device__ __forceinline__ void store(double* p, double x) { *p = x; }
  static __device__ __forceinline__ double mul(double a, double b) { return a * b; }
  static __device__ __forceinline__ double add(double a, double b) { return a + b; }
};

// Keeps an fp32 product as a real fp32 value: without it the backend folds
// round_f16(a * b) into v_fma_mixlo_f16, which rounds the exact product once (measured on
// gfx950: differs from the reference's f32-multiply-then-convert in ~1e-4 of the weights).
__device__ __forceinline__ float f32_value(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

struct Bilin {
  int x1, y1;       // top-left integer tap = floor(coord) - r
  float dx, dy;     // fractional parts, fp32 (ck:42-43)
};

__device__ __forceinline__ Bilin bilin_setup(float x0, float y0, int r) {
  Bilin b;
  const float fx = floorf(x0), fy = floorf(y0);
  b.dx = x0 - fx;
  b.dy = y0 - fy;
  // clamp far-away coordinates before the int conversion: any window that far out is empty
  const float lim = 1.0e6f;
  b.x1 = (int)fminf(fmaxf(fx, -lim), lim) - r;
  b.y1 = (int)fminf(fmaxf(fy, -lim), lim) - r;
  if (!(x0 == x0) || !(y0 == y0)) { b.x1 = -2000000; b.y1 = -2000000; }
  return b;
}

// ---- the bilinear combine of one query: (2r+1)^2 outputs from (2r+2)^2 taps ---------------------------------
// out(a, c) = ((tap[c][a] w00 + tap[c+1][a] w01) + tap[c][a+1] w10) + tap[c+1][a+1] w11, every product and every sum rounded
// to the element type (ck:55-66); output plane a (2r+1) + c.
template <typename T, int R>
struct Combine {
  typedef typename Elem<T>::work work;
  static constexpr int RD = 2 * R + 1, NT = RD + 1;
  static __device__ __forceinline__ void run(const work (&tap)[NT][NT], work w00, work w01, work w10, work w11,
                                             T* __restrict__ out, int H1W1) {
#pragma unroll
    for (int a = 0; a < RD; a++) {
#pragma unroll
      for (int c = 0; c < RD; c++) {
        work acc = Elem<T>::mul(tap[c][a], w00);  // 0 + p == p exactly
        acc = Elem<T>::add(acc, Elem<T>::mul(tap[c + 1][a], w01));
        acc = Elem<T>::add(acc, Elem<T>::mul(tap[c][a + 1], w10));
        acc = Elem<T>::add(acc, Elem<T>::mul(tap[c + 1][a + 1], w11));
        Elem<T>::store(out + (size_t)(a * RD + c) * H1W1, acc);
      }
    }
  }
};
// Half volumes: the same roundings on the packed half ALU, two outputs (x offsets a, a+1) per instruction -- 7 packed
// instructions per pair instead of 21 fp32 / convert instructions per output (the fp32 form made the lookup VALU-bound at the
// small pyramid levels: ~1300 instructions per query and level, 26 us per level at 786 432 queries).  Bit-identical:
//   * tap x weight: both are halves, the product is exact in fp32, so round_half(float(a) * float(b)) -- what c10::Half
//     does -- IS the correctly rounded half product that v_pk_mul_f16 returns;
//   * acc + product: both are halves.  If their exponents differ by <= 13 the fp32 sum is exact and both paths round it
//     once; if by more, the smaller one is below 1/4 of the larger one's half-ulp, so the exact sum, its fp32 rounding and
//     v_pk_add_f16's single rounding all return the larger operand's neighbourhood identically (no value near a tie).
// (No FMA: `#pragma clang fp contract(off)` above; f16 denormals are not flushed in either path.)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int R>
struct Combine<__half, R> {
  static constexpr int RD = 2 * R + 1, NT = RD + 1;
  static __device__ __forceinline__ void run(const float (&tap)[NT][NT], float w00, float w01, float w10, float w11,
                                             __half* __restrict__ out, int H1W1) {
    // P[j][i] = (tap[j][i], tap[j][i+1]); the taps are halves held in floats: the pack is exact
    f16x2 P[NT][NT];
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int i = 0; i < NT; i++)
        P[j][i] = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(tap[j][i], i + 1 < NT ? tap[j][i + 1] : 0.f));
    const f16x2 W00 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(w00, w00));
    const f16x2 W01 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(w01, w01));
    const f16x2 W10 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(w10, w10));
    const f16x2 W11 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(w11, w11));
    _Float16* o = reinterpret_cast<_Float16*>(out);
#pragma unroll
    for (int a = 0; a < RD; a += 2) {
#pragma unroll
      for (int c = 0; c < RD; c++) {
        f16x2 acc = P[c][a] * W00;
        acc = acc + P[c + 1][a] * W01;
        acc = acc + P[c][a + 1] * W10;
        acc = acc + P[c + 1][a + 1] * W11;
        o[(size_t)(a * RD + c) * H1W1] = acc[0];
        if (a + 1 < RD) o[(size_t)((a + 1) * RD + c) * H1W1] = acc[1];
      }
    }
  }
};

// ---- corr_index_forward ------------------------------------------------------------------
// One thread per query pixel, 64 consecutive pixels per wave: each of the (2r+1)^2 output planes
// is written as 64 consecutive elements.  A query's window is (2r+2) rows of (2r+2) contiguous
// taps in ITS OWN plane (planes of neighbouring queries are H2*W2 elements apart), so every row
// lives in one or two cache lines that no other lane shares: each row is therefore fetched with
// ONE dword-aligned wide load (16/32/64 B for f16/f32/f64 at r=3) instead of 2r+2 element loads,
// which made every line travel through L1/L2 2r+2 times.  Taps outside the plane are masked to 0
// after the load (reading the neighbouring row of the same tensor is harmless); only lanes whose
// wide load would leave the tensor fall back to per-element loads.
#ifndef CORR_MINWG
#define CORR_MINWG 1
#endif
typedef uint32_t u32a4 __attribute__((aligned(4)));

template <typename T, int NT>
struct RowLoad;  // NT taps of type T starting at p (element aligned), fp32/fp64 work values

template <int NT>
struct RowLoad<__half, NT> {
  static constexpr int NW = (NT * 2 + 2 + 3) / 4;  // dwords covering NT halfs at either parity
  static __device__ __forceinline__ void load(const __half* p, float* tap) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const u32a4* q = reinterpret_cast<const u32a4*>(reinterpret_cast<const char*>(p) - (a & 2));
    uint32_t w[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) w[k] = q[k];
    const bool odd = (a & 2) != 0;
#pragma unroll
    for (int k = 0; k < NT / 2; k++) {
      const uint32_t v = odd ? __builtin_amdgcn_alignbit(w[k + 1], w[k], 16) : w[k];
      tap[2 * k] = __half2float(__ushort_as_half((unsigned short)(v & 0xffffu)));
      tap[2 * k + 1] = __half2float(__ushort_as_half((unsigned short)(v >> 16)));
    }
  }
  static __device__ __forceinline__ size_t span_bytes() { return NW * 4; }
};

template <int NT>
struct RowLoad<float, NT> {
  static __device__ __forceinline__ void load(const float* p, float* tap) {
    const u32a4* q = reinterpret_cast<const u32a4*>(p);
#pragma unroll
    for (int k = 0; k < NT; k++) tap[k] = __uint_as_float(q[k]);
  }
  static __device__ __forceinline__ size_t span_bytes() { return NT * 4; }
};

template <int NT>
struct RowLoad<double, NT> {
  static __device__ __forceinline__ void load(const double* p, double* tap) {
#pragma unroll
    for (int k = 0; k < NT; k++) tap[k] = p[k];
  }
  static __device__ __forceinline__ size_t span_bytes() { return NT * 8; }
};

// out_bstride: elements between the outputs of consecutive batch entries ((2r+1)^2 H1W1 for the reference's operator;
// levels (2r+1)^2 H1W1 when the levels of a pyramid are written side by side, corr_pyramid_forward); cscale: the
// coordinates are multiplied by it first (1, or 2^-level: exact).
template <typename T, int R>
__global__ __launch_bounds__(256, CORR_MINWG) void corr_index_forward_kernel(const T* __restrict__ volume,
                                                                 const float* __restrict__ coords,
                                                                 T* __restrict__ corr, int H1W1,
                                                                 int H2, int W2, size_t vol_elems,
                                                                 size_t out_bstride, float cscale) {
  typedef typename Elem<T>::work work;
  constexpr int RD = 2 * R + 1, NT = RD + 1;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= H1W1) return;
  const float x0 = coords[((size_t)b * 2 + 0) * H1W1 + pix] * cscale;
  const float y0 = coords[((size_t)b * 2 + 1) * H1W1 + pix] * cscale;
  const Bilin bl = bilin_setup(x0, y0, R);
  const T* plane = volume + ((size_t)b * H1W1 + pix) * ((size_t)H2 * W2);
  const uintptr_t vbeg = reinterpret_cast<uintptr_t>(volume);
  const uintptr_t vend = vbeg + vol_elems * sizeof(T);
  // any tap of the window inside the plane at all?
  const bool xany = (bl.x1 + NT > 0) && (bl.x1 < W2);

  work tap[NT][NT];  // [row j (y)][col i (x)]
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int y1 = bl.y1 + j;
    const bool rowok = xany && (y1 >= 0) && (y1 < H2);
#pragma unroll
    for (int i = 0; i < NT; i++) tap[j][i] = (work)0;
    if (rowok) {
      const T* rp = plane + (ptrdiff_t)y1 * W2 + bl.x1;
      const uintptr_t a0 = reinterpret_cast<uintptr_t>(rp) & ~uintptr_t(3);
      if (a0 >= vbeg && a0 + RowLoad<T, NT>::span_bytes() <= vend) {
        RowLoad<T, NT>::load(rp, tap[j]);
      } else {  // first / last elements of the whole tensor only
#pragma unroll 1
        for (int i = 0; i < NT; i++) {
          const int x1 = bl.x1 + i;
          const work v = (x1 >= 0 && x1 < W2) ? Elem<T>::load(rp + i) : (work)0;
#pragma unroll
          for (int i2 = 0; i2 < NT; i2++)
            if (i2 == i) tap[j][i2] = v;
        }
      }
#pragma unroll
      for (int i = 0; i < NT; i++) {
        const int x1 = bl.x1 + i;
        if (x1 < 0 || x1 >= W2) tap[j][i] = (work)0;
      }
    }
  }
  const float one = 1.0f;
  // weights rounded to the element type (ck:55-65)
  const work w00 = Elem<T>::round((work)f32_value((one - bl.dx) * (one - bl.dy)));  // tap (a  ,c  )
  const work w01 = Elem<T>::round((work)f32_value((one - bl.dx) * bl.dy));          // tap (a  ,c+1)
  const work w10 = Elem<T>::round((work)f32_value(bl.dx * (one - bl.dy)));          // tap (a+1,c  )
  const work w11 = Elem<T>::round((work)f32_value(bl.dx * bl.dy));                  // tap (a+1,c+1)
  T* out = corr + (size_t)b * out_bstride + pix;
  Combine<T, R>::run(tap, w00, w01, w10, w11, out, H1W1);
}

// Small planes (pyramid level 3 at 48x64: 96 bytes per query in fp16, 192 in fp32): the planes of the 64 queries of a wave
// are contiguous in the volume, and the (2r+2)^2 window covers most or all of each, so the wave streams its 64 planes
// with fully coalesced 16-byte loads into LDS (plane pitch + 4 bytes: the lanes' private planes fall into different
// banks) and every lane gathers its taps from there.  Per-lane row loads straight from memory made the same lines pass
// through the texture path once per window row and fetched 1.3-1.4x the volume (profiles/r01_corr_lookup_per_level.txt).
// Arithmetic, rounding points and output layout are those of corr_index_forward_kernel.
constexpr int CS_MAXPLANE = 192;  // bytes per plane served by this kernel.  Measured at 48x64, 256 / 128 edges: 96-byte planes
                                  // (fp16 level 3) 37.6 -> 35.6 us, 192-byte planes (fp32 level 3) 39.2 -> 27.4 us; 384- and 768-byte
                                  // planes (level 2) get SLOWER this way (88 -> 150 us, 64 -> 212 us: 25-49 KB of LDS per wave leave
                                  // 3-6 waves per CU and nothing overlaps the load -> barrier -> gather sequence), so they keep the row loads
template <typename T, int R>
__global__ __launch_bounds__(64) void corr_index_forward_small(const T* __restrict__ volume,
                                                               const float* __restrict__ coords,
                                                               T* __restrict__ corr, int H1W1, int H2, int W2,
                                                               size_t out_bstride, float cscale) {
  typedef typename Elem<T>::work work;
  constexpr int RD = 2 * R + 1, NT = RD + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const int pix0 = blockIdx.x * 64, b = blockIdx.y;
  const int nq = min(64, H1W1 - pix0);
  const int PB = H2 * W2 * (int)sizeof(T);      // bytes per plane: a multiple of 16 (checked by the launcher)
  const int pitch = PB + 4;
  const unsigned char* src = reinterpret_cast<const unsigned char*>(volume + ((size_t)b * H1W1 + pix0) * ((size_t)H2 * W2));
  const int nvec = (nq * PB) >> 4;
  for (int u = lane; u < nvec; u += 64) {
    const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)u * 16);
    const int byte = u * 16, p = byte / PB, off = byte - p * PB;
    unsigned* d = reinterpret_cast<unsigned*>(smem + p * pitch + off);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  if (lane >= nq) return;
  const int pix = pix0 + lane;
  const float x0 = coords[((size_t)b * 2 + 0) * H1W1 + pix] * cscale;
  const float y0 = coords[((size_t)b * 2 + 1) * H1W1 + pix] * cscale;
  const Bilin bl = bilin_setup(x0, y0, R);
  const T* plane = reinterpret_cast<const T*>(smem + lane * pitch);
  work tap[NT][NT];  // [row j (y)][col i (x)]
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int y1 = bl.y1 + j;
    const bool rowok = (y1 >= 0) && (y1 < H2);
#pragma unroll
    for (int i = 0; i < NT; i++) {
      const int x1 = bl.x1 + i;
      const bool ok = rowok && x1 >= 0 && x1 < W2;
      tap[j][i] = ok ? Elem<T>::load(plane + (ok ? y1 * W2 + x1 : 0)) : (work)0;
    }
  }
  const float one = 1.0f;
  const work w00 = Elem<T>::round((work)f32_value((one - bl.dx) * (one - bl.dy)));  // tap (a  ,c  )   ck:55-65
  const work w01 = Elem<T>::round((work)f32_value((one - bl.dx) * bl.dy));          // tap (a  ,c+1)
  const work w10 = Elem<T>::round((work)f32_value(bl.dx * (one - bl.dy)));          // tap (a+1,c  )
  const work w11 = Elem<T>::round((work)f32_value(bl.dx * bl.dy));                  // tap (a+1,c+1)
  T* out = corr + (size_t)b * out_bstride + pix;
  Combine<T, R>::run(tap, w00, w01, w10, w11, out, H1W1);
}

template <typename T, int R>
static bool launch_corr_small(const T* v, const float* coords, T* c, int B, int HW, int H2, int W2, size_t obs,
                              float cs, hipStream_t s) {
  static const bool off = (getenv("DROID_CORR_NO_SMALL") != nullptr);  // diagnostics: the per-lane row-load kernel
  const int PB = H2 * W2 * (int)sizeof(T);
  if (off || PB > CS_MAXPLANE || (PB & 15) != 0 || (reinterpret_cast<uintptr_t>(v) & 15) != 0) return false;
  hipLaunchKernelGGL((corr_index_forward_small<T, R>), dim3((HW + 63) / 64, B), dim3(64), 64 * (PB + 4), s, v, coords, c,
                     HW, H2, W2, obs, cs);
  return true;
}

// Mid-size planes (pyramid levels 1-2 at 48x64: rows of 64 / 32 bytes, so a 128-byte line holds 2-4 window rows).  With one
// thread per query, every lane asks for its 8 window rows in 8 separate instructions, each touching 64 different lines: the
// same line is requested 2-4 times per query and, with 12 waves streaming 20+ KB each through a 32 KB L1, comes from L2
// again (PMC r02: 424 B fetched per level-2 query for ~320 touched).  Here the wave loads COOPERATIVELY: in pass `it` lane
// (q = lane >> 3, j = lane & 7) fetches row j of the window of query 8 it + q, so the 8 rows of a query sit in adjacent
// lanes of ONE instruction and the memory pipeline merges them into the 2-5 lines they occupy: every line is requested once
// per query.  The rows travel through a wave-private LDS image [query][row][dwords] (pitch odd: conflict-free both ways) to
// the lane that owns the query; arithmetic, rounding points and output layout are those of corr_index_forward_kernel.
template <typename T, int NT> struct RowWords;   // dwords a row load brings (RowLoad) and the taps' place in them
template <int NT> struct RowWords<__half, NT> { static constexpr int NW = RowLoad<__half, NT>::NW; };
template <int NT> struct RowWords<float, NT> { static constexpr int NW = NT; };

template <typename T, int R>
__global__ __launch_bounds__(256) void corr_index_forward_coop(const T* __restrict__ volume,
                                                               const float* __restrict__ coords,
                                                               T* __restrict__ corr, int H1W1, int H2, int W2,
                                                               size_t vol_elems, size_t out_bstride, float cscale) {
  typedef typename Elem<T>::work work;
  constexpr int RD = 2 * R + 1, NT = RD + 1;
  static_assert(NT == 8, "eight window rows <-> eight lanes per query");
  constexpr int NW = RowWords<T, NT>::NW;
  constexpr bool HALF = sizeof(T) == 2;
  // image row: half -- the 8 taps themselves (the loader shifts an odd-aligned row into place): 4 dwords, 16-byte LDS
  // accesses, query pitch 36 dwords (9.2 KB per wave: 16 waves per CU); float -- the 8 dwords as loaded, pitch 65
  constexpr int RW = HALF ? 4 : NW, PQ = HALF ? NT * RW + 4 : NT * RW + 1;
  __shared__ __attribute__((aligned(16))) uint32_t sm[4][64 * PQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* img = sm[wave];
  const int pix0 = blockIdx.x * 256 + wave * 64;   // first query of this wave
  const int pix = pix0 + lane;
  const int b = blockIdx.y;
  const bool valid = pix < H1W1;
  const int cpix = valid ? pix : H1W1 - 1;
  const float x0 = coords[((size_t)b * 2 + 0) * H1W1 + cpix] * cscale;
  const float y0 = coords[((size_t)b * 2 + 1) * H1W1 + cpix] * cscale;
  const Bilin bl = bilin_setup(x0, y0, R);
  const uintptr_t vbeg = reinterpret_cast<uintptr_t>(volume);
  const uintptr_t vend = vbeg + vol_elems * sizeof(T);
  const size_t plane_elems = (size_t)H2 * W2;
  const bool xany = valid && (bl.x1 + NT > 0) && (bl.x1 < W2);
  // ---- cooperative loads: pass `it` serves queries 8 it .. 8 it + 7
  const int lq = lane >> 3, lj = lane & 7;
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int q = 8 * it + lq;
    const int qx1 = __shfl(bl.x1, q), qy1 = __shfl(bl.y1, q);
    const bool qany = __shfl((int)xany, q) != 0;
    const int y1 = qy1 + lj;
    if (qany && y1 >= 0 && y1 < H2) {
      const T* rp = volume + ((size_t)b * H1W1 + (pix0 + q)) * plane_elems + (ptrdiff_t)y1 * W2 + qx1;
      const uintptr_t a0 = reinterpret_cast<uintptr_t>(rp) & ~uintptr_t(3);
      if (a0 >= vbeg && a0 + NW * 4 <= vend) {   // (rows at the two ends of the tensor: the owner loads them itself)
        const u32a4* src = reinterpret_cast<const u32a4*>(a0);
        uint32_t w[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) w[k] = src[k];
        uint32_t* d = img + q * PQ + lj * RW;
        if constexpr (HALF) {
          const bool odd = (reinterpret_cast<uintptr_t>(rp) & 2) != 0;
          uint4 v;
          v.x = odd ? __builtin_amdgcn_alignbit(w[1], w[0], 16) : w[0];
          v.y = odd ? __builtin_amdgcn_alignbit(w[2], w[1], 16) : w[1];
          v.z = odd ? __builtin_amdgcn_alignbit(w[3], w[2], 16) : w[2];
          v.w = odd ? __builtin_amdgcn_alignbit(w[4], w[3], 16) : w[3];
          *reinterpret_cast<uint4*>(d) = v;
        } else {
#pragma unroll
          for (int k = 0; k < NW; k++) d[k] = w[k];
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (!valid) return;
  // ---- the owner of the query collects its window
  const T* plane = volume + ((size_t)b * H1W1 + pix) * plane_elems;
  work tap[NT][NT];  // [row j (y)][col i (x)]
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int y1 = bl.y1 + j;
    const bool rowok = xany && (y1 >= 0) && (y1 < H2);
#pragma unroll
    for (int i = 0; i < NT; i++) tap[j][i] = (work)0;
    if (rowok) {
      const T* rp = plane + (ptrdiff_t)y1 * W2 + bl.x1;
      const uintptr_t a = reinterpret_cast<uintptr_t>(rp);
      const uintptr_t a0 = a & ~uintptr_t(3);
      if (a0 >= vbeg && a0 + NW * 4 <= vend) {
        const uint32_t* d = img + lane * PQ + j * RW;
        if constexpr (HALF) {
          const uint4 v4 = *reinterpret_cast<const uint4*>(d);
          const uint32_t w[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
          for (int k = 0; k < NT / 2; k++) {
            tap[j][2 * k] = __half2float(__ushort_as_half((unsigned short)(w[k] & 0xffffu)));
            tap[j][2 * k + 1] = __half2float(__ushort_as_half((unsigned short)(w[k] >> 16)));
          }
        } else {
#pragma unroll
          for (int k = 0; k < NT; k++) tap[j][k] = __uint_as_float(d[k]);
        }
      } else {  // first / last elements of the whole tensor only
#pragma unroll 1
        for (int i = 0; i < NT; i++) {
          const int x1 = bl.x1 + i;
          const work v = (x1 >= 0 && x1 < W2) ? Elem<T>::load(rp + i) : (work)0;
#pragma unroll
          for (int i2 = 0; i2 < NT; i2++)
            if (i2 == i) tap[j][i2] = v;
        }
      }
#pragma unroll
      for (int i = 0; i < NT; i++) {
        const int x1 = bl.x1 + i;
        if (x1 < 0 || x1 >= W2) tap[j][i] = (work)0;
      }
    }
  }
  const float one = 1.0f;
  const work w00 = Elem<T>::round((work)f32_value((one - bl.dx) * (one - bl.dy)));  // tap (a  ,c  )   ck:55-65
  const work w01 = Elem<T>::round((work)f32_value((one - bl.dx) * bl.dy));          // tap (a  ,c+1)
  const work w10 = Elem<T>::round((work)f32_value(bl.dx * (one - bl.dy)));          // tap (a+1,c  )
  const work w11 = Elem<T>::round((work)f32_value(bl.dx * bl.dy));                  // tap (a+1,c+1)
  T* out = corr + (size_t)b * out_bstride + pix;
  Combine<T, R>::run(tap, w00, w01, w10, w11, out, H1W1);
}

// rows of at most 64 bytes (two or more window rows per 128-byte line), radius 3, half / float
template <typename T, int R>
static bool launch_corr_coop(const T* v, const float* coords, T* c, int B, int HW, int H2, int W2, size_t vol_elems,
                             size_t obs, float cs, hipStream_t s) {
  if constexpr (R != 3 || sizeof(T) > 4) {
    return false;
  } else {
    static const bool off = (getenv("DROID_CORR_NO_COOP") != nullptr);  // diagnostics: the per-lane row-load kernel
    if (off || W2 * (int)sizeof(T) > 64 || (HW & 63) != 0) return false;
    hipLaunchKernelGGL((corr_index_forward_coop<T, R>), dim3((HW + 255) / 256, B), dim3(256), 0, s, v, coords, c, HW, H2, W2,
                       vol_elems, obs, cs);
    return true;
  }
}

// Any radius (slow path): taps are re-read per output.
template <typename T>
__global__ __launch_bounds__(256) void corr_index_forward_generic(const T* __restrict__ volume,
                                                                  const float* __restrict__ coords,
                                                                  T* __restrict__ corr, int H1W1,
                                                                  int H2, int W2, int r) {
  typedef typename Elem<T>::work work;
  const int rd = 2 * r + 1;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= H1W1) return;
  const Bilin bl = bilin_setup(coords[((size_t)b * 2 + 0) * H1W1 + pix],
                               coords[((size_t)b * 2 + 1) * H1W1 + pix], r);
  const T* plane = volume + ((size_t)b * H1W1 + pix) * ((size_t)H2 * W2);
  auto tapv = [&](int i, int j) -> work {
    const int x1 = bl.x1 + i, y1 = bl.y1 + j;
    if (x1 < 0 || x1 >= W2 || y1 < 0 || y1 >= H2) return (work)0;
    return Elem<T>::load(plane + (size_t)y1 * W2 + x1);
  };
  const work w00 = Elem<T>::round((work)f32_value((1.0f - bl.dx) * (1.0f - bl.dy)));
  const work w01 = Elem<T>::round((work)f32_value((1.0f - bl.dx) * bl.dy));
  const work w10 = Elem<T>::round((work)f32_value(bl.dx * (1.0f - bl.dy)));
  const work w11 = Elem<T>::round((work)f32_value(bl.dx * bl.dy));
  T* out = corr + (size_t)b * rd * rd * H1W1 + pix;
  for (int a = 0; a < rd; a++)
    for (int c = 0; c < rd; c++) {
      work acc = Elem<T>::mul(tapv(a, c), w00);
      acc = Elem<T>::add(acc, Elem<T>::mul(tapv(a, c + 1), w01));
      acc = Elem<T>::add(acc, Elem<T>::mul(tapv(a + 1, c), w10));
      acc = Elem<T>::add(acc, Elem<T>::mul(tapv(a + 1, c + 1), w11));
      Elem<T>::store(out + (size_t)(a * rd + c) * H1W1, acc);
    }
}

template <typename T>
static int corr_index_forward_t(const void* volume, const float* coords, void* corr, int B, int H1,
                                int W1, int H2, int W2, int r, hipStream_t s) {
  const int HW = H1 * W1;
  dim3 grid((HW + 255) / 256, B), block(256);
  const T* v = static_cast<const T*>(volume);
  T* c = static_cast<T*>(corr);
  const size_t vol_elems = (size_t)B * HW * H2 * W2;
  const size_t obs = (size_t)(2 * r + 1) * (2 * r + 1) * HW;
  if (r == 3 && launch_corr_small<T, 3>(v, coords, c, B, HW, H2, W2, obs, 1.0f, s)) return 0;
  if (r == 4 && launch_corr_small<T, 4>(v, coords, c, B, HW, H2, W2, obs, 1.0f, s)) return 0;
  if (r == 3 && launch_corr_coop<T, 3>(v, coords, c, B, HW, H2, W2, vol_elems, obs, 1.0f, s)) return 0;
  if (r == 3)
    hipLaunchKernelGGL((corr_index_forward_kernel<T, 3>), grid, block, 0, s, v, coords, c, HW, H2, W2, vol_elems, obs, 1.0f);
  else if (r == 4)
    hipLaunchKernelGGL((corr_index_forward_kernel<T, 4>), grid, block, 0, s, v, coords, c, HW, H2, W2, vol_elems, obs, 1.0f);
  else
    hipLaunchKernelGGL((corr_index_forward_generic<T>), grid, block, 0, s, v, coords, c, HW, H2, W2, r);
  return 0;
}

int launch_corr_index_forward(const void* volume, const float* coords, void* corr, int B, int H1,
                              int W1, int H2, int W2, int r, int dtype, hipStream_t s) {
  if (B > 65535) return DROID_E_ARG;
  switch (dtype) {
    case DROID_F16: return corr_index_forward_t<__half>(volume, coords, corr, B, H1, W1, H2, W2, r, s);
    case DROID_F32: return corr_index_forward_t<float>(volume, coords, corr, B, H1, W1, H2, W2, r, s);
    case DROID_F64: return corr_index_forward_t<double>(volume, coords, corr, B, H1, W1, H2, W2, r, s);
  }
  return DROID_E_ARG;
}

// CorrBlock.__call__ (droid_slam/modules/corr.py:40-50) for all pyramid levels: level l looks up volumes[l]
// ([B,H1,W1,H1>>l,W1>>l]) at coords * 2^-l and writes channels [l (2r+1)^2, (l+1) (2r+1)^2) of
// corr [B, levels (2r+1)^2, H1, W1] -- the tensor torch.cat(out_pyramid, dim=2) would build, without the cat
// (600 MB of extra traffic at 256 edges) and without the per-level coordinate tensors.  One launch per level.
template <typename T>
static int corr_pyramid_forward_t(const void* const* volumes, const float* coords, void* corr, int B, int H1, int W1,
                                  int r, int levels, hipStream_t s) {
  const int HW = H1 * W1, rd2 = (2 * r + 1) * (2 * r + 1);
  dim3 grid((HW + 255) / 256, B), block(256);
  const size_t obs = (size_t)levels * rd2 * HW;
  for (int l = 0; l < levels; l++) {
    const int H2 = H1 >> l, W2 = W1 >> l;
    const T* v = static_cast<const T*>(volumes[l]);
    T* c = static_cast<T*>(corr) + (size_t)l * rd2 * HW;
    const size_t vol_elems = (size_t)B * HW * H2 * W2;
    const float cs = 1.0f / (float)(1 << l);
    if (r == 3 && launch_corr_small<T, 3>(v, coords, c, B, HW, H2, W2, obs, cs, s)) continue;
    if (r == 4 && launch_corr_small<T, 4>(v, coords, c, B, HW, H2, W2, obs, cs, s)) continue;
    if (r == 3 && launch_corr_coop<T, 3>(v, coords, c, B, HW, H2, W2, vol_elems, obs, cs, s)) continue;
    if (r == 3)
      hipLaunchKernelGGL((corr_index_forward_kernel<T, 3>), grid, block, 0, s, v, coords, c, HW, H2, W2, vol_elems, obs, cs);
    else
      hipLaunchKernelGGL((corr_index_forward_kernel<T, 4>), grid, block, 0, s, v, coords, c, HW, H2, W2, vol_elems, obs, cs);
  }
  return 0;
}

int launch_corr_pyramid_forward(const void* const* volumes, const float* coords, void* corr, int B, int H1, int W1,
                                int r, int levels, int dtype, hipStream_t s) {
  if (B > 65535 || (r != 3 && r != 4) || levels < 1 || levels > 8 || (H1 >> (levels - 1)) < 1 || (W1 >> (levels - 1)) < 1)
    return DROID_E_ARG;
  switch (dtype) {
    case DROID_F16: return corr_pyramid_forward_t<__half>(volumes, coords, corr, B, H1, W1, r, levels, s);
    case DROID_F32: return corr_pyramid_forward_t<float>(volumes, coords, corr, B, H1, W1, r, levels, s);
    case DROID_F64: return corr_pyramid_forward_t<double>(volumes, coords, corr, B, H1, W1, r, levels, s);
  }
  return DROID_E_ARG;
}

// ---- corr_index_backward (ck:73-124): each query scatters into its own plane -> no races ---
template <typename T>
__global__ __launch_bounds__(256) void corr_index_backward_kernel(const float* __restrict__ coords,
                                                                  const T* __restrict__ corr_grad,
                                                                  T* __restrict__ volume_grad,
                                                                  int H1W1, int H2, int W2, int r) {
  typedef typename Elem<T>::work work;
  const int rd = 2 * r + 1;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= H1W1) return;
  const Bilin bl = bilin_setup(coords[((size_t)b * 2 + 0) * H1W1 + pix],
                               coords[((size_t)b * 2 + 1) * H1W1 + pix], r);
  T* plane = volume_grad + ((size_t)b * H1W1 + pix) * ((size_t)H2 * W2);
  const T* g = corr_grad + (size_t)b * rd * rd * H1W1 + pix;
  const work w11 = Elem<T>::round((work)f32_value(bl.dx * bl.dy));
  const work w10 = Elem<T>::round((work)f32_value(bl.dx * (1.0f - bl.dy)));
  const work w01 = Elem<T>::round((work)f32_value((1.0f - bl.dx) * bl.dy));
  const work w00 = Elem<T>::round((work)f32_value((1.0f - bl.dx) * (1.0f - bl.dy)));
  for (int i = 0; i < rd + 1; i++)
    for (int j = 0; j < rd + 1; j++) {
      const int x1 = bl.x1 + i, y1 = bl.y1 + j;
      if (x1 < 0 || x1 >= W2 || y1 < 0 || y1 >= H2) continue;
      work acc = (work)0;  // ck:106-117, same order and rounding points
      if (i > 0 && j > 0)
        acc = Elem<T>::add(acc, Elem<T>::mul(Elem<T>::load(g + (size_t)((i - 1) * rd + (j - 1)) * H1W1), w11));
      if (i > 0 && j < rd)
        acc = Elem<T>::add(acc, Elem<T>::mul(Elem<T>::load(g + (size_t)((i - 1) * rd + j) * H1W1), w10));
      if (i < rd && j > 0)
        acc = Elem<T>::add(acc, Elem<T>::mul(Elem<T>::load(g + (size_t)(i * rd + (j - 1)) * H1W1), w01));
      if (i < rd && j < rd)
        acc = Elem<T>::add(acc, Elem<T>::mul(Elem<T>::load(g + (size_t)(i * rd + j) * H1W1), w00));
      Elem<T>::store(plane + (size_t)y1 * W2 + x1, acc);
    }
}

template <typename T>
static int corr_index_backward_t(const float* coords, const void* corr_grad, void* volume_grad,
                                 int B, int H1, int W1, int H2, int W2, int r, hipStream_t s) {
  const int HW = H1 * W1;
  (void)hipMemsetAsync(volume_grad, 0, sizeof(T) * (size_t)B * HW * H2 * W2, s);
  hipLaunchKernelGGL((corr_index_backward_kernel<T>), dim3((HW + 255) / 256, B), dim3(256), 0, s,
                     coords, static_cast<const T*>(corr_grad), static_cast<T*>(volume_grad), HW, H2,
                     W2, r);
  return 0;
}

int launch_corr_index_backward(const float* coords, const void* corr_grad, void* volume_grad, int B,
                               int H1, int W1, int H2, int W2, int r, int dtype, hipStream_t s) {
  if (B > 65535) return DROID_E_ARG;
  switch (dtype) {
    case DROID_F16: return corr_index_backward_t<__half>(coords, corr_grad, volume_grad, B, H1, W1, H2, W2, r, s);
    case DROID_F32: return corr_index_backward_t<float>(coords, corr_grad, volume_grad, B, H1, W1, H2, W2, r, s);
    case DROID_F64: return corr_index_backward_t<double>(coords, corr_grad, volume_grad, B, H1, W1, H2, W2, r, s);
  }
  return DROID_E_ARG;
}

// ---- altcorr_forward (generic path) -------------------------------------------------------
// 16 lanes per query pixel, 4 queries per wave; every lane owns C/16 channels (strided by 16
// float4s so that a query's 16 lanes read one contiguous run of the channels-last row).
// corr(oy,ox) = sum over the 4 neighbouring taps of <f1, f2(tap)> * bilinear weight, channel of
// the output = ox*(2r+1)+oy (ak:109-142).
template <typename T>
__global__ __launch_bounds__(256) void altcorr_forward_generic(const T* __restrict__ fmap1,
                                                               const T* __restrict__ fmap2,
                                                               const float* __restrict__ coords,
                                                               T* __restrict__ corr, int N, int H1W1,
                                                               int H2, int W2, int C, int r) {
  typedef typename Elem<T>::work work;
  const int rd = 2 * r + 1;
  const int sub = threadIdx.x & 15;
  const int q = (blockIdx.x * 256 + threadIdx.x) >> 4;  // query index within (b, n)
  const int n = blockIdx.y, b = blockIdx.z;
  const bool qok = q < H1W1;
  const int pix = qok ? q : 0;
  const float* cp = coords + (((size_t)b * N + n) * H1W1 + pix) * 2;
  const Bilin bl = bilin_setup(cp[0], cp[1], r);
  const T* f1 = fmap1 + ((size_t)b * H1W1 + pix) * C;
  const T* f2b = fmap2 + (size_t)b * H2 * W2 * C;
  T* out = corr + (((size_t)b * N + n) * rd * rd) * H1W1 + pix;
  const work wnw = Elem<T>::round((work)f32_value(bl.dy * bl.dx));                  // ak:119-122
  const work wne = Elem<T>::round((work)f32_value(bl.dy * (1.0f - bl.dx)));
  const work wsw = Elem<T>::round((work)f32_value((1.0f - bl.dy) * bl.dx));
  const work wse = Elem<T>::round((work)f32_value((1.0f - bl.dy) * (1.0f - bl.dx)));
  // dot products of the (rd+1)^2 taps, one row of taps at a time, two rows live
  for (int ox = 0; ox < rd; ox++) {
    for (int oy = 0; oy < rd; oy++) {
      work s4[4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int iy = oy + (t >> 1), ix = ox + (t & 1);
        const int h2 = bl.y1 + iy, w2 = bl.x1 + ix;
        work s = (work)0;
        if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
          const T* f2 = f2b + ((size_t)h2 * W2 + w2) * C;
          for (int c = sub; c < C; c += 16) s = fma((work)Elem<T>::load(f1 + c), (work)Elem<T>::load(f2 + c), s);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        s4[t] = Elem<T>::round(s);
      }
      // taps: t=0 (oy,ox) se, t=1 (oy,ox+1) sw, t=2 (oy+1,ox) ne, t=3 (oy+1,ox+1) nw
      work acc = Elem<T>::mul(s4[0], wse);
      acc = Elem<T>::add(acc, Elem<T>::mul(s4[1], wsw));
      acc = Elem<T>::add(acc, Elem<T>::mul(s4[2], wne));
      acc = Elem<T>::add(acc, Elem<T>::mul(s4[3], wnw));
      if (qok && sub == 0) Elem<T>::store(out + (size_t)(ox * rd + oy) * H1W1, acc);
    }
  }
}

// ---- altcorr_forward, fp32 fast path: LDS-staged fmap2 windows ------------------------------
// One workgroup = an 8x8 tile of query pixels of one (edge, coordinate set).  The windows of
// neighbouring queries overlap almost completely when the flow is locally coherent, so the
// workgroup stages the bounding box of all 64 windows (<= ALT_MAXPOS positions) of fmap2 in LDS,
// 32 channels at a time, and every (query, tap) dot product reads its fmap2 row from LDS with
// 16-byte reads (row pitch 36 floats: conflict-free for neighbouring positions) against the
// query's fmap1 chunk held in registers.  Wave w owns tap rows j = w (mod 4) of all 64 queries;
// the (2r+2)^2 tap sums are exchanged through LDS for the bilinear combine.  Tiles whose bounding
// box does not fit (incoherent coordinates) take the per-query path of the generic kernel.
constexpr int ALT_TQ = 8;          // tile is ALT_TQ x ALT_TQ queries
constexpr int ALT_MAXPOS = 448;    // fmap2 positions staged per tile
#ifndef DROID_ALT_CH
#define DROID_ALT_CH 32
#endif
constexpr int ALT_CH = DROID_ALT_CH;          // channels per stage
constexpr int ALT_WAVES = 4;                  // waves per workgroup: wave w owns tap rows j = w (mod 4)
constexpr int ALT_THREADS = 64 * ALT_WAVES;
constexpr int ALT_PITCH = ALT_CH + 4;         // floats per staged position (pad keeps 16-B alignment)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int R>
__global__ __launch_bounds__(ALT_THREADS, 2) void altcorr_forward_tiled(const float* __restrict__ fmap1,
                                                             const float* __restrict__ fmap2,
                                                             const float* __restrict__ coords,
                                                             float* __restrict__ corr, int N, int H1,
                                                             int W1, int H2, int W2, int C) {
  constexpr int RD = 2 * R + 1, NT = RD + 1;
  constexpr int ROWS_PER_WAVE = (NT + ALT_WAVES - 1) / ALT_WAVES;
  __shared__ __attribute__((aligned(16))) float f2s[(ALT_MAXPOS + 1) * ALT_PITCH];  // +1: zero row
  __shared__ int bbox[4];
  const int tid = threadIdx.x, q = tid & 63, wave = tid >> 6;
  const int tiles_x = (W1 + ALT_TQ - 1) / ALT_TQ;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int n = blockIdx.y, b = blockIdx.z;
  const int qx = tx * ALT_TQ + (q & 7), qy = ty * ALT_TQ + (q >> 3);
  const bool qok = qx < W1 && qy < H1;
  const int H1W1 = H1 * W1;
  const int pix = qok ? qy * W1 + qx : 0;
  const float* cp = coords + (((size_t)b * N + n) * H1W1 + pix) * 2;
  const Bilin bl = bilin_setup(cp[0], cp[1], R);

  if (tid < 4) bbox[tid] = (tid < 2) ? 0x7fffffff : -0x7fffffff;  // NT is even for every radius
  __syncthreads();
  if (qok && wave == 0) {
    // windows that miss the plane completely contribute nothing and do not stretch the box
    if (bl.x1 + NT > 0 && bl.x1 < W2 && bl.y1 + NT > 0 && bl.y1 < H2) {
      atomicMin(&bbox[0], bl.x1);
      atomicMin(&bbox[1], bl.y1);
      atomicMax(&bbox[2], bl.x1 + NT);
      atomicMax(&bbox[3], bl.y1 + NT);
    }
  }
  __syncthreads();
  const int bx0 = max(bbox[0], 0), by0 = max(bbox[1], 0);
  const int bx1 = min(bbox[2], W2), by1 = min(bbox[3], H2);
  const int RW = max(bx1 - bx0, 0), RH = max(by1 - by0, 0);
  const int npos = RW * RH;
  float* out = corr + (((size_t)b * N + n) * RD * RD) * H1W1 + pix;

  if (npos > ALT_MAXPOS) {  // incoherent tile: per-query direct evaluation (same arithmetic order)
    if (!qok) return;
    const float* f1 = fmap1 + ((size_t)b * H1W1 + pix) * C;
    const float* f2b = fmap2 + (size_t)b * H2 * W2 * C;
    const float wnw = f32_value(bl.dy * bl.dx), wne = f32_value(bl.dy * (1.0f - bl.dx));
    const float wsw = f32_value((1.0f - bl.dy) * bl.dx), wse = f32_value((1.0f - bl.dy) * (1.0f - bl.dx));
    for (int o = wave; o < RD * RD; o += ALT_WAVES) {
      const int ox = o / RD, oy = o % RD;
      float s4[4];
      for (int t = 0; t < 4; t++) {
        const int h2 = bl.y1 + oy + (t >> 1), w2 = bl.x1 + ox + (t & 1);
        float s = 0.f;
        if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
          const float* f2 = f2b + ((size_t)h2 * W2 + w2) * C;
          for (int c = 0; c < C; c++) s = fmaf(f1[c], f2[c], s);
        }
        s4[t] = s;
      }
      float acc = s4[0] * wse;
      acc = acc + s4[1] * wsw;
      acc = acc + s4[2] * wne;
      acc = acc + s4[3] * wnw;
      out[(size_t)o * H1W1] = acc;
    }
    return;
  }

  // zero row for out-of-plane taps
  if (tid < ALT_PITCH) f2s[ALT_MAXPOS * ALT_PITCH + tid] = 0.f;

  float acc[ROWS_PER_WAVE][NT];
#pragma unroll
  for (int jr = 0; jr < ROWS_PER_WAVE; jr++)
#pragma unroll
    for (int i = 0; i < NT; i++) acc[jr][i] = 0.f;

  // LDS offsets (in floats) of this thread's taps; ALT_MAXPOS = the zero row
  int toff[ROWS_PER_WAVE][NT];
#pragma unroll
  for (int jr = 0; jr < ROWS_PER_WAVE; jr++) {
    const int j = wave + ALT_WAVES * jr;
    const int yy = bl.y1 + j - by0;
#pragma unroll
    for (int i = 0; i < NT; i++) {
      const int xx = bl.x1 + i - bx0;
      const bool ok = qok && j < NT && yy >= 0 && yy < RH && xx >= 0 && xx < RW;
      toff[jr][i] = (ok ? yy * RW + xx : ALT_MAXPOS) * ALT_PITCH;
    }
  }

  const float* f1p = fmap1 + ((size_t)b * H1W1 + pix) * C;
  const float* f2b = fmap2 + (size_t)b * H2 * W2 * C;
  for (int c0 = 0; c0 < C; c0 += ALT_CH) {
    __syncthreads();  // previous chunk fully consumed
    // this query's fmap1 chunk (in flight during the staging)
    f4 a1[ALT_CH / 4];
#pragma unroll
    for (int k = 0; k < ALT_CH / 4; k++) a1[k] = *reinterpret_cast<const f4*>(f1p + c0 + 4 * k);
    // stage fmap2[by0..by1, bx0..bx1, c0..c0+ALT_CH): (ALT_CH/4) threads x 16 B per position,
    // four independent loads per thread in flight before the LDS stores
    constexpr int TPP = ALT_CH / 4, PPI = ALT_THREADS / TPP;  // threads per position, positions per pass
    for (int base = 0; base < npos; base += PPI * 4) {
      f4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int p = base + u * PPI + tid / TPP;
        if (p < npos) {
          const int yy = p / RW, xx = p - yy * RW;
          v[u] = *reinterpret_cast<const f4*>(f2b + ((size_t)(by0 + yy) * W2 + (bx0 + xx)) * C + c0 + 4 * (tid % TPP));
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int p = base + u * PPI + tid / TPP;
        if (p < npos) *reinterpret_cast<f4*>(&f2s[p * ALT_PITCH + 4 * (tid % TPP)]) = v[u];
      }
    }
    __syncthreads();
    // The loop is LDS-latency bound unless a whole tap (ALT_CH floats = 8 x 16-byte reads) is in
    // flight while the previous tap is multiplied: explicit two-deep register pipeline over taps.
    {
      constexpr int NTAPS = ROWS_PER_WAVE * NT;
      f4 wa[ALT_CH / 4], wb[ALT_CH / 4];
#pragma unroll
      for (int k = 0; k < ALT_CH / 4; k++) wa[k] = *reinterpret_cast<const f4*>(&f2s[toff[0][0] + 4 * k]);
#pragma unroll
      for (int tp = 0; tp < NTAPS; tp++) {
        const int jr = tp / NT, i = tp % NT;
        if (tp + 1 < NTAPS) {
          const int jn = (tp + 1) / NT, in = (tp + 1) % NT;
#pragma unroll
          for (int k = 0; k < ALT_CH / 4; k++) {
            const f4 v = *reinterpret_cast<const f4*>(&f2s[toff[jn][in] + 4 * k]);
            if (tp & 1) wa[k] = v; else wb[k] = v;
          }
        }
        float s0 = acc[jr][i], s1 = 0.f;
#pragma unroll
        for (int k = 0; k < ALT_CH / 4; k += 2) {
          const f4 w0 = (tp & 1) ? wb[k] : wa[k];
          const f4 w1 = (tp & 1) ? wb[k + 1] : wa[k + 1];
          s0 = fmaf(a1[k][0], w0[0], s0);
          s1 = fmaf(a1[k + 1][0], w1[0], s1);
          s0 = fmaf(a1[k][1], w0[1], s0);
          s1 = fmaf(a1[k + 1][1], w1[1], s1);
          s0 = fmaf(a1[k][2], w0[2], s0);
          s1 = fmaf(a1[k + 1][2], w1[2], s1);
          s0 = fmaf(a1[k][3], w0[3], s0);
          s1 = fmaf(a1[k + 1][3], w1[3], s1);
        }
        acc[jr][i] = s0 + s1;
      }
    }
  }
  __syncthreads();
  // exchange the tap sums: taps[q][j][i], pitch NT*NT+1 to spread banks
  constexpr int TP = NT * NT + 1;
  float* taps = f2s;
#pragma unroll
  for (int jr = 0; jr < ROWS_PER_WAVE; jr++) {
    const int j = wave + ALT_WAVES * jr;
    if (j < NT) {
#pragma unroll
      for (int i = 0; i < NT; i++) taps[q * TP + j * NT + i] = acc[jr][i];
    }
  }
  __syncthreads();
  if (!qok) return;
  const float wnw = f32_value(bl.dy * bl.dx), wne = f32_value(bl.dy * (1.0f - bl.dx));          // ak:119-122
  const float wsw = f32_value((1.0f - bl.dy) * bl.dx), wse = f32_value((1.0f - bl.dy) * (1.0f - bl.dx));
  for (int o = wave; o < RD * RD; o += ALT_WAVES) {
    const int ox = o / RD, oy = o % RD;  // channel = ox*(2r+1) + oy
    const float* tq = &taps[q * TP];
    float v = tq[oy * NT + ox] * wse;                 // tap (iy=oy  , ix=ox  )
    v = v + tq[oy * NT + ox + 1] * wsw;               // tap (oy  , ox+1)
    v = v + tq[(oy + 1) * NT + ox] * wne;             // tap (oy+1, ox  )
    v = v + tq[(oy + 1) * NT + ox + 1] * wnw;         // tap (oy+1, ox+1)
    out[(size_t)o * H1W1] = v;
  }
}

// ---- altcorr_forward, fp32 matrix-core path ----------------------------------------------------
// The per-tap dot products of a block of queries against the fmap2 positions their windows cover
// are a GEMM: D[query][position] = sum_c f1[query][c] * f2[position][c].  One workgroup = a
// 16 (x) by 4 (y) tile of query pixels; wave w owns the 4x4 sub-tile x in [4w,4w+4) and multiplies
// its 16 queries (rows) against the bounding box of THEIR windows (16-position column blocks, at
// most AM_MAXBLK of them) on v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation.
//
// The box of the whole workgroup is staged in LDS in stages of 16 channels by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip), as many stages per batch as fit in the 50 KB
// buffer (all 8 for the small boxes of the coarse levels).  The LDS image is lane-linear
// (64 B per position, 16 positions per wave-instruction); bank conflicts are avoided by an XOR
// swizzle applied on the SOURCE side: 16-byte slot s of position P holds channel chunk
// s ^ ((P >> 1) & 3), so the operand reads of 8 consecutive positions cover all 32 banks.  K is
// consumed in a permuted order (k-step e of lane group g = channel 4g+e of the stage), identical
// for both operands, so that A and B fragments are plain 16-byte pieces of channels-last rows.
//
// D leaves the accumulators through LDS as dbuf[query][position] (stores with immediate offsets);
// the bilinear combine walks output columns so that tap rows are shared between consecutive
// outputs, is written exactly like the generic kernel (same order of the four weighted taps)
// and stores 64-byte runs of 16 consecutive query pixels.  Instruction count matters as much as
// the MFMA time here: each SIMD runs 12 waves of this kernel per launch, so every 1000 VALU
// instructions per wave cost ~20 us.  Computing a box instead of 64 taps per query costs box/64 redundant
// flops (2.25x for a smooth flow field) on a pipe that is otherwise idle.  A tile whose boxes do
// not fit (incoherent coordinates) is evaluated per query.
#ifdef AM_STAMPS
#define AM_LIN (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z))
// Diagnostic build only: s_memtime stamps per wave of the first 768 workgroups (three per CU).
__device__ unsigned long long g_am_stamps[768 * 4 * 32];
#define AMSTAMP(slot)                                                                              \
  do {                                                                                             \
    if ((threadIdx.x & 63) == 0 && AM_LIN < 768)                                                    \
      g_am_stamps[(AM_LIN * 4 + (threadIdx.x >> 6)) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define AMNOTE(slot, v)                                                                            \
  do {                                                                                             \
    if ((threadIdx.x & 63) == 0 && AM_LIN < 768)                                                    \
      g_am_stamps[(AM_LIN * 4 + (threadIdx.x >> 6)) * 32 + (slot)] = (unsigned long long)(v);  \
  } while (0)
#else
#define AMSTAMP(slot) do { } while (0)
#define AMNOTE(slot, v) do { } while (0)
#endif
constexpr int AM_TX = 16, AM_TY = 4;     // query tile of a workgroup
// Input element type of the matrix-core path.  A stage is 64 bytes per position whatever the type, so the LDS
// image, the swizzle and the DMA plan are shared: fp32 = 16 channels per stage on v_mfma_f32_16x16x4_f32 (4 k-steps
// per stage), fp16 = 32 channels per stage on ONE v_mfma_f32_16x16x32_f16 (16x the fp32 matrix rate; products of
// two halves are exact in fp32, accumulation is fp32, so the half path reproduces the reference's
// `.float()` evaluation of half feature maps -- modules/corr.py:120 -- up to summation order).
template <typename TI> struct AmIn;
template <> struct AmIn<float> {
  static constexpr int EPC = 4;            // elements per 16-byte chunk
  static constexpr int CH = 16;            // channels per stage
  static constexpr int MAXSTAGE = 8;       // C <= 128 on this path
};
template <> struct AmIn<__half> {
  static constexpr int EPC = 8;
  static constexpr int CH = 32;
  static constexpr int MAXSTAGE = 4;       // C <= 128
};
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void am_store(float* p, float v) { *p = v; }
__device__ __forceinline__ void am_store(__half* p, float v) { *p = __float2half_rn(v); }
constexpr int AM_MAXBLK = 15;            // 16-position blocks per wave (240 positions >= 15x16)
// D exchange: r=3 keeps 12 blocks per round (50 KB of LDS, 3 workgroups per CU; a second round serves
// blocks 13..15 of strongly diverging windows); r=4 windows need 13+ blocks even for a smooth flow
// field, so that variant exchanges all 15 at once (62 KB, 2 workgroups per CU).
template <int R> struct AmCfg {
  static constexpr int XBLK = (R <= 3) ? 12 : AM_MAXBLK;
  static constexpr int CP = 16 * XBLK + 4;      // pitch (== 4 mod 8: 2-way bank conflicts at worst)
  static constexpr int LDS_FLOATS = 64 * CP;    // staging ring and exchange share this memory
  static constexpr int MIN_WG = (R <= 3) ? 3 : 2;
};
constexpr int AM_MAXPOS = 640;           // largest workgroup box (positions) staged
constexpr int AM_MAXSLOT = (AM_MAXPOS + 63) / 64;  // DMA instructions per thread and stage

__device__ __forceinline__ int wave_min16(int v) {
  v = min(v, __shfl_xor(v, 1)); v = min(v, __shfl_xor(v, 2));
  v = min(v, __shfl_xor(v, 4)); v = min(v, __shfl_xor(v, 8));
  return v;
}
__device__ __forceinline__ int wave_max16(int v) {
  v = max(v, __shfl_xor(v, 1)); v = max(v, __shfl_xor(v, 2));
  v = max(v, __shfl_xor(v, 4)); v = max(v, __shfl_xor(v, 8));
  return v;
}

// Body shared by the two entry points below.  f1e / f2b: feature maps of this edge (channels last),
// cbase: its query coordinates, multiplied by cscale (1, or 2^-level for the pyramid entry point:
// exact), oute: its (2r+1)^2 output planes, tile: index of the 16x4 query tile.
template <int R, typename TI, typename TO>
__device__ __forceinline__ void altcorr_mfma_body(const TI* __restrict__ f1e, const TI* __restrict__ f2b,
                                                  const float* __restrict__ cbase, const float cscale,
                                                  TO* __restrict__ oute, const int tile, const int H1,
                                                  const int W1, const int H2, const int W2, const int C) {
  constexpr int RD = 2 * R + 1, NT = RD + 1;
  constexpr int AM_CH = AmIn<TI>::CH, AM_EPC = AmIn<TI>::EPC, AM_MAXSTAGE = AmIn<TI>::MAXSTAGE;
  constexpr bool HALF_IN = AM_EPC == 8;
  constexpr int AM_XBLK = AmCfg<R>::XBLK, AM_CP = AmCfg<R>::CP, AM_LDS_FLOATS = AmCfg<R>::LDS_FLOATS;
  static_assert(AM_MAXPOS * 16 <= AM_LDS_FLOATS, "one stage (64 bytes per position) of the largest box must fit");
  __shared__ __attribute__((aligned(16))) float lds[AM_LDS_FLOATS];
  __shared__ int sbox[4][4];  // per wave: x0, y0, width, height of its (clipped) box
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // keep wave-uniform values in SGPRs
  const int tiles_x = (W1 + AM_TX - 1) / AM_TX;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int H1W1 = H1 * W1;

  AMSTAMP(0);
  AMNOTE(12, __builtin_amdgcn_s_getreg((31 << 11) | 4));   // HW_ID: which CU this workgroup runs on
  AMNOTE(13, __builtin_amdgcn_s_getreg((31 << 11) | 20));  // XCC_ID
  // --- GEMM role: lane (row = lane & 15, g = lane >> 4); row = query (sy, sx) of the sub-tile
  const int row = lane & 15, g = lane >> 4;
  const int gqx = tx * AM_TX + 4 * wave + (row & 3), gqy = ty * AM_TY + (row >> 2);
  const bool gok = gqx < W1 && gqy < H1;
  const int gpix = gok ? gqy * W1 + gqx : 0;
  // coordinates first (the box and the staging plan wait for them), then the A fragments, which
  // stay in flight until the first MFMA
  float2 gc = *reinterpret_cast<const float2*>(cbase + 2 * gpix);
  f4 a_all[AM_MAXSTAGE];  // this lane's A fragments of all stages: channels 16 st + 4g .. +3 of its query
  {
    const TI* f1p = f1e + (size_t)gpix * C + AM_EPC * g;
#pragma unroll
    for (int st = 0; st < AM_MAXSTAGE; st++)
      a_all[st] = *reinterpret_cast<const f4*>(f1p + min(st * AM_CH, C - AM_CH));  // stages >= C/16 are never used
  }
  asm volatile("" : "+v"(gc.x), "+v"(gc.y));  // one 8-byte load ahead of the A loads (hipcc would sink half of it)
  const Bilin gbl = bilin_setup(gc.x * cscale, gc.y * cscale, R);
  {
    const bool hit = gok && gbl.x1 + NT > 0 && gbl.x1 < W2 && gbl.y1 + NT > 0 && gbl.y1 < H2;
    const int big = 0x3fffffff;
    const int x0 = max(wave_min16(hit ? gbl.x1 : big), 0), y0 = max(wave_min16(hit ? gbl.y1 : big), 0);
    const int x1 = min(wave_max16(hit ? gbl.x1 + NT : -big), W2), y1 = min(wave_max16(hit ? gbl.y1 + NT : -big), H2);
    if (lane == 0) {
      sbox[wave][0] = x0; sbox[wave][1] = y0;
      sbox[wave][2] = max(x1 - x0, 0); sbox[wave][3] = max(y1 - y0, 0);
    }
  }
  __syncthreads();
  AMSTAMP(1);
  int bx0 = 0x3fffffff, by0 = 0x3fffffff, bx1 = -0x3fffffff, by1 = -0x3fffffff;
  bool fits = true, two_rounds = false;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const int sx_ = __builtin_amdgcn_readfirstlane(sbox[w][0]), sy_ = __builtin_amdgcn_readfirstlane(sbox[w][1]);
    const int sw_ = __builtin_amdgcn_readfirstlane(sbox[w][2]), sh_ = __builtin_amdgcn_readfirstlane(sbox[w][3]);
    if (sw_ > 0 && sh_ > 0) {
      bx0 = min(bx0, sx_); by0 = min(by0, sy_);
      bx1 = max(bx1, sx_ + sw_); by1 = max(by1, sy_ + sh_);
      fits = fits && (sw_ * sh_ <= 16 * AM_MAXBLK);
      two_rounds = two_rounds || (sw_ * sh_ > 16 * AM_XBLK);
    }
  }
  const int BW = max(bx1 - bx0, 0), BH = max(by1 - by0, 0);
  const int npos = BW * BH;
  fits = fits && npos <= AM_MAXPOS;

  // --- output role: thread -> query (qx_l, qy_l) of the tile, outputs o = og, og+4, ...
  const int qx_l = tid & 15, qy_l = (tid >> 4) & 3, og = tid >> 6;
  const int oqx = tx * AM_TX + qx_l, oqy = ty * AM_TY + qy_l;
  const bool ook = oqx < W1 && oqy < H1;
  const int opix = ook ? oqy * W1 + oqx : 0;
#define AM_OUT_ROLE                                                                                    \
  const Bilin obl = bilin_setup(cbase[2 * opix] * cscale, cbase[2 * opix + 1] * cscale, R);            \
  const float wnw = f32_value(obl.dy * obl.dx), wne = f32_value(obl.dy * (1.0f - obl.dx));             \
  const float wsw = f32_value((1.0f - obl.dy) * obl.dx), wse = f32_value((1.0f - obl.dy) * (1.0f - obl.dx)); \
  TO* out = oute + opix;   /* weights: ak:119-122 */

  if (!fits) {  // incoherent tile: per-query direct evaluation
    if (!ook) return;
    AM_OUT_ROLE
    const TI* f1 = f1e + (size_t)opix * C;
    for (int o = og; o < RD * RD; o += 4) {
      const int ox = o / RD, oy = o % RD;
      float s4[4];
      for (int t = 0; t < 4; t++) {
        const int h2 = obl.y1 + oy + (t >> 1), w2 = obl.x1 + ox + (t & 1);
        float s = 0.f;
        if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
          const TI* f2 = f2b + ((size_t)h2 * W2 + w2) * C;
          if constexpr (HALF_IN) {
            for (int c = 0; c < C; c += 8) {
              const h8 u = *reinterpret_cast<const h8*>(f1 + c), v = *reinterpret_cast<const h8*>(f2 + c);
#pragma unroll
              for (int k = 0; k < 8; k++) s = fmaf((float)u[k], (float)v[k], s);
            }
          } else {
            for (int c = 0; c < C; c += 4) {
              const f4 u = *reinterpret_cast<const f4*>(f1 + c), v = *reinterpret_cast<const f4*>(f2 + c);
              s = fmaf(u[0], v[0], s); s = fmaf(u[1], v[1], s); s = fmaf(u[2], v[2], s); s = fmaf(u[3], v[3], s);
            }
          }
        }
        s4[t] = s;
      }
      float acc = s4[0] * wse;
      acc = acc + s4[1] * wsw;
      acc = acc + s4[2] * wne;
      acc = acc + s4[3] * wnw;
      am_store(out + (size_t)o * H1W1, acc);
    }
    return;
  }

  // --- staging plan: DMA instruction k covers positions 16k .. 16k+15 (lane>>2), 16-byte slot
  // lane&3 of each; this wave issues k = wave, wave+4, ...; source chunk = slot ^ ((P>>1)&3).
  // Stages live in a ring of `depth` LDS slots; stage s+depth-1 is requested as soon as stage s-1
  // has been consumed, so DMA latency overlaps the MFMA phases.  The DMAs are issued from inline
  // asm and retired with counted s_waitcnt vmcnt + raw s_barrier: hipcc would otherwise drain
  // every outstanding LDS-DMA (vmcnt(0)) in front of each LDS read and each __syncthreads().
  const int nk = (npos + 15) >> 4;
  const int slot_floats = nk * 256;                  // 16 positions x 64 bytes per DMA instruction
  const int nstage = C / AM_CH;                      // <= AM_MAXSTAGE (checked by the launcher)
  const int cw = nk > wave ? (nk - wave + 3) >> 2 : 0;  // DMA instructions of this wave per stage
  int depth = min(nstage, AM_LDS_FLOATS / max(slot_floats, 1));
  depth = max(1, min(depth, 56 / max((nk + 3) >> 2, 1)));   // vmcnt is a 6-bit counter
  const float rbw = 1.0f / (float)max(BW, 1);
  int soff[AM_MAXSLOT];  // float offset into fmap2 (without the channel base)
#pragma unroll
  for (int it = 0; it < AM_MAXSLOT; it++) {
    const int P = min(16 * (wave + 4 * it) + (lane >> 2), max(npos - 1, 0));  // pad lanes re-read the last row
    const int yy = (int)(((float)P + 0.5f) * rbw), xx = P - yy * BW;  // exact: P < 1024, BW < 1024
    soff[it] = ((by0 + yy) * W2 + (bx0 + xx)) * C + AM_EPC * ((lane & 3) ^ ((lane >> 3) & 3));
  }
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds);
  auto issue_stage = [&](int stage, int slot) {
    const TI* src = f2b + stage * AM_CH;
    const unsigned dst0 = lds_base + 4u * (unsigned)(slot * slot_floats);
#pragma unroll
    for (int it = 0; it < AM_MAXSLOT; it++) {
      const int k = wave + 4 * it;
      if (k < nk) {  // wave-uniform
        const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + 4u * (unsigned)(k * 256));
        const TI* gsrc = src + soff[it];
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
      }
    }
  };
#define AM_W(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
  auto wait_younger = [&](int n) {  // returns once at most n of this wave's DMAs are outstanding (rounded down)
    if (n >= 32) AM_W(32); else if (n >= 24) AM_W(24); else if (n >= 16) AM_W(16); else if (n >= 12) AM_W(12);
    else if (n >= 8) AM_W(8); else if (n >= 6) AM_W(6); else if (n >= 4) AM_W(4); else if (n >= 3) AM_W(3);
    else if (n >= 2) AM_W(2); else if (n >= 1) AM_W(1); else AM_W(0);
  };

  // --- this wave's column blocks: LDS offsets (floats) of the B operand rows
  const int sx0 = __builtin_amdgcn_readfirstlane(sbox[wave][0]), sy0 = __builtin_amdgcn_readfirstlane(sbox[wave][1]);
  const int sw = __builtin_amdgcn_readfirstlane(sbox[wave][2]), sh = __builtin_amdgcn_readfirstlane(sbox[wave][3]);
  const int nposw = sw * sh;
  const int nblk = (nposw + 15) >> 4;  // wave-uniform
  int boff[AM_MAXBLK];
  {
    const float rsw = 1.0f / (float)max(sw, 1);
    const int rbase = (sy0 - by0) * BW + (sx0 - bx0);
#pragma unroll
    for (int blk = 0; blk < AM_MAXBLK; blk++) {
      const int pp = 16 * blk + row;
      const int py = (int)(((float)pp + 0.5f) * rsw), px = pp - py * sw;
      const int r0 = (pp < nposw) ? rbase + py * BW + px : 0;
      boff[blk] = r0 * 16 + 4 * (g ^ ((r0 >> 1) & 3));
    }
  }
  f4 acc[AM_MAXBLK];
#pragma unroll
  for (int blk = 0; blk < AM_MAXBLK; blk++) acc[blk] = f4{0.f, 0.f, 0.f, 0.f};

  AMSTAMP(2);
  AMNOTE(8, nblk); AMNOTE(9, npos); AMNOTE(10, depth);
#pragma unroll
  for (int d = 0; d < AM_MAXSTAGE; d++)
    if (d < depth) issue_stage(d, d);
  // The A fragment loads were issued at kernel entry.  Retire them here, together with the first
  // `depth` stages (the compiler's vmcnt(0) for these registers also drains the DMAs it cannot
  // see), so that no compiler-generated vmcnt wait lands inside the pipelined loop.
#pragma unroll
  for (int st = 0; st < AM_MAXSTAGE; st++) asm volatile("" : "+v"(a_all[st]));
  int cur = 0;        // ring slot of stage st
  int issued = depth; // stages requested so far
#pragma unroll
  for (int st = 0; st < AM_MAXSTAGE; st++) {
    if (st < nstage) {
      if (depth == 1 && st >= 1) {  // box too large for two slots: no overlap
        __builtin_amdgcn_s_barrier();
        issue_stage(st, 0);
        issued++;
      }
      wait_younger((issued - 1 - st) * cw);
      __builtin_amdgcn_s_barrier();  // stage st visible to all waves; stage st-1 consumed by all
      if (st == 0) AMSTAMP(3);
      AMSTAMP(16 + 2 * st);
      if (depth > 1 && st >= 1 && issued < nstage) {
        issue_stage(issued, cur == 0 ? depth - 1 : cur - 1);
        issued++;
      }
      const f4 a = a_all[st];
      const float* bs = lds + cur * slot_floats;
#pragma unroll
      for (int grp = 0; grp < AM_MAXBLK / 3; grp++) {
        if (3 * grp < nblk) {  // wave-uniform; blocks past nblk multiply row 0 and are never read
          const f4 b0 = *reinterpret_cast<const f4*>(bs + boff[3 * grp]);
          const f4 b1 = *reinterpret_cast<const f4*>(bs + boff[3 * grp + 1]);
          const f4 b2 = *reinterpret_cast<const f4*>(bs + boff[3 * grp + 2]);
          f4 c0 = acc[3 * grp], c1 = acc[3 * grp + 1], c2 = acc[3 * grp + 2];
          if constexpr (HALF_IN) {  // lane group g holds k = 8g .. 8g+7 of both operands: one instruction per stage
            const h8 ah = __builtin_bit_cast(h8, a);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(h8, b0), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(h8, b1), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(h8, b2), c2, 0, 0, 0);
          } else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0[e], c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b1[e], c1, 0, 0, 0);
              c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b2[e], c2, 0, 0, 0);
            }
          }
          acc[3 * grp] = c0; acc[3 * grp + 1] = c1; acc[3 * grp + 2] = c2;
        }
      }
      AMSTAMP(17 + 2 * st);
      cur = (cur + 1 == depth) ? 0 : cur + 1;
    }
  }
#undef AM_W
  AMSTAMP(4);
  __syncthreads();
  // D exchange: dbuf[tile query][position of its wave's box]; lane holds D[query 4g+reg][16 blk + row].
  // All offsets past the per-lane base are compile-time constants (ds_write with immediates).
  float* dbuf = lds;
  {
    float* dl = dbuf + (wave * 16 + 4 * g) * AM_CP + row;
#pragma unroll
    for (int blk = 0; blk < AM_XBLK; blk++) {
      if (blk < nblk) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) dl[reg * AM_CP + 16 * blk] = acc[blk][reg];
      }
    }
  }
  __syncthreads();
  AMSTAMP(5);
  // Bilinear combine.  Wave og walks output columns ox = og, og+4, ... of all 64 queries of the
  // tile: the two tap rows of an output are shared with the next one, so a column costs
  // (2r+2) paired LDS reads for (2r+1) outputs.  Queries whose window leaves the box (plane
  // border) take the clamped + masked variant; the choice is wave-uniform.
  AM_OUT_ROLE
  const int wq = qx_l >> 2;
  const int osw = sbox[wq][2], osh = sbox[wq][3];
  const int rx = obl.x1 - sbox[wq][0], ry = obl.y1 - sbox[wq][1];  // window origin inside the wave's box
  const float* dq = dbuf + (wq * 16 + qy_l * 4 + (qx_l & 3)) * AM_CP;
  const bool inside = !ook || (rx >= 0 && ry >= 0 && rx + NT <= osw && ry + NT <= osh);
  if (AM_XBLK < AM_MAXBLK && two_rounds) {
    // some wave of this tile has more than AM_XBLK position blocks (strongly diverging windows):
    // the exchange buffer is used twice and every output sums the taps of both rounds
    constexpr int NOUT = (RD * RD + 3) / 4;
    float vacc[NOUT];
#pragma unroll
    for (int round = 0; round < 2; round++) {
      if (round == 1) {
        __syncthreads();  // round 0 fully read
        float* dl = dbuf + (wave * 16 + 4 * g) * AM_CP + row;
#pragma unroll
        for (int blk = AM_XBLK; blk < AM_MAXBLK; blk++) {
          if (blk < nblk) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) dl[reg * AM_CP + 16 * (blk - AM_XBLK)] = acc[blk][reg];
          }
        }
        __syncthreads();
      }
      const int lo = round * 16 * AM_XBLK;
#pragma unroll
      for (int m = 0; m < NOUT; m++) {
        const int o = og + 4 * m;
        const int ox = o / RD, oy = o % RD;
        float s4[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int yy = ry + oy + (t >> 1), xx = rx + ox + (t & 1);
          const int nn = yy * osw + xx - lo;
          const bool ok = yy >= 0 && yy < osh && xx >= 0 && xx < osw && nn >= 0 && nn < 16 * AM_XBLK;
          s4[t] = dq[ok ? nn : 0];
          s4[t] = ok ? s4[t] : 0.f;
        }
        float v = s4[0] * wse;
        v = v + s4[1] * wsw;
        v = v + s4[2] * wne;
        v = v + s4[3] * wnw;
        vacc[m] = (round == 0) ? v : vacc[m] + v;
      }
    }
    if (!ook) return;
#pragma unroll
    for (int m = 0; m < NOUT; m++)
      if (og + 4 * m < RD * RD) am_store(out + (size_t)(og + 4 * m) * H1W1, vacc[m]);
  } else if (__all(inside)) {
    AMNOTE(11, 1);
    if (!ook) return;
    for (int ox = og; ox < RD; ox += 4) {
      const float* dp = dq + ry * osw + rx + ox;
      TO* op = out + (size_t)ox * RD * H1W1;
      float t0 = dp[0], t1 = dp[1];
#pragma unroll
      for (int oy = 0; oy < RD; oy++) {
        dp += osw;
        const float u0 = dp[0], u1 = dp[1];
        float v = t0 * wse;   // tap (oy  , ox  )
        v = v + t1 * wsw;     // tap (oy  , ox+1)
        v = v + u0 * wne;     // tap (oy+1, ox  )
        v = v + u1 * wnw;     // tap (oy+1, ox+1)
        am_store(op + (size_t)oy * H1W1, v);
        t0 = u0; t1 = u1;
      }
    }
  } else {
    if (!ook) return;
    const int xmax = max(osw - 1, 0), ymax = max(osh - 1, 0);
    for (int ox = og; ox < RD; ox += 4) {
      const int xa = rx + ox, xb = xa + 1;
      const bool ina = xa >= 0 && xa < osw, inb = xb >= 0 && xb < osw;
      const int xac = min(max(xa, 0), xmax), xbc = min(max(xb, 0), xmax);
      TO* op = out + (size_t)ox * RD * H1W1;
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const int yy = ry + j;
        const bool iny = yy >= 0 && yy < osh;
        const float* dp = dq + min(max(yy, 0), ymax) * osw;
        float u0 = dp[xac], u1 = dp[xbc];
        u0 = (iny && ina) ? u0 : 0.f;
        u1 = (iny && inb) ? u1 : 0.f;
        if (j > 0) {
          float v = t0 * wse;
          v = v + t1 * wsw;
          v = v + u0 * wne;
          v = v + u1 * wnw;
          am_store(op + (size_t)(j - 1) * H1W1, v);
        }
        t0 = u0; t1 = u1;
      }
    }
  }
  AMSTAMP(6);
#undef AM_OUT_ROLE
}


// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
// linear id l runs virtual id (l % 8) * (total / 8) + l / 8: the tiles of one edge, whose fmap2 boxes
// overlap, then share one L2 instead of pulling every fmap2 row into all eight.
__device__ __forceinline__ unsigned am_virtual_id() {
  const unsigned total = gridDim.x * gridDim.y * gridDim.z;
  unsigned v = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if ((total & 7u) == 0) v = (v & 7u) * (total >> 3) + (v >> 3);
  return v;
}

// altcorr_forward (ak:27-142): grid (tiles, N, B)
template <int R, typename TI, typename TO>
__global__ __launch_bounds__(256, AmCfg<R>::MIN_WG) void altcorr_forward_mfma(const TI* __restrict__ fmap1,
                                                               const TI* __restrict__ fmap2,
                                                               const float* __restrict__ coords,
                                                               TO* __restrict__ corr, int N, int H1,
                                                               int W1, int H2, int W2, int C) {
  constexpr int RD = 2 * R + 1;
  const unsigned v = am_virtual_id();
  const int tile = (int)(v % gridDim.x);
  const unsigned e = v / gridDim.x;
  const int n = (int)(e % gridDim.y), b = (int)(e / gridDim.y);
  const size_t H1W1 = (size_t)H1 * W1;
  altcorr_mfma_body<R, TI, TO>(fmap1 + (size_t)b * H1W1 * C, fmap2 + (size_t)b * H2 * W2 * C,
                       coords + ((size_t)b * N + n) * H1W1 * 2, 1.0f,
                       corr + (((size_t)b * N + n) * RD * RD) * H1W1, tile, H1, W1, H2, W2, C);
}

// AltCorrBlock.corr_fn in one launch (modules/corr.py:105-125): edge e correlates frame ii[e] of
// pyramid level 0 with frame jj[e] of every level, coordinates scaled by 2^-level; no per-edge
// copies of the feature maps (`pyramid[i][:, jj]`).  grid (tiles, levels, E); output
// [E, levels*(2r+1)^2, H, W] = torch.cat of the per-level results for one coordinate set.
struct AltPyramid {
  const void* level[4];  // [frames, H >> l, W >> l, C] fp32 or fp16, channels last
};
template <int R, typename TI>
__global__ __launch_bounds__(256, AmCfg<R>::MIN_WG) void altcorr_pyramid_mfma(AltPyramid pyr,
                                                               const int64_t* __restrict__ ii,
                                                               const int64_t* __restrict__ jj,
                                                               const float* __restrict__ coords,
                                                               float* __restrict__ corr, int frames, int H1,
                                                               int W1, int C) {
  constexpr int RD = 2 * R + 1;
  const unsigned v = am_virtual_id();
  const int tile = (int)(v % gridDim.x);
  const unsigned q = v / gridDim.x;
  const int lvl = (int)(q % gridDim.y), e = (int)(q / gridDim.y);
  const int H2 = H1 >> lvl, W2 = W1 >> lvl;
  const size_t H1W1 = (size_t)H1 * W1;
  const int64_t fi = ii[e], fj = jj[e];
  float* oute = corr + ((size_t)e * gridDim.y + lvl) * RD * RD * H1W1;
  if (fi < 0 || fi >= frames || fj < 0 || fj >= frames || H2 <= 0 || W2 <= 0) {  // contract violation: zeros
    const int tiles_x = (W1 + AM_TX - 1) / AM_TX;
    const int qx = (tile % tiles_x) * AM_TX + (threadIdx.x & 15), qy = (tile / tiles_x) * AM_TY + ((threadIdx.x >> 4) & 3);
    if (qx < W1 && qy < H1)
      for (int o = threadIdx.x >> 6; o < RD * RD; o += 4) oute[(size_t)o * H1W1 + qy * W1 + qx] = 0.f;
    return;
  }
  const TI* lp = static_cast<const TI*>(lvl == 0 ? pyr.level[0] : (lvl == 1 ? pyr.level[1] : (lvl == 2 ? pyr.level[2] : pyr.level[3])));
  altcorr_mfma_body<R, TI, float>(static_cast<const TI*>(pyr.level[0]) + (size_t)fi * H1W1 * C, lp + (size_t)fj * H2 * W2 * C,
                       coords + (size_t)e * H1W1 * 2, 1.0f / (float)(1 << lvl), oute, tile, H1, W1, H2, W2, C);
}

// ---- alt-corr on the f16 matrix cores: one autonomous wave per 16 queries, all pyramid levels --------------------
// The half pyramid of the SLAM path (modules/corr.py:92-104 on video.fmaps, a torch.half buffer).  With
// v_mfma_f32_16x16x32_f16 the box GEMM costs 1/16 of its fp32 time, and what bounded the workgroup-cooperative
// kernel above -- four barriers per K stage, a staging plan for the whole workgroup, phases that wait for each
// other -- becomes the whole cost (stamps: 20 k cycles per workgroup, 2 k of them MFMA).  So here every wave is its
// own workgroup and never meets a barrier:
//   * 16 queries (a 4x4 sub-tile) against the bounding box of THEIR windows, for all levels in turn: the level-0
//     rows of the queries (fmap1 is level 0 at every level, corr.py:113) are loaded once as MFMA B operands;
//   * fmap2 positions are the A operand.  The operand layout wants lane (row = l & 15, k-group = l >> 4), which read
//     straight from memory runs at 14 B/clk/CU (tools/micro/ldmap.hip: the four lanes of a 64-byte run are 16 lanes
//     apart, every lane its own request); so blocks of 16 positions x 128 channels (4 KB) arrive by LDS-DMA with
//     four ADJACENT lanes per 64-byte run (47 B/clk/CU) into a private ring of three slots, XOR-swizzled on the
//     source side, and leave it as ds_read_b128 operands -- K is complete per block, so a block is four MFMAs
//     into its own accumulator;
//   * accumulators stay in registers until the last block, then go to LDS as D[query][position] (the lane holds
//     four consecutive positions of ONE query: one ds_write_b128 per block) over the ring's memory, and the wave
//     combines its own 16 queries: lane (query, column group) walks output columns, two FMAs per tap pair;
//   * LDS-DMA completion is counted per block with s_waitcnt vmcnt (loads, DMAs and the previous level's output
//     stores retire in order); LDS accesses of one wave execute in order, so no other synchronisation exists.
// 15.6 KB of LDS per wave: ten waves per CU hide each other's latencies.
// min / max over the 16 lanes of a DPP row in four VALU instructions (no LDS round trips: __shfl_xor goes through
// ds_bpermute, ~100 cycles each, 16 of them per level): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ int row_min16(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
  return v;
}
__device__ __forceinline__ int row_max16(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
  return v;
}
template <int R> struct AwCfg {
  static constexpr int MAXBLK = (R <= 3) ? 15 : 20;   // 16-position blocks per wave box (r=4 windows are 10x10)
  static constexpr int CP = 16 * MAXBLK + 4;          // D pitch in floats (== 4 mod 32)
  static constexpr int LDS_FLOATS = 16 * CP;
  static constexpr int NRING = 3;                     // 4-KB block slots of the staging ring (inside the D buffer)
};
struct AwArgs {
  const __half* f1;        // level-0 maps of the query frames [frames | B, H1, W1, C]
  const __half* f2[4];     // per level [frames | B, H2, W2, C]
  const int64_t* ii;       // frame of the queries per edge; null: the batch index (altcorr_forward)
  const int64_t* jj;
  const float* coords;     // [E | B, N, H1, W1, 2]
  void* corr;              // [E | B, N, levels, (2r+1)^2, H1, W1]
  int frames, N, H1, W1, C, nlevels;
  int H2[4], W2[4];
  float cscale[4];
};

template <int R, int NST, typename TO>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void altcorr_wave_f16(const AwArgs a) {
  constexpr int RD = 2 * R + 1, NT = RD + 1;
  constexpr int MAXBLK = AwCfg<R>::MAXBLK, CP = AwCfg<R>::CP, NRING = AwCfg<R>::NRING;
  static_assert(NRING * 1024 <= AwCfg<R>::LDS_FLOATS, "ring inside the D buffer");
  __shared__ __attribute__((aligned(16))) float lds[AwCfg<R>::LDS_FLOATS];
  const int lane = threadIdx.x;
  const int q = lane & 15, g = lane >> 4;
  const int C = a.C, H1 = a.H1, W1 = a.W1, H1W1 = H1 * W1;
  // XCD-aware order (am_virtual_id): the sub-tiles of one edge share an L2
  const unsigned v = am_virtual_id();
  const int tiles_x = (W1 + 3) >> 2;
  const int tile = (int)(v % gridDim.x);
  const unsigned bn = v / gridDim.x;
  const int n = (int)(bn % gridDim.y), e = (int)(bn / gridDim.y);
  const int qx = (tile % tiles_x) * 4 + (q & 3), qy = (tile / tiles_x) * 4 + (q >> 2);
  const bool ok = qx < W1 && qy < H1;
  const int pix = ok ? qy * W1 + qx : 0;
  int64_t fi = e, fj = e;
  if (a.ii) { fi = a.ii[e]; fj = a.jj[e]; }
  const bool frames_ok = fi >= 0 && fi < a.frames && fj >= 0 && fj < a.frames;
  TO* const out0 = static_cast<TO*>(a.corr) + ((size_t)e * a.N + n) * a.nlevels * (RD * RD) * (size_t)H1W1 + pix;
  if (!frames_ok) {  // contract violation: zeros
    if (ok)
      for (int o = g; o < a.nlevels * RD * RD; o += 4) am_store(out0 + (size_t)o * H1W1, 0.f);
    return;
  }
  const float2 gc = *reinterpret_cast<const float2*>(a.coords + (((size_t)e * a.N + n) * H1W1 + pix) * 2);
  // B operands: this query's level-0 row, k-group g of k-step s = channels 32 s + 8 g .. + 7
  AMSTAMP(0);
  f4 fq[NST];   // NST = C / 32 k-steps
  {
    const __half* f1p = a.f1 + ((size_t)fi * H1W1 + pix) * C + 8 * g;
#pragma unroll
    for (int s = 0; s < NST; s++) fq[s] = *reinterpret_cast<const f4*>(f1p + 32 * s);
  }
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds);
  // A-operand address inside a ring slot: k-step image s = [16 positions][64 B], slot c of position i holds chunk
  // c ^ ((i >> 1) & 3) (swizzled on the source side): conflict-free ds_read_b128
  const int aoff = q * 16 + 4 * (g ^ ((q >> 1) & 3));                 // floats
  // DMA role: lane -> position lane >> 2 of the block, 16-byte slot lane & 3 of its 64-byte k-step run
  const int dpos = lane >> 2;
  const int dchunk = 8 * ((lane & 3) ^ ((lane >> 3) & 3));            // halves
#pragma unroll
  for (int s = 0; s < NST; s++) asm volatile("" : "+v"(fq[s]));       // retire the operand loads before any LDS-DMA
  AMSTAMP(1);

  for (int lvl = 0; lvl < a.nlevels; lvl++) {
    const int H2 = a.H2[lvl], W2 = a.W2[lvl];
    const __half* f2b = a.f2[lvl] + (size_t)fj * H2 * W2 * C;
    TO* out = out0 + (size_t)lvl * (RD * RD) * H1W1;
    const Bilin bl = bilin_setup(gc.x * a.cscale[lvl], gc.y * a.cscale[lvl], R);
    const float wnw = f32_value(bl.dy * bl.dx), wne = f32_value(bl.dy * (1.0f - bl.dx));      // ak:119-122
    const float wsw = f32_value((1.0f - bl.dy) * bl.dx), wse = f32_value((1.0f - bl.dy) * (1.0f - bl.dx));
    const bool hit = ok && bl.x1 + NT > 0 && bl.x1 < W2 && bl.y1 + NT > 0 && bl.y1 < H2;
    const int big = 0x3fffffff;
    const int x0 = __builtin_amdgcn_readfirstlane(max(row_min16(hit ? bl.x1 : big), 0));
    const int y0 = __builtin_amdgcn_readfirstlane(max(row_min16(hit ? bl.y1 : big), 0));
    const int x1 = __builtin_amdgcn_readfirstlane(min(row_max16(hit ? bl.x1 + NT : -big), W2));
    const int y1 = __builtin_amdgcn_readfirstlane(min(row_max16(hit ? bl.y1 + NT : -big), H2));
    const int sw = max(x1 - x0, 0), sh = max(y1 - y0, 0);
    const int nposw = sw * sh;
    const int nblk = (nposw + 15) >> 4;    // wave-uniform
    if (nposw > 16 * MAXBLK) {             // incoherent coordinates: per-query direct evaluation
      if (ok) {
        const __half* f1 = a.f1 + ((size_t)fi * H1W1 + pix) * C;
        for (int o = g; o < RD * RD; o += 4) {
          const int ox = o / RD, oy = o % RD;
          float s4[4];
          for (int t = 0; t < 4; t++) {
            const int h2 = bl.y1 + oy + (t >> 1), w2 = bl.x1 + ox + (t & 1);
            float sacc = 0.f;
            if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
              const __half* f2 = f2b + ((size_t)h2 * W2 + w2) * C;
              for (int c = 0; c < C; c += 8) {
                const h8 u = *reinterpret_cast<const h8*>(f1 + c), w = *reinterpret_cast<const h8*>(f2 + c);
#pragma unroll
                for (int k = 0; k < 8; k++) sacc = fmaf((float)u[k], (float)w[k], sacc);
              }
            }
            s4[t] = sacc;
          }
          float acc = s4[0] * wse;
          acc = fmaf(s4[1], wsw, acc);
          acc = fmaf(s4[2], wne, acc);
          acc = fmaf(s4[3], wnw, acc);
          am_store(out + (size_t)o * H1W1, acc);
        }
      }
      continue;
    }
    // ---- box GEMM: D[position][query], block by block through the ring
    const float rsw = 1.0f / (float)max(sw, 1);
    // NST LDS-DMA instructions per block: the k-step images of block blk into slot blk % NRING.  One position decode
    // and one 32-bit byte offset per lane and block (scalar base, the level's map is < 2 GB); the k-step moves by the
    // instruction's immediate offset, which applies to the global address AND to the LDS destination
    // (tools/micro/glds_off.hip), so M0 = image base - 64 s.
    auto issue_block = [&](int blk) {
      const int P = min(16 * blk + dpos, max(nposw - 1, 0));             // pad lanes re-read the last position
      const int yy = (int)(((float)P + 0.5f) * rsw), xx = P - yy * sw;   // exact: P < 1024
      const unsigned voff = 2u * (unsigned)(((y0 + yy) * W2 + (x0 + xx)) * C + dchunk);
      const unsigned m0 = lds_base + 4096u * (unsigned)(blk % NRING);
      unsigned keep;
      if constexpr (NST == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "s_add_u32 m0, %3, 960\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:64\n\t"
                     "s_add_u32 m0, %3, 1920\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:128\n\t"
                     "s_add_u32 m0, %3, 2880\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:192\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(f2b), "s"(m0) : "memory", "scc");
      else if constexpr (NST == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "s_add_u32 m0, %3, 960\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:64\n\t"
                     "s_add_u32 m0, %3, 1920\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:128\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(f2b), "s"(m0) : "memory", "scc");
      else if constexpr (NST == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "s_add_u32 m0, %3, 960\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:64\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(f2b), "s"(m0) : "memory", "scc");
      else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(f2b), "s"(m0) : "memory", "scc");
    };
    f4 acc[MAXBLK];
#pragma unroll
    for (int blk = 0; blk < MAXBLK; blk++) acc[blk] = f4{0.f, 0.f, 0.f, 0.f};
    // the combine of the previous level has read the D buffer (LDS executes a wave's accesses in order, and its
    // results were consumed by the stores issued above): the ring may be refilled
    AMSTAMP(2 + 7 * lvl);
    AMNOTE(8 + 7 * lvl, nblk);
    static_assert(NRING == 3, "the wait ladder below assumes two younger blocks at most");
#pragma unroll
    for (int d = 0; d < NRING; d++)
      if (d < nblk) issue_block(d);
    AMSTAMP(3 + 7 * lvl);
#pragma unroll
    for (int blk = 0; blk < MAXBLK; blk++) {
      if (blk < nblk) {   // wave-uniform
        // blocks younger than blk still in flight: min(NRING-1, nblk-1-blk), NST DMAs each
        if (blk + 2 < nblk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NST) : "memory");
        else if (blk + 1 < nblk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (blk == 0) AMSTAMP(4 + 7 * lvl);
        const float* slot = lds + 1024 * (blk % NRING) + aoff;
        f4 pa[NST];
#pragma unroll
        for (int s = 0; s < NST; s++) pa[s] = *reinterpret_cast<const f4*>(slot + 256 * s);
        f4 c = acc[blk];
#pragma unroll
        for (int s = 0; s < NST; s++)
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, pa[s]), __builtin_bit_cast(h8, fq[s]), c, 0, 0, 0);
        acc[blk] = c;
        if (blk + NRING < nblk) {           // refill this block's slot: its operand reads have returned
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          issue_block(blk + NRING);
        }
      }
    }
    // ---- D -> LDS (over the ring: every block has been read): lane holds positions 16 blk + 4 g .. + 3 of query q
    AMSTAMP(5 + 7 * lvl);
    __builtin_amdgcn_wave_barrier();
    {
      float* dl = lds + q * CP + 4 * g;
#pragma unroll
      for (int blk = 0; blk < MAXBLK; blk++)
        if (blk < nblk) *reinterpret_cast<f4*>(dl + 16 * blk) = acc[blk];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    AMSTAMP(6 + 7 * lvl);
    // ---- bilinear combine of the wave's own 16 queries: lane (query q, column group g) walks output columns
    const int rx = bl.x1 - x0, ry = bl.y1 - y0;     // window origin inside the box
    const float* dq = lds + q * CP;
    const bool inside = !ok || (rx >= 0 && ry >= 0 && rx + NT <= sw && ry + NT <= sh);
    if (__all(inside)) {
      if (ok) {
        for (int ox = g; ox < RD; ox += 4) {
          const float* dp = dq + ry * sw + rx + ox;
          TO* op = out + (size_t)ox * RD * H1W1;
          float t0 = dp[0], t1 = dp[1];
#pragma unroll
          for (int oy = 0; oy < RD; oy++) {
            dp += sw;
            const float u0 = dp[0], u1 = dp[1];
            float vv = t0 * wse;            // tap (oy  , ox  )
            vv = fmaf(t1, wsw, vv);         // tap (oy  , ox+1)
            vv = fmaf(u0, wne, vv);         // tap (oy+1, ox  )
            vv = fmaf(u1, wnw, vv);         // tap (oy+1, ox+1)
            am_store(op + (size_t)oy * H1W1, vv);
            t0 = u0; t1 = u1;
          }
        }
      }
    } else if (ok) {
      const int xmax = max(sw - 1, 0), ymax = max(sh - 1, 0);
      for (int ox = g; ox < RD; ox += 4) {
        const int xa = rx + ox, xb = xa + 1;
        const bool ina = xa >= 0 && xa < sw, inb = xb >= 0 && xb < sw;
        const int xac = min(max(xa, 0), xmax), xbc = min(max(xb, 0), xmax);
        TO* op = out + (size_t)ox * RD * H1W1;
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int j = 0; j < NT; j++) {
          const int yy = ry + j;
          const bool iny = yy >= 0 && yy < sh;
          const float* dp = dq + min(max(yy, 0), ymax) * sw;
          float u0 = dp[xac], u1 = dp[xbc];
          u0 = (iny && ina) ? u0 : 0.f;
          u1 = (iny && inb) ? u1 : 0.f;
          if (j > 0) {
            float vv = t0 * wse;
            vv = fmaf(t1, wsw, vv);
            vv = fmaf(u0, wne, vv);
            vv = fmaf(u1, wnw, vv);
            am_store(op + (size_t)(j - 1) * H1W1, vv);
          }
          t0 = u0; t1 = u1;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    AMSTAMP(7 + 7 * lvl);
  }
}

template <typename TO>
static int launch_altcorr_wave_f16(const AwArgs& a, int r, int units, hipStream_t s) {
  const int tiles = ((a.W1 + 3) / 4) * ((a.H1 + 3) / 4);
  dim3 grid(tiles, a.N, units), block(64);
#define AW_LAUNCH(RR, NN) hipLaunchKernelGGL((altcorr_wave_f16<RR, NN, TO>), grid, block, 0, s, a)
  const int nst = a.C / 32;
  if (r == 3) {
    if (nst == 4) AW_LAUNCH(3, 4); else if (nst == 3) AW_LAUNCH(3, 3); else if (nst == 2) AW_LAUNCH(3, 2); else AW_LAUNCH(3, 1);
  } else {
    if (nst == 4) AW_LAUNCH(4, 4); else if (nst == 3) AW_LAUNCH(4, 3); else if (nst == 2) AW_LAUNCH(4, 2); else AW_LAUNCH(4, 1);
  }
#undef AW_LAUNCH
  return 0;
}

#ifdef AM_STAMPS
extern "C" int droid_debug_am_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_am_stamps), sizeof(unsigned long long) * 768 * 4 * 32);
}
#endif

template <typename T>
static int altcorr_forward_t(const void* f1, const void* f2, const float* coords, void* corr, int B,
                             int N, int H1, int W1, int H2, int W2, int C, int r, hipStream_t s) {
  const int HW = H1 * W1;
  dim3 grid((HW * 16 + 255) / 256, N, B), block(256);
  hipLaunchKernelGGL((altcorr_forward_generic<T>), grid, block, 0, s, static_cast<const T*>(f1),
                     static_cast<const T*>(f2), coords, static_cast<T*>(corr), N, HW, H2, W2, C, r);
  return 0;
}

int launch_altcorr_forward(const void* f1, const void* f2, const float* coords, void* corr, int B,
                           int N, int H1, int W1, int H2, int W2, int C, int r, int dtype,
                           hipStream_t s) {
  if (B > 65535 || N > 65535) return DROID_E_ARG;
  if (dtype == DROID_F32 && (C % AmIn<float>::CH) == 0 && C <= AmIn<float>::CH * AmIn<float>::MAXSTAGE && (r == 3 || r == 4) &&
      (long)H2 * W2 * C < (1l << 30)) {
    const int tiles = ((W1 + AM_TX - 1) / AM_TX) * ((H1 + AM_TY - 1) / AM_TY);
    dim3 grid(tiles, N, B), block(256);
    if (r == 3)
      hipLaunchKernelGGL((altcorr_forward_mfma<3, float, float>), grid, block, 0, s, static_cast<const float*>(f1),
                         static_cast<const float*>(f2), coords, static_cast<float*>(corr), N, H1, W1, H2, W2, C);
    else
      hipLaunchKernelGGL((altcorr_forward_mfma<4, float, float>), grid, block, 0, s, static_cast<const float*>(f1),
                         static_cast<const float*>(f2), coords, static_cast<float*>(corr), N, H1, W1, H2, W2, C);
    return 0;
  }
  // half maps (altcorr_kernel.cu:308 dispatches half): f16 matrix cores, fp32 accumulation, half output
  if (dtype == DROID_F16 && (C % 32) == 0 && C <= 128 && (r == 3 || r == 4) && (long)H2 * W2 * C < (1l << 30) &&
      (long)H1 * W1 * C < (1l << 30) && !getenv("DROID_ALTCORR_F16_WG")) {
    AwArgs a{};
    a.f1 = static_cast<const __half*>(f1);
    a.f2[0] = static_cast<const __half*>(f2);
    a.coords = coords; a.corr = corr;
    a.frames = B; a.N = N; a.H1 = H1; a.W1 = W1; a.C = C; a.nlevels = 1;
    a.H2[0] = H2; a.W2[0] = W2; a.cscale[0] = 1.0f;
    return launch_altcorr_wave_f16<__half>(a, r, B, s);
  }
  if (dtype == DROID_F16 && (C % AmIn<__half>::CH) == 0 && C <= AmIn<__half>::CH * AmIn<__half>::MAXSTAGE && (r == 3 || r == 4) &&
      (long)H2 * W2 * C < (1l << 30)) {
    const int tiles = ((W1 + AM_TX - 1) / AM_TX) * ((H1 + AM_TY - 1) / AM_TY);
    dim3 grid(tiles, N, B), block(256);
    if (r == 3)
      hipLaunchKernelGGL((altcorr_forward_mfma<3, __half, __half>), grid, block, 0, s, static_cast<const __half*>(f1),
                         static_cast<const __half*>(f2), coords, static_cast<__half*>(corr), N, H1, W1, H2, W2, C);
    else
      hipLaunchKernelGGL((altcorr_forward_mfma<4, __half, __half>), grid, block, 0, s, static_cast<const __half*>(f1),
                         static_cast<const __half*>(f2), coords, static_cast<__half*>(corr), N, H1, W1, H2, W2, C);
    return 0;
  }
  if (dtype == DROID_F32 && (C % ALT_CH) == 0 && (r == 3 || r == 4)) {
    const int tiles = ((W1 + ALT_TQ - 1) / ALT_TQ) * ((H1 + ALT_TQ - 1) / ALT_TQ);
    dim3 grid(tiles, N, B), block(ALT_THREADS);
    if (r == 3)
      hipLaunchKernelGGL((altcorr_forward_tiled<3>), grid, block, 0, s, static_cast<const float*>(f1),
                         static_cast<const float*>(f2), coords, static_cast<float*>(corr), N, H1, W1, H2, W2, C);
    else
      hipLaunchKernelGGL((altcorr_forward_tiled<4>), grid, block, 0, s, static_cast<const float*>(f1),
                         static_cast<const float*>(f2), coords, static_cast<float*>(corr), N, H1, W1, H2, W2, C);
    return 0;
  }
  switch (dtype) {
    case DROID_F16: return altcorr_forward_t<__half>(f1, f2, coords, corr, B, N, H1, W1, H2, W2, C, r, s);
    case DROID_F32: return altcorr_forward_t<float>(f1, f2, coords, corr, B, N, H1, W1, H2, W2, C, r, s);
    case DROID_F64: return altcorr_forward_t<double>(f1, f2, coords, corr, B, N, H1, W1, H2, W2, C, r, s);
  }
  return DROID_E_ARG;
}

// fused AltCorrBlock.corr_fn: see altcorr_pyramid_mfma.  Returns DROID_E_ARG for configurations the
// matrix-core path does not cover (the caller then runs altcorr_forward per level).
int launch_altcorr_pyramid_forward(const void* const* levels_dev, const int64_t* ii, const int64_t* jj,
                                   const float* coords, float* corr, int E, int frames, int H, int W, int C,
                                   int r, int nlevels, int dtype, hipStream_t s) {
  if (dtype != DROID_F32 && dtype != DROID_F16) return DROID_E_ARG;
  const int ch = dtype == DROID_F16 ? AmIn<__half>::CH : AmIn<float>::CH;
  if (nlevels < 1 || nlevels > 4 || (r != 3 && r != 4) || (C % ch) != 0 || C > 128) return DROID_E_ARG;
  if ((H >> (nlevels - 1)) < 1 || (W >> (nlevels - 1)) < 1 || (long)H * W * C >= (1l << 30)) return DROID_E_ARG;
  if ((long)E * nlevels > 65535) return DROID_E_ARG;
  AltPyramid pyr;
  for (int l = 0; l < 4; l++) pyr.level[l] = levels_dev[l < nlevels ? l : nlevels - 1];
  const int tiles = ((W + AM_TX - 1) / AM_TX) * ((H + AM_TY - 1) / AM_TY);
  dim3 grid(tiles, nlevels, E), block(256);
  if (dtype == DROID_F16 && !getenv("DROID_ALTCORR_F16_WG")) {
    AwArgs a{};
    a.f1 = static_cast<const __half*>(levels_dev[0]);
    for (int l = 0; l < nlevels; l++) {
      a.f2[l] = static_cast<const __half*>(levels_dev[l]);
      a.H2[l] = H >> l; a.W2[l] = W >> l; a.cscale[l] = 1.0f / (float)(1 << l);
    }
    a.ii = ii; a.jj = jj; a.coords = coords; a.corr = corr;
    a.frames = frames; a.N = 1; a.H1 = H; a.W1 = W; a.C = C; a.nlevels = nlevels;
    return launch_altcorr_wave_f16<float>(a, r, E, s);
  }
  if (dtype == DROID_F16) {
    if (r == 3)
      hipLaunchKernelGGL((altcorr_pyramid_mfma<3, __half>), grid, block, 0, s, pyr, ii, jj, coords, corr, frames, H, W, C);
    else
      hipLaunchKernelGGL((altcorr_pyramid_mfma<4, __half>), grid, block, 0, s, pyr, ii, jj, coords, corr, frames, H, W, C);
  } else if (r == 3)
    hipLaunchKernelGGL((altcorr_pyramid_mfma<3, float>), grid, block, 0, s, pyr, ii, jj, coords, corr, frames, H, W, C);
  else
    hipLaunchKernelGGL((altcorr_pyramid_mfma<4, float>), grid, block, 0, s, pyr, ii, jj, coords, corr, frames, H, W, C);
  return 0;
}

// ---- altcorr_backward (ak:152-286), fp32, atomics into pre-zeroed gradients ----------------
__global__ __launch_bounds__(256) void altcorr_backward_kernel(
    const float* __restrict__ fmap1, const float* __restrict__ fmap2,
    const float* __restrict__ coords, const float* __restrict__ corr_grad,
    float* __restrict__ fmap1_grad, float* __restrict__ fmap2_grad, int N, int H1W1, int H2, int W2,
    int C, int r) {
  const int rd = 2 * r + 1;
  const int sub = threadIdx.x & 15;
  const int q = (blockIdx.x * 256 + threadIdx.x) >> 4;
  const int n = blockIdx.y, b = blockIdx.z;
  if (q >= H1W1) return;
  const float* cp = coords + (((size_t)b * N + n) * H1W1 + q) * 2;
  const Bilin bl = bilin_setup(cp[0], cp[1], r);
  const float* f1 = fmap1 + ((size_t)b * H1W1 + q) * C;
  float* g1 = fmap1_grad + ((size_t)b * H1W1 + q) * C;
  const float* cg = corr_grad + (((size_t)b * N + n) * rd * rd) * H1W1 + q;
  // the query's own feature gradient is summed in registers over all taps (one atomic per channel at the end,
  // not one per tap and channel); channels c = sub + 16 m, at most 16 of them per thread (C <= 256), else atomics
  constexpr int G1MAX = 16;
  const bool g1_regs = C <= 16 * G1MAX;
  float acc1[G1MAX];
#pragma unroll
  for (int m = 0; m < G1MAX; m++) acc1[m] = 0.f;
  for (int iy = 0; iy < rd + 1; iy++)
    for (int ix = 0; ix < rd + 1; ix++) {
      const int h2 = bl.y1 + iy, w2 = bl.x1 + ix;
      if (h2 < 0 || h2 >= H2 || w2 < 0 || w2 >= W2) continue;
      float g = 0.f;  // ak:228-246
      if (iy > 0 && ix > 0) g += cg[(size_t)((iy - 1) + rd * (ix - 1)) * H1W1] * (bl.dy * bl.dx);
      if (iy > 0 && ix < rd) g += cg[(size_t)((iy - 1) + rd * ix) * H1W1] * (bl.dy * (1.f - bl.dx));
      if (iy < rd && ix > 0) g += cg[(size_t)(iy + rd * (ix - 1)) * H1W1] * ((1.f - bl.dy) * bl.dx);
      if (iy < rd && ix < rd) g += cg[(size_t)(iy + rd * ix) * H1W1] * ((1.f - bl.dy) * (1.f - bl.dx));
      const float* f2 = fmap2 + (((size_t)b * H2 + h2) * W2 + w2) * C;
      float* g2 = fmap2_grad + (((size_t)b * H2 + h2) * W2 + w2) * C;
      if (g1_regs) {
#pragma unroll
        for (int m = 0; m < G1MAX; m++) {
          const int c = sub + 16 * m;
          if (c < C) {
            acc1[m] += g * f2[c];
            atomicAdd(&g2[c], g * f1[c]);
          }
        }
      } else {
        for (int c = sub; c < C; c += 16) {
          atomicAdd(&g1[c], g * f2[c]);
          atomicAdd(&g2[c], g * f1[c]);
        }
      }
    }
  if (g1_regs) {
#pragma unroll
    for (int m = 0; m < G1MAX; m++) {
      const int c = sub + 16 * m;
      if (c < C) atomicAdd(&g1[c], acc1[m]);
    }
  }
}

// ---- altcorr_backward, tiled ------------------------------------------------------------------------------------
// The per-tap kernel above issues one global float atomic per (query, tap, channel): 805 M atomics for 32 edges at
// 48x64 x 128 channels, 2.9 ms at the chip's ~0.3 T atomics/s.  Here one workgroup owns an 8x8 tile of queries of
// one (edge, coordinate set).  The windows of neighbouring queries overlap almost completely, so the fmap2 gradient
// is formed per POSITION of the tile's bounding box as a gather over the queries whose window covers it:
//   fmap2_grad[pos] += sum_{q in tile, pos in window(q)} g(q, tap(q, pos)) fmap1[q]
// with the list of (query, tap) pairs of every position built once per workgroup in LDS (11 of the 64 queries on
// average at r = 3, all 64 in the middle of the box) and reused for all channels; a position leaves the workgroup as ONE global atomic per channel
// (~360 positions for 64 x 64 taps: 11x fewer atomics).  Accumulating per position with LDS atomics instead
// (ds_add_f32 from query-owning threads) was measured first and is slower than the global atomics it replaces
// (4.1 ms: LDS float atomics retire a few lanes per cycle).  The query's own gradient (fmap1_grad) is a register sum
// over its taps.  The 64 bilinear-combined tap gradients g (ak:228-246) of every query are formed once, in LDS.
constexpr int ABT = 8;              // tile is ABT x ABT queries
constexpr int AB_CH = 16;           // channels per pass (4 per thread)
constexpr int AB_MAXPOS = 448;      // largest box handled with hit lists (21 x 21 at r = 4); bigger boxes take per-tap atomics
constexpr int AB_HMAX = ABT * ABT;  // a position in the middle of the box is covered by every query of the tile

template <int R>
__global__ __launch_bounds__(256) void altcorr_backward_tiled(
    const float* __restrict__ fmap1, const float* __restrict__ fmap2, const float* __restrict__ coords,
    const float* __restrict__ corr_grad, float* __restrict__ fmap1_grad, float* __restrict__ fmap2_grad, int N,
    int H1, int W1, int H2, int W2, int C) {
  constexpr int RD = 2 * R + 1, NT = RD + 1, NTAP = NT * NT;
  __shared__ unsigned short hits[AB_MAXPOS][AB_HMAX + 2];   // +2: rows of 33 dwords, conflict-free across positions
  __shared__ unsigned char hcnt[AB_MAXPOS];
  __shared__ float gq[ABT * ABT][NTAP + 1];
  __shared__ __attribute__((aligned(16))) float f1s[ABT * ABT][AB_CH + 4];  // pitch 20: 16-byte reads of 8 queries cover all banks
  __shared__ int qorg[ABT * ABT][2];
  __shared__ int bbox[4];
  const int tid = threadIdx.x;
  const int qq = tid >> 2, cq = tid & 3;
  const int tiles_x = (W1 + ABT - 1) / ABT;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int n = blockIdx.y, b = blockIdx.z;
  const int qx = tx * ABT + (qq & (ABT - 1)), qy = ty * ABT + (qq / ABT);
  const bool qok = qx < W1 && qy < H1;
  const int H1W1 = H1 * W1;
  const int pix = qok ? qy * W1 + qx : 0;
  const float* cp = coords + (((size_t)b * N + n) * H1W1 + pix) * 2;
  const Bilin bl = bilin_setup(cp[0], cp[1], R);
  if (tid < 4) bbox[tid] = (tid < 2) ? 0x7fffffff : -0x7fffffff;
  __syncthreads();
  const bool hit = qok && bl.x1 + NT > 0 && bl.x1 < W2 && bl.y1 + NT > 0 && bl.y1 < H2;
  if (cq == 0) {
    qorg[qq][0] = hit ? bl.x1 : -0x3fffffff;   // queries outside the map / the image never match a position
    qorg[qq][1] = hit ? bl.y1 : -0x3fffffff;
    if (hit) {
      atomicMin(&bbox[0], max(bl.x1, 0));
      atomicMin(&bbox[1], max(bl.y1, 0));
      atomicMax(&bbox[2], min(bl.x1 + NT, W2));
      atomicMax(&bbox[3], min(bl.y1 + NT, H2));
    }
  }
  // the tap gradients of this query: g(iy, ix) = the four bilinear shares of corr_grad that tap (iy, ix) feeds
  {
    const float* cg = corr_grad + (((size_t)b * N + n) * RD * RD) * H1W1 + pix;
    const float wnw = bl.dy * bl.dx, wne = bl.dy * (1.f - bl.dx), wsw = (1.f - bl.dy) * bl.dx, wse = (1.f - bl.dy) * (1.f - bl.dx);
    for (int t = cq; t < NTAP; t += 4) {
      const int iy = t / NT, ix = t % NT;
      float g = 0.f;  // ak:228-246, same order of the four terms
      if (qok) {
        if (iy > 0 && ix > 0) g += cg[(size_t)((iy - 1) + RD * (ix - 1)) * H1W1] * wnw;
        if (iy > 0 && ix < RD) g += cg[(size_t)((iy - 1) + RD * ix) * H1W1] * wne;
        if (iy < RD && ix > 0) g += cg[(size_t)(iy + RD * (ix - 1)) * H1W1] * wsw;
        if (iy < RD && ix < RD) g += cg[(size_t)(iy + RD * ix) * H1W1] * wse;
      }
      gq[qq][t] = g;
    }
  }
  __syncthreads();
  const int bx0 = bbox[0], by0 = bbox[1];
  const int BW = max(bbox[2] - bx0, 0), BH = max(bbox[3] - by0, 0);
  const int npos = BW * BH;
  if (npos == 0) return;  // no window of the tile touches the map: all gradients are zero (outputs are pre-zeroed)
  const float* f1 = fmap1 + ((size_t)b * H1W1 + pix) * C;
  float* g1 = fmap1_grad + ((size_t)b * H1W1 + pix) * C;
  const float* f2b = fmap2 + (size_t)b * H2 * W2 * C;
  float* g2b = fmap2_grad + (size_t)b * H2 * W2 * C;
  // hit lists: the (query, tap) pairs of every box position, in query order
  if (npos <= AB_MAXPOS) {
    for (int pp = tid; pp < npos; pp += 256) {
      const int py = pp / BW, px = pp - py * BW;
      const int gx = bx0 + px, gy = by0 + py;
      int cnt = 0;
      for (int q = 0; q < ABT * ABT; q++) {
        const unsigned dx = (unsigned)(gx - qorg[q][0]), dy = (unsigned)(gy - qorg[q][1]);
        if (dx < (unsigned)NT && dy < (unsigned)NT) {
          hits[pp][cnt] = (unsigned short)((q << 8) | (dy * NT + dx));
          cnt++;
        }
      }
      hcnt[pp] = (unsigned char)cnt;
    }
  }
  __syncthreads();
  if (npos > AB_MAXPOS) {
    // incoherent tile: per-tap global atomics like altcorr_backward_kernel (each thread takes every 4th channel)
    if (!qok) return;
    for (int iy = 0; iy < NT; iy++)
      for (int ix = 0; ix < NT; ix++) {
        const int h2 = bl.y1 + iy, w2 = bl.x1 + ix;
        if (h2 < 0 || h2 >= H2 || w2 < 0 || w2 >= W2) continue;
        const float g = gq[qq][iy * NT + ix];
        const size_t po = ((size_t)h2 * W2 + w2) * C;
        for (int c = cq; c < C; c += 4) {
          atomicAdd(&g1[c], g * f2b[po + c]);
          atomicAdd(&g2b[po + c], g * f1[c]);
        }
      }
    return;
  }
  for (int c0 = 0; c0 < C; c0 += AB_CH) {
    const int cb = c0 + 4 * cq;  // this thread's 4 channels (C % 16 == 0 on this path)
    const f4 u0 = qok ? *reinterpret_cast<const f4*>(f1 + cb) : f4{0.f, 0.f, 0.f, 0.f};
    if (c0 > 0) __syncthreads();  // the previous pass has read f1s
    *reinterpret_cast<f4*>(&f1s[qq][4 * cq]) = u0;
    __syncthreads();
    // fmap2 gradient: one (position, 4 channels) unit at a time, gathered over its hit list (one channel per lane
    // with 64-byte contiguous atomics was measured slower: 1.28 vs 0.92 ms -- the gather, not the atomics, is the cost)
    for (int u = tid; u < 4 * npos; u += 256) {
      const int pp = u >> 2, cg4 = u & 3;
      const int cnt = hcnt[pp];
      if (cnt == 0) continue;
      f4 a = {0.f, 0.f, 0.f, 0.f};
      for (int h = 0; h < cnt; h++) {
        const unsigned hv = hits[pp][h];
        const float g = gq[hv >> 8][hv & 255];
        a += g * *reinterpret_cast<const f4*>(&f1s[hv >> 8][4 * cg4]);
      }
      const int py = pp / BW, px = pp - py * BW;
      float* dst = g2b + ((size_t)(by0 + py) * W2 + (bx0 + px)) * C + c0 + 4 * cg4;
#pragma unroll
      for (int k = 0; k < 4; k++) atomicAdd(&dst[k], a[k]);
    }
    // fmap1 gradient of this thread's query and channels: register sum over the window
    if (qok) {
      f4 a0 = {0.f, 0.f, 0.f, 0.f};
      for (int iy = 0; iy < NT; iy++) {
        const int h2 = bl.y1 + iy;
        if (h2 < 0 || h2 >= H2) continue;
#pragma unroll 4
        for (int ix = 0; ix < NT; ix++) {
          const int w2 = bl.x1 + ix;
          if (w2 < 0 || w2 >= W2) continue;
          a0 += gq[qq][iy * NT + ix] * *reinterpret_cast<const f4*>(f2b + ((size_t)h2 * W2 + w2) * C + cb);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) atomicAdd(&g1[cb + k], a0[k]);  // other coordinate sets (n) add to the same rows
    }
  }
}

int launch_altcorr_backward(const float* f1, const float* f2, const float* coords,
                            const float* corr_grad, float* f1g, float* f2g, int B, int N, int H1,
                            int W1, int H2, int W2, int C, int r, hipStream_t s) {
  if (B > 65535 || N > 65535) return DROID_E_ARG;
  const int HW = H1 * W1;
  static const bool per_tap = (getenv("DROID_ALTCORR_BWD_PER_TAP") != nullptr);  // diagnostics: the old kernel
  if (!per_tap && (C % AB_CH) == 0 && (r == 3 || r == 4)) {
    const int tiles = ((W1 + ABT - 1) / ABT) * ((H1 + ABT - 1) / ABT);
    if (r == 3)
      hipLaunchKernelGGL((altcorr_backward_tiled<3>), dim3(tiles, N, B), dim3(256), 0, s, f1, f2, coords, corr_grad, f1g,
                         f2g, N, H1, W1, H2, W2, C);
    else
      hipLaunchKernelGGL((altcorr_backward_tiled<4>), dim3(tiles, N, B), dim3(256), 0, s, f1, f2, coords, corr_grad, f1g,
                         f2g, N, H1, W1, H2, W2, C);
    return 0;
  }
  hipLaunchKernelGGL(altcorr_backward_kernel, dim3((HW * 16 + 255) / 256, N, B), dim3(256), 0, s, f1,
                     f2, coords, corr_grad, f1g, f2g, N, HW, H2, W2, C, r);
  return 0;
}

}  // namespace droid
