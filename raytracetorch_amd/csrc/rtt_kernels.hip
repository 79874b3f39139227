// rtt_kernels.hip — gfx950 (MI355X / CDNA4) kernels and C ABI of the RayTraceTorch ray-propagation path.
//
// Execution model
//   * one lane = one ray, or two adjacent rays held in packed registers (v_pk_*_f32); the ray state
//     (pos, dir, intensity: 7 floats) stays in VGPRs for the whole surface program, so HBM sees
//     28 B in + 28 B out per ray per trace regardless of surface count;
//   * ray storage is structure-of-arrays, a wavefront reads/writes 256 or 512 contiguous bytes per
//     component and instruction;
//   * the surface loop is wave-uniform: record `s` is addressed with a scalar index, so its fields come
//     in through scalar loads (SGPRs) and the `switch` on kinds is a scalar branch — no divergence in
//     the sequential trace.  In the nearest-hit step, lanes that won different surfaces are served by
//     a readfirstlane waterfall so the record stays scalar there as well;
//   * no MFMA: the work is elementwise geometry (a few hundred flops per 56 B), HBM-bound.
//
// Numerics follow the reference's float32 torch-CPU formulas (cited per function, paths relative to
// the reference tree).  Two builds of the same source:
//   default            product-sums may contract to FMA and normalisations / quotients use the hardware
//                      reciprocal, reciprocal-sqrt and sqrt (v_rcp_f32 / v_rsq_f32 / v_sqrt_f32, <= 1 ulp
//                      each), except the square root of the discriminant, which stays correctly rounded
//                      — the kernel is otherwise VALU-bound on divide/sqrt expansions (~10 instructions each);
//   -DRTT_STRICT_FP=1  (with -ffp-contract=off) keeps the reference's operation order with
//                      correctly-rounded divide and sqrt; used to separate rounding from logic when
//                      a parity test moves.
// Every discrete decision (hit / miss, bounds, TIR, entering) uses the same comparisons in both.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

#include "rtt_hip.h"

namespace {

constexpr float kInf = __builtin_huge_valf();
constexpr float kEps = 1e-6f;  // geom/primitives.py:6 and :21
constexpr int kBlock = 256;

#ifndef RTT_STRICT_FP
#define RTT_STRICT_FP 0
#endif
constexpr bool kStrict = RTT_STRICT_FP != 0;
// Ray planes are streamed once per launch: non-temporal accesses keep them from displacing each other in L2
#ifndef RTT_NONTEMPORAL
#define RTT_NONTEMPORAL 1
#endif

// ------------------------------------------------------------------------------------------------
// Lane types.  All per-ray arithmetic is written once over a lane type T: `float` (one ray per lane)
// or `f2` (two rays per lane).  The f32 vector ALU of gfx950 retires a wave64 mul / add / fma in four
// cycles whether it is scalar or packed, so carrying two rays per lane in v_pk_{mul,add,fma}_f32
// halves the cost of most of the arithmetic; compares, selects and the transcendental units stay
// per component.  Masks are `bool` for float and a boolean vector for f2: a per-lane bool lives in a
// scalar register pair (lane mask), so mask logic runs on the scalar unit and a select is one v_cndmask.
// ------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
typedef bool b2 __attribute__((ext_vector_type(2)));

template <class T> struct LaneOf;
template <> struct LaneOf<float> { using M = bool; static constexpr int N = 1; };
template <> struct LaneOf<f2> { using M = b2; static constexpr int N = 2; };

#define DI __device__ __forceinline__
DI bool lt(float a, float b) { return a < b; }
DI b2 lt(f2 a, f2 b) { return __builtin_convertvector(a < b, b2); }
DI bool le(float a, float b) { return a <= b; }
DI b2 le(f2 a, f2 b) { return __builtin_convertvector(a <= b, b2); }
DI bool ge(float a, float b) { return a >= b; }
DI b2 ge(f2 a, f2 b) { return __builtin_convertvector(a >= b, b2); }
DI bool gt(float a, float b) { return a > b; }
DI b2 gt(f2 a, f2 b) { return __builtin_convertvector(a > b, b2); }
DI bool isnan_(float a) { return a != a; }
DI b2 isnan_(f2 a) { return __builtin_convertvector(a != a, b2); }
DI bool and_(bool a, bool b) { return a & b; }
DI b2 and_(b2 a, b2 b) { return a & b; }
DI bool or_(bool a, bool b) { return a | b; }
DI b2 or_(b2 a, b2 b) { return a | b; }
DI bool not_(bool a) { return !a; }
DI b2 not_(b2 a) { return ~a; }
DI bool any_(bool a) { return a; }
DI bool any_(b2 a) { return a[0] | a[1]; }
DI bool all_(bool a) { return a; }
DI bool all_(b2 a) { return a[0] & a[1]; }
template <class M> DI M mask_all(bool v);
template <> DI bool mask_all<bool>(bool v) { return v; }
template <> DI b2 mask_all<b2>(bool v) { return b2{v, v}; }
DI float sel(bool m, float a, float b) { return m ? a : b; }
DI f2 sel(b2 m, f2 a, f2 b) { return f2{m[0] ? a.x : b.x, m[1] ? a.y : b.y}; }
DI float abs_(float a) { return fabsf(a); }
DI f2 abs_(f2 a) { return __builtin_elementwise_abs(a); }
DI float max_(float a, float b) { return fmaxf(a, b); }
DI f2 max_(f2 a, f2 b) { return __builtin_elementwise_max(a, b); }
// min of two values that are not NaN (callers guarantee it): compare + select, no canonicalisation
DI float min_(float a, float b) { return a < b ? a : b; }
DI f2 min_(f2 a, f2 b) { return f2{a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y}; }
DI float sqrt_ieee(float a) { return sqrtf(a); }
DI f2 sqrt_ieee(f2 a) { return f2{sqrtf(a.x), sqrtf(a.y)}; }
DI float div_ieee(float a, float b) { return a / b; }
DI f2 div_ieee(f2 a, f2 b) { return a / b; }
// 1/x, 1/sqrt(x), sqrt(x): hardware approximations (<= 1 ulp) unless the strict build is selected.
DI float rcp_(float x) { return kStrict ? 1.0f / x : __builtin_amdgcn_rcpf(x); }
DI f2 rcp_(f2 x) { return f2{rcp_(x.x), rcp_(x.y)}; }
DI float sqrt_(float x) { return kStrict ? sqrtf(x) : __builtin_amdgcn_sqrtf(x); }
DI f2 sqrt_(f2 x) { return f2{sqrt_(x.x), sqrt_(x.y)}; }
DI float rsq_(float x) { return __builtin_amdgcn_rsqf(x); }
DI f2 rsq_(f2 x) { return f2{rsq_(x.x), rsq_(x.y)}; }
// sqrt for the discriminant: hardware sqrt (<= 1 ulp) plus one Newton step on the exact residual
// fma(-s, s, x); the result is within 0.5 ulp (+ 2^-24 relative) of the true root, i.e. correctly
// rounded except for rare near-ties, at a third of the instructions of the full IEEE expansion.
DI float sqrt_disc(float x) {
  if (kStrict) return sqrtf(x);
  const float s = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rsqf(fmaxf(x, 1e-37f));
  return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
DI f2 sqrt_disc(f2 x) { return f2{sqrt_disc(x.x), sqrt_disc(x.y)}; }
DI float quot(float a, float b) { return kStrict ? a / b : a * rcp_(b); }
DI f2 quot(f2 a, f2 b) { return kStrict ? a / b : a * rcp_(b); }
// a / b from the hardware reciprocal plus one correction step on the exact residual fma(-b, q, a): within
// half an ulp (+ 2^-24 relative) of the true quotient.  Used where the quotient is a path length.
DI float quot_refined(float a, float b) {
  if (kStrict) return a / b;
  const float r = __builtin_amdgcn_rcpf(b);
  const float q = a * r;
  return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
DI f2 quot_refined(f2 a, f2 b) {
  if (kStrict) return a / b;
  const f2 r = rcp_(b);
  const f2 q = a * r;
  return __builtin_elementwise_fma(__builtin_elementwise_fma(-b, q, a), r, q);
}
template <class T> DI T splat(float v) { return T(v); }
// float and packed float lanes (as opposed to dual numbers)
template <class T> struct IsPlainLane { static constexpr bool value = false; };
template <> struct IsPlainLane<float> { static constexpr bool value = true; };
template <> struct IsPlainLane<f2> { static constexpr bool value = true; };
DI float comp(float a, int) { return a; }
DI float comp(f2 a, int k) { return k == 0 ? a.x : a.y; }
DI bool set_mask(bool, int, bool v) { return v; }
DI b2 set_mask(b2 m, int k, bool v) { return k == 0 ? b2{v, m[1]} : b2{m[0], v}; }
DI bool mcomp(bool m, int) { return m; }
DI bool mcomp(b2 m, int k) { return k == 0 ? m[0] : m[1]; }


// ------------------------------------------------------------------------------------------------
// Dual numbers for the tangent trace: a value and K directional derivatives.  Instantiating the
// visit templates with T = Dual<K> differentiates exactly the formulas the plain trace evaluates,
// branch for branch (the masks are taken on the values), which is what autograd does to the
// reference's torch ops (torch.where routes gradients by the mask, relu'/abs' follow the sign).
// ------------------------------------------------------------------------------------------------
template <int K>
struct Dual {
  static_assert(K % 2 == 0, "tangents are stored as packed pairs");
  float v;
  f2 tp[K / 2];  // tangent 2j in .x, 2j+1 in .y: the tangent arithmetic runs on v_pk_*_f32
  DI Dual() {}
  DI Dual(float x) : v(x) {
#pragma unroll
    for (int j = 0; j < K / 2; ++j) tp[j] = f2(0.0f);
  }
  DI float tan(int k) const { return (k & 1) ? tp[k >> 1].y : tp[k >> 1].x; }
};
template <int K> struct LaneOf<Dual<K>> { using M = bool; static constexpr int N = 1; };

#define RTT_DUAL_LOOP _Pragma("unroll") for (int j = 0; j < K / 2; ++j)
template <int K> DI Dual<K> operator+(Dual<K> a, Dual<K> b) { Dual<K> r; r.v = a.v + b.v; RTT_DUAL_LOOP r.tp[j] = a.tp[j] + b.tp[j]; return r; }
template <int K> DI Dual<K> operator-(Dual<K> a, Dual<K> b) { Dual<K> r; r.v = a.v - b.v; RTT_DUAL_LOOP r.tp[j] = a.tp[j] - b.tp[j]; return r; }
template <int K> DI Dual<K> operator*(Dual<K> a, Dual<K> b) { Dual<K> r; r.v = a.v * b.v; RTT_DUAL_LOOP r.tp[j] = a.tp[j] * b.v + a.v * b.tp[j]; return r; }
template <int K> DI Dual<K> operator-(Dual<K> a) { Dual<K> r; r.v = -a.v; RTT_DUAL_LOOP r.tp[j] = -a.tp[j]; return r; }
template <int K> DI Dual<K> operator+(Dual<K> a, float b) { a.v += b; return a; }
template <int K> DI Dual<K> operator+(float a, Dual<K> b) { b.v += a; return b; }
template <int K> DI Dual<K> operator-(Dual<K> a, float b) { a.v -= b; return a; }
template <int K> DI Dual<K> operator-(float a, Dual<K> b) { return (-b) + a; }
template <int K> DI Dual<K> operator*(Dual<K> a, float b) { a.v *= b; RTT_DUAL_LOOP a.tp[j] = a.tp[j] * b; return a; }
template <int K> DI Dual<K> operator*(float a, Dual<K> b) { return b * a; }
template <int K> DI bool lt(Dual<K> a, Dual<K> b) { return a.v < b.v; }
template <int K> DI bool le(Dual<K> a, Dual<K> b) { return a.v <= b.v; }
template <int K> DI bool ge(Dual<K> a, Dual<K> b) { return a.v >= b.v; }
template <int K> DI bool gt(Dual<K> a, Dual<K> b) { return a.v > b.v; }
template <int K> DI bool isnan_(Dual<K> a) { return a.v != a.v; }
template <int K> DI Dual<K> sel(bool m, Dual<K> a, Dual<K> b) { return m ? a : b; }
template <int K> DI Dual<K> abs_(Dual<K> a) { return a.v < 0.0f ? -a : (a.v > 0.0f ? a : Dual<K>(fabsf(a.v))); }
template <int K> DI Dual<K> max_(Dual<K> a, Dual<K> b) { return (a.v > b.v || b.v != b.v) ? a : b; }
template <int K> DI Dual<K> min_(Dual<K> a, Dual<K> b) { return (a.v < b.v || b.v != b.v) ? a : b; }
// f(a) with derivative fp: chain rule on every tangent
template <int K> DI Dual<K> chain(Dual<K> a, float f, float fp) { Dual<K> r; r.v = f; RTT_DUAL_LOOP r.tp[j] = a.tp[j] * fp; return r; }
template <int K> DI Dual<K> rcp_(Dual<K> a) { const float r = rcp_(a.v); return chain(a, r, -r * r); }
template <int K> DI Dual<K> sqrt_(Dual<K> a) { const float s = sqrt_(a.v); return chain(a, s, 0.5f * rcp_(s)); }
template <int K> DI Dual<K> sqrt_disc(Dual<K> a) { const float s = sqrt_disc(a.v); return chain(a, s, 0.5f * rcp_(s)); }
template <int K> DI Dual<K> sqrt_ieee(Dual<K> a) { const float s = sqrtf(a.v); return chain(a, s, 0.5f / s); }
template <int K> DI Dual<K> rsq_(Dual<K> a) { const float r = rsq_(a.v); return chain(a, r, -0.5f * r * r * r); }
template <int K> DI Dual<K> quot(Dual<K> a, Dual<K> b) { return a * rcp_(b); }
template <int K> DI Dual<K> quot_refined(Dual<K> a, Dual<K> b) {
  Dual<K> q = a * rcp_(b);
  q.v = quot_refined(a.v, b.v);
  return q;
}
template <int K> DI Dual<K> div_ieee(Dual<K> a, Dual<K> b) { return a * rcp_(b); }
template <int K> DI float comp(Dual<K> a, int) { return a.v; }
#undef RTT_DUAL_LOOP

// Philox4x32-10 (Salmon et al. 2011): counter words (c lo, c hi, word2, word3), 64-bit key.
struct Philox4 {
  uint32_t x[4];
};

DI Philox4 philox4x32_10(uint64_t counter, uint64_t key, uint32_t word2 = 0u, uint32_t word3 = 0u) {
  uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = word2, c3 = word3;
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return {{c0, c1, c2, c3}};
}

template <class T>
struct V3 {
  T x, y, z;
};

template <class T> DI V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> DI V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> DI V3<T> scale(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class T> DI T dot3(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T, class M> DI V3<T> sel3(M m, V3<T> a, V3<T> b) { return {sel(m, a.x, b.x), sel(m, a.y, b.y), sel(m, a.z, b.z)}; }

// ------------------------------------------------------------------------------------------------
// Visit configurations.  A surface visit is written once, as templates over a configuration C that
// answers "which geometry / bound / shape bound / physics is this record?".  The generic
// configuration reads the kinds from the record (scalar registers, scalar branches); specialised
// configurations fix them at compile time so that the commonest visits (lens face + Snell, lens edge,
// bounded plane, mirror cap, ...) become straight-line vector code with a handful of scalar
// instructions.  The host picks the configuration per record (classify(), below); a record that
// matches none runs the generic one, so specialisation never changes what is computed.
// ------------------------------------------------------------------------------------------------
constexpr int DYN = -1;

template <int SK, int BK, int HK, int PK, bool ROT_ID, int INV = 0>
struct Cfg {
  // SurfaceBounded.invert: read from the record (INV == DYN) or known to be clear (INV == 0)
  static __device__ __forceinline__ bool inv(const RttSurface& S) { return INV == DYN ? (S.flags & RTT_F_INVERT) != 0 : false; }
  static __device__ __forceinline__ int sk(const RttSurface& S) { return SK == DYN ? S.surf_kind : SK; }
  static __device__ __forceinline__ int bk(const RttSurface& S) { return BK == DYN ? S.bound_kind : BK; }
  static __device__ __forceinline__ int hk(const RttSurface& S) { return HK == DYN ? S.shape_kind : HK; }
  static __device__ __forceinline__ int pk(const RttSurface& S) { return PK == DYN ? S.phys_kind : PK; }
  static __device__ __forceinline__ bool has_shape(const RttSurface& S) {
    return HK == DYN ? (S.flags & RTT_F_HAS_SHAPE) != 0 : HK != RTT_SHP_NONE;
  }
  static __device__ __forceinline__ bool rs_id(const RttSurface& S) {
    return ROT_ID ? true : (S.flags & RTT_F_SHAPE_ROT_ID) != 0;
  }
  static __device__ __forceinline__ bool rj_id(const RttSurface& S) {
    return ROT_ID ? true : (S.flags & RTT_F_SURF_ROT_ID) != 0;
  }
};

using CGeneric = Cfg<DYN, DYN, DYN, DYN, false, DYN>;
using CGenericIdent = Cfg<DYN, DYN, DYN, DYN, true, DYN>;
// spherical / conic lens face of a Spheric stack, unrotated element, Snell refraction
using CFaceSnell = Cfg<RTT_SURF_QUADRIC, RTT_BND_HALF, RTT_SHP_SPHERIC_FACE, RTT_PHYS_SNELL, true>;
// cylindrical lens edge of a Spheric stack, inked (Block) or clear (Snell)
using CEdgeBlock = Cfg<RTT_SURF_CYLINDER, RTT_BND_NONE, RTT_SHP_SPHERIC_EDGE, RTT_PHYS_BLOCK, true>;
using CEdgeSnell = Cfg<RTT_SURF_CYLINDER, RTT_BND_NONE, RTT_SHP_SPHERIC_EDGE, RTT_PHYS_SNELL, true>;
// bare disk-bounded plane: sensors (Transmit) and circular stops (ApertureFilter)
using CDiskPlane = Cfg<RTT_SURF_PLANE, RTT_BND_DISK, RTT_SHP_NONE, DYN, true, DYN>;
// bare spherical mirror cap
using CMirrorCap = Cfg<RTT_SURF_QUADRIC, RTT_BND_HALF_DISK, RTT_SHP_NONE, RTT_PHYS_REFLECT, true>;
// cylindrical lens: curved faces (unrotated) and the four side planes (rotated by +-90 degrees)
using CCylFaceSnell = Cfg<RTT_SURF_QUADRIC_ZY, RTT_BND_HALF, RTT_SHP_CYL_FACE, RTT_PHYS_SNELL, true>;
using CCylSide = Cfg<RTT_SURF_PLANE, RTT_BND_NONE, RTT_SHP_CYL_EDGE, DYN, false>;

enum { OP_GENERIC = 0, OP_FACE_SNELL, OP_EDGE_BLOCK, OP_EDGE_SNELL, OP_DISK_PLANE, OP_MIRROR_CAP, OP_CYL_FACE_SNELL,
       OP_CYL_SIDE, OP_COUNT };

// one opcode byte per surface, passed by value as a kernel argument (scalar loads)
struct Ops {
  uint32_t w[RTT_MAX_SURFACES / 4];
};

__device__ __forceinline__ int op_of(const Ops& ops, int s) { return (ops.w[s >> 2] >> ((s & 3) * 8)) & 0xff; }

#define RTT_DISPATCH(op, CALL)          \
  switch (op) {                         \
    case OP_FACE_SNELL: CALL(CFaceSnell); break;       \
    case OP_EDGE_BLOCK: CALL(CEdgeBlock); break;       \
    case OP_EDGE_SNELL: CALL(CEdgeSnell); break;       \
    case OP_DISK_PLANE: CALL(CDiskPlane); break;       \
    case OP_MIRROR_CAP: CALL(CMirrorCap); break;       \
    case OP_CYL_FACE_SNELL: CALL(CCylFaceSnell); break; \
    case OP_CYL_SIDE: CALL(CCylSide); break;           \
    default: CALL(CGeneric); break;                    \
  }

// NG = 1: the program holds no record that needs the generic interpreter (the host has looked); its visit is left out.
#define RTT_DISPATCH_NG(op, CALL)                         \
  if constexpr (NG == 1) {                                \
    switch (op) {                                         \
      case OP_FACE_SNELL: CALL(CFaceSnell); break;        \
      case OP_EDGE_BLOCK: CALL(CEdgeBlock); break;        \
      case OP_EDGE_SNELL: CALL(CEdgeSnell); break;        \
      case OP_DISK_PLANE: CALL(CDiskPlane); break;        \
      case OP_MIRROR_CAP: CALL(CMirrorCap); break;        \
      case OP_CYL_FACE_SNELL: CALL(CCylFaceSnell); break; \
      default: CALL(CCylSide); break;                     \
    }                                                     \
  } else {                                                \
    RTT_DISPATCH(op, CALL)                                \
  }

// Record views.  The arithmetic reads record fields through a view so that the same templates serve the
// plain trace (fields are wave-uniform floats in scalar registers) and the tangent trace (fields are
// dual numbers carrying d(field)/d(parameter)).
struct RecF {
  const RttSurface& r;
  DI float Rs(int i) const { return r.Rs[i]; }
  DI float Rp(int i) const { return (r.flags & RTT_F_PHYS_ROT_SURF) ? r.Rj[i] : r.phys_R[i]; }
  DI float ts(int i) const { return r.ts[i]; }
  DI float Rj(int i) const { return r.Rj[i]; }
  DI float tj(int i) const { return r.tj[i]; }
  DI float geom(int i) const { return r.geom[i]; }
  DI float phys(int i) const { return r.phys[i]; }
  DI float bnd(int i) const { return r.bnd[i]; }
  DI float shp(int i) const { return r.shp[i]; }
};

// LDS view: the same fields read from a copy of the program staged in LDS (ds_read with a wave-uniform
// address).  Kept as a measured alternative to the scalar-load view (rtt_tune key 2).
struct RecL {
  const RttSurface& r;  // LDS-resident record
  DI float Rs(int i) const { return r.Rs[i]; }
  DI float Rp(int i) const { return (r.flags & RTT_F_PHYS_ROT_SURF) ? r.Rj[i] : r.phys_R[i]; }
  DI float ts(int i) const { return r.ts[i]; }
  DI float Rj(int i) const { return r.Rj[i]; }
  DI float tj(int i) const { return r.tj[i]; }
  DI float geom(int i) const { return r.geom[i]; }
  DI float phys(int i) const { return r.phys[i]; }
  DI float bnd(int i) const { return r.bnd[i]; }
  DI float shp(int i) const { return r.shp[i]; }
};

// Tangent view of a record: the 32 differentiable slots (Rs 0-8, ts 9-11, Rj 12-20, tj 21-23, geom 24-27,
// phys 28-31) carry d(slot)/d(theta_k) from a per-surface table [32][K]; bounds are not differentiated
// (masks are piecewise constant).
constexpr int kTanSlots = 32;
template <int K>
struct RecD {
  const RttSurface& r;
  const float* tan;  // [kTanSlots][K] of this surface
  DI Dual<K> mk(float v, int slot) const {
    Dual<K> d;
    d.v = v;
#pragma unroll
    for (int j = 0; j < K / 2; ++j) d.tp[j] = f2{tan[slot * K + 2 * j], tan[slot * K + 2 * j + 1]};
    return d;
  }
  DI Dual<K> Rs(int i) const { return mk(r.Rs[i], i); }
  // a Linear functor's own rotation is not a differentiable slot; the surface rotation is
  DI Dual<K> Rp(int i) const { return (r.flags & RTT_F_PHYS_ROT_SURF) ? mk(r.Rj[i], 12 + i) : Dual<K>(r.phys_R[i]); }
  DI Dual<K> ts(int i) const { return mk(r.ts[i], 9 + i); }
  DI Dual<K> Rj(int i) const { return mk(r.Rj[i], 12 + i); }
  DI Dual<K> tj(int i) const { return mk(r.tj[i], 21 + i); }
  DI Dual<K> geom(int i) const { return mk(r.geom[i], 24 + i); }
  DI Dual<K> phys(int i) const { return mk(r.phys[i], 28 + i); }
  DI float bnd(int i) const { return r.bnd[i]; }
  DI float shp(int i) const { return r.shp[i]; }
};

// v @ R, R row-major (geom/transform.py:92-93); W = 0 takes the element rotation Rs, W = 1 the surface
// rotation Rj, W = 2 the rotation of a Linear functor.  `ident` skips the products: with R exactly the identity x*1 + y*0 + z*0 == x bit for bit.
template <int W, class RV>
DI auto rot(const RV& S, int i) -> decltype(S.Rs(0)) {
  return W == 0 ? S.Rs(i) : (W == 1 ? S.Rj(i) : S.Rp(i));
}

template <int W, class T, class RV>
DI V3<T> rowmat(V3<T> v, const RV& S, bool ident) {
  if (ident) return v;
  return {v.x * rot<W>(S, 0) + v.y * rot<W>(S, 3) + v.z * rot<W>(S, 6),
          v.x * rot<W>(S, 1) + v.y * rot<W>(S, 4) + v.z * rot<W>(S, 7),
          v.x * rot<W>(S, 2) + v.y * rot<W>(S, 5) + v.z * rot<W>(S, 8)};
}

// v @ R^T (geom/primitives.py:94, geom/shape.py:85)
template <int W, class T, class RV>
DI V3<T> rowmat_t(V3<T> v, const RV& S, bool ident) {
  if (ident) return v;
  return {v.x * rot<W>(S, 0) + v.y * rot<W>(S, 1) + v.z * rot<W>(S, 2),
          v.x * rot<W>(S, 3) + v.y * rot<W>(S, 4) + v.z * rot<W>(S, 5),
          v.x * rot<W>(S, 6) + v.y * rot<W>(S, 7) + v.z * rot<W>(S, 8)};
}

// torch.min over a stacked dim propagates NaN (geom/primitives.py:34, geom/bounded.py:34)
template <class T>
DI T nanmin(T a, T b) {
  // a when a < b or a is NaN, else b (which is the smaller value or itself NaN)
  return sel(or_(lt(a, b), isnan_(a)), a, b);
}

template <class T>
struct Frames {
  V3<T> pe, de;  // element frame (de un-normalised, geom/shape.py:47)
  V3<T> pl, dl;  // surface frame
};

// Global -> element frame (+ re-normalisation, rays/ray.py:22-25 via with_coords :84-97)
// -> surface frame (geom/shape.py:37-38, geom/primitives.py:50, geom/transform.py:90-93).
template <class C, class RV, class T>
DI Frames<T> to_frames(const RV& S, V3<T> p, V3<T> d) {
  Frames<T> f;
  V3<T> den;
  if (C::has_shape(S.r)) {
    const bool id = C::rs_id(S.r);
    V3<T> ts = {T(S.ts(0)), T(S.ts(1)), T(S.ts(2))};
    f.pe = rowmat<0>(p - ts, S, id);
    f.de = rowmat<0>(d, S, id);
    const T n2 = f.de.x * f.de.x + f.de.y * f.de.y + f.de.z * f.de.z;
    if (kStrict) {
      const T n = max_(sqrt_ieee(n2), splat<T>(1e-12f));
      den = {div_ieee(f.de.x, n), div_ieee(f.de.y, n), div_ieee(f.de.z, n)};
    } else {
      // de / max(|de|, 1e-12) == de * rsqrt(max(|de|^2, 1e-24))
      den = scale(rsq_(max_(n2, splat<T>(1e-24f))), f.de);
    }
  } else {
    f.pe = p;
    f.de = d;
    den = d;
  }
  const bool idj = C::rj_id(S.r);
  V3<T> tj = {T(S.tj(0)), T(S.tj(1)), T(S.tj(2))};
  f.pl = rowmat<1>(f.pe - tj, S, idj);
  f.dl = rowmat<1>(den, S, idj);
  return f;
}

// Quadric._solve_quadratic (geom/primitives.py:290-320) and the Cone variant (:436-466).
// Roots come back with a validity mask instead of being overwritten with inf: `ok` false means the
// reference's value is inf (complex roots), which the root filter turns into a miss either way.
template <bool ABS_DISC, class T, class M>
DI void quadratic_with_linear_fallback(T A, T B, T Cc, T& t1, T& t2, M& ok) {
  const T eps = splat<T>(kEps);
  const T disc = B * B - 4.0f * A * Cc;
  const auto hit = ge(disc, splat<T>(0.0f));
  const auto lin = lt(abs_(A), eps);
  // kept (nearly) correctly rounded in both builds: the root -B - sqrt(disc) cancels for near-paraxial
  // rays and amplifies every ulp of the square root
  const T sq = ABS_DISC ? sqrt_disc(abs_(disc)) : sqrt_disc(sel(hit, disc, splat<T>(0.0f)));
  if (kStrict) {
    const T den = 2.0f * sel(lin, splat<T>(1.0f), A);
    t1 = div_ieee(-B - sq, den);
    t2 = div_ieee(-B + sq, den);
  } else {
    // |A| < eps lanes take the linear root below, so the guard A_safe = 1 of the reference is not needed
    const T r = rcp_(2.0f * A);
    t1 = (-B - sq) * r;
    t2 = (-B + sq) * r;
  }
  ok = hit;
  if (any_(lin)) {
    const T B_safe = sel(lt(abs_(B), eps), eps, B);
    const T tl = quot(-Cc, B_safe);
    t1 = sel(lin, tl, t1);
    t2 = sel(lin, tl, t2);
    ok = or_(hit, lin);
  }
}

// _solve_t of each primitive in the surface frame.  Returns the number of roots (1 or 2); `ok`
// false marks roots the reference sets to inf.
template <class C, class RV, class T, class M>
DI int solve_roots(const RV& S, V3<T> o, V3<T> d, T& t1, T& t2, M& ok) {
  const T zero = splat<T>(0.0f);
  switch (C::sk(S.r)) {
    case RTT_SURF_PLANE: {  // geom/primitives.py:124-136
      const T safe = sel(lt(abs_(d.z), splat<T>(kEps)), splat<T>(1e-8f), d.z);
      t1 = quot_refined(-o.z, safe);
      t2 = t1;
      ok = mask_all<M>(true);
      return 1;
    }
    case RTT_SURF_SPHERE: {  // geom/primitives.py:155-181
      const T b = 2.0f * dot3(o, d);
      const T c = dot3(o, o) - S.geom(1);
      const T disc = b * b - 4.0f * c;
      ok = ge(disc, zero);
      const T sq = sqrt_disc(sel(ok, disc, zero));
      t1 = (-b - sq) * 0.5f;
      t2 = (-b + sq) * 0.5f;
      return 2;
    }
    case RTT_SURF_CYLINDER: {  // geom/primitives.py:201-231
      const T A = d.x * d.x + d.y * d.y;
      const T B = 2.0f * (o.x * d.x + o.y * d.y);
      const T Cc = (o.x * o.x + o.y * o.y) - S.geom(1);
      const T disc = B * B - 4.0f * A * Cc;
      ok = ge(disc, zero);
      const T sq = sqrt_disc(abs_(disc));
      const T den = 2.0f * A;
      // A == 0 gives 1/den = inf; (-B -+ sq) * inf is inf or NaN exactly as the quotient by zero is
      t1 = quot(-B - sq, den);
      t2 = quot(-B + sq, den);
      return 2;
    }
    case RTT_SURF_QUADRIC: {  // geom/primitives.py:266-288
      const auto c = S.geom(0), c1k = S.geom(2);
      const auto two_c = 2.0f * c, two_c1k = 2.0f * c1k;
      const T A = c * (d.x * d.x + d.y * d.y) + c1k * (d.z * d.z);
      const T B = two_c * (o.x * d.x + o.y * d.y) + two_c1k * o.z * d.z - 2.0f * d.z;
      const T Cc = c * (o.x * o.x + o.y * o.y) + c1k * (o.z * o.z) - 2.0f * o.z;
      quadratic_with_linear_fallback<true>(A, B, Cc, t1, t2, ok);
      return 2;
    }
    case RTT_SURF_QUADRIC_ZY: {  // geom/primitives.py:356-377
      const auto c = S.geom(0), c1k = S.geom(2);
      const auto two_c = 2.0f * c, two_c1k = 2.0f * c1k;
      const T A = c * (d.y * d.y) + c1k * (d.z * d.z);
      const T B = two_c * (o.y * d.y) + two_c1k * o.z * d.z - 2.0f * d.z;
      const T Cc = c * (o.y * o.y) + c1k * (o.z * o.z) - 2.0f * o.z;
      quadratic_with_linear_fallback<true>(A, B, Cc, t1, t2, ok);
      return 2;
    }
    default: {  // RTT_SURF_CONE, geom/primitives.py:416-466
      const auto k2 = S.geom(1);
      const T A = d.z * d.z - k2 * (d.x * d.x + d.y * d.y);
      const T B = 2.0f * (o.z * d.z - k2 * (o.x * d.x + o.y * d.y));
      const T Cc = o.z * o.z - k2 * (o.x * o.x + o.y * o.y);
      quadratic_with_linear_fallback<false>(A, B, Cc, t1, t2, ok);
      return 2;
    }
  }
}

// SurfaceBounded.inBounds variants in the surface frame, without `invert` (geom/bounded.py).
template <class C, class RV, class T>
DI typename LaneOf<T>::M surface_in_bounds(const RV& S, V3<T> h) {
  switch (C::bk(S.r)) {
    case RTT_BND_DISK:  // bounded.py:60-64
      return le(h.x * h.x + h.y * h.y, splat<T>(S.bnd(0)));
    case RTT_BND_RECT:  // bounded.py:77-82
      return and_(le(abs_(h.x), splat<T>(S.bnd(0))), le(abs_(h.y), splat<T>(S.bnd(1))));
    case RTT_BND_ELLIPSE: {  // bounded.py:98-106
      const T dM = h.x * S.bnd(0) - h.y * S.bnd(1);
      const T dm = h.x * S.bnd(1) + h.y * S.bnd(0);
      const T a = quot(dM, splat<T>(S.bnd(2)));
      const T b = quot(dm, splat<T>(S.bnd(3)));
      return le(a * a + b * b, splat<T>(1.0f));
    }
    case RTT_BND_HALF:  // bounded.py:123-127, :171-174
      return lt(abs_(h.z * S.geom(0)), splat<T>(1.000001f));
    case RTT_BND_HALF_DISK:  // bounded.py:151-159
      return and_(lt(abs_(h.z * S.geom(0)), splat<T>(1.000001f)), le(h.x * h.x + h.y * h.y, splat<T>(S.bnd(0))));
    case RTT_BND_SINGLE_CONE:  // bounded.py:208-217
      return ge(h.z * S.geom(0), splat<T>(-kEps));
    default:  // RTT_BND_ALWAYS
      return mask_all<typename LaneOf<T>::M>(true);
  }
}

// Surface._check_t (geom/primitives.py:28-36) / SurfaceBounded._check_t (geom/bounded.py:20-36) for one
// root: true when the reference keeps the root, false when it replaces it by inf.  `ok` false stands
// for a root the reference has already set to inf: inf fails neither the epsilon test nor (as a NaN hit
// point) an un-inverted bound, so it stays inf there as here.
template <class C, class RV, class T, class M>
DI M root_kept(const RV& S, const Frames<T>& f, T r, M ok) {
  M good = and_(ok, not_(le(r, splat<T>(kEps))));
  if (C::bk(S.r) != RTT_BND_NONE) {
    V3<T> h = f.pl + scale(r, f.dl);
    auto keep = surface_in_bounds<C>(S, h);
    if (C::inv(S.r)) keep = not_(keep);
    good = and_(good, keep);
  }
  return good;
}

// min over the filtered roots (geom/primitives.py:34, geom/bounded.py:34): the column value t (inf on
// a miss) and `finite` = t < inf.
template <class C, class RV, class T, class M>
DI T solve_t_any(const RV& S, const Frames<T>& f, M& finite) {
  T t1, t2;
  M ok;
  const T inf = splat<T>(kInf);
  const int n = solve_roots<C>(S, f.pl, f.dl, t1, t2, ok);
  const M g1 = root_kept<C>(S, f, t1, ok);
  if (n == 1) {
    const T t = sel(g1, t1, inf);
    finite = lt(t, inf);
    return t;
  }
  const M g2 = root_kept<C>(S, f, t2, ok);
  const int bk = C::bk(S.r);
  if (bk != RTT_BND_NONE && bk != RTT_BND_ALWAYS && !C::inv(S.r)) {
    // a kept root passed a bound test on its hit point, so it is finite and not NaN: choose on the masks
    const M first = and_(g1, or_(not_(g2), lt(t1, t2)));
    finite = or_(g1, g2);
    return sel(finite, sel(first, t1, t2), inf);
  }
  // unbounded surfaces and inverted bounds can keep a NaN root; torch.min propagates it
  const T t = nanmin(sel(g1, t1, inf), sel(g2, t2, inf));
  finite = lt(t, inf);
  return t;
}

template <class C, class RV, class T, class M>
DI T solve_t(const RV& S, const Frames<T>& f, M& finite) {
  return solve_t_any<C>(S, f, finite);
}

DI V3<float> values(V3<float> v) { return v; }
template <int K> DI V3<float> values(V3<Dual<K>> v) { return {v.x.v, v.y.v, v.z.v}; }

// Dual-number rays: the root is found on the values (the float path above: same roots, same filters, same choice),
// and its tangents come from differentiating "t solves A t^2 + B t + C = 0" implicitly,
//     dt = -(t^2 dA + t dB + dC) / (2 A t + B),
// with A, B, C the reference's coefficient formulas of the surface kind: the derivative autograd obtains through the
// closed-form root, at a fraction of the arithmetic (no dual-number square root, reciprocal and second root).  Where
// the reference takes the linear root -C / B_safe (|A| < 1e-6) the A term is dropped, as it is there.  Planes (one
// quotient) and cones keep the explicit path.
template <class C, int K>
DI Dual<K> solve_t(const RecD<K>& S, const Frames<Dual<K>>& f, bool& finite) {
  const int sk = C::sk(S.r);
  if (sk == RTT_SURF_PLANE || sk == RTT_SURF_CONE) return solve_t_any<C>(S, f, finite);
  const RecF Sv{S.r};
  const Frames<float> fv = {values(f.pe), values(f.de), values(f.pl), values(f.dl)};
  const float t = solve_t_any<C>(Sv, fv, finite);
  Dual<K> out(t);
  if (!finite) return out;
  const V3<float> o = fv.pl, d = fv.dl;
  const V3<float> h = {o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};
  // t^2 dA + t dB + dC = go . d(o) + gd . d(d) + g0 d(geom[i0]) + g1 d(geom[i1]);  den = 2 A t + B
  V3<float> go, gd;
  float g0 = 0.0f, g1 = 0.0f, den;
  int i0 = 0, i1 = 0;
  if (sk == RTT_SURF_SPHERE) {  // t^2 + b t + c with b = 2 o.d, c = o.o - R^2 (no d.d term in the reference)
    go = {2.0f * h.x, 2.0f * h.y, 2.0f * h.z};
    gd = {2.0f * t * o.x, 2.0f * t * o.y, 2.0f * t * o.z};
    g0 = -1.0f;
    i0 = 1;
    den = 2.0f * t + 2.0f * (o.x * d.x + o.y * d.y + o.z * d.z);
  } else if (sk == RTT_SURF_CYLINDER) {
    go = {2.0f * h.x, 2.0f * h.y, 0.0f};
    gd = {t * go.x, t * go.y, 0.0f};
    g0 = -1.0f;
    i0 = 1;
    den = 2.0f * (d.x * d.x + d.y * d.y) * t + 2.0f * (o.x * d.x + o.y * d.y);
  } else {  // RTT_SURF_QUADRIC, RTT_SURF_QUADRIC_ZY
    const float c = S.r.geom[0], c1k = S.r.geom[2];
    const float kx = sk == RTT_SURF_QUADRIC ? 1.0f : 0.0f;
    const float A = c * (kx * d.x * d.x + d.y * d.y) + c1k * (d.z * d.z);
    const float B = 2.0f * c * (kx * o.x * d.x + o.y * d.y) + 2.0f * c1k * o.z * d.z - 2.0f * d.z;
    go = {2.0f * c * kx * h.x, 2.0f * c * h.y, 2.0f * c1k * h.z - 2.0f};
    i0 = 0;
    i1 = 2;
    if (fabsf(A) < kEps) {  // the reference's linear root: t = -C / B_safe
      gd = {t * 2.0f * c * kx * o.x, t * 2.0f * c * o.y, t * (2.0f * c1k * o.z - 2.0f)};
      g0 = 2.0f * t * (kx * o.x * d.x + o.y * d.y) + (kx * o.x * o.x + o.y * o.y);
      g1 = 2.0f * t * o.z * d.z + o.z * o.z;
      den = fabsf(B) < kEps ? kEps : B;
    } else {
      gd = {t * go.x, t * go.y, t * go.z};
      g0 = kx * h.x * h.x + h.y * h.y;
      g1 = h.z * h.z;
      den = 2.0f * A * t + B;
    }
  }
  const float s = -1.0f / den;
  const Dual<K> p0 = S.geom(i0), p1 = S.geom(i1);
#pragma unroll
  for (int j = 0; j < K / 2; ++j) {
    f2 num = go.x * f.pl.x.tp[j] + go.y * f.pl.y.tp[j] + go.z * f.pl.z.tp[j];
    num += gd.x * f.dl.x.tp[j] + gd.y * f.dl.y.tp[j] + gd.z * f.dl.z.tp[j];
    num += g0 * p0.tp[j] + g1 * p1.tp[j];
    out.tp[j] = s * num;
  }
  return out;
}

template <class C, class RV, class T>
DI T solve_t(const RV& S, const Frames<T>& f) {
  typename LaneOf<T>::M finite;
  return solve_t<C>(S, f, finite);
}

// HalfCyl.sagittalZ with a per-ray height (geom/bounded.py:176-186)
template <class T>
DI T sag(float c, float tz, T y) {
  const T r_sq = y * y;
  T term = 1.0f - (c * c) * r_sq;
  term = sel(gt(term, splat<T>(0.0f)), term, sel(isnan_(term), term, splat<T>(0.0f)));  // relu keeps NaN
  const T den = 1.0f + sqrt_(term);
  return quot(c * r_sq, den) + tz;
}

// Shape.inBounds(hit, j) in the element frame (geom/spherics.py:27-46, geom/cylindrics.py:23-55)
template <class C, class RV, class T>
DI typename LaneOf<T>::M shape_in_bounds(const RV& S, V3<T> h) {
  switch (C::hk(S.r)) {
    case RTT_SHP_SPHERIC_FACE:
      return le(h.x * h.x + h.y * h.y, splat<T>(S.shp(0)));
    case RTT_SHP_SPHERIC_EDGE:
      return and_(ge(h.z, splat<T>(S.shp(0))), le(h.z, splat<T>(S.shp(1))));
    case RTT_SHP_CYL_FACE:
    case RTT_SHP_CYL_EDGE: {
      const auto ap = and_(and_(le(h.x, splat<T>(S.shp(0))), ge(h.x, splat<T>(S.shp(1)))),
                           and_(le(h.y, splat<T>(S.shp(2))), ge(h.y, splat<T>(S.shp(3)))));
      if (C::hk(S.r) == RTT_SHP_CYL_FACE) return ap;
      const T zf = sag(S.shp(4), S.shp(5), h.y);
      const T zb = sag(S.shp(6), S.shp(7), h.y);
      return and_(and_(ge(h.z, zf + 1e-4f), le(h.z, zb - 1e-4f)), ap);
    }
    case RTT_SHP_POLYHEDRON: {  // geom/shape.py:122-132
      // the planes are the element's own records, which sit next to this one in the program (checked on the
      // host); the loop and the record reads are wave-uniform
      const RttSurface* first = &S.r - S.r.surface_id;
      const int count = (int)S.r.shp[0];
      auto inside = mask_all<typename LaneOf<T>::M>(true);
      for (int q = 0; q < count; ++q) {
        if (q == S.r.surface_id) continue;
        const RttSurface& Q = first[q];
        const T z = Q.Rj[6] * (h.x - Q.tj[0]) + Q.Rj[7] * (h.y - Q.tj[1]) + Q.Rj[8] * (h.z - Q.tj[2]);
        inside = and_(inside, lt(z, splat<T>(1e-4f)));
      }
      return inside;
    }
    default:
      return mask_all<typename LaneOf<T>::M>(true);
  }
}

// One column of Element.intersectTest (geom/shape.py:42-56 or geom/primitives.py:38-57), returned as
// the hit mask `t < inf` together with the frames and the un-masked t (what Shape.forward recomputes,
// geom/shape.py:78).  masked_t() gives the column value itself.
template <class C, class RV, class T>
DI typename LaneOf<T>::M test_surface(const RV& S, V3<T> p, V3<T> d, Frames<T>& f, T& t_raw) {
  f = to_frames<C>(S, p, d);
  typename LaneOf<T>::M valid;
  if (C::hk(S.r) == RTT_SHP_SPHERIC_EDGE) {
    // Slab rejection for lens edges: a kept root has t > 0, so the element-frame hit height
    // pe.z + t de.z lies beyond pe.z in the direction of de.z; a ray already past the edge's z range and
    // moving away cannot land inside it.  When that holds for every ray of the wavefront the quadratic
    // is skipped; the result (a miss) is exactly what the full evaluation returns.
    const T zero = splat<T>(0.0f);
    const auto away = or_(and_(ge(f.de.z, zero), gt(f.pe.z, splat<T>(S.shp(1)))),
                          and_(le(f.de.z, zero), lt(f.pe.z, splat<T>(S.shp(0)))));
    if (__all(all_(away))) {
      t_raw = splat<T>(kInf);
      return mask_all<typename LaneOf<T>::M>(false);
    }
  }
  t_raw = solve_t<C>(S, f, valid);
  if (C::has_shape(S.r)) {
    V3<T> he = f.pe + scale(t_raw, f.de);
    valid = and_(valid, shape_in_bounds<C>(S, he));
  }
  return valid;
}

// the [N, K] matrix entry: a Shape replaces every invalid t (NaN included) by inf, a bare Surface
// returns its t as is (geom/shape.py:50-55 vs geom/primitives.py:55-57)
template <class C, class RV, class T, class M>
DI T masked_t(const RV& S, T t_raw, M valid) {
  return C::has_shape(S.r) ? sel(valid, t_raw, splat<T>(kInf)) : t_raw;
}

// _getNormal in the surface frame
template <class C, class RV, class T>
DI V3<T> surface_normal(const RV& S, V3<T> h) {
  const T zero = splat<T>(0.0f), one = splat<T>(1.0f);
  switch (C::sk(S.r)) {
    case RTT_SURF_PLANE:  // primitives.py:138-143
      return {zero, zero, one};
    case RTT_SURF_SPHERE: {  // primitives.py:183-187
      const T R = T(S.geom(0));
      return {quot(h.x, R), quot(h.y, R), quot(h.z, R)};
    }
    case RTT_SURF_CYLINDER: {  // primitives.py:233-241
      const T R = T(S.geom(0));
      return {quot(h.x, R), quot(h.y, R), zero};
    }
    case RTT_SURF_QUADRIC:
    case RTT_SURF_QUADRIC_ZY: {  // primitives.py:330-343, :379-395
      const auto two_c = 2.0f * S.geom(0), two_c1k = 2.0f * S.geom(2);
      const T nx = C::sk(S.r) == RTT_SURF_QUADRIC ? two_c * h.x : zero;
      const T ny = two_c * h.y;
      const T nz = two_c1k * h.z - 2.0f;
      const T len = sqrt_(nx * nx + ny * ny + nz * nz) + 1e-8f;
      if (kStrict) return {-div_ieee(nx, len), -div_ieee(ny, len), -div_ieee(nz, len)};
      const T r = -rcp_(len);
      return {nx * r, ny * r, nz * r};
    }
    default: {  // cone, primitives.py:468-497
      const auto k2 = S.geom(1);
      const T nx = -k2 * h.x, ny = -k2 * h.y, nz = h.z;
      const T len = sqrt_(nx * nx + ny * ny + nz * nz);
      const auto ok = gt(len, splat<T>(1e-8f));
      const T l = len + 1e-8f;
      return {sel(ok, quot(nx, l), zero), sel(ok, quot(ny, l), zero), sel(ok, quot(nz, l), one)};
    }
  }
}

template <class T>
struct Forward {
  V3<T> hit_g, hit_l, n_g;
};

// Shape.forward / Surface.forward given frames and t (geom/shape.py:78-87, geom/primitives.py:85-96)
template <class C, class RV, class T>
DI Forward<T> surface_forward(const RV& S, V3<T> p, V3<T> d, const Frames<T>& f, T t) {
  Forward<T> o;
  o.hit_g = p + scale(t, d);
  o.hit_l = f.pl + scale(t, f.dl);
  V3<T> n = surface_normal<C>(S, o.hit_l);
  n = rowmat_t<1>(n, S, C::rj_id(S.r));
  if (C::has_shape(S.r)) n = rowmat_t<0>(n, S, C::rs_id(S.r));
  o.n_g = n;
  return o;
}

// The uniform of RefractFresnel's Monte-Carlo choice (phys/std.py:200) for the ray with index r of the call in
// bounce b of the launch: word 2*(r&1) of the Philox block with counter (r >> 1, 0, b) under `key`.  The reference
// draws fresh numbers at every visit; a ray that meets the same surface again in a later bounce of one launch gets
// another block through b.
DI float unit_uniform(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }  // [0, 1), 24 bits
DI float fresnel_uniform(uint64_t key, uint64_t r, uint32_t b, float*) {
  return unit_uniform(philox4x32_10(r >> 1, key, 0u, b).x[2 * (r & 1)]);
}
DI f2 fresnel_uniform(uint64_t key, uint64_t r, uint32_t b, f2*) {
  if ((r & 1) == 0) {  // the usual case: one block serves the lane's two rays
    const Philox4 blk = philox4x32_10(r >> 1, key, 0u, b);
    return f2{unit_uniform(blk.x[0]), unit_uniform(blk.x[2])};
  }
  return f2{fresnel_uniform(key, r, b, (float*)nullptr), fresnel_uniform(key, r + 1, b, (float*)nullptr)};
}
template <int K> DI Dual<K> fresnel_uniform(uint64_t key, uint64_t r, uint32_t b, Dual<K>*) {
  return Dual<K>(fresnel_uniform(key, r, b, (float*)nullptr));
}

// SurfaceFunction.forward(hit_local, dir_global, normal_global) -> (new_dir, imult); `ridx` is the index in
// the call of the lane's (first) ray and `bounce` the bounce of the launch, used by the stochastic functor only.
template <class C, class RV, class T>
DI void apply_physics(const RV& S, V3<T> hl, V3<T> d, V3<T> n, V3<T>& nd, T& im, uint64_t ridx, uint32_t bounce = 0u) {
  const T zero = splat<T>(0.0f), one = splat<T>(1.0f);
  switch (C::pk(S.r)) {
    case RTT_PHYS_BLOCK:  // phys/std.py:243-254
      nd = {zero, zero, zero};
      im = zero;
      return;
    case RTT_PHYS_REFLECT: {  // phys/std.py:97-108
      const T two_cos = 2.0f * dot3(d, n);
      nd = d - scale(two_cos, n);
      im = one;
      return;
    }
    case RTT_PHYS_SNELL: {  // phys/std.py:123-145
      T dot = dot3(d, n);
      const auto ent = lt(dot, zero);
      T c1 = abs_(dot);
      const T mu = sel(ent, T(S.phys(2)), T(S.phys(3)));
      T term = 1.0f - mu * mu * (1.0f - c1 * c1);
      // sqrt(relu(term)); a NaN term comes with a NaN c1, so the result is NaN through g either way
      T c2 = sqrt_(max_(term, zero));
      if constexpr (IsPlainLane<T>::value) {
        // Within a few degrees of the critical angle the root amplifies (d c2 = d term / 2 c2) the one or two ulp by
        // which the hardware reciprocal and square root leave the normal off unit length.  For those rays the normal
        // is brought to unit length to second order, n (3 - n.n) / 2, and the root is the refined one; for all
        // others this costs a compare.  (The sign of d.n does not depend on the length of n.)
        const auto near = lt(abs_(term), splat<T>(0.03f));
        if (!kStrict && any_(near)) {
          const T fix = 1.5f - 0.5f * dot3(n, n);
          const V3<T> nu = scale(fix, n);
          const T dot_u = dot3(d, nu);
          const T c1_u = abs_(dot_u);
          const T term_u = 1.0f - mu * mu * (1.0f - c1_u * c1_u);
          n = sel3(near, nu, n);
          dot = sel(near, dot_u, dot);
          c1 = sel(near, c1_u, c1);
          term = sel(near, term_u, term);
          c2 = sel(near, sqrt_disc(max_(term_u, zero)), c2);
        }
      }
      // (mu c1 - c2) * (entering ? n : -n), the sign moved onto the scalar factor (exact)
      const T g = sel(ent, mu * c1 - c2, c2 - mu * c1);
      const V3<T> refr = scale(mu, d) + scale(g, n);
      const auto tir = lt(term, zero);
      nd = refr;
      if (any_(tir)) {
        const T two_dot = 2.0f * dot;
        nd = sel3(tir, d - scale(two_dot, n), refr);
      }
      im = one;
      return;
    }
    case RTT_PHYS_FRESNEL: {  // phys/std.py:177-219
      const T dot = dot3(d, n);
      const auto ent = lt(dot, zero);
      const T ci = abs_(dot);
      const T n1 = sel(ent, T(S.phys(0)), T(S.phys(1)));
      const T n2 = sel(ent, T(S.phys(1)), T(S.phys(0)));
      const T mu = sel(ent, T(S.phys(2)), T(S.phys(3)));  // n2 / n1
      const T sin2 = mu * mu * (1.0f - ci * ci);
      const auto tir = gt(sin2, splat<T>(1.0f));
      const T ct = sqrt_(max_(1.0f - sin2, zero));
      const T n1ci = n1 * ci, n2ct = n2 * ct, n1ct = n1 * ct, n2ci = n2 * ci;
      const T rs = quot(n1ci - n2ct, n1ci + n2ct + 1e-8f);
      const T rp = quot(n1ct - n2ci, n1ct + n2ci + 1e-8f);
      const T R = sel(tir, splat<T>(1.0f), 0.5f * (rs * rs + rp * rp));
      const uint64_t key = ((uint64_t)S.r.seed[1] << 32 | S.r.seed[0]) + ((uint64_t)S.r.element_id << 8) + (uint64_t)S.r.surface_id;
      const T u = fresnel_uniform(key, ridx, bounce, (T*)nullptr);
      const auto reflect = lt(u, R);
      const T g = sel(ent, mu * ci - ct, ct - mu * ci);  // (mu ci - ct) * (entering ? n : -n)
      const V3<T> refr = scale(mu, d) + scale(g, n);
      const T two_dot = 2.0f * dot;
      nd = sel3(reflect, d - scale(two_dot, n), refr);
      im = one;
      return;
    }
    case RTT_PHYS_LINEAR: {  // phys/std.py:72-88
      V3<T> dl = rowmat<2>(d, S, false);
      // correctly rounded quotients and root: a paraxial element is usually followed by a long free path to
      // its focus, which multiplies every ulp of the outgoing slope
      const T nx = T(S.phys(0)) * hl.x + T(S.phys(2)) * div_ieee(dl.x, dl.z);
      const T ny = T(S.phys(1)) * hl.y + T(S.phys(3)) * div_ieee(dl.y, dl.z);
      // F.normalize: v / max(|v|, 1e-12); |v| >= 1 here
      const T len = sqrt_ieee(nx * nx + ny * ny + 1.0f);
      nd = rowmat_t<2>(V3<T>{div_ieee(nx, len), div_ieee(ny, len), div_ieee(splat<T>(1.0f), len)}, S, false);
      im = one;
      return;
    }
    case RTT_PHYS_APERTURE: {  // phys/filter.py:24-33
      const T m = sel(surface_in_bounds<C>(S, hl), one, zero);
      nd = scale(m, d);
      im = m;
      return;
    }
    default:  // RTT_PHYS_TRANSMIT, phys/std.py:227-235
      nd = d;
      im = one;
      return;
  }
}

// ------------------------------------------------------------------------------------------------
// Source formulas of Bundle.sample (rays/bundle.py:30-171), used by the sampling kernels further down and by the
// sequential trace that generates its own rays.
// ------------------------------------------------------------------------------------------------
// A ray's values must not depend on the kernel these formulas are inlined into (the sampling kernel, or the trace that
// generates its own rays), so no product here may be contracted with a following sum at the compiler's discretion
// (the build allows contraction everywhere): pr() hands a product on as an opaque value, which keeps every operation
// the individually rounded one of the reference's torch ops.
DI float pr(float x) {
  asm("" : "+v"(x));
  return x;
}
// torch.distributions.Uniform.rsample: low + rand * (high - low)
DI float uniform_between(float lo, float hi, float u) { return lo + pr(u * (hi - lo)); }

constexpr float kTwoPi = 6.283185307179586f;

DI void sample_local(const RttSource& src, float ua, float ub, V3<float>& p, V3<float>& d) {
  p = {0.0f, 0.0f, 0.0f};
  d = {0.0f, 0.0f, 1.0f};
  switch (src.kind) {
    case RTT_SRC_DISK: {  // rays/bundle.py:47-56: theta is drawn before r
      const float theta = uniform_between(src.par[2], src.par[3], ua);
      const float r = sqrtf(uniform_between(src.par[0], src.par[1], ub));
      float sn, cs;
      sincosf(theta, &sn, &cs);
      p = {r * cs, r * sn, 0.0f};
      break;
    }
    case RTT_SRC_LINE:  // rays/bundle.py:111-118
      p.x = uniform_between(src.par[0], src.par[1], ua);
      break;
    case RTT_SRC_FAN: {  // rays/bundle.py:136-140
      const float theta = uniform_between(src.par[0], src.par[1], ua);
      float sn, cs;
      sincosf(theta, &sn, &cs);
      d = {0.0f, sn, cs};
      break;
    }
    case RTT_SRC_POINT: {  // rays/bundle.py:65-75, :162-171: phi is drawn before theta
      const float phi = acosf(pr(-2.0f * uniform_between(src.par[0], src.par[1], ua)) + 1.0f);
      const float theta = uniform_between(src.par[2], src.par[3], ub);
      float sp, cp, st, ct;
      sincosf(phi, &sp, &cp);
      sincosf(theta, &st, &ct);
      d = {ct * sp, st * sp, cp};
      break;
    }
    case RTT_SRC_GAUSS: {  // rays/beam.py:37-41 with the normals by Box-Muller
      const float r = sqrtf(-2.0f * logf(1.0f - ua));
      float sn, cs;
      sincosf(kTwoPi * ub, &sn, &cs);
      p = {src.par[0] * (r * cs), src.par[1] * (r * sn), 0.0f};
      break;
    }
    case RTT_SRC_RAYLEIGH: {  // rays/particle.py:111-125
      const float q = 2.0f - pr(4.0f * ua);
      const float A = powf(sqrtf(pr(q * q) + 1.0f) - q, 1.0f / 3.0f);
      const float ct = A - 1.0f / A;
      const float st = sqrtf(fmaxf(1.0f - pr(ct * ct), 0.0f));
      float sn, cs;
      sincosf(kTwoPi * ub, &sn, &cs);
      d = {st * cs, st * sn, ct};
      break;
    }
    default:
      break;
  }
}

// cosine-weighted hemisphere about +z (rays/panels.py:27-37, rays/particle.py:62-68)
DI V3<float> lambert_local(float u1, float u2) {
  const float r = sqrtf(u1);
  float sn, cs;
  sincosf(kTwoPi * u2, &sn, &cs);
  return {r * cs, r * sn, sqrtf(fmaxf(1.0f - u1, 0.0f))};
}

// the sources that draw four uniforms: (ua, ub) place the ray, (uc, ud) aim it
DI void sample_local_wide(const RttSource& src, float ua, float ub, float uc, float ud, V3<float>& p, V3<float>& d) {
  if (src.kind == RTT_SRC_SPHERE) {  // rays/particle.py:34-87
    const float ct = pr(2.0f * ua) - 1.0f;
    const float st = sqrtf(fmaxf(1.0f - pr(ct * ct), 0.0f));
    float sn, cs;
    sincosf(kTwoPi * ub, &sn, &cs);
    const V3<float> n = {st * cs, st * sn, ct};
    p = scale(src.par[0], n);
    const V3<float> l = lambert_local(uc, ud);
    const float v = n.z + 1e-10f;
    const float sg = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);  // torch.sign
    const float a = -1.0f / (sg + n.z);
    const float b = n.x * n.y * a;
    const V3<float> tg = {1.0f + pr(sg * (n.x * n.x) * a), sg * b, -sg * n.x};
    const V3<float> bt = {b, sg + pr((n.y * n.y) * a), -n.y};
    d = {pr(l.x * tg.x) + pr(l.y * bt.x) + pr(l.z * n.x), pr(l.x * tg.y) + pr(l.y * bt.y) + pr(l.z * n.y),
         pr(l.x * tg.z) + pr(l.y * bt.z) + pr(l.z * n.z)};
    return;
  }
  RttSource part = src;  // RTT_SRC_PANEL: an area sampler and an emission model, each a two-uniform source
  V3<float> unused;
  if (src.area_kind == RTT_AREA_RING) {
    part.kind = RTT_SRC_DISK;
    sample_local(part, ua, ub, p, unused);
  } else {  // rays/panels.py:118-122
    p = {uniform_between(src.par[0], src.par[1], ua), uniform_between(src.par[2], src.par[3], ub), 0.0f};
  }
  if (src.emit_kind == RTT_EMIT_CONE) {  // rays/panels.py:40-62
    part.kind = RTT_SRC_POINT;
#pragma unroll
    for (int k = 0; k < 4; ++k) part.par[k] = src.epar[k];
    sample_local(part, uc, ud, unused, d);
  } else {
    d = lambert_local(uc, ud);
  }
}

struct BundleOut {
  float* __restrict__ pos;
  float* __restrict__ dir;
  int64_t ld;
  float* __restrict__ inten;
  int8_t* __restrict__ id;
  float* __restrict__ wavelength;
};

// The sources that shine along their own axis (pencil, disk, line, Gaussian beam) give every ray the same global
// direction, the normalised third column of the pose: the launcher works it out once on the host, with the very
// operations of pose_direction() below (individually rounded IEEE products, sums, square root and quotients, so the
// bits are the ones a lane would get), and the kernels take it from their arguments.
struct PosedAxis {
  int32_t known;
  float d[3];
};

// dir @ R^T, then F.normalize (geom/transform.py:262, rays/ray.py:22-25); also compiled for the host
template <class Opaque>
__host__ __device__ inline void pose_direction(const float* R, float dx, float dy, float dz, float out[3], Opaque keep) {
  float g[3];
  for (int i = 0; i < 3; ++i) g[i] = keep(keep(dx * R[3 * i]) + keep(dy * R[3 * i + 1])) + keep(dz * R[3 * i + 2]);
  const float len = fmaxf(sqrtf(keep(keep(g[0] * g[0]) + keep(g[1] * g[1])) + keep(g[2] * g[2])), 1e-12f);
  for (int i = 0; i < 3; ++i) out[i] = g[i] / len;
}

// local sample -> global ray: RayTransformBundle.transform_ (pos @ R^T + t, dir @ R^T, geom/transform.py:259-262)
// and F.normalize of Rays.__post_init__ (dir / max(|dir|, 1e-12), rays/ray.py:22-25)
DI void pose_sampled_ray(const RttSource& src, const PosedAxis& axis, V3<float> p, V3<float> d, V3<float>& pg, V3<float>& dg) {
  const float* R = src.R;
  auto row = [&](V3<float> v, int i) { return pr(v.x * R[3 * i]) + pr(v.y * R[3 * i + 1]) + pr(v.z * R[3 * i + 2]); };
  pg = {row(p, 0) + src.t[0], row(p, 1) + src.t[1], row(p, 2) + src.t[2]};
  if (axis.known) {
    dg = {axis.d[0], axis.d[1], axis.d[2]};
    return;
  }
  float o[3];
  pose_direction(R, d.x, d.y, d.z, o, [](float x) { return pr(x); });
  dg = {o[0], o[1], o[2]};
}

struct RayIO {
  const float* pos_in;
  const float* dir_in;
  const float* inten_in;
  int64_t ld_in;
  float* pos_out;
  float* dir_out;
  float* inten_out;
  int64_t ld_out;
  int64_t n;
};

struct SensorIO {
  float* hit;
  float* w;
  uint8_t* mask;
  int64_t ld;
  int64_t index0;  // index in the call of the ray that element 0 of these (possibly tile-shifted) planes holds
  // append mode (multi-bounce kernel): hits go to a compact list instead of the dense per-ray slots
  float* app_hit;      // [cap][4]: hit_local x, y, z and the intensity before the surface
  int32_t* app_key;    // [cap][2]: bounce * RTT_MAX_SURFACES + sensor slot, index of the ray in the call
  int32_t* app_count;  // number of hits offered so far (may exceed cap: the host then retries with more room)
  int32_t app_cap, bounce;
  // The list may be kept in app_parts + 1 parts (a power of two minus one as a mask, 0 = one list) of app_cap entries
  // each, part p with its counter at app_count[kHitCounterStride p]: a block appends to part blockIdx & app_parts, so
  // the thousands of wavefronts that meet a sensor in the same bounce do not all add to one address.
  int32_t app_parts;
  // first spot moments on the side (dense mode): lane_mom[0..2] += w, w x, w y of the lane's lit hits on slot mom_slot
  float* lane_mom;
  int32_t mom_slot;
  // append mode when the launch holds a compacted subset of the call's rays: the keys of the lane's rays
  // (the keys live in the lane's column of the block's LDS, kBlock entries apart: they are read at a sensor only)
  const int32_t* lane_key;
  // one bounce with dense slots: a ray writes its mask entry only on the sensor it goes to, so every lane first clears
  // the entries of its rays on the first zero_slots slots (0: the caller cleared the planes)
  int32_t zero_slots;
};

DI float hsum(float v) { return v; }
DI float hsum(f2 v) { return v.x + v.y; }

template <class T>
struct Ray {
  V3<T> p, d;
  T I;
};

// Lane type T at ray index i: float reads ray i; f2 reads rays i and i+1, with one 8-byte access when
// `vec` (planes 8-byte aligned) and both rays exist, else with two 4-byte accesses; a lane holding only
// the last ray of an odd-length bundle (nv == 1) duplicates it into its second component.
// The index type I is int64_t for absolute ray indices or uint32_t for offsets from a wave-uniform base
// pointer (a 32-bit lane offset lets the load take its base from scalar registers, no 64-bit address math).
template <class I> DI float ld_(const float* p, I i, bool, int, float*) { return p[i]; }
template <class I> DI f2 ld_(const float* p, I i, bool vec, int nv, f2*) {
  if (vec && nv == 2) return RTT_NONTEMPORAL ? __builtin_nontemporal_load(reinterpret_cast<const f2*>(p + i)) : *reinterpret_cast<const f2*>(p + i);
  return f2{p[i], p[i + (nv - 1)]};
}
template <class I> DI void st_(float* p, I i, float v, bool, int) { p[i] = v; }
template <class I> DI void st_(float* p, I i, f2 v, bool vec, int nv) {
  if (vec && nv == 2) {
    if (RTT_NONTEMPORAL) __builtin_nontemporal_store(v, reinterpret_cast<f2*>(p + i)); else *reinterpret_cast<f2*>(p + i) = v;
  } else {
    p[i] = v.x;
    if (nv == 2) p[i + 1] = v.y;
  }
}

template <class T, class I>
DI Ray<T> load_ray(const RayIO& io, I i, bool vec = false, int nv = LaneOf<T>::N) {
  Ray<T> r;
  T* tag = nullptr;
  r.p = {ld_(io.pos_in, i, vec, nv, tag), ld_(io.pos_in + io.ld_in, i, vec, nv, tag),
         ld_(io.pos_in + 2 * io.ld_in, i, vec, nv, tag)};
  r.d = {ld_(io.dir_in, i, vec, nv, tag), ld_(io.dir_in + io.ld_in, i, vec, nv, tag),
         ld_(io.dir_in + 2 * io.ld_in, i, vec, nv, tag)};
  r.I = ld_(io.inten_in, i, vec, nv, tag);
  return r;
}

template <class T, class I>
DI void store_ray(const RayIO& io, I i, const Ray<T>& r, bool vec = false, int nv = LaneOf<T>::N) {
  st_(io.pos_out, i, r.p.x, vec, nv);
  st_(io.pos_out + io.ld_out, i, r.p.y, vec, nv);
  st_(io.pos_out + 2 * io.ld_out, i, r.p.z, vec, nv);
  st_(io.dir_out, i, r.d.x, vec, nv);
  st_(io.dir_out + io.ld_out, i, r.d.y, vec, nv);
  st_(io.dir_out + 2 * io.ld_out, i, r.d.z, vec, nv);
  st_(io.inten_out, i, r.I, vec, nv);
}

// Position in an append list for every lane of the wavefront that offers an entry (`want`): one atomic add per
// wavefront (the first offering lane adds the number of offering lanes, every lane takes the base plus its rank
// among them) instead of one same-address atomic per lane, which serialise.  Lanes that do not offer get -1.  May be
// called under divergence: only the lanes that reach the call take part.
constexpr int kHitCounterStride = 32;  // int32 entries between the counters of two parts: one 128-byte line each

DI int append_slot(int32_t* count, bool want) {
  const uint64_t offering = __ballot(want);
  if (!want) return -1;
  const int lane = (int)(threadIdx.x & 63u);
  const int leader = __builtin_ctzll(offering);
  int base = 0;
  if (lane == leader) base = atomicAdd(count, (int)__popcll(offering));
  base = __shfl(base, leader, 64);
  return base + (int)__popcll(offering & ((1ull << lane) - 1ull));
}

template <class T, class M, class IX>
DI void record_sensor(const SensorIO& se, int slot, IX i, M hit, V3<T> hl, T I, int nv) {
  if (se.app_count) {
    const int part = (int)blockIdx.x & se.app_parts;
#pragma unroll
    for (int k = 0; k < LaneOf<T>::N; ++k) {
      int at = append_slot(se.app_count + kHitCounterStride * part, k < nv && mcomp(hit, k));
      if (at >= 0 && at < se.app_cap) {
        at += part * se.app_cap;
        float* h = se.app_hit + 4 * (int64_t)at;
        h[0] = comp(hl.x, k);
        h[1] = comp(hl.y, k);
        h[2] = comp(hl.z, k);
        h[3] = comp(I, k);
        se.app_key[2 * (int64_t)at] = se.bounce * RTT_MAX_SURFACES + slot;
        se.app_key[2 * (int64_t)at + 1] = se.lane_key ? se.lane_key[k * kBlock] : (int32_t)(se.index0 + (int64_t)i + k);
      }
    }
    return;
  }
  if (se.lane_mom && slot == se.mom_slot) {  // SpotSizeLoss's W, sum w x, sum w y (optim/goals.py:165-178) while the hit is in registers
    auto lit = and_(hit, gt(I, splat<T>(0.0f)));
#pragma unroll
    for (int k = 0; k < LaneOf<T>::N; ++k)
      if (k >= nv) lit = set_mask(lit, k, false);
    const T wm = sel(lit, I, splat<T>(0.0f));
    se.lane_mom[0] += hsum(wm);
    se.lane_mom[1] += hsum(wm * sel(lit, hl.x, splat<T>(0.0f)));
    se.lane_mom[2] += hsum(wm * sel(lit, hl.y, splat<T>(0.0f)));
  }
  const int64_t base = (int64_t)slot * se.ld;  // wave-uniform: the slot's planes are addressed by pointer
  uint8_t* mask = se.mask + base;
  float* hx = se.hit + 3 * base;
  float* hy = hx + se.ld;
  float* hz = hy + se.ld;
  float* w = se.w + base;
#pragma unroll
  for (int k = 0; k < LaneOf<T>::N; ++k) {
    if (k >= nv) break;
    const bool h = mcomp(hit, k);
    mask[i + k] = h ? 1 : 0;
    if (h) {
      hx[i + k] = comp(hl.x, k);
      hy[i + k] = comp(hl.y, k);
      hz[i + k] = comp(hl.z, k);
      w[i + k] = comp(I, k);
    }
  }
}

// Apply surface S to the ray(s) of a lane where hit: Element.forward + Rays.scatter_update
// (elements/parent.py:44-58, rays/ray.py:29-40, elements/sensor.py:35-37).
template <class C, class RV, class T, class M, class I>
DI void interact(const RV& S, const Frames<T>& f, T t_raw, M hit, Ray<T>& r, const SensorIO& se, I i,
                 int nv = LaneOf<T>::N) {
  const bool sensor = (S.r.flags & RTT_F_SENSOR) != 0;
  if (!sensor && !any_(hit)) return;  // lanes without a hit are done; a wavefront without hits skips the block
  Forward<T> fw = surface_forward<C>(S, r.p, r.d, f, t_raw);
  V3<T> nd;
  T im;
  apply_physics<C>(S, fw.hit_l, r.d, fw.n_g, nd, im, (uint64_t)(se.index0 + (int64_t)i), (uint32_t)se.bounce);
  if (sensor) record_sensor(se, S.r.sensor_slot, i, hit, fw.hit_l, r.I, nv);
  if (__all(all_(hit))) {  // every ray of the wavefront hits: plain moves, no selects
    r.p = fw.hit_g;
    r.d = nd;
    r.I = r.I * im;
  } else {
    r.p = sel3(hit, fw.hit_g, r.p);
    r.d = sel3(hit, nd, r.d);
    r.I = sel(hit, r.I * im, r.I);
  }
}

// test + interact of one surface (scene/sequential.py:19-34)
template <class C, class RV, class T, class I>
DI void visit_seq(const RV& S, Ray<T>& r, const SensorIO& se, I i, int nv) {
  Frames<T> f;
  T t_raw;
  const auto hit = test_surface<C>(S, r.p, r.d, f, t_raw);
  interact<C>(S, f, t_raw, hit, r, se, i, nv);
}

// ------------------------------------------------------------------------------------------------
// SequentialScene.simulate (scene/sequential.py:12-36)
// T = f2: lane l of a block works on the adjacent rays (2l, 2l+1) of its tile, read and written
// with 8-byte accesses (512 contiguous bytes per wavefront instruction) when the planes are 8-byte
// aligned (`vec`), with 4-byte accesses otherwise.  Every ray, including the last one of an odd-length
// bundle, goes through the same packed arithmetic, so a ray's result does not depend on the size or
// alignment of the batch it is traced in.
// ------------------------------------------------------------------------------------------------
// MOM: the first spot moments of sensor slot se.mom_slot are summed on the way (float per wavefront, float64 from
// there on) and each block adds its three sums to bin blockIdx.x % kMomBins of mom_partials[k * kMomBins + bin] (zeroed
// by the launcher): a few dozen float64 atomics per address, spread over the launch, and a short second kernel adds
// the bins.  (One row per block took the adding kernel 87 us for the 65536 blocks of a 32 Mi-ray launch.)
constexpr int kMomBins = 1024;

// GEN: the rays are not read but generated: lane l of a tile makes the pair of rays with global indices
// (offset + base + 2l, + 1) from one Philox block, exactly the values k_sample_bundle writes for them, traces them and
// stores the result together with the ids and wavelengths of the new bundle.  Bundle.sample + simulate then cost
// 33 B written per ray and nothing read.  The source's offset must be even.
struct GenIO {
  RttSource src;
  PosedAxis axis;
  int8_t* id;         // [n] or null
  float* wavelength;  // [n] or null
};

// TAPE: the trace leaves what a later reverse sweep needs to differentiate it without tracing again (the tape of
// torch autograd, reduced to what cannot be made again cheaply): for every surface s that some ray of an aligned group
// of kTapeGroup rays hits, the direction in front of the surface and the path length t of every ray of the group
// (t = -1 for a ray that misses: a kept root is > 1e-6), planes[(4 s + c) * ld + i], c = 0..2 direction, 3 path length;
// and per group the set of those surfaces as a bit mask, groups[i / kTapeGroup].  Nothing is written for a surface no ray
// of the group hits.  A group is the 128 rays of one wavefront (two rays per lane).
constexpr int kTapeGroup = 128;
struct TapeIO {
  float* planes;
  int64_t ld;
  uint64_t* groups;
};

template <class T, int G, bool LDS_PROG, bool MOM = false, bool GEN = false, bool TAPE = false>
__global__ void __launch_bounds__(kBlock) k_trace_seq(RayIO io, const RttSurface* __restrict__ prog_, int n_surf,
                                                      Ops ops, SensorIO se, int vec, double* __restrict__ mom_partials = nullptr,
                                                      GenIO gen = GenIO{}, TapeIO tape = TapeIO{}) {
  static_assert(!GEN || (G == 1 && LaneOf<T>::N == 2), "generated rays come in pairs, one pair per lane");
  static_assert(!TAPE || (G == 1 && LaneOf<T>::N == 2 && kBlock % 64 == 0), "a tape group is the two-ray lanes of one wavefront");
  __shared__ RttSurface lds_prog[LDS_PROG ? RTT_MAX_SURFACES : 1];
  __shared__ double mom_wave[MOM ? 3 * (kBlock / 64) : 1];
  float lane_mom[3] = {0.0f, 0.0f, 0.0f};
  if (LDS_PROG) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(prog_);
    uint32_t* dst = reinterpret_cast<uint32_t*>(lds_prog);
    for (int w = threadIdx.x; w < n_surf * (int)(sizeof(RttSurface) / 4); w += kBlock) dst[w] = src[w];
    __syncthreads();
  }
  // G groups of N rays per lane: group g of a lane sits kBlock*N rays after group g-1, so every access of
  // a wavefront stays contiguous; the G independent instruction streams share the scalar work of a visit
  // and fill each other's latency and hazard slots.
  constexpr int N = LaneOf<T>::N;
  const RttSurface* __restrict__ prog = (const RttSurface*)__builtin_assume_aligned(prog_, 256);
  const int64_t tile = (int64_t)kBlock * N * G;
  const int64_t stride = (int64_t)gridDim.x * tile;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < io.n; base += stride) {
    // the tile's planes are addressed from wave-uniform base pointers with 32-bit lane offsets
    const bool stores = !(MOM && !io.pos_out);
    const RayIO tio = {io.pos_in + base, io.dir_in + base, io.inten_in + base, io.ld_in,
                       stores ? io.pos_out + base : nullptr, stores ? io.dir_out + base : nullptr,
                       stores ? io.inten_out + base : nullptr, io.ld_out, io.n - base};
    SensorIO tse = {se.hit ? se.hit + base : nullptr, se.w ? se.w + base : nullptr,
                    se.mask ? se.mask + base : nullptr, se.ld, base};
    tse.lane_mom = MOM ? lane_mom : nullptr;
    tse.mom_slot = se.mom_slot;
    Ray<T> r[G];
    uint32_t idx[G];
    int nv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      idx[g] = ((uint32_t)g * kBlock + threadIdx.x) * N;
      const int64_t left = tio.n - (int64_t)idx[g];
      nv[g] = left >= N ? N : (left >= 1 ? 1 : 0);
    }
    // a full tile of aligned planes (wave-uniform condition) moves with one N*4-byte access per lane and
    // plane; the last tile of a bundle and unaligned planes take the general path
    const bool whole = vec != 0 && tio.n >= tile;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if constexpr (GEN) {
        // lanes past the end generate the rays that would follow and store nothing
        const uint64_t first = gen.src.offset + (uint64_t)base + idx[g];
        const Philox4 rnd = philox4x32_10(first >> 1, gen.src.seed);
        V3<float> pg[2], dg[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          V3<float> pl, dl;
          sample_local(gen.src, unit_uniform(rnd.x[2 * h]), unit_uniform(rnd.x[2 * h + 1]), pl, dl);
          pose_sampled_ray(gen.src, gen.axis, pl, dl, pg[h], dg[h]);
        }
        r[g].p = {T{pg[0].x, pg[1].x}, T{pg[0].y, pg[1].y}, T{pg[0].z, pg[1].z}};
        r[g].d = {T{dg[0].x, dg[1].x}, T{dg[0].y, dg[1].y}, T{dg[0].z, dg[1].z}};
        r[g].I = splat<T>(1.0f);
      } else if (whole) {
        r[g] = load_ray<T>(tio, idx[g], true, N);
      } else {
        // lanes past the end (nv == 0) trace a copy of the tile's first ray and store nothing
        r[g] = load_ray<T>(tio, nv[g] ? idx[g] : 0u, vec != 0 && nv[g] == N, nv[g] ? nv[g] : 1);
      }
    }
    uint64_t group_hits = 0ull;  // TAPE: surfaces hit by some ray of this wavefront's group (wave-uniform)
    for (int s = 0; s < n_surf; ++s) {
      typename std::conditional<LDS_PROG, RecL, RecF>::type S{LDS_PROG ? lds_prog[s] : prog[s]};
      if constexpr (TAPE) {
#define RTT_CALL(CFG)                                                                                   \
  {                                                                                                     \
    Frames<T> f;                                                                                        \
    T t_raw;                                                                                            \
    const auto hit = test_surface<CFG>(S, r[0].p, r[0].d, f, t_raw);                                    \
    if (__any(any_(hit))) {                                                                             \
      group_hits |= 1ull << s;                                                                          \
      if (nv[0]) {                                                                                      \
        float* tp = tape.planes + (int64_t)4 * s * tape.ld + base;                                      \
        const bool tv = whole || (vec != 0 && nv[0] == N);                                              \
        st_(tp, idx[0], r[0].d.x, tv, nv[0]);                                                           \
        st_(tp + tape.ld, idx[0], r[0].d.y, tv, nv[0]);                                                 \
        st_(tp + 2 * tape.ld, idx[0], r[0].d.z, tv, nv[0]);                                             \
        st_(tp + 3 * tape.ld, idx[0], sel(hit, t_raw, splat<T>(-1.0f)), tv, nv[0]);                     \
      }                                                                                                 \
    }                                                                                                   \
    interact<CFG>(S, f, t_raw, hit, r[0], tse, idx[0], nv[0]);                                          \
  }
        RTT_DISPATCH(op_of(ops, s), RTT_CALL)
#undef RTT_CALL
      } else {
#define RTT_CALL(CFG)                                              \
  _Pragma("unroll") for (int g = 0; g < G; ++g) visit_seq<CFG>(S, r[g], tse, idx[g], nv[g])
        RTT_DISPATCH(op_of(ops, s), RTT_CALL)
#undef RTT_CALL
      }
    }
    if constexpr (TAPE) {
      const int64_t first = base + (int64_t)(threadIdx.x & ~63u) * N;  // first ray of this wavefront's group
      if ((threadIdx.x & 63u) == 0u && first < io.n) tape.groups[first / kTapeGroup] = group_hits;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (MOM && !io.pos_out) {
        // the caller wants the sensor record and its moments only (a loss that never looks at the traced rays)
      } else if (whole) {
        store_ray<T>(tio, idx[g], r[g], true, N);
      } else if (nv[g]) {
        store_ray<T>(tio, idx[g], r[g], vec != 0, nv[g]);
      }
      if constexpr (GEN) {
        if (nv[g]) {
          if (gen.wavelength) st_(gen.wavelength + base, idx[g], splat<T>(0.0f), vec != 0, nv[g]);
          if (gen.id) {
            int8_t* id = gen.id + base + idx[g];
            if (vec != 0 && nv[g] == 2) {
              __builtin_nontemporal_store((uint16_t)((uint8_t)gen.src.ray_id * 0x0101u), reinterpret_cast<uint16_t*>(id));
            } else {
              id[0] = (int8_t)gen.src.ray_id;
              if (nv[g] == 2) id[1] = (int8_t)gen.src.ray_id;
            }
          }
        }
      }
    }
  }
  if (MOM) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float v = lane_mom[k];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) mom_wave[k * (kBlock / 64) + wv] = (double)v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      double sum = 0.0;
      for (int w = 0; w < kBlock / 64; ++w) sum += mom_wave[threadIdx.x * (kBlock / 64) + w];
      atomicAdd(&mom_partials[threadIdx.x * kMomBins + (blockIdx.x % kMomBins)], sum);
    }
  }
}

// The same trace with the position of every ray recorded after each surface (the per-bounce snapshots a
// Paths wrapper keeps, rays/ray.py:174-185): a visualisation aid, one ray per lane, generic visits.
// hist[(3 s + c) * hist_ld + i] = component c of ray i after surface s; hist_any[s] = 1 when any ray hit s.
__global__ void __launch_bounds__(kBlock) k_trace_seq_history(RayIO io, const RttSurface* __restrict__ prog, int n_surf,
                                                              SensorIO se, float* __restrict__ hist,
                                                              int32_t* __restrict__ hist_any, int64_t hist_ld) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < io.n; i += stride) {
    Ray<float> r = load_ray<float>(io, i);
    for (int s = 0; s < n_surf; ++s) {
      const RecF S{prog[s]};
      Frames<float> f;
      float t_raw;
      const bool hit = test_surface<CGeneric>(S, r.p, r.d, f, t_raw);
      interact<CGeneric>(S, f, t_raw, hit, r, se, i, 1);
      float* h = hist + (int64_t)3 * s * hist_ld + i;
      h[0] = r.p.x;
      h[hist_ld] = r.p.y;
      h[2 * hist_ld] = r.p.z;
      if (hit) hist_any[s] = 1;  // every writer stores the same value
    }
    store_ray<float>(io, i, r);
  }
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------
// Tangent of the sequential trace, contracted with upstream adjoints (backward of
// SequentialScene.simulate for the parameters of the scene).  Each lane re-traces its ray with dual
// numbers (K parameter directions at once) and accumulates
//     dtheta[k] += sum_c adj_pos[c][i] * d pos_c / d theta_k + adj_dir[c][i] * d dir_c / d theta_k
//                + sum over sensor surfaces the ray hits:  adj_sens[slot][c][i] * d hit_local_c / d theta_k
// in float64 (wave shuffle -> LDS -> one atomicAdd per block and k).  Masks, root choice, TIR and
// `entering` are piecewise constant and carry no derivative, as under autograd.
// ------------------------------------------------------------------------------------------------
struct TangentIO {
  const float* tan;      // [n_surf][kTanSlots][K]
  const float* adj_pos;  // [3][ld_adj] or null
  const float* adj_dir;  // [3][ld_adj] or null
  int64_t ld_adj;
  const float* adj_sens;  // [slots][3][sens_ld] or null
  int64_t sens_ld;
  double* dtheta;  // [K], or [3K + 2] in spot mode
  // spot mode (SPOT = true): the upstream adjoint of the spot-size loss is formed in the kernel from the
  // primal hit point on sensor slot `spot_slot`:  g = (x - cx, y - cy) (w/W) / sqrt(((x-cx)^2+(y-cy)^2) w/W)
  int spot_slot;
  float cx, cy, inv_w;
  // per-ray mode (PER_RAY = true, K = 6): the six directions are the components of the ray's own input
  // (pos x, y, z, dir x, y, z); the contraction is not summed over rays but stored, ray_adj[k][i] =
  // d(objective)/d(input component k of ray i)
  float* ray_adj;  // [6][ld_ray]
  int64_t ld_ray;
};

// NACC accumulators per lane: K (plain) or 3K + 2 (spot: A_k = sum g.dhit_k, Bx_k = sum (w/W) dx_k,
// By_k = sum (w/W) dy_k, Gx = sum gx, Gy = sum gy; the host forms dL/dtheta_k = A_k - Gx Bx_k - Gy By_k
// when the centroid is the reference point, optim/goals.py:176-182).
template <int K>
DI Dual<K> unit_seed(float v, int k) {
  Dual<K> d(v);
  if (k & 1) d.tp[k >> 1].y = 1.0f; else d.tp[k >> 1].x = 1.0f;
  return d;
}

// C is the visit configuration: the generic interpreter, or the same with both rotations known to be the identity
// for every record (the host checks), which frees the registers of the rotated frames.
template <int K, bool SPOT, bool PER_RAY = false, class C = CGeneric>
__global__ void __launch_bounds__(kBlock) k_trace_seq_tangent(RayIO io, const RttSurface* __restrict__ prog, int n_surf,
                                                              TangentIO tio) {
  static_assert(!PER_RAY || (K == 6 && !SPOT), "per-ray mode seeds the six ray components");
  using T = Dual<K>;
  constexpr int NACC = SPOT ? 3 * K + 2 : K;
  __shared__ double part[NACC][kBlock / 64];
  float acc[NACC];  // a lane adds at most n / (2048 * 256) terms here; the cross-lane sums are float64
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  // The visit below is a few thousand cycles of arithmetic at two wavefronts per SIMD, too few to hide a load:
  // the planes of the lane's next ray are requested before the arithmetic of the current one starts.
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  Ray<float> ahead = load_ray<float>(io, i < io.n ? i : 0);
  for (; i < io.n; i += stride) {
    const Ray<float> r0 = ahead;
    if (i + stride < io.n) ahead = load_ray<float>(io, i + stride);
    Ray<T> r;
    if (PER_RAY) {
      r.p = {unit_seed<K>(r0.p.x, 0), unit_seed<K>(r0.p.y, 1), unit_seed<K>(r0.p.z, 2)};
      r.d = {unit_seed<K>(r0.d.x, 3), unit_seed<K>(r0.d.y, 4), unit_seed<K>(r0.d.z, 5)};
    } else {
      r.p = {T(r0.p.x), T(r0.p.y), T(r0.p.z)};
      r.d = {T(r0.d.x), T(r0.d.y), T(r0.d.z)};
    }
    r.I = T(r0.I);
    for (int s = 0; s < n_surf; ++s) {
      const RecD<K> S{prog[s], tio.tan + (int64_t)s * kTanSlots * K};
      Frames<T> f;
      T t_raw;
      const bool hit = test_surface<C>(S, r.p, r.d, f, t_raw);
      if (!hit) continue;
      Forward<T> fw = surface_forward<C>(S, r.p, r.d, f, t_raw);
      V3<T> nd;
      T im;
      apply_physics<C>(S, fw.hit_l, r.d, fw.n_g, nd, im, (uint64_t)i);
      if (S.r.flags & RTT_F_SENSOR) {
        if (SPOT) {
          const float w = r.I.v;
          if (S.r.sensor_slot == tio.spot_slot && w > 0.0f) {
            const float wn = w * tio.inv_w;
            const float dx = fw.hit_l.x.v - tio.cx, dy = fw.hit_l.y.v - tio.cy;
            const float u = sqrtf((dx * dx + dy * dy) * wn);
            const float inv = u > 0.0f ? wn / u : 0.0f;
            const float gx = dx * inv, gy = dy * inv;
#pragma unroll
            for (int k = 0; k < K; ++k) {
              const float tx = fw.hit_l.x.tan(k), ty = fw.hit_l.y.tan(k);
              acc[k] += gx * tx + gy * ty;
              acc[K + k] += wn * tx;
              acc[2 * K + k] += wn * ty;
            }
            acc[3 * K] += gx;
            acc[3 * K + 1] += gy;
          }
        } else if (tio.adj_sens) {
          const float* a = tio.adj_sens + (int64_t)S.r.sensor_slot * 3 * tio.sens_ld + i;
          const float ax = a[0], ay = a[tio.sens_ld], az = a[2 * tio.sens_ld];
#pragma unroll
          for (int k = 0; k < K; ++k) acc[k] += ax * fw.hit_l.x.tan(k) + ay * fw.hit_l.y.tan(k) + az * fw.hit_l.z.tan(k);
        }
      }
      r.p = fw.hit_g;
      r.d = nd;
      r.I = r.I * im;
    }
    if (!SPOT && tio.adj_pos) {
      const float ax = tio.adj_pos[i], ay = tio.adj_pos[tio.ld_adj + i], az = tio.adj_pos[2 * tio.ld_adj + i];
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] += ax * r.p.x.tan(k) + ay * r.p.y.tan(k) + az * r.p.z.tan(k);
    }
    if (!SPOT && tio.adj_dir) {
      const float ax = tio.adj_dir[i], ay = tio.adj_dir[tio.ld_adj + i], az = tio.adj_dir[2 * tio.ld_adj + i];
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] += ax * r.d.x.tan(k) + ay * r.d.y.tan(k) + az * r.d.z.tan(k);
    }
    if (PER_RAY) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        tio.ray_adj[k * tio.ld_ray + i] = acc[k];
        acc[k] = 0.0f;
      }
    }
  }
  if (PER_RAY) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    const double v = wave_sum((double)acc[k]);
    if (lane == 0) part[k][wv] = v;
  }
  __syncthreads();
  if (threadIdx.x < NACC) {
    double sum = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) sum += part[threadIdx.x][w];
    atomicAdd(&tio.dtheta[threadIdx.x], sum);
  }
}

#include "rtt_adjoint.inl"

// ------------------------------------------------------------------------------------------------
// Scene.ray_cast / Scene.step (scene/base.py:144-235)
// ------------------------------------------------------------------------------------------------
// torch.min(t_matrix, dim=1): first index of the smallest value, NaN wins and sticks.
// Every ray tests every surface, so the cast runs on packed lanes (two rays per lane) like the
// sequential trace; `win` is an int per component.
DI int isel(bool m, int a, int b) { return m ? a : b; }
DI i2 isel(b2 m, int a, i2 b) { return i2{m[0] ? a : b.x, m[1] ? a : b.y}; }
template <class T> struct WinOf { using type = int; };
template <> struct WinOf<f2> { using type = i2; };
DI int icomp(int a, int) { return a; }
DI int icomp(i2 a, int k) { return k == 0 ? a.x : a.y; }
DI bool isel_lt(int s, int win) { return s < win; }
DI b2 isel_lt(int s, i2 win) { return __builtin_convertvector(i2(s) < win, b2); }
DI bool eq_(float a, float b) { return a == b; }
DI b2 eq_(f2 a, f2 b) { return __builtin_convertvector(a == b, b2); }

// Element table for culling in the nearest-hit search: element e owns the records first[e] .. first[e] +
// count[e] - 1 and lies inside the sphere (c[e], R[e]) in global coordinates; R < 0 marks an element without a
// usable bound (unbounded or inverted surfaces).  Built on the host from the surface records; n = 0 disables
// culling (more than kMaxCull elements, or no host copy of the program).
constexpr int kMaxCull = 32;
struct Cull {
  int n;
  uint8_t first[kMaxCull], count[kMaxCull];
  float c[kMaxCull][3], R[kMaxCull];
};

// torch.min over the row of distances, independent of the order in which the surfaces are visited: the
// smallest t wins, the lower surface index on equal t, and a NaN beats every number (the first NaN in program
// order wins), exactly as torch.min(t_matrix, dim=1) returns it.
template <class T>
DI void offer(T t, int s, T& tmin, typename WinOf<T>::type& win) {
  const auto lower = isel_lt(s, win);
  const auto t_nan = isnan_(t), m_nan = isnan_(tmin);
  const auto take = or_(and_(t_nan, or_(not_(m_nan), lower)),
                        and_(not_(or_(t_nan, m_nan)), or_(lt(t, tmin), and_(eq_(t, tmin), lower))));
  tmin = sel(take, t, tmin);
  win = isel(take, s, win);
}

template <class T, int NG = 0>
DI T test_one(const RttSurface* __restrict__ prog, int op, int s, V3<T> p, V3<T> d) {
  const RecF S{prog[s]};
  Frames<T> f;
  T t_raw = splat<T>(kInf), t = splat<T>(kInf);
#define RTT_CALL(CFG)                                          \
  {                                                            \
    const auto valid = test_surface<CFG>(S, p, d, f, t_raw);   \
    t = masked_t<CFG>(S, t_raw, valid);                        \
  }
  RTT_DISPATCH_NG(op, RTT_CALL)
#undef RTT_CALL
  return t;
}

// Ray against the bounding sphere (c, R) of an element, without square roots or divisions.
// With oc = c - p, a = d.d, b = oc.d, cc = oc.oc - R^2 the ray enters the sphere at (b - sqrt(b^2 - a cc)) / a.
template <class T>
struct SphereTerms {
  T a, b, cc;
};
// Sort key: b - R, a lower bound of the entry distance for a unit direction (any point q of the sphere on the
// ray has t = b + (q - c).d >= b - R).  Only the visiting order depends on it.
// Skip test: true when no point of the sphere can give this ray a distance <= tmin, i.e. the origin is
// outside, the direction is a direction (dead rays carry d = 0), and the ray either misses the sphere or
// enters it beyond tm (tmin with a margin that covers the rounding of these products): entry > tm  <=>
// u = b - tm a > 0 and u^2 > disc.  NaN anywhere makes the comparisons false, so the element is tested.  The caller
// rules out rays without a direction.
template <class T>
DI typename LaneOf<T>::M sphere_beyond(const SphereTerms<T>& q, T tm) {
  const T zero = splat<T>(0.0f);
  const T disc = q.b * q.b - q.a * q.cc;
  const T u = q.b - tm * q.a;
  const auto misses = or_(lt(disc, zero), le(q.b, zero));
  const auto beyond = and_(gt(u, zero), gt(u * u, disc));
  return and_(gt(q.cc, zero), or_(misses, beyond));
}

DI float first_comp(float v) { return v; }
DI float first_comp(f2 v) { return v.x; }

// The element table and the opcodes as the nearest-hit kernels hold them: lane e of every wavefront carries
// element e (bounding sphere, first record, record count) and lane s the opcode of surface s, fetched once per
// wavefront from the kernel arguments.  A wavefront-uniform entry is then a v_readlane away; reading it from the
// argument segment instead costs a chain of dependent loads per element and per surface test.
struct CullLanes {
  int n;      // elements (0: no culling)
  float R, cx, cy, cz;
  int fc;     // first record | record count << 8
  int op;     // lane s: opcode of surface s
};
DI CullLanes cull_lanes(const Cull& cull, const Ops& ops) {
  static_assert(RTT_MAX_SURFACES <= 64 && kMaxCull <= 64, "one lane per surface and per element");
  const int lane = threadIdx.x & 63;
  const int e = lane & (kMaxCull - 1);
  CullLanes L;
  L.n = cull.n;
  L.R = cull.R[e];
  L.cx = cull.c[e][0];
  L.cy = cull.c[e][1];
  L.cz = cull.c[e][2];
  L.fc = (int)cull.first[e] | ((int)cull.count[e] << 8);
  L.op = op_of(ops, lane & (RTT_MAX_SURFACES - 1));
  return L;
}
DI int lane_i(int v, int k) { return __builtin_amdgcn_readlane(v, k); }
DI float lane_f(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
DI float first_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
DI float push_f(int to, float v) { return __int_as_float(__builtin_amdgcn_ds_permute(to << 2, __float_as_int(v))); }

// `use_act`: rays whose `act` is false take no part in the search and do not hold an element back.
template <class T, int NG = 0>
DI void nearest(const RttSurface* __restrict__ prog, int n_surf, const CullLanes& L, V3<T> p, V3<T> d, T& tmin,
                typename WinOf<T>::type& win, uint64_t skip, typename LaneOf<T>::M act, bool use_act) {
  tmin = splat<T>(kInf);
  win = typename WinOf<T>::type(0x7fffffff);
  if (L.n == 0) {
    for (int s = 0; s < n_surf; ++s) {
      if ((skip >> s) & 1) continue;  // surfaces left out of the competition (wave-uniform)
      offer(test_one<T, NG>(prog, lane_i(L.op, s), s, p, d), s, tmin, win);
    }
  } else {
    // Front-to-back over the elements, in the order the wavefront's first ray meets their bounding spheres
    // (a heuristic: the rays of a wavefront are neighbours).  An element is skipped when, for every ray of the
    // wavefront, its sphere is missed or entered beyond the nearest hit found so far; since any t of the element
    // is at least the entry distance, a skipped element can neither win nor tie.  Every lane of the wavefront is
    // active here (the kernels clamp instead of masking), so lane e can work on element e.
    const int lane = threadIdx.x & 63;
    const float px = first_f(first_comp(p.x)), py = first_f(first_comp(p.y)), pz = first_f(first_comp(p.z));
    const float dx = first_f(first_comp(d.x)), dy = first_f(first_comp(d.y)), dz = first_f(first_comp(d.z));
    // Sort key of element `lane`: b - R, a lower bound of the entry distance for a unit direction; elements without
    // a bound come first.  The keys are compared as integers in an order that is total over every bit pattern
    // (NaN included), so the ranks are a permutation whatever the ray holds, and every element is visited once.
    const float kf = L.R >= 0.0f ? ((L.cx - px) * dx + (L.cy - py) * dy + (L.cz - pz) * dz) - L.R : -kInf;
    const uint32_t kb = (uint32_t)__float_as_int(kf);
    const int key = (int)((kb & 0x80000000u) ? ~kb : (kb | 0x80000000u)) ^ (int)0x80000000u;  // signed, monotone
    int rank = 0;  // number of elements visited before element `lane`
    for (int e = 0; e < L.n; ++e) {
      const int ke = lane_i(key, e);
      rank += (ke < key || (ke == key && e < lane)) ? 1 : 0;
    }
    rank = lane < L.n ? rank : lane;
    // lane k receives the element of rank k
    const float sR = push_f(rank, L.R), sx = push_f(rank, L.cx), sy = push_f(rank, L.cy), sz = push_f(rank, L.cz);
    const int sfc = __builtin_amdgcn_ds_permute(rank << 2, L.fc);
    // Per call: |d|^2, |d| with a margin for its rounding, and the rays that cannot be culled at all (a dead ray,
    // d = 0, "hits" wherever its origin passes the bounds; a ray that takes no part holds no element back).
    const T zero = splat<T>(0.0f);
    const T a = dot3(d, d);
    const T len = sqrt_(a) * 1.0001f;
    auto cullable = gt(a, splat<T>(1e-12f));
    for (int k = 0; k < L.n; ++k) {
      const float R = lane_f(sR, k);
      if (R >= 0.0f) {
        const V3<T> oc = {lane_f(sx, k) - p.x, lane_f(sy, k) - p.y, lane_f(sz, k) - p.z};
        const T b = dot3(oc, d);
        const T tm = tmin * 1.0001f + 1e-3f;  // margin for the rounding of these products
        // Quick test first: a point of the sphere on the ray has (b - R |d|) / |d|^2 <= t <= (b + R |d|) / |d|^2, so the
        // sphere lies beyond tm when b - R |d| > tm |d|^2 and behind the origin when b + R |d| < 0.  In a train of
        // elements along an axis that settles all but the neighbours of the ray's position.
        const T reach = R * len;
        auto out = and_(cullable, or_(gt(b - reach, tm * a), lt(b + reach, zero)));
        if (use_act) out = or_(out, not_(act));
        if (__all(all_(out))) continue;
        const SphereTerms<T> q = {a, b, dot3(oc, oc) - R * R};
        auto beyond = or_(out, and_(cullable, sphere_beyond(q, tm)));
        if (__all(all_(beyond))) continue;
      }
      const int fc = lane_i(sfc, k);
      const int s0 = fc & 0xff, s1 = s0 + (fc >> 8);
      for (int s = s0; s < s1; ++s) {
        if ((skip >> s) & 1) continue;
        offer(test_one<T, NG>(prog, lane_i(L.op, s), s, p, d), s, tmin, win);
      }
    }
  }
  win = isel(not_(lt(tmin, splat<T>(kInf))), -1, win);
}

// In the three nearest-hit kernels every lane of a wavefront stays in the loop for the same number of rounds:
// a lane past the end of the bundle works on a copy of the first ray and stores nothing (`live` false), so the
// wavefront-wide decisions of nearest() see all 64 lanes.
template <class T>
__global__ void __launch_bounds__(kBlock) k_ray_cast(const float* __restrict__ pos, const float* __restrict__ dir,
                                                     int64_t ld, int64_t n, const RttSurface* __restrict__ prog_,
                                                     int n_surf, Ops ops, Cull cull, int32_t* __restrict__ winner,
                                                     float* __restrict__ tmin_out, int vec) {
  constexpr int N = LaneOf<T>::N;
  const RttSurface* __restrict__ prog = (const RttSurface*)__builtin_assume_aligned(prog_, 256);
  const int64_t stride = (int64_t)gridDim.x * kBlock * N;
  const int64_t rounds = (n + stride - 1) / stride;
  const CullLanes cl = cull_lanes(cull, ops);
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t want = r * stride + ((int64_t)blockIdx.x * kBlock + threadIdx.x) * N;
    const bool live = want < n;
    const int64_t i = live ? want : 0;
    const int nv = (n - i) >= N ? N : 1;
    T* tag = nullptr;
    V3<T> p = {ld_(pos, i, vec != 0, nv, tag), ld_(pos + ld, i, vec != 0, nv, tag), ld_(pos + 2 * ld, i, vec != 0, nv, tag)};
    V3<T> d = {ld_(dir, i, vec != 0, nv, tag), ld_(dir + ld, i, vec != 0, nv, tag), ld_(dir + 2 * ld, i, vec != 0, nv, tag)};
    T tmin;
    typename WinOf<T>::type win;
    nearest<T>(prog, n_surf, cl, p, d, tmin, win, 0, mask_all<typename LaneOf<T>::M>(true), false);
    if (!live) continue;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if (k >= nv) break;
      if (winner) winner[i + k] = icomp(win, k);
      if (tmin_out) tmin_out[i + k] = comp(tmin, k);
    }
  }
}

// The rays of a lane through the surfaces they won (Element.forward + scatter_update, scene/base.py:213-235).
// `t` is the winner's distance from the cast itself: the reference recomputes it in Element.forward with
// the very operations of intersectTest, so both agree there by construction; re-solving here with a
// differently contracted instruction stream could flip the `t <= 1e-6` filter for rays that start on a
// surface (every bounce after the first) and turn a finite hit into an infinite one.
// `pend`: the rays that have a winner and carry intensity (active_mask = hit & (intensity > 0), scene/base.py:201).
// Rays of one wavefront may have won different surfaces: one surface per pass, taken from the first ray that is
// still waiting, with its record in scalar registers and the lane's rays on packed registers (adjacent rays mostly
// win the same surface, so both components of a lane are usually served by the same pass).  The loop runs on
// ballots, so its trip count is wave-uniform; the surface index goes through an opaque copy, otherwise the
// compiler uses the equality `win == s` to address the record per lane (vector loads, one register set per field)
// and drops the loop.
template <class T, int NG = 0>
DI void serve_winners(const RttSurface* __restrict__ prog, int opv, typename WinOf<T>::type win, T t,
                      typename LaneOf<T>::M pend, Ray<T>& r, const SensorIO& se, int64_t i, int nv) {
  constexpr int N = LaneOf<T>::N;
  using M = typename LaneOf<T>::M;
  uint64_t todo[N];
#pragma unroll
  for (int k = 0; k < N; ++k) todo[k] = __ballot(mcomp(pend, k));
  for (;;) {
    int s = -1;
#pragma unroll
    for (int k = N - 1; k >= 0; --k)
      if (todo[k]) s = __builtin_amdgcn_readlane(icomp(win, k), (int)__builtin_ctzll(todo[k]));
    if (s < 0) break;
    M mine = mask_all<M>(false);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const bool m = mcomp(pend, k) && icomp(win, k) == s;
      mine = set_mask(mine, k, m);
      todo[k] &= ~__ballot(m);
    }
    int su = s;
    asm volatile("" : "+s"(su));
    if (any_(mine)) {
      const RecF S{prog[su]};
#define RTT_CALL(CFG)                                  \
  {                                                    \
    Frames<T> f = to_frames<CFG>(S, r.p, r.d);         \
    interact<CFG>(S, f, t, mine, r, se, i, nv);        \
  }
      RTT_DISPATCH_NG(lane_i(opv, su), RTT_CALL)
#undef RTT_CALL
    }
  }
}

// Scene.simulate (scene/base.py:129-142) as ONE launch: every ray runs its own bounce loop in registers.
// The reference stops all rays together when none carries intensity; per ray that is the same thing, because
// a ray without a hit, or with intensity 0, is not touched by any later bounce (scene/base.py:201-235), so a
// lane leaves the loop when its ray missed or died.  HBM sees 28 B in and 28 B out per ray for the whole
// simulate instead of per bounce; sensor hits go to an append list keyed by (bounce, slot, ray).
// Scene.step (scene/base.py:180-235) is the same kernel with max_bounces = 1, dense sensor slots and the
// winners of the cast written out, so a loop of steps and one simulate run the same machine code per bounce.
// STAGE: one stage of a bounce loop that is run a few bounces at a time.  After a handful of bounces most rays of a
// bundle have left the scene or died while a few keep going, and a wavefront pays for a bounce as long as one of its
// rays moves.  A stage reads its rays from compact planes (the first one from the caller's planes), runs up to
// max_bounces_ bounces, then hands every ray that still moves on to the compact planes of the next stage (one atomic
// per wavefront for the places) and writes every ray that has come to rest to its place in the caller's output
// planes.  A ray carries its index in the call (sensor keys, final place) and its bounce count with it; its arithmetic
// does not depend on the rays it shares a wavefront with, so the result is that of the single loop bit for bit.
struct StageIO {
  const int32_t* count_in;  // device: number of rays in the compact input; null = the io.n rays of the call, in order
  const int32_t* orig_in;   // [count] index in the call of each ray (null = its position)
  const int32_t* taken_in;  // [count] bounces taken before this stage (null = 0)
  float* next_pos;          // compact planes of the next stage, leading dimension next_ld; null in the last stage
  float* next_dir;
  float* next_inten;
  int32_t* next_orig;
  int32_t* next_taken;
  int32_t* next_count;      // device counter of the next stage, zero at launch
  int64_t next_ld;
  int32_t bounce0;          // bounces of the loop that lie before this stage
  // A ray that comes to rest in a later stage sits anywhere in the call: seven 4-byte stores to its places in the
  // caller's planes would each dirty a 32-byte sector of its own.  Such a ray leaves one 32-byte record (position,
  // direction, intensity, bounce count) at its index in `rec` instead, the first stage marks in `late` who moved on,
  // and k_place_late writes the records to the planes in index order once the last stage is through.
  float* rec;               // [n][8]; null = every stage writes the planes itself
  uint8_t* late;            // [n]
};

#ifndef RTT_STAGE_WAVES
#define RTT_STAGE_WAVES 4
#endif
template <class T, bool SINGLE, bool STAGE = false, int NG = 0>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(STAGE ? RTT_STAGE_WAVES : 4, 4))) k_simulate_nearest(RayIO io, const RttSurface* __restrict__ prog_, int n_surf,
                                                             Ops ops, Cull cull, int max_bounces_,
                                                             int32_t* __restrict__ winner,
                                                             int32_t* __restrict__ bounces_out, SensorIO se, int vec,
                                                             StageIO st = StageIO{}) {
  constexpr int N = LaneOf<T>::N;
  using M = typename LaneOf<T>::M;
  const RttSurface* __restrict__ prog = (const RttSurface*)__builtin_assume_aligned(prog_, 256);
  const int64_t stride = (int64_t)gridDim.x * kBlock * N;
  const int64_t n_in = (STAGE && st.count_in) ? (int64_t)*st.count_in : io.n;
  // A late stage holds a few thousand rays from all over the scene: packed densely they would fill a handful of
  // wavefronts, each of which then tests every element in every bounce, one after the other.  Spread over the lanes
  // of the grid (every 2^shift-th lane holds rays) a wavefront has a few rays and a few elements, and the others run
  // beside it.  Only while the spread rays still fit the SIMDs at once: beyond that more wavefronts are more work.
  int shift = 0;
  if (STAGE && st.count_in)
    while (shift < 6 && ((n_in + N - 1) / N << (shift + 1)) <= (int64_t)2048 * 64) ++shift;  // up to 2048 wavefronts: no queueing
  const int64_t rounds = (((n_in + N - 1) / N << shift) * N + stride - 1) / stride;
  const CullLanes cl = cull_lanes(cull, ops);
  for (int64_t rnd = 0; rnd < rounds; ++rnd) {
    // a block whose first lane lies beyond the rays has nothing to do in this round or any later one (a late stage
    // fills a fraction of its grid: the others leave before they touch registers that spill)
    if ((((rnd * (stride / N) + (int64_t)blockIdx.x * kBlock) >> shift) * N) >= n_in) break;
    const int64_t slot = rnd * (stride / N) + (int64_t)blockIdx.x * kBlock + threadIdx.x;  // lane of the grid, all rounds
    const int64_t want = (slot >> shift) * N;
    const bool live = want < n_in && (slot & ((1 << shift) - 1)) == 0;
    const int64_t i = live ? want : 0;
    const int nv = (n_in - i) >= N ? N : 1;
    Ray<T> r = load_ray<T>(io, i, vec != 0, nv);
    bool open[N];   // the ray may still move: it has not missed yet
    int taken[N];   // bounces taken (STAGE: kept in LDS, see below)
    // STAGE: what a ray carries through the stages besides its state -- its index in the call, its bounce count, the
    // surface it met last -- is needed at a sensor and at the hand-over only.  It lives in the lane's column of the
    // block's LDS: in registers it pushes the bounce loop over what four wavefronts per SIMD may hold, and what the
    // compiler then spills travels to HBM and back.
    __shared__ int32_t s_carried[STAGE ? 3 * N * kBlock : 1];
    int32_t* const carried = s_carried + (STAGE ? threadIdx.x : 0);
#define RTT_ORIG(k) carried[(k) * kBlock]
#define RTT_TAKEN(k) carried[(N + (k)) * kBlock]
#define RTT_LAST(k) carried[(2 * N + (k)) * kBlock]
#pragma unroll
    for (int k = 0; k < N; ++k) {
      open[k] = live && k < nv;
      taken[k] = 0;
      if (STAGE) {
        const int64_t at = i + (k < nv ? k : 0);
        RTT_ORIG(k) = st.orig_in ? st.orig_in[at] : (int)(i + k);
        RTT_TAKEN(k) = st.taken_in ? st.taken_in[at] : 0;
        RTT_LAST(k) = 0;
      }
    }
    SensorIO seb = se;
    if (STAGE) seb.lane_key = carried;
    if (SINGLE && live) {
      for (int slot = 0; slot < se.zero_slots; ++slot) {
        uint8_t* m = se.mask + (int64_t)slot * se.ld + i;
        if (vec != 0 && nv == 2) {
          *reinterpret_cast<uint16_t*>(m) = 0;   // (ld and the ray index are even: the pair is 2-byte aligned)
        } else {
#pragma unroll
          for (int k = 0; k < N; ++k)
            if (k < nv) m[k] = 0;
        }
      }
    }
    const int max_bounces = SINGLE ? 1 : max_bounces_;  // SINGLE: Scene.step, the loop folds away
    for (int b = 0; b < max_bounces; ++b) {
      M act = mask_all<M>(false);
      bool any_act = false;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const bool a = open[k] && comp(r.I, k) > 0.0f;
        any_act |= a;
        act = set_mask(act, k, a);
      }
      if (!__any(any_act)) break;
      T tmin;
      typename WinOf<T>::type win;
      // when the cast result is wanted for every ray (also those without intensity), no ray is left out of
      // the culling decisions
      nearest<T, NG>(prog, n_surf, cl, r.p, r.d, tmin, win, 0, act, !(winner && b == 0));
      seb.bounce = (STAGE ? st.bounce0 : 0) + b;
      if (winner && b == 0 && live) {  // Scene.ray_cast result of the first bounce, for every ray
#pragma unroll
        for (int k = 0; k < N; ++k)
          if (k < nv) winner[i + k] = icomp(win, k);
      }
      M pend = mask_all<M>(false);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const bool a = mcomp(act, k);
        const bool won = icomp(win, k) >= 0;
        if (a && !won) open[k] = false;
        pend = set_mask(pend, k, a && won);
        if (STAGE) {
          if (a && won) {
            RTT_TAKEN(k) += 1;
            RTT_LAST(k) = icomp(win, k);
          }
        } else {
          taken[k] += (a && won) ? 1 : 0;
        }
      }
      serve_winners<T, NG>(prog, cl.op, win, tmin, pend, r, seb, i, nv);
    }
    if (STAGE) {
      // Places in the next stage's planes: one atomic per block (thousands of wavefronts adding to one address take
      // about 12 ns each, which at 4 Mi rays is more than a bounce costs), and within the block's range the rays are
      // laid out by the surface they met last: a counting sort in LDS (count per surface with the rank as the
      // returned value, prefix over the surfaces by one wavefront, one thread reserves the range).  The wavefronts
      // of the next stage, which take 128 consecutive rays, then mostly hold rays that sit on one surface instead of
      // a cross-section of the block.
      __shared__ int s_hist[RTT_MAX_SURFACES];
      __shared__ int s_base;
      const int lane = (int)(threadIdx.x & 63u);
      if (threadIdx.x < RTT_MAX_SURFACES) s_hist[threadIdx.x] = 0;
      __syncthreads();
      bool go_on[N];
      int rank[N], orig[N], last[N];
#pragma unroll
      for (int k = 0; k < N; ++k) {
        orig[k] = RTT_ORIG(k);
        taken[k] = RTT_TAKEN(k);
        last[k] = RTT_LAST(k);
        go_on[k] = live && k < nv && st.next_pos != nullptr && open[k] && comp(r.I, k) > 0.0f;
        rank[k] = go_on[k] ? atomicAdd(&s_hist[last[k] & (RTT_MAX_SURFACES - 1)], 1) : 0;
      }
      __syncthreads();
      if (threadIdx.x < 64) {
        static_assert(RTT_MAX_SURFACES == 64, "one lane per surface in the prefix");
        const int c = s_hist[lane];
        int incl = c;
        for (int off = 1; off < 64; off <<= 1) {
          const int up = __shfl_up(incl, off, 64);
          if (lane >= off) incl += up;
        }
        s_hist[lane] = incl - c;
        if (lane == 63) s_base = incl ? atomicAdd(st.next_count, incl) : 0;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const bool mine = live && k < nv;
        if (go_on[k]) {
          const int at = s_base + s_hist[last[k] & (RTT_MAX_SURFACES - 1)] + rank[k];
          st.next_pos[at] = comp(r.p.x, k);
          st.next_pos[st.next_ld + at] = comp(r.p.y, k);
          st.next_pos[2 * st.next_ld + at] = comp(r.p.z, k);
          st.next_dir[at] = comp(r.d.x, k);
          st.next_dir[st.next_ld + at] = comp(r.d.y, k);
          st.next_dir[2 * st.next_ld + at] = comp(r.d.z, k);
          st.next_inten[at] = comp(r.I, k);
          st.next_orig[at] = orig[k];
          st.next_taken[at] = taken[k];
        } else if (mine && st.rec && st.orig_in) {
          float4* q = reinterpret_cast<float4*>(st.rec + 8 * (int64_t)orig[k]);
          q[0] = make_float4(comp(r.p.x, k), comp(r.p.y, k), comp(r.p.z, k), comp(r.d.x, k));
          q[1] = make_float4(comp(r.d.y, k), comp(r.d.z, k), comp(r.I, k), __int_as_float(taken[k]));
        } else if (mine) {
          const int64_t o = orig[k];
          io.pos_out[o] = comp(r.p.x, k);
          io.pos_out[io.ld_out + o] = comp(r.p.y, k);
          io.pos_out[2 * io.ld_out + o] = comp(r.p.z, k);
          io.dir_out[o] = comp(r.d.x, k);
          io.dir_out[io.ld_out + o] = comp(r.d.y, k);
          io.dir_out[2 * io.ld_out + o] = comp(r.d.z, k);
          io.inten_out[o] = comp(r.I, k);
          if (bounces_out) bounces_out[o] = taken[k];
        }
        if (mine && st.late && !st.orig_in) st.late[orig[k]] = go_on[k] ? 1 : 0;   // (first stage: rays in order)
      }
      __syncthreads();  // the counts are reused by the block's next round
      continue;
    }
#undef RTT_ORIG
#undef RTT_TAKEN
#undef RTT_LAST
    if (!live) continue;
    store_ray<T>(io, i, r, vec != 0, nv);
    if (bounces_out) {
#pragma unroll
      for (int k = 0; k < N; ++k)
        if (k < nv) bounces_out[i + k] = taken[k];
    }
  }
}

// Last pass of the staged bounce loop: the rays that came to rest after the first stage, from their records to their
// places in the caller's planes.  In index order, so that what a wavefront writes to a plane lies in one stretch.
__global__ void __launch_bounds__(kBlock) k_place_late(const uint8_t* __restrict__ late, const float* __restrict__ rec,
                                                       RayIO io, int32_t* __restrict__ bounces_out) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x; o < io.n; o += stride) {
    if (!late[o]) continue;
    const float4* q = reinterpret_cast<const float4*>(rec + 8 * o);
    const float4 a = q[0], b = q[1];
    io.pos_out[o] = a.x;
    io.pos_out[io.ld_out + o] = a.y;
    io.pos_out[2 * io.ld_out + o] = a.z;
    io.dir_out[o] = a.w;
    io.dir_out[io.ld_out + o] = b.x;
    io.dir_out[2 * io.ld_out + o] = b.y;
    io.inten_out[o] = b.z;
    if (bounces_out) bounces_out[o] = __float_as_int(b.w);
  }
}

// Scene.simulate with tangents: the bounce loop of k_simulate_nearest, one ray per lane, where every ray is a
// dual number over K parameter directions.  The surface a ray goes to is decided by the nearest-hit search on the
// values alone (piecewise constant in the parameters, as under autograd: torch.min routes the gradient to the
// winning column); the visit of that surface is then the dual-number visit of the sequential tangent kernel.
// Outputs: final rays, d(pos, dir)/d(theta_k) as planes [6][K][ld_out], and per sensor hit the hit_local tangents
// [3][K] next to the appended hit record.  Intensity multipliers are 0/1 switches and carry no tangent.
struct NearestTangentIO {
  const float* tan;  // [n_surf][kTanSlots][K]
  float* ray_tan;    // [6][K][ld_out]
  float* hit_tan;    // [cap][3][K]
  int seed_rays;     // 1: directions 0..5 are the components of each ray's own input (K >= 6)
};

template <int K>
__global__ void __launch_bounds__(kBlock) k_simulate_nearest_tangent(RayIO io, const RttSurface* __restrict__ prog_, int n_surf,
                                                                     Ops ops, Cull cull, int max_bounces, SensorIO se,
                                                                     NearestTangentIO tio) {
  using T = Dual<K>;
  const RttSurface* __restrict__ prog = (const RttSurface*)__builtin_assume_aligned(prog_, 256);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t rounds = (io.n + stride - 1) / stride;
  const CullLanes cl = cull_lanes(cull, ops);
  for (int64_t rnd = 0; rnd < rounds; ++rnd) {
    const int64_t want = rnd * stride + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = want < io.n;
    const int64_t i = live ? want : 0;
    const Ray<float> r0 = load_ray<float>(io, i);
    Ray<T> r;
    r.p = {T(r0.p.x), T(r0.p.y), T(r0.p.z)};
    r.d = {T(r0.d.x), T(r0.d.y), T(r0.d.z)};
    if (K >= 6 && tio.seed_rays) {
      r.p = {unit_seed<K>(r0.p.x, 0), unit_seed<K>(r0.p.y, 1), unit_seed<K>(r0.p.z, 2)};
      r.d = {unit_seed<K>(r0.d.x, 3), unit_seed<K>(r0.d.y, 4), unit_seed<K>(r0.d.z, 5)};
    }
    r.I = T(r0.I);
    bool open = live;
    for (int b = 0; b < max_bounces; ++b) {
      const bool act = open && r.I.v > 0.0f;
      if (!__any(act)) break;
      float tmin;
      int win;
      nearest<float>(prog, n_surf, cl, V3<float>{r.p.x.v, r.p.y.v, r.p.z.v}, V3<float>{r.d.x.v, r.d.y.v, r.d.z.v}, tmin, win,
                     0, act, true);
      if (act && win < 0) open = false;
      bool pending = act && win >= 0;
      while (pending) {  // one surface per pass, its record and tangent rows addressed from scalar registers
        const int s = __builtin_amdgcn_readfirstlane(win);
        if (win == s) {
          pending = false;
          const RecD<K> S{prog[s], tio.tan + (int64_t)s * kTanSlots * K};
          Frames<T> f;
          T t_raw;
          // the dual solve repeats the intersection; should its filters disagree with the search on a borderline
          // root, the ray counts as a miss
          if (!test_surface<CGeneric>(S, r.p, r.d, f, t_raw)) {
            open = false;
            continue;
          }
          const Forward<T> fw = surface_forward<CGeneric>(S, r.p, r.d, f, t_raw);
          V3<T> nd;
          T im;
          apply_physics<CGeneric>(S, fw.hit_l, r.d, fw.n_g, nd, im, (uint64_t)i, (uint32_t)b);
          if (S.r.flags & RTT_F_SENSOR) {
            const int at = append_slot(se.app_count, true);
            if (at < se.app_cap) {
              float* h = se.app_hit + 4 * (int64_t)at;
              h[0] = fw.hit_l.x.v;
              h[1] = fw.hit_l.y.v;
              h[2] = fw.hit_l.z.v;
              h[3] = r.I.v;
              se.app_key[2 * (int64_t)at] = b * RTT_MAX_SURFACES + S.r.sensor_slot;
              se.app_key[2 * (int64_t)at + 1] = (int32_t)i;
              float* ht = tio.hit_tan + (int64_t)at * 3 * K;
#pragma unroll
              for (int k = 0; k < K; ++k) {
                ht[k] = fw.hit_l.x.tan(k);
                ht[K + k] = fw.hit_l.y.tan(k);
                ht[2 * K + k] = fw.hit_l.z.tan(k);
              }
            }
          }
          r.p = fw.hit_g;
          r.d = nd;
          r.I = T(r.I.v * im.v);
        }
      }
    }
    if (!live) continue;
    const Ray<float> out = {{r.p.x.v, r.p.y.v, r.p.z.v}, {r.d.x.v, r.d.y.v, r.d.z.v}, r.I.v};
    store_ray<float>(io, i, out);
    const T comps[6] = {r.p.x, r.p.y, r.p.z, r.d.x, r.d.y, r.d.z};
#pragma unroll
    for (int c = 0; c < 6; ++c) {
#pragma unroll
      for (int k = 0; k < K; ++k) tio.ray_tan[((int64_t)c * K + k) * io.ld_out + i] = comps[c].tan(k);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Renderer.render_3d (render/camera.py:191-257, :259-318): nearest hit over the surfaces that take part,
// the winner's global normal (Shape.forward / Surface.forward), colour = base colour of the surface times
// 0.3 + 0.7 |n . light|, clamped to [0, 1]; pixels without a hit keep the background.
// ------------------------------------------------------------------------------------------------
struct ShadeIO {
  const float* base;  // [n_surf][3] base colour per surface (device)
  uint64_t skip;      // bit s set: surface s does not take part (aperture elements)
  float light[3];
  float bg[3];
};

__global__ void __launch_bounds__(kBlock) k_render(const float* __restrict__ pos, const float* __restrict__ dir, int64_t ld,
                                                   int64_t n, const RttSurface* __restrict__ prog_, int n_surf, Ops ops,
                                                   Cull cull, ShadeIO sh, float* __restrict__ rgb) {
  const RttSurface* __restrict__ prog = (const RttSurface*)__builtin_assume_aligned(prog_, 256);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t rounds = (n + stride - 1) / stride;
  const CullLanes cl = cull_lanes(cull, ops);
  for (int64_t rnd = 0; rnd < rounds; ++rnd) {
    const int64_t want = rnd * stride + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = want < n;
    const int64_t i = live ? want : 0;
    const V3<float> p = {pos[i], pos[ld + i], pos[2 * ld + i]};
    const V3<float> d = {dir[i], dir[ld + i], dir[2 * ld + i]};
    float tmin;
    int win;
    nearest<float>(prog, n_surf, cl, p, d, tmin, win, sh.skip, true, false);
    if (!live) continue;
    V3<float> c = {sh.bg[0], sh.bg[1], sh.bg[2]};
    bool pending = win >= 0;
    while (pending) {  // one surface per pass, its record in scalar registers
      const int s = __builtin_amdgcn_readfirstlane(win);
      if (win == s) {
        const RecF S{prog[s]};
        const Frames<float> f = to_frames<CGeneric>(S, p, d);
        const Forward<float> fw = surface_forward<CGeneric>(S, p, d, f, tmin);
        const float diffuse = fw.n_g.x * sh.light[0] + fw.n_g.y * sh.light[1] + fw.n_g.z * sh.light[2];
        const float shade = 0.3f + 0.7f * fabsf(diffuse);
        const float* b = sh.base + 3 * s;
        c = {fminf(fmaxf(b[0] * shade, 0.0f), 1.0f), fminf(fmaxf(b[1] * shade, 0.0f), 1.0f),
             fminf(fmaxf(b[2] * shade, 0.0f), 1.0f)};
        pending = false;
      }
    }
    rgb[3 * i] = c.x;
    rgb[3 * i + 1] = c.y;
    rgb[3 * i + 2] = c.z;
  }
}

// ------------------------------------------------------------------------------------------------
// Element.intersectTest (elements/parent.py:30-42)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_intersect_test(const float* __restrict__ pos,
                                                           const float* __restrict__ dir, int64_t ld, int64_t n,
                                                           const RttSurface* __restrict__ prog, int n_surf,
                                                           float* __restrict__ t_out, int64_t ld_t) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    V3<float> p = {pos[i], pos[ld + i], pos[2 * ld + i]};
    V3<float> d = {dir[i], dir[ld + i], dir[2 * ld + i]};
    for (int s = 0; s < n_surf; ++s) {
      Frames<float> f;
      float t_raw;
      const RecF S{prog[s]};
      const bool valid = test_surface<CGeneric>(S, p, d, f, t_raw);
      t_out[(int64_t)s * ld_t + i] = masked_t<CGeneric>(S, t_raw, valid);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Element.forward / Shape.forward / Surface.forward on every ray (no mask)
// ------------------------------------------------------------------------------------------------
struct ForwardOut {
  float* t;
  float* hit_g;
  float* n_g;
  float* hit_l;
  float* new_dir;
  float* imult;
  int64_t ld;
};

__device__ __forceinline__ void store3(float* base, int64_t ld, int64_t i, V3<float> v) {
  if (base) {
    base[i] = v.x;
    base[ld + i] = v.y;
    base[2 * ld + i] = v.z;
  }
}

__global__ void __launch_bounds__(kBlock) k_surface_forward(const float* __restrict__ pos,
                                                            const float* __restrict__ dir, int64_t ld, int64_t n,
                                                            const RttSurface* __restrict__ prog, int index,
                                                            ForwardOut o) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const RecF S{prog[index]};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    V3<float> p = {pos[i], pos[ld + i], pos[2 * ld + i]};
    V3<float> d = {dir[i], dir[ld + i], dir[2 * ld + i]};
    Frames<float> f = to_frames<CGeneric>(S, p, d);
    const float t = solve_t<CGeneric>(S, f);
    Forward<float> fw = surface_forward<CGeneric>(S, p, d, f, t);
    if (o.t) o.t[i] = t;
    store3(o.hit_g, o.ld, i, fw.hit_g);
    store3(o.n_g, o.ld, i, fw.n_g);
    store3(o.hit_l, o.ld, i, fw.hit_l);
    if (o.new_dir || o.imult) {
      V3<float> nd;
      float im;
      apply_physics<CGeneric>(S, fw.hit_l, d, fw.n_g, nd, im, (uint64_t)i);
      store3(o.new_dir, o.ld, i, nd);
      if (o.imult) o.imult[i] = im;
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_apply_physics(const float* __restrict__ hl, const float* __restrict__ dir,
                                                          const float* __restrict__ nrm, int64_t ld, int64_t n,
                                                          const RttSurface* __restrict__ prog,
                                                          float* __restrict__ new_dir, float* __restrict__ imult) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const RecF S{prog[0]};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    V3<float> h = {hl[i], hl[ld + i], hl[2 * ld + i]};
    V3<float> d = {dir[i], dir[ld + i], dir[2 * ld + i]};
    V3<float> nn = {nrm[i], nrm[ld + i], nrm[2 * ld + i]};
    V3<float> nd;
    float im;
    apply_physics<CGeneric>(S, h, d, nn, nd, im, (uint64_t)i);
    store3(new_dir, ld, i, nd);
    imult[i] = im;
  }
}

// ------------------------------------------------------------------------------------------------
// SpotSizeLoss moments (optim/goals.py:165-183)
// ------------------------------------------------------------------------------------------------
// One entry of the spot-size sums.  pass 0: W, sum w x, sum w y; pass 1: sum sqrt(((x-cx)^2 + (y-cy)^2) w / W) with
// the per-ray term in float32 like the reference's tensors.  All sums in float64.
// pass 2: the same sum and, beside it, sum g with g = (x - cx, y - cy) (w/W) / sqrt(((x-cx)^2 + (y-cy)^2) w/W), the
// two sums the centroid term of the loss gradient needs (optim/goals.py:176-182 under autograd).
DI void spot_term(int pass, float xi, float yi, float wi, bool masked_in, double cx, double cy, float wsum,
                  double& a0, double& a1, double& a2) {
  if (!(masked_in && wi > 0.0f)) return;
  const double xd = xi, yd = yi, wd = wi;
  if (pass == 0) {
    a0 += wd;
    a1 += wd * xd;
    a2 += wd * yd;
  } else {
    const float dx = (float)(xd - cx), dy = (float)(yd - cy);
    const float wn = wi / wsum;
    const float u = sqrtf((dx * dx + dy * dy) * wn);
    a0 += (double)u;
    if (pass == 2) {
      const float inv = u > 0.0f ? wn / u : 0.0f;
      a1 += (double)(dx * inv);
      a2 += (double)(dy * inv);
    }
  }
}

// VEC: four consecutive entries per lane and trip with 16-byte loads (the planes are 16-byte aligned), so that a
// wavefront keeps 3 KB in flight instead of 0.8 KB: the pass is pure streaming and bound by the bytes in flight.
// Passes 1 and 2 with the per-ray terms in float32, like the reference's tensors (hardware sqrt and reciprocal, one
// multiplication by 1 / W instead of a division per ray); the caller adds the terms of a group of rays in float32
// and the groups in float64.
DI void spot_term_f(int pass, float xi, float yi, float wi, bool masked_in, float cx, float cy, float inv_w, float& g0,
                    float& g1, float& g2) {
  if (!(masked_in && wi > 0.0f)) return;
  const float dx = xi - cx, dy = yi - cy;
  const float wn = wi * inv_w;
  const float u = sqrt_((dx * dx + dy * dy) * wn);
  g0 += u;
  if (pass == 2) {
    const float inv = u > 0.0f ? wn * rcp_(u) : 0.0f;
    g1 += dx * inv;
    g2 += dy * inv;
  }
}

template <bool VEC, int BS = kBlock>
__global__ void __launch_bounds__(BS) k_spot_moments(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ w,
                                                         const uint8_t* __restrict__ mask, int64_t n, int pass,
                                                         double cx, double cy, double wsum,
                                                         double* __restrict__ out) {
  __shared__ double part[3][BS / 64];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  if (pass >= 3) {
    // passes 3 and 4 are pass 2 with the weight sum taken from out[0] (clamped like the reference's W) and, in pass 3,
    // the centre from out[1..2] / W: the first moments never leave the device
    wsum = fmax(out[0], 1e-12);
    if (pass == 3) {
      cx = out[1] / wsum;
      cy = out[2] / wsum;
    }
    pass = 2;
  }
  const float wsum_f = (float)wsum;
  const float cx_f = (float)cx, cy_f = (float)cy, inv_w = (float)(1.0 / wsum);
  constexpr int W = VEC ? 4 : 1;
  const int64_t stride = (int64_t)gridDim.x * BS * W;
  int64_t i = ((int64_t)blockIdx.x * BS + threadIdx.x) * W;
  if (VEC && pass != 0) {
    // two trips at once: both groups of loads are issued before the first is used (6 KB in flight per wavefront)
    for (; i + stride + 4 <= n; i += 2 * stride) {
      const int64_t j = i + stride;
      const float4 xa = *reinterpret_cast<const float4*>(x + i), xb = *reinterpret_cast<const float4*>(x + j);
      const float4 ya = *reinterpret_cast<const float4*>(y + i), yb = *reinterpret_cast<const float4*>(y + j);
      const float4 wa = *reinterpret_cast<const float4*>(w + i), wb = *reinterpret_cast<const float4*>(w + j);
      const uint32_t ma = mask ? *reinterpret_cast<const uint32_t*>(mask + i) : 0x01010101u;
      const uint32_t mb = mask ? *reinterpret_cast<const uint32_t*>(mask + j) : 0x01010101u;
      float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, h0 = 0.0f, h1 = 0.0f, h2 = 0.0f;
      spot_term_f(pass, xa.x, ya.x, wa.x, (ma & 0xffu) != 0, cx_f, cy_f, inv_w, g0, g1, g2);
      spot_term_f(pass, xa.y, ya.y, wa.y, (ma & 0xff00u) != 0, cx_f, cy_f, inv_w, g0, g1, g2);
      spot_term_f(pass, xa.z, ya.z, wa.z, (ma & 0xff0000u) != 0, cx_f, cy_f, inv_w, g0, g1, g2);
      spot_term_f(pass, xa.w, ya.w, wa.w, (ma & 0xff000000u) != 0, cx_f, cy_f, inv_w, g0, g1, g2);
      spot_term_f(pass, xb.x, yb.x, wb.x, (mb & 0xffu) != 0, cx_f, cy_f, inv_w, h0, h1, h2);
      spot_term_f(pass, xb.y, yb.y, wb.y, (mb & 0xff00u) != 0, cx_f, cy_f, inv_w, h0, h1, h2);
      spot_term_f(pass, xb.z, yb.z, wb.z, (mb & 0xff0000u) != 0, cx_f, cy_f, inv_w, h0, h1, h2);
      spot_term_f(pass, xb.w, yb.w, wb.w, (mb & 0xff000000u) != 0, cx_f, cy_f, inv_w, h0, h1, h2);
      a0 += (double)g0 + (double)h0;
      a1 += (double)g1 + (double)h1;
      a2 += (double)g2 + (double)h2;
    }
  }
  for (; i < n; i += stride) {
    if (VEC && i + 4 <= n) {
      const float4 xv = *reinterpret_cast<const float4*>(x + i);
      const float4 yv = *reinterpret_cast<const float4*>(y + i);
      const float4 wv = *reinterpret_cast<const float4*>(w + i);
      const uint32_t mv = mask ? *reinterpret_cast<const uint32_t*>(mask + i) : 0x01010101u;
      if (pass != 0) {
        float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
        spot_term_f(pass, xv.x, yv.x, wv.x, (mv & 0xffu) != 0, cx_f, cy_f, inv_w, g0, g1, g2);
        spot_term_f(pass, xv.y, yv.y, wv.y, (mv & 0xff00u) != 0, cx_f, cy_f, inv_w, g0, g1, g2);
        spot_term_f(pass, xv.z, yv.z, wv.z, (mv & 0xff0000u) != 0, cx_f, cy_f, inv_w, g0, g1, g2);
        spot_term_f(pass, xv.w, yv.w, wv.w, (mv & 0xff000000u) != 0, cx_f, cy_f, inv_w, g0, g1, g2);
        a0 += (double)g0;
        a1 += (double)g1;
        a2 += (double)g2;
        continue;
      }
      spot_term(pass, xv.x, yv.x, wv.x, (mv & 0xffu) != 0, cx, cy, wsum_f, a0, a1, a2);
      spot_term(pass, xv.y, yv.y, wv.y, (mv & 0xff00u) != 0, cx, cy, wsum_f, a0, a1, a2);
      spot_term(pass, xv.z, yv.z, wv.z, (mv & 0xff0000u) != 0, cx, cy, wsum_f, a0, a1, a2);
      spot_term(pass, xv.w, yv.w, wv.w, (mv & 0xff000000u) != 0, cx, cy, wsum_f, a0, a1, a2);
    } else {
      for (int k = 0; k < W && i + k < n; ++k)
        spot_term(pass, x[i + k], y[i + k], w[i + k], mask == nullptr || mask[i + k] != 0, cx, cy, wsum_f, a0, a1, a2);
    }
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    part[0][wv] = a0;
    part[1][wv] = a1;
    part[2][wv] = a2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < BS / 64; ++k) {
      s0 += part[0][k];
      s1 += part[1][k];
      s2 += part[2][k];
    }
    if (pass == 0) {
      atomicAdd(&out[0], s0);
      atomicAdd(&out[1], s1);
      atomicAdd(&out[2], s2);
    } else {
      atomicAdd(&out[3], s0);
      if (pass == 2) {
        atomicAdd(&out[4], s1);
        atomicAdd(&out[5], s2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Sensor.getSpotSizeParallel_xy (elements/sensor.py:87-176): per-ray-id sums over recorded hits.
// Group g = id + 128 (ids are int8).  pass 0: out[g] += (w, w x, w y); pass 1: out[g] += (w |x - cx_g|^p,
// w |y - cy_g|^p, 0) with the centres read from `centers` [256][2].  Sums are float64: a wavefront whose hits
// all carry one id (the usual case: a bundle has one id) reduces with shuffles, mixed wavefronts add into
// a per-block table in LDS; each block then adds its non-zero rows to `out`.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_spot_groups(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ w, const uint8_t* __restrict__ mask,
                                                        const int8_t* __restrict__ id, int64_t n, int pass,
                                                        const double* __restrict__ centers, float p,
                                                        double* __restrict__ out) {
  __shared__ double acc[256][3];
  for (int k = threadIdx.x; k < 256 * 3; k += kBlock) (&acc[0][0])[k] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t n_round = (n + stride - 1) / stride * stride;  // every lane of a wavefront runs every round
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_round; i += stride) {
    const bool on = i < n && (mask == nullptr || mask[i] != 0);
    const int g = on ? (int)id[i] + 128 : -1;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (on) {
      const float wi = w[i], xi = x[i], yi = y[i];
      if (pass == 0) {
        a0 = wi;
        a1 = (double)(xi * wi);  // per-hit products in float32 like the reference's tensors
        a2 = (double)(yi * wi);
      } else {
        const float dx = fabsf(xi - (float)centers[2 * g]), dy = fabsf(yi - (float)centers[2 * g + 1]);
        a0 = (double)(wi * (p == 2.0f ? dx * dx : powf(dx, p)));
        a1 = (double)(wi * (p == 2.0f ? dy * dy : powf(dy, p)));
      }
    }
    const uint64_t active = __ballot(on);
    if (active == 0) continue;
    const int g0 = __shfl(g, __builtin_ctzll(active), 64);
    if (__all(!on || g == g0)) {
      a0 = wave_sum(a0);
      a1 = wave_sum(a1);
      a2 = wave_sum(a2);
      if (lane == 0) {
        atomicAdd(&acc[g0][0], a0);
        atomicAdd(&acc[g0][1], a1);
        atomicAdd(&acc[g0][2], a2);
      }
    } else if (on) {
      atomicAdd(&acc[g][0], a0);
      atomicAdd(&acc[g][1], a1);
      atomicAdd(&acc[g][2], a2);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 256 * 3; k += kBlock) {
    const double v = (&acc[0][0])[k];
    if (v != 0.0) atomicAdd(&out[k], v);
  }
}

// ------------------------------------------------------------------------------------------------
// Bundle.sample on the device (rays/bundle.py:30-171).  One lane generates the pair of rays with global
// indices (2j, 2j+1) from one Philox4x32-10 block, applies the bundle transform and the normalisation
// of Rays.initialize, and writes the structure-of-arrays planes: a write-only kernel.
// ------------------------------------------------------------------------------------------------
DI void write_sampled_ray(const RttSource& src, const BundleOut& o, int64_t i, V3<float> pg, V3<float> dg) {
  o.pos[i] = pg.x; o.pos[o.ld + i] = pg.y; o.pos[2 * o.ld + i] = pg.z;
  o.dir[i] = dg.x; o.dir[o.ld + i] = dg.y; o.dir[2 * o.ld + i] = dg.z;
  if (o.inten) o.inten[i] = 1.0f;
  if (o.id) o.id[i] = (int8_t)src.ray_id;
  if (o.wavelength) o.wavelength[i] = 0.0f;
}

DI void store_sampled_ray(const RttSource& src, const BundleOut& o, int64_t i, V3<float> p, V3<float> d) {
  V3<float> pg, dg;
  pose_sampled_ray(src, PosedAxis{}, p, d, pg, dg);
  write_sampled_ray(src, o, i, pg, dg);
}

// `vec`: the planes are 8-byte aligned with an even leading dimension and the call starts on an even global index,
// so the two rays of a lane are one 8-byte, non-temporal store per plane (the bundle is written once and read by
// another kernel).
__global__ void __launch_bounds__(kBlock) k_sample_bundle(RttSource src, PosedAxis axis, BundleOut o, int64_t n, int vec) {
  const uint64_t first_pair = src.offset >> 1;
  const uint64_t pair = first_pair + (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const Philox4 rnd = philox4x32_10(pair, src.seed);
  V3<float> pg[2], dg[2];
  bool live[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint64_t g = 2 * pair + h;  // global ray index
    live[h] = g >= src.offset && g < src.offset + (uint64_t)n;
    V3<float> p, d;
    sample_local(src, unit_uniform(rnd.x[2 * h]), unit_uniform(rnd.x[2 * h + 1]), p, d);
    pose_sampled_ray(src, axis, p, d, pg[h], dg[h]);
  }
  const int64_t i = (int64_t)(2 * pair - src.offset);  // call index of the lane's first ray (-1 when it is not part of the call)
  if (vec && live[0] && live[1]) {
    auto put = [](float* plane, int64_t at, float a, float b) {
      __builtin_nontemporal_store(f2{a, b}, reinterpret_cast<f2*>(plane + at));
    };
    put(o.pos, i, pg[0].x, pg[1].x); put(o.pos + o.ld, i, pg[0].y, pg[1].y); put(o.pos + 2 * o.ld, i, pg[0].z, pg[1].z);
    put(o.dir, i, dg[0].x, dg[1].x); put(o.dir + o.ld, i, dg[0].y, dg[1].y); put(o.dir + 2 * o.ld, i, dg[0].z, dg[1].z);
    if (o.inten) put(o.inten, i, 1.0f, 1.0f);
    if (o.wavelength) put(o.wavelength, i, 0.0f, 0.0f);
    if (o.id) {
      const uint16_t two = (uint16_t)((uint8_t)src.ray_id * 0x0101u);
      __builtin_nontemporal_store(two, reinterpret_cast<uint16_t*>(o.id + i));
    }
    return;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (live[h]) write_sampled_ray(src, o, i + h, pg[h], dg[h]);
}

// four uniforms per ray: one Philox block each
__global__ void __launch_bounds__(kBlock) k_sample_bundle_wide(RttSource src, BundleOut o, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const Philox4 rnd = philox4x32_10(src.offset + (uint64_t)i, src.seed, 1u);
  V3<float> p, d;
  sample_local_wide(src, unit_uniform(rnd.x[0]), unit_uniform(rnd.x[1]), unit_uniform(rnd.x[2]), unit_uniform(rnd.x[3]), p, d);
  store_sampled_ray(src, o, i, p, d);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// tuning knobs (rtt_tune): rays carried per lane in the sequential trace, and the grid cap
int g_rays_per_lane = 2;
int g_program_in_lds = 0;  // 1: stage the surface program in LDS instead of reading it with scalar loads
// One tile per block by default: blocks that finish are replaced by blocks that start with their
// loads, which overlaps HBM latency with the arithmetic of the resident waves (measured on the 64 Mi
// singlet trace: 0.83 ms with a 2048-block grid-stride cap, 0.77 ms uncapped).
int g_max_blocks = 1 << 20;
int g_cull = 1;  // 0: the nearest-hit kernels test every surface for every ray (no bounding-sphere culling)

int grid_for(int64_t n, int rays_per_lane = 1) {
  const int64_t tile = (int64_t)kBlock * rays_per_lane;
  int64_t blocks = (n + tile - 1) / tile;
  if (blocks > g_max_blocks) blocks = g_max_blocks;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// Kernels that end in one atomicAdd per block keep a bounded grid (grid-stride loop over the rays):
// a hundred thousand blocks adding to the same few addresses serialise on the atomics.
int g_reduction_blocks = 2048;
int grid_for_reduction(int64_t n) {
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > g_reduction_blocks) blocks = g_reduction_blocks;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// Pick the visit configuration of one record: a specialised one only when every kind and flag it
// fixes at compile time matches; anything else runs the generic interpreter.
int classify(const RttSurface& S) {
  const int all_id = RTT_F_SHAPE_ROT_ID | RTT_F_SURF_ROT_ID;
  const bool ident = (S.flags & all_id) == all_id;
  const bool shaped = (S.flags & RTT_F_HAS_SHAPE) != 0;
  auto is = [&](int sk, int bk, int hk) { return S.surf_kind == sk && S.bound_kind == bk && S.shape_kind == hk; };
  if (shaped && ident && is(RTT_SURF_QUADRIC, RTT_BND_HALF, RTT_SHP_SPHERIC_FACE) && S.phys_kind == RTT_PHYS_SNELL)
    return OP_FACE_SNELL;
  if (shaped && ident && is(RTT_SURF_CYLINDER, RTT_BND_NONE, RTT_SHP_SPHERIC_EDGE)) {
    if (S.phys_kind == RTT_PHYS_BLOCK) return OP_EDGE_BLOCK;
    if (S.phys_kind == RTT_PHYS_SNELL) return OP_EDGE_SNELL;
  }
  if (!shaped && ident && is(RTT_SURF_PLANE, RTT_BND_DISK, RTT_SHP_NONE)) return OP_DISK_PLANE;
  if (!shaped && ident && is(RTT_SURF_QUADRIC, RTT_BND_HALF_DISK, RTT_SHP_NONE) && S.phys_kind == RTT_PHYS_REFLECT)
    return OP_MIRROR_CAP;
  if (shaped && ident && is(RTT_SURF_QUADRIC_ZY, RTT_BND_HALF, RTT_SHP_CYL_FACE) && S.phys_kind == RTT_PHYS_SNELL)
    return OP_CYL_FACE_SNELL;
  if (shaped && is(RTT_SURF_PLANE, RTT_BND_NONE, RTT_SHP_CYL_EDGE)) return OP_CYL_SIDE;
  return OP_GENERIC;
}

Ops make_ops(const RttSurface* host, int n_surf) {
  Ops ops = {};
  if (host)
    for (int s = 0; s < n_surf; ++s) ops.w[s >> 2] |= (uint32_t)classify(host[s]) << ((s & 3) * 8);
  return ops;
}

int g_lean_bounce_loop = 1;   // bounce loops of programs without a generic record run the instantiation without that visit

// Whether no record of the program needs the generic interpreter (every one matches a specialised visit): the bounce
// loops then run an instantiation that leaves it out, which takes the kernels below the register cap of four
// wavefronts per SIMD with less (or nothing) spilled.
bool no_generic_visit(const Ops& ops, const RttSurface* host, int n_surf) {
  if (!host || !g_lean_bounce_loop) return false;
  for (int s = 0; s < n_surf; ++s)
    if (((ops.w[s >> 2] >> ((s & 3) * 8)) & 0xff) == OP_GENERIC) return false;
  return true;
}

// ---- bounding spheres for the culling nearest-hit search -----------------------------------------------
// sagitta of a conic at height r, limited to the reach of the half-surface bound |z c| < 1
float conic_sag(float c, float c1k, float r) {
  if (c == 0.0f) return 0.0f;
  const float term = 1.0f - c * c1k * r * r;
  const float z = c * r * r / (1.0f + sqrtf(term > 0.0f ? term : 0.0f));
  const float reach = 1.000001f / fabsf(c);
  return fabsf(z) < reach ? z : (z < 0.0f ? -reach : reach);
}

bool is_identity(const float* R) {
  return R[0] == 1.0f && R[4] == 1.0f && R[8] == 1.0f && R[1] == 0.0f && R[2] == 0.0f && R[3] == 0.0f && R[5] == 0.0f &&
         R[6] == 0.0f && R[7] == 0.0f;
}

// Sphere around everything a ray can hit on the records [first, first + count) of one element, in global
// coordinates; returns false when the element has no finite extent this routine can derive (unbounded or
// inverted surfaces, general polyhedra, surfaces posed off the element axis).
bool element_sphere(const RttSurface* rec, int count, float c_out[3], float* R_out) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};  // box in the local frame
  auto grow = [&](float x0, float x1, float y0, float y1, float z0, float z1) {
    const float a[3] = {fminf(x0, x1), fminf(y0, y1), fminf(z0, z1)}, b[3] = {fmaxf(x0, x1), fmaxf(y0, y1), fmaxf(z0, z1)};
    for (int k = 0; k < 3; ++k) {
      lo[k] = fminf(lo[k], a[k]);
      hi[k] = fmaxf(hi[k], b[k]);
    }
  };
  const RttSurface& S0 = rec[0];
  const bool shaped = (S0.flags & RTT_F_HAS_SHAPE) != 0;
  const float* R = shaped ? S0.Rs : S0.Rj;  // local frame of the box -> global: g = t + R local
  const float* t = shaped ? S0.ts : S0.tj;
  if (!shaped && count != 1) return false;
  for (int q = 0; q < count; ++q) {
    const RttSurface& S = rec[q];
    if (S.flags & RTT_F_INVERT) return false;
    if (shaped) {
      if (!(S.flags & RTT_F_HAS_SHAPE)) return false;
      for (int k = 0; k < 9; ++k)
        if (S.Rs[k] != S0.Rs[k]) return false;
      for (int k = 0; k < 3; ++k)
        if (S.ts[k] != S0.ts[k]) return false;
      const float c = S.geom[0], c1k = S.geom[2];
      switch (S.shape_kind) {
        case RTT_SHP_SPHERIC_FACE: {
          if (S.surf_kind != RTT_SURF_QUADRIC || !is_identity(S.Rj) || S.tj[0] != 0.0f || S.tj[1] != 0.0f) return false;
          const float r = sqrtf(S.shp[0]);
          grow(-r, r, -r, r, S.tj[2], S.tj[2] + conic_sag(c, c1k, r));
          break;
        }
        case RTT_SHP_SPHERIC_EDGE: {
          if (S.surf_kind != RTT_SURF_CYLINDER || !is_identity(S.Rj) || S.tj[0] != 0.0f || S.tj[1] != 0.0f) return false;
          const float r = S.geom[0];
          grow(-r, r, -r, r, S.shp[0], S.shp[1]);
          break;
        }
        case RTT_SHP_CYL_FACE: {
          if (S.surf_kind != RTT_SURF_QUADRIC_ZY || !is_identity(S.Rj) || S.tj[0] != 0.0f || S.tj[1] != 0.0f) return false;
          const float ym = fmaxf(fabsf(S.shp[2]), fabsf(S.shp[3]));
          grow(S.shp[1], S.shp[0], S.shp[3], S.shp[2], S.tj[2], S.tj[2] + conic_sag(c, c1k, ym));
          break;
        }
        case RTT_SHP_CYL_EDGE: {
          // side planes: inside the aperture box and between the two sag curves, whose parameters the record carries
          const float ym = fmaxf(fabsf(S.shp[2]), fabsf(S.shp[3]));
          const float z0 = S.shp[5], z1 = S.shp[7];
          grow(S.shp[1], S.shp[0], S.shp[3], S.shp[2], fminf(z0, z0 + conic_sag(S.shp[4], S.shp[4], ym)),
               fmaxf(z1, z1 + conic_sag(S.shp[6], S.shp[6], ym)));
          break;
        }
        default:
          return false;
      }
    } else {
      const float c = S.geom[0], c1k = S.geom[2];
      if (S.surf_kind == RTT_SURF_PLANE && S.bound_kind == RTT_BND_DISK) {
        const float r = sqrtf(S.bnd[0]);
        grow(-r, r, -r, r, 0.0f, 0.0f);
      } else if (S.surf_kind == RTT_SURF_PLANE && S.bound_kind == RTT_BND_RECT) {
        grow(-S.bnd[0], S.bnd[0], -S.bnd[1], S.bnd[1], 0.0f, 0.0f);
      } else if (S.surf_kind == RTT_SURF_PLANE && S.bound_kind == RTT_BND_ELLIPSE) {
        const float r = fmaxf(fabsf(S.bnd[2]), fabsf(S.bnd[3]));
        grow(-r, r, -r, r, 0.0f, 0.0f);
      } else if (S.surf_kind == RTT_SURF_QUADRIC && S.bound_kind == RTT_BND_HALF_DISK) {
        const float r = sqrtf(S.bnd[0]);
        grow(-r, r, -r, r, 0.0f, conic_sag(c, c1k, r));
      } else {
        return false;
      }
    }
  }
  float cl[3], half2 = 0.0f;
  for (int k = 0; k < 3; ++k) {
    if (!(lo[k] <= hi[k]) || !std::isfinite(lo[k]) || !std::isfinite(hi[k])) return false;
    cl[k] = 0.5f * (lo[k] + hi[k]);
    const float h = 0.5f * (hi[k] - lo[k]);
    half2 += h * h;
  }
  for (int i = 0; i < 3; ++i) c_out[i] = t[i] + R[3 * i] * cl[0] + R[3 * i + 1] * cl[1] + R[3 * i + 2] * cl[2];
  const float Rb = sqrtf(half2) * 1.001f + 1e-3f;  // rounding of the records and of this estimate
  if (!std::isfinite(Rb) || !std::isfinite(c_out[0]) || !std::isfinite(c_out[1]) || !std::isfinite(c_out[2])) return false;
  *R_out = Rb;
  return true;
}

// Elements are the runs of consecutive records with one element_id.
Cull make_cull(const RttSurface* host, int n_surf) {
  Cull cull = {};
  if (!host || g_cull == 0) return cull;
  int n = 0;
  for (int s = 0; s < n_surf;) {
    int e = s + 1;
    while (e < n_surf && host[e].element_id == host[s].element_id) ++e;
    if (n == kMaxCull) {
      cull.n = 0;
      return cull;
    }
    cull.first[n] = (uint8_t)s;
    cull.count[n] = (uint8_t)(e - s);
    if (!element_sphere(host + s, e - s, cull.c[n], &cull.R[n])) cull.R[n] = -1.0f;
    ++n;
    s = e;
  }
  cull.n = n;
  return cull;
}

int check_program(const RttSurface* host, int n_surf, bool* any_sensor, int* n_slots, bool with_siblings = true) {
  if (n_surf < 1 || n_surf > RTT_MAX_SURFACES) return RTT_E_SIZE;
  if (any_sensor) *any_sensor = false;
  if (n_slots) *n_slots = 0;
  if (!host) return RTT_OK;  // caller opted out of validation
  for (int s = 0; s < n_surf; ++s) {
    const RttSurface& S = host[s];
    if (S.surf_kind < RTT_SURF_PLANE || S.surf_kind > RTT_SURF_CONE) return RTT_E_KIND;
    if (S.bound_kind < RTT_BND_NONE || S.bound_kind > RTT_BND_ALWAYS) return RTT_E_KIND;
    if (S.shape_kind < RTT_SHP_NONE || S.shape_kind > RTT_SHP_POLYHEDRON) return RTT_E_KIND;
    if (S.shape_kind == RTT_SHP_POLYHEDRON && with_siblings) {
      // the shape bound reads the sibling records first .. first + count - 1: they must exist in this program
      const int count = (int)S.shp[0], first = s - S.surface_id;
      if (count < 1 || S.surface_id < 0 || S.surface_id >= count || first < 0 || first + count > n_surf) return RTT_E_SIZE;
      for (int q = 0; q < count; ++q)
        if (host[first + q].element_id != S.element_id || host[first + q].surface_id != q) return RTT_E_KIND;
    }
    if (S.phys_kind < RTT_PHYS_TRANSMIT || S.phys_kind > RTT_PHYS_FRESNEL) return RTT_E_KIND;
    // a shape bound lives in the element frame; the two-level transform without a shape bound is allowed (one surface
    // of an element visited the way Shape.forward does, geom/shape.py:61-87)
    if (S.shape_kind != RTT_SHP_NONE && !(S.flags & RTT_F_HAS_SHAPE)) return RTT_E_KIND;
    if ((S.flags & RTT_F_SENSOR) && any_sensor) {  // slots only matter to entry points that record
      if (S.sensor_slot < 0 || S.sensor_slot >= RTT_MAX_SURFACES) return RTT_E_SIZE;
      if (any_sensor) *any_sensor = true;
      if (n_slots && S.sensor_slot + 1 > *n_slots) *n_slots = S.sensor_slot + 1;
    }
  }
  return RTT_OK;
}

bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) != 0; }

int launch_status() { return hipGetLastError() == hipSuccess ? RTT_OK : RTT_E_LAUNCH; }

// Bounces per stage of the staged bounce loop; RTT_STAGE_BOUNCES="5,2,2,3" sets another schedule for experiments (what
// is left after the listed stages goes into one last stage).  Read once, when the first staged call is made.
struct StageSchedule {
  int bounces[48];
  int stages;
};
const StageSchedule& stage_schedule() {
  static const StageSchedule schedule = [] {
    StageSchedule s = {{7, 2, 2, 3, 5, 8, 16, 1 << 30}, 8};
    if (const char* e = getenv("RTT_STAGE_BOUNCES")) {
      int k = 0;
      for (const char* q = e; *q && k < 46;) {
        const int v = atoi(q);
        if (v > 0) s.bounces[k++] = v;
        while (*q && *q != ',') ++q;
        if (*q == ',') ++q;
      }
      s.bounces[k++] = 1 << 30;
      s.stages = k;
    }
    return s;
  }();
  return schedule;
}

// The common direction of a source whose local direction is +z for every ray, computed as a lane would compute it
// (`volatile` keeps every product and sum individually rounded on the host, whatever contraction the build allows).
PosedAxis posed_axis(const RttSource& src) {
  PosedAxis a = {};
  if (src.kind == RTT_SRC_PENCIL || src.kind == RTT_SRC_DISK || src.kind == RTT_SRC_LINE || src.kind == RTT_SRC_GAUSS) {
    pose_direction(src.R, 0.0f, 0.0f, 1.0f, a.d, [](float x) {
      volatile float v = x;
      return (float)v;
    });
    a.known = 1;
  }
  return a;
}

}  // namespace

extern "C" {

int rtt_trace_seq_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                      float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                      const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, float* sens_hit,
                      float* sens_w, uint8_t* sens_mask, int64_t sens_ld, void* stream) {
  if (!pos_in || !dir_in || !inten_in || !pos_out || !dir_out || !inten_out || !prog_dev) return RTT_E_NULL;
  if (n < 0 || ld_in < n || ld_out < n) return RTT_E_SIZE;
  if (misaligned(pos_in) || misaligned(dir_in) || misaligned(inten_in) || misaligned(pos_out) ||
      misaligned(dir_out) || misaligned(inten_out) || misaligned(sens_hit) || misaligned(sens_w))
    return RTT_E_ALIGN;
  bool any_sensor;
  int rc = check_program(prog_host, n_surf, &any_sensor, nullptr);
  if (rc != RTT_OK) return rc;
  if (!prog_host) any_sensor = sens_hit != nullptr;  // unvalidated program: trust the buffers
  if (any_sensor && (!sens_hit || !sens_w || !sens_mask)) return RTT_E_NULL;
  if (any_sensor && sens_ld < n) return RTT_E_SIZE;
  if (n == 0) return RTT_OK;
  RayIO io = {pos_in, dir_in, inten_in, ld_in, pos_out, dir_out, inten_out, ld_out, n};
  SensorIO se = {sens_hit, sens_w, sens_mask, sens_ld};
  const Ops ops = make_ops(prog_host, n_surf);
  const hipStream_t st = (hipStream_t)stream;
  // two adjacent rays per lane; 8-byte accesses when every plane is 8-byte aligned
  auto aligned8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
  const int vec = (ld_in % 2 == 0) && (ld_out % 2 == 0) && aligned8(pos_in) && aligned8(dir_in) &&
                  aligned8(inten_in) && aligned8(pos_out) && aligned8(dir_out) && aligned8(inten_out);
  if (g_rays_per_lane == 1)
    hipLaunchKernelGGL((k_trace_seq<float, 1, false>), dim3(grid_for(n, 1)), dim3(kBlock), 0, st, io, prog_dev, n_surf, ops, se, 0);
  else if (g_rays_per_lane == 4)
    hipLaunchKernelGGL((k_trace_seq<f2, 2, false>), dim3(grid_for(n, 4)), dim3(kBlock), 0, st, io, prog_dev, n_surf, ops, se, vec);
  else if (g_program_in_lds)
    hipLaunchKernelGGL((k_trace_seq<f2, 1, true>), dim3(grid_for(n, 2)), dim3(kBlock), 0, st, io, prog_dev, n_surf, ops, se, vec);
  else
    hipLaunchKernelGGL((k_trace_seq<f2, 1, false>), dim3(grid_for(n, 2)), dim3(kBlock), 0, st, io, prog_dev, n_surf, ops, se, vec);
  return launch_status();
}

int rtt_sample_trace_seq_f32(const RttSource* src, float* pos_out, float* dir_out, float* inten_out, int64_t ld_out,
                             int8_t* id_out, float* wavelength_out, int64_t n, const RttSurface* prog_dev,
                             const RttSurface* prog_host, int n_surf, float* sens_hit, float* sens_w, uint8_t* sens_mask,
                             int64_t sens_ld, void* stream) {
  if (!src || !pos_out || !dir_out || !inten_out || !prog_dev) return RTT_E_NULL;
  if (n < 0 || ld_out < n) return RTT_E_SIZE;
  if (src->kind < RTT_SRC_PENCIL || src->kind > RTT_SRC_RAYLEIGH) return RTT_E_KIND;  // two uniforms per ray only
  if (src->offset & 1u) return RTT_E_ALIGN;  // a lane's pair of rays is one Philox block
  if (misaligned(pos_out) || misaligned(dir_out) || misaligned(inten_out) || misaligned(wavelength_out) ||
      misaligned(sens_hit) || misaligned(sens_w))
    return RTT_E_ALIGN;
  bool any_sensor;
  int rc = check_program(prog_host, n_surf, &any_sensor, nullptr);
  if (rc != RTT_OK) return rc;
  if (!prog_host) any_sensor = sens_hit != nullptr;
  if (any_sensor && (!sens_hit || !sens_w || !sens_mask)) return RTT_E_NULL;
  if (any_sensor && sens_ld < n) return RTT_E_SIZE;
  if (n == 0) return RTT_OK;
  RayIO io = {pos_out, dir_out, inten_out, ld_out, pos_out, dir_out, inten_out, ld_out, n};  // nothing is read
  SensorIO se = {sens_hit, sens_w, sens_mask, sens_ld};
  GenIO gen = {*src, posed_axis(*src), id_out, wavelength_out};
  const Ops ops = make_ops(prog_host, n_surf);
  auto aligned8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
  const int vec = (ld_out % 2 == 0) && aligned8(pos_out) && aligned8(dir_out) && aligned8(inten_out) &&
                  aligned8(wavelength_out) && (reinterpret_cast<uintptr_t>(id_out) & 1u) == 0;
  hipLaunchKernelGGL((k_trace_seq<f2, 1, false, false, true>), dim3(grid_for(n, 2)), dim3(kBlock), 0, (hipStream_t)stream, io,
                     prog_dev, n_surf, ops, se, vec, (double*)nullptr, gen);
  return launch_status();
}

int64_t rtt_trace_seq_moments_workspace_bytes(int64_t n) {
  return n < 0 ? 0 : (int64_t)sizeof(double) * 3 * kMomBins;
}

int rtt_trace_seq_moments_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                              float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                              const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, float* sens_hit,
                              float* sens_w, uint8_t* sens_mask, int64_t sens_ld, int mom_slot, double* mom, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  const bool no_rays_out = !pos_out && !dir_out && !inten_out;  // all three null: the traced rays are not wanted
  if (!pos_in || !dir_in || !inten_in || (!no_rays_out && (!pos_out || !dir_out || !inten_out)) || !prog_dev || !prog_host ||
      !sens_hit || !sens_w || !sens_mask || !mom || !workspace)
    return RTT_E_NULL;
  if (no_rays_out) ld_out = ld_in;
  if (n < 0 || ld_in < n || ld_out < n || sens_ld < n || mom_slot < 0 || mom_slot >= RTT_MAX_SURFACES) return RTT_E_SIZE;
  if (misaligned(pos_in) || misaligned(dir_in) || misaligned(inten_in) || misaligned(pos_out) || misaligned(dir_out) ||
      misaligned(inten_out) || misaligned(sens_hit) || misaligned(sens_w) || (reinterpret_cast<uintptr_t>(mom) & 7u) ||
      (reinterpret_cast<uintptr_t>(workspace) & 7u))
    return RTT_E_ALIGN;
  bool any_sensor;
  int n_slots = 0;
  int rc = check_program(prog_host, n_surf, &any_sensor, &n_slots);
  if (rc != RTT_OK) return rc;
  if (mom_slot >= n_slots) return RTT_E_SIZE;
  const hipStream_t st = (hipStream_t)stream;
  if (n == 0) return hipMemsetAsync(mom, 0, 3 * sizeof(double), st) == hipSuccess ? RTT_OK : RTT_E_LAUNCH;
  const int blocks = grid_for(n, 2);
  if (workspace_bytes < (int64_t)sizeof(double) * 3 * kMomBins) return RTT_E_SIZE;
  if (hipMemsetAsync(workspace, 0, sizeof(double) * 3 * kMomBins, st) != hipSuccess) return RTT_E_LAUNCH;
  RayIO io = {pos_in, dir_in, inten_in, ld_in, pos_out, dir_out, inten_out, ld_out, n};
  SensorIO se = {sens_hit, sens_w, sens_mask, sens_ld};
  se.mom_slot = mom_slot;
  const Ops ops = make_ops(prog_host, n_surf);
  auto aligned8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
  const int vec = (ld_in % 2 == 0) && (ld_out % 2 == 0) && aligned8(pos_in) && aligned8(dir_in) &&
                  aligned8(inten_in) && aligned8(pos_out) && aligned8(dir_out) && aligned8(inten_out);
  hipLaunchKernelGGL((k_trace_seq<f2, 1, false, true>), dim3(blocks), dim3(kBlock), 0, st, io, prog_dev, n_surf, ops, se, vec,
                     (double*)workspace);
  if (launch_status() != RTT_OK) return RTT_E_LAUNCH;
  hipLaunchKernelGGL(k_sum_partials, dim3(3), dim3(kBlock), 0, st, (const double*)workspace, kMomBins, (int64_t)1, (int64_t)kMomBins,
                     mom);
  return launch_status();
}

}  // extern "C"

namespace {

// every record has both rotations exactly the identity (and a host copy exists to tell)
bool all_rotations_identity(const RttSurface* prog_host, int n_surf) {
  if (!prog_host) return false;
  const int all_id = RTT_F_SHAPE_ROT_ID | RTT_F_SURF_ROT_ID;
  for (int s = 0; s < n_surf; ++s)
    if ((prog_host[s].flags & all_id) != all_id) return false;
  return true;
}

template <bool SPOT, class C>
void launch_tangent_cfg(RayIO io, const RttSurface* prog_dev, int n_surf, const TangentIO& tio, int n_theta, hipStream_t st) {
  const dim3 grid(grid_for_reduction(io.n)), block(kBlock);
  switch (n_theta) {
    case 2: hipLaunchKernelGGL((k_trace_seq_tangent<2, SPOT, false, C>), grid, block, 0, st, io, prog_dev, n_surf, tio); break;
    case 4: hipLaunchKernelGGL((k_trace_seq_tangent<4, SPOT, false, C>), grid, block, 0, st, io, prog_dev, n_surf, tio); break;
    case 6: hipLaunchKernelGGL((k_trace_seq_tangent<6, SPOT, false, C>), grid, block, 0, st, io, prog_dev, n_surf, tio); break;
    default: hipLaunchKernelGGL((k_trace_seq_tangent<8, SPOT, false, C>), grid, block, 0, st, io, prog_dev, n_surf, tio); break;
  }
}

template <bool SPOT>
int launch_tangent(RayIO io, const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, const TangentIO& tio,
                   int n_theta, hipStream_t st) {
  if (all_rotations_identity(prog_host, n_surf))
    launch_tangent_cfg<SPOT, CGenericIdent>(io, prog_dev, n_surf, tio, n_theta, st);
  else
    launch_tangent_cfg<SPOT, CGeneric>(io, prog_dev, n_surf, tio, n_theta, st);
  return launch_status();
}

int check_tangent_args(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, int64_t n,
                       const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, const float* tan_dev,
                       int n_theta, const double* out) {
  if (!pos_in || !dir_in || !inten_in || !prog_dev || !tan_dev || !out) return RTT_E_NULL;
  if (n < 0 || ld_in < n || n_theta < 2 || n_theta > 8 || (n_theta & 1)) return RTT_E_SIZE;
  if (misaligned(pos_in) || misaligned(dir_in) || misaligned(inten_in) || misaligned(tan_dev) ||
      (reinterpret_cast<uintptr_t>(out) & 7u))
    return RTT_E_ALIGN;
  bool any_sensor;
  return check_program(prog_host, n_surf, &any_sensor, nullptr);
}

}  // namespace

extern "C" {

int rtt_trace_seq_tangent_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, int64_t n,
                              const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                              const float* tan_dev, int n_theta, const float* adj_pos, const float* adj_dir,
                              int64_t ld_adj, const float* adj_sens, int64_t sens_ld, double* dtheta, void* stream) {
  int rc = check_tangent_args(pos_in, dir_in, inten_in, ld_in, n, prog_dev, prog_host, n_surf, tan_dev, n_theta, dtheta);
  if (rc != RTT_OK) return rc;
  if ((adj_pos || adj_dir) && ld_adj < n) return RTT_E_SIZE;
  if (adj_sens && sens_ld < n) return RTT_E_SIZE;
  if (misaligned(adj_pos) || misaligned(adj_dir) || misaligned(adj_sens)) return RTT_E_ALIGN;
  if (n == 0) return RTT_OK;
  RayIO io = {pos_in, dir_in, inten_in, ld_in, nullptr, nullptr, nullptr, 0, n};
  TangentIO tio = {tan_dev, adj_pos, adj_dir, ld_adj, adj_sens, sens_ld, dtheta, -1, 0.0f, 0.0f, 0.0f, nullptr, 0};
  return launch_tangent<false>(io, prog_dev, prog_host, n_surf, tio, n_theta, (hipStream_t)stream);
}

int rtt_trace_seq_ray_adjoint_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, int64_t n,
                                  const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                                  const float* zero_tan_dev, const float* adj_pos, const float* adj_dir, int64_t ld_adj,
                                  const float* adj_sens, int64_t sens_ld, float* ray_adj, int64_t ld_ray, void* stream) {
  if (!pos_in || !dir_in || !inten_in || !prog_dev || !zero_tan_dev || !ray_adj) return RTT_E_NULL;
  if (n < 0 || ld_in < n || ld_ray < n) return RTT_E_SIZE;
  if ((adj_pos || adj_dir) && ld_adj < n) return RTT_E_SIZE;
  if (adj_sens && sens_ld < n) return RTT_E_SIZE;
  if (misaligned(pos_in) || misaligned(dir_in) || misaligned(inten_in) || misaligned(zero_tan_dev) || misaligned(adj_pos) ||
      misaligned(adj_dir) || misaligned(adj_sens) || misaligned(ray_adj))
    return RTT_E_ALIGN;
  bool any_sensor;
  int rc = check_program(prog_host, n_surf, &any_sensor, nullptr);
  if (rc != RTT_OK) return rc;
  if (n == 0) return RTT_OK;
  RayIO io = {pos_in, dir_in, inten_in, ld_in, nullptr, nullptr, nullptr, 0, n};
  TangentIO tio = {zero_tan_dev, adj_pos, adj_dir, ld_adj, adj_sens, sens_ld, nullptr, -1, 0.0f, 0.0f, 0.0f, ray_adj, ld_ray};
  hipLaunchKernelGGL((k_trace_seq_tangent<6, false, true>), dim3(grid_for_reduction(n)), dim3(kBlock), 0, (hipStream_t)stream, io,
                     prog_dev, n_surf, tio);
  return launch_status();
}

int rtt_trace_seq_spot_grad_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, int64_t n,
                                const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                                const float* tan_dev, int n_theta, int sensor_slot, double cx, double cy, double wsum,
                                double* out, void* stream) {
  int rc = check_tangent_args(pos_in, dir_in, inten_in, ld_in, n, prog_dev, prog_host, n_surf, tan_dev, n_theta, out);
  if (rc != RTT_OK) return rc;
  if (sensor_slot < 0 || sensor_slot >= RTT_MAX_SURFACES || !(wsum > 0.0)) return RTT_E_SIZE;
  if (n == 0) return RTT_OK;
  RayIO io = {pos_in, dir_in, inten_in, ld_in, nullptr, nullptr, nullptr, 0, n};
  TangentIO tio = {tan_dev, nullptr, nullptr, 0, nullptr, 0, out, sensor_slot, (float)cx, (float)cy, (float)(1.0 / wsum), nullptr, 0};
  return launch_tangent<true>(io, prog_dev, prog_host, n_surf, tio, n_theta, (hipStream_t)stream);
}

}  // extern "C"

namespace {

constexpr int kAdjMaxBlocks = 2048;          // grid cap of the adjoint kernel = rows of its block-sum workspace
constexpr int kAdjMaxAcc = 128;              // accumulators per launch
constexpr int kLdsBytes = 160 * 1024;        // LDS of a gfx950 compute unit
int g_adjoint_lane = 0;                      // 0: choose by LDS need, 1: one ray per lane, 2: two rays per lane

int64_t adjoint_lds_bytes(int n_surf, int n_acc, int rays_per_lane, bool compact = false) {
  const int64_t per_ray = compact ? kAdjStart + (int64_t)n_surf * kAdjStateCompact : (int64_t)n_surf * kAdjState;
  return (int64_t)kBlock * 4 * ((int64_t)n_acc + per_ray * rays_per_lane);
}

constexpr int kMaxDevices = 64;
int g_adjoint_state = 0;                     // 0: compact states where they buy a third wavefront per SIMD, 1: full states always

int g_adjoint_lens_set = 1;                  // reverse sweep: lens-only programs run the instantiation without the other visits
int g_late_records = 0;                      // staged bounce loop: rays that rest after the first stage write the planes themselves (0) or leave records (1)
int g_adjoint_rounds = 3;                    // rounds of resident blocks in the grid of the reverse sweep (capped at 2048 blocks)

// Compute units of the current device (256 on an MI355X), asked once per device.
int compute_units() {
  static std::atomic<int> known[kMaxDevices];
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
  int n = known[dev].load(std::memory_order_acquire);
  if (n > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  known[dev].store(n, std::memory_order_release);
  return n;
}

// Dynamic LDS above 64 KB has to be granted per kernel and per device (each device holds its own copy of the code
// object): done once for the device that is current, remembered per device and instantiation.
template <class T, int BS, bool COMPACT, bool TAPE, int SET = 0>
int grant_adjoint_lds() {
  static std::atomic<bool> granted[kMaxDevices];
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return RTT_E_LAUNCH;
  if (granted[dev].load(std::memory_order_acquire)) return RTT_OK;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_seq_adjoint<T, BS, COMPACT, TAPE, SET>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          kLdsBytes) != hipSuccess)
    return RTT_E_LAUNCH;
  granted[dev].store(true, std::memory_order_release);
  return RTT_OK;
}

template <class T, int BS, bool COMPACT = false, bool TAPE = false, int SET = 0>
int launch_adjoint(const RayIO& io, const RttSurface* prog_dev, int n_surf, const Ops& ops, const AccMap& am, const AdjointIO& a,
                   int vec, int64_t lds, int blocks, hipStream_t st) {
  const int rc = grant_adjoint_lds<T, BS, COMPACT, TAPE, SET>();
  if (blocks == 0) return RTT_OK;  // a call without rays only prepares the current device, if there is one
  if (rc != RTT_OK) return rc;
  hipLaunchKernelGGL((k_trace_seq_adjoint<T, BS, COMPACT, TAPE, SET>), dim3(blocks), dim3(BS), (size_t)lds, st, io, prog_dev, n_surf, ops, am,
                     a, vec);
  return launch_status();
}

// A tape in memory: 4 n_surf planes of ld floats (ld = n rounded up to an even number, so that every plane of a 256-byte
// aligned tape is 8-byte aligned), then one 64-bit surface mask per group of kTapeGroup rays.
struct TapeLayout {
  int64_t ld, groups, bytes;
};
TapeLayout tape_layout(int64_t n, int n_surf) {
  TapeLayout t;
  t.ld = (n + 1) & ~(int64_t)1;
  t.groups = (n + kTapeGroup - 1) / kTapeGroup;
  t.bytes = (int64_t)16 * n_surf * t.ld + 8 * t.groups;
  return t;
}

// Shared body of the adjoint entry points: `partials` is the block-sum scratch ([kAdjMaxBlocks][n_acc] doubles).  With a
// tape (and spot_w in spot mode) there is no forward sweep and dir_in / inten_in are not read.
int trace_seq_adjoint(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, int64_t n,
                      const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, const uint32_t* slot_mask,
                      const float* adj_pos, const float* adj_dir, int64_t ld_adj, const float* adj_sens, int64_t sens_ld,
                      int spot_slot, int spot_centroid, double cx, double cy, double wsum, const double* spot_mom, float* ray_adj,
                      int64_t ld_ray, double* slot_adj, double* partials, int64_t partial_rows, hipStream_t st,
                      const void* tape = nullptr, int64_t tape_bytes = 0, const float* spot_w = nullptr) {
  if (!pos_in || (!tape && (!dir_in || !inten_in)) || !prog_dev || !prog_host) return RTT_E_NULL;
  if (n < 0 || ld_in < n) return RTT_E_SIZE;
  if ((adj_pos || adj_dir) && ld_adj < n) return RTT_E_SIZE;
  if (adj_sens && sens_ld < n) return RTT_E_SIZE;
  if (ray_adj && ld_ray < n) return RTT_E_SIZE;
  if (misaligned(pos_in) || misaligned(dir_in) || misaligned(inten_in) || misaligned(adj_pos) || misaligned(adj_dir) ||
      misaligned(adj_sens) || misaligned(ray_adj) || (reinterpret_cast<uintptr_t>(slot_adj) & 7u) ||
      (reinterpret_cast<uintptr_t>(partials) & 7u) || (reinterpret_cast<uintptr_t>(spot_mom) & 7u))
    return RTT_E_ALIGN;
  bool any_sensor;
  int rc = check_program(prog_host, n_surf, &any_sensor, nullptr);
  if (rc != RTT_OK) return rc;
  if (spot_slot >= RTT_MAX_SURFACES || (spot_slot >= 0 && !spot_mom && (!(wsum > 0.0) || spot_centroid))) return RTT_E_SIZE;
  AccMap am = {};
  int n_acc = 0;
  for (int s = 0; s < n_surf; ++s) {
    am.mask[s] = slot_mask ? slot_mask[s] : 0u;
    am.base[s] = (uint16_t)n_acc;
    n_acc += __builtin_popcount(am.mask[s]);
  }
  if (n_acc > kAdjMaxAcc) return RTT_E_SIZE;
  if (n_acc > 0 && (!slot_adj || !partials)) return RTT_E_NULL;
  // short programs with few slot sums keep everything in registers (two rays per lane unless one is forced); the
  // others keep the per-surface states and per-lane sums in LDS, which then sets rays per lane and occupancy
  // compact states for short programs when they buy a third wavefront per SIMD (unless tuned off)
  bool compact = g_adjoint_state == 0 && g_adjoint_lane != 1 && n_surf <= kAdjCompactSurfaces &&
                 adjoint_lds_bytes(n_surf, n_acc, 2, true) <= kLdsBytes / 3 && adjoint_lds_bytes(n_surf, n_acc, 2) > kLdsBytes / 3;
  TapeLayout tl = {};
  if (tape) {  // a tape is swept with compact states, two rays per lane, whatever the tuning says
    if (n_surf > kAdjCompactSurfaces || adjoint_lds_bytes(n_surf, n_acc, 2, true) > kLdsBytes) return RTT_E_SIZE;
    tl = tape_layout(n, n_surf);
    if (tape_bytes < tl.bytes) return RTT_E_SIZE;
    if ((reinterpret_cast<uintptr_t>(tape) & 255u) || misaligned(spot_w)) return RTT_E_ALIGN;
    if (spot_slot >= 0 && !spot_w) return RTT_E_NULL;
    compact = true;
  }
  int lanes = tape ? 2 : g_adjoint_lane;
  if (lanes == 0) lanes = compact || adjoint_lds_bytes(n_surf, n_acc, 2) <= kLdsBytes / 2 ? 2 : 1;
  if (adjoint_lds_bytes(n_surf, n_acc, lanes, compact) > kLdsBytes) return RTT_E_SIZE;
  const int64_t lds = adjoint_lds_bytes(n_surf, n_acc, lanes, compact);
  const int64_t tile = (int64_t)kBlock * lanes;
  int64_t blocks = (n + tile - 1) / tile;  // none for a call without rays, which still prepares the current device
  // About three rounds of resident blocks (registers allow 3 blocks per unit with compact states, else 2; LDS may allow
  // fewer), walked over the tiles by grid stride.  Measured at 32 Mi rays: one round of blocks that all start together
  // stays in lock step (every wavefront loads, then every wavefront computes) and is 2 % slower than three rounds.
  int64_t resident = (int64_t)compute_units() * g_adjoint_rounds * std::min<int64_t>(compact ? 3 : 2, kLdsBytes / std::max<int64_t>(lds, 1));
  if (resident > kAdjMaxBlocks) resident = kAdjMaxBlocks;
  if (blocks > resident) blocks = resident;
  if (n_acc > 0 && blocks > partial_rows) return RTT_E_SIZE;
  RayIO io = {pos_in, dir_in, inten_in, ld_in, nullptr, nullptr, nullptr, 0, n};
  AdjointIO a = {adj_pos, adj_dir, ld_adj, adj_sens, sens_ld, spot_slot, spot_centroid, (float)cx, (float)cy,
                 (spot_slot >= 0 && !spot_mom) ? (float)(1.0 / wsum) : 0.0f, spot_mom, ray_adj, ld_ray, partials, n_acc,
                 (const float*)tape, tl.ld, tape ? (const uint64_t*)((const char*)tape + (int64_t)16 * n_surf * tl.ld) : nullptr,
                 spot_w};
  const Ops ops = make_ops(prog_host, n_surf);
  auto aligned8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
  const int vec = (ld_in % 2 == 0) && aligned8(pos_in) && aligned8(dir_in) && aligned8(inten_in) && aligned8(adj_pos) &&
                  aligned8(adj_dir) && aligned8(adj_sens) && aligned8(spot_w);
  if (tape)
    rc = launch_adjoint<f2, kBlock, true, true>(io, prog_dev, n_surf, ops, am, a, vec, lds, (int)blocks, st);
  else {
    // a program of lens faces, lens edges and bounded planes only (lenses, stops, a sensor) runs the instantiation that
    // holds just these visits: a third of the code, fewer registers, nothing spilled
    bool lens_set = g_adjoint_lens_set != 0;
    for (int s = 0; s < n_surf && lens_set; ++s) {
      const int op = (ops.w[s >> 2] >> ((s & 3) * 8)) & 0xff;
      lens_set = op == OP_FACE_SNELL || op == OP_EDGE_BLOCK || op == OP_EDGE_SNELL || op == OP_DISK_PLANE;
    }
    const bool lean = g_adjoint_lens_set != 0 && no_generic_visit(ops, prog_host, n_surf);   // (no record for the interpreter)
    if (compact)
      rc = lens_set ? launch_adjoint<f2, kBlock, true, false, 1>(io, prog_dev, n_surf, ops, am, a, vec, lds, (int)blocks, st)
           : lean   ? launch_adjoint<f2, kBlock, true, false, 2>(io, prog_dev, n_surf, ops, am, a, vec, lds, (int)blocks, st)
                    : launch_adjoint<f2, kBlock, true>(io, prog_dev, n_surf, ops, am, a, vec, lds, (int)blocks, st);
    else if (lanes == 2)
      rc = lens_set ? launch_adjoint<f2, kBlock, false, false, 1>(io, prog_dev, n_surf, ops, am, a, vec, lds, (int)blocks, st)
           : lean   ? launch_adjoint<f2, kBlock, false, false, 2>(io, prog_dev, n_surf, ops, am, a, vec, lds, (int)blocks, st)
                    : launch_adjoint<f2, kBlock>(io, prog_dev, n_surf, ops, am, a, vec, lds, (int)blocks, st);
    else
      rc = launch_adjoint<float, kBlock>(io, prog_dev, n_surf, ops, am, a, 0, lds, (int)blocks, st);
  }
  if (rc != RTT_OK || n_acc == 0) return rc;
  if (n == 0) return hipMemsetAsync(slot_adj, 0, sizeof(double) * n_acc, st) == hipSuccess ? RTT_OK : RTT_E_LAUNCH;
  hipLaunchKernelGGL(k_sum_partials, dim3(n_acc), dim3(kBlock), 0, st, partials, (int)blocks, (int64_t)n_acc, (int64_t)1, slot_adj);
  return launch_status();
}

}  // namespace

// Block-sum scratch of rtt_trace_seq_adjoint_static_f32: one array per GPU the library runs on, held until the process
// ends.  In host code the name below is only a handle for hipGetSymbolAddress; kernels get the address it returns.
static __device__ __align__(256) double scratch[(size_t)kAdjMaxBlocks * kAdjMaxAcc];

namespace {

std::atomic<double*> g_scratch_ptr[kMaxDevices];

// Address of `scratch` on the current device, looked up once per device (each GPU holds its own copy at its own address).
double* scratch_on_current_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  double* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (p) return p;
  void* addr = nullptr;
  if (hipGetSymbolAddress(&addr, HIP_SYMBOL(scratch)) != hipSuccess || !addr) return nullptr;
  g_scratch_ptr[dev].store(static_cast<double*>(addr), std::memory_order_release);
  return static_cast<double*>(addr);
}

}  // namespace

extern "C" {

int rtt_trace_seq_adjoint_static_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, int64_t n,
                                     const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                                     const uint32_t* slot_mask, const float* adj_pos, const float* adj_dir, int64_t ld_adj,
                                     const float* adj_sens, int64_t sens_ld, int spot_slot, int spot_centroid, double cx, double cy,
                                     double wsum, const double* spot_mom, float* ray_adj, int64_t ld_ray, double* slot_adj,
                                     void* stream) {
  int n_acc = 0;
  if (slot_mask && n_surf >= 1 && n_surf <= RTT_MAX_SURFACES)
    for (int s = 0; s < n_surf; ++s) n_acc += __builtin_popcount(slot_mask[s]);
  if (n_acc > kAdjMaxAcc) return RTT_E_SIZE;  // more slot sums than the static scratch has columns for
  double* partials = scratch_on_current_device();
  if (!partials) return RTT_E_LAUNCH;
  return trace_seq_adjoint(pos_in, dir_in, inten_in, ld_in, n, prog_dev, prog_host, n_surf, slot_mask, adj_pos, adj_dir, ld_adj,
                           adj_sens, sens_ld, spot_slot, spot_centroid, cx, cy, wsum, spot_mom, ray_adj, ld_ray, slot_adj, partials,
                           kAdjMaxBlocks, (hipStream_t)stream);
}

int rtt_trace_seq_adjoint_max_slots(int n_surf) {
  if (n_surf < 1 || n_surf > RTT_MAX_SURFACES) return 0;
  const int64_t room = (kLdsBytes - adjoint_lds_bytes(n_surf, 0, 1)) / (kBlock * 4);
  return room < 0 ? 0 : (room > kAdjMaxAcc ? kAdjMaxAcc : (int)room);
}

int64_t rtt_trace_seq_adjoint_workspace_bytes(int n_slots) {
  return n_slots < 0 ? 0 : (int64_t)sizeof(double) * kAdjMaxBlocks * n_slots;
}

int rtt_trace_seq_adjoint_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, int64_t n,
                              const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, const uint32_t* slot_mask,
                              const float* adj_pos, const float* adj_dir, int64_t ld_adj, const float* adj_sens, int64_t sens_ld,
                              int spot_slot, int spot_centroid, double cx, double cy, double wsum, const double* spot_mom,
                              float* ray_adj, int64_t ld_ray, double* slot_adj, void* workspace, int64_t workspace_bytes,
                              void* stream) {
  int n_acc = 0;
  if (slot_mask && n_surf >= 1 && n_surf <= RTT_MAX_SURFACES)
    for (int s = 0; s < n_surf; ++s) n_acc += __builtin_popcount(slot_mask[s]);
  const int64_t rows = n_acc > 0 ? workspace_bytes / ((int64_t)sizeof(double) * n_acc) : 0;
  return trace_seq_adjoint(pos_in, dir_in, inten_in, ld_in, n, prog_dev, prog_host, n_surf, slot_mask, adj_pos, adj_dir, ld_adj,
                           adj_sens, sens_ld, spot_slot, spot_centroid, cx, cy, wsum, spot_mom, ray_adj, ld_ray, slot_adj,
                           (double*)workspace, rows, (hipStream_t)stream);
}

int64_t rtt_trace_seq_tape_bytes(int64_t n, int n_surf) {
  if (n < 0 || n_surf < 1 || n_surf > RTT_MAX_SURFACES) return 0;
  return tape_layout(n, n_surf).bytes;
}

int rtt_trace_seq_adjoint_tape_max_slots(int n_surf) {
  if (n_surf < 1 || n_surf > kAdjCompactSurfaces) return 0;
  const int64_t room = (kLdsBytes - adjoint_lds_bytes(n_surf, 0, 2, true)) / (kBlock * 4);
  return room < 0 ? 0 : (room > kAdjMaxAcc ? kAdjMaxAcc : (int)room);
}

int rtt_trace_seq_tape_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, float* pos_out,
                           float* dir_out, float* inten_out, int64_t ld_out, int64_t n, const RttSurface* prog_dev,
                           const RttSurface* prog_host, int n_surf, float* sens_hit, float* sens_w, uint8_t* sens_mask,
                           int64_t sens_ld, int mom_slot, double* mom, void* workspace, int64_t workspace_bytes, void* tape,
                           int64_t tape_bytes, void* stream) {
  const bool no_rays_out = !pos_out && !dir_out && !inten_out;
  if (!pos_in || !dir_in || !inten_in || (!no_rays_out && (!pos_out || !dir_out || !inten_out)) || !prog_dev || !prog_host || !tape)
    return RTT_E_NULL;
  if (no_rays_out) ld_out = ld_in;
  if (n < 0 || ld_in < n || ld_out < n) return RTT_E_SIZE;
  if (misaligned(pos_in) || misaligned(dir_in) || misaligned(inten_in) || misaligned(pos_out) || misaligned(dir_out) ||
      misaligned(inten_out) || misaligned(sens_hit) || misaligned(sens_w) || (reinterpret_cast<uintptr_t>(mom) & 7u) ||
      (reinterpret_cast<uintptr_t>(workspace) & 7u) || (reinterpret_cast<uintptr_t>(tape) & 255u))
    return RTT_E_ALIGN;
  bool any_sensor;
  int n_slots = 0;
  int rc = check_program(prog_host, n_surf, &any_sensor, &n_slots);
  if (rc != RTT_OK) return rc;
  if (any_sensor && (!sens_hit || !sens_w || !sens_mask)) return RTT_E_NULL;
  if (any_sensor && sens_ld < n) return RTT_E_SIZE;
  const bool moments = mom_slot >= 0;
  if (moments && (!mom || !workspace)) return RTT_E_NULL;
  if (moments && (mom_slot >= n_slots || workspace_bytes < (int64_t)sizeof(double) * 3 * kMomBins)) return RTT_E_SIZE;
  if (!moments && no_rays_out) return RTT_E_NULL;  // only the launch that sums moments can leave the rays unwritten
  const TapeLayout tl = tape_layout(n, n_surf);
  if (tape_bytes < tl.bytes) return RTT_E_SIZE;
  const hipStream_t st = (hipStream_t)stream;
  if (n == 0) return !moments || hipMemsetAsync(mom, 0, 3 * sizeof(double), st) == hipSuccess ? RTT_OK : RTT_E_LAUNCH;
  if (moments && hipMemsetAsync(workspace, 0, sizeof(double) * 3 * kMomBins, st) != hipSuccess) return RTT_E_LAUNCH;
  RayIO io = {pos_in, dir_in, inten_in, ld_in, pos_out, dir_out, inten_out, ld_out, n};
  SensorIO se = {sens_hit, sens_w, sens_mask, sens_ld};
  se.mom_slot = mom_slot;
  const TapeIO tp = {(float*)tape, tl.ld, (uint64_t*)((char*)tape + (int64_t)16 * n_surf * tl.ld)};
  const Ops ops = make_ops(prog_host, n_surf);
  auto aligned8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
  const int vec = (ld_in % 2 == 0) && (ld_out % 2 == 0) && aligned8(pos_in) && aligned8(dir_in) && aligned8(inten_in) &&
                  aligned8(pos_out) && aligned8(dir_out) && aligned8(inten_out);
  // (the grid is not capped: a block's wavefronts are the groups of its one tile)
  const dim3 grid((unsigned)((n + 2 * kBlock - 1) / (2 * kBlock)));
  if (moments) {
    hipLaunchKernelGGL((k_trace_seq<f2, 1, false, true, false, true>), grid, dim3(kBlock), 0, st, io, prog_dev, n_surf, ops, se, vec,
                       (double*)workspace, GenIO{}, tp);
    if (launch_status() != RTT_OK) return RTT_E_LAUNCH;
    hipLaunchKernelGGL(k_sum_partials, dim3(3), dim3(kBlock), 0, st, (const double*)workspace, kMomBins, (int64_t)1, (int64_t)kMomBins,
                       mom);
  } else {
    hipLaunchKernelGGL((k_trace_seq<f2, 1, false, false, false, true>), grid, dim3(kBlock), 0, st, io, prog_dev, n_surf, ops, se, vec,
                       (double*)nullptr, GenIO{}, tp);
  }
  return launch_status();
}

int rtt_trace_seq_adjoint_tape_f32(const float* pos_in, int64_t ld_in, int64_t n, const RttSurface* prog_dev,
                                   const RttSurface* prog_host, int n_surf, const uint32_t* slot_mask, const float* adj_pos,
                                   const float* adj_dir, int64_t ld_adj, const float* adj_sens, int64_t sens_ld, int spot_slot,
                                   int spot_centroid, double cx, double cy, double wsum, const double* spot_mom, const float* spot_w,
                                   float* ray_adj, int64_t ld_ray, double* slot_adj, void* workspace, int64_t workspace_bytes,
                                   const void* tape, int64_t tape_bytes, void* stream) {
  if (!tape) return RTT_E_NULL;
  int n_acc = 0;
  if (slot_mask && n_surf >= 1 && n_surf <= RTT_MAX_SURFACES)
    for (int s = 0; s < n_surf; ++s) n_acc += __builtin_popcount(slot_mask[s]);
  const int64_t rows = n_acc > 0 ? workspace_bytes / ((int64_t)sizeof(double) * n_acc) : 0;
  return trace_seq_adjoint(pos_in, nullptr, nullptr, ld_in, n, prog_dev, prog_host, n_surf, slot_mask, adj_pos, adj_dir, ld_adj,
                           adj_sens, sens_ld, spot_slot, spot_centroid, cx, cy, wsum, spot_mom, ray_adj, ld_ray, slot_adj,
                           (double*)workspace, rows, (hipStream_t)stream, tape, tape_bytes, spot_w);
}

int rtt_ray_cast_f32(const float* pos, const float* dir, int64_t ld, int64_t n, const RttSurface* prog_dev,
                     const RttSurface* prog_host, int n_surf, int32_t* winner, float* tmin, void* stream) {
  if (!pos || !dir || !prog_dev) return RTT_E_NULL;
  if (n < 0 || ld < n) return RTT_E_SIZE;
  if (misaligned(pos) || misaligned(dir) || misaligned(winner) || misaligned(tmin)) return RTT_E_ALIGN;
  int rc = check_program(prog_host, n_surf, nullptr, nullptr);
  if (rc != RTT_OK) return rc;
  if (n == 0) return RTT_OK;
  const int vec = (ld % 2 == 0) && ((reinterpret_cast<uintptr_t>(pos) | reinterpret_cast<uintptr_t>(dir)) & 7u) == 0;
  hipLaunchKernelGGL(k_ray_cast<f2>, dim3(grid_for(n, 2)), dim3(kBlock), 0, (hipStream_t)stream, pos, dir, ld, n,
                     prog_dev, n_surf, make_ops(prog_host, n_surf), make_cull(prog_host, n_surf), winner, tmin, vec);
  return launch_status();
}

int rtt_step_nearest_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                         float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                         const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, int32_t* winner,
                         float* sens_hit, float* sens_w, uint8_t* sens_mask, int64_t sens_ld, void* stream) {
  if (!pos_in || !dir_in || !inten_in || !pos_out || !dir_out || !inten_out || !prog_dev) return RTT_E_NULL;
  if (n < 0 || ld_in < n || ld_out < n) return RTT_E_SIZE;
  if (misaligned(pos_in) || misaligned(dir_in) || misaligned(inten_in) || misaligned(pos_out) ||
      misaligned(dir_out) || misaligned(inten_out) || misaligned(winner) || misaligned(sens_hit) ||
      misaligned(sens_w))
    return RTT_E_ALIGN;
  bool any_sensor;
  int n_slots = 0;
  int rc = check_program(prog_host, n_surf, &any_sensor, &n_slots);
  if (rc != RTT_OK) return rc;
  if (!prog_host) any_sensor = sens_hit != nullptr;
  if (any_sensor && (!sens_hit || !sens_w || !sens_mask)) return RTT_E_NULL;
  if (any_sensor && sens_ld < n) return RTT_E_SIZE;
  if (n == 0) return RTT_OK;
  RayIO io = {pos_in, dir_in, inten_in, ld_in, pos_out, dir_out, inten_out, ld_out, n};
  SensorIO se = {sens_hit, sens_w, sens_mask, sens_ld};
  // a bounce writes the mask entry only of rays that went to a sensor: every lane clears the entries of its rays first
  // (the kernel knows how many slots there are from the host copy of the program)
  se.zero_slots = prog_host ? n_slots : 0;
  auto aligned8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
  const int vec = (ld_in % 2 == 0) && (ld_out % 2 == 0) && aligned8(pos_in) && aligned8(dir_in) &&
                  aligned8(inten_in) && aligned8(pos_out) && aligned8(dir_out) && aligned8(inten_out) &&
                  (sens_ld % 2 == 0) && (reinterpret_cast<uintptr_t>(sens_mask) & 1u) == 0;
  const Ops ops = make_ops(prog_host, n_surf);
  if (no_generic_visit(ops, prog_host, n_surf))
    hipLaunchKernelGGL((k_simulate_nearest<f2, true, false, 1>), dim3(grid_for(n, 2)), dim3(kBlock), 0, (hipStream_t)stream, io,
                       prog_dev, n_surf, ops, make_cull(prog_host, n_surf), 1, winner, (int32_t*)nullptr, se, vec);
  else
    hipLaunchKernelGGL((k_simulate_nearest<f2, true>), dim3(grid_for(n, 2)), dim3(kBlock), 0, (hipStream_t)stream, io, prog_dev,
                       n_surf, ops, make_cull(prog_host, n_surf), 1, winner, (int32_t*)nullptr, se, vec);
  return launch_status();
}

int rtt_intersect_test_f32(const float* pos, const float* dir, int64_t ld, int64_t n, const RttSurface* prog_dev,
                           const RttSurface* prog_host, int n_surf, float* t_out, int64_t ld_t, void* stream) {
  if (!pos || !dir || !prog_dev || !t_out) return RTT_E_NULL;
  if (n < 0 || ld < n || ld_t < n) return RTT_E_SIZE;
  if (misaligned(pos) || misaligned(dir) || misaligned(t_out)) return RTT_E_ALIGN;
  int rc = check_program(prog_host, n_surf, nullptr, nullptr);
  if (rc != RTT_OK) return rc;
  if (n == 0) return RTT_OK;
  hipLaunchKernelGGL(k_intersect_test, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, pos, dir, ld, n,
                     prog_dev, n_surf, t_out, ld_t);
  return launch_status();
}

int rtt_surface_forward_f32(const float* pos, const float* dir, int64_t ld, int64_t n, const RttSurface* prog_dev,
                            const RttSurface* prog_host, int surf_index, float* t, float* hit_global,
                            float* normal_global, float* hit_local, float* new_dir, float* imult, int64_t ld_o,
                            void* stream) {
  if (!pos || !dir || !prog_dev) return RTT_E_NULL;
  if (n < 0 || ld < n || ld_o < n || surf_index < 0 || surf_index >= RTT_MAX_SURFACES) return RTT_E_SIZE;
  if (misaligned(pos) || misaligned(dir) || misaligned(t) || misaligned(hit_global) ||
      misaligned(normal_global) || misaligned(hit_local) || misaligned(new_dir) || misaligned(imult))
    return RTT_E_ALIGN;
  if (prog_host) {
    int rc = check_program(prog_host + surf_index, 1, nullptr, nullptr, false);  // forward applies no shape bound
    if (rc != RTT_OK) return rc;
  }
  if (n == 0) return RTT_OK;
  ForwardOut o = {t, hit_global, normal_global, hit_local, new_dir, imult, ld_o};
  hipLaunchKernelGGL(k_surface_forward, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, pos, dir, ld, n,
                     prog_dev, surf_index, o);
  return launch_status();
}

int rtt_apply_physics_f32(const float* hit_local, const float* dir, const float* normal, int64_t ld, int64_t n,
                          const RttSurface* prog_dev, const RttSurface* prog_host, float* new_dir, float* imult,
                          void* stream) {
  if (!hit_local || !dir || !normal || !prog_dev || !new_dir || !imult) return RTT_E_NULL;
  if (n < 0 || ld < n) return RTT_E_SIZE;
  if (misaligned(hit_local) || misaligned(dir) || misaligned(normal) || misaligned(new_dir) || misaligned(imult))
    return RTT_E_ALIGN;
  if (prog_host && (prog_host->phys_kind < RTT_PHYS_TRANSMIT || prog_host->phys_kind > RTT_PHYS_FRESNEL))
    return RTT_E_KIND;
  if (n == 0) return RTT_OK;
  hipLaunchKernelGGL(k_apply_physics, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, hit_local, dir,
                     normal, ld, n, prog_dev, new_dir, imult);
  return launch_status();
}

static int g_moments_blocks = 512;
static int g_moments_threads = 1024;
static int moments_grid(int64_t n, int threads = kBlock) {
  int64_t blocks = (n + threads - 1) / threads;
  if (blocks > g_moments_blocks) blocks = g_moments_blocks;
  return (int)(blocks < 1 ? 1 : blocks);
}

int rtt_spot_moments_f32(const float* x, const float* y, const float* w, const uint8_t* mask, int64_t n, int pass,
                         double cx, double cy, double wsum, double* out, void* stream) {
  if (!x || !y || !w || !out) return RTT_E_NULL;
  if (n < 0 || pass < 0 || pass > 4) return RTT_E_SIZE;
  if (misaligned(x) || misaligned(y) || misaligned(w) || (reinterpret_cast<uintptr_t>(out) & 7u)) return RTT_E_ALIGN;
  if (n == 0) return RTT_OK;
  auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  if (aligned16(x) && aligned16(y) && aligned16(w) && (reinterpret_cast<uintptr_t>(mask) & 3u) == 0)
    if (g_moments_threads == 1024)
      hipLaunchKernelGGL((k_spot_moments<true, 1024>), dim3(moments_grid((n + 3) / 4, 1024)), dim3(1024), 0, (hipStream_t)stream,
                         x, y, w, mask, n, pass, cx, cy, wsum, out);
    else
      hipLaunchKernelGGL(k_spot_moments<true>, dim3(moments_grid((n + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, x, y, w,
                         mask, n, pass, cx, cy, wsum, out);
  else
    hipLaunchKernelGGL(k_spot_moments<false>, dim3(grid_for_reduction(n)), dim3(kBlock), 0, (hipStream_t)stream, x, y, w, mask,
                       n, pass, cx, cy, wsum, out);
  return launch_status();
}

int rtt_trace_seq_history_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                              float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                              const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, float* sens_hit,
                              float* sens_w, uint8_t* sens_mask, int64_t sens_ld, float* hist_pos, int32_t* hist_any,
                              int64_t hist_ld, void* stream) {
  if (!pos_in || !dir_in || !inten_in || !pos_out || !dir_out || !inten_out || !prog_dev || !hist_pos || !hist_any)
    return RTT_E_NULL;
  if (n < 0 || ld_in < n || ld_out < n || hist_ld < n) return RTT_E_SIZE;
  if (misaligned(pos_in) || misaligned(dir_in) || misaligned(inten_in) || misaligned(pos_out) || misaligned(dir_out) ||
      misaligned(inten_out) || misaligned(sens_hit) || misaligned(sens_w) || misaligned(hist_pos) || misaligned(hist_any))
    return RTT_E_ALIGN;
  bool any_sensor;
  int rc = check_program(prog_host, n_surf, &any_sensor, nullptr);
  if (rc != RTT_OK) return rc;
  if (!prog_host) any_sensor = sens_hit != nullptr;
  if (any_sensor && (!sens_hit || !sens_w || !sens_mask)) return RTT_E_NULL;
  if (any_sensor && sens_ld < n) return RTT_E_SIZE;
  if (n == 0) return RTT_OK;
  RayIO io = {pos_in, dir_in, inten_in, ld_in, pos_out, dir_out, inten_out, ld_out, n};
  SensorIO se = {sens_hit, sens_w, sens_mask, sens_ld, 0};
  hipLaunchKernelGGL(k_trace_seq_history, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, io, prog_dev, n_surf, se,
                     hist_pos, hist_any, hist_ld);
  return launch_status();
}

int rtt_simulate_nearest_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                             float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                             const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, int max_bounces,
                             int32_t* bounces, float* hit_list, int32_t* hit_keys, int32_t* hit_count,
                             int64_t hit_capacity, int hit_parts, void* stream) {
  if (!pos_in || !dir_in || !inten_in || !pos_out || !dir_out || !inten_out || !prog_dev) return RTT_E_NULL;
  if (n < 0 || n > 0x7fffffff || ld_in < n || ld_out < n || max_bounces < 0 || hit_capacity < 0 ||
      hit_capacity > 0x7fffffff)
    return RTT_E_SIZE;
  if (hit_parts < 1 || hit_parts > 64 || (hit_parts & (hit_parts - 1))) return RTT_E_SIZE;
  if (misaligned(pos_in) || misaligned(dir_in) || misaligned(inten_in) || misaligned(pos_out) || misaligned(dir_out) ||
      misaligned(inten_out) || misaligned(bounces) || misaligned(hit_list) || misaligned(hit_keys) || misaligned(hit_count))
    return RTT_E_ALIGN;
  bool any_sensor;
  int rc = check_program(prog_host, n_surf, &any_sensor, nullptr);
  if (rc != RTT_OK) return rc;
  if (!prog_host) any_sensor = hit_count != nullptr;
  if (any_sensor && (!hit_count || (hit_capacity > 0 && (!hit_list || !hit_keys)))) return RTT_E_NULL;
  if (n == 0) return RTT_OK;
  RayIO io = {pos_in, dir_in, inten_in, ld_in, pos_out, dir_out, inten_out, ld_out, n};
  SensorIO se = {};
  se.app_hit = hit_list;
  se.app_key = hit_keys;
  se.app_count = any_sensor ? hit_count : nullptr;
  se.app_cap = (int32_t)(hit_capacity / hit_parts);
  se.app_parts = hit_parts - 1;
  auto aligned8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
  const int vec = (ld_in % 2 == 0) && (ld_out % 2 == 0) && aligned8(pos_in) && aligned8(dir_in) &&
                  aligned8(inten_in) && aligned8(pos_out) && aligned8(dir_out) && aligned8(inten_out);
  const Ops ops = make_ops(prog_host, n_surf);
  if (no_generic_visit(ops, prog_host, n_surf))
    hipLaunchKernelGGL((k_simulate_nearest<f2, false, false, 1>), dim3(grid_for(n, 2)), dim3(kBlock), 0, (hipStream_t)stream, io,
                       prog_dev, n_surf, ops, make_cull(prog_host, n_surf), max_bounces, (int32_t*)nullptr, bounces, se, vec);
  else
    hipLaunchKernelGGL((k_simulate_nearest<f2, false>), dim3(grid_for(n, 2)), dim3(kBlock), 0, (hipStream_t)stream, io, prog_dev,
                       n_surf, ops, make_cull(prog_host, n_surf), max_bounces, (int32_t*)nullptr, bounces, se, vec);
  return launch_status();
}

// workspace of the staged loop: kStageCounters int32 counters, then two sets of compact planes (pos 3, dir 3, intensity,
// index, bounce count: 9 words per ray) of `cap` rays each, cap = n rounded up to an even number, then the records of
// the rays that come to rest late (8 words per ray) and their marks (a byte per ray)
int64_t rtt_simulate_nearest_staged_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  const int64_t cap = (n + 1) & ~(int64_t)1;
  return 256 + ((2 * 9 * 4 * cap + 255) & ~(int64_t)255) + 8 * 4 * cap + cap;   // (a record fills one 32-byte sector)
}

int rtt_simulate_nearest_staged_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                                    float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                                    const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, int max_bounces,
                                    int32_t* bounces, float* hit_list, int32_t* hit_keys, int32_t* hit_count,
                                    int64_t hit_capacity, int hit_parts, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!pos_in || !dir_in || !inten_in || !pos_out || !dir_out || !inten_out || !prog_dev || !prog_host || !workspace)
    return RTT_E_NULL;
  if (n < 0 || n > 0x7fffffff || ld_in < n || ld_out < n || max_bounces < 0 || hit_capacity < 0 ||
      hit_capacity > 0x7fffffff || workspace_bytes < rtt_simulate_nearest_staged_workspace_bytes(n))
    return RTT_E_SIZE;
  if (hit_parts < 1 || hit_parts > 64 || (hit_parts & (hit_parts - 1))) return RTT_E_SIZE;
  if (misaligned(pos_in) || misaligned(dir_in) || misaligned(inten_in) || misaligned(pos_out) || misaligned(dir_out) ||
      misaligned(inten_out) || misaligned(bounces) || misaligned(hit_list) || misaligned(hit_keys) || misaligned(hit_count) ||
      (reinterpret_cast<uintptr_t>(workspace) & 255u))
    return RTT_E_ALIGN;
  if (pos_in == pos_out || dir_in == dir_out || inten_in == inten_out) return RTT_E_ALIGN;  // rays come to rest at different times
  bool any_sensor;
  int rc = check_program(prog_host, n_surf, &any_sensor, nullptr);
  if (rc != RTT_OK) return rc;
  for (int s = 0; s < n_surf; ++s)
    if (prog_host[s].phys_kind == RTT_PHYS_FRESNEL) return RTT_E_KIND;  // its draws are keyed by the place in the launch
  if (any_sensor && (!hit_count || (hit_capacity > 0 && (!hit_list || !hit_keys)))) return RTT_E_NULL;
  if (n == 0) return RTT_OK;
  if (max_bounces == 0)  // no stage would run: the single launch copies the rays through and writes zero bounce counts
    return rtt_simulate_nearest_f32(pos_in, dir_in, inten_in, ld_in, pos_out, dir_out, inten_out, ld_out, n, prog_dev, prog_host,
                                    n_surf, 0, bounces, hit_list, hit_keys, hit_count, hit_capacity, hit_parts, stream);
  const hipStream_t sm = (hipStream_t)stream;
  // bounces per stage: while (nearly) every ray still moves a stage boundary only costs; then rays drop out quickly,
  // later the few that are left stay for long.  Measured on the 32-surface scene at 4 Mi rays x 100 bounces: 3.3 ms with
  // this schedule, 3.4 ms with finer or coarser ones
  const StageSchedule& schedule = stage_schedule();
  const int* kStageBounces = schedule.bounces;
  const int kMaxStages = schedule.stages;
  if (hipMemsetAsync(workspace, 0, 256, sm) != hipSuccess) return RTT_E_LAUNCH;
  int32_t* counters = (int32_t*)workspace;
  const int64_t cap = (n + 1) & ~(int64_t)1;
  float* sets[2] = {(float*)((char*)workspace + 256), (float*)((char*)workspace + 256) + 9 * cap};
  float* rec = (float*)((char*)workspace + 256 + ((2 * 9 * 4 * cap + 255) & ~(int64_t)255));
  uint8_t* late = (uint8_t*)(rec + 8 * cap);
  SensorIO se = {};
  se.app_hit = hit_list;
  se.app_key = hit_keys;
  se.app_count = any_sensor ? hit_count : nullptr;
  se.app_cap = (int32_t)(hit_capacity / hit_parts);
  se.app_parts = hit_parts - 1;
  auto aligned8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
  const Ops ops = make_ops(prog_host, n_surf);
  const Cull cull = make_cull(prog_host, n_surf);
  const bool lean = no_generic_visit(ops, prog_host, n_surf);
  int done = 0;
  for (int s = 0; s < kMaxStages && done < max_bounces; ++s) {
    const int todo = kStageBounces[s] < max_bounces - done ? kStageBounces[s] : max_bounces - done;
    const bool last = done + todo >= max_bounces;
    RayIO io = {pos_in, dir_in, inten_in, ld_in, pos_out, dir_out, inten_out, ld_out, n};
    int vec = (ld_in % 2 == 0) && aligned8(pos_in) && aligned8(dir_in) && aligned8(inten_in);
    StageIO st = {};
    st.bounce0 = done;
    if (s > 0) {  // the rays the stage before handed on
      float* in = sets[(s - 1) & 1];
      io.pos_in = in;
      io.dir_in = in + 3 * cap;
      io.inten_in = in + 6 * cap;
      io.ld_in = cap;
      vec = 1;
      st.count_in = counters + s;
      st.orig_in = (const int32_t*)(in + 7 * cap);
      st.taken_in = (const int32_t*)(in + 8 * cap);
    }
    const bool records = g_late_records != 0;
    st.rec = records ? rec : nullptr;
    st.late = (records && s == 0 && !last) ? late : nullptr;
    if (!last) {
      float* out = sets[s & 1];
      st.next_pos = out;
      st.next_dir = out + 3 * cap;
      st.next_inten = out + 6 * cap;
      st.next_orig = (int32_t*)(out + 7 * cap);
      st.next_taken = (int32_t*)(out + 8 * cap);
      st.next_count = counters + s + 1;
      st.next_ld = cap;
    }
    // the number of rays of a later stage is known on the device only: a bounded grid walks over it
    int grid = grid_for(n, 2);
    if (s > 0 && grid > 8192) grid = 8192;
    if (lean)
      hipLaunchKernelGGL((k_simulate_nearest<f2, false, true, 1>), dim3(grid), dim3(kBlock), 0, sm, io, prog_dev, n_surf, ops, cull,
                         todo, (int32_t*)nullptr, bounces, se, vec, st);
    else
      hipLaunchKernelGGL((k_simulate_nearest<f2, false, true>), dim3(grid), dim3(kBlock), 0, sm, io, prog_dev, n_surf, ops, cull, todo,
                         (int32_t*)nullptr, bounces, se, vec, st);
    if (launch_status() != RTT_OK) return RTT_E_LAUNCH;
    done += todo;
    if (last && s > 0 && g_late_records) {
      RayIO place = {nullptr, nullptr, nullptr, 0, pos_out, dir_out, inten_out, ld_out, n};
      hipLaunchKernelGGL(k_place_late, dim3(grid_for(n)), dim3(kBlock), 0, sm, (const uint8_t*)late, (const float*)rec, place, bounces);
      if (launch_status() != RTT_OK) return RTT_E_LAUNCH;
    }
  }
  return RTT_OK;
}

int rtt_simulate_nearest_tangent_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                                     float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                                     const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, int max_bounces,
                                     const float* tan_dev, int n_theta, int seed_rays, float* ray_tan, float* hit_list,
                                     int32_t* hit_keys, float* hit_tan, int32_t* hit_count, int64_t hit_capacity,
                                     void* stream) {
  if (!pos_in || !dir_in || !inten_in || !pos_out || !dir_out || !inten_out || !prog_dev || !prog_host || !tan_dev || !ray_tan)
    return RTT_E_NULL;
  if (n < 0 || n > 0x7fffffff || ld_in < n || ld_out < n || max_bounces < 0 || hit_capacity < 0 ||
      hit_capacity > 0x7fffffff || n_theta < 2 || n_theta > 8 || (n_theta & 1) || (seed_rays && n_theta < 6))
    return RTT_E_SIZE;
  if (misaligned(pos_in) || misaligned(dir_in) || misaligned(inten_in) || misaligned(pos_out) || misaligned(dir_out) ||
      misaligned(inten_out) || misaligned(tan_dev) || misaligned(ray_tan) || misaligned(hit_list) || misaligned(hit_keys) ||
      misaligned(hit_tan) || misaligned(hit_count))
    return RTT_E_ALIGN;
  bool any_sensor;
  int rc = check_program(prog_host, n_surf, &any_sensor, nullptr);
  if (rc != RTT_OK) return rc;
  if (any_sensor && (!hit_count || (hit_capacity > 0 && (!hit_list || !hit_keys || !hit_tan)))) return RTT_E_NULL;
  if (n == 0) return RTT_OK;
  RayIO io = {pos_in, dir_in, inten_in, ld_in, pos_out, dir_out, inten_out, ld_out, n};
  SensorIO se = {};
  se.app_hit = hit_list;
  se.app_key = hit_keys;
  se.app_count = hit_count;
  se.app_cap = (int32_t)hit_capacity;
  const NearestTangentIO tio = {tan_dev, ray_tan, hit_tan, seed_rays ? 1 : 0};
  const dim3 grid(grid_for(n, 1)), block(kBlock);
  const Ops ops = make_ops(prog_host, n_surf);
  const Cull cull = make_cull(prog_host, n_surf);
  hipStream_t st = (hipStream_t)stream;
  switch (n_theta) {
    case 2: hipLaunchKernelGGL((k_simulate_nearest_tangent<2>), grid, block, 0, st, io, prog_dev, n_surf, ops, cull, max_bounces, se, tio); break;
    case 4: hipLaunchKernelGGL((k_simulate_nearest_tangent<4>), grid, block, 0, st, io, prog_dev, n_surf, ops, cull, max_bounces, se, tio); break;
    case 6: hipLaunchKernelGGL((k_simulate_nearest_tangent<6>), grid, block, 0, st, io, prog_dev, n_surf, ops, cull, max_bounces, se, tio); break;
    default: hipLaunchKernelGGL((k_simulate_nearest_tangent<8>), grid, block, 0, st, io, prog_dev, n_surf, ops, cull, max_bounces, se, tio); break;
  }
  return launch_status();
}

int rtt_spot_groups_f32(const float* x, const float* y, const float* w, const uint8_t* mask, const int8_t* id, int64_t n,
                        int pass, const double* centers, double norm_ord, double* out, void* stream) {
  if (!x || !y || !w || !id || !out || (pass == 1 && !centers)) return RTT_E_NULL;
  if (n < 0 || (pass != 0 && pass != 1)) return RTT_E_SIZE;
  if (misaligned(x) || misaligned(y) || misaligned(w) || (reinterpret_cast<uintptr_t>(out) & 7u) ||
      (reinterpret_cast<uintptr_t>(centers) & 7u))
    return RTT_E_ALIGN;
  if (n == 0) return RTT_OK;
  hipLaunchKernelGGL(k_spot_groups, dim3(grid_for_reduction(n)), dim3(kBlock), 0, (hipStream_t)stream, x, y, w, mask, id, n,
                     pass, centers, (float)norm_ord, out);
  return launch_status();
}

int rtt_render_f32(const float* pos, const float* dir, int64_t ld, int64_t n, const RttSurface* prog_dev,
                   const RttSurface* prog_host, int n_surf, const float* base_rgb, uint64_t skip_mask,
                   const float light_dir[3], const float background[3], float* rgb, void* stream) {
  if (!pos || !dir || !prog_dev || !base_rgb || !light_dir || !background || !rgb) return RTT_E_NULL;
  if (n < 0 || ld < n) return RTT_E_SIZE;
  if (misaligned(pos) || misaligned(dir) || misaligned(base_rgb) || misaligned(rgb)) return RTT_E_ALIGN;
  int rc = check_program(prog_host, n_surf, nullptr, nullptr);
  if (rc != RTT_OK) return rc;
  if (n == 0) return RTT_OK;
  ShadeIO sh = {base_rgb, skip_mask, {light_dir[0], light_dir[1], light_dir[2]}, {background[0], background[1], background[2]}};
  hipLaunchKernelGGL(k_render, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, pos, dir, ld, n, prog_dev, n_surf,
                     make_ops(prog_host, n_surf), make_cull(prog_host, n_surf), sh, rgb);
  return launch_status();
}

int rtt_sample_bundle_f32(const RttSource* src, float* pos, float* dir, int64_t ld, float* inten, int8_t* id,
                          float* wavelength, int64_t n, void* stream) {
  if (!src || !pos || !dir) return RTT_E_NULL;
  if (n < 0 || ld < n) return RTT_E_SIZE;
  if (src->kind < RTT_SRC_PENCIL || src->kind > RTT_SRC_SPHERE) return RTT_E_KIND;
  if (src->kind == RTT_SRC_PANEL && (src->area_kind < RTT_AREA_RECT || src->area_kind > RTT_AREA_RING ||
                                     src->emit_kind < RTT_EMIT_LAMBERT || src->emit_kind > RTT_EMIT_CONE))
    return RTT_E_KIND;
  if (misaligned(pos) || misaligned(dir) || misaligned(inten) || misaligned(wavelength)) return RTT_E_ALIGN;
  if (n == 0) return RTT_OK;
  const BundleOut o{pos, dir, ld, inten, id, wavelength};
  if (src->kind == RTT_SRC_PANEL || src->kind == RTT_SRC_SPHERE) {
    hipLaunchKernelGGL(k_sample_bundle_wide, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       *src, o, n);
    return launch_status();
  }
  const uint64_t first_pair = src->offset >> 1, last_pair = (src->offset + (uint64_t)n - 1) >> 1;
  const uint64_t pairs = last_pair - first_pair + 1;
  auto aligned8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
  const int vec = (src->offset % 2 == 0) && (ld % 2 == 0) && aligned8(pos) && aligned8(dir) && aligned8(inten) &&
                  aligned8(wavelength) && (reinterpret_cast<uintptr_t>(id) & 1u) == 0;
  hipLaunchKernelGGL(k_sample_bundle, dim3((unsigned)((pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     *src, posed_axis(*src), o, n, vec);
  return launch_status();
}

// ---- prepared launches -----------------------------------------------------------------------------------------------
// Everything a launch derives from its arguments that does not change from call to call (validated program, opcode
// and culling tables, access width, grid, kernel variant), kept in host memory behind an opaque handle: a stepping loop
// then pays a three-argument call, one memset (bounce) and the launch.
}  // extern "C"

namespace {
int g_tune_epoch = 0;  // bumped by rtt_tune: a handle prepared under other settings is stale
}

struct RttLaunch {
  int kind;  // 0: sequential trace, 1: one nearest-hit bounce
  int epoch;
  RayIO io;
  const RttSurface* prog_dev;
  int n_surf, vec, n_slots, variant;
  int lean;    // one bounce: no record of the program needs the generic visit
  int device;  // the device that was current when the handle was made (its planes live there), -1 = unknown
  int64_t sens_ld;
  Ops ops;
  Cull cull;
  unsigned grid;
};

extern "C" {

int rtt_prepare_trace_seq_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, float* pos_out,
                              float* dir_out, float* inten_out, int64_t ld_out, int64_t n, const RttSurface* prog_dev,
                              const RttSurface* prog_host, int n_surf, int64_t sens_ld, RttLaunch** out) {
  if (!out) return RTT_E_NULL;
  *out = nullptr;
  if (!pos_in || !dir_in || !inten_in || !pos_out || !dir_out || !inten_out || !prog_dev || !prog_host) return RTT_E_NULL;
  if (n < 1 || ld_in < n || ld_out < n) return RTT_E_SIZE;
  if (misaligned(pos_in) || misaligned(dir_in) || misaligned(inten_in) || misaligned(pos_out) || misaligned(dir_out) ||
      misaligned(inten_out))
    return RTT_E_ALIGN;
  bool any_sensor;
  int n_slots = 0;
  const int rc = check_program(prog_host, n_surf, &any_sensor, &n_slots);
  if (rc != RTT_OK) return rc;
  if (any_sensor && sens_ld < n) return RTT_E_SIZE;
  RttLaunch* h = new (std::nothrow) RttLaunch();
  if (!h) return RTT_E_LAUNCH;
  h->kind = 0;
  h->epoch = g_tune_epoch;
  if (hipGetDevice(&h->device) != hipSuccess) h->device = -1;
  h->io = {pos_in, dir_in, inten_in, ld_in, pos_out, dir_out, inten_out, ld_out, n};
  h->prog_dev = prog_dev;
  h->n_surf = n_surf;
  h->n_slots = any_sensor ? n_slots : 0;
  h->sens_ld = sens_ld;
  h->ops = make_ops(prog_host, n_surf);
  auto aligned8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
  h->vec = (ld_in % 2 == 0) && (ld_out % 2 == 0) && aligned8(pos_in) && aligned8(dir_in) && aligned8(inten_in) &&
           aligned8(pos_out) && aligned8(dir_out) && aligned8(inten_out);
  h->variant = g_rays_per_lane == 1 ? 1 : (g_rays_per_lane == 4 ? 4 : (g_program_in_lds ? 3 : 2));
  h->grid = (unsigned)grid_for(n, h->variant == 1 ? 1 : (h->variant == 4 ? 4 : 2));
  *out = h;
  return RTT_OK;
}

int rtt_prepare_step_nearest_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, float* pos_out,
                                 float* dir_out, float* inten_out, int64_t ld_out, int64_t n, const RttSurface* prog_dev,
                                 const RttSurface* prog_host, int n_surf, int64_t sens_ld, RttLaunch** out) {
  const int rc = rtt_prepare_trace_seq_f32(pos_in, dir_in, inten_in, ld_in, pos_out, dir_out, inten_out, ld_out, n, prog_dev,
                                           prog_host, n_surf, sens_ld, out);
  if (rc != RTT_OK) return rc;
  RttLaunch* h = *out;
  h->kind = 1;
  h->lean = no_generic_visit(h->ops, prog_host, n_surf) ? 1 : 0;
  h->cull = make_cull(prog_host, n_surf);
  h->variant = 2;
  h->grid = (unsigned)grid_for(n, 2);
  return RTT_OK;
}

int rtt_launch_prepared(const RttLaunch* h, void* sens_base, void* stream) {
  if (!h) return RTT_E_NULL;
  if (h->epoch != g_tune_epoch) return RTT_E_KIND;  // prepared under other rtt_tune settings: prepare again
  int dev = -1;
  if (h->device >= 0 && (hipGetDevice(&dev) != hipSuccess || dev != h->device)) return RTT_E_KIND;  // another device is current
  if (h->n_slots > 0 && !sens_base) return RTT_E_NULL;
  if (misaligned(sens_base)) return RTT_E_ALIGN;
  // one allocation holds the record: hit planes [slots][3][ld], intensities [slots][ld], masks [slots][ld] bytes
  SensorIO se = {};
  if (h->n_slots > 0) {
    const int64_t plane = (int64_t)h->n_slots * h->sens_ld;
    se.hit = (float*)sens_base;
    se.w = se.hit + 3 * plane;
    se.mask = (uint8_t*)(se.w + plane);
    se.ld = h->sens_ld;
  }
  const hipStream_t st = (hipStream_t)stream;
  if (h->kind == 1) {
    se.zero_slots = h->n_slots;   // the kernel clears the mask entries of its rays before the bounce
    // (8-byte accesses also need the masks' pairs 2-byte aligned: an odd sens_ld or base falls back to 4-byte accesses)
    const int vec = h->vec && (h->sens_ld % 2 == 0) && (reinterpret_cast<uintptr_t>(se.mask) & 1u) == 0;
    if (h->lean)
      hipLaunchKernelGGL((k_simulate_nearest<f2, true, false, 1>), dim3(h->grid), dim3(kBlock), 0, st, h->io, h->prog_dev, h->n_surf,
                         h->ops, h->cull, 1, (int32_t*)nullptr, (int32_t*)nullptr, se, vec);
    else
      hipLaunchKernelGGL((k_simulate_nearest<f2, true>), dim3(h->grid), dim3(kBlock), 0, st, h->io, h->prog_dev, h->n_surf, h->ops,
                         h->cull, 1, (int32_t*)nullptr, (int32_t*)nullptr, se, vec);
    return launch_status();
  }
  switch (h->variant) {
    case 1: hipLaunchKernelGGL((k_trace_seq<float, 1, false>), dim3(h->grid), dim3(kBlock), 0, st, h->io, h->prog_dev, h->n_surf, h->ops, se, 0); break;
    case 4: hipLaunchKernelGGL((k_trace_seq<f2, 2, false>), dim3(h->grid), dim3(kBlock), 0, st, h->io, h->prog_dev, h->n_surf, h->ops, se, h->vec); break;
    case 3: hipLaunchKernelGGL((k_trace_seq<f2, 1, true>), dim3(h->grid), dim3(kBlock), 0, st, h->io, h->prog_dev, h->n_surf, h->ops, se, h->vec); break;
    default: hipLaunchKernelGGL((k_trace_seq<f2, 1, false>), dim3(h->grid), dim3(kBlock), 0, st, h->io, h->prog_dev, h->n_surf, h->ops, se, h->vec); break;
  }
  return launch_status();
}

void rtt_release_prepared(RttLaunch* h) { delete h; }

int64_t rtt_sensor_record_bytes(int n_slots, int64_t sens_ld) {
  if (n_slots < 0 || sens_ld < 0) return 0;
  return ((int64_t)17 * n_slots * sens_ld + 3) & ~(int64_t)3;
}

const char* rtt_strerror(int code) {
  switch (code) {
    case RTT_OK: return "ok";
    case RTT_E_NULL: return "required pointer is NULL";
    case RTT_E_SIZE: return "size, leading dimension or index out of range";
    case RTT_E_KIND: return "surface record carries an unknown or inconsistent kind (or a prepared launch is stale)";
    case RTT_E_LAUNCH: return "kernel launch failed";
    case RTT_E_ALIGN: return "pointer is not sufficiently aligned";
    default: return "unknown error";
  }
}

int rtt_abi_version(void) { return RTT_ABI_VERSION; }

int rtt_tune(int key, int value) {
  ++g_tune_epoch;
  if (key == 0 && (value == 1 || value == 2 || value == 4)) {
    g_rays_per_lane = value;
    return RTT_OK;
  }
  if (key == 2 && (value == 0 || value == 1)) {
    g_program_in_lds = value;
    return RTT_OK;
  }
  if (key == 1 && value >= 1 && value <= (1 << 20)) {
    g_max_blocks = value;
    return RTT_OK;
  }
  if (key == 3 && (value == 0 || value == 1)) {
    g_cull = value;
    return RTT_OK;
  }
  if (key == 4 && value >= 1 && value <= (1 << 20)) {
    g_moments_blocks = value;
    return RTT_OK;
  }
  if (key == 5 && value >= 1 && value <= (1 << 20)) {
    g_reduction_blocks = value;
    return RTT_OK;
  }
  if (key == 6 && value >= 0 && value <= 2) {
    g_adjoint_lane = value;
    return RTT_OK;
  }
  if (key == 8 && (value == 256 || value == 1024)) {
    g_moments_threads = value;
    return RTT_OK;
  }
  if (key == 9 && value >= 1 && value <= 8) {
    g_adjoint_rounds = value;
    return RTT_OK;
  }
  if (key == 7 && (value == 0 || value == 1)) {
    g_adjoint_state = value;
    return RTT_OK;
  }
  if (key == 10 && (value == 0 || value == 1)) {
    g_late_records = value;
    return RTT_OK;
  }
  if (key == 11 && (value == 0 || value == 1)) {
    g_adjoint_lens_set = value;
    return RTT_OK;
  }
  if (key == 12 && (value == 0 || value == 1)) {
    g_lean_bounce_loop = value;
    return RTT_OK;
  }
  return RTT_E_SIZE;
}

int rtt_surface_record_bytes(void) { return (int)sizeof(RttSurface); }

#if RTT_PHASE_TIMERS
// Profiling build only (`make prof`): the phase clocks of the reverse sweep summed over all wavefronts since the last
// reset; out[k] for k < n.
int rtt_debug_phases(unsigned long long* out, int n, int reset) {
  unsigned long long host[kPhases] = {};
  if (hipDeviceSynchronize() != hipSuccess) return RTT_E_LAUNCH;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase), sizeof(host)) != hipSuccess) return RTT_E_LAUNCH;
  for (int k = 0; k < n && k < kPhases; ++k) out[k] = host[k];
  if (reset) {
    const unsigned long long zero[kPhases] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof(zero)) != hipSuccess) return RTT_E_LAUNCH;
  }
  return RTT_OK;
}
#endif

}  // extern "C"
