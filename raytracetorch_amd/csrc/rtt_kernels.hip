// rtt_kernels.hip — gfx950 (MI355X / CDNA4) kernels and C ABI of the RayTraceTorch ray-propagation path.
//
// Execution model
//   * one lane = one ray; the ray state (pos, dir, intensity: 7 floats) stays in VGPRs for the whole
//     surface program, so HBM sees 28 B in + 28 B out per ray per trace regardless of surface count;
//   * ray storage is structure-of-arrays, a wavefront reads/writes 256 contiguous bytes per component;
//   * the surface loop is wave-uniform: record `s` is addressed with a scalar index, so its fields come
//     in through scalar loads (SGPRs) and the `switch` on kinds is a scalar branch — no divergence in
//     the sequential trace.  In the nearest-hit step, lanes that won different surfaces are served by
//     a readfirstlane waterfall so the record stays scalar there as well;
//   * no MFMA: the work is elementwise geometry (a few hundred flops per 56 B), HBM-bound.
//
// Numerics follow the reference's float32 torch-CPU evaluation order (cited per function, paths
// relative to the reference tree); the file is built with -ffp-contract=off so no product-sum is
// fused, and hipcc's default correctly-rounded fp32 divide / sqrt is kept.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtt_hip.h"

namespace {

constexpr float kInf = __builtin_huge_valf();
constexpr float kEps = 1e-6f;  // geom/primitives.py:6 and :21
constexpr int kBlock = 256;

struct V3 {
  float x, y, z;
};

__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// v @ R, R row-major (geom/transform.py:92-93).  `ident` skips the products: with R exactly the
// identity x*1 + y*0 + z*0 == x bit for bit for finite inputs.
__device__ __forceinline__ V3 rowmat(V3 v, const float* R, bool ident) {
  if (ident) return v;
  return {v.x * R[0] + v.y * R[3] + v.z * R[6], v.x * R[1] + v.y * R[4] + v.z * R[7],
          v.x * R[2] + v.y * R[5] + v.z * R[8]};
}

// v @ R^T (geom/primitives.py:94, geom/shape.py:85)
__device__ __forceinline__ V3 rowmat_t(V3 v, const float* R, bool ident) {
  if (ident) return v;
  return {v.x * R[0] + v.y * R[1] + v.z * R[2], v.x * R[3] + v.y * R[4] + v.z * R[5],
          v.x * R[6] + v.y * R[7] + v.z * R[8]};
}

// torch.min over a stacked dim propagates NaN (geom/primitives.py:34, geom/bounded.py:34)
__device__ __forceinline__ float nanmin(float a, float b) {
  if (a != a) return a;
  if (b != b) return b;
  return a < b ? a : b;
}

struct Frames {
  V3 pe, de;  // element frame (de un-normalised, geom/shape.py:47)
  V3 pl, dl;  // surface frame
};

// Global -> element frame (+ re-normalisation, rays/ray.py:22-25 via with_coords :84-97)
// -> surface frame (geom/shape.py:37-38, geom/primitives.py:50, geom/transform.py:90-93).
__device__ __forceinline__ Frames to_frames(const RttSurface& S, V3 p, V3 d) {
  Frames f;
  V3 den;
  if (S.flags & RTT_F_HAS_SHAPE) {
    const bool id = S.flags & RTT_F_SHAPE_ROT_ID;
    V3 ts = {S.ts[0], S.ts[1], S.ts[2]};
    f.pe = rowmat(p - ts, S.Rs, id);
    f.de = rowmat(d, S.Rs, id);
    float n = sqrtf(f.de.x * f.de.x + f.de.y * f.de.y + f.de.z * f.de.z);
    n = fmaxf(n, 1e-12f);
    den = {f.de.x / n, f.de.y / n, f.de.z / n};
  } else {
    f.pe = p;
    f.de = d;
    den = d;
  }
  const bool idj = S.flags & RTT_F_SURF_ROT_ID;
  V3 tj = {S.tj[0], S.tj[1], S.tj[2]};
  f.pl = rowmat(f.pe - tj, S.Rj, idj);
  f.dl = rowmat(den, S.Rj, idj);
  return f;
}

// Quadric._solve_quadratic (geom/primitives.py:290-320) and the Cone variant (:436-466).
template <bool ABS_DISC>
__device__ __forceinline__ void quadratic_with_linear_fallback(float A, float B, float C, float& t1, float& t2) {
  const float disc = B * B - 4.0f * A * C;
  const bool hit = disc >= 0.0f;
  const bool lin = fabsf(A) < kEps;
  const float sq = ABS_DISC ? sqrtf(fabsf(disc)) : sqrtf(hit ? disc : 0.0f);
  const float A_safe = lin ? 1.0f : A;
  const float den = 2.0f * A_safe;
  t1 = (-B - sq) / den;
  t2 = (-B + sq) / den;
  t1 = hit ? t1 : kInf;
  t2 = hit ? t2 : kInf;
  if (lin) {
    const float B_safe = fabsf(B) < kEps ? kEps : B;
    t1 = t2 = -C / B_safe;
  }
}

// _solve_t of each primitive in the surface frame.  Returns the number of roots (1 or 2).
__device__ __forceinline__ int solve_roots(const RttSurface& S, V3 o, V3 d, float& t1, float& t2) {
  switch (S.surf_kind) {
    case RTT_SURF_PLANE: {  // geom/primitives.py:124-136
      const float safe = fabsf(d.z) < kEps ? 1e-8f : d.z;
      t1 = -o.z / safe;
      t2 = t1;
      return 1;
    }
    case RTT_SURF_SPHERE: {  // geom/primitives.py:155-181
      const float b = 2.0f * dot3(o, d);
      const float c = dot3(o, o) - S.geom[1];
      const float disc = b * b - 4.0f * c;
      const bool hit = disc >= 0.0f;
      const float sq = sqrtf(hit ? disc : 0.0f);
      t1 = hit ? (-b - sq) / 2.0f : kInf;
      t2 = hit ? (-b + sq) / 2.0f : kInf;
      return 2;
    }
    case RTT_SURF_CYLINDER: {  // geom/primitives.py:201-231
      const float A = d.x * d.x + d.y * d.y;
      const float B = 2.0f * (o.x * d.x + o.y * d.y);
      const float C = (o.x * o.x + o.y * o.y) - S.geom[1];
      const float disc = B * B - 4.0f * A * C;
      const bool hit = disc >= 0.0f;
      const float sq = sqrtf(fabsf(disc));
      const float den = 2.0f * A;
      t1 = hit ? (-B - sq) / den : kInf;
      t2 = hit ? (-B + sq) / den : kInf;
      return 2;
    }
    case RTT_SURF_QUADRIC: {  // geom/primitives.py:266-288
      const float c = S.geom[0], c1k = S.geom[2];
      const float two_c = 2.0f * c, two_c1k = 2.0f * c1k;
      const float A = c * (d.x * d.x + d.y * d.y) + c1k * (d.z * d.z);
      const float B = two_c * (o.x * d.x + o.y * d.y) + two_c1k * o.z * d.z - 2.0f * d.z;
      const float C = c * (o.x * o.x + o.y * o.y) + c1k * (o.z * o.z) - 2.0f * o.z;
      quadratic_with_linear_fallback<true>(A, B, C, t1, t2);
      return 2;
    }
    case RTT_SURF_QUADRIC_ZY: {  // geom/primitives.py:356-377
      const float c = S.geom[0], c1k = S.geom[2];
      const float two_c = 2.0f * c, two_c1k = 2.0f * c1k;
      const float A = c * (d.y * d.y) + c1k * (d.z * d.z);
      const float B = two_c * (o.y * d.y) + two_c1k * o.z * d.z - 2.0f * d.z;
      const float C = c * (o.y * o.y) + c1k * (o.z * o.z) - 2.0f * o.z;
      quadratic_with_linear_fallback<true>(A, B, C, t1, t2);
      return 2;
    }
    default: {  // RTT_SURF_CONE, geom/primitives.py:416-466
      const float k2 = S.geom[1];
      const float A = d.z * d.z - k2 * (d.x * d.x + d.y * d.y);
      const float B = 2.0f * (o.z * d.z - k2 * (o.x * d.x + o.y * d.y));
      const float C = o.z * o.z - k2 * (o.x * o.x + o.y * o.y);
      quadratic_with_linear_fallback<false>(A, B, C, t1, t2);
      return 2;
    }
  }
}

// SurfaceBounded.inBounds variants in the surface frame, without `invert` (geom/bounded.py).
__device__ __forceinline__ bool surface_in_bounds(const RttSurface& S, V3 h) {
  switch (S.bound_kind) {
    case RTT_BND_DISK:  // bounded.py:60-64
      return (h.x * h.x + h.y * h.y) <= S.bnd[0];
    case RTT_BND_RECT:  // bounded.py:77-82
      return (fabsf(h.x) <= S.bnd[0]) && (fabsf(h.y) <= S.bnd[1]);
    case RTT_BND_ELLIPSE: {  // bounded.py:98-106
      const float dM = h.x * S.bnd[0] - h.y * S.bnd[1];
      const float dm = h.x * S.bnd[1] + h.y * S.bnd[0];
      const float a = dM / S.bnd[2], b = dm / S.bnd[3];
      return (a * a + b * b) <= 1.0f;
    }
    case RTT_BND_HALF:  // bounded.py:123-127, :171-174
      return fabsf(h.z * S.geom[0]) < 1.000001f;
    case RTT_BND_HALF_DISK:  // bounded.py:151-159
      return (fabsf(h.z * S.geom[0]) < 1.000001f) && ((h.x * h.x + h.y * h.y) <= S.bnd[0]);
    case RTT_BND_SINGLE_CONE:  // bounded.py:208-217
      return (h.z * S.geom[0]) >= -kEps;
    default:  // RTT_BND_ALWAYS
      return true;
  }
}

// Surface._check_t (geom/primitives.py:28-36) / SurfaceBounded._check_t (geom/bounded.py:20-36)
__device__ __forceinline__ float filter_root(const RttSurface& S, const Frames& f, float r) {
  bool bad = r <= kEps;
  if (S.bound_kind != RTT_BND_NONE) {
    V3 h = f.pl + r * f.dl;
    bool keep = surface_in_bounds(S, h);
    if (S.flags & RTT_F_INVERT) keep = !keep;
    bad = bad || !keep;
  }
  return bad ? kInf : r;
}

__device__ __forceinline__ float solve_t(const RttSurface& S, const Frames& f) {
  float t1, t2;
  const int n = solve_roots(S, f.pl, f.dl, t1, t2);
  t1 = filter_root(S, f, t1);
  if (n == 1) return t1;
  t2 = filter_root(S, f, t2);
  return nanmin(t1, t2);
}

// HalfCyl.sagittalZ with a per-ray height (geom/bounded.py:176-186)
__device__ __forceinline__ float sag(float c, float tz, float y) {
  const float r_sq = y * y;
  float term = 1.0f - (c * c) * r_sq;
  term = term > 0.0f ? term : (term != term ? term : 0.0f);  // relu keeps NaN
  return (c * r_sq) / (1.0f + sqrtf(term)) + tz;
}

// Shape.inBounds(hit, j) in the element frame (geom/spherics.py:27-46, geom/cylindrics.py:23-55)
__device__ __forceinline__ bool shape_in_bounds(const RttSurface& S, V3 h) {
  switch (S.shape_kind) {
    case RTT_SHP_SPHERIC_FACE:
      return (h.x * h.x + h.y * h.y) <= S.shp[0];
    case RTT_SHP_SPHERIC_EDGE:
      return (h.z >= S.shp[0]) && (h.z <= S.shp[1]);
    case RTT_SHP_CYL_FACE:
    case RTT_SHP_CYL_EDGE: {
      const bool ap = (h.x <= S.shp[0]) && (h.x >= S.shp[1]) && (h.y <= S.shp[2]) && (h.y >= S.shp[3]);
      if (S.shape_kind == RTT_SHP_CYL_FACE) return ap;
      const float zf = sag(S.shp[4], S.shp[5], h.y);
      const float zb = sag(S.shp[6], S.shp[7], h.y);
      return (h.z >= zf + 1e-4f) && (h.z <= zb - 1e-4f) && ap;
    }
    default:
      return true;
  }
}

// One column of Element.intersectTest (geom/shape.py:42-56 or geom/primitives.py:38-57).
// `f` and the un-masked t (what Shape.forward recomputes, geom/shape.py:78) are returned as well.
__device__ __forceinline__ float test_surface(const RttSurface& S, V3 p, V3 d, Frames& f, float& t_raw) {
  f = to_frames(S, p, d);
  float t = solve_t(S, f);
  t_raw = t;
  if (S.flags & RTT_F_HAS_SHAPE) {
    V3 he = f.pe + t * f.de;
    const bool valid = (t < kInf) && shape_in_bounds(S, he);
    t = valid ? t : kInf;
  }
  return t;
}

// _getNormal in the surface frame
__device__ __forceinline__ V3 surface_normal(const RttSurface& S, V3 h) {
  switch (S.surf_kind) {
    case RTT_SURF_PLANE:  // primitives.py:138-143
      return {0.0f, 0.0f, 1.0f};
    case RTT_SURF_SPHERE:  // primitives.py:183-187
      return {h.x / S.geom[0], h.y / S.geom[0], h.z / S.geom[0]};
    case RTT_SURF_CYLINDER:  // primitives.py:233-241
      return {h.x / S.geom[0], h.y / S.geom[0], 0.0f};
    case RTT_SURF_QUADRIC:
    case RTT_SURF_QUADRIC_ZY: {  // primitives.py:330-343, :379-395
      const float two_c = 2.0f * S.geom[0], two_c1k = 2.0f * S.geom[2];
      const float nx = S.surf_kind == RTT_SURF_QUADRIC ? two_c * h.x : 0.0f;
      const float ny = two_c * h.y;
      const float nz = two_c1k * h.z - 2.0f;
      const float len = sqrtf(nx * nx + ny * ny + nz * nz) + 1e-8f;
      return {-(nx / len), -(ny / len), -(nz / len)};
    }
    default: {  // cone, primitives.py:468-497
      const float k2 = S.geom[1];
      const float nx = -k2 * h.x, ny = -k2 * h.y, nz = h.z;
      const float len = sqrtf(nx * nx + ny * ny + nz * nz);
      if (len > 1e-8f) {
        const float l = len + 1e-8f;
        return {nx / l, ny / l, nz / l};
      }
      return {0.0f, 0.0f, 1.0f};
    }
  }
}

struct Forward {
  V3 hit_g, hit_l, n_g;
};

// Shape.forward / Surface.forward given frames and t (geom/shape.py:78-87, geom/primitives.py:85-96)
__device__ __forceinline__ Forward surface_forward(const RttSurface& S, V3 p, V3 d, const Frames& f, float t) {
  Forward o;
  o.hit_g = p + t * d;
  o.hit_l = f.pl + t * f.dl;
  V3 n = surface_normal(S, o.hit_l);
  n = rowmat_t(n, S.Rj, S.flags & RTT_F_SURF_ROT_ID);
  if (S.flags & RTT_F_HAS_SHAPE) n = rowmat_t(n, S.Rs, S.flags & RTT_F_SHAPE_ROT_ID);
  o.n_g = n;
  return o;
}

// SurfaceFunction.forward(hit_local, dir_global, normal_global) -> (new_dir, imult)
__device__ __forceinline__ void apply_physics(const RttSurface& S, V3 hl, V3 d, V3 n, V3& nd, float& im) {
  switch (S.phys_kind) {
    case RTT_PHYS_BLOCK:  // phys/std.py:243-254
      nd = {0.0f, 0.0f, 0.0f};
      im = 0.0f;
      return;
    case RTT_PHYS_REFLECT: {  // phys/std.py:97-108
      const float two_cos = 2.0f * dot3(d, n);
      nd = d - two_cos * n;
      im = 1.0f;
      return;
    }
    case RTT_PHYS_SNELL: {  // phys/std.py:123-145
      const float dot = dot3(d, n);
      const bool ent = dot < 0.0f;
      const V3 ne = ent ? n : V3{-n.x, -n.y, -n.z};
      const float c1 = fabsf(dot);
      const float mu = ent ? S.phys[2] : S.phys[3];
      const float term = 1.0f - mu * mu * (1.0f - c1 * c1);
      if (term < 0.0f) {
        const float two_dot = 2.0f * dot;
        nd = d - two_dot * n;
      } else {
        const float c2 = sqrtf(term > 0.0f ? term : (term != term ? term : 0.0f));
        const float g = mu * c1 - c2;
        nd = mu * d + g * ne;
      }
      im = 1.0f;
      return;
    }
    case RTT_PHYS_APERTURE: {  // phys/filter.py:24-33
      const float m = surface_in_bounds(S, hl) ? 1.0f : 0.0f;
      nd = m * d;
      im = m;
      return;
    }
    default:  // RTT_PHYS_TRANSMIT, phys/std.py:227-235
      nd = d;
      im = 1.0f;
      return;
  }
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
};

__device__ __forceinline__ void load_ray(const RayIO& io, int64_t i, V3& p, V3& d, float& I) {
  p = {io.pos_in[i], io.pos_in[io.ld_in + i], io.pos_in[2 * io.ld_in + i]};
  d = {io.dir_in[i], io.dir_in[io.ld_in + i], io.dir_in[2 * io.ld_in + i]};
  I = io.inten_in[i];
}

__device__ __forceinline__ void store_ray(const RayIO& io, int64_t i, V3 p, V3 d, float I) {
  io.pos_out[i] = p.x;
  io.pos_out[io.ld_out + i] = p.y;
  io.pos_out[2 * io.ld_out + i] = p.z;
  io.dir_out[i] = d.x;
  io.dir_out[io.ld_out + i] = d.y;
  io.dir_out[2 * io.ld_out + i] = d.z;
  io.inten_out[i] = I;
}

__device__ __forceinline__ void record_sensor(const SensorIO& se, int slot, int64_t i, bool hit, V3 hl, float I) {
  const int64_t base = (int64_t)slot * se.ld;
  se.mask[base + i] = hit ? 1 : 0;
  if (hit) {
    se.hit[(3 * base) + i] = hl.x;
    se.hit[(3 * base) + se.ld + i] = hl.y;
    se.hit[(3 * base) + 2 * se.ld + i] = hl.z;
    se.w[base + i] = I;
  }
}

// Apply surface S to one ray if it is hit: Element.forward + Rays.scatter_update
// (elements/parent.py:44-58, rays/ray.py:29-40, elements/sensor.py:35-37).
__device__ __forceinline__ void interact(const RttSurface& S, const Frames& f, float t_raw, bool hit, V3& p, V3& d,
                                         float& I, const SensorIO& se, int64_t i) {
  Forward fw = surface_forward(S, p, d, f, t_raw);
  V3 nd;
  float im;
  apply_physics(S, fw.hit_l, d, fw.n_g, nd, im);
  if (S.flags & RTT_F_SENSOR) record_sensor(se, S.sensor_slot, i, hit, fw.hit_l, I);
  if (hit) {
    p = fw.hit_g;
    d = nd;
    I = I * im;
  }
}

// ------------------------------------------------------------------------------------------------
// SequentialScene.simulate (scene/sequential.py:12-36)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_trace_seq(RayIO io, const RttSurface* __restrict__ prog, int n_surf,
                                                      SensorIO se) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < io.n; i += stride) {
    V3 p, d;
    float I;
    load_ray(io, i, p, d, I);
    for (int s = 0; s < n_surf; ++s) {
      const RttSurface& S = prog[s];
      Frames f;
      float t_raw;
      const float t = test_surface(S, p, d, f, t_raw);
      interact(S, f, t_raw, t < kInf, p, d, I, se, i);
    }
    store_ray(io, i, p, d, I);
  }
}

// ------------------------------------------------------------------------------------------------
// Scene.ray_cast / Scene.step (scene/base.py:144-235)
// ------------------------------------------------------------------------------------------------
// torch.min(t_matrix, dim=1): first index of the smallest value, NaN wins and sticks.
__device__ __forceinline__ void nearest(const RttSurface* __restrict__ prog, int n_surf, V3 p, V3 d, float& tmin,
                                        int& win) {
  tmin = kInf;
  win = 0;
  for (int s = 0; s < n_surf; ++s) {
    Frames f;
    float t_raw;
    const float t = test_surface(prog[s], p, d, f, t_raw);
    const bool take = (tmin == tmin) && ((t != t) || (t < tmin));
    if (take) {
      tmin = t;
      win = s;
    }
  }
  if (!(tmin < kInf)) win = -1;
}

__global__ void __launch_bounds__(kBlock) k_ray_cast(const float* __restrict__ pos, const float* __restrict__ dir,
                                                     int64_t ld, int64_t n, const RttSurface* __restrict__ prog,
                                                     int n_surf, int32_t* __restrict__ winner,
                                                     float* __restrict__ tmin_out) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    V3 p = {pos[i], pos[ld + i], pos[2 * ld + i]};
    V3 d = {dir[i], dir[ld + i], dir[2 * ld + i]};
    float tmin;
    int win;
    nearest(prog, n_surf, p, d, tmin, win);
    if (winner) winner[i] = win;
    if (tmin_out) tmin_out[i] = tmin;
  }
}

__global__ void __launch_bounds__(kBlock) k_step_nearest(RayIO io, const RttSurface* __restrict__ prog, int n_surf,
                                                         int32_t* __restrict__ winner, SensorIO se) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < io.n; i += stride) {
    V3 p, d;
    float I;
    load_ray(io, i, p, d, I);
    float tmin;
    int win;
    nearest(prog, n_surf, p, d, tmin, win);
    if (winner) winner[i] = win;
    // active_mask = hit & (intensity > 0), scene/base.py:201
    bool pending = (win >= 0) && (I > 0.0f);
    // Lanes of one wavefront may have won different surfaces: serve one surface per pass with its
    // record held in scalar registers (readfirstlane picks the surface of the first pending lane).
    while (pending) {
      const int s = __builtin_amdgcn_readfirstlane(win);
      if (win == s) {
        const RttSurface& S = prog[s];
        Frames f = to_frames(S, p, d);
        const float t_raw = solve_t(S, f);
        interact(S, f, t_raw, true, p, d, I, se, i);
        pending = false;
      }
    }
    store_ray(io, i, p, d, I);
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
    V3 p = {pos[i], pos[ld + i], pos[2 * ld + i]};
    V3 d = {dir[i], dir[ld + i], dir[2 * ld + i]};
    for (int s = 0; s < n_surf; ++s) {
      Frames f;
      float t_raw;
      t_out[(int64_t)s * ld_t + i] = test_surface(prog[s], p, d, f, t_raw);
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

__device__ __forceinline__ void store3(float* base, int64_t ld, int64_t i, V3 v) {
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
  const RttSurface& S = prog[index];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    V3 p = {pos[i], pos[ld + i], pos[2 * ld + i]};
    V3 d = {dir[i], dir[ld + i], dir[2 * ld + i]};
    Frames f = to_frames(S, p, d);
    const float t = solve_t(S, f);
    Forward fw = surface_forward(S, p, d, f, t);
    if (o.t) o.t[i] = t;
    store3(o.hit_g, o.ld, i, fw.hit_g);
    store3(o.n_g, o.ld, i, fw.n_g);
    store3(o.hit_l, o.ld, i, fw.hit_l);
    if (o.new_dir || o.imult) {
      V3 nd;
      float im;
      apply_physics(S, fw.hit_l, d, fw.n_g, nd, im);
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
  const RttSurface& S = prog[0];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    V3 h = {hl[i], hl[ld + i], hl[2 * ld + i]};
    V3 d = {dir[i], dir[ld + i], dir[2 * ld + i]};
    V3 nn = {nrm[i], nrm[ld + i], nrm[2 * ld + i]};
    V3 nd;
    float im;
    apply_physics(S, h, d, nn, nd, im);
    store3(new_dir, ld, i, nd);
    imult[i] = im;
  }
}

// ------------------------------------------------------------------------------------------------
// SpotSizeLoss moments (optim/goals.py:165-183)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__global__ void __launch_bounds__(kBlock) k_spot_moments(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ w,
                                                         const uint8_t* __restrict__ mask, int64_t n, int pass,
                                                         double cx, double cy, double wsum,
                                                         double* __restrict__ out) {
  __shared__ double part[3][kBlock / 64];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const float wi = w[i];
    const bool on = (mask == nullptr || mask[i] != 0) && (wi > 0.0f);
    if (on) {
      const double xd = x[i], yd = y[i], wd = wi;
      if (pass == 0) {
        a0 += wd;
        a1 += wd * xd;
        a2 += wd * yd;
      } else {
        const double dx = xd - cx, dy = yd - cy;
        a0 += sqrt((dx * dx + dy * dy) * (wd / wsum));
      }
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
    for (int k = 0; k < kBlock / 64; ++k) {
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
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int grid_for(int64_t n) {
  // memory-bound streaming: cap the grid at 256 CUs x 8 resident blocks and grid-stride the rest
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

int check_program(const RttSurface* host, int n_surf, bool* any_sensor, int* n_slots) {
  if (n_surf < 1 || n_surf > RTT_MAX_SURFACES) return RTT_E_SIZE;
  if (any_sensor) *any_sensor = false;
  if (n_slots) *n_slots = 0;
  if (!host) return RTT_OK;  // caller opted out of validation
  for (int s = 0; s < n_surf; ++s) {
    const RttSurface& S = host[s];
    if (S.surf_kind < RTT_SURF_PLANE || S.surf_kind > RTT_SURF_CONE) return RTT_E_KIND;
    if (S.bound_kind < RTT_BND_NONE || S.bound_kind > RTT_BND_ALWAYS) return RTT_E_KIND;
    if (S.shape_kind < RTT_SHP_NONE || S.shape_kind > RTT_SHP_CYL_EDGE) return RTT_E_KIND;
    if (S.phys_kind < RTT_PHYS_TRANSMIT || S.phys_kind > RTT_PHYS_APERTURE) return RTT_E_KIND;
    if ((S.shape_kind != RTT_SHP_NONE) != ((S.flags & RTT_F_HAS_SHAPE) != 0)) return RTT_E_KIND;
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
  hipLaunchKernelGGL(k_trace_seq, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, io, prog_dev, n_surf,
                     se);
  return launch_status();
}

int rtt_ray_cast_f32(const float* pos, const float* dir, int64_t ld, int64_t n, const RttSurface* prog_dev,
                     const RttSurface* prog_host, int n_surf, int32_t* winner, float* tmin, void* stream) {
  if (!pos || !dir || !prog_dev) return RTT_E_NULL;
  if (n < 0 || ld < n) return RTT_E_SIZE;
  if (misaligned(pos) || misaligned(dir) || misaligned(winner) || misaligned(tmin)) return RTT_E_ALIGN;
  int rc = check_program(prog_host, n_surf, nullptr, nullptr);
  if (rc != RTT_OK) return rc;
  if (n == 0) return RTT_OK;
  hipLaunchKernelGGL(k_ray_cast, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, pos, dir, ld, n,
                     prog_dev, n_surf, winner, tmin);
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
  int rc = check_program(prog_host, n_surf, &any_sensor, nullptr);
  if (rc != RTT_OK) return rc;
  if (!prog_host) any_sensor = sens_hit != nullptr;
  if (any_sensor && (!sens_hit || !sens_w || !sens_mask)) return RTT_E_NULL;
  if (any_sensor && sens_ld < n) return RTT_E_SIZE;
  if (n == 0) return RTT_OK;
  RayIO io = {pos_in, dir_in, inten_in, ld_in, pos_out, dir_out, inten_out, ld_out, n};
  SensorIO se = {sens_hit, sens_w, sens_mask, sens_ld};
  hipLaunchKernelGGL(k_step_nearest, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, io, prog_dev,
                     n_surf, winner, se);
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
    int rc = check_program(prog_host + surf_index, 1, nullptr, nullptr);
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
  if (prog_host && (prog_host->phys_kind < RTT_PHYS_TRANSMIT || prog_host->phys_kind > RTT_PHYS_APERTURE))
    return RTT_E_KIND;
  if (n == 0) return RTT_OK;
  hipLaunchKernelGGL(k_apply_physics, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, hit_local, dir,
                     normal, ld, n, prog_dev, new_dir, imult);
  return launch_status();
}

int rtt_spot_moments_f32(const float* x, const float* y, const float* w, const uint8_t* mask, int64_t n, int pass,
                         double cx, double cy, double wsum, double* out, void* stream) {
  if (!x || !y || !w || !out) return RTT_E_NULL;
  if (n < 0 || (pass != 0 && pass != 1)) return RTT_E_SIZE;
  if (misaligned(x) || misaligned(y) || misaligned(w) || (reinterpret_cast<uintptr_t>(out) & 7u)) return RTT_E_ALIGN;
  if (n == 0) return RTT_OK;
  hipLaunchKernelGGL(k_spot_moments, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, x, y, w, mask, n,
                     pass, cx, cy, wsum, out);
  return launch_status();
}

const char* rtt_strerror(int code) {
  switch (code) {
    case RTT_OK: return "ok";
    case RTT_E_NULL: return "required pointer is NULL";
    case RTT_E_SIZE: return "size, leading dimension or index out of range";
    case RTT_E_KIND: return "surface record carries an unknown or inconsistent kind";
    case RTT_E_LAUNCH: return "kernel launch failed";
    case RTT_E_ALIGN: return "pointer is not sufficiently aligned";
    default: return "unknown error";
  }
}

int rtt_abi_version(void) { return RTT_ABI_VERSION; }

int rtt_surface_record_bytes(void) { return (int)sizeof(RttSurface); }

}  // extern "C"
