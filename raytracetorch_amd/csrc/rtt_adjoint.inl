// rtt_adjoint.inl — reverse-mode backward of the sequential trace (included by rtt_kernels.hip inside its
// anonymous namespace).
//
// What torch autograd does to scene/sequential.py:12-36 of the reference when element parameters or the input rays
// carry requires_grad (reached through optim/goals.py:144-187 and tests/test_optimize_singlet.py:66-116): every lane
// re-traces its ray(s) once, keeping the state in front of every surface it hits (position, direction, path length:
// seven floats per surface in a lane-private LDS column), then walks the surfaces back and propagates the adjoint of
// the ray state through the visit's formulas in reverse.  (Keeping only direction and path length and walking the
// position back, p - t d, was measured: 5 % faster, but behind a long free path the subtraction loses the low bits of
// the position and the gradients of ill-conditioned systems moved by 1e-3.)  On the way each visit yields the adjoints of the record
// slots it read (translations, curvatures, index ratios, ... — the 32 differentiable slots of a record); the slots a
// caller asked for are summed per lane in LDS, per block in float64, and a second small kernel adds the block sums.
// The host contracts these slot adjoints with d(slot)/d(parameter), so the kernel's cost does not depend on the number
// of trainable scalars, and the adjoint that arrives at the start of the ray is the gradient with respect to the input
// ray itself.
//
// Masks, the choice between roots, `entering`, total internal reflection and the Monte-Carlo branch of RefractFresnel
// are piecewise constant and carry no derivative, as under autograd (torch.where routes by the mask, relu' and abs'
// follow the sign).  The path length is differentiated implicitly from "t solves A t^2 + B t + C = 0" with the
// reference's coefficient formulas, like the tangent kernels do.

constexpr int kAdjState = 7;  // floats kept per surface and ray: p (3), d (3), t

// Phase clocks of the profiling build (-DRTT_PHASE_TIMERS=1, `make prof`): every wavefront adds up the shader clock
// between marks (after waiting for everything it has in flight, so that a wait is charged to the phase that caused it)
// and the kernel's sums end up in g_phase[], read by rtt_debug_phases().  The product build has no marks.
#ifndef RTT_PHASE_TIMERS
#define RTT_PHASE_TIMERS 0
#endif
constexpr int kPhases = 16;
#if RTT_PHASE_TIMERS
__device__ unsigned long long g_phase[kPhases];
struct PhaseTimer {
  uint64_t prev, acc[kPhases];
  DI void start() {
#pragma unroll
    for (int k = 0; k < kPhases; ++k) acc[k] = 0;
    prev = __builtin_readcyclecounter();
  }
  template <int K> DI void mark() {
    __builtin_amdgcn_s_waitcnt(0);
    const uint64_t now = __builtin_readcyclecounter();
    acc[K] += now - prev;
    prev = now;
  }
  DI void flush() {
    if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
      for (int k = 0; k < kPhases; ++k) atomicAdd(&g_phase[k], (unsigned long long)acc[k]);
    }
  }
};
#define RTT_MARK(tm, k) (tm)->template mark<k>()
#else
struct PhaseTimer {};
#define RTT_MARK(tm, k) ((void)0)
#endif

// Which record slots of which surface get an accumulator: bit j of mask[s] = slot j (Rs 0-8, ts 9-11, Rj 12-20,
// tj 21-23, geom 24-27, phys 28-31); the accumulators of a surface are consecutive, in slot order, from base[s].
struct AccMap {
  uint32_t mask[RTT_MAX_SURFACES];
  uint16_t base[RTT_MAX_SURFACES];
};

struct AdjointIO {
  const float* adj_pos;   // [3][ld_adj] adjoint of the output positions, or null
  const float* adj_dir;   // [3][ld_adj] adjoint of the output directions, or null
  int64_t ld_adj;
  const float* adj_sens;  // [slots][3][sens_ld] adjoint of the recorded hit_local, or null
  int64_t sens_ld;
  // spot mode (spot_slot >= 0): the adjoint of hit_local on that sensor slot is formed here from the primal hit,
  //   g = (x - cx, y - cy) (w/W) / sqrt(((x-cx)^2 + (y-cy)^2) w/W)  -  (w/W) (Gx, Gy)
  // for hits with intensity w > 0 (optim/goals.py:170-183).  spot_mom (device, nullable) = the six sums of the moment
  // passes [W, sum w x, sum w y, sum sqrt, Gx, Gy]: W comes from it; with spot_centroid the centre is the centroid
  // (sum w x, sum w y) / W and (Gx, Gy) the sums of the first term over all rays, otherwise the centre is (cx, cy)
  // and (Gx, Gy) = 0.  Without spot_mom: centre (cx, cy), 1 / W = inv_w, (Gx, Gy) = 0.
  int spot_slot, spot_centroid;
  float cx, cy, inv_w;
  const double* spot_mom;
  float* ray_adj;  // [6][ld_ray] adjoint of the input rays (pos x, y, z, dir x, y, z), or null
  int64_t ld_ray;
  double* partials;  // [gridDim.x][n_acc] block sums of the slot adjoints
  int n_acc;
  // sweep from a tape (TAPE kernels): what the forward trace kept, see TapeIO; spot_w = the intensities the spot
  // sensor recorded, [n] (the reverse sweep does not follow intensities)
  const float* tape_planes;
  int64_t tape_ld;
  const uint64_t* tape_groups;
  const float* spot_w;
};

template <class T>
struct SlotSink {
  float* acc;     // this lane's column of the accumulators: accumulator a lives at acc[a * stride]
  int stride;     // threads per block
  uint32_t mask;  // wanted slots of the surface (wave-uniform)
  int base;
  typename LaneOf<T>::M on;  // components whose contribution counts: hit and part of the bundle
  bool all_on;               // `on` holds for the whole wavefront: no select needed
  PhaseTimer* tm;            // profiling build only
  DI bool wants(int slot) const { return (mask >> slot) & 1u; }
  DI bool wants_any(uint32_t slots) const { return (mask & slots) != 0u; }
  DI void add(int slot, T v) const {
    if (!wants(slot)) return;  // scalar branch
    const int a = base + __builtin_popcount(mask & ((1u << slot) - 1u));
    const float x = all_on ? hsum(v) : hsum(sel(on, v, splat<T>(0.0f)));
    // the column belongs to this lane alone: a plain read-modify-write (an LDS atomic add measured slower)
    acc[a * stride] += x;
  }
};

constexpr uint32_t kSlotsRs = 0x1ffu, kSlotsRj = 0x1ffu << 12;

// dR[3i + j] += a_i b_j for the rotation whose slots start at `first`
template <class SK, class T>
DI void sink_outer(const SK& sk, int first, V3<T> a, V3<T> b) {
  sk.add(first + 0, a.x * b.x); sk.add(first + 1, a.x * b.y); sk.add(first + 2, a.x * b.z);
  sk.add(first + 3, a.y * b.x); sk.add(first + 4, a.y * b.y); sk.add(first + 5, a.y * b.z);
  sk.add(first + 6, a.z * b.x); sk.add(first + 7, a.z * b.y); sk.add(first + 8, a.z * b.z);
}

// Adjoint of SurfaceFunction.forward: given g = adjoint of new_dir, returns the adjoints of the incoming direction
// (db) and of the normal (nb), adds to the adjoint of hit_local (hlb) and hands the slot adjoints to the sink.
template <class C, class T, class SK>
DI void physics_adjoint(const RecF& S, V3<T> hl, V3<T> d, V3<T> n, V3<T> g, V3<T>& db, V3<T>& nb, V3<T>& hlb,
                        uint64_t ridx, const SK& sk) {
  const T zero = splat<T>(0.0f), one = splat<T>(1.0f);
  db = {zero, zero, zero};
  nb = {zero, zero, zero};
  switch (C::pk(S.r)) {
    case RTT_PHYS_BLOCK:  // new_dir = 0
      return;
    case RTT_PHYS_REFLECT: {  // d - 2 (d.n) n
      const T dot = dot3(d, n), gn = dot3(g, n);
      db = g - scale(2.0f * gn, n);
      nb = scale(splat<T>(-2.0f), scale(gn, d) + scale(dot, g));
      return;
    }
    case RTT_PHYS_SNELL:
    case RTT_PHYS_FRESNEL: {
      const T dot = dot3(d, n);
      const auto ent = lt(dot, zero);
      const T c1 = abs_(dot);
      const T mu = sel(ent, splat<T>(S.phys(2)), splat<T>(S.phys(3)));
      const T term = 1.0f - mu * mu * (1.0f - c1 * c1);
      const T c2 = sqrt_(max_(term, zero));
      auto mirror = lt(term, zero);  // Snell: total internal reflection
      if (C::pk(S.r) == RTT_PHYS_FRESNEL) {  // the branch the forward took: u < R, same expressions as apply_physics
        const T n1 = sel(ent, splat<T>(S.phys(0)), splat<T>(S.phys(1)));
        const T n2 = sel(ent, splat<T>(S.phys(1)), splat<T>(S.phys(0)));
        const T sin2 = mu * mu * (1.0f - c1 * c1);
        const auto tir = gt(sin2, one);
        const T ct = sqrt_(max_(1.0f - sin2, zero));
        const T n1ci = n1 * c1, n2ct = n2 * ct, n1ct = n1 * ct, n2ci = n2 * c1;
        const T rs = quot(n1ci - n2ct, n1ci + n2ct + 1e-8f);
        const T rp = quot(n1ct - n2ci, n1ct + n2ci + 1e-8f);
        const T R = sel(tir, one, 0.5f * (rs * rs + rp * rp));
        const uint64_t key = ((uint64_t)S.r.seed[1] << 32 | S.r.seed[0]) + ((uint64_t)S.r.element_id << 8) + (uint64_t)S.r.surface_id;
        mirror = lt(fresnel_uniform(key, ridx, 0u, (T*)nullptr), R);
      }
      // refraction: mu d + sg (mu c1 - c2) n, sg = +1 entering, -1 leaving
      const T sg = sel(ent, one, splat<T>(-1.0f));
      const T gn = dot3(g, n), gd = dot3(g, d);
      const T gb = sg * gn;  // adjoint of (mu c1 - c2)
      const auto lit = gt(term, zero);
      const T termb = sel(lit, -gb * 0.5f * rcp_(c2), zero);  // c2 = sqrt(relu(term))
      T mub = gd + c1 * gb - termb * (2.0f * mu * (1.0f - c1 * c1));
      const T c1b = mu * gb + termb * (2.0f * mu * mu * c1);
      const T dotb = sel(lt(dot, zero), -c1b, sel(gt(dot, zero), c1b, zero));  // c1 = |dot|
      const T coef = sg * (mu * c1 - c2);
      db = scale(mu, g) + scale(dotb, n);
      nb = scale(coef, g) + scale(dotb, d);
      if (__any(any_(mirror))) {  // d - 2 (d.n) n
        db = sel3(mirror, g - scale(2.0f * gn, n), db);
        nb = sel3(mirror, scale(splat<T>(-2.0f), scale(gn, d) + scale(dot, g)), nb);
        mub = sel(mirror, zero, mub);
      }
      sk.add(30, sel(ent, mub, zero));
      sk.add(31, sel(ent, zero, mub));
      return;
    }
    case RTT_PHYS_LINEAR: {  // phys/std.py:72-88
      const V3<T> dl = rowmat<2>(d, S, false);
      const T rz = div_ieee(one, dl.z);
      const T qx = dl.x * rz, qy = dl.y * rz;
      const T nx = splat<T>(S.phys(0)) * hl.x + splat<T>(S.phys(2)) * qx;
      const T ny = splat<T>(S.phys(1)) * hl.y + splat<T>(S.phys(3)) * qy;
      const T rl = div_ieee(one, sqrt_ieee(nx * nx + ny * ny + 1.0f));
      const V3<T> o = {nx * rl, ny * rl, rl};
      const V3<T> ob = rowmat<2>(g, S, false);  // new_dir = o @ Rp^T
      const T oo = dot3(o, ob);
      const T nxb = (ob.x - o.x * oo) * rl, nyb = (ob.y - o.y * oo) * rl;
      sk.add(28, nxb * hl.x);
      sk.add(29, nyb * hl.y);
      sk.add(30, nxb * qx);
      sk.add(31, nyb * qy);
      hlb.x = hlb.x + nxb * S.phys(0);
      hlb.y = hlb.y + nyb * S.phys(1);
      const T qxb = nxb * S.phys(2), qyb = nyb * S.phys(3);
      const V3<T> dlb = {qxb * rz, qyb * rz, -(qxb * qx + qyb * qy) * rz};
      db = rowmat_t<2>(dlb, S, false);
      if ((S.r.flags & RTT_F_PHYS_ROT_SURF) && sk.wants_any(kSlotsRj)) {
        sink_outer(sk, 12, d, dlb);  // dl = d @ Rj
        sink_outer(sk, 12, g, o);    // new_dir_i = sum_j o_j Rj[3i + j]
      }
      return;
    }
    case RTT_PHYS_APERTURE: {  // dir * m
      const T m = sel(surface_in_bounds<C>(S, hl), one, zero);
      db = scale(m, g);
      return;
    }
    default:  // transmit
      db = g;
      return;
  }
}

// Adjoint of _getNormal in the surface frame: nlb = adjoint of the local normal nl at the local hit point h.
template <class C, class T, class SK>
DI void normal_adjoint(const RecF& S, V3<T> h, V3<T> nl, V3<T> nlb, V3<T>& hb, const SK& sk) {
  const T zero = splat<T>(0.0f);
  switch (C::sk(S.r)) {
    case RTT_SURF_PLANE:
      return;
    case RTT_SURF_SPHERE: {  // h / R
      const T rR = rcp_(splat<T>(S.geom(0)));
      hb = hb + scale(rR, nlb);
      sk.add(24, -dot3(nlb, nl) * rR);
      return;
    }
    case RTT_SURF_CYLINDER: {  // (x, y, 0) / R
      const T rR = rcp_(splat<T>(S.geom(0)));
      hb.x = hb.x + nlb.x * rR;
      hb.y = hb.y + nlb.y * rR;
      sk.add(24, -(nlb.x * nl.x + nlb.y * nl.y) * rR);
      return;
    }
    case RTT_SURF_QUADRIC:
    case RTT_SURF_QUADRIC_ZY: {  // -v / (|v| + 1e-8), v = (2c x, 2c y, 2c(1+k) z - 2)
      const bool full = C::sk(S.r) == RTT_SURF_QUADRIC;
      const float two_c = 2.0f * S.geom(0), two_c1k = 2.0f * S.geom(2);
      const T vx = full ? two_c * h.x : zero, vy = two_c * h.y, vz = two_c1k * h.z - 2.0f;
      const T L = sqrt_(vx * vx + vy * vy + vz * vz);
      const T rlen = rcp_(L + 1e-8f), rL = rcp_(L);
      const T nn = dot3(nlb, nl);
      // vb = -nlb / len + (nlb . nl) nl / L
      const V3<T> vb = scale(nn * rL, nl) - scale(rlen, nlb);
      if (full) hb.x = hb.x + two_c * vb.x;
      hb.y = hb.y + two_c * vb.y;
      hb.z = hb.z + two_c1k * vb.z;
      sk.add(24, 2.0f * ((full ? h.x * vb.x : zero) + h.y * vb.y));
      sk.add(26, 2.0f * (h.z * vb.z));
      return;
    }
    default: {  // cone: v / (|v| + 1e-8) with v = (-k2 x, -k2 y, z), +z where |v| <= 1e-8
      const float k2 = S.geom(1);
      const T vx = -k2 * h.x, vy = -k2 * h.y, vz = h.z;
      const T L = sqrt_(vx * vx + vy * vy + vz * vz);
      const auto ok = gt(L, splat<T>(1e-8f));
      const T rl = rcp_(L + 1e-8f), rL = rcp_(L);
      const T nn = dot3(nlb, nl);
      V3<T> vb = scale(rl, nlb) - scale(nn * rL, nl);
      vb = sel3(ok, vb, V3<T>{zero, zero, zero});
      hb.x = hb.x - k2 * vb.x;
      hb.y = hb.y - k2 * vb.y;
      hb.z = hb.z + vb.z;
      sk.add(25, -(h.x * vb.x + h.y * vb.y));
      return;
    }
  }
}

// Adjoint of the path length: t solves A t^2 + B t + C = 0 (coefficients of the surface kind in o = pl, d = dl), so
//   dt = -(go . d(o) + gd . d(d) + g0 d(geom[i0]) + g1 d(geom[i1])) / den
// with the same coefficient vectors as the tangent kernels; tb is the adjoint of t.
template <class C, class T, class SK>
DI void root_adjoint(const RecF& S, V3<T> o, V3<T> d, T t, V3<T> h, T tb, V3<T>& ob, V3<T>& db, const SK& sk) {
  const T zero = splat<T>(0.0f);
  const T eps = splat<T>(kEps);
  V3<T> go, gd;
  T g0 = zero, g1 = zero, den;
  int i0 = 0, i1 = -1;
  switch (C::sk(S.r)) {
    case RTT_SURF_PLANE: {  // t = -o.z / safe(d.z); the guarded denominator is a constant
      const auto guard = lt(abs_(d.z), eps);
      den = sel(guard, splat<T>(1e-8f), d.z);
      go = {zero, zero, splat<T>(1.0f)};
      gd = {zero, zero, sel(guard, zero, t)};
      i0 = -1;
      break;
    }
    case RTT_SURF_SPHERE:  // t^2 + 2 (o.d) t + o.o - R^2
      go = scale(splat<T>(2.0f), h);
      gd = scale(2.0f * t, o);
      g0 = splat<T>(-1.0f);
      i0 = 1;
      den = 2.0f * t + 2.0f * dot3(o, d);
      break;
    case RTT_SURF_CYLINDER:
      go = {2.0f * h.x, 2.0f * h.y, zero};
      gd = {t * go.x, t * go.y, zero};
      g0 = splat<T>(-1.0f);
      i0 = 1;
      den = 2.0f * (d.x * d.x + d.y * d.y) * t + 2.0f * (o.x * d.x + o.y * d.y);
      break;
    case RTT_SURF_QUADRIC:
    case RTT_SURF_QUADRIC_ZY: {
      const float c = S.geom(0), c1k = S.geom(2);
      const float kx = C::sk(S.r) == RTT_SURF_QUADRIC ? 1.0f : 0.0f;
      const T A = c * (kx * d.x * d.x + d.y * d.y) + c1k * (d.z * d.z);
      go = {2.0f * c * kx * h.x, 2.0f * c * h.y, 2.0f * c1k * h.z - 2.0f};
      i0 = 0;
      i1 = 2;
      gd = scale(t, go);
      g0 = kx * h.x * h.x + h.y * h.y;
      g1 = h.z * h.z;
      den = dot3(go, d);  // 2 A t + B: the derivative of the surface equation along the ray, grad F(h) . d
      const auto lin = lt(abs_(A), eps);  // the reference's linear root -C / B_safe: the A term is dropped
      if (__any(any_(lin))) {
        const T B = 2.0f * c * (kx * o.x * d.x + o.y * d.y) + 2.0f * c1k * o.z * d.z - 2.0f * d.z;
        const V3<T> gd_lin = {t * (2.0f * c * kx) * o.x, t * (2.0f * c) * o.y, t * (2.0f * c1k * o.z - 2.0f)};
        gd = sel3(lin, gd_lin, gd);
        g0 = sel(lin, 2.0f * t * (kx * o.x * d.x + o.y * d.y) + (kx * o.x * o.x + o.y * o.y), g0);
        g1 = sel(lin, 2.0f * t * o.z * d.z + o.z * o.z, g1);
        den = sel(lin, sel(lt(abs_(B), eps), eps, B), den);
      }
      break;
    }
    default: {  // cone: A = dz^2 - k2 (dx^2 + dy^2), B = 2 (oz dz - k2 (ox dx + oy dy)), C = oz^2 - k2 (ox^2 + oy^2)
      const float k2 = S.geom(1);
      const T A = d.z * d.z - k2 * (d.x * d.x + d.y * d.y);
      const T B = 2.0f * (o.z * d.z - k2 * (o.x * d.x + o.y * d.y));
      go = {-2.0f * k2 * h.x, -2.0f * k2 * h.y, 2.0f * h.z};
      i0 = 1;
      const auto lin = lt(abs_(A), eps);
      const V3<T> gd_lin = {-2.0f * k2 * t * o.x, -2.0f * k2 * t * o.y, 2.0f * t * o.z};
      gd = sel3(lin, gd_lin, scale(t, go));
      g0 = sel(lin, -(2.0f * t * (o.x * d.x + o.y * d.y) + (o.x * o.x + o.y * o.y)), -(h.x * h.x + h.y * h.y));
      den = sel(lin, sel(lt(abs_(B), eps), eps, B), 2.0f * A * t + B);
      break;
    }
  }
  const T s = -tb * rcp_(den);
  ob = ob + scale(s, go);
  db = db + scale(s, gd);
  if (i0 >= 0) sk.add(24 + i0, s * g0);
  if (i1 >= 0) sk.add(24 + i1, s * g1);
}

// Adjoint of one surface visit for the components where `hit`: (p, d, t) are the state in front of the surface and the
// path length the forward sweep kept; pb / db come in as the adjoints of the state behind the surface and leave as
// the adjoints of the state in front of it.  `ext(hl)` supplies the adjoint that reaches hit_local from outside the
// trace (a sensor's record), evaluated at the recomputed local hit point.
template <class C, class T, class M, class SK, class Ext>
DI void visit_adjoint(const RecF& S, V3<T> p, V3<T> d, T t, M hit, V3<T>& pb, V3<T>& db, uint64_t ridx,
                      const SK& sk, Ext&& ext) {
  const T zero = splat<T>(0.0f);
  const bool shp = C::has_shape(S.r), ids = C::rs_id(S.r), idj = C::rj_id(S.r);
  // ---- the forward quantities again (to_frames, surface_forward) ----
  V3<T> q = p, pe = p, de = d, den = d;
  T rinv = splat<T>(1.0f);
  auto norm_live = mask_all<M>(true);
  if (shp) {
    q = p - V3<T>{splat<T>(S.ts(0)), splat<T>(S.ts(1)), splat<T>(S.ts(2))};
    pe = rowmat<0>(q, S, ids);
    de = rowmat<0>(d, S, ids);
    const T n2 = dot3(de, de);
    if (kStrict) {
      rinv = div_ieee(splat<T>(1.0f), max_(sqrt_ieee(n2), splat<T>(1e-12f)));
    } else {
      rinv = rsq_(max_(n2, splat<T>(1e-24f)));
    }
    norm_live = gt(n2, splat<T>(1e-24f));  // below the clamp the divisor is a constant
    den = scale(rinv, de);
  }
  const V3<T> u = pe - V3<T>{splat<T>(S.tj(0)), splat<T>(S.tj(1)), splat<T>(S.tj(2))};
  const V3<T> pl = rowmat<1>(u, S, idj);
  const V3<T> dl = rowmat<1>(den, S, idj);
  const V3<T> hl = pl + scale(t, dl);
  const V3<T> nl = surface_normal<C>(S, hl);
  const V3<T> n1 = rowmat_t<1>(nl, S, idj);
  const V3<T> n = shp ? rowmat_t<0>(n1, S, ids) : n1;
  RTT_MARK(sk.tm, 4);
  // ---- physics: new_dir(hl, d, n) ----
  V3<T> hlb = ext(hl);
  V3<T> dpb, nb;
  physics_adjoint<C>(S, hl, d, n, db, dpb, nb, hlb, ridx, sk);
  RTT_MARK(sk.tm, 5);
  // ---- normal: n = (nl @ Rj^T) @ Rs^T ----
  const V3<T> n1b = shp ? rowmat<0>(nb, S, ids) : nb;
  const V3<T> nlb = rowmat<1>(n1b, S, idj);
  normal_adjoint<C>(S, hl, nl, nlb, hlb, sk);
  RTT_MARK(sk.tm, 6);
  // ---- hit_local = pl + t dl, hit_global = p + t d ----
  V3<T> plb = hlb, dlb = scale(t, hlb);
  const T tb = dot3(hlb, dl) + dot3(pb, d);
  V3<T> p_in = pb;
  V3<T> d_in = dpb + scale(t, pb);
  root_adjoint<C>(S, pl, dl, t, hl, tb, plb, dlb, sk);
  RTT_MARK(sk.tm, 7);
  // ---- frames: pl = (pe - tj) @ Rj, dl = den @ Rj, den = de / |de|, pe = (p - ts) @ Rs, de = d @ Rs ----
  const V3<T> ub = rowmat_t<1>(plb, S, idj);
  const V3<T> denb = rowmat_t<1>(dlb, S, idj);
  sk.add(21, -ub.x);
  sk.add(22, -ub.y);
  sk.add(23, -ub.z);
  V3<T> deb = denb;
  if (shp) {
    const T along = dot3(den, denb);
    deb = denb - scale(along, den);
    if (!__all(all_(norm_live))) deb = sel3(norm_live, deb, denb);
    deb = scale(rinv, deb);
    const V3<T> qb = rowmat_t<0>(ub, S, ids);
    p_in = p_in + qb;
    sk.add(9, -qb.x);
    sk.add(10, -qb.y);
    sk.add(11, -qb.z);
    d_in = d_in + rowmat_t<0>(deb, S, ids);
    if (sk.wants_any(kSlotsRs)) {
      sink_outer(sk, 0, q, ub);   // pe = q @ Rs
      sink_outer(sk, 0, d, deb);  // de = d @ Rs
      sink_outer(sk, 0, nb, n1);  // n_i = sum_j n1_j Rs[3i + j]
    }
  } else {
    p_in = p_in + ub;
    d_in = d_in + denb;
  }
  if (sk.wants_any(kSlotsRj)) {
    sink_outer(sk, 12, u, plb);
    sink_outer(sk, 12, den, dlb);
    sink_outer(sk, 12, n1b, nl);  // n1_i = sum_j nl_j Rj[3i + j]
  }
  if (sk.all_on) {  // every component of the wavefront hit: plain moves
    pb = p_in;
    db = d_in;
  } else {
    pb = sel3(hit, p_in, pb);
    db = sel3(hit, d_in, db);
  }
  RTT_MARK(sk.tm, 8);
  (void)zero;
}

template <class T> struct HitBits { uint64_t b[LaneOf<T>::N]; };

template <class T, class M>
DI void set_bits(HitBits<T>& hb, int s, M hit) {
#pragma unroll
  for (int k = 0; k < LaneOf<T>::N; ++k) hb.b[k] |= (uint64_t)(mcomp(hit, k) ? 1u : 0u) << s;
}
template <class T>
DI typename LaneOf<T>::M get_bits(const HitBits<T>& hb, int s) {
  auto m = mask_all<typename LaneOf<T>::M>(false);
#pragma unroll
  for (int k = 0; k < LaneOf<T>::N; ++k) m = set_mask(m, k, ((hb.b[k] >> s) & 1ull) != 0);
  return m;
}

template <class I> DI float ldp_(const float* p, I i, bool, int, float*) { return p[i]; }
template <class I> DI f2 ldp_(const float* p, I i, bool vec, int nv, f2*) {
  if (vec && nv == 2) return *reinterpret_cast<const f2*>(p + i);
  return f2{p[i], p[i + (nv - 1)]};
}

// a 64-bit value that every lane of the wavefront holds alike, moved to scalar registers
DI uint64_t uniform64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// f(S), f(S + 1), ..., f(E - 1) with the index a compile-time constant in every call
template <int S, int E, class F>
DI void unrolled(F&& f) {
  if constexpr (S < E) {
    f(S);
    unrolled<S + 1, E>(f);
  }
}
// Where the forward sweep keeps the state in front of every surface.
// LDS (NS == 0): a lane-private column of dynamic LDS, any number of surfaces, rolled loops.  Per surface either the
// full state (position, direction, path length: 7 values) or, COMPACT, direction and path length only (4 values) next
// to the position the ray started from (3 values, once): the position in front of surface s is then made again by
// walking p <- p + t d over the surfaces before s, the very operations of the forward sweep, so it has the same bits.
// (Walking back from the end, p - t d, does not: see the head of this file.)  The walk makes the reverse sweep
// quadratic in the number of surfaces; it is used for short programs, where the smaller footprint buys a third
// wavefront per SIMD.
constexpr int kAdjStateCompact = 4, kAdjStart = 3;
constexpr int kAdjCompactSurfaces = 6;  // longest program swept with compact states (the walk is quadratic)
template <class T, int BS, bool COMPACT>
struct AdjState {
  static constexpr int kPer = COMPACT ? kAdjStateCompact : kAdjState;
  T* base;
  DI AdjState(float* lds_dyn, int n_acc) : base(reinterpret_cast<T*>(lds_dyn + (int64_t)n_acc * BS) + threadIdx.x) {}
  DI T* of(int s) const { return base + ((int64_t)s * kPer + (COMPACT ? kAdjStart : 0)) * BS; }
  DI void put_start(V3<T> p) {
    base[0] = p.x; base[BS] = p.y; base[2 * BS] = p.z;
  }
  DI V3<T> get_start() const { return {base[0], base[BS], base[2 * BS]}; }
  DI void put(int s, V3<T> p, V3<T> d, T t) {
    T* st = of(s);
    if (COMPACT) {
      st[0] = d.x; st[BS] = d.y; st[2 * BS] = d.z; st[3 * BS] = t;
    } else {
      st[0] = p.x; st[BS] = p.y; st[2 * BS] = p.z;
      st[3 * BS] = d.x; st[4 * BS] = d.y; st[5 * BS] = d.z; st[6 * BS] = t;
    }
  }
  DI void get_dir(int s, V3<T>& d, T& t) const {
    const T* st = of(s);
    d = {st[0], st[BS], st[2 * BS]};
    t = st[3 * BS];
  }
  DI void get(int s, V3<T>& p, V3<T>& d, T& t) const {
    const T* st = of(s);
    p = {st[0], st[BS], st[2 * BS]};
    d = {st[3 * BS], st[4 * BS], st[5 * BS]};
    t = st[6 * BS];
  }
};
// Dynamic LDS = float acc[n_acc][BS], then T state[n_surf][7][BS] or, COMPACT, T start[3][BS] and
//   T state[n_surf][4][BS]; BS threads per block.  With two rays per lane a four-surface program gets two wavefronts
//   per SIMD with the full states and three with the compact ones.
// TAPE (with COMPACT): there is no forward sweep; the states come from the tape the forward trace wrote (direction and
//   path length per surface that the ray's group hit), staged into the same LDS layout, and the positions are walked
//   from the input positions as above.  io.dir_in and io.inten_in are not read.
// SET = 1: the program holds nothing but spherical lens faces, lens edges and bounded planes (a lens system in front of a
// sensor or a stop); the visits of the other configurations are left out of this instantiation.  SET = 2: every record
// matches a specialised visit; the generic interpreter's visit is left out.
#define RTT_DISPATCH_SET(op, CALL)                  \
  if constexpr (SET == 1) {                         \
    switch (op) {                                   \
      case OP_FACE_SNELL: CALL(CFaceSnell); break;  \
      case OP_EDGE_BLOCK: CALL(CEdgeBlock); break;  \
      case OP_EDGE_SNELL: CALL(CEdgeSnell); break;  \
      default: CALL(CDiskPlane); break;             \
    }                                               \
  } else if constexpr (SET == 2) {                  \
    constexpr int NG = 1;                           \
    RTT_DISPATCH_NG(op, CALL)                       \
  } else {                                          \
    RTT_DISPATCH(op, CALL)                          \
  }

template <class T, int BS, bool COMPACT = false, bool TAPE = false, int SET = 0>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(COMPACT ? 3 : 2, 4))) k_trace_seq_adjoint(RayIO io, const RttSurface* __restrict__ prog_, int n_surf,
                                                              Ops ops, AccMap am, AdjointIO a, int vec) {
  static_assert(!TAPE || (COMPACT && LaneOf<T>::N == 2 && BS % 64 == 0), "a tape is swept with compact states, two rays per lane");
  extern __shared__ float lds_dyn[];
  constexpr int N = LaneOf<T>::N;
  using M = typename LaneOf<T>::M;
  using State = AdjState<T, BS, COMPACT>;
  using Sink = SlotSink<T>;
  const RttSurface* __restrict__ prog = (const RttSurface*)__builtin_assume_aligned(prog_, 256);
  float* acc = lds_dyn + threadIdx.x;
  PhaseTimer timer, *tm = &timer;
#if RTT_PHASE_TIMERS
  timer.start();
#endif
  (void)tm;
  for (int k = 0; k < a.n_acc; ++k) acc[k * BS] = 0.0f;
  float Gx = 0.0f, Gy = 0.0f, cx = a.cx, cy = a.cy, inv_w = a.inv_w;
  if (a.spot_slot >= 0 && a.spot_mom) {
    const double W = fmax(a.spot_mom[0], 1e-12);  // the reference clamps the weight sum (optim/goals.py:173)
    inv_w = (float)(1.0 / W);
    if (a.spot_centroid) {
      cx = (float)(a.spot_mom[1] / W);
      cy = (float)(a.spot_mom[2] / W);
      Gx = (float)a.spot_mom[4];
      Gy = (float)a.spot_mom[5];
    }
  }
  const int64_t tile = (int64_t)BS * N;
  const int64_t stride = (int64_t)gridDim.x * tile;
  uint64_t group_mask = 0ull;  // TAPE: surfaces the group of the next tile hit (wave-uniform, kept in scalar registers)
  if constexpr (TAPE) {
    const int64_t first = (int64_t)blockIdx.x * tile + (int64_t)(threadIdx.x & ~63u) * N;
    if (first < io.n) group_mask = uniform64(a.tape_groups[first / kTapeGroup]);
  }
  for (int64_t base = (int64_t)blockIdx.x * tile; base < io.n; base += stride) {
    const RayIO tio = {io.pos_in + base, io.dir_in + base, io.inten_in + base, io.ld_in, nullptr, nullptr, nullptr, 0, io.n - base};
    const uint32_t idx = threadIdx.x * N;
    const int64_t left = tio.n - (int64_t)idx;
    const int nv = left >= N ? N : (left >= 1 ? 1 : 0);
    // lanes past the end trace a copy of the tile's first ray and contribute nothing
    Ray<T> r;
    if constexpr (TAPE) {
      T* tagp = nullptr;
      const uint32_t ati = nv ? idx : 0u;
      const bool vl = vec != 0 && nv == N;
      r.p = {ld_(tio.pos_in, ati, vl, nv ? nv : 1, tagp), ld_(tio.pos_in + tio.ld_in, ati, vl, nv ? nv : 1, tagp),
             ld_(tio.pos_in + 2 * tio.ld_in, ati, vl, nv ? nv : 1, tagp)};
    } else {
      r = load_ray<T>(tio, nv ? idx : 0u, vec != 0 && nv == N, nv ? nv : 1);
    }
    M live = mask_all<M>(false);
#pragma unroll
    for (int k = 0; k < N; ++k) live = set_mask(live, k, k < nv);
    RTT_MARK(tm, 0);  // ray loads
    const uint64_t ridx = (uint64_t)(base + idx);
    HitBits<T> bits;
#pragma unroll
    for (int k = 0; k < N; ++k) bits.b[k] = 0ull;
    T w_spot = splat<T>(0.0f);
    State state(lds_dyn, a.n_acc);
    uint64_t any_hit = 0ull, all_hit = 0ull;  // COMPACT: surfaces that some / every ray of the wavefront hit (wave-uniform)
    if constexpr (COMPACT) state.put_start(r.p);
    // ---- forward sweep: keep the state in front of every surface that is hit ----
    auto forward_visit = [&](const int s) __attribute__((always_inline)) {
      const RecF S{prog[s]};
      Frames<T> f;
      T t_raw;
      M hit;
#define RTT_CALL(CFG)                                                                                  \
  {                                                                                                    \
    hit = test_surface<CFG>(S, r.p, r.d, f, t_raw);                                                    \
    if (__any(any_(hit))) {                                                                            \
      state.put(s, r.p, r.d, t_raw);                                                                   \
      if (COMPACT) any_hit |= 1ull << s;                                                               \
      if (COMPACT && __all(all_(hit))) all_hit |= 1ull << s;                                           \
      const Forward<T> fw = surface_forward<CFG>(S, r.p, r.d, f, t_raw);                               \
      V3<T> nd;                                                                                        \
      T im;                                                                                            \
      apply_physics<CFG>(S, fw.hit_l, r.d, fw.n_g, nd, im, ridx, 0u);                                  \
      if ((S.r.flags & RTT_F_SENSOR) && S.r.sensor_slot == a.spot_slot) w_spot = sel(hit, r.I, w_spot); \
      if (__all(all_(hit))) {                                                                          \
        r.p = fw.hit_g;                                                                                \
        r.d = nd;                                                                                      \
        r.I = r.I * im;                                                                                \
      } else {                                                                                         \
        r.p = sel3(hit, fw.hit_g, r.p);                                                                \
        r.d = sel3(hit, nd, r.d);                                                                      \
        r.I = sel(hit, r.I * im, r.I);                                                                 \
      }                                                                                                \
    }                                                                                                  \
  }
      RTT_DISPATCH_SET(op_of(ops, s), RTT_CALL)
#undef RTT_CALL
      set_bits(bits, s, hit);
    };
    if constexpr (TAPE) {
      // ---- no forward sweep: stage the tape of this wavefront's group ----
      // All loads of a tile are issued before the first one is waited for (one trip to memory per tile, not one per
      // surface), and the group's surface mask was asked for a tile ahead.
      any_hit = group_mask;
      {
        const int64_t next_first = base + stride + (int64_t)(threadIdx.x & ~63u) * N;
        group_mask = next_first < io.n ? uniform64(a.tape_groups[next_first / kTapeGroup]) : 0ull;
      }
      T* tagp = nullptr;
      const uint32_t ati = nv ? idx : 0u;
      const bool vl = vec != 0 && nv == N;
      if (a.spot_slot >= 0 && a.spot_w) w_spot = ld_(a.spot_w + base, ati, vl, nv ? nv : 1, tagp);
      constexpr int kBatch = 3;  // surfaces whose planes are in flight together (their values wait in registers)
      uint64_t left_s = any_hit;
      while (left_s != 0ull) {
        int sb[kBatch];
        V3<T> dq[kBatch];
        T tq[kBatch];
        unrolled<0, kBatch>([&](const int j) __attribute__((always_inline)) {
          sb[j] = left_s != 0ull ? __builtin_ctzll(left_s) : -1;
          left_s &= left_s - 1ull;
          if (sb[j] < 0) return;
          const float* tp = a.tape_planes + (int64_t)4 * sb[j] * a.tape_ld + base;
          dq[j] = {ld_(tp, ati, vl, nv ? nv : 1, tagp), ld_(tp + a.tape_ld, ati, vl, nv ? nv : 1, tagp),
                   ld_(tp + 2 * a.tape_ld, ati, vl, nv ? nv : 1, tagp)};
          tq[j] = ld_(tp + 3 * a.tape_ld, ati, vl, nv ? nv : 1, tagp);
        });
        unrolled<0, kBatch>([&](const int j) __attribute__((always_inline)) {
          if (sb[j] < 0) return;
          const M hit = and_(gt(tq[j], splat<T>(0.0f)), live);   // a ray that missed carries t = -1
          state.put(sb[j], r.p, dq[j], tq[j]);
          if (__all(all_(hit))) all_hit |= 1ull << sb[j];
          set_bits(bits, sb[j], hit);
        });
      }
    } else {
      for (int s = 0; s < n_surf; ++s) forward_visit(s);
    }
    RTT_MARK(tm, 1);  // forward sweep (or the staging of the tape)
    // ---- adjoint of the final state ----
    T* tag = nullptr;
    const T zero = splat<T>(0.0f);
    V3<T> pb = {zero, zero, zero}, db = {zero, zero, zero};
    const uint32_t at = nv ? idx : 0u;
    const bool vload = vec != 0 && nv == N && (a.ld_adj % 2 == 0);
    if (a.adj_pos) {
      const float* ap = a.adj_pos + base;
      pb = {ldp_(ap, at, vload, nv ? nv : 1, tag), ldp_(ap + a.ld_adj, at, vload, nv ? nv : 1, tag),
            ldp_(ap + 2 * a.ld_adj, at, vload, nv ? nv : 1, tag)};
    }
    if (a.adj_dir) {
      const float* ad = a.adj_dir + base;
      db = {ldp_(ad, at, vload, nv ? nv : 1, tag), ldp_(ad + a.ld_adj, at, vload, nv ? nv : 1, tag),
            ldp_(ad + 2 * a.ld_adj, at, vload, nv ? nv : 1, tag)};
    }
    // ---- reverse sweep ----
    auto reverse_visit = [&](const int s) __attribute__((always_inline)) {
      const M hit = get_bits(bits, s);
      if (!__any(any_(hit))) return;
      const RecF S{prog[s]};
      V3<T> p, d;
      T t;
      if constexpr (COMPACT) {
        p = state.get_start();
        for (int j = 0; j < s; ++j) {  // the forward sweep's moves again, on the surfaces in front of s
          if (!((any_hit >> j) & 1ull)) continue;
          V3<T> dj;
          T tj;
          state.get_dir(j, dj, tj);
          const V3<T> moved = p + scale(tj, dj);
          p = ((all_hit >> j) & 1ull) ? moved : sel3(get_bits(bits, j), moved, p);
        }
        state.get_dir(s, d, t);
      } else {
        state.get(s, p, d, t);
      }
      const M on = and_(hit, live);
      const bool all_on = __all(all_(on));
      const Sink sk = {acc, BS, am.mask[s], (int)am.base[s], on, all_on, tm};
      RTT_MARK(tm, 2);  // upstream adjoints, state of the visit, positions walked
      const bool sensor = (S.r.flags & RTT_F_SENSOR) != 0;
      const int slot = S.r.sensor_slot;
      auto ext = [&](V3<T> hl) -> V3<T> {
        V3<T> g = {zero, zero, zero};
        if (!sensor) return g;
        if (a.spot_slot >= 0) {
          if (slot != a.spot_slot) return g;
          const T wn = w_spot * inv_w;
          const T dx = hl.x - cx, dy = hl.y - cy;
          const T uu = sqrt_ieee((dx * dx + dy * dy) * wn);
          const T inv = sel(gt(uu, zero), div_ieee(wn, uu), zero);
          const auto on = and_(gt(w_spot, zero), hit);  // (a recorded intensity exists only where the sensor was hit)
          g.x = sel(on, dx * inv - wn * Gx, zero);
          g.y = sel(on, dy * inv - wn * Gy, zero);
          return g;
        }
        if (a.adj_sens) {
          const float* as = a.adj_sens + (int64_t)slot * 3 * a.sens_ld + base;
          const bool vs = vec != 0 && nv == N && (a.sens_ld % 2 == 0);
          g = {ldp_(as, at, vs, nv ? nv : 1, tag), ldp_(as + a.sens_ld, at, vs, nv ? nv : 1, tag),
               ldp_(as + 2 * a.sens_ld, at, vs, nv ? nv : 1, tag)};
          // entries of rays that did not hit the sensor are not part of the record
          g = sel3(hit, g, V3<T>{zero, zero, zero});
        }
        return g;
      };
      RTT_MARK(tm, 3);  // record flags, dispatch
#define RTT_CALL(CFG) visit_adjoint<CFG>(S, p, d, t, hit, pb, db, ridx, sk, ext)
      RTT_DISPATCH_SET(op_of(ops, s), RTT_CALL)
#undef RTT_CALL
    };
    for (int s = n_surf - 1; s >= 0; --s) reverse_visit(s);
    if (a.ray_adj && nv) {
      float* ra = a.ray_adj + base;
      const T comps[6] = {pb.x, pb.y, pb.z, db.x, db.y, db.z};
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        float* plane = ra + (int64_t)c * a.ld_ray;
#pragma unroll
        for (int k = 0; k < N; ++k)
          if (k < nv) plane[idx + k] = comp(comps[c], k);
      }
    }
  }
  RTT_MARK(tm, 9);
#if RTT_PHASE_TIMERS
  timer.flush();
#endif
  // ---- block sums of the slot adjoints, float64 ----
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  for (int k = wv; k < a.n_acc; k += BS / 64) {
    const float* col = lds_dyn + (int64_t)k * BS;
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < BS / 64; ++j) v += (double)col[lane + 64 * j];
    v = wave_sum(v);
    if (lane == 0) a.partials[(int64_t)blockIdx.x * a.n_acc + k] = v;  // read back by k_sum_partials with strides (n_acc, 1)
  }
}

// out[k] = sum over blocks b of partials[b * stride_b + k * stride_k]; one block per accumulator, fixed order: the
// result does not depend on the order in which the blocks of the first kernel finished
__global__ void __launch_bounds__(kBlock) k_sum_partials(const double* __restrict__ partials, int n_blocks, int64_t stride_b,
                                                         int64_t stride_k, double* __restrict__ out) {
  __shared__ double part[kBlock / 64];
  const int k = blockIdx.x;
  double v = 0.0;
  for (int b = threadIdx.x; b < n_blocks; b += kBlock) v += partials[(int64_t)b * stride_b + (int64_t)k * stride_k];
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) part[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) s += part[w];
    out[k] = s;
  }
}
