/*
 * rtt_hip.h — C ABI of librtt_hip.so, the MI355X (gfx950) implementation of the RayTraceTorch
 * batched ray-propagation hot path.
 *
 * The upstream project is pure Python on PyTorch and has no FFI of its own; every entry point
 * below replaces a *Python* call chain of the reference, cited as file:line relative to the
 * reference tree.  All pointers except `prog` records passed by the host packer are DEVICE
 * pointers owned by the caller.  No entry point allocates, frees or synchronises; work is
 * enqueued on `stream` (a hipStream_t passed as void*).  Return value: 0 on success, a negative
 * RTT_E_* code otherwise (see rtt_strerror).
 *
 * Ray storage is structure-of-arrays: a "plane triple" pointer `p` with leading dimension `ld`
 * addresses component k of ray i at p[k*ld + i]  (k = 0,1,2 for x,y,z).
 */
#ifndef RTT_HIP_H
#define RTT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTT_ABI_VERSION 6
#define RTT_MAX_SURFACES 64 /* upper bound on n_surf accepted by every entry point */

/* error codes */
#define RTT_OK 0
#define RTT_E_NULL (-1)      /* a required pointer is NULL */
#define RTT_E_SIZE (-2)      /* n < 0, n_surf out of range, ld < n, index out of range */
#define RTT_E_KIND (-3)      /* a surface record carries an unknown kind */
#define RTT_E_LAUNCH (-4)    /* hipLaunchKernel reported an error */
#define RTT_E_ALIGN (-5)     /* a pointer is not 4-byte aligned */

/* surface geometry kinds (geom/primitives.py:119-497) */
enum {
  RTT_SURF_PLANE = 0,      /* z = 0                                   primitives.py:119-143 */
  RTT_SURF_QUADRIC = 1,    /* c(x^2+y^2)+c(1+k)z^2-2z = 0             primitives.py:244-343 */
  RTT_SURF_QUADRIC_ZY = 2, /* c y^2 + c(1+k) z^2 - 2z = 0             primitives.py:346-395 */
  RTT_SURF_CYLINDER = 3,   /* x^2+y^2 = R^2                           primitives.py:190-241 */
  RTT_SURF_SPHERE = 4,     /* |p|^2 = R^2                             primitives.py:146-187 */
  RTT_SURF_CONE = 5        /* z^2 - s^2 (x^2+y^2) = 0                 primitives.py:398-497 */
};

/* surface-level bounds, evaluated in the surface frame at both roots (geom/bounded.py:20-36) */
enum {
  RTT_BND_NONE = 0,        /* plain Surface: only the t <= 1e-6 filter primitives.py:28-36 */
  RTT_BND_DISK = 1,        /* x^2+y^2 <= bnd[0]                       bounded.py:60-64   */
  RTT_BND_RECT = 2,        /* |x|<=bnd[0] & |y|<=bnd[1]               bounded.py:77-82   */
  RTT_BND_ELLIPSE = 3,     /* bnd = cos, sin, r_major, r_minor        bounded.py:98-106  */
  RTT_BND_HALF = 4,        /* |z*c| < 1+1e-6                          bounded.py:123-127,171-174 */
  RTT_BND_HALF_DISK = 5,   /* HALF & x^2+y^2 <= bnd[0]                bounded.py:151-159 */
  RTT_BND_SINGLE_CONE = 6, /* z*slope >= -1e-6                        bounded.py:208-217 */
  RTT_BND_ALWAYS = 7       /* bounded surface whose inBounds is constant true */
};

/* shape-level bounds, evaluated in the element frame (geom/shape.py:46-55) */
enum {
  RTT_SHP_NONE = 0,         /* element.shape is a bare Surface                              */
  RTT_SHP_SPHERIC_FACE = 1, /* x^2+y^2 <= shp[0]                       spherics.py:40-46    */
  RTT_SHP_SPHERIC_EDGE = 2, /* shp[0] <= z <= shp[1]                   spherics.py:34-39    */
  RTT_SHP_CYL_FACE = 3,     /* box aperture shp[0..3]                  cylindrics.py:31-42  */
  RTT_SHP_CYL_EDGE = 4,     /* box & sag(y) limits, shp[4..7]=c0,tz0,c1,tz1 cylindrics.py:44-55 */
  RTT_SHP_POLYHEDRON = 5    /* CvxPolyhedron: n_q . (hit - t_q) < 1e-4 for every other plane q of the element,
                               n_q = row 2 of Rj_q, t_q = tj_q, read from the element's own records: the
                               shp[0] records of one polyhedron are consecutive in the program and ordered by
                               surface_id                                   shape.py:104-132 */
};

/* surface physics (phys/std.py, phys/filter.py) */
enum {
  RTT_PHYS_TRANSMIT = 0, /* std.py:221-235 */
  RTT_PHYS_BLOCK = 1,    /* std.py:237-254 */
  RTT_PHYS_REFLECT = 2,  /* std.py:91-108  */
  RTT_PHYS_SNELL = 3,    /* std.py:111-145; phys[0]=ior_in, phys[1]=ior_out,
                            phys[2]=ior_out/ior_in, phys[3]=ior_in/ior_out */
  RTT_PHYS_APERTURE = 4, /* filter.py:10-33: dir*=m, imult=m, m = surface bound without invert */
  RTT_PHYS_LINEAR = 5,   /* std.py:35-88: paraxial element; phys = Cx, Cy, Dx, Dy; the functor's rotation is
                            phys_R, or Rj when RTT_F_PHYS_ROT_SURF is set (elements/ideal.py:47-55) */
  RTT_PHYS_FRESNEL = 6   /* std.py:148-219: Snell refraction or mirror reflection, chosen per ray by
                            u < R(Fresnel); phys as RTT_PHYS_SNELL.  u is word 2*(i&1) of the Philox4x32-10
                            block with counter (i >> 1, 0, b) and key seed + surface position in the program,
                            u = (x >> 8) * 2^-24, i = index of the ray in the call, b = bounce of the launch (0 in
                            the sequential trace), so that a ray meeting the surface again draws a new number */
};

/* flags */
#define RTT_F_HAS_SHAPE 1    /* two-level transform with re-normalisation (geom/shape.py:37-38) */
#define RTT_F_SHAPE_ROT_ID 2 /* Rs is exactly the identity */
#define RTT_F_SURF_ROT_ID 4  /* Rj is exactly the identity */
#define RTT_F_INVERT 8       /* SurfaceBounded.invert (geom/bounded.py:29-30) */
#define RTT_F_SENSOR 16      /* Sensor element: record hit_local, intensity (elements/sensor.py:35-37) */
#define RTT_F_PHYS_ROT_SURF 32 /* RTT_PHYS_LINEAR rotates with the surface transform Rj (LinearElement) */

/* One record per (element, surface), 256 bytes, packed on the host from the live parameters. */
typedef struct RttSurface {
  int32_t surf_kind, bound_kind, shape_kind, phys_kind;
  int32_t flags, element_id, surface_id, sensor_slot;
  float Rs[9], ts[3]; /* element (Shape) transform: local = (p - ts) @ Rs   geom/transform.py:90-93 */
  float Rj[9], tj[3]; /* surface transform inside the element frame                              */
  float geom[4];      /* quadrics: c, k, c*(1+k), 0 | cylinder/sphere: R, R*R | cone: s, s*s     */
  float bnd[4];
  float shp[8];
  float phys[4];
  float phys_R[9];    /* rotation of a Linear functor that carries its own transform (phys/std.py:51-54) */
  uint32_t seed[2];   /* Philox key (lo, hi) of RTT_PHYS_FRESNEL, drawn by the host for every trace */
  float pad[1];
} RttSurface;

/* ---- fused sequential trace: SequentialScene.simulate (scene/sequential.py:12-36) --------------
 * For s = 0..n_surf-1 in order: test surface s (Element.intersectTest column, elements/parent.py:30-42,
 * geom/shape.py:25-59), and for rays with t < inf apply Element.forward (elements/parent.py:44-58)
 * and Rays.scatter_update (rays/ray.py:29-40).  In-place operation (out == in) is allowed.
 * Sensor surfaces (RTT_F_SENSOR) additionally write, for slot k = sensor_slot, hit_local to
 * sens_hit[(3*k+c)*sens_ld + i], the pre-update intensity to sens_w[k*sens_ld + i] and 1/0 to
 * sens_mask[k*sens_ld + i]; all three may be NULL when no surface carries RTT_F_SENSOR. */
int rtt_trace_seq_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                      float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                      const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                      float* sens_hit, float* sens_w, uint8_t* sens_mask, int64_t sens_ld,
                      void* stream);

/* The same trace, and beside it the first spot-size moments of one sensor slot (optim/goals.py:165-178 right behind
 * scene/sequential.py:12-36): mom[0..2] = sum w, sum w x, sum w y over the hits recorded on slot mom_slot with intensity
 * w > 0, overwritten.  The sums are formed while the hit is in registers (float32 per wavefront, float64 from there on),
 * which saves the pass of rtt_spot_moments_f32 (pass 0) over the record.  workspace: device scratch of
 * rtt_trace_seq_moments_workspace_bytes(n) bytes (24 KB: 1024 bins per sum that the blocks add into with float64
 * atomics; cleared by the call).  The sensor buffers and prog_host are required. 
 * pos_out, dir_out and inten_out may all three be NULL: the traced rays are then not written (a loss that only reads
 * the sensor saves 28 B per ray). */
int64_t rtt_trace_seq_moments_workspace_bytes(int64_t n);
int rtt_trace_seq_moments_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                              float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                              const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                              float* sens_hit, float* sens_w, uint8_t* sens_mask, int64_t sens_ld,
                              int mom_slot, double* mom, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- the same trace with per-surface position snapshots: SequentialScene.simulate on rays wrapped in Paths
 * (rays/ray.py:100-207; Paths.scatter_update :174-185 snapshots every ray's position after each surface that
 * was hit by at least one ray).  hist_pos[(3*s + c)*hist_ld + i] receives component c of ray i after surface s,
 * hist_any[s] (zeroed by the caller) is set to 1 when some ray hit surface s.  A visualisation aid: one ray per
 * lane, not tuned. */
int rtt_trace_seq_history_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                              float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                              const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                              float* sens_hit, float* sens_w, uint8_t* sens_mask, int64_t sens_ld,
                              float* hist_pos, int32_t* hist_any, int64_t hist_ld, void* stream);

/* ---- backward of the sequential trace w.r.t. scene parameters (what torch autograd does to
 * scene/sequential.py:12-36 when c, k, trans, rot_vec, ior_* carry requires_grad; reached in the
 * reference through optim/goals.py:144-187 and tests/test_optimize_singlet.py:66-116).
 * Forward-mode: each ray is re-traced with n_theta (2, 4, 6 or 8) parameter directions; tan_dev is
 * [n_surf][32][n_theta] = d(record slot)/d(theta_k) for the slots Rs[9], ts[3], Rj[9], tj[3], geom[4],
 * phys[4] in that order.  The per-ray tangents are contracted with the upstream adjoints of the output
 * rays (adj_pos, adj_dir: plane triples, nullable) and of the recorded sensor hit points
 * (adj_sens[(3*slot+c)*sens_ld + i], nullable) and summed into dtheta[n_theta] (float64, device,
 * zeroed by the caller). */
int rtt_trace_seq_tangent_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, int64_t n,
                              const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                              const float* tan_dev, int n_theta, const float* adj_pos, const float* adj_dir,
                              int64_t ld_adj, const float* adj_sens, int64_t sens_ld, double* dtheta, void* stream);

/* The same backward w.r.t. the input rays themselves (torch autograd follows rays.pos / rays.dir back to whatever
 * made them: a trainable bundle pose geom/transform.py:247-264, an earlier trace): the six directions are the
 * components of each ray's own input, and the contraction with the upstream adjoints is stored per ray instead of
 * summed: ray_adj[k*ld_ray + i] = d(objective)/d(component k of ray i), k = pos x, y, z, dir x, y, z.
 * zero_tan_dev: [n_surf][32][6] zeros in device memory (the scene parameters do not move here). */
int rtt_trace_seq_ray_adjoint_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, int64_t n,
                                  const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                                  const float* zero_tan_dev, const float* adj_pos, const float* adj_dir, int64_t ld_adj,
                                  const float* adj_sens, int64_t sens_ld, float* ray_adj, int64_t ld_ray, void* stream);

/* ---- fused backward of the spot-size loss (optim/goals.py:144-187 followed by .backward()):
 * the same tangent trace, with the loss adjoint formed in the kernel from the primal hit point on sensor
 * slot `sensor_slot`, g = (x-cx, y-cy)(w/W)/sqrt(((x-cx)^2+(y-cy)^2) w/W) for hits with intensity w > 0.
 * out[3*n_theta + 2] (float64, device, zeroed by the caller) receives
 *   A_k = sum g . d(hit_xy)/d theta_k,  Bx_k = sum (w/W) dx/d theta_k,  By_k = sum (w/W) dy/d theta_k,  Gx, Gy = sum g,
 * from which dL/d theta_k = A_k - Gx Bx_k - Gy By_k (centroid as reference point) or A_k (fixed target).
 * All five are plain sums over rays, so ranks add them before combining. */
int rtt_trace_seq_spot_grad_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, int64_t n,
                                const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                                const float* tan_dev, int n_theta, int sensor_slot, double cx, double cy, double wsum,
                                double* out, void* stream);

/* ---- reverse-mode backward of the sequential trace (the same autograd semantics as above: scene/sequential.py:12-36
 * under optim/goals.py:144-187 and tests/test_optimize_singlet.py:66-116 of the reference; gradient masks and shared
 * Parameters, geom/transform.py:29-46 and elements/lens.py:42-56, are the caller's contraction).
 * Every ray is re-traced once keeping its state in front of each surface it hits, then the adjoint of the ray state is
 * swept back through the surfaces.  The cost does not depend on the number of trainable scalars: the kernel returns the
 * adjoints of the record slots themselves (slot numbering as in the tangent table: Rs 0-8, ts 9-11, Rj 12-20, tj 21-23,
 * geom 24-27, phys 28-31), summed over the rays, and the caller contracts them with d(slot)/d(parameter).
 *   slot_mask  host array [n_surf]: bit j of slot_mask[s] asks for the adjoint of slot j of surface s (NULL: none);
 *   slot_adj   device array of one float64 per set bit, ordered by surface and, within a surface, by slot; overwritten;
 *   adj_pos, adj_dir, adj_sens   upstream adjoints as for rtt_trace_seq_tangent_f32 (each nullable);
 *   spot_slot >= 0 instead forms the upstream adjoint of the spot-size loss in the kernel, on the hits of that sensor
 *     slot with intensity w > 0:  g = (x-cx, y-cy)(w/W)/sqrt(((x-cx)^2+(y-cy)^2) w/W) - (w/W)(Gx, Gy).
 *     With spot_mom (device memory: the six float64 sums of rtt_spot_moments_f32, [W, sum w x, sum w y, sum sqrt, Gx, Gy],
 *     reduced over ranks) W = max(spot_mom[0], 1e-12) and, when spot_centroid != 0, the centre is the centroid
 *     spot_mom[1..2] / W and (Gx, Gy) = spot_mom[4..5]; with spot_centroid == 0 the centre is (cx, cy) and (Gx, Gy) = 0.
 *     Without spot_mom (NULL): centre (cx, cy), W = wsum, (Gx, Gy) = 0 — a fixed target with sums known on the host.
 *     Reading the sums from device memory lets a caller chain moments and backward without a host round trip;
 *   ray_adj    (nullable) [6][ld_ray]: adjoint of the input rays, as rtt_trace_seq_ray_adjoint_f32 returns it;
 *   workspace  device scratch for the per-block sums, rtt_trace_seq_adjoint_workspace_bytes(number of set bits) bytes.
 * Limits: rtt_trace_seq_adjoint_max_slots(n_surf) set bits per call (the per-surface ray states and the per-lane sums
 * live in the 160 KB of LDS of a compute unit); RTT_E_SIZE beyond that — split the mask over several calls, or use the
 * tangent entry points.  prog_host is required. */
int rtt_trace_seq_adjoint_max_slots(int n_surf);
int64_t rtt_trace_seq_adjoint_workspace_bytes(int n_slots);
int rtt_trace_seq_adjoint_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, int64_t n,
                              const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, const uint32_t* slot_mask,
                              const float* adj_pos, const float* adj_dir, int64_t ld_adj, const float* adj_sens, int64_t sens_ld,
                              int spot_slot, int spot_centroid, double cx, double cy, double wsum, const double* spot_mom,
                              float* ray_adj, int64_t ld_ray, double* slot_adj, void* workspace, int64_t workspace_bytes,
                              void* stream);

/* The same backward for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): no counterpart in the reference.  Identical to rtt_trace_seq_adjoint_f32 in
 * arguments (without the workspace) and in results, bit for bit; the per-block sums go to a static device array of the
 * library (2 MiB = 2048 blocks x 128 slot sums of float64 per GPU, held until the process ends) instead of a caller's
 * workspace.  At most 128 set bits in slot_mask (RTT_E_SIZE beyond).  Neither this call nor anything it calls
 * allocates or frees.  All calls on one GPU share that array: calls that overlap in time on one GPU (different
 * streams, different threads) race on it and must be serialised by the caller; calls on different GPUs are independent. */
int rtt_trace_seq_adjoint_static_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in, int64_t n,
                                     const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                                     const uint32_t* slot_mask, const float* adj_pos, const float* adj_dir, int64_t ld_adj,
                                     const float* adj_sens, int64_t sens_ld, int spot_slot, int spot_centroid, double cx, double cy,
                                     double wsum, const double* spot_mom, float* ray_adj, int64_t ld_ray, double* slot_adj,
                                     void* stream);

/* ---- the differentiable trace with a tape: what torch autograd keeps between the forward of scene/sequential.py:12-36
 * and its backward (optim/goals.py:144-187, tests/test_optimize_singlet.py:66-116), reduced to what the backward cannot
 * make again cheaply.  rtt_trace_seq_tape_f32 is rtt_trace_seq_f32 (mom_slot < 0) or rtt_trace_seq_moments_f32
 * (mom_slot >= 0; then pos_out, dir_out, inten_out may all be NULL) that also fills `tape`: per surface that some ray of
 * an aligned group of 128 rays hits, the direction in front of it and the path length of every ray of the group
 * (16 B per ray and surface hit), and per group the set of those surfaces.  `tape` is device memory of
 * rtt_trace_seq_tape_bytes(n, n_surf) bytes, 256-byte aligned, opaque to the caller.  prog_host is required.
 * rtt_trace_seq_adjoint_tape_f32 is rtt_trace_seq_adjoint_f32 without its forward sweep: it reads the input positions
 * and the tape of the same rays and program (the directions and intensities are not read), at about two thirds of the
 * arithmetic.  spot_w: in spot mode, the device plane [n] of intensities the spot sensor recorded (sens_w of that slot
 * as the forward call wrote it).  Limits: programs of at most 6 surfaces and rtt_trace_seq_adjoint_tape_max_slots(n_surf)
 * slot sums per call (0 = no tape for this program: use rtt_trace_seq_adjoint_f32); RTT_E_SIZE beyond. */
int64_t rtt_trace_seq_tape_bytes(int64_t n, int n_surf);
int rtt_trace_seq_adjoint_tape_max_slots(int n_surf);
int rtt_trace_seq_tape_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                           float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                           const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                           float* sens_hit, float* sens_w, uint8_t* sens_mask, int64_t sens_ld,
                           int mom_slot, double* mom, void* workspace, int64_t workspace_bytes,
                           void* tape, int64_t tape_bytes, void* stream);
int rtt_trace_seq_adjoint_tape_f32(const float* pos_in, int64_t ld_in, int64_t n,
                                   const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, const uint32_t* slot_mask,
                                   const float* adj_pos, const float* adj_dir, int64_t ld_adj, const float* adj_sens, int64_t sens_ld,
                                   int spot_slot, int spot_centroid, double cx, double cy, double wsum, const double* spot_mom,
                                   const float* spot_w, float* ray_adj, int64_t ld_ray, double* slot_adj, void* workspace,
                                   int64_t workspace_bytes, const void* tape, int64_t tape_bytes, void* stream);

/* ---- nearest-hit cast: Scene.ray_cast (scene/base.py:144-178) ---------------------------------
 * winner[i] = index of the first surface with the smallest t (or -1 when every t is inf),
 * tmin[i] = that t.  Either output may be NULL. */
int rtt_ray_cast_f32(const float* pos, const float* dir, int64_t ld, int64_t n,
                     const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                     int32_t* winner, float* tmin, void* stream);

/* ---- one non-sequential bounce: Scene.step (scene/base.py:180-235) ----------------------------
 * ray_cast + winner dispatch + scatter_update for rays with a hit and intensity > 0.
 * winner (nullable) receives the ray_cast result.  Sensor outputs as in rtt_trace_seq_f32; a bounce writes a 1 only for
 * rays that went to a sensor, so the kernel first clears the mask entries of all rays on all slots (with a host copy of
 * the program, from which it knows the slots; without one the caller passes the mask planes zeroed). */
int rtt_step_nearest_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                         float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                         const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                         int32_t* winner, float* sens_hit, float* sens_w, uint8_t* sens_mask,
                         int64_t sens_ld, void* stream);

/* ---- single-bounce shaded view: Renderer.render_3d (render/camera.py:191-257) and its colouring
 * (render/camera.py:259-318).  Nearest hit over the surfaces whose bit in skip_mask is clear (the reference
 * leaves aperture elements out of the competition), normal of the winner (geom/shape.py:61-87), colour =
 * base_rgb[3*s..] * (0.3 + 0.7 |n . light_dir|) clamped to [0,1]; background where nothing is hit.
 * base_rgb is a device array [n_surf][3]; light_dir and background are host arrays; rgb is [n][3]. */
int rtt_render_f32(const float* pos, const float* dir, int64_t ld, int64_t n,
                   const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                   const float* base_rgb, uint64_t skip_mask, const float light_dir[3],
                   const float background[3], float* rgb, void* stream);

/* ---- the whole bounce loop: Scene.simulate (scene/base.py:129-142) -------------------------------
 * Up to max_bounces repetitions of Scene.step in one launch, each ray running its own loop in registers (a ray
 * that misses, or whose intensity is 0, is not touched by later bounces, so stopping it early changes nothing).
 * bounces[i] (nullable) = number of surfaces ray i interacted with.  Sensor hits are appended, in no particular
 * order, to hit_list[k] = (hit_local x, y, z, intensity before the surface) with hit_keys[k] =
 * (bounce * RTT_MAX_SURFACES + sensor_slot, ray index); *hit_count (zeroed by the caller) counts every hit
 * offered, also those beyond the capacity, which are dropped: a caller that finds a count above its capacity
 * repeats the call with more room.  Sorting the keys restores the reference's order (by bounce, then by ray).
 * The list is kept in hit_parts parts (a power of two, 1 .. 64) of hit_capacity / hit_parts entries each: part p
 * holds entries [p cap_p, p cap_p + min(count_p, cap_p)) and its counter is hit_count[32 p] (one 128-byte line per
 * counter; the caller zeroes 32 hit_parts int32).  A block appends to part blockIdx mod hit_parts, so the wavefronts
 * that meet a sensor in the same bounce do not all add to one address (at 4 Mi rays a single counter costs more than
 * the bounces themselves); one part is right for a few thousand rays. */
int rtt_simulate_nearest_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                             float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                             const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, int max_bounces,
                             int32_t* bounces, float* hit_list, int32_t* hit_keys, int32_t* hit_count,
                             int64_t hit_capacity, int hit_parts, void* stream);

/* The same loop run in stages of a few bounces (scene/base.py:129-142): after each stage the rays that still move are
 * packed into compact planes for the next one and the rays that have come to rest are written to their places, so
 * wavefronts stay full of moving rays however unevenly a bundle thins out.  Same outputs as rtt_simulate_nearest_f32,
 * bit for bit (a ray's arithmetic does not depend on the rays it shares a wavefront with).  Needs a host copy of the
 * program, output planes distinct from the input planes (RTT_E_ALIGN), a program without RTT_PHYS_FRESNEL (its draws
 * are keyed by the place in the launch: RTT_E_KIND) and a 256-byte aligned device workspace of
 * rtt_simulate_nearest_staged_workspace_bytes(n) bytes.  Pays off for bundles of many rays and loops of many bounces;
 * for a few thousand rays the single launch is faster. */
int64_t rtt_simulate_nearest_staged_workspace_bytes(int64_t n);
int rtt_simulate_nearest_staged_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                                    float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                                    const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, int max_bounces,
                                    int32_t* bounces, float* hit_list, int32_t* hit_keys, int32_t* hit_count,
                                    int64_t hit_capacity, int hit_parts, void* workspace, int64_t workspace_bytes,
                                    void* stream);

/* Scene.simulate carrying derivatives (the reference gets them from torch autograd through scene/base.py:129-235;
 * optim/goals.py:63-96 relies on that): the bounce loop above where every ray is a dual number over n_theta (2, 4,
 * 6 or 8) parameter directions given by the tangent table tan_dev [n_surf][32][n_theta] (layout as for
 * rtt_trace_seq_tangent_f32).  The winning surface of a bounce is chosen on the values (piecewise constant in the
 * parameters).  ray_tan [6][n_theta][ld_out] receives d(pos x, y, z, dir x, y, z)/d(theta_k) of the final rays,
 * hit_tan [hit_capacity][3][n_theta] the tangents of hit_local for the entry of the same index in hit_list.
 * With seed_rays != 0 (n_theta >= 6) directions 0..5 are the components (pos x, y, z, dir x, y, z) of each ray's own
 * input on top of the table, so that ray_tan[c][j][i] is the Jacobian of ray i: the reference's autograd reaches
 * the rays' origin (a trainable bundle pose, an earlier step) the same way.
 * prog_host is required.  hit_count must be zeroed by the caller; a count above hit_capacity means the list was
 * truncated and the call must be repeated with more room. */
int rtt_simulate_nearest_tangent_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                                     float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                                     const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, int max_bounces,
                                     const float* tan_dev, int n_theta, int seed_rays, float* ray_tan, float* hit_list,
                                     int32_t* hit_keys, float* hit_tan, int32_t* hit_count, int64_t hit_capacity,
                                     void* stream);

/* ---- Element.intersectTest (elements/parent.py:30-42; geom/shape.py:25-59; primitives.py:38-57)
 * t_out[s*ld_t + i] = distance to surface s of the program, inf on a miss. */
int rtt_intersect_test_f32(const float* pos, const float* dir, int64_t ld, int64_t n,
                           const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf,
                           float* t_out, int64_t ld_t, void* stream);

/* ---- Element.forward / Shape.forward / Surface.forward for ONE surface, every ray, no mask
 * (elements/parent.py:44-58; geom/shape.py:61-87; geom/primitives.py:59-96).
 * Outputs (each nullable): t[n], hit_global/normal_global/hit_local/new_dir plane triples with
 * leading dimension ld_o, imult[n]. */
int rtt_surface_forward_f32(const float* pos, const float* dir, int64_t ld, int64_t n,
                            const RttSurface* prog_dev, const RttSurface* prog_host, int surf_index,
                            float* t, float* hit_global, float* normal_global, float* hit_local,
                            float* new_dir, float* imult, int64_t ld_o, void* stream);

/* ---- SurfaceFunction.forward(hit_local, ray_dir, normal) on its own (phys/std.py:16-32,91-145,221-254;
 * phys/filter.py:24-33).  Inputs and new_dir are plane triples with leading dimension ld; only the
 * physics (and, for RTT_PHYS_APERTURE, bound) fields of the single record are used. */
int rtt_apply_physics_f32(const float* hit_local, const float* dir, const float* normal, int64_t ld,
                          int64_t n, const RttSurface* prog_dev, const RttSurface* prog_host,
                          float* new_dir, float* imult, void* stream);

/* ---- spot-size moments: SpotSizeLoss.forward (optim/goals.py:165-183) -------------------------
 * pass 0: out[0..2] += sum w, sum w x, sum w y over entries with mask != 0 and w > 0.
 * pass 1: out[3] += sum sqrt(((x-cx)^2 + (y-cy)^2) * (w / wsum)) over the same entries.
 * pass 2: pass 1 and out[4..5] += sum g, g = (x-cx, y-cy)(w/wsum)/sqrt(((x-cx)^2+(y-cy)^2) w/wsum): what the centroid
 *         term of the loss gradient needs (rtt_trace_seq_adjoint_f32, spot_mom); `out` then holds 6 doubles.
 * pass 3: pass 2 with wsum = max(out[0], 1e-12) and the centre (out[1], out[2]) / wsum read from `out` itself, i.e. from
 *         the (rank-reduced) result of pass 0: the chain pass 0 -> centroid -> pass 3 needs no host round trip.
 * pass 4: pass 2 with wsum = max(out[0], 1e-12) read from `out` and the centre (cx, cy) as given (fixed target).
 * `out` is 4 doubles in device memory that the caller zeroes before pass 0. */
int rtt_spot_moments_f32(const float* x, const float* y, const float* w, const uint8_t* mask, int64_t n,
                         int pass, double cx, double cy, double wsum, double* out, void* stream);

/* ---- ray sources: Bundle.sample(N) (rays/bundle.py:30-37) generated on the device ----------------
 * Local sample (rays/bundle.py:40-171) -> RayTransformBundle.transform_ (geom/transform.py:247-264:
 * pos @ R^T + t, dir @ R^T) -> Rays.initialize (rays/ray.py:42-82: dir / max(|dir|, 1e-12), intensity 1,
 * wavelength 0, id = ray_id as int8).  The reference draws its uniforms from torch.distributions.Uniform
 * (low + rand * (high - low)); here rand comes from Philox4x32-10 keyed by `seed`: the pair of rays with
 * global indices (2j, 2j+1) uses counter (j lo, j hi, 0, 0) and the outputs (x0, x1 | x2, x3) as its
 * (first, second) uniform, u = (x >> 8) * 2^-24.  Sources that draw four uniforms per ray (RTT_SRC_PANEL,
 * RTT_SRC_SPHERE) use counter (i lo, i hi, 1, 0) of the ray's global index i: (x0, x1) are the two position
 * draws, (x2, x3) the two direction draws.  A ray's values depend only on (seed, global index), so shards
 * that pass their global `offset` reproduce the single-call bundle. */
enum {
  RTT_SRC_PENCIL = 0, /* Bundle: pos 0, dir +z                                 rays/bundle.py:24-28   */
  RTT_SRC_DISK = 1,   /* DiskSample: theta = U(par[2], par[3]) first, then r = sqrt(U(par[0], par[1]));
                         pos (r cos, r sin, 0), dir +z                         rays/bundle.py:40-56,83-98 */
  RTT_SRC_LINE = 2,   /* pos (U(par[0], par[1]), 0, 0), dir +z                 rays/bundle.py:101-118 */
  RTT_SRC_FAN = 3,    /* theta = U(par[0], par[1]); dir (0, sin, cos)          rays/bundle.py:121-140 */
  RTT_SRC_POINT = 4,  /* phi = acos(1 - 2 U(par[0], par[1])) first, then theta = U(par[2], par[3]);
                         dir (cos(theta) sin(phi), sin(theta) sin(phi), cos(phi)) rays/bundle.py:58-80,143-171 */
  RTT_SRC_GAUSS = 5,  /* GaussianBeam: pos (par[0] z0, par[1] z1, 0), z0 and z1 independent standard normals
                         (Box-Muller of the two uniforms: sqrt(-2 ln(1-ua)) (cos, sin)(2 pi ub)), dir +z.
                         The reference draws torch.normal values: same distribution, other stream.  rays/beam.py:9-41 */
  RTT_SRC_RAYLEIGH = 6, /* cos(theta) = A - 1/A, A = cbrt(sqrt(q^2+1) - q), q = 2 - 4 ua; phi = 2 pi ub;
                         dir (sin cos(phi), sin sin(phi), cos)                 rays/particle.py:90-125 */
  RTT_SRC_PANEL = 7,  /* PanelSource: position by `area_kind` from par (RTT_AREA_RECT: x = U(par[0], par[1]) first,
                         then y = U(par[2], par[3]); RTT_AREA_RING: as RTT_SRC_DISK), direction by `emit_kind`
                         from epar (RTT_EMIT_LAMBERT: (sqrt(u1) cos(2 pi u2), sqrt(u1) sin(2 pi u2), sqrt(1-u1));
                         RTT_EMIT_CONE: as RTT_SRC_POINT with epar for par)      rays/panels.py:10-154 */
  RTT_SRC_SPHERE = 8  /* LambertianSphere of radius par[0]: outward normal n from cos_t = 2 u1 - 1, phi = 2 pi u2,
                         pos = par[0] n; cosine-weighted direction about n in the Frisvad basis rays/particle.py:9-87 */
};
enum { RTT_AREA_RECT = 0, RTT_AREA_RING = 1 };
enum { RTT_EMIT_LAMBERT = 0, RTT_EMIT_CONE = 1 };

typedef struct RttSource {
  int32_t kind, ray_id;
  float par[4];
  float R[9], t[3]; /* RayTransformBundle: rotation matrix (row-major) and translation */
  uint64_t seed;    /* Philox key */
  uint64_t offset;  /* global index of the first ray of this call */
  int32_t area_kind, emit_kind; /* RTT_SRC_PANEL only */
  float epar[4];                /* RTT_SRC_PANEL with RTT_EMIT_CONE */
} RttSource;

/* pos, dir: plane triples with leading dimension ld; inten, id, wavelength: [n], each nullable. */
int rtt_sample_bundle_f32(const RttSource* src, float* pos, float* dir, int64_t ld, float* inten, int8_t* id,
                          float* wavelength, int64_t n, void* stream);

/* ---- Bundle.sample(N) followed by SequentialScene.simulate(rays) in one launch ---------------------
 * (rays/bundle.py:30-37 + scene/sequential.py:12-36; the pair of calls optim/goals.py:150-156 and the reference's
 * scripts make for every fresh bundle.)  Each lane generates its two rays exactly as rtt_sample_bundle_f32 does for the
 * same `src` (same Philox blocks, same formulas), traces them as rtt_trace_seq_f32 does and writes the traced planes
 * together with the ids and wavelengths of the new bundle: 33 B written per ray, nothing read.  The result equals the
 * two calls made one after the other.  Restrictions: a source that draws two uniforms per ray (kind <=
 * RTT_SRC_RAYLEIGH, else RTT_E_KIND) and an even src->offset (a lane's pair is one Philox block, else RTT_E_ALIGN);
 * callers outside them sample first and trace second.  id_out and wavelength_out are nullable; the sensor arguments
 * are those of rtt_trace_seq_f32. */
int rtt_sample_trace_seq_f32(const RttSource* src, float* pos_out, float* dir_out, float* inten_out, int64_t ld_out,
                             int8_t* id_out, float* wavelength_out, int64_t n, const RttSurface* prog_dev,
                             const RttSurface* prog_host, int n_surf, float* sens_hit, float* sens_w,
                             uint8_t* sens_mask, int64_t sens_ld, void* stream);

/* ---- per-id spot sizes: Sensor.getSpotSizeParallel_xy (elements/sensor.py:87-176) ---------------
 * Hits are given densely (x, y, w, id per ray; mask != 0 marks a recorded hit, NULL = all).  Group g = id + 128.
 * pass 0: out[3g + (0,1,2)] += sum w, sum w x, sum w y.
 * pass 1: out[3g + (0,1)] += sum w |x - centers[2g]|^norm_ord, sum w |y - centers[2g+1]|^norm_ord.
 * out is [256][3] float64 in device memory (zeroed by the caller before each pass), centers [256][2] float64. */
int rtt_spot_groups_f32(const float* x, const float* y, const float* w, const uint8_t* mask, const int8_t* id,
                        int64_t n, int pass, const double* centers, double norm_ord, double* out, void* stream);

/* ---- prepared launches: the stepping loops of the reference (benchmarks/sim_benchmark.py:107-124: scene.step() twenty
 * times on the same rays; scene/sequential.py:12-36 called per optimisation step on the same planes) repeat one launch
 * with the same arguments.  rtt_prepare_* validates the arguments of rtt_trace_seq_f32 / rtt_step_nearest_f32 once
 * (program checks, opcode and culling tables, access width, grid) and returns an opaque handle in host memory (the one
 * thing this library allocates, freed by rtt_release_prepared); rtt_launch_prepared enqueues the launch.  The sensor
 * record of a launch is ONE device allocation of rtt_sensor_record_bytes(slots, sens_ld) bytes that the caller passes per
 * launch (NULL for a program without sensor surfaces): hit planes [slots][3][sens_ld] float, then intensities
 * [slots][sens_ld] float, then masks [slots][sens_ld] bytes, as rtt_trace_seq_f32 / rtt_step_nearest_f32 fill their three
 * buffers (the bounce clears the masks first).  The ray planes, the device program and sens_ld are bound by the handle:
 * the caller keeps them alive and prepares again when any of them, or an rtt_tune setting, changes (RTT_E_KIND: stale;
 * also when the launch is made with another device current than the handle was prepared with).
 * A handle may be launched from one thread at a time.  n >= 1; prog_host is required. */
typedef struct RttLaunch RttLaunch;
int rtt_prepare_trace_seq_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                              float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                              const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, int64_t sens_ld,
                              RttLaunch** out);
int rtt_prepare_step_nearest_f32(const float* pos_in, const float* dir_in, const float* inten_in, int64_t ld_in,
                                 float* pos_out, float* dir_out, float* inten_out, int64_t ld_out, int64_t n,
                                 const RttSurface* prog_dev, const RttSurface* prog_host, int n_surf, int64_t sens_ld,
                                 RttLaunch** out);
int rtt_launch_prepared(const RttLaunch* launch, void* sens_base, void* stream);
void rtt_release_prepared(RttLaunch* launch);
int64_t rtt_sensor_record_bytes(int n_slots, int64_t sens_ld);

/* Launch tuning, process-wide, optional.  key 0: rays carried per lane by rtt_trace_seq_f32 (1, 2 or 4);
 * key 1: cap on the number of 256-thread blocks per launch (the remainder is grid-stride);
 * key 2: 1 stages the surface program in LDS, 0 (default) reads it with scalar loads;
 * key 3: 0 makes the nearest-hit kernels test every surface for every ray, 1 (default) lets them skip elements
 * whose bounding sphere lies beyond the nearest hit found so far (same results);
 * key 4: number of blocks of rtt_spot_moments_f32 (default 512: every block ends in float64 atomics on the same
 * few addresses, which serialise at about 12 ns each, so more blocks cost more than they stream);
 * key 5: cap on the blocks of the other reducing kernels (tangent traces), default 2048;
 * key 6: rays per lane of rtt_trace_seq_adjoint_f32: 0 (default) two when the LDS need allows two blocks per compute
 * unit, else one; 1 or 2 force the choice;
 * key 7: the per-surface ray states of rtt_trace_seq_adjoint_f32 in LDS: 0 (default) compact (direction and path length,
 * positions walked again) for programs of at most 6 surfaces where that lets a third wavefront per SIMD in, 1 always
 * the full states;
 * key 8: threads per block of rtt_spot_moments_f32 on 16-byte aligned records, 1024 (default) or 256;
 * key 9: size of the grid of the reverse sweeps in rounds of resident blocks, 1 .. 8 (default 3, at most 2048 blocks);
 * key 10: rtt_simulate_nearest_staged_f32: 0 (default) every stage writes the rays that come to rest to the output
 * planes itself (seven scattered 4-byte stores for a ray that rests after the first stage), 1 such rays leave one
 * 32-byte record each and a last pass writes them to the planes in index order (measured on the 32-surface scene,
 * 4 Mi rays x 100 bounces: 8 % fewer HBM bytes, 1.5 % more time: the last pass).  Same results;
 * key 11: rtt_trace_seq_adjoint_f32 on a program of spherical lens faces, lens edges and bounded planes only: 1 (default)
 * runs the instantiation of the kernel that holds just these visits (fewer registers, nothing spilled: 3 % off a
 * spot-size step at 32 Mi rays), 0 the general one;
 * key 12: rtt_step_nearest_f32 / rtt_simulate_nearest_f32 / rtt_simulate_nearest_staged_f32 (and a prepared bounce) on a
 * program in which every record matches a specialised visit: 1 (default) runs the instantiation without the generic
 * interpreter (no registers spilled at four wavefronts per SIMD: on the 32-surface scene 3 % off the one-bounce step,
 * 15 % fewer HBM bytes and 1 % less time for the 100-bounce loop), 0 the general one.  Same results, bit for bit.
 * rtt_trace_seq_adjoint_f32 has the same instantiation for such programs (switched with key 11).
 * No counterpart in the reference. */
int rtt_tune(int key, int value);

const char* rtt_strerror(int code);
int rtt_abi_version(void);
int rtt_surface_record_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* RTT_HIP_H */
