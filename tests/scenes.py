"""Scene builders shared by the golden-vector generator and the tests.

Every builder takes a namespace `ns` exposing the reference's public class names
(``SingletLens``, ``Sensor``, ``Disk``, ``RayTransform`` ...).  ``oracle/make_goldens.py`` passes a
namespace built from the upstream reference, the tests pass one built from ``raytracetorch_amd``;
constructor calls are therefore literally identical on both sides.
"""
import math
import types

import torch


def namespace_from_package(pkg):
    ns = types.SimpleNamespace()
    for mod in (pkg.geom, pkg.phys, pkg.elements, pkg.scene, pkg.rays):
        for k in dir(mod):
            if not k.startswith("_"):
                setattr(ns, k, getattr(mod, k))
    for k in ("HalfSphere", "HalfCyl", "BoundedHalfSphere", "SingleCone"):
        setattr(ns, k, getattr(pkg.geom.bounded, k))
    ns.Cylinder = pkg.geom.primitives.Cylinder
    ns.Cone = pkg.geom.primitives.Cone
    ns.SpotSizeLoss = pkg.optim.goals.SpotSizeLoss
    for k in ("FocalLengthLoss", "SpotTargetLoss"):
        setattr(ns, k, getattr(pkg.optim.goals, k))
    for k in ("ThicknessConstraint", "SpacingConstraint", "SystemLengthConstraint"):
        setattr(ns, k, getattr(pkg.optim.constraints, k))
    ns.ParabolicMirror, ns.ParabolicMirrorXZ = pkg.elements.mirror.ParabolicMirror, pkg.elements.mirror.ParabolicMirrorXZ
    ns.CylindricalMirror = pkg.elements.mirror.CylindricalMirror
    for k in ("Camera", "OrbitCamera", "Renderer"):
        setattr(ns, k, getattr(pkg.render, k))
    ns.Paths = pkg.rays.Paths
    ns.ElementCustom = pkg.elements.parent.ElementCustom
    ns.Fuzzy = pkg.phys.filter.Fuzzy
    ns.IdealMirror = pkg.elements.ideal.IdealMirror
    ns.IdealCylThinLens = pkg.elements.ideal.IdealCylThinLens
    return ns


def T(ns, z=0.0, x=0.0, y=0.0, rot=None):
    return ns.RayTransform(translation=[x, y, z], rotation=rot)


# ---------------------------------------------------------------------------------------------
# ray sources (deterministic given the torch seed)
# ---------------------------------------------------------------------------------------------
def disk_bundle(ns, radius=4.0, z=-30.0, x=0.0, y=0.0, rot=None, ray_id=0):
    return ns.CollimatedDisk(radius=radius, ray_id=ray_id,
                             transform=ns.RayTransformBundle(translation=[x, y, z], rotation=rot))


def gaussian_rays(ns, n, z=-10.0, spread=1.0, slope=0.1):
    """C1 input: pos=(N(0,1), N(0,1), z), dir=normalize(slope*N, slope*N, 1)."""
    pos = torch.randn(n, 3) * spread
    pos[:, 2] = z
    d = torch.randn(n, 3) * slope
    d[:, 2] = 1.0
    return ns.Rays.initialize(pos, d)


# ---------------------------------------------------------------------------------------------
# elements / systems
# ---------------------------------------------------------------------------------------------
def singlet(ns, **kw):
    args = dict(c1=0.05, c2=-0.05, d=10.0, t=3.0, ior_glass=1.5, ior_media=1.0, inked=True)
    args.update(kw)
    return ns.SingletLens(**args)


def system5(ns):
    """C3: lens, stop, lens, mirror, sensor = 9 surfaces."""
    return [
        singlet(ns, transform=T(ns, 0.0)),
        ns.CircularAperture(3.5, invert=True, transform=T(ns, 5.0)),
        ns.SingletLens(c1=0.02, c2=-0.03, d=16.0, t=2.0, ior_glass=1.6, inked=True, transform=T(ns, 10.0)),
        ns.SphericalMirror(c1=-0.01, d=0.0, diameter=30.0, transform=T(ns, 40.0)),
        ns.Sensor(ns.Disk(20.0, transform=T(ns, 25.0))),
    ]


def naive32(ns, z0=0.0):
    """C4: 6 singlets + 2 cylindrical singlets + mirror + sensor = 32 surfaces (`z0` moves the whole group along z)."""
    els = [ns.SingletLens(c1=0.03, c2=-0.02, d=12.0, t=2.0, ior_glass=1.5, inked=True,
                          transform=T(ns, z0 + 8.0 * i)) for i in range(6)]
    els += [ns.CylSingletLens(c1=0.02, c2=-0.02, height=14.0, width=14.0, t=2.0, ior_glass=1.5, inked=True,
                              transform=T(ns, z0 + 50.0 + 8.0 * i)) for i in range(2)]
    els.append(ns.SphericalMirror(c1=-0.01, d=0.0, diameter=30.0, transform=T(ns, z0 + 80.0)))
    els.append(ns.Sensor(ns.Disk(20.0, transform=T(ns, z0 + 70.0))))
    return els


def triplet_train(ns, count=10, grads=False, sensors=()):
    """`count` weak triplets in a row, alternately converging and diverging so that a collimated beam stays inside
    them (7 surfaces each: ten of them are more surfaces than one launch takes); a sensor behind every triplet whose
    index is in `sensors`."""
    els = []
    for i in range(count):
        sgn = 1.0 if i % 2 == 0 else -1.0
        els.append(ns.TripletLens(c1=0.012 * sgn, c2=-0.006 * sgn, c3=0.004 * sgn, c4=-0.010 * sgn, d=12.0, t1=2.0, t2=1.0,
                                  t3=2.5, ior_glass1=1.5, ior_glass2=1.65, ior_glass3=1.55, c1_grad=grads, c4_grad=grads,
                                  ior_glass2_grad=grads, transform=T(ns, 9.0 * i, x=0.05 * (i % 3 - 1), rot=[0.002 * i, -0.001 * i, 0.0])))
        if i in sensors:
            els.append(ns.Sensor(ns.Disk(9.0, transform=T(ns, 9.0 * i + 4.5))))   # (a triplet is centred on its pose: +-2.75)
    return els


def spot_system(ns, grads=True):
    """C5: trainable singlet + sensor near the back focal plane."""
    lens = singlet(ns, c1_grad=grads, c2_grad=grads, t_grad=grads, ior_glass_grad=grads)
    sensor = ns.Sensor(ns.Disk(6.0, transform=T(ns, 19.0)))
    return lens, sensor


def tilt_system(ns, rot):
    """Singlet whose element pose (translation and rotation), curvatures and indices are all trainable, and a sensor
    whose pose is trainable too."""
    lens = singlet(ns, c1_grad=True, c2_grad=True, t_grad=True, ior_glass_grad=True, ior_media_grad=True,
                   transform=ns.RayTransform(translation=[0.2, -0.1, 1.0], rotation=rot, trans_grad=True, rot_grad=True))
    sensor = ns.Sensor(ns.Disk(8.0, transform=ns.RayTransform(translation=[0.0, 0.0, 19.0], rotation=[0.0, 0.0, 0.0],
                                                              trans_grad=True, rot_grad=True)))
    return lens, sensor


def folded_system(ns):
    """Nearest-hit scene with several bounces per ray and trainable poses, curvatures and index: a sensor that rays
    cross twice, a singlet, and a concave mirror that sends the beam back through both.  All coordinates stay below
    0.5: a ray that starts on a surface finds that surface again at |t| of a few ulp of its position, and the
    reference's `t <= 1e-6` filter only removes that root while positions are this small (at |z| = 40 a float32
    trace re-hits the surface it stands on for a third of the rays)."""
    lens = singlet(ns, c1=5.0, c2=-5.0, d=0.1, t=0.03, c1_grad=True, c2_grad=True, t_grad=True, ior_glass_grad=True,
                   transform=ns.RayTransform(translation=[0.001, -0.0005, 0.0], rotation=[0.0, 0.0, 0.0], trans_grad=True, rot_grad=True))
    mirror = ns.SphericalMirror(c1=-2.0, d=0.0, diameter=0.3, c1_grad=True,
                                transform=ns.RayTransform(translation=[0.0, 0.0, 0.4], rotation=[0.01, -0.02, 0.0],
                                                          trans_grad=True, rot_grad=True))
    sensor = ns.Sensor(ns.Disk(0.12, transform=ns.RayTransform(translation=[0.0, 0.0, -0.1], trans_grad=True)))
    return lens, mirror, sensor


def folded_bundle(ns):
    return ns.CollimatedDisk(radius=0.03, ray_id=0,
                             transform=ns.RayTransformBundle(translation=[0.0, 0.0, -0.3], rotation=[0.01, -0.01, 0.0]))


def manual_spot_loss(ns, lens, rays, target_z=100.0):
    """The closure of the reference's tests/test_optimize_singlet.py:66-116: two Element calls chained by assigning
    pos / dir, projection onto the plane z = target_z, mean squared distance from the axis."""
    p1, d1, _ = lens(rays, surf_idx=0)
    rays.pos = p1
    rays.dir = d1
    p2, d2, _ = lens(rays, surf_idx=1)
    t_dist = (target_z - p2[:, 2]) / (d2[:, 2] + 1e-6)
    x, y = p2[:, 0] + t_dist * d2[:, 0], p2[:, 1] + t_dist * d2[:, 1]
    return torch.mean(x ** 2 + y ** 2)


def manual_lens(ns):
    return ns.SingletLens(c1=0.016667, c2=-0.00283, d=25.4, t=4.0, ior_glass=1.5168, ior_media=1.0, c1_grad=True,
                          c2_grad=True, t_grad=True, ior_glass_grad=True, inked=False)


def soft_edge(hit_local, unit=1.0):
    """Intensity profile of the Fuzzy filter in the goldens: a Gaussian apodisation that is exactly zero far out."""
    r2 = (hit_local[:, 0] ** 2 + hit_local[:, 1] ** 2) / (unit * unit)
    return torch.exp(-r2 / 4.0) * (r2 < 9.0).to(hit_local.dtype)


def soft_edge_small(hit_local):
    return soft_edge(hit_local, unit=0.01)


def fuzzy_system(ns, small=False):
    """Sensor, singlet, Fuzzy apodiser on a disk, sensor: the second sensor sees apodised intensities.  `small`
    is the same system at 1/100 scale, where a float32 bounce loop does not re-hit the surface a ray stands on."""
    u = 0.01 if small else 1.0
    lens = singlet(ns, c1=5.0, c2=-5.0, d=0.1, t=0.03) if small else singlet(ns)
    return [ns.Sensor(ns.Disk(8.0 * u, transform=T(ns, -5.0 * u))), lens,
            ns.ElementCustom(ns.Disk(6.0 * u, transform=T(ns, 8.0 * u)), ns.Fuzzy(soft_edge_small if small else soft_edge), "cpu"),
            ns.Sensor(ns.Disk(6.0 * u, transform=T(ns, 19.0 * u)))]


def paraxial_system(ns):
    """Elements of every kind that has a paraxial matrix, posed and partly trainable, in axial order."""
    tf = lambda z, **kw: ns.RayTransform(translation=[0.2, -0.1, z], rotation=[0.01, -0.02, 0.0], **kw)
    return [
        ns.SingletLens(c1=0.02, c2=-0.02, d=10.0, t=3.0, ior_glass=1.4, c1_grad=True, c2_grad=True, t_grad=True,
                       ior_glass_grad=True, transform=tf(0.0, trans_grad=True)),
        ns.CylSingletLens(c1=0.01, c2=-0.015, height=8.0, width=8.0, t=2.0, ior_glass=1.6, c1_grad=True, transform=tf(10.0)),
        ns.DoubletLens(c1=0.02, c2=-0.03, c3=-0.01, d=10.0, t1=3.0, t2=1.5, ior_glass1=1.5, ior_glass2=1.7, transform=tf(20.0)),
        ns.TripletLens(c1=0.02, c2=-0.03, c3=0.01, c4=-0.02, d=10.0, t1=2.0, t2=1.0, t3=2.0, ior_glass1=1.5, ior_glass2=1.6,
                       ior_glass3=1.7, transform=tf(30.0)),
        ns.IdealThinLens(focal=40.0, focal_grad=True, diameter=10.0, transform=tf(40.0)),
        ns.CircularAperture(3.0, transform=tf(45.0)),
        ns.SphericalMirror(c1=-0.01, d=0.0, diameter=20.0, c1_grad=True, transform=tf(50.0)),
        ns.CylindricalMirror(c1=-0.01, d=0.0, transform=tf(60.0)),
        ns.ParabolicMirror(c1=0.005, d=0.0, transform=tf(70.0)),
        ns.ParabolicMirrorXZ(c1=0.004, d=0.0, transform=tf(80.0)),
    ]


def trainable_bundle(ns, case):
    """A disk source whose pose is trainable, sized for the tilt system ('seq') or the folded system ('naive')."""
    if case == "seq":
        return ns.CollimatedDisk(radius=3.5, ray_id=0, transform=ns.RayTransformBundle(
            translation=[0.1, -0.2, -30.0], rotation=[0.02, 0.01, 0.0], trans_grad=True, rot_grad=True))
    return ns.CollimatedDisk(radius=0.03, ray_id=0, transform=ns.RayTransformBundle(
        translation=[0.0, 0.0, -0.3], rotation=[0.01, -0.01, 0.0], trans_grad=True, rot_grad=True))


def folded_gradients(lens, mirror, sensor):
    s0, s1 = lens.shape.surfaces[0], lens.shape.surfaces[1]
    return dict(lens_rot=lens.shape.transform.rot_vec.grad, lens_trans=lens.shape.transform.trans.grad,
                c1=s0.c.grad, c2=s1.c.grad, t1=s1.transform.trans.grad, ior_glass=lens.ior_glass.grad,
                mirror_c=mirror.shape.c.grad, mirror_rot=mirror.shape.transform.rot_vec.grad,
                mirror_trans=mirror.shape.transform.trans.grad, sensor_trans=sensor.shape.transform.trans.grad)


def tilt_gradients(lens, sensor):
    s0, s1 = lens.shape.surfaces[0], lens.shape.surfaces[1]
    return dict(lens_rot=lens.shape.transform.rot_vec.grad, lens_trans=lens.shape.transform.trans.grad,
                c1=s0.c.grad, c2=s1.c.grad, t1=s1.transform.trans.grad, ior_glass=lens.ior_glass.grad,
                ior_media=lens.ior_media.grad, sensor_rot=sensor.shape.transform.rot_vec.grad,
                sensor_trans=sensor.shape.transform.trans.grad)


# name -> (elements builder, rays builder, n_rays)
def sequential_cases(ns):
    hp = math.pi / 2
    return {
        "singlet": (lambda: [singlet(ns)], lambda n: disk_bundle(ns).sample(n), 4096),
        "singlet_tilted_bundle": (lambda: [singlet(ns)],
                                  lambda n: disk_bundle(ns, rot=[-0.087, 0.0, 0.0]).sample(n), 4096),
        "singlet_decentred_bundle": (lambda: [singlet(ns)],
                                     lambda n: disk_bundle(ns, radius=3.0, x=1.5, y=-2.5).sample(n), 4096),
        "singlet_posed": (lambda: [singlet(ns, transform=T(ns, 2.0, x=0.3, y=-0.2, rot=[0.05, -0.03, 0.4]))],
                          lambda n: disk_bundle(ns, radius=4.5).sample(n), 4096),
        "singlet_plano": (lambda: [singlet(ns, c1=0.0, c2=-0.08)], lambda n: disk_bundle(ns).sample(n), 4096),
        "singlet_clear_edge": (lambda: [singlet(ns, inked=False)],
                               lambda n: disk_bundle(ns, radius=6.0, rot=[0.12, 0.0, 0.0], y=3.0).sample(n), 4096),
        "singlet_tir": (lambda: [singlet(ns, c1=0.09, c2=-0.09, t=4.0, ior_glass=1.9)],
                        lambda n: ns.PointSource(NA=0.95, ray_id=2, transform=ns.RayTransformBundle(
                            translation=[0.0, 0.0, 0.0])).sample(n), 4096),
        "singlet_sensor": (lambda: list(spot_system(ns, grads=False)), lambda n: disk_bundle(ns).sample(n), 4096),
        "system5": (lambda: system5(ns), lambda n: disk_bundle(ns).sample(n), 4096),
        "doublet": (lambda: [ns.DoubletLens(c1=0.04, c2=-0.03, c3=-0.01, d=12.0, t1=3.0, t2=1.5,
                                            ior_glass1=1.5, ior_glass2=1.7, transform=T(ns, 1.0)),
                             ns.Sensor(ns.Disk(10.0, transform=T(ns, 30.0)))],
                    lambda n: disk_bundle(ns, radius=5.0, rot=[0.02, 0.03, 0.0]).sample(n), 4096),
        "triplet": (lambda: [ns.TripletLens(c1=0.05, c2=-0.02, c3=0.01, c4=-0.04, d=12.0, t1=2.0, t2=1.0, t3=2.5,
                                            ior_glass1=1.5, ior_glass2=1.65, ior_glass3=1.55)],
                    lambda n: disk_bundle(ns, radius=5.5).sample(n), 2048),
        "dead_rays": (lambda: [singlet(ns), ns.Sensor(ns.Disk(50.0, transform=T(ns, 12.0)))],
                      lambda n: disk_bundle(ns, radius=6.5).sample(n), 2048),
        "aperture_plain": (lambda: [ns.CircularAperture(2.0, invert=False, transform=T(ns, 1.0)),
                                    ns.Sensor(ns.Disk(30.0, transform=T(ns, 6.0)))],
                           lambda n: disk_bundle(ns).sample(n), 1024),
        "aperture_invert": (lambda: [ns.CircularAperture(2.0, invert=True, transform=T(ns, 1.0)),
                                     ns.RectangularAperture(1.5, 0.5, invert=True, transform=T(ns, 2.0, rot=[0.0, 0.0, 0.3])),
                                     ns.EllipticAperture(1.2, 0.4, rot=0.5, invert=True, transform=T(ns, 3.0)),
                                     ns.Sensor(ns.Disk(30.0, transform=T(ns, 6.0)))],
                            lambda n: disk_bundle(ns).sample(n), 2048),
        "mirrors": (lambda: [ns.ParabolicMirror(c1=-0.02, d=0.0, transform=T(ns, 20.0)),
                             ns.CylindricalMirror(c1=0.03, d=0.0, transform=T(ns, -5.0, rot=[0.0, 0.1, 0.0])),
                             ns.ParabolicMirrorXZ(c1=-0.01, d=0.0, transform=T(ns, 25.0))],
                    lambda n: disk_bundle(ns, z=-2.0).sample(n), 2048),
        # 2f -> 2f imaging through an ideal thin lens (tests/test_ideal.py:55-107 of the reference), tilted lens
        "ideal_thin_lens": (lambda: [ns.IdealThinLens(focal=50.0, diameter=40.0, transform=T(ns, 0.0, rot=[0.02, -0.01, 0.0])),
                                     ns.Sensor(ns.Disk(30.0, transform=T(ns, 100.0)))],
                            lambda n: ns.PointSource(NA=0.15, ray_id=1, transform=ns.RayTransformBundle(
                                translation=[0.5, -0.3, -100.0])).sample(n), 2048),
        "ideal_mirror": (lambda: [ns.IdealMirror(radius_x=80.0, radius_y=120.0, transform=T(ns, 5.0)),
                                  ns.Sensor(ns.Disk(30.0, transform=T(ns, 25.0)))],
                         lambda n: disk_bundle(ns, radius=5.0, rot=[0.03, 0.0, 0.0]).sample(n), 1024),
        # convex polyhedron as a traced shape: a posed reflecting box, then a refracting one
        "box": (lambda: [ns.ElementCustom(ns.Box(length=6.0, width=4.0, height=5.0,
                                                 transform=T(ns, 2.0, x=0.5, rot=[0.3, -0.4, 0.1])), ns.Reflect(), None),
                         ns.ElementCustom(ns.Box(length=3.0, width=8.0, height=8.0, transform=T(ns, -12.0, rot=[0.0, 0.2, 0.0])),
                                          ns.RefractSnell(1.0, 1.5), None)],
                lambda n: disk_bundle(ns, radius=4.5, rot=[0.05, -0.02, 0.0]).sample(n), 2048),
        "cyl_singlet": (lambda: [ns.CylSingletLens(c1=0.04, c2=-0.03, height=12.0, width=10.0, t=2.5,
                                                   ior_glass=1.5, inked=False, transform=T(ns, 0.0, rot=[0.0, 0.0, 0.2]))],
                        lambda n: disk_bundle(ns, radius=7.5, rot=[0.05, 0.02, 0.0]).sample(n), 4096),
    }


def primitive_cases(ns):
    """Bare surfaces exercised through Surface.intersectTest / Surface.forward."""
    hp = math.pi / 2
    return {
        "plane_c1": (lambda: ns.Plane(ns.RayTransform(translation=torch.zeros(3))),
                     lambda: gaussian_rays(ns, 1024)),
        "plane_posed": (lambda: ns.Plane(T(ns, 3.0, x=1.0, rot=[0.3, -0.2, 0.1])), lambda: gaussian_rays(ns, 512)),
        "sphere": (lambda: ns.Sphere(radius=torch.tensor(10.0)), lambda: gaussian_rays(ns, 512, z=-20.0, spread=6.0)),
        "cylinder": (lambda: ns.Cylinder(5.0, transform=T(ns, 0.0, rot=[hp, 0.0, 0.0])),
                     lambda: gaussian_rays(ns, 512, z=-20.0, spread=4.0)),
        "quadric_parabola": (lambda: ns.Quadric(c=torch.tensor(0.1), k=torch.tensor(-1.0)),
                             lambda: gaussian_rays(ns, 512, spread=3.0)),
        "quadric_hyperbola": (lambda: ns.Quadric(c=0.08, k=-2.5, transform=T(ns, 1.0)),
                              lambda: gaussian_rays(ns, 512, spread=3.0, slope=0.3)),
        "quadric_zy": (lambda: ns.QuadricZY(c=0.07, k=-0.5, transform=T(ns, 0.5, rot=[0.0, 0.0, 0.7])),
                       lambda: gaussian_rays(ns, 512, spread=3.0, slope=0.2)),
        "cone": (lambda: ns.Cone(slope=0.5, transform=T(ns, 2.0)), lambda: gaussian_rays(ns, 512, spread=3.0, slope=0.3)),
        "single_cone": (lambda: ns.SingleCone(slope=-0.7, transform=T(ns, 2.0)),
                        lambda: gaussian_rays(ns, 512, spread=3.0, slope=0.3)),
        "half_sphere": (lambda: ns.HalfSphere(curvature=0.1, curvature_grad=False, transform=T(ns, -1.0)),
                        lambda: gaussian_rays(ns, 512, spread=5.0, slope=0.4)),
        "bounded_half_sphere": (lambda: ns.BoundedHalfSphere(curvature=-0.05, diameter=8.0, transform=T(ns, 4.0)),
                                lambda: gaussian_rays(ns, 512, spread=3.0, slope=0.2)),
        "half_cyl": (lambda: ns.HalfCyl(curvature=0.09, curvature_grad=False, transform=T(ns, 0.0)),
                     lambda: gaussian_rays(ns, 512, spread=5.0, slope=0.4)),
        "disk_inverted": (lambda: ns.Disk(1.0, invert=True, transform=T(ns, 0.0)), lambda: gaussian_rays(ns, 256)),
        "rectangle": (lambda: ns.Rectangle(1.0, 0.5, transform=T(ns, 0.0, rot=[0.0, 0.0, 0.3])),
                      lambda: gaussian_rays(ns, 256)),
        "ellipse": (lambda: ns.Ellipse(1.5, 0.5, 0.4, transform=T(ns, 0.0)), lambda: gaussian_rays(ns, 256)),
    }


def kat_rays(ns):
    """Known-answer inputs of the reference's tests/test_primitive.py."""
    R = ns.Rays.initialize
    return {
        "kat_plane": (lambda: ns.Plane(ns.RayTransform(translation=torch.zeros(3))),
                      lambda: R([[0, 0, -10], [1, 1, -5], [-2, -2, -10]], [[0, 0, 1]] * 3)),
        "kat_sphere": (lambda: ns.Sphere(radius=torch.tensor(10.0)),
                       lambda: R([[0, 0, 0], [0, 0, -20], [0, 15, 0]], [[0, 0, 1], [0, 0, 1], [0, 1, 0]])),
        "kat_parabola": (lambda: ns.Quadric(c=torch.tensor(0.1), k=torch.tensor(-1.0)),
                         lambda: R([[0, 2, -10], [0, 5, -10]], [[0, 0, 1], [0, 0, 1]])),
        "kat_implicit": (lambda: ns.Quadric(c=torch.tensor(0.1), k=torch.tensor(0.0)),
                         lambda: R([[0, 0, -20], [1, 1, -20], [5, 5, -20], [0.1, 8, -20]],
                                   [[0, 0, 1], [0, 0, 1], [0, 0.1, 1], [0, 0.1, 1]])),
    }


def random_system(ns, rng):
    """The same random constructor arguments for either namespace."""
    els = []
    z = 0.0
    for _ in range(int(rng.integers(1, 4))):
        kind = rng.choice(["singlet", "singlet", "doublet", "stop", "mirror", "cyl"])
        rot = [float(x) for x in rng.normal(0, 0.03, 3)] if rng.random() < 0.5 else None
        tr = ns.RayTransform(translation=[float(rng.normal(0, 0.2)), float(rng.normal(0, 0.2)), z], rotation=rot)
        if kind == "singlet":
            els.append(ns.SingletLens(c1=float(rng.uniform(-0.06, 0.06)), c2=float(rng.uniform(-0.06, 0.06)), d=12.0,
                                      t=float(rng.uniform(3.0, 5.0)), ior_glass=float(rng.uniform(1.4, 1.9)),
                                      inked=bool(rng.random() < 0.5), transform=tr))
        elif kind == "doublet":
            els.append(ns.DoubletLens(c1=float(rng.uniform(0.0, 0.05)), c2=float(rng.uniform(-0.04, 0.0)),
                                      c3=float(rng.uniform(-0.03, 0.01)), d=12.0, t1=3.0, t2=2.0,
                                      ior_glass1=1.5, ior_glass2=float(rng.uniform(1.55, 1.8)), transform=tr))
        elif kind == "stop":
            els.append(ns.CircularAperture(float(rng.uniform(2.0, 4.0)), invert=True, transform=tr))
        elif kind == "mirror":
            els.append(ns.SphericalMirror(c1=float(rng.uniform(-0.03, 0.03)), d=0.0, diameter=20.0, transform=tr))
        else:
            els.append(ns.CylSingletLens(c1=float(rng.uniform(0.0, 0.05)), c2=float(rng.uniform(-0.05, 0.0)), height=12.0,
                                         width=11.0, t=3.0, ior_glass=1.6, inked=bool(rng.random() < 0.5), transform=tr))
        z += float(rng.uniform(6.0, 10.0))
    els.append(ns.Sensor(ns.Disk(15.0, transform=ns.RayTransform(translation=[0.0, 0.0, z]))))
    return els


def random_system_extra(ns, rng):
    """Random systems over the kinds random_system does not draw: ideal thin lenses, reflecting / refracting
    boxes, triplets, rectangular and elliptic stops, parabolic and cylindrical mirrors."""
    els = []
    z = 0.0
    for _ in range(int(rng.integers(1, 4))):
        kind = rng.choice(["ideal", "box_reflect", "box_refract", "triplet", "rect_stop", "ellipse_stop", "parabolic", "cyl_mirror"])
        rot = [float(x) for x in rng.normal(0, 0.03, 3)] if rng.random() < 0.5 else None
        tr = ns.RayTransform(translation=[float(rng.normal(0, 0.2)), float(rng.normal(0, 0.2)), z], rotation=rot)
        if kind == "ideal":
            els.append(ns.IdealThinLens(focal=float(rng.uniform(20.0, 80.0)) * (1 if rng.random() < 0.7 else -1),
                                        diameter=float(rng.uniform(8.0, 14.0)), transform=tr))
        elif kind in ("box_reflect", "box_refract"):
            box = ns.Box(length=float(rng.uniform(2.0, 4.0)), width=float(rng.uniform(6.0, 14.0)),
                         height=float(rng.uniform(6.0, 14.0)), transform=tr)
            fn = ns.Reflect() if kind == "box_reflect" else ns.RefractSnell(1.0, float(rng.uniform(1.3, 1.8)))
            els.append(ns.ElementCustom(box, fn, None))
        elif kind == "triplet":
            els.append(ns.TripletLens(c1=float(rng.uniform(0.0, 0.05)), c2=float(rng.uniform(-0.03, 0.0)),
                                      c3=float(rng.uniform(0.0, 0.02)), c4=float(rng.uniform(-0.05, 0.0)), d=12.0,
                                      t1=2.0, t2=1.0, t3=2.5, ior_glass1=1.5, ior_glass2=float(rng.uniform(1.55, 1.75)),
                                      ior_glass3=1.55, transform=tr))
        elif kind == "rect_stop":
            els.append(ns.RectangularAperture(float(rng.uniform(1.5, 4.0)), float(rng.uniform(1.0, 3.0)), invert=True, transform=tr))
        elif kind == "ellipse_stop":
            els.append(ns.EllipticAperture(float(rng.uniform(2.5, 4.0)), float(rng.uniform(1.0, 2.0)), rot=float(rng.uniform(0, 1.5)),
                                           invert=True, transform=tr))
        elif kind == "parabolic":
            els.append(ns.ParabolicMirror(c1=float(rng.uniform(-0.03, 0.03)), d=0.0, transform=tr))
        else:
            els.append(ns.CylindricalMirror(c1=float(rng.uniform(-0.03, 0.03)), d=0.0, transform=tr))
        z += float(rng.uniform(6.0, 10.0))
    els.append(ns.Sensor(ns.Disk(15.0, transform=ns.RayTransform(translation=[0.0, 0.0, z]))))
    return els


def make_all_trainable(els):
    """requires_grad on every floating-point parameter of the elements (tolerances excepted); returns
    {name: parameter} with names that are the same in the reference and in the package."""
    import torch.nn as nn
    named = dict(nn.ModuleList(els).named_parameters())
    out = {}
    for name, p in named.items():
        if p.is_floating_point() and "epsilon" not in name:
            p.requires_grad_(True)
            out[name] = p
    return out


def fuzz_objective(ns, els, rays, wp, wd, ws):
    """Random projection of the traced rays and of the sensor hits (the last element is the sensor)."""
    r = ns.SequentialScene(els).simulate(rays)
    obj = (r.pos * wp).sum() + (r.dir * wd).sum()
    if els[-1].hitLocs:
        locs, w, _ = els[-1].getHitsTensors()
        obj = obj + (locs * ws[: locs.shape[0]] * w[:, None]).sum()
    return obj


def fuzz_step_objective(ns, els, rays, wp, wd, ws):
    """One nearest-hit bounce (Scene.step) of the same random systems; projection of the stepped rays and of what the
    sensor recorded, if it was the nearest surface of any ray."""
    scene = ns.Scene()
    for el in els:
        scene.add_element(el)
    scene._build_index_maps()
    scene.rays = rays
    scene.step()
    r = scene.rays
    obj = (r.pos * wp).sum() + (r.dir * wd).sum()
    if els[-1].hitLocs:
        locs, w, _ = els[-1].getHitsTensors()
        obj = obj + (locs * ws[: locs.shape[0]] * w[:, None]).sum()
    return obj


# ---------------------------------------------------------------------------------------------
# ray-source cases (rays/bundle.py)
# ---------------------------------------------------------------------------------------------
def bundle_cases(ns):
    """name -> (constructor, device-sampler kind, parameters, number of uniform draws per ray[, panel description:
    dict(area=, emit=, epar=)])."""
    import math
    pose = lambda: ns.RayTransformBundle(rotation=[0.1, -0.2, 0.05], translation=[1.0, -2.0, -30.0])
    return {
        "pencil": (lambda: ns.Bundle(ray_id=3, transform=pose()), 0, lambda b: (), 0),
        "disk": (lambda: ns.CollimatedDisk(radius=4.0, ray_id=1, transform=pose()), 1,
                 lambda b: (0.0, float(b.radius2), 0.0, 2 * math.pi), 2),
        "disk_unposed": (lambda: ns.CollimatedDisk(radius=2.5, ray_id=0), 1,
                         lambda b: (0.0, float(b.radius2), 0.0, 2 * math.pi), 2),
        "line": (lambda: ns.CollimatedLine(length=3.0, ray_id=2, transform=pose()), 2, lambda b: (-3.0, 3.0), 1),
        "fan": (lambda: ns.Fan(angle=0.6, ray_id=4, transform=pose()), 3, lambda b: (-0.3, 0.3), 1),
        "point": (lambda: ns.PointSource(NA=0.25, ray_id=5, transform=pose()), 4,
                  lambda b: (0.0, float(b.F_phi_max), 0.0, 2 * math.pi), 2),
        # normals instead of uniforms: the golden keeps the reference's standard-normal draws
        "gauss": (lambda: ns.GaussianBeam(2.0, 1.2, ray_id=6, transform=pose()), 5, lambda b: (0.5, 0.3), 2),
        "rayleigh": (lambda: ns.RayleighScatter(ray_id=7, transform=pose()), 6, lambda b: (), 2),
        "panel_rect": (lambda: ns.RectangularPanel(3.0, 1.0, ray_id=8, transform=pose()), 7,
                       lambda b: (-1.5, 1.5, -0.5, 0.5), 4, dict(area=0, emit=0, epar=())),
        "panel_ring_cone": (lambda: ns.RingSource(1.0, 2.0, ray_id=9, transform=pose(), emission=ns.SolidAngleEmission(0.4)),
                            7, lambda b: (1.0, 4.0, 0.0, 2 * math.pi), 4,
                            dict(area=1, emit=1, epar=(0.0, (1 - math.cos(0.4)) / math.pi, 0.0, 2 * math.pi))),
        "sphere": (lambda: ns.LambertianSphere(2.0, ray_id=10, transform=pose()), 8, lambda b: (2.0,), 4),
    }


# ---------------------------------------------------------------------------------------------
# physics functors called on their own (phys/std.py): name -> (constructor, inputs(n) -> (hit_local, dir, normal))
# ---------------------------------------------------------------------------------------------
def physics_cases(ns):
    def unit(v):
        return v / v.norm(dim=1, keepdim=True)

    def inputs(n):
        hl = torch.randn(n, 3) * 3.0
        d = unit(torch.randn(n, 3) * torch.tensor([1.0, 1.0, 0.6]) + torch.tensor([0.0, 0.0, 0.8]))
        nrm = unit(torch.randn(n, 3) * 0.5 + torch.tensor([0.0, 0.0, -1.0]))
        return hl, d, nrm

    return {
        "fresnel_air_glass": (lambda: ns.RefractFresnel(ior_in=1.0, ior_out=1.5), inputs),
        "fresnel_glass_air": (lambda: ns.RefractFresnel(ior_in=1.5, ior_out=1.0), inputs),   # total internal reflection occurs
        "linear_posed": (lambda: ns.Linear(Cx=-0.02, Cy=-0.03, Dx=1.0, Dy=0.9,
                                           transform=ns.RayTransform(rotation=[0.1, 0.05, -0.2])), inputs),
        "linear_plain": (lambda: ns.Linear(Cx=-0.05, Cy=-0.05), inputs),
    }


# ---------------------------------------------------------------------------------------------
# renderer cases (render/camera.py): name -> (elements, camera keyword arguments)
# ---------------------------------------------------------------------------------------------
def render_cases(ns):
    cam = dict(position=(30.0, 20.0, -40.0), look_at=(0.0, 0.0, 15.0), up_vector=(0.0, 1.0, 0.0), fov_deg=40.0,
               width=96, height=64)
    return {
        "system5": (lambda: system5(ns), cam),
        "box_and_lens": (lambda: [ns.ElementCustom(ns.Box(length=6.0, width=4.0, height=5.0,
                                                          transform=T(ns, 20.0, x=6.0, rot=[0.3, -0.4, 0.1])), ns.Block(), None),
                                  singlet(ns, inked=False), ns.CylSingletLens(c1=0.04, c2=-0.03, height=12.0, width=10.0, t=2.5,
                                                                             ior_glass=1.7, inked=True, transform=T(ns, 10.0))],
                         dict(cam, position=(-14.0, 10.0, -18.0), look_at=(2.0, 0.0, 10.0), fov_deg=60.0)),
        # the scene of the reference's static render script (tests/render_static.py:23-66): elements assembled by hand
        # around bare spheres, a blocking box, an orbit camera, a grey background
        "spheres_and_box": (lambda: hand_built_elements(ns),
                            dict(position=(-30.0, 0.0, 0.0), look_at=(0.0, 0.0, 0.0), up_vector=(0.0, 1.0, 0.0), fov_deg=40.0,
                                 width=96, height=96),
                            dict(camera="OrbitCamera", background_color=(0.9, 0.9, 0.9))),
    }


def hand_built_elements(ns):
    e1 = ns.Element()
    e1.shape = ns.Sphere(radius=2.0)
    e1.surface_functions.append(ns.Reflect())
    e2 = ns.Element()
    e2.shape = ns.Sphere(radius=2.0, transform=ns.RayTransform(translation=[0, -5, 0]))
    e2.surface_functions.append(ns.Reflect())
    e3 = ns.Element()
    e3.shape = ns.Box(height=7, width=7, length=7, transform=ns.RayTransform(translation=[0, 0, -20]))
    for _ in range(len(e3.shape)):
        e3.surface_functions.append(ns.Block())
    return [e1, e2, e3]


def render_setup(ns, name, device=None):
    """(scene, camera, renderer) of a render case; elements go in through `scene.elements.append` like in the
    reference's script."""
    case = render_cases(ns)[name]
    mk_elements, cam_kw = case[0], case[1]
    opts = case[2] if len(case) > 2 else {}
    scene = ns.Scene()
    for el in mk_elements():
        scene.elements.append(el)
    scene._build_index_maps()
    kw = dict(cam_kw) if device is None else dict(cam_kw, device=device)
    cam = getattr(ns, opts.get("camera", "Camera"))(**kw)
    rend = ns.Renderer(scene, background_color=opts["background_color"]) if "background_color" in opts else ns.Renderer(scene)
    return scene, cam, rend


# ---------------------------------------------------------------------------------------------
# bare lens shapes scanned by a row of parallel rays (the constructions of the reference's profile scripts,
# tests/test_spherics.py:108-146 and tests/test_cylindrics.py:106-122: tensor-valued arguments, a shape rotated by
# 90 degrees, a box rotated by 45 degrees)
# ---------------------------------------------------------------------------------------------
def profile_shapes(ns):
    t = torch.tensor
    out = [
        ("biconvex", ns.Singlet(C1=1 / t(50.0), C2=1 / t(-50.0), D=t(25.0), T=t(10.0)), 30.0),
        ("biconvex_flipped", ns.Singlet(C1=1 / t(50.0), C2=1 / t(-50.0), D=t(25.0), T=t(10.0),
                                        transform=ns.RayTransform(rotation=t([0, math.pi / 2, 0]))), 30.0),
        ("meniscus", ns.Singlet(C1=1 / t(30.0), C2=1 / t(100.0), D=t(20.0), T=t(5.0)), 24.0),
        ("plano", ns.Singlet(C1=t(0.0), C2=1 / t(-40.0), D=t(25.0), T=t(8.0)), 30.0),
        ("half_sphere", ns.Singlet(C1=t(0.0), C2=1 / t(-12.6), D=t(25.0), T=t(12.7)), 30.0),
        ("doublet", ns.Doublet(C1=1 / t(50.0), C2=1 / t(-50.0), C3=1 / t(-200.0), T1=t(10.0), T2=t(5.0), D=t(25.0)), 30.0),
        ("triplet", ns.Triplet(C1=1 / t(40.0), C2=1 / t(30.0), C3=1 / t(30.0), C4=1 / t(-40.0),
                               T1=t(8.0), T2=t(4.0), T3=t(8.0), D=t(20.0)), 24.0),
        ("box45", ns.Box(length=t(1.0), width=t(1.0), height=t(1.0),
                         transform=ns.RayTransform(rotation=t([0, math.pi / 4, 0]))), 3.0),
        ("cyl_biconvex", ns.CylSinglet(C1=1 / t(50.0), C2=1 / t(-50.0), width=t(30.0), height=t(20.0), T=t(10.0)), 45.0),
        ("cyl_plano", ns.CylSinglet(C1=1 / t(30.0), C2=t(0.0), width=t(20.0), height=t(25.0), T=t(8.0)), 37.5),
        ("cyl_biconcave", ns.CylSinglet(C1=t(1 / -50.0), C2=t(1 / 50.0), width=t(40.0), height=t(20.0), T=t(2.0)), 60.0),
    ]
    return out


def profile_rays(ns, extent, axis, n=200):
    """`n` rays along +z from z = -100, spread over `extent` along x (axis 0) or y (axis 1)."""
    coords = torch.linspace(-extent / 2, extent / 2, n)
    pos = torch.zeros(n, 3)
    pos[:, axis] = coords
    pos[:, 2] = -100.0
    d = torch.tensor([[0.0, 0.0, 1.0]]).expand(n, 3).contiguous()
    return ns.Rays.initialize(pos, d)
