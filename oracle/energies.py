"""oracle/energies.py — TEST INFRASTRUCTURE ONLY (never imported by the product path stark_amd/).

CPU restatement of the reference's energy *definitions*, one function per potential registry key. Every function
receives `b`: the potential's bound inputs in the reference's binding order (the order of the `mws.make_*` calls in
the cited constructor), each entry a list of `stride` scalars (oracle.ad.D2 for DoFs, ndarray[E] otherwise), and returns
the per-element energy as a D2 (value, gradient, Hessian w.r.t. the element's velocity DoFs).

All DoFs are next-step velocities: x1 = x0 + dt*v1 (stark/src/models/time_integration.cpp:3-11).
"""
from __future__ import annotations

import numpy as np

from . import ad
from .ad import D2, add, cross, det3, dot, frob_sq, inv2, inv3, matmul, norm, normalized, scale, sqnorm, sub, trace, transpose, where


def _x1(x0, v1, dt):
    return [x0[i] + dt * v1[i] for i in range(3)]


def _cols_to_matrix(cols):
    """Matrix whose COLUMNS are the given vectors (reference: Matrix(collect_scalars({..}), {n,3}).transpose())."""
    return [[cols[j][i] for j in range(len(cols))] for i in range(len(cols[0]))]


def _identity(n):
    return [[1.0 if i == j else 0.0 for j in range(n)] for i in range(n)]


def _msub(A, B):
    return [[A[i][j] - B[i][j] for j in range(len(A[0]))] for i in range(len(A))]


def _mscale(s, A):
    return [[s * A[i][j] for j in range(len(A[0]))] for i in range(len(A))]


# ----------------------------------------------------------------------------------------------------------------------
# stark/src/models/deformables/point/EnergyLumpedInertia.cpp:12-49
# bindings: v1*, x0, v0, a, f, volume, density, damping, is_quasistatic, dt, gravity
def EnergyLumpedInertia(b):
    v1, x0, v0, a, f, (volume,), (density,), (damping,), (is_quasistatic,), (dt,), gravity = b
    mass = volume * density
    x1 = _x1(x0, v1, dt)
    xhat = [x0[i] + dt * v0[i] for i in range(3)]
    dev = sub(x1, xhat)
    dev2 = sub(x1, x0)
    E_inertia = 0.5 * mass * (dot(dev, dev) / (dt ** 2) + dot(dev2, dev2) * damping / dt)
    f_ext = [mass * (a[i] + gravity[i]) + f[i] for i in range(3)]
    E_ext = -dot(f_ext, x1)
    return E_ext + where(is_quasistatic > 0.5, 0.0, E_inertia)


# stark/src/models/deformables/point/EnergyPrescribedPositions.cpp:15-32
# bindings: v1*, x0, x1_prescribed, k, dt
def EnergyPrescribedPositions(b):
    v1, x0, target, (k,), (dt,) = b
    x1 = _x1(x0, v1, dt)
    return 0.5 * k * sqnorm(sub(x1, target))


# ----------------------------------------------------------------------------------------------------------------------
def _stable_neohookean_density(F1, e, nu):
    # stark/src/models/deformables/volume/EnergyTetStrain.cpp:50-61
    mu = e / (2.0 * (1.0 + nu))
    lam = (e * nu) / ((1.0 + nu) * (1.0 - 2.0 * nu))
    mu_ = 4.0 / 3.0 * mu
    lam_ = lam + 5.0 / 6.0 * mu
    detF = det3(F1)
    Ic = frob_sq(F1)
    alpha = 1.0 + mu_ / lam_ - mu_ / (4.0 * lam_)
    return 0.5 * mu_ * (Ic - 3.0) + 0.5 * lam_ * (detF - alpha).powN(2) - 0.5 * mu_ * (Ic + 1.0).log()


# stark/src/models/deformables/volume/EnergyTetStrain.cpp:12-78
# bindings: v1[4]*, x0[4], X[4], scale, e, nu, strain_limit, strain_limit_stiffness, damping, dt
def EnergyTetStrain(b):
    v1, x0, X = b[0:4], b[4:8], b[8:12]
    (scale_,), (e,), (nu,), (strain_limit,), (sl_k,), (damping,), (dt,) = b[12:19]
    x1 = [_x1(x0[i], v1[i], dt) for i in range(4)]
    Xs = [scale(scale_, X[i]) for i in range(4)]
    DX = _cols_to_matrix([sub(Xs[1], Xs[0]), sub(Xs[2], Xs[0]), sub(Xs[3], Xs[0])])
    DXinv = inv3(DX)
    Dx1 = _cols_to_matrix([sub(x1[1], x1[0]), sub(x1[2], x1[0]), sub(x1[3], x1[0])])
    F1 = matmul(Dx1, DXinv)
    I = _identity(3)
    E1 = _mscale(0.5, _msub(matmul(transpose(F1), F1), I))
    vol = det3(DX) / 6.0
    Dx0 = _cols_to_matrix([sub(x0[1], x0[0]), sub(x0[2], x0[0]), sub(x0[3], x0[0])])
    F0 = matmul(Dx0, DXinv)
    E0 = _mscale(0.5, _msub(matmul(transpose(F0), F0), I))
    dE_dt = [[(E1[i][j] - E0[i][j]) / dt for j in range(3)] for i in range(3)]
    elastic = _stable_neohookean_density(F1, e, nu)
    damp = 0.5 * damping * frob_sq(dE_dt)
    trE = trace(E1)
    devE = [[E1[i][j] - (trE / 3.0) * I[i][j] for j in range(3)] for i in range(3)]
    dev_norm = frob_sq(devE).sqrt()
    largest = trE / 3.0 + np.sqrt(2.0 / 3.0) * dev_norm
    dl = largest - strain_limit
    sl = where(dl.v > 0.0, sl_k * dl.powN(3) / 3.0, 0.0)
    return vol * (elastic + damp + sl)


# stark/src/models/deformables/volume/EnergyTetStrain.cpp:80-123
# bindings: v1[4]*, x0[4], X[4], scale, e, nu, dt
def EnergyTetStrain_Elasticity_Only(b):
    v1, x0, X = b[0:4], b[4:8], b[8:12]
    (scale_,), (e,), (nu,), (dt,) = b[12:16]
    x1 = [_x1(x0[i], v1[i], dt) for i in range(4)]
    Xs = [scale(scale_, X[i]) for i in range(4)]
    DX = _cols_to_matrix([sub(Xs[1], Xs[0]), sub(Xs[2], Xs[0]), sub(Xs[3], Xs[0])])
    DXinv = inv3(DX)
    Dx1 = _cols_to_matrix([sub(x1[1], x1[0]), sub(x1[2], x1[0]), sub(x1[3], x1[0])])
    F1 = matmul(Dx1, DXinv)
    vol = det3(DX) / 6.0
    return vol * _stable_neohookean_density(F1, e, nu)


# ----------------------------------------------------------------------------------------------------------------------
def _triangle_jacobian(x):
    # stark/src/models/deformables/deformable_tools.cpp:7-21 (rest configuration projected to its own plane)
    u = normalized(sub(x[1], x[0]))
    n = cross(u, sub(x[2], x[0]))
    v = normalized(cross(u, n))
    P = [u, v]  # 2x3
    Xp = [[dot(P[r], x[k]) for r in range(2)] for k in range(3)]
    return _cols_to_matrix([sub(Xp[1], Xp[0]), sub(Xp[2], Xp[0])])  # 2x2


def _eigenvalues_sym_2x2(A):
    # stark/src/models/deformables/deformable_tools.cpp:26-36
    a, bb, c = A[0][0], A[1][1], A[0][1]
    delta = (4.0 * c.powN(2) + (a - bb).powN(2)).sqrt()
    return [0.5 * (a + bb + delta), 0.5 * (a + bb - delta)]


def _triangle_common(b, full):
    v1, x0, X = b[0:3], b[3:6], b[6:9]
    if full:
        (scale_,), (thickness,), (e,), (nu,), (damping,), (strain_limit,), (sl_k,), (inflation,), (dt,) = b[9:18]
    else:
        (scale_,), (thickness,), (e,), (nu,), (inflation,), (dt,) = b[9:15]
    x1 = [_x1(x0[i], v1[i], dt) for i in range(3)]
    Xs = [scale(scale_, X[i]) for i in range(3)]
    rest_area = 0.5 * norm(cross(sub(Xs[0], Xs[2]), sub(Xs[1], Xs[2])))
    DXinv = inv2(_triangle_jacobian(Xs))
    Dx1 = _cols_to_matrix([sub(x1[1], x1[0]), sub(x1[2], x1[0])])  # 3x2
    F1 = matmul(Dx1, DXinv)  # 3x2
    C1 = matmul(transpose(F1), F1)
    mu = e / (2.0 * (1.0 + nu))
    lam = (e * nu) / ((1.0 + nu) * (1.0 - nu))  # 2D
    area = 0.5 * norm(cross(sub(x1[0], x1[2]), sub(x1[1], x1[2])))
    J = area / rest_area
    Ic = trace(C1)
    logJ = J.log()
    elastic = 0.5 * mu * (Ic - 2.0) - mu * logJ + 0.5 * lam * logJ.powN(2)
    n0 = scale(-1.0, normalized(cross(sub(x0[1], x0[0]), sub(x0[2], x0[0]))))
    infl = inflation * dot(n0, add(add(x1[0], x1[1]), x1[2])) / 3.0
    total = elastic + infl
    if full:
        I = _identity(2)
        E1 = _mscale(0.5, _msub(C1, I))
        Dx0 = _cols_to_matrix([sub(x0[1], x0[0]), sub(x0[2], x0[0])])
        F0 = matmul(Dx0, DXinv)
        E0 = _mscale(0.5, _msub(matmul(transpose(F0), F0), I))
        dE_dt = [[(E1[i][j] - E0[i][j]) / dt for j in range(2)] for i in range(2)]
        damp = 0.5 * damping * frob_sq(dE_dt)
        s = _eigenvalues_sym_2x2(E1)
        sl = 0.0
        for i in range(2):
            dl = s[i] - strain_limit
            sl = sl + where(dl.v > 0.0, sl_k * dl.powN(3) / 3.0, 0.0)
        total = total + damp + sl
    return thickness * rest_area * total


# stark/src/models/deformables/surface/EnergyTriangleStrain.cpp:13-80
def EnergyTriangleStrain(b):
    return _triangle_common(b, True)


# stark/src/models/deformables/surface/EnergyTriangleStrain.cpp:82-129
def EnergyTriangleStrain_Elasticity_Only(b):
    return _triangle_common(b, False)


# ----------------------------------------------------------------------------------------------------------------------
_DS_EPS = 1e-12


def _dihedral(x):
    # stark/src/models/deformables/surface/EnergyDiscreteShells.cpp:12-23
    e0, e1, e2 = sub(x[1], x[0]), sub(x[2], x[0]), sub(x[3], x[0])
    n0 = cross(e0, e1)
    n1 = scale(-1.0, cross(e0, e2))
    c = (1.0 - _DS_EPS) * dot(normalized(n0), normalized(n1))
    return c.acos() if isinstance(c, D2) else np.arccos(c)


# stark/src/models/deformables/surface/EnergyDiscreteShells.cpp:26-62
# bindings: v1[4]*, x0[4], rest_angle, rest_edge_length, rest_height, scale, stiffness, damping, dt
def EnergyDiscreteShells(b):
    v1, x0 = b[0:4], b[4:8]
    (rest_angle,), (rest_len,), (rest_h,), (scale_,), (k,), (damping,), (dt,) = b[8:15]
    x1 = [_x1(x0[i], v1[i], dt) for i in range(4)]
    ratio = (rest_len * scale_) / (rest_h * scale_)
    da1 = _dihedral(x1)
    dd = da1 - rest_angle
    E_b = k * (dd * dd) * ratio
    da0 = _dihedral(x0)
    E_d = damping * 1.0 / dt * (0.5 * da1.powN(2) - da0 * da1) * ratio
    return E_b + E_d


# stark/src/models/deformables/surface/EnergyDiscreteShells.cpp:64-92
# bindings: v1[4]*, x0[4], K(4), coef, stiffness, dt
def EnergyBendingFlat(b):
    v1, x0 = b[0:4], b[4:8]
    K, (coef,), (k,), (dt,) = b[8:12]
    x1 = [_x1(x0[i], v1[i], dt) for i in range(4)]
    P = 0.0
    for d in range(3):
        x = [x1[0][d], x1[1][d], x1[2][d], x1[3][d]]
        Kx = dot(K, x)
        P = P + 0.5 * k * coef * (Kx * Kx)
    return P


# ----------------------------------------------------------------------------------------------------------------------
# Rigid bodies. stark/src/models/rigidbodies/rigidbody_transformations.cpp:54-160
def _quat_to_rotation(q):
    qw, qx, qy, qz = q
    tx, ty, tz = 2.0 * qx, 2.0 * qy, 2.0 * qz
    twx, twy, twz = tx * qw, ty * qw, tz * qw
    txx, txy, txz = tx * qx, ty * qx, tz * qx
    tyy, tyz, tzz = ty * qy, tz * qy, tz * qz
    return [[1.0 - (tyy + tzz), txy - twz, txz + twy],
            [txy + twz, 1.0 - (txx + tzz), tyz - twx],
            [txz - twy, tyz + twx, 1.0 - (txx + tyy)]]


def _quat_mul(q1, q2):
    a, b, c, d = q1
    e, f, g, h = q2
    return [a * e - b * f - c * g - d * h, b * e + a * f + c * h - d * g, a * g - b * h + c * e + d * f, a * h + b * g - c * f + d * e]


def _R1(q0, w1, dt):
    # quat_time_integration_as_rotation_matrix: normalize(q0 + 0.5 dt (0,w) * q0)
    w_ = [0.0 * w1[0], w1[0], w1[1], w1[2]]
    p = _quat_mul(w_, q0)
    q1 = [q0[i] + 0.5 * dt * p[i] for i in range(4)]
    n = (q1[0] * q1[0] + q1[1] * q1[1] + q1[2] * q1[2] + q1[3] * q1[3]).sqrt()
    return _quat_to_rotation([x / n for x in q1])


def _matvec(R, x):
    return [dot(R[i], x) for i in range(3)]


def _rb_x1(v1, w1, t0, q0, x_loc, dt):
    # RigidBodyDynamics::get_x1 (RigidBodyDynamics.cpp:46-62)
    R1 = _R1(q0, w1, dt)
    t1 = [t0[i] + dt * v1[i] for i in range(3)]
    return add(t1, _matvec(R1, x_loc))


def _rb_x0(t0, q0, x_loc):
    return add(t0, _matvec(_quat_to_rotation(q0), x_loc))


def _rb_d1(w1, q0, d_loc, dt):
    return _matvec(_R1(q0, w1, dt), d_loc)


# stark/src/models/rigidbodies/EnergyRigidBodyInertia.cpp:13-39
# bindings: v1*, v0, a, force, mass, linear_damping, is_quasistatic, dt, gravity
def EnergyRigidBodyInertia_Linear(b):
    v1, v0, a, f, (m,), (damping,), (is_q,), (dt,), gravity = b
    dev = sub(v1, v0)
    E_inertia = 0.5 * m * dot(dev, dev) + 0.5 * m * dot(v1, v1) * damping * dt
    f_ext = [m * (a[i] + gravity[i]) + f[i] for i in range(3)]
    E_ext = -1.0 * dt * dot(f_ext, v1)
    return E_ext + where(is_q > 0.5, 0.0, E_inertia)


# stark/src/models/rigidbodies/EnergyRigidBodyInertia.cpp:42-67
# bindings: w1*, w0, aa, torque, J0_glob(9), angular_damping, is_quasistatic, dt
def EnergyRigidBodyInertia_Angular(b):
    w1, w0, aa, t, J, (damping,), (is_q,), (dt,) = b
    Jm = [[J[3 * i + j] for j in range(3)] for i in range(3)]
    dev = sub(w1, w0)
    E_inertia = 0.5 * (dot(dev, _matvec(Jm, dev)) + dot(w1, _matvec(Jm, w1)) * damping * dt)
    t_ext = add(_matvec(Jm, aa), t)
    E_ext = -1.0 * dt * dot(t_ext, w1)
    return E_ext + where(is_q > 0.5, 0.0, E_inertia)


# stark/src/models/rigidbodies/EnergyRigidBodyConstraints.cpp:30-45; RigidBodyConstraints.h:110-113
def rb_constraint_global_points(b):
    loc, target, (k,), (active,), (dt,), v1, w1, t0, q0 = b
    p = _rb_x1(v1, w1, t0, q0, loc, dt)
    return 0.5 * k * sqnorm(sub(target, p)), active


# EnergyRigidBodyConstraints.cpp:47-62; RigidBodyConstraints.h:150-153
def rb_constraint_global_directions(b):
    d_loc, target, (k,), (active,), (dt,), w1, q0 = b
    d = _rb_d1(w1, q0, d_loc, dt)
    return 0.5 * k * sqnorm(sub(target, d)), active


# EnergyRigidBodyConstraints.cpp:64-80; RigidBodyConstraints.h:191-194
def rb_constraint_points(b):
    a_loc, b_loc, (k,), (active,), (dt,), v1a, w1a, t0a, q0a, v1b, w1b, t0b, q0b = b
    a1 = _rb_x1(v1a, w1a, t0a, q0a, a_loc, dt)
    b1 = _rb_x1(v1b, w1b, t0b, q0b, b_loc, dt)
    return 0.5 * k * sqnorm(sub(b1, a1)), active


def _sq_distance_point_line(p, a, b_):
    # stark/src/models/distances.cpp:61-68
    ab = sub(b_, a)
    ap = sub(p, a)
    e = dot(ap, ab)
    return dot(ap, ap) - e * e / dot(ab, ab)


# EnergyRigidBodyConstraints.cpp:82-99; RigidBodyConstraints.h:228-231
def rb_constraint_point_on_axis(b):
    a_loc, da_loc, b_loc, (k,), (active,), (dt,), v1a, w1a, t0a, q0a, v1b, w1b, t0b, q0b = b
    a1 = _rb_x1(v1a, w1a, t0a, q0a, a_loc, dt)
    da1 = _rb_d1(w1a, q0a, da_loc, dt)
    b1 = _rb_x1(v1b, w1b, t0b, q0b, b_loc, dt)
    return 0.5 * k * _sq_distance_point_line(b1, a1, add(a1, da1)), active


# EnergyRigidBodyConstraints.cpp:101-118; RigidBodyConstraints.h:265-268
def rb_constraint_distances(b):
    a_loc, b_loc, (target,), (k,), (active,), (dt,), v1a, w1a, t0a, q0a, v1b, w1b, t0b, q0b = b
    a1 = _rb_x1(v1a, w1a, t0a, q0a, a_loc, dt)
    b1 = _rb_x1(v1b, w1b, t0b, q0b, b_loc, dt)
    return 0.5 * k * (target - norm(sub(b1, a1))).powN(2), active


# EnergyRigidBodyConstraints.cpp:120-138; RigidBodyConstraints.h:305-311
def rb_constraint_distance_limits(b):
    a_loc, b_loc, (dmin,), (dmax,), (k,), (active,), (dt,), v1a, w1a, t0a, q0a, v1b, w1b, t0b, q0b = b
    a1 = _rb_x1(v1a, w1a, t0a, q0a, a_loc, dt)
    b1 = _rb_x1(v1b, w1b, t0b, q0b, b_loc, dt)
    length = norm(sub(b1, a1))
    E_min = where(length.v < dmin, k * (dmin - length).powN(2) / 2.0, 0.0)
    E_max = where(length.v > dmax, k * (length - dmax).powN(2) / 2.0, 0.0)
    return E_min + E_max, active


# EnergyRigidBodyConstraints.cpp:140-156; RigidBodyConstraints.h:357-360
def rb_constraint_directions(b):
    da_loc, db_loc, (k,), (active,), (dt,), w1a, q0a, w1b, q0b = b
    da = _rb_d1(w1a, q0a, da_loc, dt)
    db = _rb_d1(w1b, q0b, db_loc, dt)
    return 0.5 * k * sqnorm(sub(db, da)), active


# EnergyRigidBodyConstraints.cpp:158-175; RigidBodyConstraints.h:409-413
def rb_constraint_angle_limits(b):
    da_loc, db_loc, (max_distance,), (k,), (active,), (dt,), w1a, q0a, w1b, q0b = b
    da = _rb_d1(w1a, q0a, da_loc, dt)
    db = _rb_d1(w1b, q0b, db_loc, dt)
    length = norm(sub(db, da))
    return where(length.v > max_distance, k * (length - max_distance).powN(3) / 3.0, 0.0), active


# EnergyRigidBodyConstraints.cpp:177-196; RigidBodyConstraints.h:455-466
def rb_constraint_damped_spring(b):
    a_loc, b_loc, (rest,), (k,), (damping,), (active,), (dt,), v1a, w1a, t0a, q0a, v1b, w1b, t0b, q0b = b
    a1 = _rb_x1(v1a, w1a, t0a, q0a, a_loc, dt)
    b1 = _rb_x1(v1b, w1b, t0b, q0b, b_loc, dt)
    a0 = _rb_x0(t0a, q0a, a_loc)
    b0 = _rb_x0(t0b, q0b, b_loc)
    l1 = norm(sub(b1, a1))
    l0 = norm(sub(b0, a0))
    return 0.5 * k * (l1 - rest).powN(2) + 0.5 * damping * ((l1 - l0) / dt).powN(2), active


def _c1_controller(da1, va1, vb1, target_v, max_force, delay, dt):
    # RigidBodyConstraints.h:54-69
    v = dot(da1, sub(vb1, va1))
    k = max_force / delay
    eps = delay / 2.0
    dv = v - target_v
    E_c = 0.5 * k * dv.powN(2) * dt
    E_r = max_force * (dv - eps) * dt
    E_l = -1.0 * E_r
    return where(dv.v < -delay, E_l, where(dv.v < delay, E_c, E_r))


# EnergyRigidBodyConstraints.cpp:198-218
def rb_constraint_linear_velocity(b):
    da_loc, (target_v,), (max_force,), (delay,), (active,), va1, vb1, wa1, qa0, (dt,) = b
    da1 = _rb_d1(wa1, qa0, da_loc, dt)
    return _c1_controller(da1, va1, vb1, target_v, max_force, delay, dt), active


# EnergyRigidBodyConstraints.cpp:220-238
def rb_constraint_angular_velocity(b):
    da_loc, (target_w,), (max_torque,), (delay,), (active,), wa1, wb1, qa0, (dt,) = b
    da1 = _rb_d1(wa1, qa0, da_loc, dt)
    return _c1_controller(da1, wa1, wb1, target_w, max_torque, delay, dt), active


# ----------------------------------------------------------------------------------------------------------------------
# IPC contact + friction. stark/src/models/interactions/EnergyFrictionalContact.cpp
# Every potential is a sequence of symbol-getter groups (:1358-1423) read in binding order, followed by a tail.
class _Cur:
    def __init__(self, b):
        self.b = b
        self.i = 0

    def take(self, n=1):
        r = self.b[self.i:self.i + n]
        self.i += n
        return r

    def scalar(self):
        return self.take()[0][0]

    def d_x1(self, k):      # _get_d_x1 (:1393-1399): v1[k]*, x0[k], dt
        v1 = self.take(k)
        x0 = self.take(k)
        dt = self.scalar()
        return [_x1(x0[i], v1[i], dt) for i in range(k)]

    def d_v1(self, k):      # _get_d_v1 (:1389-1392)
        return self.take(k)

    def rest(self, k):      # _get_d_X / _get_rb_X
        return self.take(k)

    def rb(self, k, vel):   # _get_rb_x1 / _get_rb_v1 (:1358-1369): dt, x_loc[k], v1*, w1*, t0, q0_
        dt = self.scalar()
        xl = self.take(k)
        v1, w1, t0, q0 = self.take(4)
        R1 = _R1(q0, w1, dt)
        out = []
        for i in range(k):
            r = _matvec(R1, xl[i])
            if vel:
                out.append(add(v1, cross(w1, r)))   # global_point_velocity_in_rigid_body
            else:
                out.append(add([t0[j] + dt * v1[j] for j in range(3)], r))
        return out


def _distance_point_point(p, q):
    return sqnorm(sub(p, q)).sqrt()


def _distance_point_line(p, a, b_):
    return _sq_distance_point_line(p, a, b_).sqrt()


def _distance_point_plane(p, a, b_, c):
    # stark/src/models/distances.cpp:69-77
    n = normalized(cross(sub(a, c), sub(b_, c)))
    d = dot(sub(p, a), n)
    return (d * d).sqrt()


def _distance_line_line(a, b_, p, q):
    # stark/src/models/distances.cpp:83-89
    n = cross(sub(b_, a), sub(q, p))
    l = dot(sub(p, a), n)
    return (l * l / sqnorm(n)).sqrt()


def _barrier(d, dhat, k):
    return k * (dhat - d).powN(3) / 3.0       # cubic (:1225-1237)


def _mollifier(ea, eb, ra, rb_):
    # :1251-1259
    eps_x = 1e-3 * sqnorm(sub(ra[0], ra[1])) * sqnorm(sub(rb_[0], rb_[1]))
    x = sqnorm(cross(sub(ea[1], ea[0]), sub(eb[1], eb[0])))
    r = x / eps_x
    f = (-1.0 * r + 2.0) * r
    return where(x.v > eps_x, 1.0, f)


def _pt_contact(A, KA, B, KB, DIST):
    def f(b):
        c = _Cur(b)
        a = c.d_x1(KA) if A == "d" else c.rb(KA, False)
        bb = c.d_x1(KB) if B == "d" else c.rb(KB, False)
        dhat = c.scalar() + c.scalar()
        k = c.scalar()
        if DIST == 0:
            d = _distance_point_point(a[0], bb[0])
        elif DIST == 1:
            d = _distance_point_line(a[0], bb[0], bb[1])
        elif DIST == 2:
            d = _distance_point_plane(a[0], bb[0], bb[1], bb[2])
        elif DIST == 3:
            d = _distance_point_line(bb[0], a[0], a[1])
        else:
            d = _distance_point_plane(bb[0], a[0], a[1], a[2])
        return _barrier(d, dhat, k)
    return f


def _ee_contact(A, PA, B, PB, DIST):
    def side(c, S, P):
        e = c.d_x1(2) if S == "d" else c.rb(2, False)
        r = c.rest(2)
        p = None
        if P:
            p = (c.d_x1(1) if S == "d" else c.rb(1, False))[0]
        return e, r, p

    def f(b):
        c = _Cur(b)
        ea, ra, p = side(c, A, PA)
        eb, rb_, q = side(c, B, PB)
        k = c.scalar()
        dhat = c.scalar() + c.scalar()
        if DIST == 0:
            d = _distance_point_point(p, q)
        elif DIST == 1:
            d = _distance_point_line(p, eb[0], eb[1])
        elif DIST == 2:
            d = _distance_line_line(ea[0], ea[1], eb[0], eb[1])
        else:
            d = _distance_point_line(q, ea[0], ea[1])
        return _mollifier(ea, eb, ra, rb_) * _barrier(d, dhat, k)
    return f


def _friction(A, KA, B, KB, KIND):
    nbary = 0 if KIND == 0 else (3 if KIND in (2, 5) else 2)

    def f(b):
        c = _Cur(b)
        a = c.d_v1(KA) if A == "d" else c.rb(KA, True)
        bb = c.d_v1(KB) if B == "d" else c.rb(KB, True)
        bary = c.take()[0] if nbary else None

        def comb(pts, n):
            acc = scale(bary[0], pts[0])
            for i in range(1, n):
                acc = add(acc, scale(bary[i], pts[i]))
            return acc
        if KIND == 0:
            v = sub(bb[0], a[0])
        elif KIND == 1:
            v = sub(comb(bb, 2), a[0])
        elif KIND == 2:
            v = sub(comb(bb, 3), a[0])
        elif KIND == 3:
            va = add(a[0], scale(bary[0], sub(a[1], a[0])))
            vb = add(bb[0], scale(bary[1], sub(bb[1], bb[0])))
            v = sub(vb, va)
        elif KIND == 4:
            v = sub(comb(a, 2), bb[0])
        else:
            v = sub(comb(a, 3), bb[0])
        # _friction_potential (:1260-1278), C0
        T = c.take()[0]
        mu, fn, epsv, dt = c.scalar(), c.scalar(), c.scalar(), c.scalar()
        ut0 = dot(T[0:3], v) * dt + 1.13e-9
        ut1 = dot(T[3:6], v) * dt - 1.07e-9
        u = (ut0 * ut0 + ut1 * ut1).sqrt()
        epsu = dt * epsv
        k = mu * fn / epsu
        eps = mu * fn / (2.0 * k)
        return where(u.v < epsu, 0.5 * k * u.powN(2), mu * fn * (u - eps))
    return f


_CONTACT = {
    "contact_d_d_pt_pp_cubic": _pt_contact("d", 1, "d", 1, 0), "contact_d_d_pt_pe_cubic": _pt_contact("d", 1, "d", 2, 1),
    "contact_d_d_pt_pt_cubic": _pt_contact("d", 1, "d", 3, 2), "contact_d_d_ee_pp_cubic": _ee_contact("d", True, "d", True, 0),
    "contact_d_d_ee_pe_cubic": _ee_contact("d", True, "d", False, 1), "contact_d_d_ee_ee_cubic": _ee_contact("d", False, "d", False, 2),
    "contact_rb_rb_pt_pp_cubic": _pt_contact("rb", 1, "rb", 1, 0), "contact_rb_rb_pt_pe_cubic": _pt_contact("rb", 1, "rb", 2, 1),
    "contact_rb_rb_pt_pt_cubic": _pt_contact("rb", 1, "rb", 3, 2), "contact_rb_rb_ee_pp_cubic": _ee_contact("rb", True, "rb", True, 0),
    "contact_rb_rb_ee_pe_cubic": _ee_contact("rb", True, "rb", False, 1), "contact_rb_rb_ee_ee_cubic": _ee_contact("rb", False, "rb", False, 2),
    "contact_rb_d_pt_pp_cubic": _pt_contact("rb", 1, "d", 1, 0), "contact_rb_d_pt_pe_cubic": _pt_contact("rb", 1, "d", 2, 1),
    "contact_rb_d_pt_pt_cubic": _pt_contact("rb", 1, "d", 3, 2), "contact_rb_d_pt_ep_cubic": _pt_contact("rb", 2, "d", 1, 3),
    "contact_rb_d_pt_tp_cubic": _pt_contact("rb", 3, "d", 1, 4), "contact_rb_d_ee_pp_cubic": _ee_contact("rb", True, "d", True, 0),
    "contact_rb_d_ee_pe_cubic": _ee_contact("rb", True, "d", False, 1), "contact_rb_d_ee_ee_cubic": _ee_contact("rb", False, "d", False, 2),
    "contact_rb_d_ee_ep_cubic": _ee_contact("rb", False, "d", True, 3),
    "friction_d_d_pp_C0": _friction("d", 1, "d", 1, 0), "friction_d_d_pe_C0": _friction("d", 1, "d", 2, 1), "friction_d_d_pt_C0": _friction("d", 1, "d", 3, 2),
    "friction_d_d_ee_C0": _friction("d", 2, "d", 2, 3), "friction_rb_rb_pp_C0": _friction("rb", 1, "rb", 1, 0), "friction_rb_rb_pe_C0": _friction("rb", 1, "rb", 2, 1),
    "friction_rb_rb_pt_C0": _friction("rb", 1, "rb", 3, 2), "friction_rb_rb_ee_C0": _friction("rb", 2, "rb", 2, 3), "friction_rb_d_pp_C0": _friction("rb", 1, "d", 1, 0),
    "friction_rb_d_pe_C0": _friction("rb", 1, "d", 2, 1), "friction_rb_d_pt_C0": _friction("rb", 1, "d", 3, 2), "friction_rb_d_ee_C0": _friction("rb", 2, "d", 2, 3),
    "friction_rb_d_ep_C0": _friction("rb", 2, "d", 1, 4), "friction_rb_d_tp_C0": _friction("rb", 3, "d", 1, 5),
}

# ----------------------------------------------------------------------------------------------------------------------
# Rods. stark/src/models/deformables/line/EnergySegmentStrain.cpp:11-55 (complete), :57-88 (elasticity only)
# bindings: v1[2]*, x0[2], X[2], scale, section_radius, youngs_modulus, [strain_damping, strain_limit, strain_limit_stiffness,] dt
def _segment_strain(b, full):
    v1, x0, X = b[0:2], b[2:4], b[4:6]
    (scale_,), (radius,), (youngs,) = b[6:9]
    if full:
        (damping,), (strain_limit,), (sl_k,), (dt,) = b[9:13]
    else:
        (dt,) = b[9]
    x1 = [_x1(x0[i], v1[i], dt) for i in range(2)]
    Xs = [scale(scale_, X[i]) for i in range(2)]
    section_area = np.pi * radius ** 2
    l_rest = norm(sub(Xs[0], Xs[1]))
    l = norm(sub(x1[0], x1[1]))
    e = (l - l_rest) / l_rest
    volume = section_area * l_rest
    E = volume * youngs * e.powN(2) / 2.0
    if full:
        over = e - strain_limit
        E = E + where(over.v > 0.0, volume * sl_k * over.powN(3) / 3.0, 0.0)
        l0 = norm(sub(x0[1], x0[0]))
        e0 = (l0 - l_rest) / l_rest
        E = E + dt * damping * ((e - e0) / dt).powN(2) / 2.0
    return E


def EnergySegmentStrain(b):
    return _segment_strain(b, True)


def EnergySegmentStrain_Elasticity_Only(b):
    return _segment_strain(b, False)


# ----------------------------------------------------------------------------------------------------------------------
# Attachments. stark/src/models/interactions/EnergyAttachments.cpp:17-35, 37-60, 62-85, 87-111, 113-135
def EnergyAttachments_d_d_p_p(b):
    v1a, v1b, x0a, x0b, (k,), (dt,) = b
    return 0.5 * k * sqnorm(sub(_x1(x0b, v1b, dt), _x1(x0a, v1a, dt)))


def _comb(w, pts):
    out = scale(w[0], pts[0])
    for wi, pi in zip(w[1:], pts[1:]):
        out = add(out, scale(wi, pi))
    return out


def EnergyAttachments_d_d_p_e(b):
    v1, x0, bary, (k,), (dt,) = b[0:3], b[3:6], b[6], b[7], b[8]
    x1 = [_x1(x0[i], v1[i], dt) for i in range(3)]
    return 0.5 * k * sqnorm(sub(_comb(bary, x1[1:3]), x1[0]))


def EnergyAttachments_d_d_p_t(b):
    v1, x0, bary, (k,), (dt,) = b[0:4], b[4:8], b[8], b[9], b[10]
    x1 = [_x1(x0[i], v1[i], dt) for i in range(4)]
    return 0.5 * k * sqnorm(sub(_comb(bary, x1[1:4]), x1[0]))


def EnergyAttachments_d_d_e_e(b):
    v1, x0, bary_0, bary_1, (k,), (dt,) = b[0:4], b[4:8], b[8], b[9], b[10], b[11]
    x1 = [_x1(x0[i], v1[i], dt) for i in range(4)]
    return 0.5 * k * sqnorm(sub(_comb(bary_1, x1[2:4]), _comb(bary_0, x1[0:2])))


def EnergyAttachments_rb_d(b):
    (k,), (dt,), v1_d, x0_d, x_loc, v1, w1, t0, q0 = b
    x1_rb = _rb_x1(v1, w1, t0, q0, x_loc, dt)
    return 0.5 * k * sqnorm(sub(_x1(x0_d, v1_d, dt), x1_rb))


REGISTRY = {f.__name__: f for f in [
    EnergyLumpedInertia, EnergyPrescribedPositions, EnergyTetStrain, EnergyTetStrain_Elasticity_Only,
    EnergyTriangleStrain, EnergyTriangleStrain_Elasticity_Only, EnergyDiscreteShells, EnergyBendingFlat,
    EnergyRigidBodyInertia_Linear, EnergyRigidBodyInertia_Angular, rb_constraint_global_points, rb_constraint_global_directions,
    rb_constraint_points, rb_constraint_point_on_axis, rb_constraint_distances, rb_constraint_distance_limits, rb_constraint_directions,
    rb_constraint_angle_limits, rb_constraint_damped_spring, rb_constraint_linear_velocity, rb_constraint_angular_velocity,
    EnergySegmentStrain, EnergySegmentStrain_Elasticity_Only, EnergyAttachments_d_d_p_p, EnergyAttachments_d_d_p_e, EnergyAttachments_d_d_p_t,
    EnergyAttachments_d_d_e_e, EnergyAttachments_rb_d,
]}
REGISTRY.update(_CONTACT)
