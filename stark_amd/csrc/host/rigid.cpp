// host/rigid.cpp — rigid body state, inertia and constraint tables of the host layer (see sim.hpp).
// Reference: stark/src/models/rigidbodies/{RigidBodyDynamics,EnergyRigidBodyInertia,EnergyRigidBodyConstraints,RigidBodies,
// RigidBodyHandler}.cpp and RigidBodyConstraints.h. The energies themselves are device kernels (stark_amd/csrc/energies.hpp);
// this file owns the arrays they bind, in the reference's binding order, and the host-side callbacks.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "sim.hpp"

namespace mistark {

namespace {
struct Binder
{
    std::vector<mistark_binding> b;
    mistark_ctx* ctx;
    Stark& stark;
    Binder(mistark_ctx* c, Stark& s) : ctx(c), stark(s) {}
    int add(const double* host, int64_t n_items, int stride, int col)
    {
        const int id = mistark_array(ctx, host, n_items, stride);
        stark.check(id);
        b.push_back({id, stride, col});
        return id;
    }
    void add_id(int id, int stride, int col) { b.push_back({id, stride, col}); }
    template <std::size_t N>
    void potential(const char* name, const std::vector<std::array<int32_t, N>>& conn)
    {
        stark.check(mistark_potential(ctx, name, conn.empty() ? nullptr : conn[0].data(), (int32_t)conn.size(), (int32_t)N, b.data(), (int32_t)b.size()));
    }
};
const Vec3 ZERO = {0.0, 0.0, 0.0};
}  // namespace

// ======================================================================================================================
// RigidBodyDynamics  (RigidBodyDynamics.cpp)
// ======================================================================================================================
RigidBodyDynamics::RigidBodyDynamics(Stark& s) : stark(s)
{
    // RigidBodyDynamics.cpp:9-15: add_dof(v1, "rigid.v1"), add_dof(w1, "rigid.w1") + two callbacks
    stark.add_model(this);
    stark.callbacks->add_before_time_step([this]() { _before_time_step(); });
    stark.callbacks->add_on_time_step_accepted([this]() { _on_time_step_accepted(); });
}
int RigidBodyDynamics::add(const std::string& label)
{
    const int id = (int)t0.size();
    for (auto* v : {&t0, &t1, &v0, &v1, &w0, &w1, &a, &aa, &force, &torque}) v->push_back(ZERO);
    for (auto* q : {&q0, &q1, &q0_}) q->push_back({1.0, 0.0, 0.0, 0.0});
    R0.push_back(mat_identity());
    R1.push_back(mat_identity());
    labels.push_back(label.empty() ? "rb_" + std::to_string(id) : label);
    stark.mark_registration_dirty();
    return id;
}
Vec3 RigidBodyDynamics::get_x1(int b, const Vec3& x_loc, double dt) const
{
    // integrate_loc_point (rigidbody_transformations.cpp:38-43)
    const Mat3 R = quat_to_matrix(quat_time_integration(q0[b], w1[b], dt));
    return R * x_loc + (t0[b] + dt * v1[b]);
}
Vec3 RigidBodyDynamics::get_d1(int b, const Vec3& d_loc, double dt) const { return quat_to_matrix(quat_time_integration(q0[b], w1[b], dt)) * d_loc; }
void RigidBodyDynamics::register_dofs(mistark_ctx* ctx)
{
    const int64_t n = get_n_bodies();
    stark.check(mistark_add_dof_set(ctx, "rigid.v1", n ? v1[0].data() : nullptr, 3 * n));
    stark.check(mistark_add_dof_set(ctx, "rigid.w1", n ? w1[0].data() : nullptr, 3 * n));
    id_v1 = id_w1 = id_v0 = id_w0 = id_a = id_aa = id_force = id_torque = id_t0 = id_q0_ = -1;
    if (n == 0) return;
    stark.check(id_v1 = mistark_array(ctx, v1[0].data(), n, 3));
    stark.check(id_w1 = mistark_array(ctx, w1[0].data(), n, 3));
    stark.check(id_v0 = mistark_array(ctx, v0[0].data(), n, 3));
    stark.check(id_w0 = mistark_array(ctx, w0[0].data(), n, 3));
    stark.check(id_a = mistark_array(ctx, a[0].data(), n, 3));
    stark.check(id_aa = mistark_array(ctx, aa[0].data(), n, 3));
    stark.check(id_force = mistark_array(ctx, force[0].data(), n, 3));
    stark.check(id_torque = mistark_array(ctx, torque[0].data(), n, 3));
    stark.check(id_t0 = mistark_array(ctx, t0[0].data(), n, 3));
    stark.check(id_q0_ = mistark_array(ctx, q0_[0].data(), n, 4));
}
void RigidBodyDynamics::_before_time_step()
{
    // RigidBodyDynamics.cpp:139-151: q0_ <- q0, v1 = w1 = 0; then the (small) state goes to the device
    if (get_n_bodies() == 0) return;
    q0_ = q0;
    std::fill(v1.begin(), v1.end(), ZERO);
    std::fill(w1.begin(), w1.end(), ZERO);
    for (int id : {id_t0, id_q0_, id_v0, id_w0, id_a, id_aa, id_force, id_torque}) stark.check(mistark_upload(stark.ctx, id));
    stark.check(mistark_array_fill(stark.ctx, id_v1, 0.0));
    stark.check(mistark_array_fill(stark.ctx, id_w1, 0.0));
}
void RigidBodyDynamics::fetch_velocities()
{
    if (get_n_bodies() == 0 || !stark.ctx || id_v1 < 0) return;
    stark.check(mistark_download(stark.ctx, id_v1));
    stark.check(mistark_download(stark.ctx, id_w1));
}
void RigidBodyDynamics::_on_time_step_accepted()
{
    // RigidBodyDynamics.cpp:152-166
    if (get_n_bodies() == 0) return;
    fetch_velocities();
    const double dt = stark.dt;
    for (int i = 0; i < get_n_bodies(); i++) {
        t1[i] = t0[i] + dt * v1[i];
        q1[i] = quat_time_integration(q0[i], w1[i], dt);
        R1[i] = quat_to_matrix(q1[i]);
    }
    t0 = t1;
    q0 = q1;
    R0 = R1;
    v0 = v1;
    w0 = w1;
}

// ======================================================================================================================
// EnergyRigidBodyInertia  (EnergyRigidBodyInertia.cpp)
// ======================================================================================================================
EnergyRigidBodyInertia::EnergyRigidBodyInertia(Stark& s, spRigidBodyDynamics r) : stark(s), rb(r)
{
    stark.add_model(this);
    stark.callbacks->add_before_time_step([this]() { _before_time_step(); });
}
void EnergyRigidBodyInertia::add(int rb_idx, double m, const Mat3& inertia_loc)
{
    if (rb_idx != (int)mass.size()) throw std::runtime_error("EnergyRigidBodyInertia::add() found non-consecutive rigid body added");
    conn.push_back({(int32_t)mass.size()});
    mass.push_back(m);
    J_loc.push_back(inertia_loc);
    J0_glob.push_back(inertia_loc);
    linear_damping.push_back(0.0);
    angular_damping.push_back(0.0);
    is_quasistatic.push_back(0.0);
    stark.mark_registration_dirty();
}
void EnergyRigidBodyInertia::register_potentials(mistark_ctx* ctx)
{
    id_J0 = -1;
    if (conn.empty()) return;
    const int64_t n = (int64_t)mass.size();
    {
        // EnergyRigidBodyInertia.cpp:17-25
        Binder B(ctx, stark);
        B.add_id(rb->id_v1, 3, 0);
        B.add_id(rb->id_v0, 3, 0);
        B.add_id(rb->id_a, 3, 0);
        B.add_id(rb->id_force, 3, 0);
        B.add(mass.data(), n, 1, 0);
        B.add(linear_damping.data(), n, 1, 0);
        B.add(is_quasistatic.data(), n, 1, 0);
        B.add_id(stark.dt_array(), 1, -1);
        B.add_id(stark.gravity_array(), 3, -1);
        B.potential("EnergyRigidBodyInertia_Linear", conn);
    }
    {
        // EnergyRigidBodyInertia.cpp:46-53
        Binder B(ctx, stark);
        B.add_id(rb->id_w1, 3, 0);
        B.add_id(rb->id_w0, 3, 0);
        B.add_id(rb->id_aa, 3, 0);
        B.add_id(rb->id_torque, 3, 0);
        id_J0 = B.add(J0_glob[0].data(), n, 9, 0);
        B.add(angular_damping.data(), n, 1, 0);
        B.add(is_quasistatic.data(), n, 1, 0);
        B.add_id(stark.dt_array(), 1, -1);
        B.potential("EnergyRigidBodyInertia_Angular", conn);
    }
}
void EnergyRigidBodyInertia::_before_time_step()
{
    // EnergyRigidBodyInertia.cpp:83-104: J0_glob = R0 J_loc R0^T
    if (conn.empty()) return;
    for (size_t i = 0; i < mass.size(); i++) J0_glob[i] = matmul(matmul(rb->R0[i], J_loc[i]), transpose(rb->R0[i]));
    if (id_J0 >= 0) stark.check(mistark_upload(stark.ctx, id_J0));
}

// ======================================================================================================================
// RigidBodyHandler  (RigidBodyHandler.cpp)
// ======================================================================================================================
RigidBodyHandler& RigidBodyHandler::set_translation(const Vec3& t)
{
    rb->t1[idx] = t;
    rb->t0[idx] = t;
    return *this;
}
RigidBodyHandler& RigidBodyHandler::add_translation(const Vec3& t)
{
    rb->t1[idx] = rb->t1[idx] + t;
    rb->t0[idx] = rb->t0[idx] + t;
    return *this;
}
RigidBodyHandler& RigidBodyHandler::set_rotation(const Quat& q)
{
    rb->q0[idx] = rb->q1[idx] = rb->q0_[idx] = quat_normalized(q);  // (q0_ too: the device reads it before the first step's callbacks)
    rb->R0[idx] = rb->R1[idx] = quat_to_matrix(rb->q0[idx]);
    return *this;
}
RigidBodyHandler& RigidBodyHandler::set_rotation(double angle_deg, const Vec3& axis) { return set_rotation(quat_angle_axis(deg2rad(angle_deg), normalized(axis))); }
RigidBodyHandler& RigidBodyHandler::add_rotation(const Quat& q)
{
    // RigidBodyHandler.cpp:75-83: rotates the body about the global origin
    const Quat qn = quat_normalized(q);
    const Mat3 R = quat_to_matrix(qn);
    rb->t0[idx] = R * rb->t0[idx];
    rb->t1[idx] = R * rb->t1[idx];
    return set_rotation(quat_mul(qn, rb->q0[idx]));
}
RigidBodyHandler& RigidBodyHandler::add_rotation(double angle_deg, const Vec3& axis, const Vec3& pivot)
{
    add_translation(-1.0 * pivot);
    add_rotation(quat_angle_axis(deg2rad(angle_deg), normalized(axis)));
    return add_translation(pivot);
}
RigidBodyHandler& RigidBodyHandler::set_velocity(const Vec3& v)
{
    rb->v0[idx] = v;  // RigidBodyHandler.cpp:136-141
    rb->v1[idx] = v;
    return *this;
}
RigidBodyHandler& RigidBodyHandler::set_angular_velocity(const Vec3& w)
{
    rb->w0[idx] = w;
    rb->w1[idx] = w;
    return *this;
}
RigidBodyHandler& RigidBodyHandler::set_force_at_centroid(const Vec3& f)
{
    rb->force[idx] = f;
    return *this;
}
RigidBodyHandler& RigidBodyHandler::set_torque(const Vec3& t)
{
    rb->torque[idx] = t;
    return *this;
}
// RigidBodyHandler.cpp:98-127
RigidBodyHandler& RigidBodyHandler::add_force_at_centroid(const Vec3& f)
{
    rb->force[idx] = rb->force[idx] + f;
    return *this;
}
RigidBodyHandler& RigidBodyHandler::add_force_at(const Vec3& f, const Vec3& p)
{
    rb->force[idx] = rb->force[idx] + f;
    rb->torque[idx] = rb->torque[idx] + cross(p - rb->t1[idx], f);
    return *this;
}
RigidBodyHandler& RigidBodyHandler::add_torque(const Vec3& t)
{
    rb->torque[idx] = rb->torque[idx] + t;
    return *this;
}
RigidBodyHandler& RigidBodyHandler::set_linear_damping(double d)
{
    inertia->linear_damping[idx] = d;
    return *this;
}
RigidBodyHandler& RigidBodyHandler::set_angular_damping(double d)
{
    inertia->angular_damping[idx] = d;
    return *this;
}

// ======================================================================================================================
// EnergyRigidBodyConstraints  (EnergyRigidBodyConstraints.cpp, RigidBodyConstraints.h)
// ======================================================================================================================
namespace {
struct KindInfo
{
    const char* name;
    int n_vecs, n_scalars;
    bool two_bodies, has_stiffness, normalize[3];
};
// column meaning per kind (RigidBodyConstraints.h): v0 v1 v2 | s0 s1 s2
const KindInfo KINDS[EnergyRigidBodyConstraints::N_KINDS] = {
    {"rb_constraint_global_points", 2, 0, false, true, {false, false, false}},      // loc, target_glob
    {"rb_constraint_global_directions", 2, 0, false, true, {true, true, false}},    // d_loc, target_d_glob
    {"rb_constraint_points", 2, 0, true, true, {false, false, false}},              // a_loc, b_loc
    {"rb_constraint_point_on_axis", 3, 0, true, true, {false, true, false}},        // a_loc, da_loc, b_loc
    {"rb_constraint_distances", 2, 1, true, true, {false, false, false}},           // a_loc, b_loc | target_distance
    {"rb_constraint_distance_limits", 2, 2, true, true, {false, false, false}},     // a_loc, b_loc | min, max
    {"rb_constraint_directions", 2, 0, true, true, {true, true, false}},            // da_loc, db_loc
    {"rb_constraint_angle_limits", 2, 1, true, true, {true, true, false}},          // da_loc, db_loc | max_distance
    {"rb_constraint_damped_spring", 2, 2, true, true, {false, false, false}},       // a_loc, b_loc | rest_length, damping
    {"rb_constraint_linear_velocity", 1, 3, true, false, {true, false, false}},     // da_loc | target_v, max_force, delay
    {"rb_constraint_angular_velocity", 1, 3, true, false, {true, false, false}},    // da_loc | target_w, max_torque, delay
};
double sq_distance_point_line(const Vec3& p, const Vec3& a, const Vec3& b) { return dot(cross(a - p, b - p), cross(a - p, b - p)) / dot(b - a, b - a); }
}  // namespace

EnergyRigidBodyConstraints::EnergyRigidBodyConstraints(Stark& s, spRigidBodyDynamics r) : stark(s), rb(r)
{
    for (int& id : id_stiffness) id = -1;
    for (int k = 0; k < N_KINDS; k++) id_v0[k] = id_v1[k] = -1;
    stark.add_model(this);
    stark.callbacks->newton->add_is_converged_state_valid([this]() { return _is_converged_state_valid(); });
    stark.callbacks->add_on_time_step_accepted([this]() { _on_time_step_accepted(); });
}
int EnergyRigidBodyConstraints::add(Kind kind, int a, int b, const Vec3* vecs, int n_vecs, const double* scalars, int n_scalars, double stiffness, double tolerance)
{
    const KindInfo& K = KINDS[kind];
    if (n_vecs != K.n_vecs || n_scalars != K.n_scalars) throw std::runtime_error(std::string("EnergyRigidBodyConstraints::add(): wrong parameter count for ") + K.name);
    Table& T = tables[kind];
    const int id = (int)T.conn.size();
    T.conn.push_back({id, a, K.two_bodies ? b : a});
    std::vector<Vec3>* vc[3] = {&T.v0, &T.v1, &T.v2};
    for (int i = 0; i < 3; i++) vc[i]->push_back(i < n_vecs ? (K.normalize[i] ? normalized(vecs[i]) : vecs[i]) : ZERO);
    T.v0_rest.push_back(T.v0.back());
    std::vector<double>* sc[3] = {&T.s0, &T.s1, &T.s2};
    for (int i = 0; i < 3; i++) sc[i]->push_back(i < n_scalars ? scalars[i] : 0.0);
    T.stiffness.push_back(stiffness);
    T.tolerance.push_back(tolerance);
    T.is_active.push_back(1.0);
    stark.mark_registration_dirty();
    return id;
}
void EnergyRigidBodyConstraints::register_potentials(mistark_ctx* ctx)
{
    for (int kind = 0; kind < N_KINDS; kind++) {
        id_stiffness[kind] = id_v0[kind] = id_v1[kind] = -1;
        Table& T = tables[kind];
        if (T.conn.empty()) continue;
        const int64_t n = (int64_t)T.conn.size();
        Binder B(ctx, stark);
        auto vec = [&](std::vector<Vec3>& v) { B.add(v[0].data(), n, 3, 0); };
        auto sca = [&](std::vector<double>& v) { return B.add(v.data(), n, 1, 0); };
        auto x1 = [&](int col) {  // RigidBodyDynamics::get_x1 / get_x1_d1 / get_x0_x1: v1, w1, t0, q0_ of the body
            B.add_id(rb->id_v1, 3, col);
            B.add_id(rb->id_w1, 3, col);
            B.add_id(rb->id_t0, 3, col);
            B.add_id(rb->id_q0_, 4, col);
        };
        auto d1 = [&](int col) {  // RigidBodyDynamics::get_d1: w1, q0_
            B.add_id(rb->id_w1, 3, col);
            B.add_id(rb->id_q0_, 4, col);
        };
        const int dt = stark.dt_array();
        switch (kind) {  // binding order of EnergyRigidBodyConstraints.cpp:30-247
            case GlobalPoints: vec(T.v0); id_v1[kind] = B.add(T.v1[0].data(), n, 3, 0); id_stiffness[kind] = sca(T.stiffness); sca(T.is_active); B.add_id(dt, 1, -1); x1(1); break;
            case GlobalDirections: id_v0[kind] = B.add(T.v0[0].data(), n, 3, 0); vec(T.v1); id_stiffness[kind] = sca(T.stiffness); sca(T.is_active); B.add_id(dt, 1, -1); d1(1); break;
            case Points: vec(T.v0); vec(T.v1); id_stiffness[kind] = sca(T.stiffness); sca(T.is_active); B.add_id(dt, 1, -1); x1(1); x1(2); break;
            case PointOnAxes: vec(T.v0); vec(T.v1); vec(T.v2); id_stiffness[kind] = sca(T.stiffness); sca(T.is_active); B.add_id(dt, 1, -1); x1(1); x1(2); break;
            case Distances: vec(T.v0); vec(T.v1); sca(T.s0); id_stiffness[kind] = sca(T.stiffness); sca(T.is_active); B.add_id(dt, 1, -1); x1(1); x1(2); break;
            case DistanceLimits: vec(T.v0); vec(T.v1); sca(T.s0); sca(T.s1); id_stiffness[kind] = sca(T.stiffness); sca(T.is_active); B.add_id(dt, 1, -1); x1(1); x1(2); break;
            case Directions: vec(T.v0); vec(T.v1); id_stiffness[kind] = sca(T.stiffness); sca(T.is_active); B.add_id(dt, 1, -1); d1(1); d1(2); break;
            case AngleLimits: vec(T.v0); vec(T.v1); sca(T.s0); id_stiffness[kind] = sca(T.stiffness); sca(T.is_active); B.add_id(dt, 1, -1); d1(1); d1(2); break;
            case DampedSprings: vec(T.v0); vec(T.v1); sca(T.s0); id_stiffness[kind] = sca(T.stiffness); sca(T.s1); sca(T.is_active); B.add_id(dt, 1, -1); x1(1); x1(2); break;
            case LinearVelocity:
                vec(T.v0); sca(T.s0); sca(T.s1); sca(T.s2); sca(T.is_active);
                B.add_id(rb->id_v1, 3, 1); B.add_id(rb->id_v1, 3, 2); B.add_id(rb->id_w1, 3, 1); B.add_id(rb->id_q0_, 4, 1); B.add_id(dt, 1, -1);
                break;
            case AngularVelocity:
                vec(T.v0); sca(T.s0); sca(T.s1); sca(T.s2); sca(T.is_active);
                B.add_id(rb->id_w1, 3, 1); B.add_id(rb->id_w1, 3, 2); B.add_id(rb->id_q0_, 4, 1); B.add_id(dt, 1, -1);
                break;
            default: break;
        }
        if (KINDS[kind].two_bodies) {
            B.potential(KINDS[kind].name, T.conn);
        } else {
            // one body: the reference's table is {idx, rb} (EnergyRigidBodyConstraints.cpp:30,47: LabelledConnectivity<2>); the third
            // column of the host table repeats the body and is not handed over
            std::vector<std::array<int32_t, 2>> c2;
            for (const auto& r : T.conn) c2.push_back({r[0], r[1]});
            B.potential(KINDS[kind].name, c2);
        }
    }
}
bool EnergyRigidBodyConstraints::_is_converged_state_valid()
{
    // EnergyRigidBodyConstraints.cpp:250-262: harden every constraint beyond tolerance; the step is redone if any was
    rb->fetch_velocities();
    return _adjust_constraints_stiffness(1.0, stiffness_hard_multiplier, false);
}
void EnergyRigidBodyConstraints::_on_time_step_accepted() { _adjust_constraints_stiffness(soft_constraint_capacity_hardening_point, stiffness_soft_multiplier, true); }
bool EnergyRigidBodyConstraints::_adjust_constraints_stiffness(double cap, double multiplier, bool are_positions_set)
{
    // EnergyRigidBodyConstraints.cpp:277-398. (`cap` is unused by the reference as well.)
    (void)cap;
    const double dt = stark.dt;
    auto X = [&](int b, const Vec3& loc) { return are_positions_set ? rb->get_position_at(b, loc) : rb->get_x1(b, loc, dt); };
    auto D = [&](int b, const Vec3& loc) { return are_positions_set ? rb->get_direction(b, loc) : rb->get_d1(b, loc, dt); };
    bool is_valid = true;
    for (int kind = 0; kind <= AngleLimits; kind++) {
        Table& T = tables[kind];
        bool changed = false;
        for (size_t i = 0; i < T.conn.size(); i++) {
            const int idx = T.conn[i][0], a = T.conn[i][1], b = T.conn[i][2];
            double C = 0.0;  // violation in m or deg
            switch (kind) {
                case GlobalPoints: C = norm(X(a, T.v0[idx]) - T.v1[idx]); break;
                case GlobalDirections: C = rad2deg(std::asin(norm(D(a, T.v0[idx]) - T.v1[idx]))); break;
                case Points: C = norm(X(b, T.v1[idx]) - X(a, T.v0[idx])); break;
                case PointOnAxes: {
                    const Vec3 a1 = X(a, T.v0[idx]), da1 = D(a, T.v1[idx]), b1 = X(b, T.v2[idx]);
                    C = std::sqrt(sq_distance_point_line(b1, a1, a1 + da1));
                    break;
                }
                case Distances: C = std::abs(norm(X(b, T.v1[idx]) - X(a, T.v0[idx])) - T.s0[idx]); break;
                case DistanceLimits: {
                    const double d = norm(X(b, T.v1[idx]) - X(a, T.v0[idx]));
                    C = d < T.s0[idx] ? std::abs(d - T.s0[idx]) : (d > T.s1[idx] ? std::abs(d - T.s1[idx]) : 0.0);
                    break;
                }
                case Directions: C = rad2deg(std::asin(norm(D(b, T.v1[idx]) - D(a, T.v0[idx])))); break;
                case AngleLimits: {
                    const double d = norm(D(b, T.v1[idx]) - D(a, T.v0[idx]));
                    if (d > T.s0[idx]) {
                        const double c = d - T.s0[idx];
                        C = rad2deg(std::acos((2.0 - c * c) / 2.0));  // angle_of_opening_distance
                    }
                    break;
                }
                default: break;
            }
            if (C > T.tolerance[idx]) {
                is_valid = false;
                T.stiffness[idx] *= multiplier;
                changed = true;
            }
        }
        if (changed && id_stiffness[kind] >= 0) stark.check(mistark_upload(stark.ctx, id_stiffness[kind]));
    }
    return is_valid;
}

std::array<double, 2> EnergyRigidBodyConstraints::measure(Kind kind, int idx, int which) const
{
    const Table& T = tables[kind];
    if (idx < 0 || idx >= (int)T.conn.size()) throw std::runtime_error("EnergyRigidBodyConstraints::measure(): bad constraint index");
    const int a = T.conn[idx][1], b = T.conn[idx][2];
    auto X = [&](int body, const Vec3& loc) { return rb->get_position_at(body, loc); };
    auto D = [&](int body, const Vec3& loc) { return rb->get_direction(body, loc); };
    auto V = [&](int body, const Vec3& loc) { return rb->v1[body] + cross(rb->w1[body], rb->get_position_at(body, loc) - rb->t1[body]); };
    const double k = T.stiffness[idx];
    const double EPS = 100.0 * std::numeric_limits<double>::epsilon();  // RigidBodyConstraints.h:52
    auto c1_controller = [](const Vec3& da1, const Vec3& va1, const Vec3& vb1, double target, double max_force, double delay) -> std::array<double, 2> {
        // RigidBodyConstraints.h:70-85
        const double dv = dot(da1, vb1 - va1) - target;
        if (dv < -delay) return {dv, -max_force};
        if (dv < delay) return {dv, -(max_force / delay) * dv};
        return {dv, max_force};
    };
    switch (kind) {
        case GlobalPoints: {  // :114-119
            const double C = norm(X(a, T.v0[idx]) - T.v1[idx]);
            return {C, k * C};
        }
        case GlobalDirections: {  // :154-164
            const Vec3 u = D(a, T.v0[idx]) - T.v1[idx];
            const double C = norm(u);
            const Vec3 force = (-k * C / (C + EPS)) * u;
            return {rad2deg(std::asin(C)), norm(cross(T.v1[idx], force))};
        }
        case Points: {  // :195-200
            const double C = norm(X(b, T.v1[idx]) - X(a, T.v0[idx]));
            return {C, k * C};
        }
        case PointOnAxes: {  // :232-236
            const Vec3 a1 = X(a, T.v0[idx]), da1 = D(a, T.v1[idx]), b1 = X(b, T.v2[idx]);
            const double C = std::sqrt(sq_distance_point_line(b1, a1, a1 + da1));
            return {C, k * C};
        }
        case Distances: {  // :269-275 (signed)
            const double C = norm(X(b, T.v1[idx]) - X(a, T.v0[idx])) - T.s0[idx];
            return {C, -k * C};
        }
        case DistanceLimits: {  // :312-327 (signed)
            const double d = norm(X(b, T.v1[idx]) - X(a, T.v0[idx]));
            if (d < T.s0[idx]) return {d - T.s0[idx], -k * (d - T.s0[idx])};
            if (d > T.s1[idx]) return {d - T.s1[idx], -k * (d - T.s1[idx])};
            return {0.0, 0.0};
        }
        case Directions: {  // :361-371
            const Vec3 da = D(a, T.v0[idx]), u = D(b, T.v1[idx]) - da;
            const double C = norm(u);
            const Vec3 force = (k * C / (C + EPS)) * u;
            return {rad2deg(std::asin(C)), norm(cross(da, force))};
        }
        case AngleLimits: {  // :414-427
            const Vec3 da = D(a, T.v0[idx]), u = D(b, T.v1[idx]) - da;
            const double d = norm(u);
            if (d > T.s0[idx]) {
                const double C = d - T.s0[idx];
                const Vec3 force = (k * C * C / (d + EPS)) * u;
                return {rad2deg(std::acos((2.0 - C * C) / 2.0)), norm(cross(da, force))};
            }
            return {0.0, 0.0};
        }
        case DampedSprings: {  // :467-481
            const Vec3 a1 = X(a, T.v0[idx]), b1 = X(b, T.v1[idx]);
            if (which == 0) {
                const double C = norm(b1 - a1) - T.s0[idx];
                return {C, -k * C};
            }
            const double dv = dot(V(b, T.v1[idx]) - V(a, T.v0[idx]), normalized(b1 - a1));
            return {dv, -T.s1[idx] * dv};
        }
        case LinearVelocity:  // :511-514
            return c1_controller(D(a, T.v0[idx]), rb->v1[a], rb->v1[b], T.s0[idx], T.s1[idx], T.s2[idx]);
        case AngularVelocity: {  // :544-548
            const auto r = c1_controller(D(a, T.v0[idx]), rb->w1[a], rb->w1[b], T.s0[idx], T.s1[idx], T.s2[idx]);
            return {rad2deg(r[0]), r[1]};
        }
        default: break;
    }
    throw std::runtime_error("EnergyRigidBodyConstraints::measure(): bad kind");
}

// ======================================================================================================================
// RigidBodies  (RigidBodies.cpp)
// ======================================================================================================================
RigidBodies::RigidBodies(Stark& stark, spRigidBodyDynamics r) : rb(r)
{
    inertia = std::make_shared<EnergyRigidBodyInertia>(stark, rb);
    constraints = std::make_shared<EnergyRigidBodyConstraints>(stark, rb);
}
RigidBodyHandler RigidBodies::add(double mass, const Mat3& inertia_local)
{
    const int idx = rb->add();
    inertia->add(idx, mass, inertia_local);
    return RigidBodyHandler(rb.get(), inertia.get(), idx);
}
using C = EnergyRigidBodyConstraints;
int RigidBodies::add_constraint_global_point(const RigidBodyHandler& body, const Vec3& p)
{
    const Vec3 v[2] = {body.transform_global_to_local_point(p), p};
    return constraints->add(C::GlobalPoints, body.get_idx(), -1, v, 2, nullptr, 0, default_stiffness, default_tolerance_in_m);
}
int RigidBodies::add_constraint_global_direction(const RigidBodyHandler& body, const Vec3& d)
{
    const Vec3 v[2] = {body.transform_global_to_local_direction(d), d};
    return constraints->add(C::GlobalDirections, body.get_idx(), -1, v, 2, nullptr, 0, default_stiffness, default_tolerance_in_deg);
}
int RigidBodies::add_constraint_point(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p)
{
    const Vec3 v[2] = {a.transform_global_to_local_point(p), b.transform_global_to_local_point(p)};
    return constraints->add(C::Points, a.get_idx(), b.get_idx(), v, 2, nullptr, 0, default_stiffness, default_tolerance_in_m);
}
int RigidBodies::add_constraint_point_on_axis(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p, const Vec3& d)
{
    const Vec3 v[3] = {a.transform_global_to_local_point(p), a.transform_global_to_local_direction(d), b.transform_global_to_local_point(p)};
    return constraints->add(C::PointOnAxes, a.get_idx(), b.get_idx(), v, 3, nullptr, 0, default_stiffness, default_tolerance_in_m);
}
int RigidBodies::add_constraint_distance(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& ag, const Vec3& bg)
{
    const Vec3 v[2] = {a.transform_global_to_local_point(ag), b.transform_global_to_local_point(bg)};
    const double s[1] = {norm(ag - bg)};
    return constraints->add(C::Distances, a.get_idx(), b.get_idx(), v, 2, s, 1, default_stiffness, default_tolerance_in_m);
}
int RigidBodies::add_constraint_distance_limits(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& ag, const Vec3& bg, double min_distance, double max_distance)
{
    const double d = norm(ag - bg);
    if (d < min_distance || max_distance < d) throw std::runtime_error("RigidBodies::add_constraint_distance_limits() got a rest distance out of limits");
    const Vec3 v[2] = {a.transform_global_to_local_point(ag), b.transform_global_to_local_point(bg)};
    const double s[2] = {min_distance, max_distance};
    return constraints->add(C::DistanceLimits, a.get_idx(), b.get_idx(), v, 2, s, 2, default_stiffness, default_tolerance_in_m);
}
int RigidBodies::add_constraint_direction(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& d)
{
    const Vec3 v[2] = {a.transform_global_to_local_direction(d), b.transform_global_to_local_direction(d)};
    return constraints->add(C::Directions, a.get_idx(), b.get_idx(), v, 2, nullptr, 0, default_stiffness, default_tolerance_in_deg);
}
int RigidBodies::add_constraint_angle_limit(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& d, double admissible_angle_deg)
{
    const Vec3 v[2] = {a.transform_global_to_local_direction(d), b.transform_global_to_local_direction(d)};
    const double s[1] = {std::sqrt(1.0 + 1.0 - 2.0 * std::cos(deg2rad(admissible_angle_deg)))};  // opening_distance_of_angle
    return constraints->add(C::AngleLimits, a.get_idx(), b.get_idx(), v, 2, s, 1, default_stiffness, default_tolerance_in_deg);
}
int RigidBodies::add_constraint_spring(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& ag, const Vec3& bg, double stiffness, double damping)
{
    const Vec3 v[2] = {a.transform_global_to_local_point(ag), b.transform_global_to_local_point(bg)};
    const double s[2] = {norm(ag - bg), damping};
    return constraints->add(C::DampedSprings, a.get_idx(), b.get_idx(), v, 2, s, 2, stiffness, 0.0);
}
int RigidBodies::add_constraint_linear_velocity(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& d, double target_v, double max_force, double delay)
{
    if (max_force < 0.0) throw std::runtime_error("RigidBodies::add_constraint_linear_velocity() got a negative force");
    const Vec3 v[1] = {a.transform_global_to_local_direction(d)};
    const double s[3] = {target_v, max_force, delay};
    return constraints->add(C::LinearVelocity, a.get_idx(), b.get_idx(), v, 1, s, 3, 0.0, 0.0);
}
int RigidBodies::add_constraint_angular_velocity(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& d, double target_w, double max_abs_torque, double delay)
{
    if (max_abs_torque < 0.0) throw std::runtime_error("RigidBodies::add_constraint_angular_velocity() got a negative torque");
    const Vec3 v[1] = {a.transform_global_to_local_direction(d)};
    const double s[3] = {target_w, max_abs_torque, delay};
    return constraints->add(C::AngularVelocity, a.get_idx(), b.get_idx(), v, 1, s, 3, 0.0, 0.0);
}
static Vec3 orthogonal_to(const Vec3& d)
{
    // RigidBodies.cpp:257, :279
    const Vec3 ex = {1.0, 0.0, 0.0}, ey = {0.0, 1.0, 0.0};
    return dot(d, ex) < 0.5 ? cross(d, ex) : cross(d, ey);
}
void RigidBodies::add_constraint_fix(const RigidBodyHandler& body)
{
    add_constraint_global_point(body, body.get_translation());
    add_constraint_global_direction(body, {0.0, 0.0, 1.0});
    add_constraint_global_direction(body, {1.0, 0.0, 0.0});
}
void EnergyRigidBodyConstraints::set_global_target_point(int idx, const Vec3& p)
{
    Table& T = tables[GlobalPoints];
    if (idx < 0 || idx >= (int)T.v1.size()) throw std::runtime_error("set_global_target_point(): no such constraint");
    T.v1[idx] = p;
    if (stark.ctx && id_v1[GlobalPoints] >= 0) stark.check(mistark_upload(stark.ctx, id_v1[GlobalPoints]));
}
void EnergyRigidBodyConstraints::set_global_direction_rotation(int idx, const Mat3& R)
{
    Table& T = tables[GlobalDirections];
    if (idx < 0 || idx >= (int)T.v0.size()) throw std::runtime_error("set_global_direction_rotation(): no such constraint");
    T.v0[idx] = R * T.v0_rest[idx];
    if (stark.ctx && id_v0[GlobalDirections] >= 0) stark.check(mistark_upload(stark.ctx, id_v0[GlobalDirections]));
}
void RigidBodies::set_fix_transformation(int anchor_point, int z_lock, int x_lock, const Vec3& translation, const Mat3& rotation)
{
    constraints->set_global_target_point(anchor_point, translation);
    constraints->set_global_direction_rotation(z_lock, rotation);
    constraints->set_global_direction_rotation(x_lock, rotation);
}
void RigidBodies::add_constraint_attachment(const RigidBodyHandler& a, const RigidBodyHandler& b)
{
    add_constraint_point(a, b, 0.5 * (a.get_translation() + b.get_translation()));
    add_constraint_direction(a, b, {0.0, 0.0, 1.0});
    add_constraint_direction(a, b, {1.0, 0.0, 0.0});
}
void RigidBodies::add_constraint_point_with_angle_limit(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p, const Vec3& d, double deg)
{
    add_constraint_point(a, b, p);
    add_constraint_angle_limit(a, b, d, deg);
}
void RigidBodies::add_constraint_hinge(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p, const Vec3& d)
{
    add_constraint_point(a, b, p);
    add_constraint_direction(a, b, d);
}
void RigidBodies::add_constraint_hinge_with_angle_limit(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p, const Vec3& d, double deg)
{
    const Vec3 u = orthogonal_to(d);
    add_constraint_hinge(a, b, p, d);
    add_constraint_angle_limit(a, b, u, deg);
}
void RigidBodies::add_constraint_slider(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p, const Vec3& d)
{
    add_constraint_point_on_axis(a, b, p, d);
    add_constraint_direction(a, b, d);
}
void RigidBodies::add_constraint_prismatic_slider(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p, const Vec3& d)
{
    const Vec3 u = orthogonal_to(d);
    add_constraint_slider(a, b, p, d);
    add_constraint_direction(a, b, u);
}
void RigidBodies::add_constraint_spring_with_limits(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& ag, const Vec3& bg, double stiffness, double min_length,
                                                    double max_length, double damping)
{
    add_constraint_spring(a, b, ag, bg, stiffness, damping);
    add_constraint_distance_limits(a, b, ag, bg, min_length, max_length);
}
void RigidBodies::add_constraint_prismatic_press(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p, const Vec3& d, double target_v, double max_force, double delay)
{
    add_constraint_prismatic_slider(a, b, p, d);
    add_constraint_linear_velocity(a, b, d, target_v, max_force, delay);
}
void RigidBodies::add_constraint_motor(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p, const Vec3& d, double target_w, double max_torque, double delay)
{
    add_constraint_hinge(a, b, p, d);
    add_constraint_angular_velocity(a, b, d, target_w, max_torque, delay);
}

}  // namespace mistark
