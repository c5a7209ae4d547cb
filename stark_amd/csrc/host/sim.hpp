// host/sim.hpp — C++ host mirror of the reference's time-step driver, state containers and energy registration
// classes for the hot path, calling the device engine through the C ABI (include/mistark.h) only.
//
// Same class names, Params fields, add(...) signatures, callback hooks and accept/retry/halve-dt policy as
//   stark::core::Stark / Settings / Callbacks      stark/src/core/{Stark,Settings,Callbacks}.*
//   stark::PointDynamics, PointSetHandler          stark/src/models/deformables/PointDynamics.*, PointSetHandler.*
//   stark::EnergyLumpedInertia                     stark/src/models/deformables/point/EnergyLumpedInertia.*
//   stark::EnergyPrescribedPositions               stark/src/models/deformables/point/EnergyPrescribedPositions.*
//   stark::EnergyTetStrain                         stark/src/models/deformables/volume/EnergyTetStrain.*
//   stark::EnergyTriangleStrain                    stark/src/models/deformables/surface/EnergyTriangleStrain.*
//   stark::EnergyDiscreteShells                    stark/src/models/deformables/surface/EnergyDiscreteShells.*
//   stark::Deformables, DeformablesPresets, Simulation   stark/src/models/{deformables/Deformables.h,presets/*,Simulation.*}
// Eigen::Vector3d is replaced by the layout-compatible std::array<double,3> (mistark::Vec3).
// What differs by design: potentials are registered by name + bound arrays (no symbolic lambda, no JIT), the state
// lives on the MI355X between callbacks, and host arrays are mirrors refreshed at the documented sync points.
#pragma once
#include <array>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/mistark.h"
#include "../../../include/mistark_contact.h"
#include "mesh.hpp"

namespace mistark {

// ---- stark::core::Settings (stark/src/core/Settings.h:10-49, defaults Settings.cpp:43-50) ---------------------------
struct Settings
{
    struct Output
    {
        std::string simulation_name = "";
        std::string output_directory = "";
        bool enable_output = false;        // console line per step (Stark.cpp:182-189)
        bool enable_frame_writes = false;  // VTK frames through the write_frame callbacks (Stark.cpp:314-338); needs output_directory
        int fps = 30;                      // < 0: a frame after every accepted time step
    } output;
    struct Simulation
    {
        Vec3 gravity = {0.0, 0.0, -9.81};
        bool init_frictional_contact = true;
        double max_time_step_size = 1.0 / 30.0;
        bool use_adaptive_time_step = true;
        double time_step_size_success_multiplier = 1.05;
        double time_step_size_lower_bound = 1e-6;
    } simulation;
    mistark_newton_settings newton;
    struct Execution
    {
        double allowed_execution_time = std::numeric_limits<double>::max();
        double end_simulation_time = std::numeric_limits<double>::max();
        int end_frame = std::numeric_limits<int>::max();
        int device = 0;                   // MI355X ordinal (replaces n_threads)
        bool mirror_state_to_host = true; // refresh PointDynamics host arrays after every accepted step
        // multi-GPU sharding (include/mistark.h "multi-GPU"): one Simulation per GPU, all built identically
        int rank = 0, world = 1;
        std::string rccl_unique_id;        // 128 bytes from mistark_dist_unique_id (rank 0), shared by the launcher
        mistark_local_group* local_group = nullptr;  // in-process group instead of RCCL (tests)
        mistark_ipc_comm* ipc_comm = nullptr;        // IPC windows instead of RCCL (mistark.h "IPC windows")
    } execution;
    Settings() { mistark_newton_default_settings(&newton); }
};

// ---- stark::core::Callbacks (stark/src/core/Callbacks.h:13-84) + symx::SolverCallbacks (solver_utils.h:29-117) ------
struct SolverCallbacks
{
    std::vector<std::function<void()>> before_energy_evaluation;
    std::vector<std::function<bool()>> is_initial_state_valid, is_intermediate_state_valid, is_converged, is_converged_state_valid;
    std::vector<std::function<void()>> on_intermediate_state_invalid, on_armijo_fail;
    std::vector<std::function<double()>> max_allowed_step;
    void add_before_energy_evaluation(std::function<void()> f) { before_energy_evaluation.push_back(f); }
    void add_is_initial_state_valid(std::function<bool()> f) { is_initial_state_valid.push_back(f); }
    void add_is_intermediate_state_valid(std::function<bool()> f) { is_intermediate_state_valid.push_back(f); }
    void add_on_intermediate_state_invalid(std::function<void()> f) { on_intermediate_state_invalid.push_back(f); }
    void add_on_armijo_fail(std::function<void()> f) { on_armijo_fail.push_back(f); }
    void add_is_converged(std::function<bool()> f) { is_converged.push_back(f); }
    void add_is_converged_state_valid(std::function<bool()> f) { is_converged_state_valid.push_back(f); }
    void add_max_allowed_step(std::function<double()> f) { max_allowed_step.push_back(f); }
};
struct Callbacks
{
    std::shared_ptr<SolverCallbacks> newton = std::make_shared<SolverCallbacks>();
    std::vector<std::function<void()>> before_simulation, before_time_step, after_time_step, on_time_step_accepted;
    std::vector<std::function<bool()>> should_continue_execution;
    std::vector<std::function<void()>> write_frame;
    void add_write_frame(std::function<void()> f) { write_frame.push_back(f); }
    void add_before_simulation(std::function<void()> f) { before_simulation.push_back(f); }
    void add_before_time_step(std::function<void()> f) { before_time_step.push_back(f); }
    void add_after_time_step(std::function<void()> f) { after_time_step.push_back(f); }
    void add_on_time_step_accepted(std::function<void()> f) { on_time_step_accepted.push_back(f); }
    void add_should_continue_execution(std::function<bool()> f) { should_continue_execution.push_back(f); }
};

// A model that owns host arrays / connectivity and (re)registers them with the engine (the reference's constructors call
// global_potential->add_potential / add_dof once with lambdas; we call the C ABI with the current pointers and sizes).
struct Registrable
{
    virtual ~Registrable() = default;
    virtual void register_dofs(mistark_ctx* ctx) {}
    virtual void register_potentials(mistark_ctx* ctx) {}
};

// ---- stark::core::Stark (stark/src/core/Stark.h:12-46, Stark.cpp:133-313) ---------------------------------------------
class Stark
{
public:
    Settings settings;  // (const in the reference; the Newton settings may be replaced between steps here)
    mistark_ctx* ctx = nullptr;
    std::shared_ptr<Callbacks> callbacks = std::make_shared<Callbacks>();
    double current_time = 0.0;
    int current_time_step = 0;
    int current_frame = 0;
    double next_frame_time = 0.0;
    std::string get_frame_path(const std::string& name) const;  // Stark.cpp:245-248
    double dt = -1.0;        // bound by address into the potentials, like mws.make_scalar(stark.dt)
    Vec3 gravity = {0.0, 0.0, -9.81};
    mistark_newton_stats last_stats{};
    int last_newton_result = MISTARK_RUNNING;
    // accumulated over the run (the reference's Logger series)
    long total_newton_iterations = 0, total_cg_iterations = 0, total_linear_solves = 0, failed_steps = 0;
    double total_newton_time = 0.0, total_linear_solve_time = 0.0, total_eval_pgh_time = 0.0, total_eval_p_time = 0.0, total_project_time = 0.0,
           total_assembly_time = 0.0, total_callback_time = 0.0, total_step_time = 0.0;
    long total_evaluations = 0;

    explicit Stark(const Settings& settings);
    ~Stark();
    Stark(const Stark&) = delete;
    bool run_one_step();
    void begin_time_step();           // before_time_step callbacks (friction tables, rigid-body caches, v1 = 0), without solving
    void before_energy_evaluation();  // the Newton callback of that name (contact tables at the current DoFs)
    bool run(double duration, std::function<void()> callback = nullptr);
    void add_model(Registrable* m) { models.push_back(m); registration_dirty = true; }
    void mark_registration_dirty() { registration_dirty = true; }
    int dt_array() const { return dt_array_id; }
    int gravity_array() const { return gravity_array_id; }
    void ensure_registered();  // (re)register every model with the engine if anything changed
    void check(int rc) const;  // throws std::runtime_error with mistark_last_error on rc < 0

private:
    std::vector<Registrable*> models;
    bool is_init = false;
    bool registration_dirty = true;
    int dt_array_id = -1, gravity_array_id = -1;
    double dt_uploaded = -1.0;
    void _initialize();
    void _write_frame();
};

// ---- stark::PointSetHandler / PointDynamics ------------------------------------------------------------------------------
class PointDynamics;
class PointSetHandler
{
    int idx = -1;
    PointDynamics* dyn = nullptr;

public:
    PointSetHandler() = default;
    PointSetHandler(PointDynamics* dyn, int idx) : idx(idx), dyn(dyn) {}
    int get_idx() const { return idx; }
    bool is_valid() const { return dyn != nullptr; }
    int get_begin() const;
    int get_end() const;
    int size() const;
    int get_global_index(int local_index) const;
    template <std::size_t N>
    std::array<int, N> get_global_indices(const std::array<int, N>& loc) const
    {
        std::array<int, N> g;
        for (std::size_t i = 0; i < N; i++) g[i] = get_global_index(loc[i]);
        return g;
    }
    std::vector<int> all() const;
    Vec3 get_position(int local_index) const;
    Vec3 get_rest_position(int local_index) const;
    // PointSetHandler.cpp:104-129 (before the first step: edits the host arrays the registration starts from)
    PointSetHandler& add_displacement(const Vec3& displacement, bool also_at_rest_pose = true);
    PointSetHandler& add_rotation(double angle_deg, const Vec3& axis, const Vec3& pivot = {0.0, 0.0, 0.0}, bool also_at_rest_pose = true);
};

class PointDynamics : public Registrable
{
public:
    // AoS arrays, all point sets concatenated (stark/src/models/deformables/PointDynamics.h:16-23)
    std::vector<Vec3> X, x0, x1, v0, v1, a, f;
    std::vector<int> set_begin;  // per set; set i = [set_begin[i], set_begin[i+1])
    std::vector<std::string> labels;
    // engine array ids (valid after registration)
    int id_X = -1, id_x0 = -1, id_v0 = -1, id_v1 = -1, id_a = -1, id_f = -1, dof_set = -1;

    explicit PointDynamics(Stark& stark);
    PointSetHandler add(const std::vector<Vec3>& x, const std::string& label = "");
    int size() const { return (int)X.size(); }
    int get_begin(int s) const { return set_begin[s]; }
    int get_end(int s) const { return set_begin[s + 1]; }
    int get_global_index(int s, int local) const
    {
        // IntervalVector::_assert_existing_set / _assert_local_idx (stark/src/models/IntervalVector.h:194-199): the reference exits, this throws
        if (s < 0 || s + 1 >= (int)set_begin.size()) throw std::runtime_error("PointDynamics: point set " + std::to_string(s) + " does not exist");
        if (local < 0 || set_begin[s] + local >= set_begin[s + 1]) throw std::runtime_error("PointDynamics: local index " + std::to_string(local) + " out of the range of point set " + std::to_string(s));
        return set_begin[s] + local;
    }
    Vec3 get_x1(int global_index, double dt) const { return x0[global_index] + dt * v1[global_index]; }
    void mirror_to_host();   // device -> host for x0, v0, v1 (x1 recomputed)
    void upload_state();     // host -> device after the user edited positions / velocities / forces
    void mark_state_edited();  // host arrays were edited before/between steps: re-register (uploads them)
    void register_dofs(mistark_ctx* ctx) override;

private:
    Stark& stark;
    void _before_time_step();
    void _on_time_step_accepted();
};
using spPointDynamics = std::shared_ptr<PointDynamics>;

// ---- energies ------------------------------------------------------------------------------------------------------------
#define MISTARK_HANDLER(Model, ParamsT)                                              \
    struct Handler                                                                   \
    {                                                                                \
        Model* model = nullptr;                                                      \
        int idx = -1;                                                                \
        Handler() = default;                                                         \
        Handler(Model* m, int i) : model(m), idx(i) {}                               \
        int get_idx() const { return idx; }                                          \
        bool is_valid() const { return model != nullptr; }                           \
        ParamsT get_params() const { return model->get_params(*this); }              \
        void set_params(const ParamsT& p) { model->set_params(*this, p); }           \
    };

class EnergyLumpedInertia : public Registrable
{
public:
    struct Params
    {
        double density = 1.0, damping = 0.0;
        bool quasistatic = false;
    };
    MISTARK_HANDLER(EnergyLumpedInertia, Params)
    EnergyLumpedInertia(Stark& stark, spPointDynamics dyn);
    Handler add(const PointSetHandler& set, const std::vector<int>& points, const std::vector<double>& lumped_volume, const Params& params);
    Handler add(const PointSetHandler& set, const std::vector<double>& lumped_volume, const Params& params);
    Handler add(const PointSetHandler& set, const std::vector<std::array<int, 2>>& segments, const Params& params);
    Handler add(const PointSetHandler& set, const std::vector<std::array<int, 3>>& triangles, const Params& params);
    Handler add(const PointSetHandler& set, const std::vector<std::array<int, 4>>& tets, const Params& params);
    Params get_params(const Handler& h) const;
    void set_params(const Handler& h, const Params& p);
    double get_mass(const Handler& h) const;
    void register_potentials(mistark_ctx* ctx) override;

private:
    Stark& stark;
    spPointDynamics dyn;
    std::vector<std::array<int32_t, 3>> conn;  // idx, glob, group
    std::vector<double> density, damping, is_quasistatic, lumped_volume;
    bool params_dirty = true;
};

class EnergyPrescribedPositions : public Registrable
{
public:
    struct Params
    {
        double stiffness = 1e3;
        double tolerance = std::numeric_limits<double>::max();
    };
    MISTARK_HANDLER(EnergyPrescribedPositions, Params)
    EnergyPrescribedPositions(Stark& stark, spPointDynamics dyn);
    Handler add(const PointSetHandler& set, const std::vector<int>& points, const Params& params);
    Handler add_inside_aabb(const PointSetHandler& set, const Vec3& aabb_center, const Vec3& aabb_dim, const Params& params);
    Handler add_outside_aabb(const PointSetHandler& set, const Vec3& aabb_center, const Vec3& aabb_dim, const Params& params);
    Params get_params(const Handler& h) const;
    void set_params(const Handler& h, const Params& p);
    void set_transformation(const Handler& h, const Vec3& t, const std::array<double, 9>& R);
    void set_target_position(const Handler& h, int prescribed_idx, const Vec3& t);
    void register_potentials(mistark_ctx* ctx) override;

private:
    Stark& stark;
    spPointDynamics dyn;
    std::vector<std::array<int32_t, 3>> conn;  // idx, point, group
    std::vector<Vec3> target_positions, rest_positions;
    std::vector<double> stiffness, tolerance;
    std::vector<std::array<int, 2>> group_begin_end;
    int id_target = -1, id_stiffness = -1;
    bool targets_dirty = false, stiffness_dirty = false;
    bool _is_converged_state_valid();
    void _before_energy_evaluation();
};

// stark::EnergySegmentStrain (stark/src/models/deformables/line/EnergySegmentStrain.*): rods
class EnergySegmentStrain : public Registrable
{
public:
    struct Params
    {
        bool elasticity_only = false;
        double scale = 1.0, section_radius = 5e-3, youngs_modulus = 1e3, damping = 0.0;
        double strain_limit = std::numeric_limits<double>::max(), strain_limit_stiffness = 1e3;
    };
    MISTARK_HANDLER(EnergySegmentStrain, Params)
    EnergySegmentStrain(Stark& stark, spPointDynamics dyn);
    Handler add(const PointSetHandler& set, const std::vector<std::array<int, 2>>& segments, const Params& params);
    Params get_params(const Handler& h) const;
    void set_params(const Handler& h, const Params& p);
    void register_potentials(mistark_ctx* ctx) override;

private:
    Stark& stark;
    spPointDynamics dyn;
    std::vector<std::array<int32_t, 4>> conn_elasticity_only, conn_complete;  // idx, group, i, j
    std::vector<char> elasticity_only;
    std::vector<double> scale, section_radius, youngs_modulus, strain_damping, strain_limit, strain_limit_stiffness;
};

class EnergyTetStrain : public Registrable
{
public:
    struct Params
    {
        bool elasticity_only = false;
        double scale = 1.0, youngs_modulus = 1e3, poissons_ratio = 0.3, damping = 0.0;
        double strain_limit = std::numeric_limits<double>::max(), strain_limit_stiffness = 1e3;
    };
    MISTARK_HANDLER(EnergyTetStrain, Params)
    EnergyTetStrain(Stark& stark, spPointDynamics dyn);
    Handler add(const PointSetHandler& set, const std::vector<std::array<int, 4>>& tets, const Params& params);
    Params get_params(const Handler& h) const;
    void set_params(const Handler& h, const Params& p);
    void register_potentials(mistark_ctx* ctx) override;

private:
    Stark& stark;
    spPointDynamics dyn;
    std::vector<std::array<int32_t, 6>> conn_elasticity_only, conn_complete;  // idx, group, i, j, k, l
    std::vector<char> elasticity_only;
    std::vector<double> scale, youngs_modulus, poissons_ratio, strain_damping, strain_limit, strain_limit_stiffness;
};

class EnergyTriangleStrain : public Registrable
{
public:
    struct Params
    {
        bool elasticity_only = false;
        double scale = 1.0, thickness = 1e-3, youngs_modulus = 1e3, poissons_ratio = 0.3, damping = 0.0;
        double strain_limit = std::numeric_limits<double>::max(), strain_limit_stiffness = 1e3, inflation = 0.0;
    };
    MISTARK_HANDLER(EnergyTriangleStrain, Params)
    EnergyTriangleStrain(Stark& stark, spPointDynamics dyn);
    Handler add(const PointSetHandler& set, const std::vector<std::array<int, 3>>& triangles, const Params& params);
    Params get_params(const Handler& h) const;
    void set_params(const Handler& h, const Params& p);
    void register_potentials(mistark_ctx* ctx) override;

private:
    Stark& stark;
    spPointDynamics dyn;
    std::vector<std::array<int32_t, 5>> conn_elasticity_only, conn_complete;  // idx, group, i, j, k
    std::vector<char> elasticity_only;
    std::vector<double> scale, thickness, youngs_modulus, poissons_ratio, strain_damping, strain_limit, strain_limit_stiffness, inflation;
};

class EnergyDiscreteShells : public Registrable
{
public:
    struct Params
    {
        double scale = 1.0, stiffness = 1e3, damping = 0.0;
        bool flat_rest_angle = false;
    };
    MISTARK_HANDLER(EnergyDiscreteShells, Params)
    EnergyDiscreteShells(Stark& stark, spPointDynamics dyn);
    Handler add(const PointSetHandler& set, const std::vector<std::array<int, 3>>& triangles, const Params& params);
    Params get_params(const Handler& h) const;
    void set_params(const Handler& h, const Params& p);
    void register_potentials(mistark_ctx* ctx) override;

private:
    Stark& stark;
    spPointDynamics dyn;
    std::vector<std::array<int32_t, 6>> conn_flat_rest, conn_complete;  // idx, group, v0..v3
    std::vector<double> scale, bending_stiffness, bending_damping;
    std::vector<char> flat_rest_angle;
    // per hinge (indexed by "idx"): the two tables share the index space of their own connectivity
    std::vector<double> rest_dihedral_angle_rad, rest_edge_length, rest_height;  // complete
    std::vector<double> bergou_coef;                                             // flat
    std::vector<std::array<double, 4>> bergou_K;                                 // flat
};

// ---- rigid bodies ---------------------------------------------------------------------------------------------------------------------
// stark::RigidBodyDynamics (stark/src/models/rigidbodies/RigidBodyDynamics.*): the state of the (few) bodies is kept on the host and
// pushed to the device before every step; the solved v1 / w1 are the only values read back.
class RigidBodyDynamics : public Registrable
{
public:
    std::vector<Vec3> t0, t1, v0, v1, w0, w1, a, aa, force, torque;
    std::vector<Quat> q0, q1, q0_;
    std::vector<Mat3> R0, R1;
    std::vector<std::string> labels;
    int id_v1 = -1, id_w1 = -1, id_v0 = -1, id_w0 = -1, id_a = -1, id_aa = -1, id_force = -1, id_torque = -1, id_t0 = -1, id_q0_ = -1;

    explicit RigidBodyDynamics(Stark& stark);
    int add(const std::string& label = "");
    int get_n_bodies() const { return (int)t0.size(); }
    Vec3 get_x1(int rb, const Vec3& x_loc, double dt) const;  // integrate_loc_point
    Vec3 get_d1(int rb, const Vec3& d_loc, double dt) const;  // integrate_loc_direction
    Vec3 get_position_at(int rb, const Vec3& x_loc) const { return R1[rb] * x_loc + t1[rb]; }
    Vec3 get_direction(int rb, const Vec3& d_loc) const { return R1[rb] * d_loc; }
    void fetch_velocities();  // device -> host for v1, w1
    void register_dofs(mistark_ctx* ctx) override;

private:
    Stark& stark;
    void _before_time_step();
    void _on_time_step_accepted();
};
using spRigidBodyDynamics = std::shared_ptr<RigidBodyDynamics>;

// stark::EnergyRigidBodyInertia (stark/src/models/rigidbodies/EnergyRigidBodyInertia.*)
class EnergyRigidBodyInertia : public Registrable
{
public:
    std::vector<double> mass, linear_damping, angular_damping, is_quasistatic;
    std::vector<Mat3> J_loc, J0_glob;
    EnergyRigidBodyInertia(Stark& stark, spRigidBodyDynamics rb);
    void add(int rb_idx, double mass, const Mat3& inertia_loc);
    void register_potentials(mistark_ctx* ctx) override;

private:
    Stark& stark;
    spRigidBodyDynamics rb;
    std::vector<std::array<int32_t, 1>> conn;
    int id_J0 = -1;
    void _before_time_step();
};

// stark::RigidBodyHandler (stark/src/models/rigidbodies/RigidBodyHandler.*), the part scenes use
class RigidBodyHandler
{
    RigidBodyDynamics* rb = nullptr;
    EnergyRigidBodyInertia* inertia = nullptr;
    int idx = -1;

public:
    RigidBodyHandler() = default;
    RigidBodyHandler(RigidBodyDynamics* rb, EnergyRigidBodyInertia* inertia, int idx) : rb(rb), inertia(inertia), idx(idx) {}
    int get_idx() const { return idx; }
    bool is_valid() const { return rb != nullptr; }
    Vec3 get_translation() const { return rb->t1[idx]; }
    Quat get_quaternion() const { return rb->q1[idx]; }
    Vec3 get_velocity() const { return rb->v1[idx]; }
    Vec3 get_angular_velocity() const { return rb->w1[idx]; }
    RigidBodyHandler& set_translation(const Vec3& t);
    RigidBodyHandler& add_translation(const Vec3& t);
    RigidBodyHandler& set_rotation(const Quat& q);
    RigidBodyHandler& set_rotation(double angle_deg, const Vec3& axis);
    RigidBodyHandler& add_rotation(const Quat& q);
    RigidBodyHandler& add_rotation(double angle_deg, const Vec3& axis, const Vec3& pivot = {0.0, 0.0, 0.0});
    RigidBodyHandler& set_velocity(const Vec3& v);
    RigidBodyHandler& set_angular_velocity(const Vec3& w);
    RigidBodyHandler& set_force_at_centroid(const Vec3& f);
    RigidBodyHandler& add_force_at_centroid(const Vec3& f);
    RigidBodyHandler& add_force_at(const Vec3& f, const Vec3& application_point_glob);
    RigidBodyHandler& set_torque(const Vec3& t);
    RigidBodyHandler& add_torque(const Vec3& t);
    Vec3 get_velocity_at(const Vec3& x_loc) const { return rb->v1[idx] + cross(rb->w1[idx], rb->get_position_at(idx, x_loc) - rb->t1[idx]); }
    RigidBodyHandler& set_linear_damping(double d);
    RigidBodyHandler& set_angular_damping(double d);
    Vec3 transform_global_to_local_point(const Vec3& x) const { return transpose(rb->R1[idx]) * (x - rb->t1[idx]); }
    Vec3 transform_global_to_local_direction(const Vec3& d) const { return transpose(rb->R1[idx]) * d; }
    Vec3 transform_local_to_global_point(const Vec3& x) const { return rb->R1[idx] * x + rb->t1[idx]; }
};

// stark::EnergyRigidBodyConstraints + the RigidBodyConstraints containers (EnergyRigidBodyConstraints.*, RigidBodyConstraints.h).
// One table per constraint kind: connectivity rows {idx, a[, b]} and per-constraint parameter columns, bound in the reference's order.
class EnergyRigidBodyConstraints : public Registrable
{
public:
    enum Kind { GlobalPoints = 0, GlobalDirections, Points, PointOnAxes, Distances, DistanceLimits, Directions, AngleLimits, DampedSprings, LinearVelocity, AngularVelocity, N_KINDS };
    struct Table
    {
        std::vector<std::array<int32_t, 3>> conn;      // idx, a, b (b unused for the global kinds)
        std::vector<Vec3> v0, v1, v2;                   // up to three vector columns (meaning per kind)
        std::vector<Vec3> v0_rest;                      // v0 as registered (GlobalDirections: d_loc_rest, RigidBodyConstraints.h)
        std::vector<double> s0, s1, s2;                 // up to three scalar columns
        std::vector<double> stiffness, tolerance, is_active;
        bool values_dirty = false;
    };
    Table tables[N_KINDS];
    double stiffness_hard_multiplier = 2.0, stiffness_soft_multiplier = 1.05, soft_constraint_capacity_hardening_point = 0.5;

    EnergyRigidBodyConstraints(Stark& stark, spRigidBodyDynamics rb);
    int add(Kind kind, int a, int b, const Vec3* vecs, int n_vecs, const double* scalars, int n_scalars, double stiffness, double tolerance);
    // RBCGlobalPointHandler::set_global_target_point / RBCGlobalDirectionHandler::set_rotation (rigidbody_constraints_ui.h:72,91)
    void set_global_target_point(int idx, const Vec3& p);
    void set_global_direction_rotation(int idx, const Mat3& R);
    // The handlers' measurements (rigidbody_constraints_ui.h:75-330 with the static functions of RigidBodyConstraints.h): {violation, force or
    // torque} of constraint `idx` of `kind` at the current state; which = 1 selects the damper of a damped spring.
    std::array<double, 2> measure(Kind kind, int idx, int which = 0) const;
    void register_potentials(mistark_ctx* ctx) override;

private:
    Stark& stark;
    spRigidBodyDynamics rb;
    int id_stiffness[N_KINDS];
    int id_v0[N_KINDS], id_v1[N_KINDS];
    bool _is_converged_state_valid();
    void _on_time_step_accepted();
    bool _adjust_constraints_stiffness(double cap, double multiplier, bool are_positions_set);
    void _upload_stiffness();
};

// stark::RigidBodies (stark/src/models/rigidbodies/RigidBodies.*)
class RigidBodies
{
    double default_stiffness = 1e6, default_tolerance_in_m = 0.001, default_tolerance_in_deg = 1.0;
    spRigidBodyDynamics rb;

public:
    std::shared_ptr<EnergyRigidBodyInertia> inertia;
    std::shared_ptr<EnergyRigidBodyConstraints> constraints;
    RigidBodies(Stark& stark, spRigidBodyDynamics rb);
    void set_default_constraint_stiffness(double k) { default_stiffness = k; }
    void set_default_constraint_distance_tolerance(double t) { default_tolerance_in_m = t; }
    void set_default_constraint_angle_tolerance(double t) { default_tolerance_in_deg = t; }
    RigidBodyHandler add(double mass, const Mat3& inertia_local);
    RigidBodyHandler handler(int idx) { return RigidBodyHandler(rb.get(), inertia.get(), idx); }
    // each returns the index of the (last) base constraint it created
    int add_constraint_global_point(const RigidBodyHandler& body, const Vec3& p_glob);
    int add_constraint_global_direction(const RigidBodyHandler& body, const Vec3& d_glob);
    int add_constraint_point(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p_glob);
    int add_constraint_point_on_axis(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p_glob, const Vec3& d_glob);
    int add_constraint_distance(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& a_glob, const Vec3& b_glob);
    int add_constraint_distance_limits(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& a_glob, const Vec3& b_glob, double min_distance, double max_distance);
    int add_constraint_direction(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& d_glob);
    int add_constraint_angle_limit(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& d_glob, double admissible_angle_deg);
    int add_constraint_spring(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& a_glob, const Vec3& b_glob, double stiffness, double damping = 0.0);
    int add_constraint_linear_velocity(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& d_glob, double target_v, double max_force, double delay = 0.01);
    int add_constraint_angular_velocity(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& d_glob, double target_w, double max_abs_torque, double delay = 0.01);
    void add_constraint_fix(const RigidBodyHandler& body);
    // RBCFixHandler::set_transformation (rigidbody_constraints_ui.h:369-375) on the three base constraints of a fix
    void set_fix_transformation(int anchor_point, int z_lock, int x_lock, const Vec3& translation, const Mat3& rotation);
    void add_constraint_attachment(const RigidBodyHandler& a, const RigidBodyHandler& b);
    void add_constraint_point_with_angle_limit(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p_glob, const Vec3& d_glob, double admissible_angle_deg);
    void add_constraint_hinge(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p_glob, const Vec3& d_glob);
    void add_constraint_hinge_with_angle_limit(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p_glob, const Vec3& d_glob, double admissible_angle_deg);
    void add_constraint_slider(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p_glob, const Vec3& d_glob);
    void add_constraint_prismatic_slider(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p_glob, const Vec3& d_glob);
    void add_constraint_spring_with_limits(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& a_glob, const Vec3& b_glob, double stiffness, double min_length, double max_length, double damping = 0.0);
    void add_constraint_prismatic_press(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p_glob, const Vec3& d_glob, double target_v, double max_force, double delay = 0.01);
    void add_constraint_motor(const RigidBodyHandler& a, const RigidBodyHandler& b, const Vec3& p_glob, const Vec3& d_glob, double target_w, double max_torque, double delay = 0.01);
};

// ---- stark::EnergyFrictionalContact (stark/src/models/interactions/EnergyFrictionalContact.*) ---------------------------------------
// Host half of the contact model: parameters, collision mesh registration, stiffness adaptation and the callbacks; detection,
// classification and the 35 potentials' tables are the device contact module's (include/mistark_contact.h).
class EnergyFrictionalContact : public Registrable
{
public:
    struct GlobalParams
    {
        double default_contact_thickness = -1.0;
        double min_contact_stiffness = 1e6, max_contact_stiffness = 1e20, friction_stick_slide_threshold = 0.1;
        bool collisions_enabled = true, friction_enabled = true, triangle_point_enabled = true, edge_edge_enabled = true, intersection_test_enabled = true;
    };
    struct Params
    {
        double contact_thickness = 0.0;
    };
    struct Handler
    {
        EnergyFrictionalContact* model = nullptr;
        int idx = -1;
        int get_idx() const { return idx; }
        bool is_valid() const { return model != nullptr; }
        void set_contact_thickness(double t) { model->set_contact_thickness(*this, t); }
        void set_friction(const Handler& other, double mu) { model->set_friction(*this, other, mu); }
        void disable_collision(const Handler& other) { model->disable_collision(*this, other); }
    };
    EnergyFrictionalContact(Stark& stark, spPointDynamics dyn, spRigidBodyDynamics rb);
    GlobalParams get_global_params() const { return global_params; }
    void set_global_params(const GlobalParams& p);
    double get_contact_stiffness() const { return contact_stiffness; }
    Handler add_triangles(const PointSetHandler& set, const std::vector<std::array<int, 3>>& triangles, const Params& params);
    Handler add_triangles(const PointSetHandler& set, const std::vector<std::array<int, 3>>& triangles, const std::vector<int>& point_set_map, const Params& params);
    Handler add_edges(const PointSetHandler& set, const std::vector<std::array<int, 2>>& edges, const Params& params);
    Handler add_triangles(const RigidBodyHandler& rb, const std::vector<Vec3>& vertices, const std::vector<std::array<int, 3>>& triangles, const Params& params);
    void set_contact_thickness(const Handler& obj, double t);
    void set_friction(const Handler& a, const Handler& b, double mu);
    void disable_collision(const Handler& a, const Handler& b);
    bool is_empty() const { return meshes.empty(); }
    bool is_active() const { return is_initialized; }
    int64_t last_n_contacts = 0, last_n_friction_contacts = 0, n_detections = 0;
    void register_potentials(mistark_ctx* ctx) override;

private:
    struct Mesh
    {
        int kind, idx_in_ps;
        std::vector<int32_t> verts;  // index in the physical system's vertex array
        std::vector<std::array<int, 3>> triangles;
        std::vector<std::array<int, 2>> edges;
    };
    Stark& stark;
    spPointDynamics dyn;
    spRigidBodyDynamics rb;
    bool is_initialized = false;
    GlobalParams global_params;
    double contact_stiffness = 1e6;
    std::vector<double> contact_thicknesses;
    std::vector<Vec3> rigidbody_local_vertices;
    std::vector<Mesh> meshes;
    std::vector<std::array<double, 3>> friction_pairs;  // a, b, mu
    std::vector<std::array<int, 2>> disabled_pairs;
    int id_k = -1;
    double k_uploaded = -1.0;
    double _init_contact_thickness(double t) const;
    void _sync_stiffness();
    void _before_time_step();
    void _before_energy_evaluation();
    bool _is_intermediate_state_valid(bool is_initial_check);
    void _on_intermediate_state_invalid();
    void _on_time_step_accepted();
    bool _should_continue_execution();
};
using ContactHandler = EnergyFrictionalContact::Handler;

// ---- stark::EnergyAttachments (stark/src/models/interactions/EnergyAttachments.*) ------------------------------------------------------
// Penalty springs gluing material points: deformable point to point / edge / triangle, edge to edge, and rigid body to deformable point.
class EnergyAttachments : public Registrable
{
public:
    struct Params
    {
        double stiffness = 1e3;
        double tolerance = std::numeric_limits<double>::max();
    };
    MISTARK_HANDLER(EnergyAttachments, Params)
    EnergyAttachments(Stark& stark, spPointDynamics dyn, spRigidBodyDynamics rb);
    Handler add(const PointSetHandler& set_0, const PointSetHandler& set_1, const std::vector<int>& points_0, const std::vector<int>& points_1, const Params& params);
    Handler add(const PointSetHandler& set_0, const PointSetHandler& set_1, const std::vector<int>& points, const std::vector<std::array<int, 2>>& edges,
                const std::vector<std::array<double, 2>>& bary, const Params& params);
    Handler add(const PointSetHandler& set_0, const PointSetHandler& set_1, const std::vector<int>& points, const std::vector<std::array<int, 3>>& triangles,
                const std::vector<std::array<double, 3>>& bary, const Params& params);
    Handler add(const PointSetHandler& set_0, const PointSetHandler& set_1, const std::vector<std::array<int, 2>>& edges_0, const std::vector<std::array<int, 2>>& edges_1,
                const std::vector<std::array<double, 2>>& bary_0, const std::vector<std::array<double, 2>>& bary_1, const Params& params);
    Handler add(const RigidBodyHandler& rb, const PointSetHandler& set, const std::vector<Vec3>& rb_points_loc, const std::vector<int>& set_points, const Params& params);
    Handler add(const RigidBodyHandler& rb, const PointSetHandler& set, const std::vector<int>& points, const Params& params);
    // EnergyAttachments.cpp:229-297, 334-360: attach the points closer than `distance` to a triangle mesh at its nearest vertex / edge / face
    struct MultiHandler
    {
        std::array<Handler, 3> handlers;  // point-point, point-edge, point-triangle
    };
    MultiHandler add_by_distance(const PointSetHandler& set_0, const PointSetHandler& set_1, const std::vector<int>& points, const std::vector<std::array<int, 3>>& triangles,
                                 double distance, const Params& params);
    Handler add_by_distance(const RigidBodyHandler& rb, const PointSetHandler& set, const std::vector<Vec3>& loc_vertices, const std::vector<std::array<int, 3>>& triangles,
                            const std::vector<int>& set_points, double distance, const Params& params);
    Params get_params(const Handler& h) const;
    void set_params(const Handler& h, const Params& p);
    void register_potentials(mistark_ctx* ctx) override;

private:
    enum Type { PointPoint = 0, PointEdge, PointTriangle, EdgeEdge, RigidDeformable, N_TYPES };
    Stark& stark;
    spPointDynamics dyn;
    spRigidBodyDynamics rb;
    std::vector<std::array<int32_t, 3>> conn_p_p;  // group, a, b
    std::vector<std::array<int32_t, 5>> conn_p_e;  // idx, group, p, e0, e1
    std::vector<std::array<int32_t, 6>> conn_p_t;  // idx, group, p, t0, t1, t2
    std::vector<std::array<int32_t, 6>> conn_e_e;  // idx, group, ea0, ea1, eb0, eb1
    std::vector<std::array<int32_t, 4>> conn_rb_d; // idx, group, rb, p
    std::vector<std::array<double, 2>> bary_p_e, bary_e_e_0, bary_e_e_1;
    std::vector<std::array<double, 3>> bary_p_t;
    std::vector<Vec3> rb_points_loc;
    std::vector<double> stiffness[N_TYPES], tolerance[N_TYPES];  // per group
    int id_stiffness[N_TYPES] = {-1, -1, -1, -1, -1};
    std::vector<std::pair<int, int>> handlers_map;  // handler -> (type, group)
    Handler new_handler(int type, const Params& params, int& group);
    bool _is_converged_state_valid();
};
class MeshOutput;
struct Interactions
{
    std::shared_ptr<MeshOutput> output;  // (the reference keeps one output object per subsystem; one serves both here)
    std::shared_ptr<EnergyAttachments> attachments;
    std::shared_ptr<EnergyFrictionalContact> contact;
};

// ---- frame output (stark/src/models/deformables/DeformablesMeshOutput.*, rigidbodies/RigidBodiesMeshOutput.*) ------
// Legacy binary VTK files, one per output label and frame, written from the host mirror of the state (positions leave the device once
// per frame, not per step). Meshes with the same label are merged into one file.
void write_VTK(const std::string& path, const std::vector<Vec3>& vertices, const int* conn, size_t n_cells, int nodes_per_cell);
class MeshOutput
{
public:
    MeshOutput(Stark& stark, spPointDynamics dyn, spRigidBodyDynamics rb);
    void add_point_set(const std::string& label, const PointSetHandler& set);
    void add_segment_mesh(const std::string& label, const PointSetHandler& set, const std::vector<std::array<int, 2>>& conn);
    void add_triangle_mesh(const std::string& label, const PointSetHandler& set, const std::vector<std::array<int, 3>>& conn);
    // triangles in the numbering of a vertex subset (point_set_map: local vertex -> index in the point set): the surface of a volume, written
    // with its own vertices only (DeformablesMeshOutput.cpp:49-60, DeformablesPresets.cpp:74-76)
    void add_triangle_mesh(const std::string& label, const PointSetHandler& set, const std::vector<std::array<int, 3>>& conn, const std::vector<int>& point_set_map);
    void add_tet_mesh(const std::string& label, const PointSetHandler& set, const std::vector<std::array<int, 4>>& conn);
    void add_triangle_mesh(const std::string& label, const RigidBodyHandler& rb, const std::vector<Vec3>& local_vertices, const std::vector<std::array<int, 3>>& conn);
    int frames_written = 0;

private:
    struct Mesh
    {
        std::string label;
        int nodes_per_cell = 0;
        int point_set = -1, rigid_body = -1;
        std::vector<Vec3> local_vertices;  // rigid bodies
        std::vector<int> conn;
        std::vector<int> point_set_map;    // non-empty: the mesh uses these points of the set only, in this order
    };
    Stark& stark;
    spPointDynamics dyn;
    spRigidBodyDynamics rb;
    std::vector<Mesh> meshes;
    void add(const std::string& label, int nodes_per_cell, int point_set, int rigid_body, const std::vector<Vec3>& loc, const int* conn, size_t n);
    void _write_frame();
};

// ---- stark::Deformables + presets + Simulation ---------------------------------------------------------------------------
struct Deformables
{
    spPointDynamics point_sets;
    std::shared_ptr<EnergyLumpedInertia> lumped_inertia;
    std::shared_ptr<EnergyPrescribedPositions> prescribed_positions;
    std::shared_ptr<EnergySegmentStrain> segment_strain;
    std::shared_ptr<EnergyTriangleStrain> triangle_strain;
    std::shared_ptr<EnergyDiscreteShells> discrete_shells;
    std::shared_ptr<EnergyTetStrain> tet_strain;
    Deformables(Stark& stark, spPointDynamics dyn);
};

namespace Line {
struct Params
{
    EnergyLumpedInertia::Params inertia;
    EnergySegmentStrain::Params strain;
    EnergyFrictionalContact::Params contact;
    static Params Elastic_Rubberband();  // stark/src/models/presets/deformables_preset_types.cpp:17-37
};
struct Handler
{
    PointSetHandler point_set;
    EnergyLumpedInertia::Handler inertia;
    EnergySegmentStrain::Handler strain;
    ContactHandler contact;
};
struct VCH
{
    std::vector<Vec3> vertices;
    std::vector<std::array<int, 2>> segments;
    Handler handler;
};
}  // namespace Line
namespace Surface {
struct Params
{
    EnergyLumpedInertia::Params inertia;
    EnergyTriangleStrain::Params strain;
    EnergyDiscreteShells::Params bending;
    EnergyFrictionalContact::Params contact;
    static Params Cotton_Fabric();  // stark/src/models/presets/deformables_preset_types.cpp:44-58
};
struct Handler
{
    PointSetHandler point_set;
    EnergyLumpedInertia::Handler inertia;
    EnergyTriangleStrain::Handler strain;
    EnergyDiscreteShells::Handler bending;
    ContactHandler contact;
};
struct VCH
{
    std::vector<Vec3> vertices;
    std::vector<std::array<int, 3>> triangles;
    Handler handler;
};
}  // namespace Surface
namespace Volume {
struct Params
{
    EnergyLumpedInertia::Params inertia;
    EnergyTetStrain::Params strain;
    EnergyFrictionalContact::Params contact;
    static Params Soft_Rubber();  // stark/src/models/presets/deformables_preset_types.cpp:70-80
};
struct Handler
{
    PointSetHandler point_set;
    EnergyLumpedInertia::Handler inertia;
    EnergyTetStrain::Handler strain;
    ContactHandler contact;
};
struct VCH
{
    std::vector<Vec3> vertices;
    std::vector<std::array<int, 4>> tets;
    Handler handler;
};
}  // namespace Volume

namespace RigidBody {
struct Handler
{
    RigidBodyHandler rigidbody;
    ContactHandler contact;
};
struct VCH
{
    std::vector<Vec3> vertices;
    std::vector<std::array<int, 3>> triangles;
    Handler handler;
};
}  // namespace RigidBody

class DeformablesPresets
{
    std::shared_ptr<Deformables> deformables;
    std::shared_ptr<Interactions> interactions;

public:
    DeformablesPresets(std::shared_ptr<Deformables> d, std::shared_ptr<Interactions> i) : deformables(d), interactions(i) {}
    Line::Handler add_line(const std::string& label, const std::vector<Vec3>& vertices, const std::vector<std::array<int, 2>>& segments, const Line::Params& params);
    Line::VCH add_line_as_segments(const std::string& label, const Vec3& begin, const Vec3& end, int n_segments, const Line::Params& params);
    Surface::Handler add_surface(const std::string& label, const std::vector<Vec3>& vertices, const std::vector<std::array<int, 3>>& triangles, const Surface::Params& params);
    Surface::VCH add_surface_grid(const std::string& label, const std::array<double, 2>& dim, const std::array<int, 2>& subdivisions, const Surface::Params& params);
    Volume::Handler add_volume(const std::string& label, const std::vector<Vec3>& vertices, const std::vector<std::array<int, 4>>& tets, const Volume::Params& params);
    Volume::VCH add_volume_grid(const std::string& label, const Vec3& dim, const std::array<int, 3>& subdivisions, const Volume::Params& params);
};
// stark::RigidBodyPresets (stark/src/models/presets/RigidBodyPresets.cpp)
class RigidBodyPresets
{
    std::shared_ptr<RigidBodies> rigidbodies;
    std::shared_ptr<Interactions> interactions;

public:
    RigidBodyPresets(std::shared_ptr<RigidBodies> r, std::shared_ptr<Interactions> i) : rigidbodies(r), interactions(i) {}
    RigidBody::Handler add(const std::string& label, double mass, const Mat3& inertia_local, const std::vector<Vec3>& vertices, const std::vector<std::array<int, 3>>& triangles,
                           const EnergyFrictionalContact::Params& contact = {});
    RigidBody::VCH add_box(const std::string& label, double mass, const Vec3& size, const EnergyFrictionalContact::Params& contact = {});
};
struct Presets
{
    std::shared_ptr<DeformablesPresets> deformables;
    std::shared_ptr<RigidBodyPresets> rigidbodies;
};

class Simulation
{
    Stark stark;

public:
    std::shared_ptr<Deformables> deformables;
    std::shared_ptr<RigidBodies> rigidbodies;
    std::shared_ptr<Interactions> interactions;
    std::shared_ptr<Presets> presets;
    explicit Simulation(const Settings& settings);
    Stark& get_stark() { return stark; }
    double get_time() const { return stark.current_time; }
    double get_time_step_size() const { return stark.dt; }
    void run_one_time_step() { stark.run_one_step(); }
    void run(double duration, std::function<void()> cb = nullptr) { stark.run(duration, cb); }
};

}  // namespace mistark
