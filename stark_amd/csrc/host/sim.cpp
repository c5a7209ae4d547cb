// host/sim.cpp — see sim.hpp. Reference citations are given per function.
#include "sim.hpp"
#include "bind.hpp"

#include <limits>
#include <chrono>
#include <cstdio>
#include <stdexcept>

namespace mistark {

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ======================================================================================================================
// Stark  (stark/src/core/Stark.cpp)
// ======================================================================================================================
Stark::Stark(const Settings& s) : settings(s)
{
    dt = settings.simulation.max_time_step_size;  // Stark.cpp:68
    gravity = settings.simulation.gravity;
}
Stark::~Stark()
{
    if (ctx) mistark_destroy(ctx);
}
void Stark::check(int rc) const
{
    if (rc < 0) throw std::runtime_error(std::string("mistark: ") + (ctx ? mistark_last_error(ctx) : "no context"));
}
void Stark::ensure_registered()
{
    if (!registration_dirty) return;
    mistark_ctx* old = ctx;
    if (old) {
        // keep the current device state: mirror it into the host arrays the new registration starts from
        for (auto* m : models)
            if (auto* pd = dynamic_cast<PointDynamics*>(m)) pd->mirror_to_host();
        ctx = nullptr;
    }
    // device < 0: a registration-only context (no GPU needed; tests/test_shim_cpu.py compares what this mirror registers with what the
    // reference's own classes register through the SymX shim)
    const int rc = settings.execution.device < 0 ? mistark_create_dry(&ctx) : mistark_create(settings.execution.device, &ctx);
    if (rc != 0) throw std::runtime_error("mistark_create failed (" + std::to_string(rc) + "): no MI355X visible; the hot path has no CPU fallback");
    if (settings.execution.world > 1 && old) {
        check(mistark_dist_move(ctx, old));  // (a communicator is created once: an RCCL unique id is single-use)
    } else if (settings.execution.world > 1) {
        const auto& ex = settings.execution;
        if (ex.local_group) check(mistark_dist_init_local(ctx, ex.local_group, ex.rank));
        else if (ex.ipc_comm) check(mistark_dist_init_ipc(ctx, ex.ipc_comm));
        else if (ex.rccl_unique_id.size() == 128) check(mistark_dist_init_rccl(ctx, ex.rank, ex.world, ex.rccl_unique_id.data()));
        else throw std::runtime_error("multi-GPU run without a communicator id");
    }
    for (auto* m : models) m->register_dofs(ctx);
    dt_array_id = mistark_array(ctx, &dt, 1, 1);
    check(dt_array_id);
    gravity_array_id = mistark_array(ctx, gravity.data(), 1, 3);
    check(gravity_array_id);
    for (auto* m : models) m->register_potentials(ctx);
    if (settings.execution.device >= 0) {
        // the rest positions of the points per block row (rigid bodies: NaN). Sharded runs partition the rows by them (recursive coordinate
        // bisection; rigid bodies keep the last rank), one GPU orders its solver's rows along a space-filling curve through them
        const int64_t nbr = mistark_ndofs(ctx) / 3;
        std::vector<double> xyz((size_t)(3 * nbr), std::numeric_limits<double>::quiet_NaN());
        for (auto* m : models)
            if (auto* pd = dynamic_cast<PointDynamics*>(m)) {
                const int64_t row0 = mistark_dof_set_first_row(ctx, pd->dof_set);
                if (row0 < 0) continue;
                for (size_t i = 0; i < pd->size(); i++)
                    for (int d = 0; d < 3; d++) xyz[(size_t)(3 * (row0 + (int64_t)i) + d)] = pd->X[i][d];
            }
        check(mistark_dist_set_row_coords(ctx, xyz.data(), nbr));
    }
    dt_uploaded = dt;
    registration_dirty = false;
    if (old) mistark_destroy(old);
}
void Stark::_initialize()
{
    // Stark.cpp:284-313 (no JIT: kernels are compiled ahead of time)
    is_init = true;
    ensure_registered();
    if (mistark_ndofs(ctx) == 0) throw std::runtime_error("Stark::_initialize(): no degrees of freedom");
    for (auto& f : callbacks->before_simulation) f();
    bool valid = true;
    for (auto& f : callbacks->newton->is_initial_state_valid) valid = valid && f();
    if (!valid) throw std::runtime_error("Initial state is not valid");
    _write_frame();  // Stark.cpp:303
}

namespace {
// trampoline from the C callbacks of mistark_newton_solve to the std::function lists
struct CbCtx
{
    SolverCallbacks* cb;
};
void cb_before_eval(void* u) { for (auto& f : ((CbCtx*)u)->cb->before_energy_evaluation) f(); }
int cb_initial_valid(void* u) { bool v = true; for (auto& f : ((CbCtx*)u)->cb->is_initial_state_valid) v = v && f(); return v; }
int cb_intermediate_valid(void* u) { bool v = true; for (auto& f : ((CbCtx*)u)->cb->is_intermediate_state_valid) v = v && f(); return v; }
void cb_on_invalid(void* u) { for (auto& f : ((CbCtx*)u)->cb->on_intermediate_state_invalid) f(); }
void cb_on_armijo(void* u) { for (auto& f : ((CbCtx*)u)->cb->on_armijo_fail) f(); }
// SolverCallbacks::run_is_converged starts from `false` and ANDs (solver_utils.h:51-58, :100-103): never true
int cb_is_converged(void* u) { bool v = false; for (auto& f : ((CbCtx*)u)->cb->is_converged) v = v && f(); return v; }
int cb_converged_valid(void* u) { bool v = true; for (auto& f : ((CbCtx*)u)->cb->is_converged_state_valid) v = v && f(); return v; }
double cb_max_step(void* u) { double s = 1.0; for (auto& f : ((CbCtx*)u)->cb->max_allowed_step) s = std::min(s, f()); return s; }
}  // namespace

// The first half of run_one_step up to the Newton solve, and the Newton callback that precedes every evaluation, as separate calls: a
// stage-by-stage comparison with the reference (tests/test_gpu_fullsize.py) puts the engine at a dumped state the way the reference's
// harness does (callbacks->run_before_time_step(); set_dofs; newton->run_before_energy_evaluation()).
void Stark::begin_time_step()
{
    if (!is_init) _initialize();
    ensure_registered();
    for (auto& f : callbacks->before_time_step) f();
    ensure_registered();
    if (dt != dt_uploaded) {
        check(mistark_upload(ctx, dt_array_id));
        dt_uploaded = dt;
    }
}
void Stark::before_energy_evaluation()
{
    for (auto& f : callbacks->newton->before_energy_evaluation) f();
}
bool Stark::run_one_step()
{
    struct StepTimer
    {
        double& acc;
        double t0;
        ~StepTimer() { acc += now_s() - t0; }
    } step_timer{total_step_time, now_s()};
    if (!is_init) _initialize();
    ensure_registered();
    // Stark.cpp:145
    bool cont = true;
    for (auto& f : callbacks->should_continue_execution) cont = cont && f();
    if (!cont) return false;
    for (auto& f : callbacks->before_time_step) f();  // Stark.cpp:154
    ensure_registered();                               // a callback may have added objects
    if (dt != dt_uploaded) {
        check(mistark_upload(ctx, dt_array_id));
        dt_uploaded = dt;
    }

    CbCtx cc{callbacks->newton.get()};
    mistark_newton_callbacks cb{};
    cb.user = &cc;
    SolverCallbacks& n = *callbacks->newton;
    if (!n.before_energy_evaluation.empty()) cb.before_energy_evaluation = cb_before_eval;
    if (!n.is_initial_state_valid.empty()) cb.is_initial_state_valid = cb_initial_valid;
    if (!n.is_intermediate_state_valid.empty()) cb.is_intermediate_state_valid = cb_intermediate_valid;
    if (!n.on_intermediate_state_invalid.empty()) cb.on_intermediate_state_invalid = cb_on_invalid;
    if (!n.on_armijo_fail.empty()) cb.on_armijo_fail = cb_on_armijo;
    if (!n.is_converged.empty()) cb.is_converged = cb_is_converged;
    if (!n.is_converged_state_valid.empty()) cb.is_converged_state_valid = cb_converged_valid;
    if (!n.max_allowed_step.empty()) cb.max_allowed_step = cb_max_step;

    const double t0 = now_s();
    mistark_newton_stats st{};
    const int result = mistark_newton_solve(ctx, &settings.newton, &cb, &st);  // Stark.cpp:158
    check(result);
    last_stats = st;
    last_newton_result = result;
    total_newton_iterations += st.newton_iterations;
    total_cg_iterations += st.cg_iterations;
    total_linear_solves += st.n_linear_solves;
    total_newton_time += st.t_total;
    total_linear_solve_time += st.t_linear_solve;
    total_eval_pgh_time += st.t_eval_pgh;
    total_eval_p_time += st.t_eval_p;
    total_project_time += st.t_project;
    total_assembly_time += st.t_assembly;
    total_callback_time += st.t_callbacks;
    total_evaluations += st.n_evaluations;

    if (result == MISTARK_SUCCESSFUL) {
        for (auto& f : callbacks->on_time_step_accepted) f();  // Stark.cpp:164-170
        for (auto& f : callbacks->after_time_step) f();
        current_time += dt;
        current_time_step++;
        _write_frame();  // Stark.cpp:201
        const double dt_taken = dt;
        dt = std::min(settings.simulation.max_time_step_size, dt * settings.simulation.time_step_size_success_multiplier);
        if (settings.output.enable_output) {
            const double runtime = now_s() - t0;
            std::printf("%d. dt: %5.2f ms | #newton: %2d | ph: %4.1f%% | #CG/newton: %4d | ls (cap|max|inv|bt): %2d|%2d|%2d|%2d| runtime: %6.1f ms | cr: %6.1f\n",
                        current_time_step - 1, 1000.0 * dt_taken, st.newton_iterations, 100.0 * st.projected_hessians_ratio,
                        st.newton_iterations > 0 ? st.cg_iterations / st.newton_iterations : 0, st.ls_cap_iterations, st.ls_max_iterations, st.ls_inv_iterations,
                        st.ls_bt_iterations, 1000.0 * runtime, runtime / dt_taken);
        }
        return true;
    }
    // failure: do not advance time (Stark.cpp:214-241)
    failed_steps++;
    if (result == MISTARK_INVALID_CONVERGED_STATE || result == MISTARK_TOO_MANY_INVALID_INTERMEDIATE_ITERATIONS) return true;  // a callback hardened a parameter: retry
    if (!settings.simulation.use_adaptive_time_step) return false;
    dt /= 2.0;
    if (dt < settings.simulation.time_step_size_lower_bound) return false;
    return true;
}
std::string Stark::get_frame_path(const std::string& name) const
{
    return settings.output.output_directory + "/" + settings.output.simulation_name + "_" + name + "_" + std::to_string(current_frame);
}
void Stark::_write_frame()
{
    // Stark.cpp:314-338
    if (!settings.output.enable_frame_writes) return;
    auto write_frame_impl = [&]() {
        if (settings.output.fps != 0)
            for (auto& f : callbacks->write_frame) f();
        if (settings.output.enable_output) std::printf("[Frame: %d] Time: %.3f s\n", current_frame, current_time);
        current_frame++;
    };
    if (settings.output.fps < 0) {
        write_frame_impl();
    } else if (current_frame == 0) {
        write_frame_impl();
        next_frame_time += 1.0 / (double)settings.output.fps;
    } else {
        while (current_time > next_frame_time + 100.0 * std::numeric_limits<double>::epsilon()) {
            write_frame_impl();
            next_frame_time += 1.0 / (double)settings.output.fps;
        }
    }
}
bool Stark::run(double duration, std::function<void()> callback)
{
    // Stark.cpp:79-132
    const double begin_time = current_time;
    const double t0 = now_s();
    bool success = false;
    while (current_time <= settings.execution.end_simulation_time && (current_time - begin_time) <= duration && current_frame <= settings.execution.end_frame &&
           (now_s() - t0) <= settings.execution.allowed_execution_time) {
        if (callback) callback();
        success = run_one_step();
        if (!success) break;
    }
    return success;
}

// ======================================================================================================================
// PointDynamics  (stark/src/models/deformables/PointDynamics.cpp)
// ======================================================================================================================
int PointSetHandler::get_begin() const { return dyn->get_begin(idx); }
int PointSetHandler::get_end() const { return dyn->get_end(idx); }
int PointSetHandler::size() const { return dyn->get_end(idx) - dyn->get_begin(idx); }
int PointSetHandler::get_global_index(int local_index) const { return dyn->get_global_index(idx, local_index); }
std::vector<int> PointSetHandler::all() const
{
    std::vector<int> v(size());
    for (int i = 0; i < size(); i++) v[i] = i;
    return v;
}
Vec3 PointSetHandler::get_position(int i) const { return dyn->x1[get_global_index(i)]; }
Vec3 PointSetHandler::get_rest_position(int i) const { return dyn->X[get_global_index(i)]; }
PointSetHandler& PointSetHandler::add_displacement(const Vec3& d, bool also_at_rest_pose)
{
    for (int i = get_begin(); i < get_end(); i++) {
        dyn->x1[i] = dyn->x1[i] + d;
        dyn->x0[i] = dyn->x0[i] + d;
        if (also_at_rest_pose) dyn->X[i] = dyn->X[i] + d;
    }
    dyn->mark_state_edited();
    return *this;
}
PointSetHandler& PointSetHandler::add_rotation(double angle_deg, const Vec3& axis, const Vec3& pivot, bool also_at_rest_pose)
{
    const Mat3 R = quat_to_matrix(quat_angle_axis(deg2rad(angle_deg), normalized(axis)));
    for (int i = get_begin(); i < get_end(); i++) {
        dyn->x0[i] = R * (dyn->x0[i] - pivot) + pivot;  // rotate_deg(point, R, pivot)
        dyn->x1[i] = dyn->x0[i];
        if (also_at_rest_pose) dyn->X[i] = dyn->x0[i];
    }
    dyn->mark_state_edited();
    return *this;
}

PointDynamics::PointDynamics(Stark& s) : stark(s)
{
    // PointDynamics.cpp:5-10: add_dof(v1, "soft.v1") + two callbacks
    set_begin.push_back(0);
    stark.add_model(this);
    stark.callbacks->add_before_time_step([this]() { _before_time_step(); });
    stark.callbacks->add_on_time_step_accepted([this]() { _on_time_step_accepted(); });
}
PointSetHandler PointDynamics::add(const std::vector<Vec3>& x, const std::string& label)
{
    // PointDynamics.cpp:12-26
    const int set = (int)set_begin.size() - 1;
    X.insert(X.end(), x.begin(), x.end());
    x0.insert(x0.end(), x.begin(), x.end());
    x1.insert(x1.end(), x.begin(), x.end());
    const Vec3 zero = {0.0, 0.0, 0.0};
    v0.insert(v0.end(), x.size(), zero);
    v1.insert(v1.end(), x.size(), zero);
    a.insert(a.end(), x.size(), zero);
    f.insert(f.end(), x.size(), zero);
    set_begin.push_back((int)X.size());
    labels.push_back(label.empty() ? "point_set_" + std::to_string(set) : label);
    stark.mark_registration_dirty();
    return PointSetHandler(this, set);
}
void PointDynamics::register_dofs(mistark_ctx* ctx)
{
    const int64_t n = (int64_t)size();
    dof_set = mistark_add_dof_set(ctx, "soft.v1", n ? v1[0].data() : nullptr, 3 * n);
    stark.check(dof_set);
    if (n == 0) return;
    stark.check(id_v1 = mistark_array(ctx, v1[0].data(), n, 3));
    stark.check(id_X = mistark_array(ctx, X[0].data(), n, 3));
    stark.check(id_x0 = mistark_array(ctx, x0[0].data(), n, 3));
    stark.check(id_v0 = mistark_array(ctx, v0[0].data(), n, 3));
    stark.check(id_a = mistark_array(ctx, a[0].data(), n, 3));
    stark.check(id_f = mistark_array(ctx, f[0].data(), n, 3));
}
void PointDynamics::_before_time_step()
{
    // v1 <- 0 (PointDynamics.cpp:58-62), on the device and in the host mirror
    if (size() == 0) return;
    std::fill(v1.begin(), v1.end(), Vec3{0.0, 0.0, 0.0});
    stark.check(mistark_array_fill(stark.ctx, id_v1, 0.0));
}
void PointDynamics::_on_time_step_accepted()
{
    // x1 = x0 + dt v1; x0 <- x1; v0 <- v1 (PointDynamics.cpp:64-78), on the device
    if (size() == 0) return;
    stark.check(mistark_array_axpby(stark.ctx, id_x0, 1.0, id_x0, stark.dt, id_v1));
    stark.check(mistark_array_axpby(stark.ctx, id_v0, 1.0, id_v1, 0.0, -1));
    if (stark.settings.execution.mirror_state_to_host) mirror_to_host();
}
void PointDynamics::mirror_to_host()
{
    if (size() == 0 || !stark.ctx || id_x0 < 0) return;
    stark.check(mistark_download(stark.ctx, id_x0));
    stark.check(mistark_download(stark.ctx, id_v0));
    stark.check(mistark_dofs_to_host_arrays(stark.ctx));
    x1 = x0;
}
void PointDynamics::mark_state_edited()
{
    if (stark.ctx) upload_state();
}
void PointDynamics::upload_state()
{
    if (size() == 0 || !stark.ctx || id_x0 < 0) return;
    for (int id : {id_X, id_x0, id_v0, id_a, id_f}) stark.check(mistark_upload(stark.ctx, id));
    stark.check(mistark_dofs_from_host_arrays(stark.ctx));
}

// ======================================================================================================================
// helpers
// ======================================================================================================================

// ======================================================================================================================
// EnergyLumpedInertia  (stark/src/models/deformables/point/EnergyLumpedInertia.cpp)
// ======================================================================================================================
EnergyLumpedInertia::EnergyLumpedInertia(Stark& s, spPointDynamics d) : stark(s), dyn(d) { stark.add_model(this); }
void EnergyLumpedInertia::register_potentials(mistark_ctx* ctx)
{
    if (conn.empty()) return;
    // binding order of EnergyLumpedInertia.cpp:17-27
    BindList B(ctx, stark);
    B.add_id(dyn->id_v1, 3, 1);
    B.add_id(dyn->id_x0, 3, 1);
    B.add_id(dyn->id_v0, 3, 1);
    B.add_id(dyn->id_a, 3, 1);
    B.add_id(dyn->id_f, 3, 1);
    B.add(lumped_volume.data(), (int64_t)lumped_volume.size(), 1, 0);
    B.add(density.data(), (int64_t)density.size(), 1, 2);
    B.add(damping.data(), (int64_t)damping.size(), 1, 2);
    B.add(is_quasistatic.data(), (int64_t)is_quasistatic.size(), 1, 2);
    B.add_id(stark.dt_array(), 1, -1);
    B.add_id(stark.gravity_array(), 3, -1);
    B.potential("EnergyLumpedInertia", conn);
}
EnergyLumpedInertia::Handler EnergyLumpedInertia::add(const PointSetHandler& set, const std::vector<int>& points, const std::vector<double>& vol, const Params& params)
{
    const int group = (int)density.size();
    density.push_back(params.density);
    damping.push_back(params.damping);
    is_quasistatic.push_back(params.quasistatic ? 1.0 : 0.0);
    for (size_t i = 0; i < points.size(); i++) {
        lumped_volume.push_back(vol[i]);
        conn.push_back({(int32_t)conn.size(), set.get_global_index(points[i]), group});
    }
    stark.mark_registration_dirty();
    return Handler(this, group);
}
EnergyLumpedInertia::Handler EnergyLumpedInertia::add(const PointSetHandler& set, const std::vector<double>& vol, const Params& params)
{
    const int n = set.size();
    if ((int)vol.size() != n) throw std::runtime_error("EnergyLumpedInertia::add(): lumped_volume.size() != n");
    std::vector<int> points;
    std::vector<double> nz;
    for (int i = 0; i < n; i++)
        if (vol[i] > 0.0) {
            points.push_back(i);
            nz.push_back(vol[i]);
        }
    return add(set, points, nz, params);
}
EnergyLumpedInertia::Handler EnergyLumpedInertia::add(const PointSetHandler& set, const std::vector<std::array<int, 2>>& segments, const Params& params)
{
    // EnergyLumpedInertia.cpp:95-115: half of each segment's rest length to either end point
    std::vector<double> vol(set.size(), 0.0);
    for (const auto& e : segments) {
        const auto g = set.get_global_indices(e);
        const double l = 0.5 * norm(dyn->X[g[0]] - dyn->X[g[1]]);
        vol[e[0]] += l;
        vol[e[1]] += l;
    }
    return add(set, vol, params);
}
EnergyLumpedInertia::Handler EnergyLumpedInertia::add(const PointSetHandler& set, const std::vector<std::array<int, 3>>& tris, const Params& params)
{
    std::vector<double> vol(set.size(), 0.0);
    for (const auto& t : tris) {
        const auto g = set.get_global_indices(t);
        const double l = triangle_area(dyn->X[g[0]], dyn->X[g[1]], dyn->X[g[2]]) / 3.0;
        for (int k = 0; k < 3; k++) vol[t[k]] += l;
    }
    return add(set, vol, params);
}
EnergyLumpedInertia::Handler EnergyLumpedInertia::add(const PointSetHandler& set, const std::vector<std::array<int, 4>>& tets, const Params& params)
{
    std::vector<double> vol(set.size(), 0.0);
    for (const auto& t : tets) {
        const auto g = set.get_global_indices(t);
        const double l = unsigned_tetra_volume(dyn->X[g[0]], dyn->X[g[1]], dyn->X[g[2]], dyn->X[g[3]]) / 4.0;
        for (int k = 0; k < 4; k++) vol[t[k]] += l;
    }
    return add(set, vol, params);
}
EnergyLumpedInertia::Params EnergyLumpedInertia::get_params(const Handler& h) const
{
    Params p;
    p.density = density[h.idx];
    p.damping = damping[h.idx];
    p.quasistatic = is_quasistatic[h.idx] > 0.5;
    return p;
}
void EnergyLumpedInertia::set_params(const Handler& h, const Params& p)
{
    density[h.idx] = p.density;
    damping[h.idx] = p.damping;
    is_quasistatic[h.idx] = p.quasistatic ? 1.0 : 0.0;
    stark.mark_registration_dirty();
}
double EnergyLumpedInertia::get_mass(const Handler& h) const
{
    double m = 0.0;
    for (size_t i = 0; i < conn.size(); i++)
        if (conn[i][2] == h.idx) m += density[h.idx] * lumped_volume[i];
    return m;
}

// ======================================================================================================================
// EnergyPrescribedPositions  (stark/src/models/deformables/point/EnergyPrescribedPositions.cpp)
// ======================================================================================================================
EnergyPrescribedPositions::EnergyPrescribedPositions(Stark& s, spPointDynamics d) : stark(s), dyn(d)
{
    stark.add_model(this);
    stark.callbacks->newton->add_is_converged_state_valid([this]() { return _is_converged_state_valid(); });
    stark.callbacks->add_before_time_step([this]() { _before_energy_evaluation(); });
}
void EnergyPrescribedPositions::register_potentials(mistark_ctx* ctx)
{
    if (conn.empty()) return;
    // binding order of EnergyPrescribedPositions.cpp:20-24
    BindList B(ctx, stark);
    B.add_id(dyn->id_v1, 3, 1);
    B.add_id(dyn->id_x0, 3, 1);
    B.add(target_positions[0].data(), (int64_t)target_positions.size(), 3, 0);
    id_target = B.b.back().array;
    B.add(stiffness.data(), (int64_t)stiffness.size(), 1, 2);
    id_stiffness = B.b.back().array;
    B.add_id(stark.dt_array(), 1, -1);
    B.potential("EnergyPrescribedPositions", conn);
    targets_dirty = stiffness_dirty = false;
}
void EnergyPrescribedPositions::_before_energy_evaluation()
{
    // host-side edits of targets / stiffness (scripted boundary conditions, hardening) reach the device before the solve
    if (!stark.ctx || id_target < 0) return;
    if (targets_dirty) stark.check(mistark_upload(stark.ctx, id_target));
    if (stiffness_dirty) stark.check(mistark_upload(stark.ctx, id_stiffness));
    targets_dirty = stiffness_dirty = false;
}
EnergyPrescribedPositions::Handler EnergyPrescribedPositions::add(const PointSetHandler& set, const std::vector<int>& points, const Params& params)
{
    const int group = (int)stiffness.size();
    stiffness.push_back(params.stiffness);
    tolerance.push_back(params.tolerance);
    const int begin = (int)target_positions.size();
    for (int p : points) {
        const int g = set.get_global_index(p);
        target_positions.push_back(dyn->x1[g]);
        rest_positions.push_back(dyn->x1[g]);
        conn.push_back({(int32_t)conn.size(), g, group});
    }
    group_begin_end.push_back({begin, (int)target_positions.size()});
    stark.mark_registration_dirty();
    return Handler(this, group);
}
EnergyPrescribedPositions::Handler EnergyPrescribedPositions::add_inside_aabb(const PointSetHandler& set, const Vec3& c, const Vec3& dim, const Params& params)
{
    // Eigen::AlignedBox3d::contains is inclusive on both ends
    std::vector<int> pts;
    for (int i = 0; i < set.size(); i++) {
        const Vec3 p = set.get_position(i);
        bool in = true;
        for (int k = 0; k < 3; k++) in = in && (p[k] >= c[k] - 0.5 * dim[k]) && (p[k] <= c[k] + 0.5 * dim[k]);
        if (in) pts.push_back(i);
    }
    return add(set, pts, params);
}
EnergyPrescribedPositions::Handler EnergyPrescribedPositions::add_outside_aabb(const PointSetHandler& set, const Vec3& c, const Vec3& dim, const Params& params)
{
    // EnergyPrescribedPositions.cpp:66-78
    std::vector<int> pts;
    for (int i = 0; i < set.size(); i++) {
        const Vec3 p = set.get_position(i);
        bool in = true;
        for (int k = 0; k < 3; k++) in = in && (p[k] >= c[k] - 0.5 * dim[k]) && (p[k] <= c[k] + 0.5 * dim[k]);
        if (!in) pts.push_back(i);
    }
    return add(set, pts, params);
}
EnergyPrescribedPositions::Params EnergyPrescribedPositions::get_params(const Handler& h) const { return Params{stiffness[h.idx], tolerance[h.idx]}; }
void EnergyPrescribedPositions::set_params(const Handler& h, const Params& p)
{
    stiffness[h.idx] = p.stiffness;
    tolerance[h.idx] = p.tolerance;
    stiffness_dirty = true;
}
void EnergyPrescribedPositions::set_transformation(const Handler& h, const Vec3& t, const std::array<double, 9>& R)
{
    const auto [b, e] = group_begin_end[h.idx];
    for (int i = b; i < e; i++) {
        const Vec3& r = rest_positions[i];
        target_positions[i] = {R[0] * r[0] + R[1] * r[1] + R[2] * r[2] + t[0], R[3] * r[0] + R[4] * r[1] + R[5] * r[2] + t[1], R[6] * r[0] + R[7] * r[1] + R[8] * r[2] + t[2]};
    }
    targets_dirty = true;
}
void EnergyPrescribedPositions::set_target_position(const Handler& h, int prescribed_idx, const Vec3& t)
{
    target_positions[group_begin_end[h.idx][0] + prescribed_idx] = t;
    targets_dirty = true;
}
bool EnergyPrescribedPositions::_is_converged_state_valid()
{
    // EnergyPrescribedPositions.cpp:131-156: harden the stiffness (x2) of the first violating group and fail the step
    bool any_finite = false;
    for (double t : tolerance) any_finite = any_finite || t < std::numeric_limits<double>::max();
    if (!any_finite || conn.empty()) return true;
    stark.check(mistark_dofs_to_host_arrays(stark.ctx));
    stark.check(mistark_download(stark.ctx, dyn->id_x0));
    for (const auto& c : conn) {
        const Vec3 x1 = dyn->get_x1(c[1], stark.dt);
        const Vec3 d = x1 - target_positions[c[0]];
        const double tol = tolerance[c[2]];
        if (dot(d, d) > tol * tol) {
            stiffness[c[2]] *= 2.0;
            stark.check(mistark_upload(stark.ctx, id_stiffness));
            return false;
        }
    }
    return true;
}

// ======================================================================================================================
// EnergyTetStrain  (stark/src/models/deformables/volume/EnergyTetStrain.cpp)
// ======================================================================================================================
EnergyTetStrain::EnergyTetStrain(Stark& s, spPointDynamics d) : stark(s), dyn(d) { stark.add_model(this); }
void EnergyTetStrain::register_potentials(mistark_ctx* ctx)
{
    const int64_t ng = (int64_t)youngs_modulus.size();
    if (!conn_complete.empty()) {
        // EnergyTetStrain.cpp:20-29
        BindList B(ctx, stark);
        for (int k = 0; k < 4; k++) B.add_id(dyn->id_v1, 3, 2 + k);
        for (int k = 0; k < 4; k++) B.add_id(dyn->id_x0, 3, 2 + k);
        for (int k = 0; k < 4; k++) B.add_id(dyn->id_X, 3, 2 + k);
        B.add(scale.data(), ng, 1, 1);
        B.add(youngs_modulus.data(), ng, 1, 1);
        B.add(poissons_ratio.data(), ng, 1, 1);
        B.add(strain_limit.data(), ng, 1, 1);
        B.add(strain_limit_stiffness.data(), ng, 1, 1);
        B.add(strain_damping.data(), ng, 1, 1);
        B.add_id(stark.dt_array(), 1, -1);
        B.potential("EnergyTetStrain", conn_complete);
    }
    if (!conn_elasticity_only.empty()) {
        // EnergyTetStrain.cpp:87-93
        BindList B(ctx, stark);
        for (int k = 0; k < 4; k++) B.add_id(dyn->id_v1, 3, 2 + k);
        for (int k = 0; k < 4; k++) B.add_id(dyn->id_x0, 3, 2 + k);
        for (int k = 0; k < 4; k++) B.add_id(dyn->id_X, 3, 2 + k);
        B.add(scale.data(), ng, 1, 1);
        B.add(youngs_modulus.data(), ng, 1, 1);
        B.add(poissons_ratio.data(), ng, 1, 1);
        B.add_id(stark.dt_array(), 1, -1);
        B.potential("EnergyTetStrain_Elasticity_Only", conn_elasticity_only);
    }
}
EnergyTetStrain::Handler EnergyTetStrain::add(const PointSetHandler& set, const std::vector<std::array<int, 4>>& tets, const Params& p)
{
    const int group = (int)youngs_modulus.size();
    elasticity_only.push_back(p.elasticity_only);
    scale.push_back(p.scale);
    youngs_modulus.push_back(p.youngs_modulus);
    poissons_ratio.push_back(p.poissons_ratio);
    strain_damping.push_back(p.damping);
    strain_limit.push_back(p.strain_limit);
    strain_limit_stiffness.push_back(p.strain_limit_stiffness);
    auto& conn = p.elasticity_only ? conn_elasticity_only : conn_complete;
    for (const auto& t : tets) {
        const auto g = set.get_global_indices(t);
        conn.push_back({(int32_t)conn.size(), group, g[0], g[1], g[2], g[3]});
    }
    stark.mark_registration_dirty();
    return Handler(this, group);
}
EnergyTetStrain::Params EnergyTetStrain::get_params(const Handler& h) const
{
    const int g = h.idx;
    Params p;
    p.elasticity_only = elasticity_only[g];
    p.scale = scale[g];
    p.youngs_modulus = youngs_modulus[g];
    p.poissons_ratio = poissons_ratio[g];
    p.damping = strain_damping[g];
    p.strain_limit = strain_limit[g];
    p.strain_limit_stiffness = strain_limit_stiffness[g];
    return p;
}
void EnergyTetStrain::set_params(const Handler& h, const Params& p)
{
    const int g = h.idx;
    if ((bool)elasticity_only[g] != p.elasticity_only) throw std::runtime_error("EnergyTetStrain::set_params(): elasticity_only cannot be changed");
    scale[g] = p.scale;
    youngs_modulus[g] = p.youngs_modulus;
    poissons_ratio[g] = p.poissons_ratio;
    strain_damping[g] = p.damping;
    strain_limit[g] = p.strain_limit;
    strain_limit_stiffness[g] = p.strain_limit_stiffness;
    stark.mark_registration_dirty();
}

// ======================================================================================================================
// EnergyTriangleStrain  (stark/src/models/deformables/surface/EnergyTriangleStrain.cpp)
// ======================================================================================================================
EnergyTriangleStrain::EnergyTriangleStrain(Stark& s, spPointDynamics d) : stark(s), dyn(d) { stark.add_model(this); }
void EnergyTriangleStrain::register_potentials(mistark_ctx* ctx)
{
    const int64_t ng = (int64_t)youngs_modulus.size();
    for (int full = 1; full >= 0; full--) {
        auto& conn = full ? conn_complete : conn_elasticity_only;
        if (conn.empty()) continue;
        BindList B(ctx, stark);
        for (int k = 0; k < 3; k++) B.add_id(dyn->id_v1, 3, 2 + k);
        for (int k = 0; k < 3; k++) B.add_id(dyn->id_x0, 3, 2 + k);
        for (int k = 0; k < 3; k++) B.add_id(dyn->id_X, 3, 2 + k);
        B.add(scale.data(), ng, 1, 1);
        B.add(thickness.data(), ng, 1, 1);
        B.add(youngs_modulus.data(), ng, 1, 1);
        B.add(poissons_ratio.data(), ng, 1, 1);
        if (full) {
            B.add(strain_damping.data(), ng, 1, 1);
            B.add(strain_limit.data(), ng, 1, 1);
            B.add(strain_limit_stiffness.data(), ng, 1, 1);
        }
        B.add(inflation.data(), ng, 1, 1);
        B.add_id(stark.dt_array(), 1, -1);
        B.potential(full ? "EnergyTriangleStrain" : "EnergyTriangleStrain_Elasticity_Only", conn);
    }
}
EnergyTriangleStrain::Handler EnergyTriangleStrain::add(const PointSetHandler& set, const std::vector<std::array<int, 3>>& tris, const Params& p)
{
    const int group = (int)youngs_modulus.size();
    elasticity_only.push_back(p.elasticity_only);
    scale.push_back(p.scale);
    thickness.push_back(p.thickness);
    youngs_modulus.push_back(p.youngs_modulus);
    poissons_ratio.push_back(p.poissons_ratio);
    strain_damping.push_back(p.damping);
    strain_limit.push_back(p.strain_limit);
    strain_limit_stiffness.push_back(p.strain_limit_stiffness);
    inflation.push_back(p.inflation);
    auto& conn = p.elasticity_only ? conn_elasticity_only : conn_complete;
    for (const auto& t : tris) {
        const auto g = set.get_global_indices(t);
        conn.push_back({(int32_t)conn.size(), group, g[0], g[1], g[2]});
    }
    stark.mark_registration_dirty();
    return Handler(this, group);
}
EnergyTriangleStrain::Params EnergyTriangleStrain::get_params(const Handler& h) const
{
    const int g = h.idx;
    Params p;
    p.elasticity_only = elasticity_only[g];
    p.scale = scale[g];
    p.thickness = thickness[g];
    p.youngs_modulus = youngs_modulus[g];
    p.poissons_ratio = poissons_ratio[g];
    p.damping = strain_damping[g];
    p.strain_limit = strain_limit[g];
    p.strain_limit_stiffness = strain_limit_stiffness[g];
    p.inflation = inflation[g];
    return p;
}
void EnergyTriangleStrain::set_params(const Handler& h, const Params& p)
{
    const int g = h.idx;
    if ((bool)elasticity_only[g] != p.elasticity_only) throw std::runtime_error("EnergyTriangleStrain::set_params(): elasticity_only cannot be changed");
    scale[g] = p.scale;
    thickness[g] = p.thickness;
    youngs_modulus[g] = p.youngs_modulus;
    poissons_ratio[g] = p.poissons_ratio;
    strain_damping[g] = p.damping;
    strain_limit[g] = p.strain_limit;
    strain_limit_stiffness[g] = p.strain_limit_stiffness;
    inflation[g] = p.inflation;
    stark.mark_registration_dirty();
}

// ======================================================================================================================
// EnergyDiscreteShells  (stark/src/models/deformables/surface/EnergyDiscreteShells.cpp)
// ======================================================================================================================
EnergyDiscreteShells::EnergyDiscreteShells(Stark& s, spPointDynamics d) : stark(s), dyn(d) { stark.add_model(this); }
void EnergyDiscreteShells::register_potentials(mistark_ctx* ctx)
{
    const int64_t ng = (int64_t)bending_stiffness.size();
    if (!conn_complete.empty()) {
        // EnergyDiscreteShells.cpp:33-41
        BindList B(ctx, stark);
        for (int k = 0; k < 4; k++) B.add_id(dyn->id_v1, 3, 2 + k);
        for (int k = 0; k < 4; k++) B.add_id(dyn->id_x0, 3, 2 + k);
        B.add(rest_dihedral_angle_rad.data(), (int64_t)rest_dihedral_angle_rad.size(), 1, 0);
        B.add(rest_edge_length.data(), (int64_t)rest_edge_length.size(), 1, 0);
        B.add(rest_height.data(), (int64_t)rest_height.size(), 1, 0);
        B.add(scale.data(), ng, 1, 1);
        B.add(bending_stiffness.data(), ng, 1, 1);
        B.add(bending_damping.data(), ng, 1, 1);
        B.add_id(stark.dt_array(), 1, -1);
        B.potential("EnergyDiscreteShells", conn_complete);
    }
    if (!conn_flat_rest.empty()) {
        // EnergyDiscreteShells.cpp:71-76
        BindList B(ctx, stark);
        for (int k = 0; k < 4; k++) B.add_id(dyn->id_v1, 3, 2 + k);
        for (int k = 0; k < 4; k++) B.add_id(dyn->id_x0, 3, 2 + k);
        B.add(bergou_K[0].data(), (int64_t)bergou_K.size(), 4, 0);
        B.add(bergou_coef.data(), (int64_t)bergou_coef.size(), 1, 0);
        B.add(bending_stiffness.data(), ng, 1, 1);
        B.add_id(stark.dt_array(), 1, -1);
        B.potential("EnergyBendingFlat", conn_flat_rest);
    }
}
EnergyDiscreteShells::Handler EnergyDiscreteShells::add(const PointSetHandler& set, const std::vector<std::array<int, 3>>& tris, const Params& p)
{
    // EnergyDiscreteShells.cpp:94-171. NOTE (reference behaviour kept): the per-hinge tables (rest angle / length / height
    // and Bergou K / coef) are all appended for every hinge, whichever table its connectivity row goes to, and indexed by
    // the row's own "idx" column.
    const int group = (int)bending_stiffness.size();
    scale.push_back(p.scale);
    bending_stiffness.push_back(p.stiffness);
    bending_damping.push_back(p.damping);
    flat_rest_angle.push_back(p.flat_rest_angle);
    if (p.flat_rest_angle && p.scale != 1.0) throw std::runtime_error("EnergyDiscreteShells::add(): scale must be 1.0 if flat_rest_angle");
    std::vector<std::array<int, 4>> hinges;
    find_internal_angles(hinges, tris, set.size());
    auto cot = [](const Vec3& v, const Vec3& w) { return dot(v, w) / norm(cross(v, w)); };
    auto& conn = p.flat_rest_angle ? conn_flat_rest : conn_complete;
    for (const auto& h : hinges) {
        const auto g = set.get_global_indices(h);
        conn.push_back({(int32_t)conn.size(), group, g[0], g[1], g[2], g[3]});
        const Vec3 e0 = dyn->X[g[1]] - dyn->X[g[0]], e1 = dyn->X[g[2]] - dyn->X[g[0]], e2 = dyn->X[g[3]] - dyn->X[g[0]];
        const Vec3 e3 = dyn->X[g[2]] - dyn->X[g[1]], e4 = dyn->X[g[3]] - dyn->X[g[1]];
        const double len = norm(e0);
        rest_edge_length.push_back(len);
        const Vec3 n0 = cross(e0, e1), n1 = -1.0 * cross(e0, e2);
        const double nn0 = norm(n0), nn1 = norm(n1);
        rest_dihedral_angle_rad.push_back(std::acos((1.0 - 1e-12) * dot((1.0 / nn0) * n0, (1.0 / nn1) * n1)));
        const double A0 = 0.5 * nn0, A1 = 0.5 * nn1;
        rest_height.push_back(1.0 / 6.0 * (2.0 * A0 / len + 2.0 * A1 / len));
        const Vec3 me0 = -1.0 * e0;
        const double c01 = cot(e0, e1), c02 = cot(e0, e2), c03 = cot(me0, e3), c04 = cot(me0, e4);
        bergou_coef.push_back(3.0 / (A0 + A1) * 0.5);
        bergou_K.push_back({c03 + c04, c01 + c02, -c01 - c03, -c02 - c04});
    }
    stark.mark_registration_dirty();
    return Handler(this, group);
}
EnergyDiscreteShells::Params EnergyDiscreteShells::get_params(const Handler& h) const
{
    Params p;
    p.scale = scale[h.idx];
    p.stiffness = bending_stiffness[h.idx];
    p.damping = bending_damping[h.idx];
    p.flat_rest_angle = flat_rest_angle[h.idx];
    return p;
}
void EnergyDiscreteShells::set_params(const Handler& h, const Params& p)
{
    if ((bool)flat_rest_angle[h.idx] != p.flat_rest_angle) throw std::runtime_error("EnergyDiscreteShells::set_params(): flat_rest_angle cannot be changed");
    scale[h.idx] = p.scale;
    bending_stiffness[h.idx] = p.stiffness;
    bending_damping[h.idx] = p.damping;
    stark.mark_registration_dirty();
}

// ======================================================================================================================
// Deformables, presets, Simulation
// ======================================================================================================================
Deformables::Deformables(Stark& stark, spPointDynamics dyn) : point_sets(dyn)
{
    // construction order = potential registration order of the reference (stark/src/models/deformables/Deformables.cpp)
    lumped_inertia = std::make_shared<EnergyLumpedInertia>(stark, dyn);
    prescribed_positions = std::make_shared<EnergyPrescribedPositions>(stark, dyn);
    segment_strain = std::make_shared<EnergySegmentStrain>(stark, dyn);
    triangle_strain = std::make_shared<EnergyTriangleStrain>(stark, dyn);
    discrete_shells = std::make_shared<EnergyDiscreteShells>(stark, dyn);
    tet_strain = std::make_shared<EnergyTetStrain>(stark, dyn);
}
Surface::Params Surface::Params::Cotton_Fabric()
{
    Params p;
    p.inertia.density = 0.2;
    p.inertia.damping = 0.1;
    p.strain.elasticity_only = false;
    p.strain.thickness = 0.001;
    p.strain.youngs_modulus = 5e3;
    p.strain.poissons_ratio = 0.3;
    p.strain.strain_limit = 0.1;
    p.strain.strain_limit_stiffness = 1e6;
    p.strain.damping = 0.1 * p.strain.thickness * p.strain.youngs_modulus;
    p.bending.flat_rest_angle = true;
    p.bending.stiffness = 1e-6;
    p.bending.damping = 0.1 * p.bending.stiffness;
    return p;
}
Volume::Params Volume::Params::Soft_Rubber()
{
    Params p;
    p.inertia.density = 1000.0;
    p.inertia.damping = 0.1;
    p.strain.elasticity_only = false;
    p.strain.youngs_modulus = 1e4;
    p.strain.poissons_ratio = 0.3;
    p.strain.strain_limit = 1.0;
    p.strain.strain_limit_stiffness = 1e2;
    p.strain.damping = 0.0;
    return p;
}
Surface::Handler DeformablesPresets::add_surface(const std::string& label, const std::vector<Vec3>& V, const std::vector<std::array<int, 3>>& T, const Surface::Params& p)
{
    // DeformablesPresets.cpp:31-44
    PointSetHandler ps = deformables->point_sets->add(V, label);
    auto inertia = deformables->lumped_inertia->add(ps, T, p.inertia);
    auto strain = deformables->triangle_strain->add(ps, T, p.strain);
    auto bending = deformables->discrete_shells->add(ps, T, p.bending);
    ContactHandler contact = interactions->contact->add_triangles(ps, T, p.contact);
    if (!label.empty() && interactions->output) interactions->output->add_triangle_mesh(label, ps, T);
    return {ps, inertia, strain, bending, contact};
}
Surface::VCH DeformablesPresets::add_surface_grid(const std::string& label, const std::array<double, 2>& dim, const std::array<int, 2>& sub, const Surface::Params& p)
{
    std::vector<Vec3> V;
    std::vector<std::array<int, 3>> T;
    generate_triangle_grid(V, T, {0.0, 0.0}, dim, sub);
    auto h = add_surface(label, V, T, p);
    return {V, T, h};
}
Volume::Handler DeformablesPresets::add_volume(const std::string& label, const std::vector<Vec3>& V, const std::vector<std::array<int, 4>>& T, const Volume::Params& p)
{
    // DeformablesPresets.cpp:65-79: the collision mesh is the surface of the tet mesh
    PointSetHandler ps = deformables->point_sets->add(V, label);
    auto inertia = deformables->lumped_inertia->add(ps, T, p.inertia);
    auto strain = deformables->tet_strain->add(ps, T, p.strain);
    ContactHandler contact;
    std::vector<std::array<int, 3>> surface;
    std::vector<int> tri_to_tet_map;
    const bool with_output = !label.empty() && interactions->output;
    if (interactions->contact->is_active() || with_output) find_surface(surface, tri_to_tet_map, V, T);
    if (interactions->contact->is_active()) contact = interactions->contact->add_triangles(ps, surface, tri_to_tet_map, p.contact);
    // the frame of a volume is its surface with the surface's own vertices (DeformablesPresets.cpp:74-76)
    if (with_output) interactions->output->add_triangle_mesh(label, ps, surface, tri_to_tet_map);
    return {ps, inertia, strain, contact};
}
Volume::VCH DeformablesPresets::add_volume_grid(const std::string& label, const Vec3& dim, const std::array<int, 3>& sub, const Volume::Params& p)
{
    std::vector<Vec3> V;
    std::vector<std::array<int, 4>> T;
    generate_tet_grid(V, T, {0.0, 0.0, 0.0}, dim, sub);
    auto h = add_volume(label, V, T, p);
    return {V, T, h};
}
RigidBody::Handler RigidBodyPresets::add(const std::string& label, double mass, const Mat3& inertia_local, const std::vector<Vec3>& V, const std::vector<std::array<int, 3>>& T,
                                         const EnergyFrictionalContact::Params& cp)
{
    // RigidBodyPresets.cpp:11-26
    RigidBodyHandler body = rigidbodies->add(mass, inertia_local);
    ContactHandler contact;
    if (interactions->contact->is_active()) contact = interactions->contact->add_triangles(body, V, T, cp);
    if (!label.empty() && interactions->output) interactions->output->add_triangle_mesh(label, body, V, T);
    return {body, contact};
}
RigidBody::VCH RigidBodyPresets::add_box(const std::string& label, double mass, const Vec3& size, const EnergyFrictionalContact::Params& cp)
{
    std::vector<Vec3> V;
    std::vector<std::array<int, 3>> T;
    make_box(V, T, size);
    auto h = add(label, mass, inertia_tensor_box(mass, size), V, T, cp);
    return {V, T, h};
}
Simulation::Simulation(const Settings& settings) : stark(settings)
{
    // Simulation.cpp:84-100: deformables, rigid bodies, interactions (= registration order of DoF sets and potentials)
    auto pd = std::make_shared<PointDynamics>(stark);
    auto rbd = std::make_shared<RigidBodyDynamics>(stark);
    deformables = std::make_shared<Deformables>(stark, pd);
    rigidbodies = std::make_shared<RigidBodies>(stark, rbd);
    interactions = std::make_shared<Interactions>();
    interactions->attachments = std::make_shared<EnergyAttachments>(stark, pd, rbd);  // Interactions.cpp:8-9: attachments, then contact
    interactions->contact = std::make_shared<EnergyFrictionalContact>(stark, pd, rbd);
    interactions->output = std::make_shared<MeshOutput>(stark, pd, rbd);
    presets = std::make_shared<Presets>();
    presets->deformables = std::make_shared<DeformablesPresets>(deformables, interactions);
    presets->rigidbodies = std::make_shared<RigidBodyPresets>(rigidbodies, interactions);
}

}  // namespace mistark
