// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY (never linked/imported by the product path).
//
// Drives the UNMODIFIED reference (built by oracle/Makefile into oracle/_ref/libstark_ref.a) through its public
// C++ API only, to
//   (1) `dump`  — write golden fixtures: for a named synthetic scene, the full evaluator input (every potential's
//                 connectivity table + every bound array, in the reference's binding order) and the reference's own
//                 stage outputs (E, grad, element Hessians, projected Hessians, assembled BSR triplets, PCG solution,
//                 Newton iterates / iteration counts).
//   (2) `time`  — run a scene for a number of steps and print the reference's own Newton / linear-solve timings as one
//                 JSON line (the "reference OpenMP CPU path" baseline of bench.py's cpu_baseline leg).
//
// Reference API used (all public):
//   stark::Simulation / presets / deformables / rigidbodies / interactions   stark/src/models/Simulation.h:13-42
//   stark::core::Stark {global_potential, context, callbacks, dt, gravity}   stark/src/core/Stark.h:12-46
//   symx::GlobalPotential::{get_potentials,get_dof_maps,get_dofs,set_dofs}    symx/src/solver/GlobalPotential.h:64-77
//   symx::Potential::{get_name,get_mws,has_conditional}                       symx/src/solver/Potential.h:13-35
//   symx::MappedWorkspace<double>::{maps,conn}                                symx/src/compile/MappedWorkspace.h:46-56
//   symx::SecondOrderCompiledPotential / Assembly / ElementHessians           symx/src/solver/second_order/*.h
//   bsm::BlockedSparseMatrix::{to_triplets,prepare_preconditioning}, bsm::solve_pcg   BlockedSparseMatrix/*.h
#include <stark>
#include "ipc_toolkit_geometry_functions.h"       // TriangleMeshCollisionDetection/src (narrow-phase classification)
#include "models/interactions/friction_geometry.h"  // stark/src
#include <Eigen/Sparse>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <functional>
#include <map>
#include <optional>
#include <sstream>
#include <string>
#include <vector>
#include <filesystem>
#include <omp.h>

namespace fs = std::filesystem;

// --------------------------------------------------------------------------------------------------------------------
// .npy writer (v1.0, little endian, C order)
// --------------------------------------------------------------------------------------------------------------------
static void write_npy(const std::string& path, const char* descr, size_t itemsize, const void* data, const std::vector<size_t>& shape)
{
    std::string shape_str = "(";
    size_t n = 1;
    for (size_t i = 0; i < shape.size(); i++) { shape_str += std::to_string(shape[i]) + ","; n *= shape[i]; }
    shape_str += ")";
    std::string header = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + shape_str + ", }";
    size_t total = 10 + header.size() + 1;
    size_t pad = (64 - total % 64) % 64;
    header += std::string(pad, ' ') + "\n";
    std::ofstream f(path, std::ios::binary);
    const char magic[] = "\x93NUMPY\x01\x00";
    f.write(magic, 8);
    uint16_t hl = (uint16_t)header.size();
    f.write((const char*)&hl, 2);
    f.write(header.data(), header.size());
    f.write((const char*)data, n * itemsize);
}
static void npy_f64(const std::string& p, const double* d, const std::vector<size_t>& s) { write_npy(p, "<f8", 8, d, s); }
static void npy_f32(const std::string& p, const float* d, const std::vector<size_t>& s) { write_npy(p, "<f4", 4, d, s); }
static void npy_i32(const std::string& p, const int32_t* d, const std::vector<size_t>& s) { write_npy(p, "<i4", 4, d, s); }

// --------------------------------------------------------------------------------------------------------------------
// Scenes (deterministic, no RNG). Parameters are passed as key=value on the command line.
// --------------------------------------------------------------------------------------------------------------------
struct Args
{
    std::map<std::string, std::string> kv;
    double d(const std::string& k, double def) const { auto it = kv.find(k); return it == kv.end() ? def : std::stod(it->second); }
    int i(const std::string& k, int def) const { auto it = kv.find(k); return it == kv.end() ? def : std::stoi(it->second); }
    std::string s(const std::string& k, const std::string& def) const { auto it = kv.find(k); return it == kv.end() ? def : it->second; }
};

// A collision mesh as handed to EnergyFrictionalContact::add_triangles (EnergyFrictionalContact.cpp:54-91), recorded so
// that fixtures carry the detector's inputs next to the reference's contact tables
struct ContactMeshRecord
{
    std::string kind;                          // "d" (deformable) or "rb"
    int idx_in_ps = 0;                         // point set index / rigid body index
    std::vector<int> verts;                    // per collision vertex: global index in the physical system's vertex array
    std::vector<std::array<int, 3>> triangles; // local connectivity
    std::vector<std::array<int, 2>> edges;     // find_edges_from_simplices(triangles, n_vertices)
    double thickness = 0.0;
};
struct Scene
{
    std::unique_ptr<stark::Simulation> sim;
    std::vector<ContactMeshRecord> contact_meshes;
    std::vector<std::array<double, 3>> friction_pairs;  // (mesh a, mesh b, mu)
    int n_rb_collision_vertices = 0;
    std::shared_ptr<void> keep;         // data a scene's user-defined potentials are bound to
    std::function<void()> before_step;  // the scene's script (Simulation::run(duration, callback) calls it before every time step)
    void step() { if (before_step) before_step(); sim->run_one_time_step(); }
    void record_deformable(const stark::PointSetHandler& ps, const std::vector<std::array<int, 3>>& tris, const std::vector<int>& map, double thickness)
    {
        ContactMeshRecord m;
        m.kind = "d";
        m.idx_in_ps = ps.get_idx();
        for (int l : map) m.verts.push_back(ps.get_global_index(l));
        m.triangles = tris;
        m.edges = stark::find_edges_from_simplices(tris, (int)map.size());
        m.thickness = thickness;
        contact_meshes.push_back(m);
    }
    // edge-only collision mesh (EnergyFrictionalContact::add_edges, EnergyFrictionalContact.cpp:66-77): the segments as given
    void record_deformable_edges(const stark::PointSetHandler& ps, const std::vector<std::array<int, 2>>& segments, double thickness)
    {
        ContactMeshRecord m;
        m.kind = "d";
        m.idx_in_ps = ps.get_idx();
        for (int l : ps.all()) m.verts.push_back(ps.get_global_index(l));
        m.edges = segments;
        m.thickness = thickness;
        contact_meshes.push_back(m);
    }
    void record_rigid(const stark::RigidBodyHandler& rb, int n_vertices, const std::vector<std::array<int, 3>>& tris, double thickness)
    {
        ContactMeshRecord m;
        m.kind = "rb";
        m.idx_in_ps = rb.get_idx();
        for (int l = 0; l < n_vertices; l++) m.verts.push_back(n_rb_collision_vertices + l);
        n_rb_collision_vertices += n_vertices;
        m.triangles = tris;
        m.edges = stark::find_edges_from_simplices(tris, n_vertices);
        m.thickness = thickness;
        contact_meshes.push_back(m);
    }
    void record_friction(int a, int b, double mu) { friction_pairs.push_back({ (double)a, (double)b, mu }); }
    std::string json;  // scene description echoed into the manifest so the build can construct the identical scene
};

static stark::Settings base_settings(const Args& a, const std::string& name)
{
    stark::Settings settings = stark::Settings();
    settings.output.simulation_name = name;
    settings.output.output_directory = a.s("outdir", "/tmp/mistark_oracle_out");
    settings.output.codegen_directory = a.s("codegen", "/tmp/mistark_oracle_codegen");
    settings.output.enable_frame_writes = a.i("frames", 0) != 0;  // mode `frames`: the reference writes its VTK frames (one per time step at fps = 30)
    if (a.i("frames", 0) != 0) settings.output.fps = 30;
    settings.output.enable_output = a.i("verbose", 0) != 0 || a.i("frames", 0) != 0;
    settings.output.console_verbosity = (symx::Verbosity)a.i("verbosity", (int)symx::Verbosity::Summary);  // 3 = Full: one line per Newton iteration
    settings.output.file_verbosity = symx::Verbosity::Minimal;
    settings.execution.n_threads = a.i("threads", 1);
    settings.simulation.max_time_step_size = a.d("dt", 1.0 / 30.0);
    settings.simulation.use_adaptive_time_step = a.i("adaptive", 1) != 0;
    if (a.s("solver", "pcg") == "llt") settings.newton.linear_solver = symx::LinearSolver::DirectLLT;  // NewtonsMethod.cpp:395-418
    const std::string proj = a.s("projection", "Progressive");
    if (proj == "Progressive") settings.newton.projection_mode = symx::ProjectionToPD::Progressive;
    else if (proj == "ProjectedNewton") settings.newton.projection_mode = symx::ProjectionToPD::ProjectedNewton;
    else if (proj == "Newton") settings.newton.projection_mode = symx::ProjectionToPD::Newton;
    else if (proj == "ProjectOnDemand") settings.newton.projection_mode = symx::ProjectionToPD::ProjectOnDemand;
    // the rest of the Newton driver's knobs (solver_utils.h:173-259), for the fixtures that pin its non-default branches
    settings.newton.project_to_pd_use_mirroring = a.i("mirroring", settings.newton.project_to_pd_use_mirroring ? 1 : 0) != 0;
    settings.newton.project_on_demand_countdown = a.i("countdown", settings.newton.project_on_demand_countdown);
    settings.newton.step_cap = a.d("step_cap", settings.newton.step_cap);
    settings.simulation.gravity[2] = a.d("gz", settings.simulation.gravity[2]);
    return settings;
}

// cfg-2 style tet beam: generate_tet_grid(center 0, dims {lx,ly,lz}, {nx,ny,nz}), Soft_Rubber, nodes with x < -lx/2+1e-3 prescribed
static Scene scene_tetbeam(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "tetbeam");
    settings.simulation.init_frictional_contact = false;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const int nx = a.i("nx", 8), ny = a.i("ny", 2), nz = a.i("nz", 2);
    const double lx = a.d("lx", 4.0), ly = a.d("ly", 1.0), lz = a.d("lz", 1.0);
    auto material = stark::Volume::Params::Soft_Rubber();
    material.strain.elasticity_only = a.i("eo", 1) != 0;
    material.strain.damping = a.d("strain_damping", material.strain.damping);
    material.strain.strain_limit = a.d("strain_limit", material.strain.strain_limit);
    material.strain.strain_limit_stiffness = a.d("strain_limit_stiffness", material.strain.strain_limit_stiffness);
    material.strain.youngs_modulus = a.d("E", material.strain.youngs_modulus);
    material.strain.poissons_ratio = a.d("nu", material.strain.poissons_ratio);
    auto [V, T, H] = sim.presets->deformables->add_volume_grid("beam", { lx, ly, lz }, { nx, ny, nz }, material);
    auto bc = stark::EnergyPrescribedPositions::Params().set_stiffness(a.d("bc_stiffness", 1e7));
    sim.deformables->prescribed_positions->add_inside_aabb(H.point_set, { -0.5 * lx, 0.0, 0.0 }, { 2e-3, 2.0 * ly, 2.0 * lz }, bc);
    std::ostringstream js;
    js << "{\"kind\":\"tetbeam\",\"nx\":" << nx << ",\"ny\":" << ny << ",\"nz\":" << nz << ",\"lx\":" << lx << ",\"ly\":" << ly << ",\"lz\":" << lz
       << ",\"eo\":" << (material.strain.elasticity_only ? 1 : 0) << "}";
    sc.json = js.str();
    return sc;
}

// Hanging cloth (examples/main.cpp hanging_cloth): n x n Cotton_Fabric, two corners prescribed, no contact
static Scene scene_cloth(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "cloth");
    settings.simulation.init_frictional_contact = false;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const int n = a.i("n", 8);
    const double d = a.d("size", 1.0);
    const double hd = 0.5 * d;
    auto material = stark::Surface::Params::Cotton_Fabric();
    material.strain.elasticity_only = a.i("eo", 0) != 0;
    material.bending.flat_rest_angle = a.i("flat", 1) != 0;
    material.bending.stiffness = a.d("bend_stiffness", material.bending.stiffness);
    material.bending.damping = a.d("bend_damping", material.bending.damping);
    material.strain.inflation = a.d("inflation", 0.0);
    auto [V, T, H] = sim.presets->deformables->add_surface_grid("cloth", { d, d }, { n, n }, material);
    auto bc = stark::EnergyPrescribedPositions::Params().set_stiffness(1e6);
    sim.deformables->prescribed_positions->add_inside_aabb(H.point_set, { hd, hd, 0.0 }, { 0.001, 0.001, 0.001 }, bc);
    sim.deformables->prescribed_positions->add_inside_aabb(H.point_set, { -hd, hd, 0.0 }, { 0.001, 0.001, 0.001 }, bc);
    std::ostringstream js;
    js << "{\"kind\":\"cloth\",\"n\":" << n << ",\"size\":" << d << ",\"eo\":" << (material.strain.elasticity_only ? 1 : 0)
       << ",\"flat\":" << (material.bending.flat_rest_angle ? 1 : 0) << "}";
    sc.json = js.str();
    return sc;
}

// examples/main.cpp hanging_deformable_box (:76-107): n^3 Soft_Rubber box (E = 1e4, damping + strain limiting on), two top corners prescribed
static Scene scene_hangingbox(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "hangingbox");
    settings.simulation.init_frictional_contact = false;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const int n = a.i("n", 10);
    const double d = a.d("size", 0.5), hd = 0.5 * d;
    auto material = stark::Volume::Params::Soft_Rubber();
    material.strain.youngs_modulus = 1e4;
    material.strain.elasticity_only = false;
    auto [V, T, H] = sim.presets->deformables->add_volume_grid("box", { d, d, d }, { n, n, n }, material);
    auto bc = stark::EnergyPrescribedPositions::Params().set_stiffness(1e7);
    sim.deformables->prescribed_positions->add_inside_aabb(H.point_set, { hd, hd, hd }, { 0.001, 0.001, 0.001 }, bc);
    sim.deformables->prescribed_positions->add_inside_aabb(H.point_set, { -hd, hd, hd }, { 0.001, 0.001, 0.001 }, bc);
    std::ostringstream js;
    js << "{\"kind\":\"hangingbox\",\"n\":" << n << ",\"size\":" << d << "}";
    sc.json = js.str();
    return sc;
}

// pystark/pystark/test_sim.py:4-31 (the reference's Python smoke test): s x s Cotton_Fabric cloth, n x n, two corners prescribed with the
// default EnergyPrescribedPositions::Params, no contact, run for 1 s
static Scene scene_pycloth(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "pycloth");
    settings.simulation.init_frictional_contact = false;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const int n = a.i("n", 32);
    const double s = a.d("size", 0.5);
    auto [V, T, H] = sim.presets->deformables->add_surface_grid("cloth", { s, s }, { n, n }, stark::Surface::Params::Cotton_Fabric());
    const Eigen::Vector3d dim = 0.001 * Eigen::Vector3d::Ones();
    sim.deformables->prescribed_positions->add_inside_aabb(H.point_set, { 0.5 * s, 0.5 * s, 0.0 }, dim, stark::EnergyPrescribedPositions::Params());
    sim.deformables->prescribed_positions->add_inside_aabb(H.point_set, { 0.5 * s, -0.5 * s, 0.0 }, dim, stark::EnergyPrescribedPositions::Params());
    std::ostringstream js;
    js << "{\"kind\":\"pycloth\",\"n\":" << n << ",\"size\":" << s << "}";
    sc.json = js.str();
    return sc;
}

// cfg-4 style block WITHOUT contact: generate_tet_grid(center (0,0,0.6), {lx,ly,lz}, {nx,ny,nz}), Soft_Rubber (full potential),
// bottom face (z = 0.6 - lz/2) prescribed, gravity compresses the block
static Scene scene_tetblock(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "tetblock");
    settings.simulation.init_frictional_contact = false;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const int nx = a.i("nx", 4), ny = a.i("ny", 4), nz = a.i("nz", 4);
    const double lx = a.d("lx", 1.0), ly = a.d("ly", 1.0), lz = a.d("lz", 1.0);
    auto material = stark::Volume::Params::Soft_Rubber();
    material.strain.elasticity_only = a.i("eo", 0) != 0;
    auto [V, T] = stark::generate_tet_grid({ 0.0, 0.0, 0.6 }, { lx, ly, lz }, { nx, ny, nz });
    auto H = sim.presets->deformables->add_volume("block", V, T, material);
    auto bc = stark::EnergyPrescribedPositions::Params().set_stiffness(a.d("bc_stiffness", 1e7));
    sim.deformables->prescribed_positions->add_inside_aabb(H.point_set, { 0.0, 0.0, 0.6 - 0.5 * lz }, { 2.0 * lx, 2.0 * ly, 2e-3 }, bc);
    std::ostringstream js;
    js << "{\"kind\":\"tetblock\",\"nx\":" << nx << ",\"ny\":" << ny << ",\"nz\":" << nz << ",\"lx\":" << lx << ",\"ly\":" << ly << ",\"lz\":" << lz
       << ",\"eo\":" << (material.strain.elasticity_only ? 1 : 0) << "}";
    sc.json = js.str();
    return sc;
}

// Chain of rigid boxes exercising every rigid-body potential: box0 fixed (global point + 2 global directions), then one
// constraint type per link (tests/rb_constraints.cpp and examples/rb_constraint_test_scenes.cpp use the same building blocks)
static Scene scene_rbchain(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "rbchain");
    settings.simulation.init_frictional_contact = false;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    auto make_box = [&](double x, double y, double z) {
        auto [V, T, box] = sim.presets->rigidbodies->add_box("box", 1.0 + 0.1 * x, { 0.1, 0.12, 0.08 });
        box.rigidbody.set_translation({ x, y, z });
        box.rigidbody.add_rotation(20.0 * x + 5.0, Eigen::Vector3d(1.0, 0.5, -0.3).normalized());
        return box.rigidbody;
    };
    auto rbs = sim.rigidbodies;
    rbs->set_default_constraint_stiffness(a.d("rb_stiffness", 1e4));
    auto b0 = make_box(0.0, 0.0, 0.0);
    rbs->add_constraint_fix(b0);
    auto b1 = make_box(0.15, 0.0, 0.0);
    rbs->add_constraint_point(b0, b1, { 0.07, 0.02, 0.01 });
    auto b2 = make_box(0.30, 0.02, 0.0);
    rbs->add_constraint_point_on_axis(b1, b2, { 0.22, 0.0, 0.01 }, Eigen::Vector3d(1.0, 0.2, 0.0).normalized());
    auto b3 = make_box(0.45, 0.0, 0.03);
    rbs->add_constraint_distance(b2, b3, { 0.33, 0.0, 0.0 }, { 0.42, 0.01, 0.02 });
    auto b4 = make_box(0.60, 0.0, 0.0);
    rbs->add_constraint_distance_limits(b3, b4, { 0.48, 0.0, 0.0 }, { 0.57, 0.0, 0.0 }, 0.08995, 0.09005);
    rbs->add_constraint_direction(b3, b4, Eigen::Vector3d(0.0, 1.0, 0.2).normalized());
    auto b5 = make_box(0.75, 0.0, 0.0);
    rbs->add_constraint_point(b4, b5, { 0.67, 0.0, 0.0 });
    rbs->add_constraint_angle_limit(b4, b5, Eigen::Vector3d(1.0, 0.0, 0.0), 0.5);
    auto b6 = make_box(0.90, 0.0, 0.0);
    rbs->add_constraint_spring(b5, b6, { 0.78, 0.0, 0.0 }, { 0.87, 0.01, 0.0 }, 200.0, 3.0);
    auto b7 = make_box(1.05, 0.0, 0.0);
    rbs->add_constraint_point_on_axis(b6, b7, { 0.97, 0.0, 0.0 }, Eigen::Vector3d(1.0, 0.0, 0.0));
    rbs->add_constraint_linear_velocity(b6, b7, Eigen::Vector3d(1.0, 0.0, 0.0), 0.3, 5.0, 0.05);
    auto b8 = make_box(1.20, 0.0, 0.0);
    rbs->add_constraint_hinge(b7, b8, { 1.12, 0.0, 0.0 }, Eigen::Vector3d(0.0, 1.0, 0.0));
    rbs->add_constraint_angular_velocity(b7, b8, Eigen::Vector3d(0.0, 1.0, 0.0), 1.0, 2.0, 0.1);
    sc.json = "{\"kind\":\"rbchain\"}";
    return sc;
}

// Contact zoo: a fixed rigid box, a cloth hovering over it inside the contact distance, a second rigid box and a soft
// tet block over the cloth, a third rigid box beside the first. IPC contact + friction between every pair.
static Scene scene_contactmix(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "contactmix");
    settings.simulation.init_frictional_contact = true;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const double th = a.d("thickness", 0.01);
    const double gap = a.d("gap", 0.012);
    auto gp = stark::EnergyFrictionalContact::GlobalParams();
    gp.default_contact_thickness = th;
    gp.min_contact_stiffness = a.d("kmin", 1e5);
    gp.friction_stick_slide_threshold = a.d("epsv", 0.1);
    sim.interactions->contact->set_global_params(gp);
    const int n = a.i("n", 6);

    auto [bV, bT, boxA] = sim.presets->rigidbodies->add_box("boxA", 1.0, 0.3);
    sim.rigidbodies->add_constraint_fix(boxA.rigidbody);
    sc.record_rigid(boxA.rigidbody, (int)bV.size(), bT, th);

    auto cm = stark::Surface::Params::Cotton_Fabric();
    auto [cV, cT, cloth] = sim.presets->deformables->add_surface_grid("cloth", { 0.5, 0.5 }, { n, n }, cm);
    sc.record_deformable(cloth.point_set, cT, cloth.point_set.all(), th);
    cloth.point_set.add_rotation(10.0, Eigen::Vector3d::UnitZ());
    cloth.point_set.add_displacement({ 0.01, 0.005, 0.15 + gap });

    auto [b2V, b2T, boxB] = sim.presets->rigidbodies->add_box("boxB", 0.2, 0.1);
    sc.record_rigid(boxB.rigidbody, (int)b2V.size(), b2T, th);
    boxB.rigidbody.add_rotation(25.0, Eigen::Vector3d::UnitZ());
    boxB.rigidbody.add_translation({ -0.08, 0.06, 0.15 + 2.0 * gap + 0.05 });

    auto [b3V, b3T, boxC] = sim.presets->rigidbodies->add_box("boxC", 0.5, 0.2);
    sc.record_rigid(boxC.rigidbody, (int)b3V.size(), b3T, th);
    boxC.rigidbody.add_rotation(3.0, Eigen::Vector3d(0.2, 0.3, 1.0).normalized());
    boxC.rigidbody.add_translation({ 0.15 + 0.1 + gap + 0.004, 0.01, 0.03 });

    auto vm = stark::Volume::Params::Soft_Rubber();
    auto [sV, sT] = stark::generate_tet_grid({ 0.12, -0.1, 0.15 + 2.0 * gap + 0.04 }, { 0.08, 0.08, 0.08 }, { 2, 2, 2 });
    auto soft = sim.presets->deformables->add_volume("soft", sV, sT, vm);
    {
        // the preset extracts the collision surface with find_surface (DeformablesPresets.cpp:66-72); same call, same result
        auto [surf, map] = stark::find_surface(sV, sT);
        sc.record_deformable(soft.point_set, surf, map, th);
    }

    const double mu = a.d("mu", 0.5);
    auto ct = sim.interactions->contact;
    ct->set_friction(boxA.contact, cloth.contact, mu);
    ct->set_friction(boxB.contact, cloth.contact, mu);
    ct->set_friction(boxA.contact, boxC.contact, mu);
    ct->set_friction(soft.contact, cloth.contact, mu);
    ct->set_friction(cloth.contact, cloth.contact, mu);
    sc.record_friction(0, 1, mu);
    sc.record_friction(2, 1, mu);
    sc.record_friction(0, 3, mu);
    sc.record_friction(4, 1, mu);
    sc.record_friction(1, 1, mu);
    sc.json = "{\"kind\":\"contactmix\"}";
    return sc;
}

// configs[4] of BASELINE.json ("mixed scene: soft body + shell + hinged rigid bodies, full coupling"): a Soft_Rubber tet block resting
// over a fixed floor box, a Cotton_Fabric cloth over the block, a chain of boxes joined by add_constraint_hinge (first link fixed)
// over the cloth. Rigid bodies are registered first (see scene_blockbox about the registration order and friction).
static Scene scene_mixed(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "mixed");
    settings.simulation.init_frictional_contact = true;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const int nx = a.i("nx", 26), ny = a.i("ny", 26), nz = a.i("nz", 25), nc = a.i("nc", 128), nrb = a.i("nrb", 16);
    const double L = a.d("L", 1.0), gap = a.d("gap", 0.0015), th = a.d("thickness", 1e-3), mu = a.d("mu", 0.5), bx = a.d("bx", 3.0), bz = a.d("bz", 0.1);
    const double link = a.d("link", 0.05), cloth_size = a.d("cloth", 1.2);
    auto gp = stark::EnergyFrictionalContact::GlobalParams();
    gp.default_contact_thickness = th;
    gp.min_contact_stiffness = a.d("kmin", 1e8);
    sim.interactions->contact->set_global_params(gp);
    auto ct = sim.interactions->contact;

    auto [fV, fT, floor] = sim.presets->rigidbodies->add_box("floor", 1.0, { bx, bx, bz });
    sc.record_rigid(floor.rigidbody, (int)fV.size(), fT, th);
    sim.rigidbodies->add_constraint_fix(floor.rigidbody);

    const double z_block = 0.5 * bz + gap + 0.5 * L;
    const double z_cloth = 0.5 * bz + gap + L + gap;
    const double z_chain = z_cloth + gap + 0.5 * link;
    const double pitch = 1.5 * link;  // link length = link, spacing between links = link / 2
    std::vector<stark::RigidBodyHandler> links;
    std::vector<stark::EnergyFrictionalContact::Handler> link_contacts;
    for (int i = 0; i < nrb; i++) {
        auto [V, T, h] = sim.presets->rigidbodies->add_box("link", 0.2, { link, link, link });
        h.rigidbody.set_translation({ (i - 0.5 * (nrb - 1)) * pitch, 0.0, z_chain });
        sc.record_rigid(h.rigidbody, (int)V.size(), T, th);
        links.push_back(h.rigidbody);
        link_contacts.push_back(h.contact);
    }
    sim.rigidbodies->add_constraint_fix(links[0]);
    for (int i = 0; i + 1 < nrb; i++) {
        const Eigen::Vector3d mid((i + 0.5 - 0.5 * (nrb - 1)) * pitch, 0.0, z_chain);
        sim.rigidbodies->add_constraint_hinge(links[i], links[i + 1], mid, Eigen::Vector3d::UnitY());
        ct->disable_collision(link_contacts[i], link_contacts[i + 1]);
    }

    auto vm = stark::Volume::Params::Soft_Rubber();
    auto [sV, sT] = stark::generate_tet_grid({ 0.0, 0.0, z_block }, { L, L, L }, { nx, ny, nz });
    auto soft = sim.presets->deformables->add_volume("block", sV, sT, vm);
    {
        auto [surf, map] = stark::find_surface(sV, sT);
        sc.record_deformable(soft.point_set, surf, map, th);
    }
    auto cm = stark::Surface::Params::Cotton_Fabric();
    auto [cV, cT, cloth] = sim.presets->deformables->add_surface_grid("cloth", { cloth_size * L, cloth_size * L }, { nc, nc }, cm);
    sc.record_deformable(cloth.point_set, cT, cloth.point_set.all(), th);
    cloth.point_set.add_displacement({ 0.0, 0.0, z_cloth });

    if (mu > 0.0) {
        ct->set_friction(floor.contact, soft.contact, mu);
        ct->set_friction(soft.contact, cloth.contact, mu);
        for (int i = 0; i < nrb; i++) ct->set_friction(link_contacts[i], cloth.contact, mu);
        sc.record_friction(0, nrb + 1, mu);
        sc.record_friction(nrb + 1, nrb + 2, mu);
        for (int i = 0; i < nrb; i++) sc.record_friction(1 + i, nrb + 2, mu);
    }
    std::ostringstream js;
    js << "{\"kind\":\"mixed\",\"nx\":" << nx << ",\"ny\":" << ny << ",\"nz\":" << nz << ",\"nc\":" << nc << ",\"nrb\":" << nrb << ",\"L\":" << L << ",\"gap\":" << gap
       << ",\"thickness\":" << th << ",\"mu\":" << mu << ",\"bx\":" << bx << ",\"bz\":" << bz << ",\"link\":" << link << ",\"cloth\":" << cloth_size
       << ",\"kmin\":" << gp.min_contact_stiffness << "}";
    sc.json = js.str();
    return sc;
}

// Vertex/edge contacts: box corners aimed at a corner and at an edge of a fixed box, a soft block corner aimed at another
// corner -> the point-point and point-edge rows of the rb-rb and rb-deformable tables that flat contacts never produce
static Scene scene_contactcorners(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "contactcorners");
    settings.simulation.init_frictional_contact = true;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const double th = a.d("thickness", 0.01);
    const double g = a.d("gap", 0.012);
    auto gp = stark::EnergyFrictionalContact::GlobalParams();
    gp.default_contact_thickness = th;
    gp.min_contact_stiffness = a.d("kmin", 1e5);
    sim.interactions->contact->set_global_params(gp);
    const double h = 0.15, s3 = g / std::sqrt(3.0), s2 = g / std::sqrt(2.0);

    auto [aV, aT, boxA] = sim.presets->rigidbodies->add_box("boxA", 1.0, 2.0 * h);
    sim.rigidbodies->add_constraint_fix(boxA.rigidbody);
    sc.record_rigid(boxA.rigidbody, (int)aV.size(), aT, th);

    // corner of D on the diagonal of A's (+,+,+) corner
    auto [dV, dT, boxD] = sim.presets->rigidbodies->add_box("boxD", 0.2, 0.1);
    sc.record_rigid(boxD.rigidbody, (int)dV.size(), dT, th);
    boxD.rigidbody.add_translation({ h + s3 + 0.05, h + s3 + 0.05, h + s3 + 0.05 });

    // corner of E over the middle of A's edge (x = +h, z = +h)
    auto [eV, eT, boxE] = sim.presets->rigidbodies->add_box("boxE", 0.2, 0.1);
    sc.record_rigid(boxE.rigidbody, (int)eV.size(), eT, th);
    boxE.rigidbody.add_translation({ h + s2 + 0.05, 0.013 + 0.05, h + s2 + 0.05 });

    // corner of a soft block on the diagonal of A's (-,-,+) corner
    auto vm = stark::Volume::Params::Soft_Rubber();
    auto [sV, sT] = stark::generate_tet_grid({ -h - s3 - 0.04, -h - s3 - 0.04, h + s3 + 0.04 }, { 0.08, 0.08, 0.08 }, { 1, 1, 1 });
    auto soft = sim.presets->deformables->add_volume("soft", sV, sT, vm);
    {
        auto [surf, map] = stark::find_surface(sV, sT);
        sc.record_deformable(soft.point_set, surf, map, th);
    }
    const double mu = a.d("mu", 0.5);
    auto ct = sim.interactions->contact;
    ct->set_friction(boxA.contact, boxD.contact, mu);
    ct->set_friction(boxA.contact, boxE.contact, mu);
    ct->set_friction(boxA.contact, soft.contact, mu);
    sc.record_friction(0, 1, mu);
    sc.record_friction(0, 2, mu);
    sc.record_friction(0, 3, mu);
    sc.json = "{\"kind\":\"contactcorners\"}";
    return sc;
}

// Rods in contact (edge-only collision meshes, Line presets): a rod lying diagonally over a fixed rigid box, a second rod crossing
// over the first, and a cloth patch over both
static Scene scene_contactrods(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "contactrods");
    settings.simulation.init_frictional_contact = true;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const double th = a.d("thickness", 0.01), g = a.d("gap", 0.012), h = 0.15;
    auto gp = stark::EnergyFrictionalContact::GlobalParams();
    gp.default_contact_thickness = th;
    gp.min_contact_stiffness = a.d("kmin", 1e5);
    sim.interactions->contact->set_global_params(gp);

    auto [aV, aT, boxA] = sim.presets->rigidbodies->add_box("boxA", 1.0, 2.0 * h);
    sim.rigidbodies->add_constraint_fix(boxA.rigidbody);
    sc.record_rigid(boxA.rigidbody, (int)aV.size(), aT, th);

    auto lp = stark::Line::Params::Elastic_Rubberband();
    auto [r1V, r1S, rod1] = sim.presets->deformables->add_line_as_segments("rod1", { -0.2, -0.12, h + g }, { 0.2, 0.1, h + g }, 7, lp);
    sc.record_deformable_edges(rod1.point_set, r1S, th);
    auto [r2V, r2S, rod2] = sim.presets->deformables->add_line_as_segments("rod2", { -0.1, 0.17, h + 2.0 * g }, { 0.13, -0.18, h + 2.0 * g + 0.004 }, 5, lp);
    sc.record_deformable_edges(rod2.point_set, r2S, th);

    auto [cV, cT, cloth] = sim.presets->deformables->add_surface_grid("cloth", { 0.2, 0.2 }, { 4, 4 }, stark::Surface::Params::Cotton_Fabric());
    sc.record_deformable(cloth.point_set, cT, cloth.point_set.all(), th);
    cloth.point_set.add_rotation(7.0, Eigen::Vector3d::UnitZ());
    cloth.point_set.add_displacement({ 0.01, 0.0, h + 3.0 * g + 0.002 });

    const double mu = a.d("mu", 0.5);
    auto ct = sim.interactions->contact;
    ct->set_friction(boxA.contact, rod1.contact, mu);
    ct->set_friction(rod1.contact, rod2.contact, mu);
    ct->set_friction(rod2.contact, cloth.contact, mu);
    ct->set_friction(rod1.contact, cloth.contact, mu);
    sc.record_friction(0, 1, mu);
    sc.record_friction(1, 2, mu);
    sc.record_friction(2, 3, mu);
    sc.record_friction(1, 3, mu);
    sc.json = "{\"kind\":\"contactrods\"}";
    return sc;
}

// Hello-world of the reference's README (cfg 1) without the spin script: Cotton_Fabric cloth over a fixed rigid box
static Scene scene_clothbox(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "clothbox");
    settings.simulation.init_frictional_contact = true;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const int n = a.i("n", 32);
    const double th = a.d("thickness", 0.0025), gap = a.d("gap", 0.04), mu = a.d("mu", 0.5), size = a.d("size", 0.4), bs = a.d("box", 0.08);
    auto gp = stark::EnergyFrictionalContact::GlobalParams();
    gp.default_contact_thickness = th;
    gp.min_contact_stiffness = a.d("kmin", gp.min_contact_stiffness);
    sim.interactions->contact->set_global_params(gp);
    auto [cV, cT, cloth] = sim.presets->deformables->add_surface_grid("cloth", { size, size }, { n, n }, stark::Surface::Params::Cotton_Fabric());
    // tilt (degrees about y through the cloth's centre; round 6): the cloth's lowest edge stays `gap` above the box's top face, the rest rises to
    // gap + size sin(tilt). A flat cloth dropped parallel to the floor can pass through it between two steps unnoticed (no CCD in this contact
    // model; the intersection check only sees edges that cross the surface); a tilted one straddles the surface whenever it penetrates.
    const double tilt = a.d("tilt", 0.0);
    if (tilt != 0.0) {
        cloth.point_set.add_rotation(tilt, Eigen::Vector3d::UnitY());
        cloth.point_set.add_displacement({ 0.0, 0.0, 0.5 * size * std::sin(std::abs(tilt) * M_PI / 180.0) });
    }
    sc.record_deformable(cloth.point_set, cT, cloth.point_set.all(), th);
    auto [bV, bT, box] = sim.presets->rigidbodies->add_box("box", 1.0, bs);
    sc.record_rigid(box.rigidbody, (int)bV.size(), bT, th);
    const double ox = a.d("ox", 0.0), oy = a.d("oy", 0.0);  // (the box moved off the cloth's axes: see scene_blockbox)
    box.rigidbody.add_translation({ ox, oy, -0.5 * bs - gap });
    auto fix = sim.rigidbodies->add_constraint_fix(box.rigidbody);
    const double spin = a.d("spin", 0.0);  // README.md:84-91: the script turns the fixed box by 90 deg/s about z
    if (spin != 0.0) {
        stark::Simulation* ps = sc.sim.get();
        const Eigen::Vector3d anchor(0.0, 0.0, -0.5 * bs - gap);
        sc.before_step = [fix, ps, anchor, spin]() mutable { fix.set_transformation(anchor, spin * ps->get_time(), Eigen::Vector3d::UnitZ()); };
    }
    if (mu > 0.0) {
        sim.interactions->contact->set_friction(cloth.contact, box.contact, mu);
        sc.record_friction(0, 1, mu);
    }
    std::ostringstream js;
    js << "{\"kind\":\"clothbox\",\"spin\":" << spin << ",\"n\":" << n << ",\"thickness\":" << th << ",\"gap\":" << gap << ",\"mu\":" << mu << ",\"size\":" << size << ",\"box\":" << bs
       << ",\"kmin\":" << gp.min_contact_stiffness << ",\"ox\":" << ox << ",\"oy\":" << oy << ",\"tilt\":" << tilt << "}";
    sc.json = js.str();
    return sc;
}

// cfg 4: Soft_Rubber tet block dropped on a fixed rigid box, IPC contact + friction
static Scene scene_blockbox(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "blockbox");
    settings.simulation.init_frictional_contact = true;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const int nx = a.i("nx", 44), ny = a.i("ny", 44), nz = a.i("nz", 43);
    const double L = a.d("L", 1.0), gap = a.d("gap", 0.05), th = a.d("thickness", 1e-3), mu = a.d("mu", 0.5), bx = a.d("bx", 3.0), bz = a.d("bz", 0.1);
    auto gp = stark::EnergyFrictionalContact::GlobalParams();
    gp.default_contact_thickness = th;
    gp.min_contact_stiffness = a.d("kmin", 1e8);
    sim.interactions->contact->set_global_params(gp);
    // Registration order matters for friction: with the deformable registered FIRST, edge-edge friction rows take the
    // reference's "deformable -> rigid" branch (EnergyFrictionalContact.cpp:763-766), whose barycentric pair is stored in (A, B) =
    // (deformable, rigid) order but consumed as (rigid, deformable) (:1200-1207 with :1348-1356); the result then depends on the
    // orientation of the collision edges, i.e. on find_surface's unordered_map iteration order. boxfirst=1 avoids that branch.
    const bool boxfirst = a.i("boxfirst", 1) != 0;
    auto vm = stark::Volume::Params::Soft_Rubber();
    // ox / oy: the block moved off the box's axes. Centred, the nodes of the bottom face with x = -y lie EXACTLY above the diagonal edge of the
    // box's top face, and ~80 edge-edge pairs have their closest point exactly at an edge endpoint: whether such a pair counts as edge-edge
    // or edge-point is decided by the last bit of a product (i.e. by how the compiler contracted the multiply-adds), the energy and gradient
    // are the same either way, the Hessian is not.
    const double ox = a.d("ox", 0.0), oy = a.d("oy", 0.0);
    auto [sV, sT] = stark::generate_tet_grid({ ox, oy, 0.5 * bz + gap + 0.5 * L }, { L, L, L }, { nx, ny, nz });
    std::optional<stark::Volume::Handler> soft_;
    std::optional<stark::RigidBody::Handler> box_;
    auto add_block = [&]() {
        soft_.emplace(sim.presets->deformables->add_volume("block", sV, sT, vm));
        auto [surf, map] = stark::find_surface(sV, sT);
        sc.record_deformable(soft_->point_set, surf, map, th);
    };
    auto add_the_box = [&]() {
        auto [bV, bT, bh] = sim.presets->rigidbodies->add_box("box", 1.0, { bx, bx, bz });
        box_.emplace(bh);
        sc.record_rigid(box_->rigidbody, (int)bV.size(), bT, th);
        sim.rigidbodies->add_constraint_fix(box_->rigidbody);
    };
    if (boxfirst) { add_the_box(); add_block(); }
    else { add_block(); add_the_box(); }
    if (mu > 0.0) {
        sim.interactions->contact->set_friction(soft_->contact, box_->contact, mu);
        sc.record_friction(0, 1, mu);
    }
    std::ostringstream js;
    js << "{\"kind\":\"blockbox\",\"nx\":" << nx << ",\"ny\":" << ny << ",\"nz\":" << nz << ",\"L\":" << L << ",\"gap\":" << gap << ",\"thickness\":" << th
       << ",\"mu\":" << mu << ",\"bx\":" << bx << ",\"bz\":" << bz << ",\"kmin\":" << gp.min_contact_stiffness << ",\"boxfirst\":" << (boxfirst ? 1 : 0) << ",\"ox\":" << ox << ",\"oy\":" << oy << "}";
    sc.json = js.str();
    return sc;
}

// examples/main.cpp hanging_net: an n x n net of Elastic_Rubberband segments (the edges of a triangle grid) fixed along its perimeter
static Scene scene_hangingnet(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "hangingnet");
    settings.simulation.init_frictional_contact = false;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const int n = a.i("n", 20);
    const double d = a.d("size", 1.0);
    auto [V, T] = stark::generate_triangle_grid({ 0.0, 0.0 }, { d, d }, { n, n });
    auto E = stark::find_edges_from_simplices(T, (int)V.size());
    auto H = sim.presets->deformables->add_line("segments", V, E, stark::Line::Params::Elastic_Rubberband());
    sim.deformables->prescribed_positions->add_outside_aabb(H.point_set, { 0.0, 0.0, 0.0 }, { d - 0.001, d - 0.001, d - 0.001 }, stark::EnergyPrescribedPositions::Params());
    std::ostringstream js;
    js << "{\"kind\":\"hangingnet\",\"n\":" << n << ",\"size\":" << d << "}";
    sc.json = js.str();
    return sc;
}

// Rods + attachments: a cloth hanging from two rods (complete / elasticity-only segment strain) by a
// point-point and a point-edge attachment, a free rod riding on the cloth (point-triangle and edge-edge attachments) and a free
// rigid box hanging from the cloth's far edge (rigid-deformable attachments). No contact.
static Scene scene_attachzoo(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "attachzoo");
    settings.simulation.init_frictional_contact = false;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const int n = a.i("n", 6);
    const double d = a.d("size", 0.4), hd = 0.5 * d, h = d / n;
    const double k = a.d("k", 1e4), tol = a.d("tol", std::numeric_limits<double>::max());
    auto [cV, cT, cloth] = sim.presets->deformables->add_surface_grid("cloth", { d, d }, { n, n }, stark::Surface::Params::Cotton_Fabric());
    auto find = [&](double x, double y) {
        int best = 0;
        for (int i = 0; i < (int)cV.size(); i++)
            if ((cV[i] - Eigen::Vector3d(x, y, 0.0)).norm() < (cV[best] - Eigen::Vector3d(x, y, 0.0)).norm()) best = i;
        return best;
    };
    auto bc = stark::EnergyPrescribedPositions::Params().set_stiffness(1e6);
    auto att = stark::EnergyAttachments::Params().set_stiffness(k).set_tolerance(tol);

    // rod A: complete model, hangs the corner (-hd, -hd) by a point-point attachment
    auto lpA = stark::Line::Params::Elastic_Rubberband();
    lpA.strain.strain_limit = a.d("rod_strain_limit", 0.01);
    lpA.strain.damping = a.d("rod_damping", 1e-3);
    auto [aV, aS, rodA] = sim.presets->deformables->add_line_as_segments("rodA", { -hd, -hd, 0.3 }, { -hd, -hd, 0.0 }, 5, lpA);
    sim.deformables->prescribed_positions->add(rodA.point_set, { 0 }, bc);
    sim.interactions->attachments->add(rodA.point_set, cloth.point_set, std::vector<int>{ 5 }, std::vector<int>{ find(-hd, -hd) }, att);

    // rod B: elasticity only, ends on the boundary edge next to the corner (hd, -hd): point-edge attachment
    auto lpB = stark::Line::Params::Elastic_Rubberband();
    lpB.strain.elasticity_only = true;
    const int e0 = find(hd, -hd), e1 = find(hd - h, -hd);
    const Eigen::Vector3d endB = 0.3 * cV[e0] + 0.7 * cV[e1];
    auto [bV, bS, rodB] = sim.presets->deformables->add_line_as_segments("rodB", endB + Eigen::Vector3d(0.0, 0.0, 0.3), endB, 4, lpB);
    sim.deformables->prescribed_positions->add(rodB.point_set, { 0 }, bc);
    sim.interactions->attachments->add(rodB.point_set, cloth.point_set, std::vector<int>{ 4 }, std::vector<std::array<int, 2>>{ { e0, e1 } }, std::vector<std::array<double, 2>>{ { 0.3, 0.7 } }, att);

    // rod C: free, lies on the cloth; one end sits in a triangle, one segment is tied to a cloth edge
    const std::array<int, 3> tri = cT[cT.size() / 2];
    const Eigen::Vector3d pC = 0.2 * cV[tri[0]] + 0.3 * cV[tri[1]] + 0.5 * cV[tri[2]];
    auto lpC = stark::Line::Params::Elastic_Rubberband();
    auto [rV, rS, rodC] = sim.presets->deformables->add_line_as_segments("rodC", pC, pC + Eigen::Vector3d(3.0 * h, 0.7 * h, 0.0), 3, lpC);
    sim.interactions->attachments->add(rodC.point_set, cloth.point_set, std::vector<int>{ 0 }, std::vector<std::array<int, 3>>{ tri }, std::vector<std::array<double, 3>>{ { 0.2, 0.3, 0.5 } }, att);
    {
        const Eigen::Vector3d mid = 0.5 * (rV[2] + rV[3]);
        const int q0 = find(mid.x(), mid.y());
        const int q1 = find(cV[q0].x() + h, cV[q0].y());
        sim.interactions->attachments->add(rodC.point_set, cloth.point_set, std::vector<std::array<int, 2>>{ { 2, 3 } }, std::vector<std::array<int, 2>>{ { q0, q1 } }, std::vector<std::array<double, 2>>{ { 0.5, 0.5 } },
                                           std::vector<std::array<double, 2>>{ { 0.4, 0.6 } }, att);
    }

    // free rigid box hanging from the cloth's y = +hd edge
    const double bs = a.d("box", 0.15);
    auto [xV, xT, box] = sim.presets->rigidbodies->add_box("box", a.d("box_mass", 0.3), bs);
    box.rigidbody.add_rotation(10.0, Eigen::Vector3d::UnitX());
    box.rigidbody.add_translation({ 0.0, hd + 0.5 * bs, 0.0 });
    std::vector<int> edge_points;
    for (int i = 0; i < (int)cV.size(); i++)
        if (std::abs(cV[i].y() - hd) < 1e-9 && std::abs(cV[i].x()) < 0.5 * bs + 1e-9) edge_points.push_back(i);
    sim.interactions->attachments->add(box.rigidbody, cloth.point_set, edge_points, att);

    std::ostringstream js;
    js.precision(17);
    js << "{\"kind\":\"attachzoo\",\"n\":" << n << ",\"size\":" << d << ",\"k\":" << k << ",\"tol\":" << (tol > 1e300 ? -1.0 : tol) << ",\"box\":" << bs
       << ",\"box_mass\":" << a.d("box_mass", 0.3) << ",\"rod_strain_limit\":" << lpA.strain.strain_limit << ",\"rod_damping\":" << lpA.strain.damping << "}";
    sc.json = js.str();
    return sc;
}

// EnergyAttachments::add_by_distance (SURVEY.md §8(f) rank 4 remainder): a patch of cloth glued onto a larger cloth wherever it is closer
// than a distance (it overhangs one corner, so vertices, boundary edges and faces are all nearest entities), and a free rigid box glued
// under the cloth by the distance to its surface mesh. No contact.
static Scene scene_attachdist(const Args& a)
{
    Scene sc;
    stark::Settings settings = base_settings(a, "attachdist");
    settings.simulation.init_frictional_contact = false;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const int n = a.i("n", 6);
    const double d = a.d("size", 0.4), hd = 0.5 * d;
    const double k = a.d("k", 1e4), tol = a.d("tol", std::numeric_limits<double>::max());
    const double gap = a.d("gap", 0.004), dist = a.d("dist", 0.08), box_dist = a.d("box_dist", 0.02);
    auto [cV, cT, cloth] = sim.presets->deformables->add_surface_grid("cloth", { d, d }, { n, n }, stark::Surface::Params::Cotton_Fabric());
    std::vector<int> corners;
    for (int i = 0; i < (int)cV.size(); i++)
        if (std::abs(std::abs(cV[i].x()) - hd) < 1e-9 && std::abs(cV[i].y() + hd) < 1e-9) corners.push_back(i);
    sim.deformables->prescribed_positions->add(cloth.point_set, corners, stark::EnergyPrescribedPositions::Params().set_stiffness(1e6));
    auto att = stark::EnergyAttachments::Params().set_stiffness(k).set_tolerance(tol);

    // the patch: a little above the cloth, turned, hanging over the corner (+hd, +hd)
    auto [pV, pT, patch] = sim.presets->deformables->add_surface_grid("patch", { 0.45 * d, 0.45 * d }, { 3, 3 }, stark::Surface::Params::Cotton_Fabric());
    patch.point_set.add_rotation(a.d("turn", 17.0), Eigen::Vector3d::UnitZ());
    patch.point_set.add_displacement({ 0.41 * d, 0.37 * d, gap });
    std::vector<int> all_patch((size_t)pV.size());
    for (int i = 0; i < (int)pV.size(); i++) all_patch[i] = i;
    sim.interactions->attachments->add_by_distance(patch.point_set, cloth.point_set, all_patch, cT, dist, att);

    // the box: its top face just under the cloth's middle
    const double bs = a.d("box", 0.12);
    auto [xV, xT, box] = sim.presets->rigidbodies->add_box("box", a.d("box_mass", 0.2), bs);
    box.rigidbody.add_rotation(a.d("box_turn", 25.0), Eigen::Vector3d::UnitZ());
    box.rigidbody.add_translation({ -0.11 * d, -0.07 * d, -0.5 * bs - gap });
    std::vector<int> all_cloth((size_t)cV.size());
    for (int i = 0; i < (int)cV.size(); i++) all_cloth[i] = i;
    sim.interactions->attachments->add_by_distance(box.rigidbody, cloth.point_set, xV, xT, all_cloth, box_dist, att);

    std::ostringstream js;
    js.precision(17);
    js << "{\"kind\":\"attachdist\",\"n\":" << n << ",\"size\":" << d << ",\"k\":" << k << ",\"tol\":" << (tol > 1e300 ? -1.0 : tol) << ",\"gap\":" << gap
       << ",\"dist\":" << dist << ",\"box_dist\":" << box_dist << ",\"turn\":" << a.d("turn", 17.0) << ",\"box\":" << bs << ",\"box_mass\":" << a.d("box_mass", 0.2) << ",\"box_turn\":" << a.d("box_turn", 25.0) << "}";
    sc.json = js.str();
    return sc;
}

// User-defined potentials through GlobalPotential::add_potential, as README.md:109-126 / examples/main.cpp:666-692 show it: a clamped
// Soft_Rubber block whose vertices are attracted by a magnet ("magnetic": EnergyMagneticAttraction, verbatim from the README) or by several
// weighted poles through a summation loop ("foreach": MappedWorkspace::add_for_each, MappedWorkspace.h:123-130). Names the engine has no
// kernel for: through the shim they run SymX's op sequence on the device interpreter (tests/shim/shim_check.cpp builds the same scenes).
struct UserPotentialData
{
    double magnet_force = 100.0;
    Eigen::Vector3d magnet_center = { 0.3, 0.2, 1.6 };
    symx::LabelledConnectivity<1> vertices{ { "point" } };
    std::vector<std::array<double, 4>> poles = { { 0.3, 0.2, 1.6, 0.6 }, { -0.4, 0.1, 1.7, 0.4 }, { 0.0, -0.5, 1.5, 0.5 } };
};
static Scene scene_userpot(const Args& a, const std::string& kind)
{
    Scene sc;
    stark::Settings settings = base_settings(a, kind);
    settings.simulation.init_frictional_contact = false;
    sc.sim = std::make_unique<stark::Simulation>(settings);
    auto& sim = *sc.sim;
    const int n = a.i("n", 3);
    auto [V, T] = stark::generate_tet_grid({ 0.0, 0.0, 0.6 }, { 1.0, 1.0, 1.0 }, { n, n, n });
    auto block = sim.presets->deformables->add_volume("block", V, T, stark::Volume::Params::Soft_Rubber());
    sim.deformables->prescribed_positions->add_inside_aabb(block.point_set, { 0.0, 0.0, 0.1 }, { 2.0, 2.0, 2e-3 }, stark::EnergyPrescribedPositions::Params().set_stiffness(1e7));
    auto data = std::make_shared<UserPotentialData>();
    data->magnet_force = a.d("k", 20.0);
    sc.keep = data;
    for (int v = 0; v < (int)block.point_set.size(); v++) data->vertices.push_back({ block.point_set.get_global_index(v) });
    stark::core::Stark* stark_core = &sim.get_stark();
    stark::PointDynamics* dyn = sim.deformables->point_sets.get();
    UserPotentialData* d = data.get();
    if (kind == "magnetic") {
        stark_core->global_potential->add_potential("EnergyMagneticAttraction", d->vertices,
            [stark_core, dyn, d](symx::MappedWorkspace<double>& mws, symx::Element& elem)
            {
                symx::Vector v1 = mws.make_vector(dyn->v1.data, elem["point"]);
                symx::Vector x0 = mws.make_vector(dyn->x0.data, elem["point"]);
                symx::Scalar dt = mws.make_scalar(stark_core->dt);
                symx::Scalar k = mws.make_scalar(d->magnet_force);
                symx::Vector m = mws.make_vector(d->magnet_center);
                symx::Vector x1 = stark::time_integration(x0, v1, dt);
                symx::Vector r = x1 - m;
                symx::Scalar dist = r.norm();
                return -k / dist;
            });
    } else {
        stark_core->global_potential->add_potential("EnergyMultipoleAttraction", d->vertices,
            [stark_core, dyn, d](symx::MappedWorkspace<double>& mws, symx::Element& elem)
            {
                symx::Vector v1 = mws.make_vector(dyn->v1.data, elem["point"]);
                symx::Vector x0 = mws.make_vector(dyn->x0.data, elem["point"]);
                symx::Scalar dt = mws.make_scalar(stark_core->dt);
                symx::Scalar k = mws.make_scalar(d->magnet_force);
                symx::Vector x1 = stark::time_integration(x0, v1, dt);
                return mws.add_for_each(d->poles, [&](symx::Vector& pole) {
                    symx::Vector r = x1 - symx::Vector({ pole[0], pole[1], pole[2] });
                    return -k * pole[3] / r.norm();
                });
            });
    }
    std::ostringstream js;
    js << "{\"kind\":\"" << kind << "\",\"n\":" << n << "}";
    sc.json = js.str();
    return sc;
}

static Scene make_scene(const std::string& name, const Args& a)
{
    if (name == "magnetic" || name == "foreach") return scene_userpot(a, name);
    if (name == "hangingnet") return scene_hangingnet(a);
    if (name == "attachzoo") return scene_attachzoo(a);
    if (name == "attachdist") return scene_attachdist(a);
    if (name == "clothbox") return scene_clothbox(a);
    if (name == "blockbox") return scene_blockbox(a);
    if (name == "mixed") return scene_mixed(a);
    if (name == "contactrods") return scene_contactrods(a);
    if (name == "contactcorners") return scene_contactcorners(a);
    if (name == "contactmix") return scene_contactmix(a);
    if (name == "rbchain") return scene_rbchain(a);
    if (name == "tetblock") return scene_tetblock(a);
    if (name == "tetbeam") return scene_tetbeam(a);
    if (name == "hangingbox") return scene_hangingbox(a);
    if (name == "pycloth") return scene_pycloth(a);
    if (name == "cloth") return scene_cloth(a);
    std::cerr << "unknown scene " << name << std::endl;
    exit(2);
}

// --------------------------------------------------------------------------------------------------------------------
// Generic snapshot of the evaluator inputs + the reference's stage outputs at the CURRENT state
// --------------------------------------------------------------------------------------------------------------------
static std::string jstr(const std::string& s) { return "\"" + s + "\""; }

static void dump_snapshot(Scene& sc, const std::string& dir, const Args& a)
{
    fs::create_directories(dir);
    stark::core::Stark& st = sc.sim->get_stark();
    auto gp = st.global_potential;
    const auto& potentials = gp->get_potentials();
    const auto& dof_maps = gp->get_dof_maps();
    const std::vector<int32_t> dof_offsets = gp->get_dofs_offsets();
    const int ndofs = gp->get_total_n_dofs();

    std::ostringstream man;
    man.precision(17);
    man << "{\n\"scene\":" << sc.json << ",\n\"dt\":" << st.dt << ",\n\"gravity\":[" << st.gravity[0] << "," << st.gravity[1] << "," << st.gravity[2] << "],\n";
    man << "\"ndofs\":" << ndofs << ",\n\"dof_sets\":[";
    for (int i = 0; i < gp->get_n_dof_sets(); i++) {
        man << (i ? "," : "") << "{\"label\":" << jstr(gp->get_dof_label(i)) << ",\"offset\":" << dof_offsets[i] << ",\"size\":" << gp->get_n_dofs(i) << "}";
    }
    man << "],\n";

    // Unique arrays keyed by (host id, stride)
    std::map<std::pair<std::uintptr_t, int>, int> array_ids;
    std::map<std::uintptr_t, int> dof_set_of_id;
    for (int i = 0; i < (int)dof_maps.size(); i++) dof_set_of_id[dof_maps[i].id()] = i;
    auto get_array = [&](const symx::DataMap<const double>& m) {
        auto key = std::make_pair(m.id(), (int)m.stride);
        auto it = array_ids.find(key);
        if (it != array_ids.end()) return it->second;
        const int k = (int)array_ids.size();
        array_ids[key] = k;
        const size_t n = (size_t)m.n_elements();
        // NB: for DoF arrays n_elements() of the potential-side map is the number of items (e.g. points)
        npy_f64(dir + "/a" + std::to_string(k) + ".npy", m.data(), { n, (size_t)m.stride });
        return k;
    };

    // The DoF vector
    std::vector<double> u(ndofs);
    gp->get_dofs(u.data());
    npy_f64(dir + "/dofs.npy", u.data(), { (size_t)ndofs });

    // Per-potential inputs and per-potential reference outputs
    const int nthreads = 1;  // deterministic element order
    Eigen::VectorXd grad_total = Eigen::VectorXd::Zero(ndofs);
    double E_total = 0.0;
    man << "\"potentials\":[\n";
    for (int pi = 0; pi < (int)potentials.size(); pi++) {
        const symx::Potential& pot = *potentials[pi];
        auto mws = pot.get_mws();
        const int n_elem = mws->conn.n_elements();
        const int stride = mws->conn.stride;
        const std::string P = dir + "/p" + std::to_string(pi);
        if (n_elem > 0) npy_i32(P + "_conn.npy", mws->conn.data(), { (size_t)n_elem, (size_t)stride });
        man << (pi ? ",\n" : "") << "{\"name\":" << jstr(pot.get_name()) << ",\"n_elem\":" << n_elem << ",\"conn_stride\":" << stride
            << ",\"has_condition\":" << (pot.has_conditional() ? 1 : 0) << ",\"bindings\":[";
        for (size_t bi = 0; bi < mws->maps.size(); bi++) {
            const auto& m = mws->maps[bi];
            int dof_set = -1;
            auto it = dof_set_of_id.find(m.id());
            if (it != dof_set_of_id.end()) dof_set = it->second;
            int arr = -1;
            if (n_elem > 0 || m.connectivity_index < 0) arr = get_array(m);
            man << (bi ? "," : "") << "{\"array\":" << arr << ",\"stride\":" << m.stride << ",\"conn\":" << m.connectivity_index << ",\"dof_set\":" << dof_set << "}";
        }
        man << "]";

        if (n_elem > 0) {
            // the energy (and condition) as SymX's own straight-line op sequence (symx/src/compile/Sequence.h:24-41; semantics of an op:
            // Compilation.cpp:381-469): rows {type, dst, a, b, cond} + one constant per op. What a generic evaluator consumes.
            auto dump_ops = [&](const symx::Scalar& expr, const std::string& tag) {
                symx::Sequence seq({ expr });
                std::vector<int32_t> rows;
                std::vector<double> consts;
                for (const auto& op : seq.ops) {
                    rows.insert(rows.end(), { (int32_t)op.type, op.dst, op.a, op.b, op.cond });
                    consts.push_back(op.constant);
                }
                npy_i32(P + "_" + tag + ".npy", rows.data(), { seq.ops.size(), (size_t)5 });
                npy_f64(P + "_" + tag + "c.npy", consts.data(), { seq.ops.size() });
                return seq.get_n_inputs();
            };
            man << ",\"n_inputs\":" << dump_ops(pot.get_expression(), "ops");
            if (pot.has_conditional()) dump_ops(pot.get_condition(), "cops");
            symx::DeferredParallelTasks tasks;
            symx::SecondOrderCompiledPotential cp(pot, dof_maps, st.context->compilation_directory, tasks);
            tasks.run(8);
            symx::Assembly as;
            as.start(dof_offsets, nthreads, true, true);
            cp.evaluate_P__dP_du__local_d2P_du2(as);
            as.stop(true, true);
            const double E = as.E.get_solution();
            const Eigen::VectorXd& g = as.grad.get_solution();
            E_total += E;
            grad_total += g;
            auto& hs = as.element_hessians->hessians;
            const int m = (int)hs.size();
            man << ",\"E\":" << E << ",\"n_hessians\":" << m;
            if (m > 0) {
                const int nb = hs[0].n_blocks_per_dim;
                std::vector<int32_t> rows((size_t)m * nb);
                std::vector<double> vals((size_t)m * 9 * nb * nb);
                for (int e = 0; e < m; e++) {
                    std::memcpy(&rows[(size_t)e * nb], hs[e].block_rows, nb * sizeof(int32_t));
                    std::memcpy(&vals[(size_t)e * 9 * nb * nb], hs[e].values, 9 * nb * nb * sizeof(double));
                }
                npy_i32(P + "_hrows.npy", rows.data(), { (size_t)m, (size_t)nb });
                npy_f64(P + "_hvals.npy", vals.data(), { (size_t)m, (size_t)(3 * nb), (size_t)(3 * nb) });
                man << ",\"n_blocks\":" << nb;

                // Projection of every element Hessian (project_to_PD.cpp:12-82 semantics, eps = 1e-10, no mirroring)
                as.element_hessians->project_to_PD_inplace__all(1e-10, false);
                for (int e = 0; e < m; e++) {
                    std::memcpy(&vals[(size_t)e * 9 * nb * nb], hs[e].values, 9 * nb * nb * sizeof(double));
                }
                npy_f64(P + "_hvals_proj.npy", vals.data(), { (size_t)m, (size_t)(3 * nb), (size_t)(3 * nb) });
            }
            npy_f64(P + "_grad.npy", g.data(), { (size_t)ndofs });
            // Energy-only and energy+gradient variants must agree with the full one (they are separate JIT kernels)
            {
                symx::Assembly as2;
                as2.start(dof_offsets, nthreads, false, false);
                cp.evaluate_P(as2);
                as2.stop(false, false);
                man << ",\"E_only\":" << as2.E.get_solution();
            }
        }
        man << "}";
    }
    man << "\n],\n";
    man << "\"n_arrays\":" << array_ids.size() << ",\n";

    // Global evaluation through the reference's own SecondOrderCompiledGlobal, assembly, preconditioner and PCG
    {
        symx::SecondOrderCompiledGlobal g(gp, st.context);
        double E = 0.0;
        Eigen::VectorXd grad(ndofs);
        const int nt_save = st.context->n_threads;
        st.context->n_threads = nthreads;
        auto eh = g.evaluate_P__dP_du__local_d2P_du2(E, grad);
        man << "\"E\":" << E << ",\n\"E_sum_of_potentials\":" << E_total << ",\n\"n_hessians\":" << eh->size() << ",\n";
        npy_f64(dir + "/grad.npy", grad.data(), { (size_t)ndofs });
        auto A = eh->assemble_global(nthreads, ndofs);
        std::vector<Eigen::Triplet<double>> trip;
        A->to_triplets(trip);
        // to scalar COO (row, col, val); values are the float-stored entries widened to double
        std::vector<int32_t> tr(trip.size()), tc(trip.size());
        std::vector<double> tv(trip.size());
        for (size_t i = 0; i < trip.size(); i++) { tr[i] = trip[i].row(); tc[i] = trip[i].col(); tv[i] = trip[i].value(); }
        npy_i32(dir + "/A_rows.npy", tr.data(), { trip.size() });
        npy_i32(dir + "/A_cols.npy", tc.data(), { trip.size() });
        npy_f64(dir + "/A_vals.npy", tv.data(), { trip.size() });
        man << "\"nnz_scalar\":" << trip.size() << ",\n";

        // Linear solve exactly as NewtonsMethod::_solve_linear_system (NewtonsMethod.cpp:421-446)
        const double residual = grad.cwiseAbs().maxCoeff();
        const double forcing = std::min(1e-2, residual * std::min(0.5, std::sqrt(residual)));
        const double abs_tol = std::max(forcing, 1e-12);
        A->set_preconditioner(bsm::Preconditioner::BlockDiagonal);
        A->prepare_preconditioning(nthreads);
        Eigen::VectorXd du = Eigen::VectorXd::Zero(ndofs);
        Eigen::VectorXd rhs = -grad;
        bsm::PCGContext ctx;
        bsm::PCGInfo info = bsm::solve_pcg(*A, du.data(), rhs.data(), ndofs, abs_tol, 1e-4, 10000, nthreads, true, ctx);
        npy_f64(dir + "/pcg_x.npy", du.data(), { (size_t)ndofs });
        man << "\"residual\":" << residual << ",\n\"pcg\":{\"abs_tol\":" << abs_tol << ",\"rel_tol\":1e-4,\"converged\":" << (info.converged ? 1 : 0)
            << ",\"iterations\":" << info.n_iterations << ",\"error\":" << info.error << ",\"indefinite\":" << (info.found_indefiniteness ? 1 : 0) << "},\n";
        // SpMV probe: y = A * x with x_i = sin(0.37 i)
        Eigen::VectorXd xin(ndofs), yout(ndofs);
        for (int i = 0; i < ndofs; i++) xin[i] = std::sin(0.37 * i);
        A->spmxv_from_ptr(yout.data(), xin.data(), nthreads);
        npy_f64(dir + "/spmv_y.npy", yout.data(), { (size_t)ndofs });
        // Preconditioner probe: z = M^-1 x
        Eigen::VectorXd z(ndofs);
        A->apply_preconditioning(z.data(), xin.data(), nthreads);
        npy_f64(dir + "/prec_z.npy", z.data(), { (size_t)ndofs });
        st.context->n_threads = nt_save;
    }
    // Collision meshes handed to the contact model (detector inputs; the tables above are the reference's detector outputs)
    if (!sc.contact_meshes.empty()) {
        man << "\"contact\":{\"stiffness\":" << sc.sim->interactions->contact->get_contact_stiffness() << ",\"meshes\":[";
        for (size_t k = 0; k < sc.contact_meshes.size(); k++) {
            const ContactMeshRecord& m = sc.contact_meshes[k];
            const std::string P = dir + "/cm" + std::to_string(k);
            std::vector<int32_t> v(m.verts.begin(), m.verts.end()), t, e;
            for (auto& tri : m.triangles) t.insert(t.end(), tri.begin(), tri.end());
            for (auto& ed : m.edges) e.insert(e.end(), ed.begin(), ed.end());
            npy_i32(P + "_verts.npy", v.data(), { v.size() });
            npy_i32(P + "_tris.npy", t.data(), { m.triangles.size(), (size_t)3 });
            npy_i32(P + "_edges.npy", e.data(), { m.edges.size(), (size_t)2 });
            man << (k ? "," : "") << "{\"kind\":" << jstr(m.kind) << ",\"idx_in_ps\":" << m.idx_in_ps << ",\"thickness\":" << m.thickness << "}";
        }
        man << "],\"friction\":[";
        for (size_t k = 0; k < sc.friction_pairs.size(); k++)
            man << (k ? "," : "") << "[" << (int)sc.friction_pairs[k][0] << "," << (int)sc.friction_pairs[k][1] << "," << sc.friction_pairs[k][2] << "]";
        man << "]},\n";
    }
    man << "\"end\":0\n}\n";
    std::ofstream(dir + "/manifest.json") << man.str();
}

// Deterministic perturbation of the DoFs so that every derivative path is exercised: u_i += amp * sin(1.3 i + 0.7)
static void perturb_dofs(stark::core::Stark& st, double amp)
{
    const int n = st.global_potential->get_total_n_dofs();
    std::vector<double> u(n);
    st.global_potential->get_dofs(u.data());
    for (int i = 0; i < n; i++) u[i] += amp * std::sin(1.3 * i + 0.7);
    st.global_potential->set_dofs(u.data());
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::cerr << "usage: ref_harness dump|time|traj <scene> [key=value ...]" << std::endl;
        return 2;
    }
    const std::string mode = argv[1];
    const std::string scene_name = argv[2];
    Args a;
    for (int i = 3; i < argc; i++) {
        std::string s = argv[i];
        auto p = s.find('=');
        if (p != std::string::npos) a.kv[s.substr(0, p)] = s.substr(p + 1);
    }
    if (mode == "geom") {
        // Known-answer vectors of the narrow-phase classification (ipc_toolkit_geometry_functions.cpp), the edge-triangle
        // intersection test and the friction geometry (friction_geometry.cpp) on seeded pseudo-random primitives
        const std::string dir = a.s("out", "/tmp/mistark_fixture");
        fs::create_directories(dir);
        const int n = a.i("n", 400);
        uint64_t state = 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() {  // xorshift64*, uniform in [-1, 1)
            state ^= state >> 12; state ^= state << 25; state ^= state >> 27;
            return (double)((state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0 * 2.0 - 1.0;
        };
        auto V = [](const double* x) { return tmcd::Vec3d(x[0], x[1], x[2]); };
        auto Ev = [](const double* x) { return Eigen::Vector3d(x[0], x[1], x[2]); };
        std::vector<double> pt_in((size_t)n * 12), pt_d2(n), ee_in((size_t)n * 12), ee_d2(n), et_in((size_t)n * 15);
        std::vector<int32_t> pt_type(n), ee_type(n), et_hit(n);
        std::vector<double> fr_pt((size_t)n * 9), fr_pe((size_t)n * 8), fr_pp((size_t)n * 6), fr_ee((size_t)n * 8);
        for (int i = 0; i < n; i++) {
            // point - triangle: the point is placed over/around the triangle so that every region occurs
            double* x = &pt_in[(size_t)i * 12];
            for (int k = 3; k < 12; k++) x[k] = rnd();
            {
                const double a0 = 1.25 * rnd() + 0.25, b0 = 1.25 * rnd() + 0.25, h = 0.2 * rnd();
                const tmcd::Vec3d t0 = V(x + 3), t1 = V(x + 6), t2 = V(x + 9);
                const tmcd::Vec3d nrm = (t1 - t0).cross(t2 - t0);
                const tmcd::Vec3d p = t0 + (t1 - t0) * a0 + (t2 - t0) * b0 + nrm * h;
                for (int k = 0; k < 3; k++) x[k] = p[k];
                tmcd::PointTriangleDistanceType ty;
                pt_d2[i] = tmcd::point_triangle_sq_distance(ty, p, t0, t1, t2);
                pt_type[i] = (int)ty;
                const auto bary = stark::barycentric_point_triangle(Ev(x), Ev(x + 3), Ev(x + 6), Ev(x + 9));
                const auto T = stark::projection_matrix_triangle(Ev(x + 3), Ev(x + 6), Ev(x + 9));
                for (int k = 0; k < 3; k++) fr_pt[(size_t)i * 9 + k] = bary[k];
                for (int k = 0; k < 6; k++) fr_pt[(size_t)i * 9 + 3 + k] = T[k];
                const auto b2 = stark::barycentric_point_edge(Ev(x), Ev(x + 3), Ev(x + 6));
                const auto T2 = stark::projection_matrix_point_edge(Ev(x), Ev(x + 3), Ev(x + 6));
                for (int k = 0; k < 2; k++) fr_pe[(size_t)i * 8 + k] = b2[k];
                for (int k = 0; k < 6; k++) fr_pe[(size_t)i * 8 + 2 + k] = T2[k];
                const auto T3 = stark::projection_matrix_point_point(Ev(x), Ev(x + 3));
                for (int k = 0; k < 6; k++) fr_pp[(size_t)i * 6 + k] = T3[k];
            }
            // edge - edge: every fourth pair nearly parallel
            double* y = &ee_in[(size_t)i * 12];
            for (int k = 0; k < 12; k++) y[k] = rnd();
            if (i % 4 == 3)
                for (int k = 0; k < 3; k++) y[9 + k] = y[6 + k] + (y[3 + k] - y[k]) * (0.5 + 0.4 * rnd()) + 1e-7 * rnd();
            {
                tmcd::EdgeEdgeDistanceType ty;
                ee_d2[i] = tmcd::edge_edge_sq_distance(ty, V(y), V(y + 3), V(y + 6), V(y + 9), 1e-30);
                ee_type[i] = (int)ty;
                const auto b = stark::barycentric_edge_edge(Ev(y), Ev(y + 3), Ev(y + 6), Ev(y + 9));
                const auto T = stark::projection_matrix_edge_edge(Ev(y), Ev(y + 3), Ev(y + 6), Ev(y + 9));
                for (int k = 0; k < 2; k++) fr_ee[(size_t)i * 8 + k] = b[k];
                for (int k = 0; k < 6; k++) fr_ee[(size_t)i * 8 + 2 + k] = T[k];
            }
            // edge - triangle intersection: the edge is aimed at a point around the triangle
            double* w = &et_in[(size_t)i * 15];
            for (int k = 6; k < 15; k++) w[k] = rnd();
            {
                const double a0 = 0.8 * rnd() + 0.3, b0 = 0.8 * rnd() + 0.3;
                const tmcd::Vec3d t0 = V(w + 6), t1 = V(w + 9), t2 = V(w + 12);
                const tmcd::Vec3d c = t0 + (t1 - t0) * a0 + (t2 - t0) * b0;
                const tmcd::Vec3d dir(rnd(), rnd(), rnd());
                const double s0 = 0.5 + rnd(), s1 = 0.5 + rnd();
                const tmcd::Vec3d q0 = c - dir * s0, q1 = c + dir * s1;
                for (int k = 0; k < 3; k++) { w[k] = q0[k]; w[3 + k] = q1[k]; }
                et_hit[i] = tmcd::is_edge_intersecting_triangle(q0, q1, t0, t1, t2) ? 1 : 0;
            }
        }
        npy_f64(dir + "/pt_in.npy", pt_in.data(), { (size_t)n, 12 });
        npy_f64(dir + "/pt_d2.npy", pt_d2.data(), { (size_t)n });
        npy_i32(dir + "/pt_type.npy", pt_type.data(), { (size_t)n });
        npy_f64(dir + "/ee_in.npy", ee_in.data(), { (size_t)n, 12 });
        npy_f64(dir + "/ee_d2.npy", ee_d2.data(), { (size_t)n });
        npy_i32(dir + "/ee_type.npy", ee_type.data(), { (size_t)n });
        npy_f64(dir + "/et_in.npy", et_in.data(), { (size_t)n, 15 });
        npy_i32(dir + "/et_hit.npy", et_hit.data(), { (size_t)n });
        npy_f64(dir + "/fr_pt.npy", fr_pt.data(), { (size_t)n, 9 });
        npy_f64(dir + "/fr_pe.npy", fr_pe.data(), { (size_t)n, 8 });
        npy_f64(dir + "/fr_pp.npy", fr_pp.data(), { (size_t)n, 6 });
        npy_f64(dir + "/fr_ee.npy", fr_ee.data(), { (size_t)n, 8 });
        return 0;
    }
    symx::suppress_compiler_output(true);
    Scene sc = make_scene(scene_name, a);
    stark::core::Stark& st = sc.sim->get_stark();
    const int steps = a.i("steps", 2);

    if (mode == "prime") {
        // JIT-compile (or load) every kernel of the scene. Must be run once before `dump` on a cold cache: the
        // reference's cache key hashes the expression graph, which symbolic differentiation (cold cache only) mutates,
        // so a second compiled object created in the same process after a cold compile would never hit the cache.
        sc.step();
        return 0;
    }
    if (mode == "dump") {
        // Run `steps` time steps, then start the next one by hand, perturb v1 and snapshot
        // steps = 0: snapshot at the initial configuration (needs a primed JIT cache, see `prime`)
        for (int s = 0; s < steps; s++) sc.step();
        std::vector<double> u_conv(st.global_potential->get_total_n_dofs());
        st.global_potential->get_dofs(u_conv.data());
        st.callbacks->run_before_time_step();  // v1 <- 0; friction tables; rb caches
        // restart from 60% of the last converged velocities plus a deterministic perturbation
        for (auto& v : u_conv) v *= 0.6;
        st.global_potential->set_dofs(u_conv.data());
        perturb_dofs(st, a.d("amp", 0.05));
        st.callbacks->newton->run_before_energy_evaluation();
        dump_snapshot(sc, a.s("out", "/tmp/mistark_fixture"), a);
        return 0;
    }
    if (mode == "frames") {
        // frame files exactly as the reference writes them (DeformablesMeshOutput / RigidBodiesMeshOutput through Stark::_write_frame,
        // Stark.cpp:314-338): the initial frame and one per time step; also the YAML log (Logger::save_to_disk) and the run summary
        for (int s = 0; s < steps; s++) sc.step();
        st.print();
        return 0;
    }
    if (mode == "slimdump") {
        // Stage outputs of the reference at FULL size (BASELINE configs at their own sizes), at a state anyone can reproduce without the
        // evaluator inputs: after `steps` time steps (0: the initial configuration), start the next step by hand, set every velocity DoF to the
        // closed-form perturbation u_i = amp sin(1.3 i + 0.7), refresh the contact tables, and record: E, the whole gradient, its max
        // norm, per-potential element counts and energies, the connectivity tables of the contact / friction potentials (= the contact
        // sets), the pattern size, the SpMV probe y = A sin(0.37 i), and the PCG outcome at the Newton forcing tolerance.
        const std::string dir = a.s("out", "/tmp/mistark_slim");
        fs::create_directories(dir);
        for (int s = 0; s < steps; s++) sc.step();
        {
            // grid scenes start with exactly parallel edges, where the edge-edge classification of the (lagged, start-of-step) friction
            // tables hangs on the orientation of the collision edges, i.e. on find_surface's hash order: a closed-form displacement of
            // every point, x0_i[d] += xamp sin(0.9 (3 i + d) + 0.3), takes the start geometry out of that degenerate position
            const double xamp = a.d("xamp", 0.0);
            auto& ps = *sc.sim->deformables->point_sets;
            if (xamp != 0.0)
                for (int i = 0; i < (int)ps.size(); i++)
                    for (int d = 0; d < 3; d++) ps.x0.data[(size_t)i][d] += xamp * std::sin(0.9 * (3.0 * i + d) + 0.3);
        }
        st.callbacks->run_before_time_step();
        auto gp = st.global_potential;
        const int ndofs = gp->get_total_n_dofs();
        {
            std::vector<double> u(ndofs);
            const double amp = a.d("amp", 0.01);
            for (int i = 0; i < ndofs; i++) u[i] = amp * std::sin(1.3 * i + 0.7);
            gp->set_dofs(u.data());
        }
        st.callbacks->newton->run_before_energy_evaluation();
        std::ostringstream man;
        man.precision(17);
        man << "{\"scene\":" << sc.json << ",\"dt\":" << st.dt << ",\"ndofs\":" << ndofs << ",\"amp\":" << a.d("amp", 0.01) << ",\"xamp\":" << a.d("xamp", 0.0) << ",\"steps\":" << steps << ",\n\"potentials\":[";
        const auto& potentials = gp->get_potentials();
        for (int pi = 0; pi < (int)potentials.size(); pi++) {
            const symx::Potential& pot = *potentials[pi];
            auto mws = pot.get_mws();
            const int n_elem = mws->conn.n_elements();
            const std::string nm = pot.get_name();
            man << (pi ? "," : "") << "{\"name\":" << jstr(nm) << ",\"n_elem\":" << n_elem << ",\"conn_stride\":" << mws->conn.stride << "}";
            if (n_elem > 0 && (nm.find("contact_") == 0 || nm.find("friction_") == 0))
                npy_i32(dir + "/t_" + nm + ".npy", mws->conn.data(), { (size_t)n_elem, (size_t)mws->conn.stride });
        }
        man << "],\n";
        if (sc.sim->interactions && sc.sim->interactions->contact) man << "\"contact_stiffness\":" << sc.sim->interactions->contact->get_contact_stiffness() << ",\n";
        symx::SecondOrderCompiledGlobal g(gp, st.context);
        double E = 0.0;
        Eigen::VectorXd grad(ndofs);
        const int nthreads = st.context->n_threads;
        auto eh = g.evaluate_P__dP_du__local_d2P_du2(E, grad);
        man << "\"E\":" << E << ",\"n_hessians\":" << eh->size() << ",\"residual\":" << grad.cwiseAbs().maxCoeff() << ",\n";
        npy_f64(dir + "/grad.npy", grad.data(), { (size_t)ndofs });
        auto A = eh->assemble_global(nthreads, ndofs);
        std::vector<Eigen::Triplet<double>> trip;
        A->to_triplets(trip);
        man << "\"nnz_scalar\":" << trip.size() << ",\n";
        trip.clear();
        trip.shrink_to_fit();
        const double residual = grad.cwiseAbs().maxCoeff();
        const double forcing = std::min(1e-2, residual * std::min(0.5, std::sqrt(residual)));
        const double abs_tol = std::max(forcing, 1e-12);
        A->set_preconditioner(bsm::Preconditioner::BlockDiagonal);
        A->prepare_preconditioning(nthreads);
        Eigen::VectorXd xin(ndofs), yout(ndofs);
        for (int i = 0; i < ndofs; i++) xin[i] = std::sin(0.37 * i);
        A->spmxv_from_ptr(yout.data(), xin.data(), nthreads);
        npy_f64(dir + "/spmv_y.npy", yout.data(), { (size_t)ndofs });
        Eigen::VectorXd du = Eigen::VectorXd::Zero(ndofs);
        Eigen::VectorXd rhs = -grad;
        bsm::PCGContext ctx;
        bsm::PCGInfo info = bsm::solve_pcg(*A, du.data(), rhs.data(), ndofs, abs_tol, 1e-4, 10000, nthreads, true, ctx);
        man << "\"pcg\":{\"abs_tol\":" << abs_tol << ",\"rel_tol\":1e-4,\"converged\":" << (info.converged ? 1 : 0) << ",\"iterations\":" << info.n_iterations
            << ",\"error\":" << info.error << ",\"indefinite\":" << (info.found_indefiniteness ? 1 : 0) << ",\"x_dot_rhs\":" << du.dot(rhs) << ",\"x_norm\":" << du.norm() << "},\n";
        // the same system with every element Hessian projected to PSD (project_to_PD.cpp:12-82, eps 1e-10): the solve a projected Newton step does
        {
            eh->project_to_PD_inplace__all(1e-10, false);
            auto Ap = eh->assemble_global(nthreads, ndofs);
            Ap->set_preconditioner(bsm::Preconditioner::BlockDiagonal);
            Ap->prepare_preconditioning(nthreads);
            Ap->spmxv_from_ptr(yout.data(), xin.data(), nthreads);
            npy_f64(dir + "/spmv_y_proj.npy", yout.data(), { (size_t)ndofs });
            Eigen::VectorXd dup = Eigen::VectorXd::Zero(ndofs);
            bsm::PCGContext ctx2;
            bsm::PCGInfo ip = bsm::solve_pcg(*Ap, dup.data(), rhs.data(), ndofs, abs_tol, 1e-4, 10000, nthreads, true, ctx2);
            man << "\"pcg_projected\":{\"abs_tol\":" << abs_tol << ",\"rel_tol\":1e-4,\"converged\":" << (ip.converged ? 1 : 0) << ",\"iterations\":" << ip.n_iterations
                << ",\"error\":" << ip.error << ",\"indefinite\":" << (ip.found_indefiniteness ? 1 : 0) << ",\"x_dot_rhs\":" << dup.dot(rhs) << ",\"x_norm\":" << dup.norm() << "},\n";
        }
        man << "\"end\":0}\n";
        std::ofstream(dir + "/slim.json") << man.str();
        return 0;
    }
    if (mode == "traj") {
        // Evaluation-point trace of `steps` time steps. SolverCallbacks::run_is_converged short-circuits on its `false`
        // default (solver_utils.h:51-58) so it cannot be used as a per-iteration hook; before_energy_evaluation runs
        // before every full evaluation (NewtonsMethod.cpp:101) and before every Armijo energy evaluation (:573), so the
        // recorded DoF vectors are: iterate k, its line-search trial points, iterate k+1 (= accepted trial), ...
        // Consecutive duplicates are removed.
        const std::string dir = a.s("out", "/tmp/mistark_traj");
        fs::create_directories(dir);
        st.callbacks->run_before_time_step();  // fills the per-step caches (rigid-body q0_, J0_glob; v1 = 0) the potentials read
        const bool slim = a.i("slim", 0) != 0;  // larger scenes: only the step log and the final state (no evaluator inputs, no iterates)
        if (!slim) dump_snapshot(sc, dir, a);  // evaluator inputs + stage outputs at the initial state (t = 0, v1 = 0)
        std::vector<std::vector<double>> iterates;
        std::vector<int> iter_step;
        int cur_step = 0;
        st.callbacks->newton->add_before_energy_evaluation([&]() {
            std::vector<double> u(st.global_potential->get_total_n_dofs());
            st.global_potential->get_dofs(u.data());
            if (!iterates.empty() && iter_step.back() == cur_step && iterates.back() == u) return;
            if (slim) iterates.clear();  // (keep the last one for the duplicate test only)
            iterates.push_back(u);
            iter_step.push_back(cur_step);
        });
        // `maxstep`: a user callback of SolverCallbacks::add_max_allowed_step (the hook a CCD would use; no STARK model registers one) that
        // allows this fraction of every step: the [max] stage of the line search (NewtonsMethod.cpp:494-506)
        // `vamp`: start velocities v0_i[d] = vamp sin(1.3 (3 i + d) + 0.7) on every point (a violent first step: inverted and indefinite
        // elements, invalid line-search candidates), set before the first time step
        const double vamp = a.d("vamp", 0.0);
        if (vamp != 0.0) {
            auto& pts = *sc.sim->deformables->point_sets;
            for (int i = 0; i < pts.size(); i++)
                for (int d = 0; d < 3; d++) pts.v0.data[(size_t)i][d] = vamp * std::sin(1.3 * (3.0 * i + d) + 0.7);
        }
        const double maxstep = a.d("maxstep", 1.0);
        if (maxstep < 1.0) st.callbacks->newton->add_max_allowed_step([maxstep]() { return maxstep; });
        std::ostringstream man;
        man.precision(17);
        man << "{\"scene\":" << sc.json << ",\"steps\":[";
        auto& lg = *st.context->logger;
        for (cur_step = 0; cur_step < steps; cur_step++) {
            const double dt_used = st.dt;
            const bool have = cur_step > 0;
            const int n_a = have ? lg.get_int("newton_iterations") : 0, l_a = have ? lg.get_timer_count("linear_system_solve") : 0;
            sc.step();
            man << (cur_step ? "," : "") << "{\"dt\":" << dt_used << ",\"time\":" << st.current_time << ",\"newton\":" << lg.get_int("newton_iterations") - n_a
                << ",\"linear_solves\":" << lg.get_timer_count("linear_system_solve") - l_a << "}";
        }
        for (const char* key : {"ls_cap", "ls_max", "ls_inv"}) {
            man << "],\n\"" << key << "\":[";
            const auto& v = lg.get_int_series(key);
            for (size_t i = 0; i < v.size(); i++) man << (i ? "," : "") << v[i];
        }
        for (const char* key : {"n_projected_hessians", "n_hessians"}) {  // (logged as doubles, NewtonsMethod.cpp:199-203)
            man << "],\n\"" << key << "\":[";
            const auto& v = lg.get_double_series(key);
            for (size_t i = 0; i < v.size(); i++) man << (i ? "," : "") << (long long)v[i];
        }
        man << "],\n\"newton_iterations\":[";
        { const auto& v = lg.get_int_series("newton_iterations"); for (size_t i = 0; i < v.size(); i++) man << (i ? "," : "") << v[i]; }
        man << "],\n\"cg_iterations\":[";
        { const auto& v = lg.get_int_series("cg_iterations"); for (size_t i = 0; i < v.size(); i++) man << (i ? "," : "") << v[i]; }
        man << "],\n\"ls_bt\":[";
        { const auto& v = lg.get_int_series("ls_bt"); for (size_t i = 0; i < v.size(); i++) man << (i ? "," : "") << v[i]; }
        man << "],\n\"n_iterates\":" << iterates.size() << ",\"iter_step\":[";
        for (size_t i = 0; i < iter_step.size(); i++) man << (i ? "," : "") << iter_step[i];
        man << "]}\n";
        const size_t nd = st.global_potential->get_total_n_dofs();
        std::vector<double> flat(iterates.size() * nd);
        for (size_t i = 0; i < iterates.size(); i++) std::memcpy(&flat[i * nd], iterates[i].data(), nd * sizeof(double));
        if (!slim) npy_f64(dir + "/iterates.npy", flat.data(), { iterates.size(), nd });
        auto& ps = *sc.sim->deformables->point_sets;
        if (ps.size() > 0) {
            npy_f64(dir + "/x_end.npy", ps.x0.data[0].data(), { (size_t)ps.size(), 3 });
            npy_f64(dir + "/v_end.npy", ps.v0.data[0].data(), { (size_t)ps.size(), 3 });
        }
        std::ofstream(dir + "/traj.json") << man.str();
        return 0;
    }
    if (mode == "time") {
        // warm-up steps (first step builds the sparsity pattern and JIT-loads) then timed steps
        const int warm = a.i("warmup", 1);
        for (int s = 0; s < warm; s++) sc.step();
        auto& lg = *st.context->logger;
        // (the logger's labels exist once a step has run)
        const int newton0 = warm > 0 ? lg.get_int("newton_iterations") : 0;
        const double ls0 = warm > 0 ? lg.get_timer_total("linear_system_solve") : 0.0;
        const int lsn0 = warm > 0 ? lg.get_timer_count("linear_system_solve") : 0;
        // (per time step: Newton iterations, linear solves and CG iterations as the reference's logger counts them)
        std::ostringstream per_step;
        const double t0 = omp_get_wtime();
        for (int s = 0; s < steps; s++) {
            const bool have = warm > 0 || s > 0;
            const int n_a = have ? lg.get_int("newton_iterations") : 0, l_a = have ? lg.get_timer_count("linear_system_solve") : 0, c_a = have ? lg.get_int("cg_iterations") : 0;
            sc.step();
            per_step << (s ? "," : "") << "[" << lg.get_int("newton_iterations") - n_a << "," << lg.get_timer_count("linear_system_solve") - l_a << ","
                     << lg.get_int("cg_iterations") - c_a << "]";
        }
        const double t1 = omp_get_wtime();
        const int newton = lg.get_int("newton_iterations") - newton0;
        const double ls = lg.get_timer_total("linear_system_solve") - ls0;
        const int lsn = lg.get_timer_count("linear_system_solve") - lsn0;
        std::printf("{\"scene\":%s,\"threads\":%d,\"steps\":%d,\"newton_iterations\":%d,\"wall_s\":%.6f,\"newton_steps_per_s\":%.6f,\"linear_solve_s\":%.6f,\"ms_per_linear_solve\":%.6f,\"linear_solves\":%d,\"ndofs\":%d,\"per_step\":[%s]}\n",
            sc.json.c_str(), st.settings.execution.n_threads, steps, newton, t1 - t0, newton / (t1 - t0), ls, lsn > 0 ? 1000.0 * ls / lsn : 0.0, lsn,
            st.global_potential->get_total_n_dofs(), per_step.str().c_str());
        return 0;
    }
    std::cerr << "unknown mode " << mode << std::endl;
    return 2;
}
