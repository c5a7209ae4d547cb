// Test driver of the SymX shim (shim/): the UNMODIFIED reference (stark/src/**, compiled against shim/include/symx in place) builds scenes
// through its own stark::Simulation; the first time step makes the shim's symx::NewtonsMethod register everything with libmistark.
//   MISTARK_SHIM_DRY=1 MISTARK_SHIM_DESCRIBE=<file> shim_check <scene>     registration only (no GPU): the JSON of mistark_describe
//   shim_check <scene> [steps]                                             on a machine with an MI355X: real time steps on the engine
// Built by oracle/Makefile (target _ref/shim_check) only where /root/reference exists; test infrastructure, not shipped.
#include <stark>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <set>
#include <vector>

// `shim_check tmcd_broad`: tmcd::ProximityDetection used directly — in the shim_cd build that is the stand-in on the device detector, in the
// plain shim build the reference's own detector — on a tilted 7 x 7 cloth patch over a box: run(), then get_broad_phase_results(), printed as
// sorted rows. tests/test_gpu_contact.py lays the two binaries' outputs side by side.
static int tmcd_broad_phase_listing()
{
    std::vector<std::array<double, 3>> cv, bv;
    std::vector<std::array<int32_t, 3>> ct, bt;
    const int n = 6;
    for (int j = 0; j <= n; j++)
        for (int i = 0; i <= n; i++) cv.push_back({ -0.3 + 0.1 * i, -0.3 + 0.1 * j + 0.013 * i, 0.002 + 0.003 * i + 0.0011 * j * j });
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            const int a = j * (n + 1) + i, b = a + 1, c = a + n + 1, d = c + 1;
            ct.push_back({ a, b, d });
            ct.push_back({ a, d, c });
        }
    for (int k = 0; k < 8; k++) bv.push_back({ (k & 1) ? 0.5 : -0.5, (k & 2) ? 0.5 : -0.5, (k & 4) ? 0.0 : -0.2 });
    const int quads[6][4] = { { 0, 1, 3, 2 }, { 4, 6, 7, 5 }, { 0, 4, 5, 1 }, { 2, 3, 7, 6 }, { 0, 2, 6, 4 }, { 1, 5, 7, 3 } };
    for (auto& q : quads) {
        bt.push_back({ q[0], q[1], q[2] });
        bt.push_back({ q[0], q[2], q[3] });
    }
    auto edges_of = [](const std::vector<std::array<int32_t, 3>>& T) {
        std::set<std::array<int32_t, 2>> S;
        for (auto& t : T)
            for (int k = 0; k < 3; k++) S.insert({ std::min(t[k], t[(k + 1) % 3]), std::max(t[k], t[(k + 1) % 3]) });
        return std::vector<std::array<int32_t, 2>>(S.begin(), S.end());
    };
    const auto ce = edges_of(ct), be = edges_of(bt);
    tmcd::ProximityDetection pd;
    pd.set_n_threads(1);
    pd.add_mesh(&cv[0][0], (int32_t)cv.size(), &ct[0][0], (int32_t)ct.size(), &ce[0][0], (int32_t)ce.size());
    pd.add_mesh(&bv[0][0], (int32_t)bv.size(), &bt[0][0], (int32_t)bt.size(), &be[0][0], (int32_t)be.size());
    pd.add_blacklist(1, 1);
    pd.activate_point_triangle(true);
    pd.activate_edge_edge(true);
    const auto& res = pd.run(0.004);
    const size_t n_narrow = res.point_triangle.point_point.size() + res.point_triangle.point_edge.size() + res.point_triangle.point_triangle.size() +
                            res.edge_edge.point_point.size() + res.edge_edge.point_edge.size() + res.edge_edge.edge_edge.size();
    const auto& bp = pd.get_broad_phase_results();
    auto rows = [](const std::vector<std::pair<tmcd::SetIndex, tmcd::SetIndex>>& v) {
        std::vector<std::array<int32_t, 4>> r;
        for (auto& p : v) r.push_back({ p.first.set, p.first.idx, p.second.set, p.second.idx });
        std::sort(r.begin(), r.end());
        return r;
    };
    std::cout << "{\"narrow_pairs\":" << n_narrow;
    for (int l = 0; l < 2; l++) {
        const auto r = rows(l == 0 ? bp.point_triangle : bp.edge_edge);
        std::cout << ",\"" << (l == 0 ? "point_triangle" : "edge_edge") << "\":[";
        for (size_t k = 0; k < r.size(); k++) std::cout << (k ? "," : "") << "[" << r[k][0] << "," << r[k][1] << "," << r[k][2] << "," << r[k][3] << "]";
        std::cout << "]";
    }
    std::cout << "}" << std::endl;
    return 0;
}

int main(int argc, char** argv)
{
    const std::string scene = argc > 1 ? argv[1] : "blockbox";
    if (scene == "tmcd_broad") return tmcd_broad_phase_listing();
    const int steps = argc > 2 ? std::atoi(argv[2]) : 1;
    stark::Settings settings = stark::Settings();
    settings.output.output_directory = "/tmp/mistark_shim_out";
    settings.output.codegen_directory = "/tmp/mistark_shim_codegen";
    // SHIM_SCRATCH=<dir>: a directory of this run's own for both (tests running several of these side by side, pytest -n: two runs writing the
    // shared defaults at once is what made tests/test_shim_cpu.py fail now and then under xdist)
    if (const char* scratch = std::getenv("SHIM_SCRATCH")) {
        settings.output.output_directory = std::string(scratch) + "/out";
        settings.output.codegen_directory = std::string(scratch) + "/codegen";
    }
    // argv[4]: an output directory; the reference then writes its frames (VTK), its YAML log and its run summary there, fed by the shim
    const bool with_output = argc > 4;
    if (with_output) settings.output.output_directory = argv[4];
    settings.output.simulation_name = with_output ? scene : "shim_check";
    settings.output.enable_frame_writes = with_output;
    settings.output.enable_output = with_output;
    settings.output.fps = 30;
    settings.execution.n_threads = 1;
    settings.simulation.init_frictional_contact = scene == "blockbox" || scene == "mixed" || scene == "benchblock";
    if (const char* env = std::getenv("SHIM_THREADS")) settings.execution.n_threads = std::atoi(env);  // (the reference's host-side work: contact detection)
    stark::Simulation sim(settings);
    if (scene == "blockbox") {
        // the scene of tests/test_shim_cpu.py: rigid box (registered first) + fixed, 2 x 2 x 2 Soft_Rubber block, frictional contact
        auto gp = stark::EnergyFrictionalContact::GlobalParams();
        gp.default_contact_thickness = 1e-3;
        sim.interactions->contact->set_global_params(gp);
        auto [bV, bT, box] = sim.presets->rigidbodies->add_box("box", 1.0, { 1.0, 1.0, 0.1 });
        sim.rigidbodies->add_constraint_fix(box.rigidbody);
        auto [sV, sT] = stark::generate_tet_grid({ 0.0, 0.0, 0.6 }, { 1.0, 1.0, 1.0 }, { 2, 2, 2 });
        auto block = sim.presets->deformables->add_volume("block", sV, sT, stark::Volume::Params::Soft_Rubber());
        sim.interactions->contact->set_friction(block.contact, box.contact, 0.5);
    } else if (scene == "benchblock" || scene == "benchclamped") {
        // bench.py's workload through the reference's own classes (SHIM_GRID=nx,ny,nz, default 10,10,10): benchblock = configs[3] (block 1.5 mm
        // above a fixed rigid box {3,3,0.1}, thickness 1e-3, mu 0.5, kmin 1e8; the reference's host detection feeds the contact tables),
        // benchclamped = the contact-free variant (bottom face clamped). For timing the drop-in path beside the engine's own scene mirror.
        int g[3] = { 10, 10, 10 };
        if (const char* env = std::getenv("SHIM_GRID")) std::sscanf(env, "%d,%d,%d", &g[0], &g[1], &g[2]);
        if (scene == "benchblock") {
            auto gp = stark::EnergyFrictionalContact::GlobalParams();
            gp.default_contact_thickness = 1e-3;
            gp.min_contact_stiffness = 1e8;
            sim.interactions->contact->set_global_params(gp);
            auto [bV, bT, box] = sim.presets->rigidbodies->add_box("box", 1.0, { 3.0, 3.0, 0.1 });
            sim.rigidbodies->add_constraint_fix(box.rigidbody);
            auto [sV, sT] = stark::generate_tet_grid({ 0.0, 0.0, 0.05 + 0.0015 + 0.5 }, { 1.0, 1.0, 1.0 }, { g[0], g[1], g[2] });
            auto block = sim.presets->deformables->add_volume("block", sV, sT, stark::Volume::Params::Soft_Rubber());
            sim.interactions->contact->set_friction(block.contact, box.contact, 0.5);
        } else {
            auto [sV, sT] = stark::generate_tet_grid({ 0.0, 0.0, 0.6 }, { 1.0, 1.0, 1.0 }, { g[0], g[1], g[2] });
            auto block = sim.presets->deformables->add_volume("block", sV, sT, stark::Volume::Params::Soft_Rubber());
            sim.deformables->prescribed_positions->add_inside_aabb(block.point_set, { 0.0, 0.0, 0.1 }, { 2.0, 2.0, 2e-3 }, stark::EnergyPrescribedPositions::Params().set_stiffness(1e7));
        }
    } else if (scene == "mixed") {
        // BASELINE configs[4] in small (oracle/ref_harness.cpp scene_mixed; tests/test_gpu_scene.py _build_mixed): floor, chain of hinged
        // boxes, tet block, cloth; contact and friction between the layers
        const int nx = 2, nc = 4, nrb = 3;
        const double L = 0.4, gap = 0.003, th = 0.002, mu = 0.5, bx = 1.2, bz = 0.1, link = 0.04, cloth_size = 1.2;
        auto gp = stark::EnergyFrictionalContact::GlobalParams();
        gp.default_contact_thickness = th;
        gp.min_contact_stiffness = 1e6;
        sim.interactions->contact->set_global_params(gp);
        auto ct = sim.interactions->contact;
        auto [fV, fT, floor] = sim.presets->rigidbodies->add_box("floor", 1.0, { bx, bx, bz });
        sim.rigidbodies->add_constraint_fix(floor.rigidbody);
        const double z_block = 0.5 * bz + gap + 0.5 * L, z_cloth = 0.5 * bz + gap + L + gap, z_chain = z_cloth + gap + 0.5 * link, pitch = 1.5 * link;
        std::vector<stark::RigidBodyHandler> links;
        std::vector<stark::EnergyFrictionalContact::Handler> link_contacts;
        for (int i = 0; i < nrb; i++) {
            auto [V, T, h] = sim.presets->rigidbodies->add_box("link", 0.2, { link, link, link });
            h.rigidbody.set_translation({ (i - 0.5 * (nrb - 1)) * pitch, 0.0, z_chain });
            links.push_back(h.rigidbody);
            link_contacts.push_back(h.contact);
        }
        sim.rigidbodies->add_constraint_fix(links[0]);
        for (int i = 0; i + 1 < nrb; i++) {
            sim.rigidbodies->add_constraint_hinge(links[i], links[i + 1], Eigen::Vector3d((i + 0.5 - 0.5 * (nrb - 1)) * pitch, 0.0, z_chain), Eigen::Vector3d::UnitY());
            ct->disable_collision(link_contacts[i], link_contacts[i + 1]);
        }
        auto [sV, sT] = stark::generate_tet_grid({ 0.0, 0.0, z_block }, { L, L, L }, { nx, nx, nx });
        auto soft = sim.presets->deformables->add_volume("block", sV, sT, stark::Volume::Params::Soft_Rubber());
        auto [cV, cT, cloth] = sim.presets->deformables->add_surface_grid("cloth", { cloth_size * L, cloth_size * L }, { nc, nc }, stark::Surface::Params::Cotton_Fabric());
        cloth.point_set.add_displacement({ 0.0, 0.0, z_cloth });
        ct->set_friction(floor.contact, soft.contact, mu);
        ct->set_friction(soft.contact, cloth.contact, mu);
        for (int i = 0; i < nrb; i++) ct->set_friction(link_contacts[i], cloth.contact, mu);
    } else if (scene == "magnetic" || scene == "inplace" || scene == "inplace_sparse" || scene == "foreach") {
        // handled below (user-defined potentials need state that outlives this block)
    } else if (scene == "tetbeam") {
        auto [sV, sT] = stark::generate_tet_grid({ 0.0, 0.0, 0.0 }, { 4.0, 1.0, 1.0 }, { 4, 1, 1 });
        auto beam = sim.presets->deformables->add_volume("beam", sV, sT, stark::Volume::Params::Soft_Rubber());
        sim.deformables->prescribed_positions->add_inside_aabb(beam.point_set, { -2.0, 0.0, 0.0 }, { 2e-3, 2.0, 2.0 }, stark::EnergyPrescribedPositions::Params().set_stiffness(1e7));
    } else if (scene == "cloth") {
        auto [cV, cT, cloth] = sim.presets->deformables->add_surface_grid("cloth", { 0.4, 0.4 }, { 4, 4 }, stark::Surface::Params::Cotton_Fabric());
        sim.deformables->prescribed_positions->add_inside_aabb(cloth.point_set, { -0.2, 0.0, 0.0 }, { 2e-3, 2.0, 2.0 }, stark::EnergyPrescribedPositions::Params().set_stiffness(1e7));
    } else {
        std::cerr << "unknown scene " << scene << std::endl;
        return 2;
    }
    // ---- user-defined potentials through GlobalPotential::add_potential, exactly as README.md:109-126 / examples/main.cpp:666-692 do it: a name
    // the engine has no kernel for, so the shim hands SymX's op sequence of the expression to the device interpreter
    double magnet_force = 20.0;
    if (const char* env = std::getenv("SHIM_MAGNET_K")) magnet_force = std::atof(env);
    Eigen::Vector3d magnet_center = { 0.3, 0.2, 1.6 };
    symx::LabelledConnectivity<1> user_vertices{ { "point" } };
    std::vector<Eigen::Vector3d> targets;   // "inplace": one target per vertex, rewritten by a Newton callback
    double pull_stiffness = 2e3;
    std::vector<std::array<double, 4>> poles = { { 0.3, 0.2, 1.6, 0.6 }, { -0.4, 0.1, 1.7, 0.4 }, { 0.0, -0.5, 1.5, 0.5 } };
    long n_evaluations = 0;
    if (scene == "magnetic" || scene == "inplace" || scene == "inplace_sparse" || scene == "foreach") {
        int n = scene == "magnetic" || scene == "foreach" ? 3 : 18;
        if (const char* env = std::getenv("SHIM_GRID")) n = std::atoi(env);
        auto [sV, sT] = stark::generate_tet_grid({ 0.0, 0.0, 0.6 }, { 1.0, 1.0, 1.0 }, { n, n, n });
        auto block = sim.presets->deformables->add_volume("block", sV, sT, stark::Volume::Params::Soft_Rubber());
        sim.deformables->prescribed_positions->add_inside_aabb(block.point_set, { 0.0, 0.0, 0.1 }, { 2.0, 2.0, 2e-3 }, stark::EnergyPrescribedPositions::Params().set_stiffness(1e7));
        for (int v = 0; v < (int)block.point_set.size(); v++) user_vertices.push_back({ block.point_set.get_global_index(v) });
        stark::core::Stark& stark_core = sim.get_stark();
        stark::PointDynamics* dyn = sim.deformables->point_sets.get();
        if (scene == "magnetic") {
            stark_core.global_potential->add_potential("EnergyMagneticAttraction", user_vertices,
                [&, dyn](symx::MappedWorkspace<double>& mws, symx::Element& elem)
                {
                    symx::Vector v1 = mws.make_vector(dyn->v1.data, elem["point"]);
                    symx::Vector x0 = mws.make_vector(dyn->x0.data, elem["point"]);
                    symx::Scalar dt = mws.make_scalar(stark_core.dt);
                    symx::Scalar k = mws.make_scalar(magnet_force);
                    symx::Vector m = mws.make_vector(magnet_center);
                    symx::Vector x1 = stark::time_integration(x0, v1, dt);
                    symx::Vector r = x1 - m;
                    symx::Scalar d = r.norm();
                    return -k / d;
                });
        } else if (scene == "foreach") {
            // a summation loop (MappedWorkspace::add_for_each, MappedWorkspace.h:123-130; what SymX's fem integrators use for quadrature rules):
            // attraction to several poles {x, y, z, weight}, summed per vertex BEFORE the element's projection to PD
            stark_core.global_potential->add_potential("EnergyMultipoleAttraction", user_vertices,
                [&, dyn](symx::MappedWorkspace<double>& mws, symx::Element& elem)
                {
                    symx::Vector v1 = mws.make_vector(dyn->v1.data, elem["point"]);
                    symx::Vector x0 = mws.make_vector(dyn->x0.data, elem["point"]);
                    symx::Scalar dt = mws.make_scalar(stark_core.dt);
                    symx::Scalar k = mws.make_scalar(magnet_force);
                    symx::Vector x1 = stark::time_integration(x0, v1, dt);
                    return mws.add_for_each(poles, [&](symx::Vector& pole) {
                        symx::Vector r = x1 - symx::Vector({ pole[0], pole[1], pole[2] });
                        return -k * pole[3] / r.norm();
                    });
                });
        } else {
            for (int v = 0; v < (int)block.point_set.size(); v++) targets.push_back(dyn->x0.data[(size_t)block.point_set.get_global_index(v)] + Eigen::Vector3d(0.0, 0.0, 0.02));
            stark_core.global_potential->add_potential("EnergyPullToTargets", user_vertices,
                [&, dyn](symx::MappedWorkspace<double>& mws, symx::Element& elem)
                {
                    symx::Vector v1 = mws.make_vector(dyn->v1.data, elem["point"]);
                    symx::Vector x0 = mws.make_vector(dyn->x0.data, elem["point"]);
                    symx::Vector t = mws.make_vector(targets, elem["point"]);
                    symx::Scalar dt = mws.make_scalar(stark_core.dt);
                    symx::Scalar k = mws.make_scalar(pull_stiffness);
                    symx::Vector x1 = stark::time_integration(x0, v1, dt);
                    return 0.5 * k * (x1 - t).squared_norm();
                });
            // a Newton callback that REWRITES the large target array in place (same address, same size) in the middle of a solve: at its 4th
            // energy evaluation every target moves ("inplace"), or only three of them do ("inplace_sparse": what a sampled check cannot see)
            // (SHIM_EDIT_AT: the evaluation at which the edit happens; 1 = before anything has been evaluated: every evaluation of the run sees the
            // edited targets — the run a solve REDONE after a missed edit must reproduce)
            long edit_at = 4;
            if (const char* env = std::getenv("SHIM_EDIT_AT")) edit_at = std::atol(env);
            stark_core.callbacks->newton->add_before_energy_evaluation([&, edit_at, sparse = scene == "inplace_sparse"]() {
                if (++n_evaluations != edit_at) return;
                if (sparse) {
                    for (size_t v : { (size_t)100, targets.size() / 2 + 7, targets.size() - 50 })   /* none of them inside a sampled window */ targets[v] += Eigen::Vector3d(0.3, 0.0, 0.0);
                } else {
                    for (auto& t : targets) t += Eigen::Vector3d(0.01, -0.005, 0.03);
                }
            });
        }
    }
    // the first step registers everything and builds the sparsity pattern: timed apart from the rest
    const char* dry_env = std::getenv("MISTARK_SHIM_DRY");
    const bool dry = dry_env && dry_env[0] == '1';
    auto newton_total = [&] {
        long n = 0;
        if (dry) return n;  // (registration only: nothing is solved, the logger has no such series)
        for (int v : sim.get_stark().context->logger->get_int_series("newton_iterations")) n += v;
        return n;
    };
    const auto t0 = std::chrono::steady_clock::now();
    if (steps > 0) sim.run_one_time_step();
    const auto t1 = std::chrono::steady_clock::now();
    const long n1 = newton_total();
    for (int s = 1; s < steps; s++) sim.run_one_time_step();
    const auto t2 = std::chrono::steady_clock::now();
    const long n2 = newton_total();
    const double first_s = std::chrono::duration<double>(t1 - t0).count(), rest_s = std::chrono::duration<double>(t2 - t1).count();
    std::cout << "shim_check: " << steps << " step(s) of '" << scene << "' done" << std::endl;
    std::cout.precision(6);
    std::cout << "{\"scene\":\"" << scene << "\",\"steps\":" << steps << ",\"first_step_s\":" << first_s << ",\"newton_first\":" << n1 << ",\"rest_s\":" << rest_s
              << ",\"newton_rest\":" << (n2 - n1) << ",\"newton_steps_per_s_after_first\":" << (rest_s > 0 ? (n2 - n1) / rest_s : 0.0) << "}" << std::endl;
    if (with_output) sim.get_stark().print();  // run summary + final YAML log (Stark.cpp:254-282)
    if (argc > 3) {
        // what the run produced: Newton iterations per solve (the series the shim logs like the reference, NewtonsMethod.cpp:249) and the
        // positions of all points
        auto& st = sim.get_stark();
        std::ofstream f(argv[3]);
        f.precision(17);
        f << "{\"newton_iterations\":[";
        const auto& it = st.context->logger->get_int_series("newton_iterations");
        for (size_t i = 0; i < it.size(); i++) f << (i ? "," : "") << it[i];
        f << "],\"time\":" << st.current_time << ",\"x\":[";
        auto& ps = *sim.deformables->point_sets;
        for (int i = 0; i < (int)ps.size(); i++) f << (i ? "," : "") << "[" << ps.x0.data[(size_t)i][0] << "," << ps.x0.data[(size_t)i][1] << "," << ps.x0.data[(size_t)i][2] << "]";
        f << "]}\n";
    }
    return 0;
}
