// newton.cpp — host control flow of Newton's method over the device engine.
//
// Mirrors symx::NewtonsMethod::solve (symx/src/solver/NewtonsMethod.cpp:28-252), _project_and_assemble (:254-352),
// _increase/_decrease_projection (:354-386), _solve_linear_system (:388-457) and _line_search_inplace (:459-641):
// same decisions in the same order, every vector stays on the device, only scalars cross PCIe.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "engine.hpp"

namespace mistark {

namespace {
struct Timer
{
    double& acc;
    std::chrono::steady_clock::time_point t0;
    explicit Timer(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~Timer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
// GPU-side stage times without synchronising: mark(stage) records an event on the engine's stream ("from here on the queued work
// belongs to `stage`"); collect() runs after the solve's final synchronisation and adds the time between consecutive marks to the stage
// of the earlier one. (Host timers around asynchronous launches would charge a stage's GPU time to whoever synchronises next; the
// earlier version synchronised after every stage instead, which left the GPU idle while the host prepared the next one.)
enum Stage { ST_EVAL_PGH, ST_EVAL_P, ST_ASSEMBLY, ST_PROJECT, ST_SOLVE, ST_OTHER, ST_COUNT };
struct GpuStages
{
    Context& c;
    std::vector<std::pair<int, size_t>> marks;
    size_t used = 0;
    explicit GpuStages(Context& ctx) : c(ctx) {}
    void mark(int stage)
    {
        if (used == c.stage_ev.size()) {
            hipEvent_t e;
            MS_CHECK(hipEventCreate(&e));
            c.stage_ev.push_back(e);
        }
        MS_CHECK(hipEventRecord(c.stage_ev[used], c.stream));
        marks.emplace_back(stage, used++);
    }
    void collect(double* seconds /*[ST_COUNT]*/)
    {
        if (marks.empty()) return;
        MS_CHECK(hipEventSynchronize(c.stage_ev[marks.back().second]));
        for (size_t i = 0; i + 1 < marks.size(); i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c.stage_ev[marks[i].second], c.stage_ev[marks[i + 1].second]) == hipSuccess) seconds[marks[i].first] += 1e-3 * ms;
        }
    }
};
}  // namespace

int newton_solve(Context& c, const mistark_newton_settings& s, const mistark_newton_callbacks* cb, mistark_newton_stats& st)
{
    Timer t_total(st.t_total);
    prepare(c);
    GpuStages gs(c);
    double stage_s[ST_COUNT] = {0, 0, 0, 0, 0, 0};
    const int64_t ndofs = c.ndofs;
    auto sync = [&]() { MS_CHECK(hipStreamSynchronize(c.stream)); };
    auto call_void = [&](void (*f)(void*)) {
        if (cb && f) {
            Timer t(st.t_callbacks);
            sync();
            f(cb->user);
        }
    };
    auto call_bool = [&](int (*f)(void*), bool def) -> bool {
        if (cb && f) {
            Timer t(st.t_callbacks);
            sync();
            return f(cb->user) != 0;
        }
        return def;
    };

    double E0 = 0.0, du_dot_grad = 0.0, res_0 = std::numeric_limits<double>::max();
    // MISTARK_NEWTON_TRACE=1: one line per Newton iteration and per linear solve on stderr (what the reference prints at Verbosity::Full)
    static const bool trace = std::getenv("MISTARK_NEWTON_TRACE") != nullptr;
    int result = MISTARK_RUNNING;
    int pdn_countdown = 0;
    double ppn_threshold = -1.0;

    c.newton_log.clear();
    if (!call_bool(cb ? cb->is_initial_state_valid : nullptr, true)) result = MISTARK_INVALID_INITIAL_STATE;

    int it = -1;
    while (result == MISTARK_RUNNING) {
        it++;
        if (it == s.max_iterations) {
            result = s.max_iterations_as_success ? MISTARK_SUCCESSFUL : MISTARK_TOO_MANY_ITERATIONS;
            break;
        }
        // default residual: ||grad||_inf (solver_utils.h:28), read back with the energy. Progressive / no projection never reads the
        // double blocks of an element it does not project: the closed-form tets write float blocks only (eval: lazy)
        const bool lazy = c.lazy_allowed && (s.projection_mode == MISTARK_PROJ_PROGRESSIVE || s.projection_mode == MISTARK_PROJ_NEWTON) && s.linear_solver != MISTARK_SOLVER_DIRECT_LLT;
        // the large potentials start now, beside the callback (the contact search: 0.4 ms of small launches and read-backs; a caller's host code)
        if (cb && cb->before_energy_evaluation) eval_prelaunch(c, MISTARK_EVAL_P_G_H, lazy);
        call_void(cb ? cb->before_energy_evaluation : nullptr);
        double residual = 0.0;
        {
            gs.mark(ST_EVAL_PGH);
            eval(c, MISTARK_EVAL_P_G_H, &E0, nullptr, &residual, lazy);
            st.n_evaluations++;
            gs.mark(ST_OTHER);
        }
        if (it == 0) res_0 = residual;
        c.newton_log.push_back(mistark_newton_iteration{});
        c.newton_log.back().residual = residual;
        if (trace) std::fprintf(stderr, "         %d. r0: %.2e | \n", it, residual);
        if (!std::isfinite(residual) || !std::isfinite(E0)) {
            // NaN / inf energy or gradient (k_max_abs turns a NaN entry into +inf): the reference would spin forever here (every comparison
            // below is false and the projection threshold becomes NaN too); report the failure instead so that the time step is halved / the run stops
            result = MISTARK_LINEAR_SYSTEM_SOLVE_FAILURE;
            break;
        }
        if (residual < s.bailout_residual) {
            result = MISTARK_SUCCESSFUL;
            break;
        }
        if (it >= s.min_iterations) {
            if (residual < s.residual_tolerance_abs) {
                result = MISTARK_SUCCESSFUL;
                break;
            }
            if (it > 0 && residual / res_0 < s.residual_tolerance_rel) {
                result = MISTARK_SUCCESSFUL;
                break;
            }
        }

        double du_max_solved = 0.0;
        bool assembled = false;  // "hess == nullptr" in the reference
        bool solved = false;
        while (!solved) {
            // ---- _project_and_assemble ------------------------------------------------------------------------
            bool all_projected = false;
            if (s.projection_mode == MISTARK_PROJ_PROGRESSIVE && !assembled) {
                gs.mark(ST_ASSEMBLY);
                assemble(c);
                assembled = true;
            }
            bool reassemble = !assembled;  // projections performed after assembly update the matrix in place (update_global)
            {
                gs.mark(ST_PROJECT);
                switch (s.projection_mode) {
                    case MISTARK_PROJ_NEWTON: break;
                    case MISTARK_PROJ_PROJECTED_NEWTON: {
                        int64_t np = 0;
                        project(c, s.projection_eps, s.project_to_pd_use_mirroring, nullptr, false, 0.0, nullptr, &np, nullptr);
                        all_projected = true;
                        break;
                    }
                    case MISTARK_PROJ_ON_DEMAND:
                        if (pdn_countdown > 0) {
                            int64_t np = 0;
                            project(c, s.projection_eps, s.project_to_pd_use_mirroring, nullptr, false, 0.0, nullptr, &np, nullptr);
                            all_projected = true;
                        }
                        break;
                    case MISTARK_PROJ_PROGRESSIVE:
                        if (ppn_threshold > 0.0) {
                            if (ppn_threshold < 1e-12) ppn_threshold = 0.0;
                            int all_active = 0;
                            int64_t np = 0;
                            // (the round started beside the solve that has just failed, if this is it: only its last phase is left)
                            if (!project_spec_adopt(c, s.projection_eps, s.project_to_pd_use_mirroring, ppn_threshold, &all_active, &np))
                                project(c, s.projection_eps, s.project_to_pd_use_mirroring, nullptr, true, ppn_threshold, &all_active, &np, nullptr);
                            all_projected = all_active != 0;
                        }
                        break;
                    default: throw Error("unknown projection mode");
                }
            }
            if (reassemble || !c.matrix_current) {
                gs.mark(ST_ASSEMBLY);
                assemble(c);
                assembled = true;
            }

            // ---- _solve_linear_system ---------------------------------------------------------------------------
            mistark_pcg_info info{};
            {
                gs.mark(ST_SOLVE);
                const double forcing = std::min(1e-2, residual * std::min(0.5, std::sqrt(residual)));
                const double abs_tol = std::max(forcing, s.cg_abs_tolerance);
                if (s.linear_solver == MISTARK_SOLVER_DIRECT_LLT) {
                    vec_neg(c, c.tmp_a.p, c.grad.p, ndofs);
                    info.converged = direct_llt(c, c.tmp_a.p, c.du.p) ? 1 : 0;  // NewtonsMethod.cpp:395-418
                } else if (c.world > 1) {
                    vec_neg(c, c.tmp_a.p, c.grad.p, ndofs);
                    pcg(c, c.tmp_a.p, abs_tol, s.cg_rel_tolerance, s.cg_max_iterations, s.cg_stop_on_indefiniteness, &info);
                } else {
                    // Should this solve fail, the retry projects the rows above the NEXT threshold (_increase_projection below): that round's
                    // selection and eigen-projections start now, beside the solve (kernels.hip: project_speculate)
                    if (s.projection_mode == MISTARK_PROJ_PROGRESSIVE && !all_projected) {
                        double next = (ppn_threshold < 0.0 ? residual : ppn_threshold) * s.ppn_tightening_factor;
                        if (next > 0.0) {
                            if (next < 1e-12) next = 0.0;  // (what the retry would pass: see above)
                            if (next > 0.0) project_speculate_request(c, s.projection_eps, s.project_to_pd_use_mirroring, next);
                        }
                    }
                    pcg(c, c.grad.p, abs_tol, s.cg_rel_tolerance, s.cg_max_iterations, s.cg_stop_on_indefiniteness, &info, -1.0);  // A du = -g
                }
                st.cg_iterations += info.n_iterations;
                st.n_linear_solves++;
                c.newton_log.back().linear_solves++;
                c.newton_log.back().cg_iterations_last = info.n_iterations;
                c.newton_log.back().cg_iterations_all += info.n_iterations;
                if (trace) std::fprintf(stderr, "            solve: #CG %5d | tol %.2e | %s%s| ppn threshold %.2e\n", info.n_iterations, abs_tol, info.converged ? "converged " : "",
                                        info.found_indefiniteness ? "indefinite " : "", ppn_threshold);
                gs.mark(ST_OTHER);
            }
            const bool ok = info.converged != 0;
            const bool can_project_more = (s.projection_mode != MISTARK_PROJ_NEWTON) && !all_projected;
            if (!ok && !can_project_more) {
                result = MISTARK_LINEAR_SYSTEM_SOLVE_FAILURE;
                break;
            }
            bool descends = false;
            if (ok) {
                reduce_dot_and_max_abs(c, c.du.p, c.grad.p, ndofs, &du_dot_grad, &du_max_solved);  // (max |du| is needed right after the loop)
                descends = du_dot_grad < 0.0;
                if (!descends && !can_project_more) {
                    result = MISTARK_STEP_DOES_NOT_DESCEND;
                    break;
                }
            }
            if (ok && descends) {
                project_spec_discard(c);  // (the round started beside this solve is not needed)
                solved = true;
                break;
            }
            // _increase_projection
            if (s.projection_mode == MISTARK_PROJ_ON_DEMAND) pdn_countdown = s.project_on_demand_countdown;
            else if (s.projection_mode == MISTARK_PROJ_PROGRESSIVE) {
                if (ppn_threshold < 0.0) ppn_threshold = residual;  // = grad.cwiseAbs().maxCoeff()
                ppn_threshold *= s.ppn_tightening_factor;
            }
        }
        project_spec_discard(c);
        if (result != MISTARK_RUNNING) break;

        // _decrease_projection
        if (s.projection_mode == MISTARK_PROJ_ON_DEMAND) pdn_countdown--;
        else if (s.projection_mode == MISTARK_PROJ_PROGRESSIVE) ppn_threshold *= s.ppn_release_factor;

        st.n_hessians += (int64_t)c.n_elem_total;
        st.n_projected_hessians += c.n_projected_total;
        {
            mistark_newton_iteration& rec = c.newton_log.back();
            rec.logged = 1;
            rec.n_hessians = (int64_t)c.n_elem_total;
            rec.n_projected_hessians = c.n_projected_total;
            rec.du_max = du_max_solved;
        }

        double du_max = du_max_solved;
        if (trace) std::fprintf(stderr, "            du: %.1e | E0: %.10e\n", du_max, E0);
        if (it >= s.min_iterations && du_max < s.step_tolerance) {
            result = MISTARK_SUCCESSFUL;
            break;
        }

        // ---- _line_search_inplace ---------------------------------------------------------------------------------
        const int ls0[4] = {st.ls_cap_iterations, st.ls_max_iterations, st.ls_inv_iterations, st.ls_bt_iterations};
        {
            double* u0 = c.tmp_b.p;  // dofs_before_ls
            copy_async(c.stream, u0, c.u.p, (size_t)ndofs * sizeof(double));
            auto apply = [&](double step) { vec_axpby(c, c.u.p, 1.0, u0, step, c.du.p, ndofs); };
            double retraction = 1.0;
            if (du_max > s.step_cap) {
                retraction *= s.step_cap / du_max;
                vec_axpby(c, c.du.p, retraction, c.du.p, 0.0, nullptr, ndofs);
                du_max = s.step_cap;
                st.ls_cap_iterations++;
            }
            double max_step = 1.0;
            if (cb && cb->max_allowed_step) {
                Timer t(st.t_callbacks);
                sync();
                max_step = std::min(1.0, cb->max_allowed_step(cb->user));
            }
            if (max_step < 1.0) {
                retraction *= max_step;
                vec_axpby(c, c.du.p, max_step, c.du.p, 0.0, nullptr, ndofs);
                du_max *= max_step;
                st.ls_max_iterations++;
            }
            double step = 1.0;
            apply(step);
            int inv_it = 0;
            for (; inv_it < s.max_backtracking_invalid_state_iterations; ++inv_it) {
                if (call_bool(cb ? cb->is_intermediate_state_valid : nullptr, true)) break;
                step *= 0.5;
                apply(step);
                st.ls_inv_iterations++;
            }
            if (inv_it == s.max_backtracking_invalid_state_iterations) {
                call_void(cb ? cb->on_intermediate_state_invalid : nullptr);
                result = MISTARK_TOO_MANY_INVALID_INTERMEDIATE_ITERATIONS;
            } else if (s.enable_armijo_backtracking) {
                const double expected = s.line_search_armijo_beta * du_dot_grad * retraction;
                double E_threshold = E0 + expected * step;
                double E1 = 0.0;
                int k = 0;
                for (; k < s.max_backtracking_armijo_iterations; ++k) {
                    // (Starting the NEXT iteration's gradient / Hessian kernel here as well, on the bet that the candidate is accepted, was
                    // measured: the search chain slows down by what the evaluation gains — the GPU is busy either way.)
                    if (cb && cb->before_energy_evaluation) eval_prelaunch(c, MISTARK_EVAL_P, false);
                    call_void(cb ? cb->before_energy_evaluation : nullptr);
                    {
                        gs.mark(ST_EVAL_P);
                        eval(c, MISTARK_EVAL_P, &E1, nullptr);
                        st.n_evaluations++;
                        gs.mark(ST_OTHER);
                    }
                    if (E1 < E_threshold) break;
                    step *= 0.5;
                    apply(step);
                    E_threshold = E0 + expected * step;
                    st.ls_bt_iterations++;
                }
                if (k == s.max_backtracking_armijo_iterations) {
                    call_void(cb ? cb->on_armijo_fail : nullptr);
                    result = MISTARK_TOO_MANY_ARMIJO_ITERATIONS;
                }
            }
        }
        {
            mistark_newton_iteration& rec = c.newton_log.back();
            rec.line_search = 1;
            rec.ls_cap = st.ls_cap_iterations - ls0[0];
            rec.ls_max = st.ls_max_iterations - ls0[1];
            rec.ls_inv = st.ls_inv_iterations - ls0[2];
            rec.ls_bt = st.ls_bt_iterations - ls0[3];
        }
        // user convergence (NewtonsMethod.cpp:221)
        if (it >= s.min_iterations && call_bool(cb ? cb->is_converged : nullptr, false)) {
            result = MISTARK_SUCCESSFUL;
            break;
        }
    }
    if (result == MISTARK_SUCCESSFUL) {
        if (!call_bool(cb ? cb->is_converged_state_valid : nullptr, true)) result = MISTARK_INVALID_CONVERGED_STATE;
    }
    st.newton_iterations = it;
    if (st.n_hessians > 0) st.projected_hessians_ratio = (double)st.n_projected_hessians / (double)st.n_hessians;
    gs.mark(ST_OTHER);
    sync();
    gs.collect(stage_s);
    st.t_eval_pgh += stage_s[ST_EVAL_PGH];
    st.t_eval_p += stage_s[ST_EVAL_P];
    st.t_assembly += stage_s[ST_ASSEMBLY];
    st.t_project += stage_s[ST_PROJECT];
    st.t_linear_solve += stage_s[ST_SOLVE];
    return result;
}

}  // namespace mistark
