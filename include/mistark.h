/* mistark.h — C ABI of libmistark.so, the MI355X-native engine behind STARK's per-Newton-step hot path.
 *
 * This is the drop-in boundary (SURVEY.md §8b "New C ABI"): plain pointers and sizes, int status returns
 * (0 = ok, otherwise mistark_last_error()), no exceptions / exit() across the boundary, one host thread per context.
 * Host buffers are caller-owned (the reference's DataMap lambdas, symx/src/compile/data_maps.h:97-105); device buffers
 * are library-owned. Each entry point cites the reference interface it replaces.
 *
 * Mapping to the reference's registration API (symx/src/solver/GlobalPotential.h:37-70):
 *   GlobalPotential::add_dof(arr, label)                 -> mistark_add_dof_set
 *   mws.make_scalar / make_vector / make_matrix(arr,..)  -> mistark_array + one mistark_binding per call, in call order
 *   GlobalPotential::add_potential(name, conn, lambda)   -> mistark_potential (name selects the hand-written kernel;
 *                                                           the symbolic lambda is not needed)
 *   GlobalPotential::get_dofs / set_dofs                 -> mistark_get_dofs / mistark_set_dofs
 *   SecondOrderCompiledGlobal::evaluate_*                -> mistark_eval
 *   NewtonsMethod::_project_and_assemble                 -> mistark_assemble / mistark_project
 *   NewtonsMethod::_solve_linear_system (bsm::solve_pcg) -> mistark_pcg
 *   NewtonsMethod::solve                                 -> mistark_newton_solve
 */
#ifndef MISTARK_H
#define MISTARK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mistark_ctx mistark_ctx;

/* ---- context -------------------------------------------------------------------------------------------------- */
int mistark_create(int device, mistark_ctx** out);
void mistark_destroy(mistark_ctx* ctx);
/* A registration-only context: no GPU is touched (process-wide from the first call on), DoF sets / arrays / potentials are recorded,
 * every evaluation entry point fails. For checking on a machine without a GPU what a caller registers (mistark_describe). */
int mistark_create_dry(mistark_ctx** out);
/* The registration as JSON (DoF sets; per potential: name, connectivity stride, element count, bindings as [array, stride, connectivity
 * column, items] with arrays named "dof:<label>" or "a<k>" in order of first use). Returns the length needed including the terminator;
 * writes at most cap bytes. */
int64_t mistark_describe(mistark_ctx* ctx, char* buf, int64_t cap);
const char* mistark_last_error(mistark_ctx* ctx);
/* Version / build info string (static storage). */
const char* mistark_version(void);

/* ---- DoF sets and bound arrays ---------------------------------------------------------------------------------- */
/* GlobalPotential::add_dof (GlobalPotential.h:55-61): `host` holds n_scalars doubles (3 per DoF block, AoS as
 * PointDynamics::v1, stark/src/models/deformables/PointDynamics.h:16-23). Sets are concatenated in registration order.
 * Returns the set index (>= 0) or < 0 on error. The size may be changed later with mistark_resize_dof_set. */
int mistark_add_dof_set(mistark_ctx* ctx, const char* label, double* host, int64_t n_scalars);
int mistark_resize_dof_set(mistark_ctx* ctx, int set, double* host, int64_t n_scalars);

/* A bound input array (DataMap, data_maps.h:97-121): n_items x stride doubles, row-major, caller-owned.
 * Arrays are identified by (host pointer, stride); binding the host pointer of a DoF set yields a view of the DoF
 * vector. Returns the array id (>= 0). Re-binding an existing (pointer, stride) updates n_items. */
int mistark_array(mistark_ctx* ctx, const double* host, int64_t n_items, int stride);
/* The view of DoF set `set` as an array of the given stride, by set index instead of by host address (an EMPTY set has no address to be
 * recognised by: SymX identifies DoF maps by container identity, DataMap::id, data_maps.h:97-105). Returns the array id. */
int mistark_dof_array(mistark_ctx* ctx, int set, int stride);
/* Re-point an array id at a (possibly reallocated / resized) host buffer. */
int mistark_array_rebind(mistark_ctx* ctx, int array, const double* host, int64_t n_items);
/* host -> device / device -> host copies of one array (all arrays if array < 0). */
int mistark_upload(mistark_ctx* ctx, int array);
int mistark_download(mistark_ctx* ctx, int array);
/* Device-side vector helpers used by the state containers (PointDynamics.cpp:58-78): dst = a*x + b*y (x,y,dst array
 * ids of equal size; y may be -1), fill. */
int mistark_array_axpby(mistark_ctx* ctx, int dst, double a, int x, double b, int y);
int mistark_array_fill(mistark_ctx* ctx, int dst, double value);

/* ---- potentials ----------------------------------------------------------------------------------------------------- */
typedef struct mistark_binding
{
    int32_t array;     /* id from mistark_array */
    int32_t stride;    /* doubles per item (must match the kernel's expectation) */
    int32_t conn_col;  /* connectivity column that indexes the array, -1 = global value (item 0) */
} mistark_binding;

/* GlobalPotential::add_potential(name, LabelledConnectivity<N>&, ...) (GlobalPotential.h:37-52). `name` is the
 * reference's registry key (e.g. "EnergyTetStrain"); bindings are given in the order of the reference's mws.make_*
 * calls for that potential. conn: n_elem x conn_stride int32, row-major, copied at call time.
 * Returns the potential id (>= 0). Calling again with an existing name replaces connectivity and bindings. */
int mistark_potential(mistark_ctx* ctx, const char* name, const int32_t* conn, int32_t n_elem, int32_t conn_stride,
                      const mistark_binding* bindings, int32_t n_bindings);

/* A potential WITHOUT a hand-written kernel (user-defined energies, README.md:109-126 of the reference): the energy is handed over as
 * SymX's own straight-line op sequence and interpreted on the device (slow path; see csrc/custom.hip). The reference-side shim builds
 * it with `symx::Sequence seq({potential.get_expression()})` (symx/src/compile/Sequence.h:24-41) and copies `seq.ops`:
 *   ops       n_ops rows {type, dst, a, b, cond} with symx::ExprType codes (symx/src/symbol/Expr.h:12-43) and the value numbering of
 *             symx/src/compile/Compilation.cpp:381-469: values < n_inputs are the gathered inputs (the bindings flattened in order),
 *             a Symbol op binds output 0 (the energy), Branch ops are the if (value > 0) / else / endif markers
 *   constants one double per op (used by ConstantFloat ops)
 *   cond_*    optional second sequence: the element is active iff its value is > 0 (Potential::get_condition, conditional potentials)
 * Bindings on DoF arrays define the local DoF blocks exactly as for mistark_potential. Limits: 96 inputs, 256 live temporaries. */
int mistark_potential_custom(mistark_ctx* ctx, const char* name, const int32_t* conn, int32_t n_elem, int32_t conn_stride, const mistark_binding* bindings, int32_t n_bindings,
                             const int32_t* ops, const double* constants, int32_t n_ops, int32_t n_inputs, const int32_t* cond_ops, const double* cond_constants,
                             int32_t n_cond_ops);
/* A custom potential does not stay interpreted: at its first evaluation the op sequence is EMITTED as HIP source (one statement per op, the
 * temporaries local variables, Branch markers real branches, the gather unrolled — the mirror of the reference's scalar emitter,
 * symx/src/compile/Compilation.cpp:381-469), compiled for gfx950 by hipRTC and cached on disk by a hash of the source (MISTARK_RTC_CACHE,
 * default /tmp/mistark_rtc_cache_<uid>). The interpreter remains the fallback (no libhiprtc, a failed build, sequences of more than
 * MISTARK_RTC_MAX_OPS = 3000 ops, option "custom_rtc" = 0). mistark_custom_emit returns what the emitter writes for a sequence — the source
 * in `out` (truncated to cap) and its length — and, with compile != 0, runs hipRTC on it and returns the size of the code object; < 0 with the
 * message in `out` on failure. in_dof[k]: the local DoF component (3 * block + c) input k seeds, -1 for inputs that are not DoFs. Needs no
 * context and no GPU. */
int64_t mistark_custom_emit(const char* name, const int32_t* strides, int32_t n_bindings, const int32_t* in_dof, const int32_t* ops, const double* constants, int32_t n_ops,
                            int32_t n_inputs, const int32_t* cond_ops, const double* cond_constants, int32_t n_cond_ops, int32_t n_blocks, int32_t compile, char* out, int64_t cap);
/* Summation loop of a custom potential (MappedWorkspace::add_for_each, symx/src/compile/MappedWorkspace.h:123-130; used by SymX's fem
 * integrators for quadrature rules): inputs [first_input, first_input + stride) — covered by a binding like any other input — take the
 * n_iterations rows of `data` (copied) one after the other, and the element's energy, gradient and Hessian are the sums over the rows in
 * that order (CompiledInLoop_run.h:375-400). One summation per potential, as in the reference (MappedWorkspace.h:525-530). */
int mistark_potential_custom_set_summation(mistark_ctx* ctx, int potential, int32_t first_input, int32_t stride, int32_t n_iterations, const double* data);
/* Marks a potential whose connectivity changes inside the Newton loop (the reference's contact tables are refilled in
 * before_energy_evaluation, EnergyFrictionalContact.cpp:117-119). Its Hessian blocks go to a second, small block-CSR part
 * (A = A_static + A_dynamic) so that a connectivity update only re-patterns that part, not the whole matrix. */
/* Introspection of a registration (what MappedWorkspace holds for a potential, MappedWorkspace.h:329-333): the connectivity table as
 * registered (conn may be NULL to query the sizes; device-side contact tables: mistark_contact_get_table), and the caller's array behind
 * binding `binding` (pointer, items, stride). */
int mistark_potential_table(mistark_ctx* ctx, int potential, int32_t* conn, int64_t* n_elem, int32_t* conn_stride);
int mistark_potential_binding_data(mistark_ctx* ctx, int potential, int binding, const double** host, int64_t* n_items, int32_t* stride);
int mistark_potential_set_dynamic(mistark_ctx* ctx, int potential, int dynamic);
/* Id of the first potential registered under `name` (the ids mistark_potential returned, in registration order), -1 if there is none. */
int mistark_find_potential(mistark_ctx* ctx, const char* name);
/* LabelledConnectivity::clear() + push_back() (symx/src/compile/LabelledConnectivity.h): replaces the rows of a potential. */
int mistark_potential_update_connectivity(mistark_ctx* ctx, int potential, const int32_t* conn, int32_t n_elem);
/* Number of potentials known to the engine and their registry names (for the shim's "unknown name" error path). */
int mistark_n_supported_potentials(void);
const char* mistark_supported_potential(int i);

/* ---- DoF vector --------------------------------------------------------------------------------------------------- */
int64_t mistark_ndofs(mistark_ctx* ctx);
int mistark_get_dofs(mistark_ctx* ctx, double* u_host);        /* GlobalPotential::get_dofs (GlobalPotential.cpp:111-121) */
int mistark_set_dofs(mistark_ctx* ctx, const double* u_host);  /* GlobalPotential::set_dofs (GlobalPotential.cpp:134-144) */
/* device DoFs -> the host arrays registered with mistark_add_dof_set, and back */
int mistark_dofs_to_host_arrays(mistark_ctx* ctx);
/* The same, skipped when the DoF vector on the device has not changed since the last transfer to the caller's arrays (a drop-in's Newton
 * callbacks all read the DoFs from the caller's arrays; several of them run at the same iterate: intersection check, contact update,
 * convergence test). The caller must not have written its DoF arrays in between (mistark_dofs_from_host_arrays / mistark_set_dofs say so). */
int mistark_dofs_to_host_arrays_if_changed(mistark_ctx* ctx);
int mistark_dofs_from_host_arrays(mistark_ctx* ctx);

/* ---- evaluation ----------------------------------------------------------------------------------------------------- */
enum { MISTARK_EVAL_P = 0, MISTARK_EVAL_P_G = 1, MISTARK_EVAL_P_G_H = 2 };
/* SecondOrderCompiledGlobal::evaluate_P / _P__dP_du / _P__dP_du__local_d2P_du2 (SecondOrderCompiledGlobal.cpp:72-142).
 * E: total energy; grad_host (may be NULL): ndofs doubles. Element Hessians stay on the device. */
int mistark_eval(mistark_ctx* ctx, int mode, double* E, double* grad_host);
/* Parity access: element Hessians of one potential, n_elem x (3 NB) x (3 NB) row-major doubles, and the global block
 * rows n_elem x NB (ElementHessians::HessianView, ElementHessians.h:33-39). Either pointer may be NULL. */
int mistark_get_element_hessians(mistark_ctx* ctx, int potential, double* values, int32_t* block_rows, int32_t* nb_out);
int mistark_get_element_energies(mistark_ctx* ctx, int potential, double* values);

/* ---- projection + assembly ------------------------------------------------------------------------------------------ */
/* ElementHessians::project_to_PD_inplace__all / project_to_PD_for_update__selectively (ElementHessians.cpp:48-67,79-182;
 * project_to_PD.cpp:12-32). active_blocks: NULL = all elements, else ndofs/3 flags; only not-yet-projected elements
 * touching an active block are projected. Returns counts through the out-pointers (may be NULL). */
int mistark_project(mistark_ctx* ctx, double eps, int mirroring, const uint8_t* active_blocks, int64_t* n_projected_now,
                    int64_t* n_changed_now);
/* Same selection rule evaluated on the device from the current gradient: block active iff max|grad_block| >= threshold
 * (NewtonsMethod.cpp:314-323). all_active_out: 1 if every block was active. */
int mistark_project_by_gradient(mistark_ctx* ctx, double eps, int mirroring, double threshold, int* all_active_out,
                                int64_t* n_projected_now);
/* ElementHessians::assemble_global (ElementHessians.cpp:224-256): element blocks -> float 3x3-block CSR. */
int mistark_assemble(mistark_ctx* ctx);
/* Parity access to the assembled matrix (BlockedSparseMatrix::to_triplets): block CSR, vals row-major 3x3 floats. */
int mistark_get_bsr(mistark_ctx* ctx, int64_t* n_block_rows, int64_t* nnzb, int64_t* row_ptr, int32_t* cols, float* vals);
/* Probes: y = A x (BlockedSparseMatrix::spmxv_from_ptr), z = M^-1 x (prepare/apply_preconditioning). Host vectors. */
int mistark_spmv(mistark_ctx* ctx, const double* x_host, double* y_host);
int mistark_apply_preconditioner(mistark_ctx* ctx, const double* x_host, double* z_host);

/* ---- linear solve --------------------------------------------------------------------------------------------------- */
typedef struct mistark_pcg_info
{
    int32_t converged;
    int32_t n_iterations;
    int32_t found_indefiniteness;
    int32_t reserved;
    double error;
} mistark_pcg_info;
/* bsm::solve_pcg (BlockedSparseMatrix/solve_pcg.h:83-232) with x0 = 0 and rhs = -grad of the last evaluation
 * (NewtonsMethod.cpp:392,431-446). The solution stays on the device as the Newton step `du`; du_host may be NULL. */
int mistark_pcg(mistark_ctx* ctx, double abs_tol, double rel_tol, int max_iter, int stop_on_indefiniteness, double* du_host,
                mistark_pcg_info* info);
/* Same with an explicit host rhs (parity tests). */
int mistark_pcg_rhs(mistark_ctx* ctx, const double* rhs_host, double abs_tol, double rel_tol, int max_iter,
                    int stop_on_indefiniteness, double* x_host, mistark_pcg_info* info);
/* symx::LinearSolver::DirectLLT as a staged call (NewtonsMethod.cpp:395-418: Eigen::SimplicialLLT of the assembled matrix): x = A^-1 rhs by a
 * Cholesky factorisation in double on the device — dense up to 3072 unknowns, block-tridiagonal on a reverse Cuthill-McKee ordering while the
 * band is cheap, multifrontal on a nested-dissection ordering beyond (option "llt_multifrontal": 1 = always, -1 = never). *success = 0 when a
 * pivot is not positive (the reference's "solve failed"). */
int mistark_direct_llt_rhs(mistark_ctx* ctx, const double* rhs_host, double* x_host, int* success);

/* ---- Newton's method -------------------------------------------------------------------------------------------------- */
/* symx::SolverReturn (symx/src/solver/solver_utils.h:15-26) */
enum
{
    MISTARK_SUCCESSFUL = 0,
    MISTARK_RUNNING = 1,
    MISTARK_INVALID_INITIAL_STATE = 2,
    MISTARK_TOO_MANY_ITERATIONS = 3,
    MISTARK_TOO_MANY_ARMIJO_ITERATIONS = 4,
    MISTARK_LINEAR_SYSTEM_SOLVE_FAILURE = 5,
    MISTARK_TOO_MANY_INVALID_INTERMEDIATE_ITERATIONS = 6,
    MISTARK_STEP_DOES_NOT_DESCEND = 7,
    MISTARK_INVALID_CONVERGED_STATE = 8
};
/* symx::ProjectionToPD (solver_utils.h:137-143) */
enum { MISTARK_PROJ_NEWTON = 0, MISTARK_PROJ_PROJECTED_NEWTON = 1, MISTARK_PROJ_ON_DEMAND = 2, MISTARK_PROJ_PROGRESSIVE = 3 };

/* symx::NewtonSettings (solver_utils.h:173-259) with STARK's overrides as defaults (stark/src/core/Settings.cpp:43-50) */
/* symx::LinearSolver. DirectLLT (NewtonsMethod.cpp:395-418) is a dense Cholesky for small systems (<= 3072 unknowns); larger ones are an error. */
enum
{
    MISTARK_SOLVER_BDPCG = 0,
    MISTARK_SOLVER_DIRECT_LLT = 1
};
typedef struct mistark_newton_settings
{
    int32_t max_iterations;
    int32_t min_iterations;
    double residual_tolerance_abs;
    double residual_tolerance_rel;
    double step_tolerance;
    int32_t max_iterations_as_success;
    double step_cap;
    int32_t enable_armijo_backtracking;
    double line_search_armijo_beta;
    int32_t max_backtracking_armijo_iterations;
    int32_t max_backtracking_invalid_state_iterations;
    int32_t projection_mode;
    double projection_eps;
    int32_t project_to_pd_use_mirroring;
    int32_t project_on_demand_countdown;
    double ppn_tightening_factor;
    double ppn_release_factor;
    int32_t cg_max_iterations;
    double cg_abs_tolerance;
    double cg_rel_tolerance;
    int32_t cg_stop_on_indefiniteness;
    double bailout_residual;
    int32_t linear_solver; /* symx::LinearSolver (solver_utils.h): MISTARK_SOLVER_BDPCG (default) or MISTARK_SOLVER_DIRECT_LLT */
} mistark_newton_settings;
void mistark_newton_default_settings(mistark_newton_settings* s);

/* NewtonsMethod::SolveStats (NewtonsMethod.h:32-43) + wall-clock per stage (the reference's Logger timers) */
typedef struct mistark_newton_stats
{
    int32_t newton_iterations;
    int32_t cg_iterations;
    int32_t ls_cap_iterations;
    int32_t ls_max_iterations;
    int32_t ls_inv_iterations;
    int32_t ls_bt_iterations;
    int64_t n_hessians;
    int64_t n_projected_hessians;
    double projected_hessians_ratio;
    int32_t n_linear_solves;
    int32_t n_evaluations;
    double t_eval_pgh;
    double t_eval_p;
    double t_project;
    double t_assembly;
    double t_linear_solve;
    double t_callbacks;
    double t_total;
} mistark_newton_stats;

/* One record per Newton iteration of the last mistark_newton_solve: what the reference appends to its Logger series inside the loop
 * (NewtonsMethod.cpp:198-207 "n_hessians", "n_projected_hessians", "cg_iterations" = the iterations of the LAST linear solve of the
 * iteration, :488-594 "ls_cap", "ls_max", "ls_inv", "ls_bt") and prints at Verbosity::Full (r0, du). `logged` = the iteration got past
 * its linear solves (the first group of series has an entry), `line_search` = its line search ran (the ls_* series have one). */
typedef struct mistark_newton_iteration
{
    double residual;             /* r0 at the iteration's evaluation */
    double du_max;               /* max |du| of the accepted solve */
    int32_t linear_solves;       /* solves of this iteration (progressive projection retries included) */
    int32_t cg_iterations_last;  /* the reference's per-iteration "cg_iterations" */
    int32_t cg_iterations_all;   /* over all solves of the iteration */
    int32_t logged;
    int64_t n_hessians;
    int64_t n_projected_hessians;
    int32_t line_search;
    int32_t ls_cap, ls_max, ls_inv, ls_bt;
    int32_t reserved;
} mistark_newton_iteration;
/* Copies up to `cap` records into `out`; *n = number of records the last solve produced. */
int mistark_newton_iteration_log(mistark_ctx* ctx, mistark_newton_iteration* out, int32_t cap, int32_t* n);

/* symx::SolverCallbacks (solver_utils.h:29-117). Any pointer may be NULL. Callbacks run on the calling thread; they may
 * call mistark_* functions on the same context (e.g. download positions, update contact connectivity). */
typedef struct mistark_newton_callbacks
{
    void* user;
    void (*before_energy_evaluation)(void* user);
    int (*is_initial_state_valid)(void* user);
    int (*is_intermediate_state_valid)(void* user);
    void (*on_intermediate_state_invalid)(void* user);
    void (*on_armijo_fail)(void* user);
    int (*is_converged)(void* user);
    int (*is_converged_state_valid)(void* user);
    double (*max_allowed_step)(void* user);
} mistark_newton_callbacks;

/* NewtonsMethod::solve (NewtonsMethod.cpp:28-252). Returns a MISTARK_* SolverReturn code (>= 0) or < 0 on engine error. */
int mistark_newton_solve(mistark_ctx* ctx, const mistark_newton_settings* settings, const mistark_newton_callbacks* callbacks,
                         mistark_newton_stats* stats);

/* ---- options --------------------------------------------------------------------------------------------------------- */
/* Engine switches (all default 0): "force_generic" = evaluate every potential through the generic hyper-dual kernels (the
 * closed-form kernels are then cross-checked against them; "generic_contact" = the same for the contact and friction potentials only;
 * "contact_closed_min_lanes" = N: closed-form contact kernels for tables with at least N (contact, DoF pair) lanes, 0 = always, default
 * -1 = by potential, see launch_eval);
 * "generic_inertia" = EnergyLumpedInertia through the generic kernel (six lanes per node; default: one lane per node, the same bits);
 * "pin_host_arrays" = page-lock the caller's large DoF and bound arrays where they are (hipHostRegister, checked; released when an array is
 * rebound or resized and with the context; a range that cannot be locked stays pageable) so that transfers to and from them are direct DMA;
 * "atomic_assembly" = scatter assembly with float atomics
 * instead of the deterministic gather; "proj_variant" = PSD projection cross-checks, bits: 1 = eigen-decomposition with the
 * matrix in LDS instead of registers, 2 = one launch per potential instead of one for all short lists, 4 = IEEE division /
 * square root for the rotation angles, 8 = the full-size matrix for translation-invariant elements too (default: their reduced
 * matrix); "spmv_chunk_tiles" = tiles per SpMV chunk (0 = by matrix size); "no_contact_cache" = run every contact search in full
 * (no answer from the installed tables, no shared box list, count read back before the sort); "lazy_hessians" = 0: the Newton loop
 * keeps the double element-Hessian pool (default 1: float upper triangles, doubles recomputed for projected elements);
 * "lazy_eval" = staged mistark_eval calls take the lazy path too; "no_grad_gather" = gradient by atomics in arrival order instead of
 * the per-potential pools summed in list order;
 * "no_multi_eval_p" = one launch per potential in energy-only evaluations (default: the small potentials share one launch);
 * "no_eval_prelaunch" = do not start the large potentials' kernels ahead of the callback that precedes an evaluation;
 * "no_pattern_overlap" / "no_eval_overlap" / "no_bounded_pattern" = switch off, one by one, the side stream for the contact part's
 * pattern, the auxiliary stream for small potentials, the device-side counts of the pattern build; "fuse_dir" = direction update
 * inside the SpMV (measured slower, a cross-check); "kernel_dbg" = measurement switches inside kernels.
 * Returns 0, or < 0 for an unknown name. The environment variable
 * MISTARK_OPTIONS="name=value,name=value" applies the same switches inside mistark_create (for a process that cannot be
 * edited: a test suite, a profiler run); a bad entry makes mistark_create fail with -6. MISTARK_POISON=1 fills every fresh
 * device allocation with a NaN pattern (finds reads of memory nobody wrote; the GPU test suite passes with it).
 * MISTARK_NEWTON_TRACE=1 prints one line per Newton iteration and linear solve on stderr (residual, CG iterations, tolerance,
 * converged / indefinite, projection threshold, max |du|): what the reference prints at symx::Verbosity::Full. */
int mistark_set_option(mistark_ctx* ctx, const char* name, int value);

/* ---- timers ------------------------------------------------------------------------------------------------------- */
/* Average duration in ms of the SpMV kernel over the launches since the last reset, measured with HIP events on the
 * engine's stream; n = number of launches measured. Used by bench.py for the roofline figure. */
int mistark_spmv_timing(mistark_ctx* ctx, int reset, double* avg_ms, int64_t* n, double* bytes_per_launch);
/* Average duration in ms of the EMPTY event brackets recorded right behind the timed launches: what a pair of event records costs the
 * stream by itself (call before the reset of mistark_spmv_timing). */
int mistark_spmv_event_overhead(mistark_ctx* ctx, double* avg_ms);
/* The same sampled launches on the device's constant clock: every workgroup records when it started and finished, the host takes
 * max(end) - min(start) — the launch's execution time inside the solver loop, which is what rocprofv3's kernel trace reports, without
 * the dispatch latency and marker packets an event bracket contains (call before the reset of mistark_spmv_timing). */
int mistark_spmv_device_clock(mistark_ctx* ctx, double* avg_ms, int64_t* n);
/* Micro-benchmark: n back-to-back SpMV launches on the currently assembled matrix, average duration in microseconds. */
int mistark_spmv_bench(mistark_ctx* ctx, int n_launches, double* avg_us);
/* Waits until everything queued on the engine's stream has finished (entry points that return values already do; assemble / project /
 * axpby only enqueue). For timing from the host. */
int mistark_sync(mistark_ctx* ctx);
/* Event counters of the context, by name (tests assert that a feature under test actually ran): "proj_speculated" / "proj_adopted" (projection
 * rounds started beside a solve / taken over by the retry, option proj_speculation), "dof_skips_verified" (MISTARK_VERIFY_DOF_SKIP=1: DoF
 * transfers skipped at an unchanged iterate and checked against a real transfer), "fused_solves" / "unfused_solves" (sharded PCG),
 * "rtc_builds" / "rtc_launches" / "rtc_build_ms" (user-defined potentials: kernels emitted and compiled by hipRTC, their launches, build time),
 * "multi_pgh_launches" (evaluations whose contact / friction tables shared one launch; option no_multi_eval_pgh = 1: one launch per table),
 * "contact_searches" / "contact_repeated_searches" (barrier-table searches that ran on the device / that ran at the state the previous one had
 * searched: 0 unless the option no_contact_cache is set), "eval_pgh_issue_us" / "eval_pgh_wait_us" (host microseconds of the P+g+H evaluations:
 * issuing their launches / waiting for their read-backs). */
int mistark_get_counter(mistark_ctx* ctx, const char* name, int64_t* out);

/* ---- multi-GPU: one problem sharded over `world` ranks, one engine context (and one process) per GPU ------------------------
 * Block rows are partitioned over the ranks (owner map: mistark_dist_set_row_owner, or a graph partition of the potentials' connectivity).
 * A rank evaluates every element that touches one of its rows (interface elements are evaluated by both sides: no gradient / Hessian
 * traffic), assembles and solves ITS rows of the system (block-Jacobi PCG on a row-sharded matrix: the ghosts of the search direction
 * from their owners and the three dot products of solve_pcg.h:180,201,217 are exchanged in every iteration, (r.r, r.z) fused), and
 * holds the whole state (DoFs, bound arrays, contact tables), so every rank runs the same host code and takes the same decisions.
 * All exchanges are all-gathers on the engine's stream: stores into IPC windows of the peers (below), ncclAllGather over xGMI (RCCL), or a copy
 * kernel for ranks inside one process.
 * Reductions are done by every rank in rank order: identical bits everywhere. Call right after mistark_create, before the first
 * evaluation. With N ranks mistark_get_bsr / mistark_apply_preconditioner are not available (single-rank accessors). */
int mistark_shard_range(int64_t n, int rank, int world, int64_t* begin, int64_t* end); /* [n*rank/world, n*(rank+1)/world) */
/* The built-in graph partition as a host function (no GPU, no context): tables of elements given by their block rows (rows[t][e * nb[t] + k]),
 * hub[r] != 0 marks rows kept out of the graph (given to the last rank; NULL: none). Breadth-first order from a pseudo-peripheral row, cut
 * into `world` pieces of equal element incidence. */
int mistark_partition_rows(int64_t n_block_rows, int world, int n_tables, const int32_t* const* rows, const int64_t* n_elem, const int32_t* nb, const uint8_t* hub,
                           int32_t* owner_out);
int mistark_dist_unique_id(char out[128]);  /* rank 0: ncclGetUniqueId, to be broadcast by the launcher (e.g. torch.distributed) */
int mistark_dist_init_rccl(mistark_ctx* ctx, int rank, int world, const char unique_id[128]);
/* Moves the communicator (rank, world, transport) of `from` to `ctx`; `from` becomes a single-rank context (a scene that registers
 * again keeps its communicator: an RCCL unique id is single-use). */
int mistark_dist_move(mistark_ctx* ctx, mistark_ctx* from);
/* one-rank RCCL round trip (all-gather of `inout`) through the dlopen'ed entry points the N-rank path uses */
int mistark_dist_rccl_selftest(mistark_ctx* ctx, double* inout, int64_t n);
/* Explicit partition: owner[r] in [0, world) for every block row r of the flat DoF vector (e.g. slabs along the longest axis from the
 * scene's positions). owner == NULL returns to the built-in graph partition. */
int mistark_dist_set_row_owner(mistark_ctx* ctx, const int32_t* owner, int64_t n_block_rows);
int64_t mistark_dof_set_first_row(mistark_ctx* ctx, int set);  /* first block row of a DoF set in the flat DoF vector (< 0: no such set) */
/* A position per block row (xyz[3 r ..], NaN for rows without one: rigid bodies keep the last rank): the rows are partitioned by recursive
 * coordinate bisection, weighted by element incidences — box-like parts with few interface rows. The host mirror passes the rest positions
 * of its point sets. An explicit owner map takes precedence; without either the built-in graph partition is used. */
int mistark_dist_set_row_coords(mistark_ctx* ctx, const double* xyz, int64_t n_block_rows);
/* Block rows that potentials with device-side connectivity may reference on any rank; every rank keeps them as ghosts. The contact system
 * registers the collision vertices of its deformable meshes itself; small DoF sets (rigid bodies) are always shared. */
int mistark_dist_add_shared_rows(mistark_ctx* ctx, const int32_t* rows, int64_t n);
/* out[0..15): block rows owned by this rank, ghosts, rows it sends, elements it evaluates, blocks of its static / contact matrix part, linear
 * solves that took the fused iteration (one exposed exchange, ranks exchanging through windows) and the five-launch one; [8] world and [9] rank
 * of the context, [10] transport (0 none, 1 in-process group, 2 RCCL, 3 IPC windows), [11] ranks the transport itself counts (RCCL:
 * ncclCommCount of the communicator), [12] contact searches whose sweep was dealt out to the ranks (keys all-gathered and merged), [13] elements
 * of all potentials with fixed connectivity as registered (the unsharded problem) and [14] those of them this rank evaluates */
int mistark_dist_info(mistark_ctx* ctx, int64_t* out, int n);
int mistark_dist_get_row_owner(mistark_ctx* ctx, int32_t* owner);
/* ---- IPC windows: one process per rank, no library in the data path -----------------------------------------------------------------
 * Every rank owns a window of device memory that the other ranks map with hipIpcOpenMemHandle (the peers' GPUs through the xGMI aperture,
 * or the same GPU when several ranks share one device — how a one-GPU box runs the real multi-process launch path). An exchange is: the
 * producing kernel stores its values into every rank's window as 8-byte {sequence tag, 32 data bits} granules (system-scope write-through
 * stores), the consuming kernel polls its own window until the granules carry the tag. No host involvement, no library launch, no separate
 * flag: one exchange costs the one-way store latency. Waits are bounded (MISTARK_IPC_TIMEOUT_S, default 30 s): a peer that never delivers
 * turns into an error at the host's next synchronisation point instead of a hung GPU.
 *   create (allocates and zeroes the window; `handle_out` = its 64-byte hipIpcMemHandle_t)  ->  the launcher all-gathers the handles (bench.py:
 *   torch.distributed / gloo)  ->  connect (handles of all ranks in rank order, n_bytes = world * 64)  ->  mistark_dist_init_ipc(ctx, comm).
 * window_bytes: at least 48 bytes per DoF of the largest problem (smaller messages than a slot travel in one piece, longer ones in several);
 * 0 = the minimum. The communicator must outlive the contexts using it. */
typedef struct mistark_ipc_comm mistark_ipc_comm;
mistark_ipc_comm* mistark_ipc_comm_create(int device, int rank, int world, int64_t window_bytes, char handle_out[64]);
int mistark_ipc_comm_connect(mistark_ipc_comm* comm, const char* handles, int64_t n_bytes);
const char* mistark_ipc_comm_last_error(mistark_ipc_comm* comm);
void mistark_ipc_comm_destroy(mistark_ipc_comm* comm);
int mistark_dist_init_ipc(mistark_ctx* ctx, mistark_ipc_comm* comm);
/* `iters` all-gathers of n doubles with predictable values, every received value checked; avg_us[0] = wall time of one exchange + stream
 * synchronisation, avg_us[1] = of one exchange in a train enqueued back to back. Collective: every rank calls it with the same arguments. */
int mistark_ipc_comm_selftest(mistark_ipc_comm* comm, int64_t n, int iters, double avg_us[2]);
/* Pre-flight of the windows: one tagged granule over every ordered pair of ranks (ping-pong, `iters` >= 2 exchanges per pair, the first of a
 * pair discarded), every wait bounded by timeout_s (clamped to 2 s). half_rtt_us[p] = half the best round trip with peer p in microseconds
 * (0 for the own rank, < 0 where no granule came back). Returns the number of peers that answered (world - 1 = the windows deliver), < 0 on
 * error. Collective: every rank calls it with the same arguments, a host barrier in front (the kernels must start within the time-out of each
 * other). A launcher that sees fewer than world - 1 on any rank falls back to RCCL and says so (bench.py). */
int mistark_ipc_comm_preflight(mistark_ipc_comm* comm, int iters, double timeout_s, double* half_rtt_us /* world */);
/* The two all-reduces the reference's parallel reductions turn into on N GPUs, on RCCL whatever transport the engine runs on: ncclAllReduce
 * (f64, sum) of n_big doubles (the gradient of the shared DoFs: symx SecondOrderCompiledGlobal.cpp:72-142) and of 3 doubles (the dot products of
 * a CG iteration: BlockedSparseMatrix/solve_pcg.h:180,201,217), `reps` launches each back to back between HIP events, results checked.
 * out[0] = ncclCommCount, out[1] = microseconds per all-reduce of n_big doubles, out[2] = of 3 doubles, out[3] = seconds in ncclCommInitRank.
 * Collective; one rank per device (RCCL refuses two ranks on one device: the error text lands in err). No engine context involved. */
int mistark_rccl_allreduce_bench(int device, int rank, int world, const char unique_id[128], int64_t n_big, int reps, double* out /* 4 */, char* err, int err_len);
/* Measurement: solo durations (microseconds) of the two kernels of the fused PCG iteration on THIS rank's rows — out[0] = the SpMV with its
 * halo polls, out[1] = the vector kernel (whose workgroup 0 reduces and pushes the rank's three sums) — replayed from the rank's
 * last converged solve on the messages still in its window (every poll answered at once). Not a collective: the caller lets the ranks take
 * turns (a host barrier between them), so the figures are kernels running alone even when all ranks share one GPU; no rank may start a solve
 * in between. Needs ranks that exchange through windows and a converged solve on the current matrix. */
int mistark_dist_fused_bench(mistark_ctx* ctx, int n_launches, double* out);
/* the same sharded path with several contexts inside one process (one host thread per context, one device, one shared stream), used by
 * the single-GPU tests */
typedef struct mistark_local_group mistark_local_group;
mistark_local_group* mistark_local_group_create(int world);
void mistark_local_group_destroy(mistark_local_group* group);
int mistark_dist_init_local(mistark_ctx* ctx, mistark_local_group* group, int rank);

#ifdef __cplusplus
}
#endif
#endif
