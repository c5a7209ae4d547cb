// symx::NewtonsMethod on libmistark — replaces symx/src/solver/NewtonsMethod.h:28-97 (same name, same public members) in a STARK build.
//
//   NewtonsMethod::create(global_potential, context, callbacks)   Stark.cpp:295  -> nothing is compiled; the first solve() registers
//   newton->settings = settings.newton                            Stark.cpp:296  -> translated into mistark_newton_settings per solve
//   newton->solve()                                               Stark.cpp:158  -> register / refresh, mistark_newton_solve, DoFs back
//   newton->get_last_solve_stats()                                Stark.cpp:176
//   newton->print_summary()                                       Stark.cpp:278
// Registration walks GlobalPotential::get_potentials(): name -> kernel, MappedWorkspace::conn -> connectivity table, MappedWorkspace::maps
// (the mws.make_* calls in order: which array, stride, connectivity column; MappedWorkspace.h:287-291,329-333) -> bindings; the symbolic
// expression of a recognised potential is never evaluated. A name without a hand-written kernel is handed over as SymX's own op
// sequence (mistark_potential_custom). Host arrays stay caller-owned: sizes and pointers are re-read at every solve and, for the
// potentials whose tables change inside the Newton loop (contacts), after every before_energy_evaluation callback.
#pragma once
#include <memory>

#include <solver/GlobalPotential.h>
#include <solver/solver_utils.h>

struct mistark_ctx;

namespace symx
{
	// Public surface = what stark/src and user code touch of the reference's class: SolveStats (NewtonsMethod.h:32-43), the two public fields,
	// create / solve / get_last_solve_stats / print_summary (:79-87). Everything behind it lives in shim/src/NewtonsMethod.cpp.
	class NewtonsMethod
	{
	public:
		struct SolveStats
		{
			// (names and types are the interface: Stark.cpp:176-207 and the summary printer read them)
			int newton_iterations = 0, cg_iterations = 0;
			int ls_cap_iterations = 0, ls_max_iterations = 0, ls_inv_iterations = 0, ls_bt_iterations = 0;
			uint64_t n_hessians = 0, n_projected_hessians = 0;
			double projected_hessians_ratio = 0.0;
		};

		spSolverCallbacks callbacks;   // Stark.cpp:295 hands these over at creation; user code may add to them later
		NewtonSettings settings;       // Stark.cpp:296 overwrites this before the first solve

		NewtonsMethod(spGlobalPotential gp, spContext ctx, spSolverCallbacks cbs = nullptr);
		~NewtonsMethod();
		static std::shared_ptr<NewtonsMethod> create(spGlobalPotential gp, spContext ctx, spSolverCallbacks cbs = nullptr);

		SolverReturn solve();                                                  // register / refresh, solve on the engine, DoFs back
		const SolveStats& get_last_solve_stats() const { return stats; }
		void print_summary(double total_time = -1.0) const;

		mistark_ctx* engine() const;                                           // extension: nullptr before the first solve
		long mistark_stale_solves() const;                                     // extension: solves that had to be redone because a callback edited a large input in place unseen by the sampled checks (0 in a healthy run; also the logger entry "mistark_stale_solves")

	private:
		struct Impl;
		std::unique_ptr<Impl> impl;
		spGlobalPotential global_potential;
		spContext context;
		SolveStats stats;
	};
	using spNewtonsMethod = std::shared_ptr<NewtonsMethod>;
}
