import sys
sys.path.insert(0, ".")
from bench import build_scene
from stark_amd import sim as S
sim = build_scene(S, 44, 44, 43, 0)
for i in range(6):
    sim.run_one_step()
    st = sim.info().last_stats
    print("step", i, "newton", st.newton_iterations, "cg", st.cg_iterations, "solves", st.n_linear_solves, "n_hess", st.n_hessians, "n_proj", st.n_projected_hessians, "ratio %.3f" % st.projected_hessians_ratio,
          "t_proj %.1f ms t_ls %.1f ms t_eval %.1f ms total %.1f ms" % (1e3*st.t_project, 1e3*st.t_linear_solve, 1e3*st.t_eval_pgh, 1e3*st.t_total))
