import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import bench
from stark_amd import capi, sim as S
for off in [(0.0,0.0),(0.00137,-0.00053)]:
    sim = bench.build_scene(S, 44,44,43, 0, "contact", offset=off)
    bench.run_newton_steps(sim, S, capi, 4)
    def ctr(n):
        v=C.c_int64(); assert capi.lib().mistark_get_counter(sim.engine_handle(), n, C.byref(v))==0; return v.value
    names=[b"evt_tet_us", b"evt_small_us", b"evt_gather_us", b"evt_main_us", b"evt_pattern_us", b"evt_n"]
    e0=[ctr(n) for n in names]
    a=(ctr(b"eval_pgh_issue_us"), ctr(b"eval_pgh_wait_us")); i0=sim.info()
    newton,n_ls,n_cg,t_ls = bench.run_newton_steps(sim, S, capi, 20)
    b=(ctr(b"eval_pgh_issue_us"), ctr(b"eval_pgh_wait_us")); i1=sim.info()
    nev = i1.n_evaluations - i0.n_evaluations if hasattr(i1,'n_evaluations') else None
    print(off, "newton", newton, "eval PGH host issue us total", b[0]-a[0], "wait us total", b[1]-a[1], "stage eval_pgh s", i1.total_eval_pgh_time-i0.total_eval_pgh_time)
    e1=[ctr(n) for n in names]
    n=max(e1[5]-e0[5],1)
    print("   GPU stamps per evaluation, us from the tets' start (MISTARK_EVAL_EVENTS=1):", {k.decode(): round((y-x)/n,1) for k,x,y in zip(names[:5],e0,e1)}, "n", n)
    sim.close()
