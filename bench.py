#!/usr/bin/env python3
"""bench.py — Newton-steps/s and ms/linear-solve of the MI355X engine on BASELINE.json's headline workload.

A "step" is ONE NEWTON ITERATION of the hot path (contact detection -> evaluate E/grad/element Hessians -> [project] ->
assemble -> block-Jacobi PCG -> intersection check -> line search), the reference's `newton_iterations` counter
(NewtonsMethod.cpp:243,249). The workload is configs[3]: a 1M-tet Soft_Rubber (stable Neo-Hookean, full EnergyTetStrain)
block generate_tet_grid{44,44,43} resting on a fixed rigid box {3,3,0.1} with IPC frictional contact (thickness 1e-3,
mu 0.5, min contact stiffness 1e8), gravity, dt = 1/30. The block starts 1.5 mm above the box (inside the 2 mm barrier range)
instead of SURVEY.md's 5 cm so that the barrier is active from the first Newton step of the timed region.
`--scene clamped` runs the contact-free variant (bottom face clamped) used by earlier profiles.
Synthetic data only. Inputs are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W         (N>1: launched by torch.distributed.run, one rank per GPU)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


GAP, THICKNESS, MU, KMIN, BOX = 0.0015, 1e-3, 0.5, 1e8, (3.0, 3.0, 0.1)


def build_scene(S, nx, ny, nz, device, scene="contact", eo=False, offset=(0.0, 0.0)):
    st = S.default_settings()
    st.device = device
    st.mirror_state_to_host = 0      # state stays in HBM between steps
    st.init_frictional_contact = 1 if scene == "contact" else 0
    sim = S.Simulation(st)
    p = S.soft_rubber()
    p.elasticity_only = 1 if eo else 0
    if scene != "contact":
        ps = sim.add_volume_grid("block", (0.0, 0.0, 0.6), (1.0, 1.0, 1.0), (nx, ny, nz), p)
        sim.prescribe_inside_aabb(ps, (0.0, 0.0, 0.1), (2.0, 2.0, 2e-3), 1e7)
        return sim
    gp = S.contact_global_params()
    gp.default_contact_thickness = THICKNESS
    gp.min_contact_stiffness = KMIN
    sim.set_contact_global_params(gp)
    # (box registered first: see oracle/ref_harness.cpp scene_blockbox on why the order matters for the reference's friction)
    rb = sim.add_rigid_box("box", 1.0, BOX)
    sim.rb_add_constraint("fix", rb)
    # (offset: tests move the block off the box's diagonal, above which the centred grid has nodes exactly on a contact-classification tie)
    ps = sim.add_volume_grid("block", (offset[0], offset[1], 0.5 * BOX[2] + GAP + 0.5), (1.0, 1.0, 1.0), (nx, ny, nz), p)
    sim.set_friction(sim.contact_group("d", ps), sim.contact_group("rb", rb), MU)
    return sim


def run_newton_steps(sim, S, capi, n_steps):
    """Runs time steps until exactly n_steps Newton iterations have been executed. Returns (newton, linear_solves, cg, t_linear)."""
    base = sim.info()
    n0 = base.total_newton_iterations
    ls0, cg0, tl0 = base.total_linear_solves, base.total_cg_iterations, base.total_linear_solve_time
    ns = S.default_settings().newton
    while True:
        done = sim.info().total_newton_iterations - n0
        remaining = n_steps - done
        if remaining <= 0:
            break
        ns.max_iterations = int(remaining)
        ns.max_iterations_as_success = 1
        sim.set_newton_settings(ns)
        if not sim.run_one_step():
            raise RuntimeError("simulation stopped")
        if sim.info().total_newton_iterations - n0 == done and sim.info().last_stats.newton_iterations == 0:
            # a converged time step costs evaluations but no Newton iteration: keep stepping (state advances)
            continue
    i = sim.info()
    return i.total_newton_iterations - n0, i.total_linear_solves - ls0, i.total_cg_iterations - cg0, i.total_linear_solve_time - tl0


def host_cpu():
    """CPU model, physical cores and logical CPUs of this host (from /proc/cpuinfo)."""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                phys = v
            elif k == "core id":
                cores.add((phys, v))
    except OSError:
        pass
    logical = logical or (os.cpu_count() or 1)
    return model, (len(cores) or logical), logical


def cpu_baseline(nx, ny, nz, scene="contact", offset=(0.0, 0.0), sweep=(8, 16, 32, 64, "physical"), steps=4):
    """The UNMODIFIED reference (oracle/_ref/ref_harness, built by oracle/Makefile) timed on this host's cores at several thread counts
    (it does not scale monotonically: 8 threads beat 64 on this scene); `value` is the BEST of them, every leg is reported. `steps` time steps
    after one warm-up step: four of them hold about the Newton iterations the GPU's timed window holds (iterations 5..24 of the run)."""
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    model, physical, logical = host_cpu()
    base = {"unit": "Newton-steps/s", "cores": None, "kind": "reference", "cpu_model": model, "physical_cores": physical, "logical_cpus": logical}
    if not os.path.exists(harness):
        return dict(base, value=None, cores=0, sample="unavailable: oracle/_ref/ref_harness not built")
    common = ["nx=%d" % nx, "ny=%d" % ny, "nz=%d" % nz, "codegen=/tmp/mistark_bench_codegen", "outdir=/tmp/mistark_bench_out"]
    name = "tetblock"
    if scene == "contact":
        name = "blockbox"
        common += ["L=1", "gap=%g" % GAP, "thickness=%g" % THICKNESS, "mu=%g" % MU, "kmin=%g" % KMIN, "bx=%g" % BOX[0], "bz=%g" % BOX[2], "boxfirst=1",
                   "ox=%.17g" % offset[0], "oy=%.17g" % offset[1]]

    def run(n_threads, n_steps):
        out = subprocess.run([harness, "time", name] + common + ["threads=%d" % n_threads, "steps=%d" % n_steps, "warmup=1"], check=True, capture_output=True, timeout=1500).stdout.decode()
        return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])

    # ("physical": one thread per physical core, what SURVEY 8d asks for beside the 8-thread figure; the reference scales negatively past 16
    # threads on this scene, so the best leg is a small one — cores_swept says what was tried)
    threads = sorted({max(1, min(physical if t == "physical" else t, logical)) for t in sweep})
    try:
        subprocess.run([harness, "prime", name, "nx=2", "ny=2", "nz=2"] + common[3:] + ["threads=%d" % threads[0]], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    except Exception as e:  # noqa: BLE001
        return dict(base, value=None, cores=0, sample="failed: %r" % (e,))
    legs = {}
    for t in threads:
        try:
            r = run(t, steps)
            legs[str(t)] = {"value": r["newton_steps_per_s"], "ms_per_linear_solve": r["ms_per_linear_solve"], "newton_iterations": r["newton_iterations"],
                            "linear_solves": r.get("linear_solves"), "wall_s": r["wall_s"]}
        except Exception as e:  # noqa: BLE001
            legs[str(t)] = {"value": None, "sample": "failed: %r" % (e,)}
    ok = {t: v for t, v in legs.items() if v.get("value")}
    if not ok:
        return dict(base, value=None, cores=0, sample="failed", by_threads=legs)
    best = max(ok, key=lambda t: ok[t]["value"])
    # The reference's own build says -march=native (stark/CMakeLists.txt:25); oracle/_ref is x86-64-v3 so that it runs on any host of the pool.
    # Where the host has AVX-512 the best leg is repeated with the x86-64-v4 build of the same sources (oracle/_ref_v4, `make -C oracle v4`) and the
    # faster of the two is the baseline: the GPU is compared with the best the reference does on this box.
    builds = {"x86-64-v3": ok[best]["value"]}
    harness_v4 = os.path.join(ROOT, "oracle", "_ref_v4", "ref_harness")
    try:
        has_avx512 = any(" avx512f" in l for l in open("/proc/cpuinfo") if l.startswith("flags"))
    except OSError:
        has_avx512 = False
    if has_avx512 and os.path.exists(harness_v4):
        try:
            harness = harness_v4
            r = run(int(best), steps)
            builds["x86-64-v4"] = r["newton_steps_per_s"]
            if r["newton_steps_per_s"] > ok[best]["value"]:
                ok[best] = {"value": r["newton_steps_per_s"], "ms_per_linear_solve": r["ms_per_linear_solve"], "newton_iterations": r["newton_iterations"],
                            "linear_solves": r.get("linear_solves"), "wall_s": r["wall_s"]}
        except Exception as e:  # noqa: BLE001
            builds["x86-64-v4"] = "failed: %r" % (e,)
    base["builds"] = builds
    return dict(base, value=ok[best]["value"], cores=int(best), cores_swept=threads, value_at_physical_cores=legs.get(str(min(physical, logical)), {}).get("value"),
                ms_per_linear_solve=ok[best]["ms_per_linear_solve"], newton_iterations=ok[best]["newton_iterations"],
                linear_solves=ok[best]["linear_solves"], wall_s=ok[best]["wall_s"], by_threads=legs,
                sample="same scene, %d time steps after 1 warm-up step (pattern build + JIT excluded) at %s threads; value = the best leg (%s threads)" % (steps, "/".join(str(t) for t in threads), best))


def profile_traffic():
    """HBM-side bytes per real SpMV launch from the newest committed rocprofv3 PMC passes (profiles/<tag>_pmc_fetch.txt / _pmc_write.txt,
    made by tools/profile_round.sh from this command): 2 x FETCH_SIZE (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md
    "HBM") + WRITE_SIZE, KB -> bytes. Counters cannot be collected inside a timed run; this is a profile-derived figure and labelled so."""
    import glob
    import re

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_fetch.txt"))):
        w = f.replace("_pmc_fetch.txt", "_pmc_write.txt")
        if not os.path.exists(w):
            continue
        vals = []
        for path in (f, w):
            m = re.findall(r"k_spmv_fused: .* avg counter value over those: ([0-9.]+)", open(path).read())
            vals.append(float(m[-1]) if m else None)
        if None not in vals:
            tag = os.path.basename(f)[:-len("_pmc_fetch.txt")]
            key = [int(x) for x in re.findall(r"\d+", tag)]
            if best is None or key > best[0]:
                best = (key, tag, (2.0 * vals[0] + vals[1]) * 1024.0)
    return (best[2], "profiles/%s_pmc_{fetch,write}.txt" % best[1]) if best else (None, None)


def profile_kernel_trace():
    """Average duration of the SpMV launches that did work in the newest committed rocprofv3 kernel trace of this command
    (profiles/<tag>_kernel_stats.txt, last line, written by profiles/summarize_rocpd.py): (ms, source) or (None, None)."""
    import glob
    import re

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats.txt"))):
        m = re.findall(r"k_spmv launches > 5 us \(real work\): n=(\d+) avg=([0-9.]+) us", open(f).read())
        if m:
            tag = os.path.basename(f)[:-len("_kernel_stats.txt")]
            key = [int(x) for x in re.findall(r"\d+", tag)]
            if best is None or key > best[0]:
                best = (key, tag, float(m[-1][1]) * 1e-3)
    return (best[2], "profiles/%s_kernel_stats.txt" % best[1]) if best else (None, None)


PINNED_OFFSET = (0.00137, -0.00053)   # the placement on which the reference reproduces itself and the engine's counts equal its log
HBM_GRID = (88, 88, 86)               # 7.99 M tets: the matrix (790 MB) is out of reach of the 256 MiB Infinity Cache
PREFLIGHT_TIMEOUT_S = 2.0             # every wait of the windows' pre-flight gives up after this long (N > 1)
RCCL_LEG_REPS = 200                   # all-reduces per size in the RCCL leg of an N > 1 line
RCCL_LEG_TIMEOUT_S = 180.0            # a communicator that does not come up within this long is reported, not waited for


def pinned_placement_run(S, capi, nx, ny, nz, device, steps, warmup, with_cpu):
    """The same workload with the block moved PINNED_OFFSET off the box's axes: the scene the parity tests pin entry by entry against the
    reference's step log (tests/test_gpu_fullsize.py). Same measurement as the headline figure, reported beside it."""
    sim = build_scene(S, nx, ny, nz, device, "contact", offset=PINNED_OFFSET)
    run_newton_steps(sim, S, capi, warmup)
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    newton, n_ls, n_cg, t_ls = run_newton_steps(sim, S, capi, steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    sim.close()
    out = {"offset": list(PINNED_OFFSET), "value": newton / el, "unit": "Newton-steps/s", "ms_per_step": 1e3 * el / max(newton, 1), "ms_per_linear_solve": 1e3 * t_ls / max(n_ls, 1),
           "linear_solves": n_ls, "cg_iterations": n_cg, "newton_iterations": newton, "steps": steps, "warmup": warmup,
           "pinned_by": "tests/test_gpu_fullsize.py::test_full_size_first_time_steps_equal_the_reference_log_off_the_degenerate_placement"}
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(nx, ny, nz, "contact", PINNED_OFFSET, sweep=(8, 16, 32), steps=3)
    return out


EXTRA_WINDOWS = 2   # further windows of `steps` Newton iterations behind the headline one (N = 1), each from the same start state


def extra_windows(S, capi, nx, ny, nz, device, scene, offset, steps, warmup, first):
    """`value` is the FIRST window (the contract's W warm-up + K timed iterations). A window of 20 iterations is 0.12 s against a box-to-box and
    run-to-run spread of a few percent, so the line also carries EXTRA_WINDOWS more windows measured the same way, each on a freshly
    built scene (the same start state: the run is deterministic, every window executes the same Newton iterations) and their median."""
    import statistics

    import torch
    vals, iters = [first["value"]], [first["newton_iterations"]]
    ms_solve = [first["ms_per_linear_solve"]]
    for _ in range(EXTRA_WINDOWS):
        sim = build_scene(S, nx, ny, nz, device, scene, offset=offset)
        if warmup > 0:
            run_newton_steps(sim, S, capi, warmup)
        else:
            sim.prepare()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        newton, n_ls, n_cg, t_ls = run_newton_steps(sim, S, capi, steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        sim.close()
        vals.append(newton / el)
        iters.append(newton)
        ms_solve.append(1e3 * t_ls / max(n_ls, 1))
    return {"values": [round(v, 3) for v in vals], "median": round(statistics.median(vals), 3), "min": round(min(vals), 3), "max": round(max(vals), 3),
            "spread_rel": round((max(vals) - min(vals)) / statistics.median(vals), 4), "newton_iterations": iters,
            "ms_per_linear_solve": [round(v, 4) for v in ms_solve], "windows": len(vals),
            "what": "window 0 is `value` (driver-timed contract); windows 1.. = the same W warm-up + K timed Newton iterations on a freshly built scene "
                    "(same start state, same iterations), timed between device synchronisations"}


def secondary_run(S, capi, device, steps, warmup, set_dist, barrier, dist, torch, one_device=False, world=1):
    """The same scene at HBM_GRID on the ranks of this run: `steps` Newton iterations after `warmup`, timed like the headline figure (barrier
    + synchronisation on both sides, maximum over the ranks). Returns (dict, sim) — the caller closes the scene."""
    nx, ny, nz = HBM_GRID
    sim = build_scene(S, nx, ny, nz, device, "contact")
    if set_dist is not None:
        set_dist(sim)
    if one_device and world > 1:
        sim.prepare()
        capi.lib().mistark_set_option(sim.engine_handle(), b"spmv_grid_cap", max(8, (1024 // world) // 8 * 8))
        capi.lib().mistark_set_option(sim.engine_handle(), b"no_eval_prelaunch", 1)
    run_newton_steps(sim, S, capi, max(warmup, 1))
    barrier()
    t0 = time.perf_counter()
    newton, n_ls, n_cg, t_ls = run_newton_steps(sim, S, capi, steps)
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    out = {"grid": "%d,%d,%d" % HBM_GRID, "tets": 12 * nx * ny * nz, "dofs": sim.info().ndofs, "n_gpus": world, "value": newton / el, "unit": "Newton-steps/s",
           "ms_per_step": 1e3 * el / max(newton, 1), "steps": steps, "warmup": max(warmup, 1), "ms_per_linear_solve": 1e3 * t_ls / max(n_ls, 1), "linear_solves": n_ls,
           "cg_iterations_per_solve": n_cg / max(n_ls, 1), "scaling": "strong"}
    if set_dist is not None:
        sim.close()
        return out
    return out, sim


def hbm_resident_pass(S, capi, device, steps, warmup, torch):
    """SpMV pass on the same scene at HBM_GRID (the secondary workload's Newton iterations first, then 50 back-to-back launches between HIP
    events): the roofline figure with the matrix streaming from HBM instead of the Infinity Cache. Returns (roofline dict, secondary dict)."""
    import ctypes as C
    nx, ny, nz = HBM_GRID

    def barrier():
        torch.cuda.synchronize()

    sec, sim = secondary_run(S, capi, device, steps, warmup, None, barrier, None, torch)
    _, _, nbytes = sim.spmv_timing(reset=-1)
    us = C.c_double()
    if capi.lib().mistark_spmv_bench(sim.engine_handle(), 50, C.byref(us)) != 0:
        raise RuntimeError(capi.lib().mistark_last_error(sim.engine_handle()))
    ndofs = sim.info().ndofs
    sim.close()
    gbs = nbytes / (us.value * 1e-6) / 1e9
    return ({"grid": "%d,%d,%d" % HBM_GRID, "tets": 12 * nx * ny * nz, "dofs": ndofs, "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": us.value * 1e-3, "launches_timed": 50,
            "achieved": gbs, "unit": "GB/s", "frac": gbs / 8000.0, "timing": "HIP events around 50 back-to-back launches (mistark_spmv_bench)"}, sec)



# Keys of the one JSON line (tests/test_bench_cli_cpu.py pins them; tests/test_gpu_multiprocess.py checks a real N > 1 line against them).
LINE_KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
             "preflight", "peer_latency_us", "rccl", "stages_ms_per_newton_iteration", "ms_per_linear_solve", "cg_iterations_per_solve", "linear_solves",
             "cg_iterations", "sharded_cg_kernels_us", "newton_iterations", "host_timers_s", "contact", "roofline", "cpu_baseline", "value_windows"]
VALUE_WINDOWS_KEYS = ["values", "median", "min", "max", "spread_rel", "newton_iterations", "ms_per_linear_solve", "windows"]
RCCL_LEG_KEYS = ["ranks", "allreduce_ndofs_us", "allreduce_3_us"]                      # always present, null when the leg was refused or failed
STAGE_TABLE_KEYS = ["measured", "model", "model_one_gpu", "model_speedup", "measured_over_model"]
STAGE_KEYS = ["linear_solve", "evaluation_assembly_projection", "contact_callbacks", "total"]
PREFLIGHT_KEYS = ["timeout_s", "peer_latency_us", "ok", "wall_s"]

# DESIGN.md "Multi-GPU": what the step of configs[3] is expected to cost per Newton iteration on N GPUs — the figures the >= 6x of north_star was
# judged against ("2.3x at 8"). Inputs measured with all ranks on ONE device (profiles/r03_shardN_bench.json, r02_shard8_stages.txt):
#   kernels of one fused CG iteration on a rank's rows (SpMV with halo polls + vector kernel), microseconds, solo
MODEL_CG_KERNELS_US = {1: 38.7, 2: 28.0, 4: 19.0, 8: 14.2}
MODEL_CG_EXPOSED_US = 2.0          # the one exposed exchange of an iteration (assumed: cannot be measured on one device)
MODEL_SOLVE_OVERHEAD_MS = 0.11     # per linear solve: prologue, preconditioner, gather of the solution, read-backs
MODEL_EVAL_1GPU_MS = 1.5           # evaluation + assembly + projection per Newton iteration on one GPU
MODEL_EVAL_FLOOR_MS = 0.1          # their small kernels and exchanges, which do not shrink with the shard
MODEL_CONTACT_REPLICATED_MS = 0.4  # contact search + pattern per Newton iteration: box build, sorts, routing, host chain (replicated)
MODEL_CONTACT_SWEEP_MS = 0.4       # ... and the sweeps, dealt out to the ranks


def model_ms_per_newton(world, cg_per_newton, solves_per_newton, max_element_share):
    """The model's stage column for `world` ranks at this run's own CG / solve counts. max_element_share = the largest share of all elements
    a rank evaluates (1 / world + the interface elements evaluated on both sides)."""
    kern = MODEL_CG_KERNELS_US.get(world)
    if kern is None:   # between the measured points: the SpMV's share shrinks with the rows, the vector kernel's 7-8 us do not
        kern = 8.0 + (MODEL_CG_KERNELS_US[1] - 8.0) / world
    m = {"linear_solve": cg_per_newton * (kern + (MODEL_CG_EXPOSED_US if world > 1 else 0.0)) * 1e-3 + solves_per_newton * MODEL_SOLVE_OVERHEAD_MS,
         "evaluation_assembly_projection": MODEL_EVAL_1GPU_MS * max_element_share + (MODEL_EVAL_FLOOR_MS if world > 1 else 0.0),
         "contact_callbacks": MODEL_CONTACT_REPLICATED_MS + MODEL_CONTACT_SWEEP_MS / world}
    m["total"] = sum(m.values())
    return {k: round(v, 3) for k, v in m.items()}


def stage_table(per_rank_ms, world, newton, n_ls, n_cg, ranks_seen, total_ms):
    """Measured: per stage the slowest rank's GPU-event time per Newton iteration of the timed region. Model: DESIGN.md's column for this N and,
    as the divisor, for one GPU — `model_speedup` is what the model predicts for this N, `measured_total` / a one-GPU line's ms_per_step is what the
    driver's scaling curve will show."""
    worst = {k: max(r[k] for r in per_rank_ms) for k in per_rank_ms[0]}
    measured = {"linear_solve": worst["linear_solve"], "evaluation_assembly_projection": worst["eval_pgh"] + worst["eval_p"] + worst["project"] + worst["assembly"],
                "contact_callbacks": worst["callback"], "total": total_ms}
    # the largest share of the unsharded problem's elements a rank evaluates (1 / world + interface elements, which every side evaluates)
    share = max((r["elements_evaluated"] / max(r["elements_total"], 1) for r in (ranks_seen or [])), default=1.0 / world)
    cgn, sn = n_cg / max(newton, 1), n_ls / max(newton, 1)
    model_n = model_ms_per_newton(world, cgn, sn, share)
    model_1 = model_ms_per_newton(1, cgn, sn, 1.0)
    return {"measured": {k: round(v, 3) for k, v in measured.items()}, "model": model_n, "model_one_gpu": model_1,
            "model_speedup": round(model_1["total"] / model_n["total"], 2), "measured_over_model": round(measured["total"] / model_n["total"], 2),
            "cg_iterations_per_newton_iteration": round(cgn, 2), "linear_solves_per_newton_iteration": round(sn, 2), "largest_element_share": round(share, 4),
            "ranks_on_one_device": bool(os.environ.get("MISTARK_BENCH_DEVICE")),
            "note": "measured = slowest rank per stage (GPU events on the engine's stream); with all ranks on one device the ranks time-share it and the "
                    "measured column proves the path, not the model"}


def rccl_allreduce_leg(capi, dist, rank, world, device, ranks_seen, broadcast_uid, allgather, ndofs):
    """The RCCL leg of an N > 1 line: one communicator over all ranks (ncclCommInitRank through the engine's own dlopen'ed entry points),
    RCCL_LEG_REPS x ncclAllReduce(f64, sum) of `ndofs` doubles (the gradient) and of 3 doubles (the dot products of a CG iteration,
    solve_pcg.h:180,201,217), results checked. Runs whatever transport the engine itself took, in a thread the launcher gives up on after
    RCCL_LEG_TIMEOUT_S. Returns (dict for the line, hung?)."""
    import ctypes as C
    import threading

    devices = {(r["device"], r["pci_bus_id"]) for r in (ranks_seen or [])}
    if len(devices) < world:
        return ({"ranks": None, "refused": "ranks share a device (%d distinct device(s) for %d ranks): RCCL needs one device per rank" % (len(devices), world),
                 "allreduce_ndofs_us": None, "allreduce_3_us": None}, False)
    uid = broadcast_uid(required=False)   # (None on EVERY rank when rank 0 could not make one: a refused leg, not an exception on one rank)
    if uid is None:
        return ({"ranks": None, "refused": "rank 0 could not create an ncclUniqueId", "allreduce_ndofs_us": None, "allreduce_3_us": None}, False)
    out = (C.c_double * 4)()
    err = C.create_string_buffer(512)
    box = {}

    def work():
        box["rc"] = capi.lib().mistark_rccl_allreduce_bench(device, rank, world, uid, int(ndofs), RCCL_LEG_REPS, out, err, 512)

    t = threading.Thread(target=work, daemon=True)
    t0 = time.perf_counter()
    t.start()
    t.join(RCCL_LEG_TIMEOUT_S)
    hung = t.is_alive()
    mine = {"rank": rank, "hung": hung, "rc": box.get("rc"), "error": err.value.decode(errors="replace") if box.get("rc") not in (0, None) else None,
            "ranks": int(out[0]) if box.get("rc") == 0 else None, "allreduce_ndofs_us": out[1] if box.get("rc") == 0 else None,
            "allreduce_3_us": out[2] if box.get("rc") == 0 else None, "comm_init_s": out[3] if box.get("rc") == 0 else None}
    every = allgather(mine)   # (gloo: host side, works whether or not RCCL came up)
    any_hung = any(e["hung"] for e in every)
    ok = [e for e in every if e["rc"] == 0]
    leg = {"ranks": ok[0]["ranks"] if ok else None, "ranks_reported_by_every_rank": [e["ranks"] for e in every],
           "ndofs": int(ndofs), "repetitions": RCCL_LEG_REPS,
           # slowest rank's average: what an exchange costs the job
           "allreduce_ndofs_us": round(max(e["allreduce_ndofs_us"] for e in ok), 2) if len(ok) == world else None,
           "allreduce_3_us": round(max(e["allreduce_3_us"] for e in ok), 2) if len(ok) == world else None,
           "allreduce_ndofs_algbw_GBps": round(8.0 * ndofs / (max(e["allreduce_ndofs_us"] for e in ok) * 1e-6) / 1e9, 1) if len(ok) == world else None,
           "comm_init_s": round(max(e["comm_init_s"] for e in ok), 2) if ok else None, "wall_s": round(time.perf_counter() - t0, 2),
           "results_checked": "every element of the first all-reduce of each size against the closed-form sum over the ranks",
           "what": "ncclAllReduce(f64, sum) back to back on one stream between HIP events after 5 warm-up launches, one communicator over all ranks "
                   "(mistark_rccl_allreduce_bench: the engine's dlopen'ed librccl entry points); slowest rank's average"}
    errs = [e["error"] for e in every if e["error"]]
    if errs:
        leg["error"] = errs[0]
    if any_hung:
        leg["error"] = "timed out after %.0f s on rank(s) %s" % (RCCL_LEG_TIMEOUT_S, [e["rank"] for e in every if e["hung"]])
    return leg, any_hung


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves (one process per GPU, torch.distributed.run on
    127.0.0.1) with this very command line; rank 0's JSON line is the only thing on stdout. Refuses to pretend: fewer visible GPUs than ranks
    is an error unless MISTARK_BENCH_DEVICE says that all ranks are to share one device (test boxes)."""
    import socket

    import torch

    have = torch.cuda.device_count()
    if have < n and os.environ.get("MISTARK_BENCH_DEVICE") is None:
        print("bench.py: --gpus %d but only %d GPU(s) visible; set MISTARK_BENCH_DEVICE=<id> to run all ranks on one device (a path test, not a "
              "scaling measurement)" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (window handles, RCCL)
    env.setdefault("OMP_NUM_THREADS", "1")
    env["MISTARK_BENCH_SELF_LAUNCHED"] = "1"
    if env.get("MISTARK_BENCH_DEVICE") is not None:
        env.setdefault("MISTARK_IPC_TIMEOUT_S", "10")   # (ranks sharing a device: a stalled exchange should fail the path test quickly)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks' stdout carries rank 0's JSON line — and whatever libraries print there (gloo announces its connections on stdout): only the
    # JSON line goes to this process's stdout, the rest to stderr
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
    for raw in p.stdout:
        line = raw.decode(errors="replace")
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
    sys.stdout.flush()
    return p.wait()


def drop_in_run(S, capi, nx, ny, nz, device, time_steps=16):
    """The same workload through the actual drop-in: oracle/_ref/shim_check_cd is the UNMODIFIED reference — its own stark::Simulation,
    EnergyFrictionalContact, callbacks, time stepping — compiled against shim/include/symx and shim/include_cd and linked with libmistark.so, i.e.
    what a STARK user gets by swapping the two headers. `value` = Newton iterations per second over time steps 2..time_steps (the first one
    registers everything and builds the pattern); `mirror_same_window` = this repo's own scene mirror over the same time steps; `mirror_ratio` =
    drop-in / mirror. The shim's own statistics (seconds inside the reference's callbacks, inside the shim's sync, bringing DoFs to the host) come
    from MISTARK_SHIM_STATS=1."""
    import re

    exe = os.path.join(ROOT, "oracle", "_ref", "shim_check_cd")
    if not os.path.exists(exe):
        return {"unavailable": "oracle/_ref/shim_check_cd not built (needs /root/reference in the build container: make -C oracle shim_cd)"}
    env = dict(os.environ, SHIM_GRID="%d,%d,%d" % (nx, ny, nz), MISTARK_SHIM_STATS="1", MISTARK_DEVICE=str(device))
    env.setdefault("SHIM_THREADS", "1")   # (the reference's own host-side work per callback is small: one thread beats sixteen on it, measured 1.29 s against 1.79 s)
    r = subprocess.run([exe, "benchblock", str(time_steps)], env=env, capture_output=True, timeout=1200)
    if r.returncode != 0:
        return {"unavailable": "shim_check_cd failed: " + r.stderr.decode()[-300:]}
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    err = r.stderr.decode()
    out = {"value": line["newton_steps_per_s_after_first"], "unit": "Newton-steps/s", "newton_iterations": line["newton_rest"], "time_steps": time_steps - 1,
           "wall_s": line["rest_s"], "first_time_step_s": line["first_step_s"], "binary": "oracle/_ref/shim_check_cd benchblock (unmodified stark/src/** on shim/include/symx + shim/include_cd)"}
    m = re.search(r"of ([0-9.e+-]+) s in solve\(\): ([0-9.e+-]+) s in the caller's callbacks, ([0-9.e+-]+) s in sync\(\) .*?, ([0-9.e+-]+) s bringing DoFs", err)
    if m:
        out.update(solve_s=float(m.group(1)), callbacks_s=float(m.group(2)), sync_s=float(m.group(3)), dofs_to_host_s=float(m.group(4)))
    # the mirror over the same time steps
    sim = build_scene(S, nx, ny, nz, device, "contact")
    import torch
    if not sim.run_one_step():
        raise RuntimeError("mirror: simulation stopped")
    torch.cuda.synchronize()
    i0 = sim.info()
    t0 = time.perf_counter()
    for _ in range(time_steps - 1):
        if not sim.run_one_step():
            raise RuntimeError("mirror: simulation stopped")
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    i1 = sim.info()
    sim.close()
    n = i1.total_newton_iterations - i0.total_newton_iterations
    out["mirror_same_window"] = {"value": n / el, "newton_iterations": n, "wall_s": el}
    out["mirror_ratio"] = out["value"] / (n / el)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--grid", type=str, default="44,44,43", help="hexahedra per dimension (12 tets each)")
    ap.add_argument("--scene", type=str, default="contact", choices=["contact", "clamped"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements of the default workload (pinned placement, HBM-resident SpMV pass)")
    ap.add_argument("--offset", type=str, default="0,0", help="block moved off the box's axes by (ox, oy) metres: the placement on which the reference reproduces itself "
                                                              "and the engine's Newton / solve / CG counts equal its log (DESIGN.md section 5); default: centred")
    a = ap.parse_args()
    nx, ny, nz = [int(v) for v in a.grid.split(",")]
    offset = tuple(float(v) for v in a.offset.split(","))

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    import torch

    from stark_amd import capi
    from stark_amd import sim as S

    # N > 1 on the default workload also runs the same scene at HBM_GRID (7.99 M tets: each of 8 ranks then holds what one GPU holds at
    # configs[3], the regime the row-sharded design is for) and reports it as config.secondary; N = 1 reports the same run beside the
    # HBM-resident SpMV figure, so that a reader has the one-GPU number the N-GPU one divides by
    secondary_wanted = a.scene == "contact" and (nx, ny, nz) == (44, 44, 43) and offset == (0.0, 0.0) and not a.no_extras
    dist = None
    uid = None
    comm = None
    transport = None
    transport_requested = None
    fallback_reason = None
    preflight = None
    ipc_selftest_us = None
    # (MISTARK_BENCH_DEVICE: all ranks on one device — only to exercise the N > 1 launch path on a single-GPU box)
    device = int(os.environ.get("MISTARK_BENCH_DEVICE", local_rank))
    if world > 1:
        # torch.distributed (gloo) only launches / synchronises the ranks and carries the transport's bootstrap data (64-byte window handles, or
        # the 128-byte RCCL id); the exchanges of the path (boundary values, dot products, energies) are issued by the engine itself on its
        # HIP stream (stark_amd/csrc/dist.hip): stores into the peers' IPC windows (default), or ncclAllGather (MISTARK_BENCH_TRANSPORT=rccl,
        # and the fallback when the windows cannot be set up — RCCL needs one device per rank)
        import ctypes as C

        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: gloo must not go looking for an interface through a hostname that may not resolve
        torch.cuda.set_device(device)
        sys.stdout.flush()
        saved_fd = os.dup(1)   # (gloo prints its connection report to the C stdout: keep this process's stdout for the one JSON line)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo")
            dist.barrier()
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

        def allgather_bytes(b):
            out = [None] * world
            dist.all_gather_object(out, b)
            return out

        transport = os.environ.get("MISTARK_BENCH_TRANSPORT", "ipc")
        transport_requested = transport
        if transport == "ipc":
            # Every phase ends in an all-gather of what went wrong, so that ALL ranks take the same way out: create the window (no collective
            # inside) -> exchange handles -> connect -> pre-flight (one tagged granule over every ordered pair of ranks, each wait bounded by
            # 2 s) -> self-test (all-gathers of 1024 doubles, every value checked). Whatever fails, everybody falls back to RCCL and the line says why.
            def agree(err):
                errs = allgather_bytes(err)
                return next((e for e in errs if e), None)

            n_rows = (nx + 1) * (ny + 1) * (nz + 1) + nx * ny * nz + 64
            if secondary_wanted:   # the windows also carry the 8 M-tet secondary workload
                n_rows = max(n_rows, (HBM_GRID[0] + 1) * (HBM_GRID[1] + 1) * (HBM_GRID[2] + 1) + HBM_GRID[0] * HBM_GRID[1] * HBM_GRID[2] + 64)
            err = None
            try:
                comm = capi.IpcComm(device, rank, world, max(64 * 3 * n_rows, 32 << 20))
            except Exception as e:  # noqa: BLE001
                err = "create: %r" % (e,)
            fallback_reason = agree(err)
            if fallback_reason is None:
                handles = allgather_bytes(comm.handle)
                try:
                    comm.connect(handles)
                except Exception as e:  # noqa: BLE001
                    err = "connect: %r" % (e,)
                fallback_reason = agree(err)
            if fallback_reason is None:
                t_pf = time.perf_counter()
                row, n_ok = None, 0
                dist.barrier()   # (the pre-flight's kernels must start within its time-out of each other)
                try:
                    n_ok, row = comm.preflight(16, PREFLIGHT_TIMEOUT_S)
                    if n_ok != world - 1:
                        err = "pre-flight: rank %d got an answer from %d of %d peers within %.1f s" % (rank, n_ok, world - 1, PREFLIGHT_TIMEOUT_S)
                except Exception as e:  # noqa: BLE001
                    err = "pre-flight: %r" % (e,)
                rows = allgather_bytes(row)
                preflight = {"timeout_s": PREFLIGHT_TIMEOUT_S, "exchanges_per_pair": 16, "wall_s": round(time.perf_counter() - t_pf, 3),
                             "what": "ping-pong of one 8-byte tagged granule between every ordered pair of ranks through the mapped windows; "
                                     "peer_latency_us[a][b] = half the best round trip rank a measured with rank b on its device clock (null: nothing came back)",
                             "peer_latency_us": [[None if v is None else round(v, 3) for v in r] if r else None for r in rows]}
                fallback_reason = agree(err)
                preflight["ok"] = fallback_reason is None
            if fallback_reason is None:
                try:
                    ipc_selftest_us = comm.selftest(1024, 50)[1]
                except Exception as e:  # noqa: BLE001
                    err = "self-test: %r" % (e,)
                fallback_reason = agree(err)
            if fallback_reason is not None:
                if rank == 0:
                    print("bench: IPC windows unavailable (%s); falling back to RCCL" % fallback_reason, file=sys.stderr)
                if comm is not None:
                    try:
                        comm.close()
                    except Exception:  # noqa: BLE001
                        pass
                comm, transport = None, "rccl"
        def broadcast_uid(required=True):
            """A fresh ncclUniqueId from rank 0 (one per communicator: an id is single-use). Rank 0's failure travels through the SAME
            broadcast as the id, so every rank learns of it together: all raise (required) or all get None (the caller refuses its leg)."""
            box = [(None, None)]
            if rank == 0:
                buf = C.create_string_buffer(128)
                try:
                    rc = capi.lib().mistark_dist_unique_id(buf)
                    box[0] = (buf.raw, None) if rc == 0 else (None, "ncclGetUniqueId failed (rc %d)" % rc)
                except Exception as e:  # noqa: BLE001
                    box[0] = (None, "ncclGetUniqueId: %r" % (e,))
            dist.broadcast_object_list(box, src=0)
            uid_raw, uid_err = box[0]
            if uid_raw is None and required:
                raise RuntimeError(uid_err or "no ncclUniqueId from rank 0")
            return uid_raw

        if transport == "rccl":
            uid = broadcast_uid()

    sim = build_scene(S, nx, ny, nz, device, a.scene, offset=offset)
    if world > 1:
        if comm is not None:
            sim.set_dist_ipc(comm, rank, world)
        else:
            sim.set_dist_rccl(rank, world, uid)
    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ranks_seen = None
    if world > 1:
        # what actually runs, rank by rank, as the engine and the runtime report it (not what the command line asked for)
        import ctypes as _Cd
        sim.prepare()
        di = (_Cd.c_int64 * 15)()
        if capi.lib().mistark_dist_info(sim.engine_handle(), di, 15) != 0:
            raise RuntimeError(capi.lib().mistark_last_error(sim.engine_handle()))
        props = torch.cuda.get_device_properties(device)
        mine = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(), "device": device, "device_name": props.name,
                "pci_bus_id": getattr(props, "pci_bus_id", None), "engine_world": int(di[8]), "engine_rank": int(di[9]),
                "transport": {0: None, 1: "in-process", 2: "rccl", 3: "ipc"}.get(int(di[10])), "transport_ranks": int(di[11]),
                "rows_owned": int(di[0]), "ghost_rows": int(di[1]), "elements_evaluated": int(di[14]), "elements_total": int(di[13])}
        ranks_seen = allgather_bytes(mine)

    if world > 1 and os.environ.get("MISTARK_BENCH_DEVICE") is not None:
        # all ranks on ONE device (test boxes): kernels that poll for a peer's data hold their workgroup slots, so the ranks' SpMV grids
        # together must leave room for the kernels they are waiting for (2048 resident 256-thread workgroups per MI355X)
        sim.prepare()
        if capi.lib().mistark_set_option(sim.engine_handle(), b"spmv_grid_cap", max(8, (1024 // world) // 8 * 8)) != 0:
            raise RuntimeError("spmv_grid_cap")
        # ... and every further stream per process oversubscribes the ONE device's hardware queues (measured with two ranks: everything,
        # the linear solves included, 2.7 times slower once each rank opens the stream of the early evaluation): off when ranks share a device
        if capi.lib().mistark_set_option(sim.engine_handle(), b"no_eval_prelaunch", 1) != 0:
            raise RuntimeError("no_eval_prelaunch")
    # warm-up (includes sparsity-pattern construction and first-touch allocations)
    if a.warmup > 0:
        run_newton_steps(sim, S, capi, a.warmup)
    else:
        sim.prepare()  # (--warmup 0: the engine and its registration exist from here on; pattern build and first-touch allocations are then timed)
    # A/B runs of engine options (measurement only; the engine exists after the first step): MISTARK_BENCH_OPTS="no_eager_assembly=1,fuse_dir=1"
    for kv in filter(None, os.environ.get("MISTARK_BENCH_OPTS", "").split(",")):
        k, v = kv.split("=")
        if capi.lib().mistark_set_option(sim.engine_handle(), k.encode(), int(v)) != 0:
            raise RuntimeError("%s: %s" % (kv, capi.lib().mistark_last_error(sim.engine_handle())))

    sim.spmv_timing(reset=-1 if os.environ.get("MISTARK_BENCH_NO_SPMV_SAMPLING") else 1)  # start SpMV timing for the timed region (the variable: A/B of what the sampling itself costs)
    barrier()
    info0 = sim.info()
    if os.environ.get("MISTARK_ALLOC_TRACE"):
        print("[alloc] ---- timed region begins", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    newton, n_ls, n_cg, t_ls = run_newton_steps(sim, S, capi, a.steps)
    barrier()
    t1 = time.perf_counter()
    if os.environ.get("MISTARK_ALLOC_TRACE"):
        print("[alloc] ---- timed region ends", file=sys.stderr, flush=True)
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    import ctypes as _C0
    _ov = _C0.c_double()
    spmv_ev_overhead_ms = _ov.value if capi.lib().mistark_spmv_event_overhead(sim.engine_handle(), _C0.byref(_ov)) == 0 and _ov.value > 0 else None
    if spmv_ev_overhead_ms is None and _ov.value > 0:
        spmv_ev_overhead_ms = _ov.value
    _clk_ms, _clk_n = _C0.c_double(), _C0.c_int64()
    spmv_clk_ms, spmv_clk_n = None, 0
    if capi.lib().mistark_spmv_device_clock(sim.engine_handle(), _C0.byref(_clk_ms), _C0.byref(_clk_n)) == 0 and _clk_n.value > 0:
        spmv_clk_ms, spmv_clk_n = _clk_ms.value, _clk_n.value
    spmv_ms, spmv_n, spmv_bytes = sim.spmv_timing(reset=-1)
    # N > 1 over windows: solo durations of the fused PCG iteration's two kernels on every rank's rows (the ranks take turns: valid on a box
    # where all ranks share one GPU, too)
    fused_kernels_us = None
    if world > 1 and comm is not None:
        import ctypes as _C1
        mine = None
        capi.lib().mistark_set_option(sim.engine_handle(), b"spmv_grid_cap", 0)  # (a rank measuring alone needs no cap: the grid a GPU of its own would get)
        for turn in range(world):  # one rank at a time, the others idle on the host: solo kernel durations
            if turn == rank:
                buf = (_C1.c_double * 2)()
                if capi.lib().mistark_dist_fused_bench(sim.engine_handle(), 200, buf) == 0:
                    mine = [round(buf[i], 2) for i in range(2)]
                else:
                    mine = capi.lib().mistark_last_error(sim.engine_handle()).decode()
            dist.barrier()
        every = allgather_bytes(mine)
        if all(isinstance(v, list) for v in every):
            fused_kernels_us = {"spmv_with_halo": [v[0] for v in every], "vector_kernel": [v[1] for v in every]}
        else:
            fused_kernels_us = {"unavailable": [v for v in every if not isinstance(v, list)][0]}
    spmv_b2b_ms = None
    if rank == 0:
        import ctypes as _C
        us = _C.c_double()
        if capi.lib().mistark_spmv_bench(sim.engine_handle(), 100, _C.byref(us)) == 0:
            spmv_b2b_ms = us.value * 1e-3
    info = sim.info()
    stage = {k: getattr(info, "total_" + k + "_time") - getattr(info0, "total_" + k + "_time") for k in ["newton", "linear_solve", "eval_pgh", "eval_p", "project", "assembly", "callback", "step"]}
    contact_info = sim.contact_info() if a.scene == "contact" else None
    if contact_info is not None:
        # barrier-table searches that ran on the device (requests answered from the installed tables are not counted) and how many of them ran at
        # the very state the previous one had searched (0 since round 5: DESIGN.md section 8, "A search that found the tables unchanged was run twice")
        import ctypes as _C3
        for key in ("contact_searches", "contact_repeated_searches"):
            v = _C3.c_int64()
            if capi.lib().mistark_get_counter(sim.engine_handle(), key.encode(), _C3.byref(v)) == 0:
                contact_info[key + "_since_start"] = int(v.value)
    if world > 1 and ranks_seen is not None:
        import ctypes as _Cd2
        di2 = (_Cd2.c_int64 * 13)()
        capi.lib().mistark_dist_info(sim.engine_handle(), di2, 13)
        counts = allgather_bytes((int(di2[12]), int(di2[6]), int(di2[7])))
        for r, (ns, nf, nu) in zip(ranks_seen, counts):
            r.update(contact_searches_with_the_sweep_dealt_out=ns, linear_solves_fused_iteration=nf, linear_solves_five_launch_iteration=nu)
    secondary = None
    if world > 1 and secondary_wanted:
        sim.close()
        sim = None
        uid2 = broadcast_uid() if comm is None else None   # (RCCL: the second scene gets a communicator of its own)
        try:
            secondary = secondary_run(S, capi, device, a.steps, a.warmup,
                                      (lambda s2: s2.set_dist_ipc(comm, rank, world)) if comm is not None else (lambda s2: s2.set_dist_rccl(rank, world, uid2)),
                                      barrier, dist, torch, one_device=os.environ.get("MISTARK_BENCH_DEVICE") is not None, world=world)
        except Exception as e:  # noqa: BLE001
            secondary = {"unavailable": repr(e)}
        # (a rank that failed above must not leave the others waiting in the next collective with a half-run scene: agree on the outcome)
        if any(isinstance(x, dict) and "unavailable" in x for x in allgather_bytes(secondary)):
            secondary = next(x for x in allgather_bytes(secondary) if isinstance(x, dict) and "unavailable" in x)

    rccl_leg = None
    stages_table = None
    rccl_hung = False
    if world == 1:
        # the one-GPU column of the scaling table, measured (GPU-event stage timers of this run) beside DESIGN.md's model of it
        stages_table = stage_table([{k: 1e3 * v / max(newton, 1) for k, v in stage.items()}], 1, newton, n_ls, n_cg, None, 1e3 * elapsed / max(newton, 1))
    if world > 1:
        # per-stage milliseconds per Newton iteration as measured in this run (GPU-event stage timers of every rank, the slowest rank counts)
        # beside the model DESIGN.md ("Multi-GPU") derives the expected scaling from
        per_rank = allgather_bytes({k: 1e3 * v / max(newton, 1) for k, v in stage.items()})
        stages_table = stage_table(per_rank, world, newton, n_ls, n_cg, ranks_seen, 1e3 * elapsed / max(newton, 1))
        rccl_leg, rccl_hung = rccl_allreduce_leg(capi, dist, rank, world, device, ranks_seen, broadcast_uid, allgather_bytes, info.ndofs)
    if rank == 0:
        traffic, traffic_src = profile_traffic()
        trace_ms, trace_src = profile_kernel_trace()
        n_tets = 12 * nx * ny * nz
        default_workload = a.scene == "contact" and (nx, ny, nz) == (44, 44, 43) and world == 1
        achieved_events = (spmv_bytes / (spmv_ms * 1e-3)) / 1e9 if spmv_ms > 0 else 0.0
        achieved_clock = (spmv_bytes / (spmv_clk_ms * 1e-3)) / 1e9 if spmv_clk_ms else None
        achieved_b2b = (spmv_bytes / (spmv_b2b_ms * 1e-3)) / 1e9 if spmv_b2b_ms else None
        # The headline figure is MEASURED IN THIS RUN: HIP events on the engine's stream around 100 back-to-back SpMV launches on the matrix the
        # timed region ended with (one event pair per batch: the 5 us an event pair costs the stream do not dilute a 20 us launch). Beside it, also
        # live: the sampled launches INSIDE the timed region on the device clock (first wavefront in to last wavefront out) and between HIP events
        # (dispatch and marker packets inside). The committed rocprofv3 kernel trace of this command is a side field
        # (`rocprofv3_profile_launch_ms`); `profile_agreement` = live / profile duration: a build whose SpMV regressed, or a stale profile, shows there.
        if world == 1 and achieved_b2b:
            achieved, basis, basis_ms, basis_n = achieved_b2b, "live, this run: HIP events around 100 back-to-back launches on the engine's stream right after the timed region (same matrix, same vectors)", spmv_b2b_ms, 100
        elif achieved_clock:
            achieved, basis, basis_ms, basis_n = achieved_clock, "device clock of sampled launches inside the timed region", spmv_clk_ms, spmv_clk_n
        elif achieved_events > 0:
            achieved, basis, basis_ms, basis_n = achieved_events, "HIP event bracket of sampled launches inside the timed region", spmv_ms, spmv_n
        else:  # a very short timed region may hold no sampled launch (only every 32nd SpMV of a solve is sampled)
            achieved, basis, basis_ms, basis_n = achieved_b2b or 0.0, "HIP events around 100 back-to-back launches after the timed region", spmv_b2b_ms, 100
        out = {
            "metric": "Newton-steps/s",
            # N>1: ONE scene, elements of every potential sharded over the GPUs (strong scaling: the work is fixed)
            "value": newton / elapsed,
            "unit": "Newton-steps/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1000.0 * elapsed / max(newton, 1),
            "higher_is_better": True,
            "scaling": "strong",  # N > 1 shards ONE fixed problem
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("configs[3]: tet block generate_tet_grid{%d,%d,%d} = %d tets / %d DoF, Soft_Rubber stable Neo-Hookean (full EnergyTetStrain) + lumped "
                             "inertia on a fixed rigid box {3,3,0.1}, IPC barrier + lagged friction (thickness 1e-3, mu 0.5, kmin 1e8), device proximity/intersection "
                             "detection every evaluation, gravity, dt=1/30, initial gap 1.5 mm%s" % (nx, ny, nz, n_tets, info.ndofs, "" if offset == (0.0, 0.0) else ", block moved (%g, %g) m off the box's axes" % offset)) if a.scene == "contact" else
                            ("tet block generate_tet_grid{%d,%d,%d} = %d tets / %d DoF, Soft_Rubber, bottom face clamped, gravity, dt=1/30, NO contact" % (nx, ny, nz, n_tets, info.ndofs)),
                "step": "one Newton iteration (contact detection, eval P+g+H, assembly, block-Jacobi PCG, intersection check, line search)",
                "parallelism": "single GPU" if world == 1 else ("block rows partitioned over %d GPUs; every rank evaluates the elements touching its rows (interface elements on both sides) and assembles "
                                                                   "and solves its rows: row-sharded block-Jacobi PCG, ghosts of p and the dot products exchanged %s "
                                                                   "in every iteration; the contact search's sweep dealt out to the ranks (keys all-gathered); state, line search, box sort and table routing replicated" %
                                                                   (world, "by stores into the peers' IPC windows (hipIpc; 8-byte tagged granules, no library call)" if transport == "ipc" else "by ncclAllGather (RCCL over xGMI)")),
                "transport": transport,
                "transport_requested": transport_requested,
                # not None: the windows were asked for and something on the way (creation, handle exchange, mapping, pre-flight, self-test) failed
                # on some rank; every rank then took RCCL (ncclAllGather on the engine's stream)
                "transport_fallback_reason": fallback_reason,
                # wall time of one all-gather of 1024 doubles through the windows in a train of 50 enqueued back to back (push kernel + polling
                # kernel: two boundaries, the one-way latency and the ranks' skew), measured by the transport's self-test before the scene is built
                "ipc_allgather_1024_doubles_us": ipc_selftest_us,
                "ranks_on_one_device": bool(os.environ.get("MISTARK_BENCH_DEVICE")) if world > 1 else None,
                # one entry per rank, as the engine and the runtime report it: device, the transport in use and the number of ranks the transport
                # itself counts (RCCL: ncclCommCount of the communicator)
                "ranks_seen": ranks_seen,
                "distinct_devices": len({(r["device"], r["pci_bus_id"]) for r in ranks_seen}) if ranks_seen else None,
                "launched_by": "bench.py itself (torch.distributed.run, one process per GPU)" if os.environ.get("MISTARK_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if world > 1 else None),
                "projection": "Progressive",
                # the same scene at 88 x 88 x 86 hexahedra (7.99 M tets / 4.1 M DoF) on the same ranks: Newton-steps/s of ONE problem 8 times the size
                "secondary": secondary,
            },
            # ---- N > 1 only (null at N = 1): the windows' pre-flight with its peer-latency matrix, the RCCL leg (ncclAllReduce of ndofs and of 3
            # doubles on a communicator of all ranks, whatever transport the engine itself uses) and measured stage times beside DESIGN.md's model
            "preflight": preflight,
            "peer_latency_us": preflight["peer_latency_us"] if preflight else None,
            "rccl": rccl_leg,
            "stages_ms_per_newton_iteration": stages_table,
            "ms_per_linear_solve": 1000.0 * t_ls / max(n_ls, 1),
            "cg_iterations_per_solve": n_cg / max(n_ls, 1),
            "linear_solves": n_ls,
            "cg_iterations": n_cg,
            # N > 1: what ONE CG iteration costs a rank in kernels (solo, microseconds, per rank): the iteration is these two launches, with one
            # exposed wait (the vector kernel's for the slowest rank's three sums) and the halo hidden behind the SpMV's interior rows
            "sharded_cg_kernels_us": fused_kernels_us,
            "newton_iterations": newton,
            "host_timers_s": {k: round(v, 6) for k, v in stage.items()},
            "contact": contact_info,
            "roofline": {
                "kernel": "k_spmv_fused (3x3-block CSR in row-aligned chunks, float values, double vectors)" if world == 1 else
                          "k_spmv_halo (the same SpMV over rank 0's rows of the sharded matrix, ghost columns polled from the IPC window): bytes and duration of ONE GPU's launch",
                "bound": "hbm",
                "achieved": achieved,
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": achieved / 8000.0,
                "basis": basis,
                "avg_launch_ms": basis_ms,
                "launches_timed": basis_n,
                "algorithmic_bytes_per_launch": spmv_bytes,
                # not measured in this run (PMC counters need their own rocprofv3 passes): taken from the newest committed profile of the same
                # command, named in traffic_source; null for other workloads or when no such profile exists
                "traffic": traffic if default_workload else None,
                "traffic_source": traffic_src if default_workload else None,
                # the matrix (99 MB) and the vectors fit the 256 MiB Infinity Cache: FETCH_SIZE counts cache hits too, so `achieved` is fabric-side
                # bandwidth; a plain float4 stream of the same value buffer reaches 6.3 TB/s on this box (tools/spmv_sweep.py, variant 9)
                "frac_of_stream_ceiling": achieved / 6300.0,
                "working_set": "Infinity-Cache resident (matrix 99 MB + vectors)" if (nx, ny, nz) == (44, 44, 43) else None,
                # ---- measured live in THIS run, every 32nd SpMV launch of the timed region:
                "live": {
                    # every workgroup of a sampled launch stamps its start and end on the device's constant clock (s_memrealtime); duration = max(end) - min(start)
                    "device_clock": {"launch_ms": spmv_clk_ms, "launches": spmv_clk_n, "achieved": achieved_clock, "frac": achieved_clock / 8000.0 if achieved_clock else None},
                    # the same launches between a pair of HIP events on the engine's stream (dispatch latency and the marker packets included);
                    # event_pair_overhead_ms: an EMPTY bracket recorded right behind each
                    "event_bracket": {"launch_ms": spmv_ms, "launches": spmv_n, "achieved": achieved_events, "frac": achieved_events / 8000.0, "event_pair_overhead_ms": spmv_ev_overhead_ms},
                    # 100 launches back to back after the timed region, one event pair around the batch
                    "back_to_back": {"launch_ms": spmv_b2b_ms, "launches": 100, "achieved": achieved_b2b, "frac": achieved_b2b / 8000.0 if achieved_b2b else None},
                },
                # the committed rocprofv3 kernel trace of this command (under the tracer; its duration spans the dispatch packet)
                "rocprofv3_profile_launch_ms": trace_ms if default_workload else None,
                "rocprofv3_profile_source": trace_src if default_workload else None,
                "profile_agreement": (basis_ms / trace_ms) if (default_workload and trace_ms and basis_ms) else None,
            },
        }
        out["value_windows"] = None
        if world == 1 and not a.no_extras:
            sim.close()
            sim = None
            try:
                out["value_windows"] = extra_windows(S, capi, nx, ny, nz, device, a.scene, offset, a.steps, a.warmup, out)
            except Exception as e:  # noqa: BLE001
                out["value_windows"] = {"unavailable": repr(e)}
        if default_workload and offset == (0.0, 0.0) and not a.no_extras:
            if sim is not None:
                sim.close()
            sim = None
            try:
                hb, sec = hbm_resident_pass(S, capi, device, a.steps, a.warmup, torch)
                out["roofline"]["hbm_resident"] = hb
                out["config"]["secondary"] = sec
            except Exception as e:  # noqa: BLE001
                out["roofline"]["hbm_resident"] = {"unavailable": repr(e)}
            try:
                out["drop_in"] = drop_in_run(S, capi, nx, ny, nz, device)
            except Exception as e:  # noqa: BLE001
                out["drop_in"] = {"unavailable": repr(e)}
            try:
                out["pinned_placement"] = pinned_placement_run(S, capi, nx, ny, nz, device, a.steps, a.warmup, not a.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                out["pinned_placement"] = {"unavailable": repr(e)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nx, ny, nz, a.scene, offset)
        else:
            out["cpu_baseline"] = {"value": None, "unit": "Newton-steps/s", "cores": 0, "kind": "reference", "sample": "skipped (N>1 or --no-cpu-baseline)"}
        print(json.dumps(out))
    if rccl_hung:
        # a thread of this process still sits in ncclCommInitRank / an all-reduce that never completed: the line is out, leave without the
        # orderly shutdown that would wait for it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    if sim is not None:
        sim.close()
    if dist is not None:
        dist.barrier()  # (nobody unmaps a window another rank may still be storing into)
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
