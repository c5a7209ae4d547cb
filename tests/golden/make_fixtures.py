#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference): `make -C oracle` builds oracle/_ref/ref_harness, which this
script drives (`prime` then `dump`/`traj`) and whose .npy/.json output it packs into one compressed .npz per fixture.
The fixtures are data only: evaluator inputs (connectivity tables + bound arrays in the reference's binding order) and
the reference's stage outputs. Usage: python tests/golden/make_fixtures.py [name ...]
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
OUT = os.path.join(ROOT, "tests", "golden")

# name -> (mode, scene, args)
FIXTURES = {
    # cfg-2 family (tet beam), elasticity-only and full (damping + strain limiting active)
    "tetbeam_eo_4x1x1": ("dump", "tetbeam", "nx=4 ny=1 nz=1 eo=1 steps=2 amp=0.05"),
    "tetbeam_full_4x1x1": ("dump", "tetbeam", "nx=4 ny=1 nz=1 eo=0 steps=2 amp=0.05 strain_damping=0.3 strain_limit=0.02 strain_limit_stiffness=1e4"),
    "tetbeam_eo_8x2x2": ("dump", "tetbeam", "nx=8 ny=2 nz=2 eo=1 steps=2 amp=0.02"),
    "tetbeam_softrubber_6x2x2": ("dump", "tetbeam", "nx=6 ny=2 nz=2 eo=0 steps=2 amp=0.02"),
    # strong perturbation: indefinite element Hessians -> exercises the PSD projection
    "tetbeam_eo_4x1x1_big": ("dump", "tetbeam", "nx=4 ny=1 nz=1 eo=1 steps=1 amp=1.5"),
    # cloth family (cfg-1/3 energies without contact)
    "cloth_flat_6": ("dump", "cloth", "n=6 flat=1 eo=0 steps=2 amp=0.05"),
    "cloth_shells_6": ("dump", "cloth", "n=6 flat=0 eo=0 steps=2 amp=0.05 bend_stiffness=1e-3 bend_damping=1e-4"),
    "cloth_eo_infl_5": ("dump", "cloth", "n=5 flat=1 eo=1 steps=2 amp=0.05 inflation=50"),
    # rigid bodies: chain of boxes exercising the 2 inertia + 11 constraint potentials
    "rbchain": ("dump", "rbchain", "steps=2 amp=0.3"),
    "traj_rbchain": ("traj", "rbchain", "steps=3"),
    # IPC contact + friction zoo (cloth over a fixed rigid box, rigid box + soft block over the cloth, rigid-rigid pair):
    # at t = 0 (all barrier tables of the initial gaps) and after one step
    "contactmix_t0": ("dump", "contactmix", "steps=0 amp=0.03"),
    "contactmix_t1": ("dump", "contactmix", "steps=1 amp=0.003"),
    # box / soft-block corners aimed at a corner and an edge of a fixed box: vertex-vertex and vertex-edge rows
    "contactcorners_t0": ("dump", "contactcorners", "steps=0 amp=0.01"),
    # rods (edge-only collision meshes) on a rigid box, across each other and under a cloth
    "contactrods_t0": ("dump", "contactrods", "steps=0 amp=0.03"),
    "contactrods_t1": ("dump", "contactrods", "steps=1 amp=0.003"),
    # known-answer vectors of the narrow-phase classification, edge-triangle intersection and friction geometry
    "contact_geometry": ("geom", "none", "n=600"),
    # rods + attachments (§8(f) rank 1): both segment-strain potentials and the five attachment potentials
    "attachzoo": ("dump", "attachzoo", "steps=2 amp=0.05"),
    "traj_attachzoo": ("traj", "attachzoo", "steps=4"),
    "attachdist": ("dump", "attachdist", "steps=1 amp=0.02"),
    "traj_attachdist": ("traj", "attachdist", "steps=4"),
    # the reference's example scene hanging_net (examples/main.cpp:12-39) at 12 x 12
    "traj_hangingnet_12": ("traj", "hangingnet", "n=12 steps=4"),
    # Newton trajectories (iterates after every Newton iteration)
    "traj_tetbeam_eo_8x2x2": ("traj", "tetbeam", "nx=8 ny=2 nz=2 eo=1 steps=3"),
    "traj_cloth_flat_8": ("traj", "cloth", "n=8 flat=1 eo=0 steps=3"),
    # the same beam with the reference's DirectLLT linear solver (exact Newton steps)
    "traj_tetbeam_llt_6x2x2": ("traj", "tetbeam", "nx=6 ny=2 nz=2 eo=1 steps=3 solver=llt"),
    # DirectLLT beyond the one-workgroup dense path (> 3072 unknowns: block-tridiagonal Cholesky): a 6 000-tet beam and configs[1] at full size
    "traj_tetbeam_llt_20x5x5": ("traj", "tetbeam", "nx=20 ny=5 nz=5 eo=0 steps=3 slim=1 solver=llt threads=8"),
    "traj_cfg1_tetbeam_llt_52x13x13": ("traj", "tetbeam", "nx=52 ny=13 nz=13 eo=0 steps=1 slim=1 solver=llt threads=8"),
    # configs[1] at FULL size (105 k tets, Soft_Rubber with damping + strain limiting, clamped end, no contact) and a mid-size configs[3]
    # (12 k-tet block on a fixed box, contact + friction): step log and final state only
    "traj_cfg1_tetbeam_52x13x13": ("traj", "tetbeam", "nx=52 ny=13 nz=13 eo=0 steps=3 slim=1 threads=8"),
    # the reference's Python smoke test (pystark/pystark/test_sim.py): hanging 32 x 32 cloth, 1 s = 30 steps
    "traj_cfg_pystark_hanging_cloth": ("traj", "pycloth", "n=32 steps=30 slim=1 threads=8"),
    # the reference's example hanging_deformable_box (examples/main.cpp:76-107): 12 k tets, 6 steps
    "traj_cfg_example_hanging_box": ("traj", "hangingbox", "n=10 steps=6 slim=1 threads=8"),
    "traj_cfg3_blockbox_10": ("traj", "blockbox", "nx=10 ny=10 nz=10 L=0.5 gap=0.002 thickness=0.002 bx=1.5 kmin=1e6 steps=4 boxfirst=1 slim=1 threads=8"),
    # configs[0] = the README's spinning box under a 32 x 32 cloth (README.md:53-95: thickness 2.5 mm, no friction, box turned 90 deg/s by
    # the per-step script), 10 frames: step log and final state
    "traj_cfg0_spinning_box_cloth_32": ("traj", "clothbox", "n=32 mu=0 spin=90 steps=10 slim=1 threads=8"),
    # configs[4] at fixture size: tet block on a fixed floor + cloth over it + chain of 4 hinged boxes over the cloth, contact + friction
    # between the layers (step log and final state)
    "traj_cfg4_mixed_small": ("traj", "mixed", "nx=4 ny=4 nz=4 nc=10 nrb=4 L=0.4 gap=0.003 thickness=0.002 bx=1.2 kmin=1e6 link=0.04 steps=5 slim=1 threads=8"),
    # BASELINE configs[3], [2], [4] at their FULL sizes: stage outputs of the reference at a closed-form state (`slimdump`: all velocity DoFs =
    # amp sin(1.3 i + 0.7) at the initial configuration): E, the whole gradient, the contact / friction tables, pattern size, SpMV probes
    # (every 4th entry, float32: the matrix is float) of the assembled and of the PSD-projected matrix, PCG outcomes on both
    "slim_cfg3_blockbox_44x44x43": ("slimdump", "blockbox", "nx=44 ny=44 nz=43 L=1 gap=0.0015 thickness=0.001 mu=0.5 kmin=1e8 bx=3 bz=0.1 boxfirst=1 threads=8 steps=0 amp=0.01 xamp=2e-4"),
    "slim_cfg2_clothbox_256": ("slimdump", "clothbox", "n=256 size=1 box=2 gap=0.0015 thickness=0.001 mu=0.5 threads=8 steps=0 amp=0.005 xamp=2e-4"),
    "slim_cfg4_mixed_26x26x25": ("slimdump", "mixed", "threads=8 steps=0 amp=0.01 xamp=2e-4"),
    # configs[3] at FULL size over its first time steps: the reference's own per-step log (Newton iterations, linear solves) with 8 and with 4
    # threads — on this placement the reference's runs are not reproducible from the fourth attempt on (21 linear solves there in most runs,
    # 17 in another with the same build and thread count; see the offset variant below for why) — and the per-iteration CG series + the end
    # state of its first time step (two attempts: the first one hardens the rigid-body constraint)
    "steplog_cfg3_blockbox_44x44x43": ("steplog", "blockbox", "nx=44 ny=44 nz=43 L=1 gap=0.0015 thickness=0.001 mu=0.5 kmin=1e8 bx=3 bz=0.1 boxfirst=1 steps=5"),
    # The same with the block moved 1.37 mm / -0.53 mm off the box's axes. Centred, the bottom-face nodes with x = -y lie exactly above the
    # diagonal edge of the box's top face and 78 edge-edge pairs sit on a classification tie (closest point exactly at an edge endpoint:
    # edge-edge or edge-point is decided by the last bit of a product). Off the axes the reference's 8- and 4-thread logs are identical
    # and reproducible.
    "steplog_cfg3_offset_44x44x43": ("steplog", "blockbox", "nx=44 ny=44 nz=43 L=1 gap=0.0015 thickness=0.001 mu=0.5 kmin=1e8 bx=3 bz=0.1 boxfirst=1 ox=0.00137 oy=-0.00053 steps=8"),
    # configs[2] as BASELINE describes it — a 256 x 256 Cotton_Fabric cloth DROPPED on a fixed floor from 5 cm (z = 0.05, 60 time
    # steps; free fall, impact around the fourth step, settling under IPC contact + friction): the reference's per-step log with 8 and with 4 threads
    "steplog_cfg2_clothbox_drop_256": ("steplog", "clothbox", "n=256 size=1 box=2 gap=0.05 thickness=0.001 mu=0.5 steps=60"),
    # configs[2] as a WELL-POSED dynamic scene (round 6): the same cloth tilted 3 degrees about y, its lowest edge 2 mm above the
    # floor (the contact distance), released. A tilted cloth straddles the floor's surface whenever it penetrates, so the reference's intersection
    # check sees it (the flat cloth of the 5 cm drop passes through between two steps). Logs of the reference with 8 and with 4 threads and the
    # sampled state after 10 steps of a third run (4 threads). The reference's runs of this scene are NOT reproducible: they either land in three
    # steps of 32 / 56-61 / 45-59 Newton iterations and rest from the ninth step on, or take 37-38 iterations in the first step and then fail
    # attempts and halve dt — which one a run takes changed between two invocations with the same thread count. The fixture holds both kinds.
    # tmcd::ProximityDetection::get_broad_phase_results() of the reference's OWN detector (tests/shim/shim_check.cpp `tmcd_broad`, the plain shim build
    # links the unmodified TriangleMeshCollisionDetection): candidate pairs of a tilted 7 x 7 cloth patch over a box, enlargement 4 mm, as sorted rows
    "tmcd_broad_phase_listing": ("shimcheck", "tmcd_broad", ""),
    "steplog_cfg2_tilted_256": ("steplog", "clothbox", "n=256 size=1 box=2 gap=0.002 tilt=3 thickness=0.001 mu=0.5 steps=30 traj_steps=10 traj_threads=4"),
    # what the reference WRITES for a run: its VTK frames, its YAML log and its run summary (tet beam, 2 time steps)
    "frames_tetbeam_4x1x1": ("frames", "tetbeam", "nx=4 ny=1 nz=1 eo=0 steps=2 frames=1"),
    # contact scenes (cfg 1 / cfg 4 at fixture size): cloth resting on a fixed rigid box, soft block pressed on a fixed rigid box
    "traj_clothbox_8": ("traj", "clothbox", "n=8 gap=0.004 steps=4"),
    "traj_blockbox_3": ("traj", "blockbox", "nx=3 ny=3 nz=3 L=0.2 gap=0.004 thickness=0.005 bx=0.5 kmin=1e5 steps=6 boxfirst=1"),
    # block registered first, no friction (with friction this order makes the reference's result depend on an unordered_map walk)
    # user-defined potentials (README.md:109-126: EnergyMagneticAttraction; the same with a summation loop, MappedWorkspace::add_for_each):
    # names the engine has no kernel for — through the shim they run SymX's op sequence on the device interpreter
    "traj_user_magnetic_3": ("traj", "magnetic", "n=3 k=20 steps=5 slim=1"),
    "traj_user_foreach_3": ("traj", "foreach", "n=3 k=20 steps=5 slim=1"),
    # The Newton driver's non-default branches (NewtonsMethod.cpp:254-386 _increase/_decrease_projection, :459-641 line search): the same beam
    # thrown about by start velocities of 40 m/s (`vamp`: inverted / indefinite elements in the first steps) under each projection mode —
    # Newton fails its steps and halves dt, ProjectedNewton projects everything, ProjectOnDemand counts down 4 iterations after a failure,
    # Progressive tightens and releases —, with mirroring, with a step cap ([cap]) and with a max_allowed_step callback ([max]); an
    # elasticity-only beam at dt = 0.2 that runs ProjectOnDemand through several countdown cycles; the block on the box under ProjectedNewton /
    # ProjectOnDemand / Newton, and thrown at the box (3 m/s: an invalid line-search candidate, [inv])
    "traj_tetbeam_big_progressive": ("traj", "tetbeam", "nx=8 ny=2 nz=2 eo=0 steps=3 slim=1 vamp=40 projection=Progressive"),
    "traj_tetbeam_big_newton": ("traj", "tetbeam", "nx=8 ny=2 nz=2 eo=0 steps=3 slim=1 vamp=40 projection=Newton"),
    "traj_tetbeam_big_projected": ("traj", "tetbeam", "nx=8 ny=2 nz=2 eo=0 steps=3 slim=1 vamp=40 projection=ProjectedNewton"),
    "traj_tetbeam_big_ondemand": ("traj", "tetbeam", "nx=8 ny=2 nz=2 eo=0 steps=3 slim=1 vamp=40 projection=ProjectOnDemand"),
    "traj_tetbeam_big_mirror": ("traj", "tetbeam", "nx=8 ny=2 nz=2 eo=0 steps=3 slim=1 vamp=40 projection=ProjectedNewton mirroring=1"),
    "traj_tetbeam_big_ondemand_eo": ("traj", "tetbeam", "nx=8 ny=2 nz=2 eo=1 steps=3 slim=1 vamp=15 dt=0.2 projection=ProjectOnDemand"),
    "traj_tetbeam_big_cap": ("traj", "tetbeam", "nx=8 ny=2 nz=2 eo=0 steps=3 slim=1 vamp=40 step_cap=8"),
    "traj_tetbeam_big_max": ("traj", "tetbeam", "nx=8 ny=2 nz=2 eo=0 steps=3 slim=1 vamp=40 maxstep=0.6"),
    "traj_blockbox_3_projected": ("traj", "blockbox", "nx=3 ny=3 nz=3 L=0.2 gap=0.004 thickness=0.005 bx=0.5 kmin=1e5 steps=6 boxfirst=1 slim=1 projection=ProjectedNewton"),
    "traj_blockbox_3_ondemand": ("traj", "blockbox", "nx=3 ny=3 nz=3 L=0.2 gap=0.004 thickness=0.005 bx=0.5 kmin=1e5 steps=6 boxfirst=1 slim=1 projection=ProjectOnDemand"),
    "traj_blockbox_3_newton": ("traj", "blockbox", "nx=3 ny=3 nz=3 L=0.2 gap=0.004 thickness=0.005 bx=0.5 kmin=1e5 steps=6 boxfirst=1 slim=1 projection=Newton"),
    "traj_blockbox_3_thrown": ("traj", "blockbox", "nx=3 ny=3 nz=3 L=0.2 gap=0.004 thickness=0.005 bx=0.5 kmin=1e5 steps=4 boxfirst=1 slim=1 vamp=3"),
    "traj_blockbox_3_nofriction": ("traj", "blockbox", "nx=3 ny=3 nz=3 L=0.2 gap=0.004 thickness=0.005 bx=0.5 kmin=1e5 steps=6 boxfirst=0 mu=0"),
}


# steplog_cfg3_nofma.npz is not made by this script's harness but by the SAME reference compiled with -ffp-contract=off (make -C oracle nofma;
# the command is in the fixture's `recipe` entry): the time-step logs of configs[3] on the centred and on the offset placement, see
# tests/test_oracle_golden.py::test_the_references_own_log_on_exact_ties_depends_on_its_compile_flags


def run(cmd):
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)


def pack(name):
    mode, scene, args = FIXTURES[name]
    args = args.split()
    if mode == "shimcheck":   # output of oracle/_ref/shim_check <scene> (the reference's classes driven directly), one JSON line
        exe = os.path.join(os.path.dirname(HARNESS), "shim_check")
        txt = subprocess.run([exe, scene] + args, check=True, capture_output=True).stdout.decode().strip().splitlines()[-1]
        json.loads(txt)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), listing_json=np.frombuffer(txt.encode(), dtype=np.uint8),
                            harness_args=np.frombuffer(("shim_check " + scene).encode(), dtype=np.uint8))
        print(name, "%.1f KB" % (os.path.getsize(os.path.join(OUT, name + ".npz")) / 1024))
        return
    tmp = tempfile.mkdtemp(prefix="mistark_fx_")
    try:
        scene_args = [a for a in args if not a.startswith(("steps=", "amp=", "xamp=", "frames=", "vamp=", "maxstep="))]
        if mode != "geom":
            run([HARNESS, "prime", scene] + scene_args)
        if mode == "steplog":
            # (traj_steps / traj_threads: how many steps, with how many threads, the run that dumps the sampled end state takes; default 2 / 8)
            traj_steps = next((a.split("=")[1] for a in args if a.startswith("traj_steps=")), "2")
            traj_threads = next((a.split("=")[1] for a in args if a.startswith("traj_threads=")), "8")
            args = [a for a in args if not a.startswith("traj_")]
            scene_args = [a for a in scene_args if not a.startswith("traj_")]
            data = {}
            for threads in (8, 4):
                out = subprocess.run([HARNESS, "time", scene] + args + ["warmup=0", "threads=%d" % threads, "outdir=" + tmp], check=True, capture_output=True).stdout.decode()
                txt = [l for l in out.splitlines() if l.startswith("{")][-1]
                json.loads(txt)
                data["time_t%d_json" % threads] = np.frombuffer(txt.encode(), dtype=np.uint8)
            run([HARNESS, "traj", scene] + scene_args + ["steps=" + traj_steps, "slim=1", "threads=" + traj_threads, "out=" + tmp])  # (the first attempt ends in an invalid converged state: constraint hardening, the step is redone)
            txt = open(os.path.join(tmp, "traj.json")).read()
            json.loads(txt)
            data["traj_json"] = np.frombuffer(txt.encode(), dtype=np.uint8)
            for k in ("x_end", "v_end"):
                data[k + "_every64"] = np.load(os.path.join(tmp, k + ".npy"))[::64]
            data["harness_args"] = np.frombuffer((mode + " " + scene + " " + " ".join(args)).encode(), dtype=np.uint8)
            np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
            print(name, "%.1f KB" % (os.path.getsize(os.path.join(OUT, name + ".npz")) / 1024))
            return
        if mode == "frames":
            out = subprocess.run([HARNESS, mode, scene] + args + ["outdir=" + tmp], check=True, capture_output=True).stdout
            open(os.path.join(tmp, "console.txt"), "wb").write(out)
        else:
            run([HARNESS, mode, scene] + args + ["out=" + tmp])
        data = {}
        for fn in sorted(os.listdir(tmp)):
            p = os.path.join(tmp, fn)
            if fn.endswith(".npy"):
                data[fn[:-4]] = np.load(p)
                if mode == "slimdump" and fn.startswith("spmv_y"):
                    data[fn[:-4]] = data[fn[:-4]][::4].astype(np.float32)
            elif mode == "frames" and fn.endswith((".vtk", ".yaml", ".txt")):
                key = fn.replace(".", "_") if not fn.endswith(".yaml") else "log_yaml"
                data[key] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
            elif fn.endswith(".json"):
                txt = open(p).read()
                json.loads(txt)  # validate
                data[fn[:-5] + "_json"] = np.frombuffer(txt.encode(), dtype=np.uint8)
        data["harness_args"] = np.frombuffer((mode + " " + scene + " " + " ".join(args)).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
        print(name, "%.1f KB" % (os.path.getsize(os.path.join(OUT, name + ".npz")) / 1024))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    names = sys.argv[1:] or list(FIXTURES)
    for n in names:
        pack(n)
