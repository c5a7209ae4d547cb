"""GPU: the reference's own automated test suite, tests/rb_constraints.cpp (13 TEST_CASEs), against the MI355X engine.

Each case builds the same two-box scene through the host mirror of stark::RigidBodies, simulates until the constraint carries the applied
load in steady state and asserts the reference's physical invariants with the reference's tolerances: violation within the constraint's
tolerance, reaction force / torque within 1e-3 (relative) of the applied load (tests/rb_constraints.cpp:56-58 etc.).
Settings as the reference's (tests/rb_constraints.cpp:27-46: dt = 2 ms, DirectLLT, residual tolerance 1e-6, no step tolerance, no gravity, no
contact), except fixed MASS / PERTURBATION values instead of std::random_device (two draws of its range); the second case of every test
runs with the engine's default solver, the block-Jacobi PCG.
The simulated time is the reference's 3 s only for the cases that need it to reach steady state; the static ones stop at 1 s."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [(37.25, 52.5), (3.5, 91.0)]  # (MASS, PERTURBATION): rng() in [0, 100], PERTURBATION = rng() + 10
SOLVER = {37.25: 1, 3.5: 0}          # MISTARK_SOLVER_DIRECT_LLT for the first case, MISTARK_SOLVER_BDPCG for the second


def _sim(S, gravity=(0.0, 0.0, 0.0), mass=37.25):
    st = S.default_settings()
    st.newton.linear_solver = SOLVER[mass]
    st.gravity[0], st.gravity[1], st.gravity[2] = gravity
    st.init_frictional_contact = 0
    st.max_time_step_size = 0.002
    st.newton.residual_tolerance_abs = 1e-6
    st.newton.step_tolerance = 0.0
    return S.Simulation(st)


def _box(S, sim, mass, translation=None):
    b = sim.rb_add(mass, S.inertia_tensor_box(mass, (0.1, 0.1, 0.1)))
    if translation is not None:
        sim.rb_set_translation(b, translation)
    return b


def _within_rel(value, target, rel=1e-3):
    return abs(value - target) <= rel * max(abs(value), abs(target))


@pytest.mark.parametrize("mass,pert", CASES)
def test_inertia(mass, pert):
    from stark_amd import sim as S
    sim = _sim(S, gravity=(pert, 0.0, 0.0), mass=mass)
    box0 = _box(S, sim, mass)
    c = sim.rb_add_constraint("global_point", box0, -1, sim.rb_state(box0)[0])
    assert sim.run(1.0)
    C, f, tol = sim.rb_constraint_measure("global_point", c)
    assert abs(C) <= tol
    assert _within_rel(f, pert * mass)
    sim.close()


@pytest.mark.parametrize("mass,pert", CASES)
def test_global_point(mass, pert):
    from stark_amd import sim as S
    sim = _sim(S, mass=mass)
    box0 = _box(S, sim, mass)
    c = sim.rb_add_constraint("global_point", box0, -1, sim.rb_state(box0)[0])
    sim.rb_add_force_at_centroid(box0, (pert, 0, 0))
    assert sim.run(1.0)
    C, f, tol = sim.rb_constraint_measure("global_point", c)
    assert abs(C) <= tol
    assert _within_rel(f, pert)
    sim.close()


@pytest.mark.parametrize("mass,pert", CASES)
def test_global_direction(mass, pert):
    from stark_amd import sim as S
    sim = _sim(S, mass=mass)
    box0 = _box(S, sim, mass)
    c = sim.rb_add_constraint("global_direction", box0, -1, (0.0, 0.0, 1.0))
    sim.rb_add_torque(box0, (pert, 0, 0))
    assert sim.run(1.0)
    C, t, tol = sim.rb_constraint_measure("global_direction", c)
    assert abs(C) <= tol
    assert _within_rel(t, pert)
    sim.close()


def _fixed_pair(S, sim, mass, translation):
    box0 = _box(S, sim, mass)
    sim.rb_add_constraint("fix", box0)
    box1 = _box(S, sim, mass, translation)
    return box0, box1


@pytest.mark.parametrize("mass,pert", CASES)
def test_point(mass, pert):
    from stark_amd import sim as S
    sim = _sim(S, mass=mass)
    box0, box1 = _fixed_pair(S, sim, mass, (0.1, 0.0, 0.0))
    c = sim.rb_add_constraint("point", box0, box1, (0.05, 0.0, 0.0))
    sim.rb_add_force_at_centroid(box1, (pert, 0, 0))
    assert sim.run(1.0)
    C, f, tol = sim.rb_constraint_measure("point", c)
    assert abs(C) <= tol
    assert _within_rel(f, pert)
    sim.close()


@pytest.mark.parametrize("mass,pert", CASES)
def test_point_on_axis(mass, pert):
    from stark_amd import sim as S
    sim = _sim(S, mass=mass)
    box0, box1 = _fixed_pair(S, sim, mass, (0.1, 0.0, 0.0))
    c = sim.rb_add_constraint("point_on_axis", box0, box1, (0.0, 0.0, 0.0), (0.0, 0.0, 1.0))
    sim.rb_add_force_at_centroid(box1, (pert, 0, 0))
    assert sim.run(1.0)
    C, f, tol = sim.rb_constraint_measure("point_on_axis", c)
    assert abs(C) <= tol
    assert _within_rel(f, pert)
    sim.close()


@pytest.mark.parametrize("mass,pert", CASES)
def test_distance(mass, pert):
    from stark_amd import sim as S
    sim = _sim(S, mass=mass)
    box0, box1 = _fixed_pair(S, sim, mass, (1.0, 0.0, 0.0))
    c = sim.rb_add_constraint("distance", box0, box1, sim.rb_state(box0)[0], sim.rb_state(box1)[0])
    sim.rb_add_force_at_centroid(box1, (pert, 0, 0))
    assert sim.run(1.0)
    C, f, tol = sim.rb_constraint_measure("distance", c)
    assert abs(C) <= tol
    assert _within_rel(f, -pert)
    sim.close()


@pytest.mark.parametrize("mass,pert", CASES)
@pytest.mark.parametrize("sign", [1.0, -1.0], ids=["max", "min"])
def test_distance_limits(mass, pert, sign):
    from stark_amd import sim as S
    sim = _sim(S, mass=mass)
    box0, box1 = _fixed_pair(S, sim, mass, (1.0, 0.0, 0.0))
    c = sim.rb_add_constraint("distance_limits", box0, box1, sim.rb_state(box0)[0], sim.rb_state(box1)[0], 0.99, 1.01)
    sim.rb_add_force_at_centroid(box1, (sign * pert, 0, 0))
    assert sim.run(3.0)
    C, f, tol = sim.rb_constraint_measure("distance_limits", c)
    assert abs(C) <= tol
    assert _within_rel(f, -sign * pert)
    sim.close()


@pytest.mark.parametrize("mass,pert", CASES)
def test_direction(mass, pert):
    from stark_amd import sim as S
    sim = _sim(S, mass=mass)
    box0, box1 = _fixed_pair(S, sim, mass, (0.0, 0.0, 0.1))
    c = sim.rb_add_constraint("direction", box0, box1, (0.0, 0.0, 1.0))
    sim.rb_add_torque(box1, (pert, 0, 0))
    assert sim.run(1.0)
    C, t, tol = sim.rb_constraint_measure("direction", c)
    assert abs(C) <= tol
    assert _within_rel(t, pert)
    sim.close()


@pytest.mark.parametrize("mass,pert", CASES)
def test_angle_limit(mass, pert):
    from stark_amd import sim as S
    sim = _sim(S, mass=mass)
    box0, box1 = _fixed_pair(S, sim, mass, (0.0, 0.0, 0.1))
    c = sim.rb_add_constraint("angle_limit", box0, box1, (0.0, 0.0, 1.0), 25.0)
    sim.rb_add_torque(box1, (pert, 0, 0))
    assert sim.run(3.0)
    C, t, tol = sim.rb_constraint_measure("angle_limit", c)
    assert abs(C) <= tol
    assert _within_rel(t, pert)
    sim.close()


@pytest.mark.parametrize("mass,pert", CASES)
def test_spring(mass, pert):
    from stark_amd import sim as S
    sim = _sim(S, mass=mass)
    stiffness, perturbation, damping = 1000.0, 1.0, 1.0
    box0, box1 = _fixed_pair(S, sim, mass, (0.2, 0.0, 0.0))
    c = sim.rb_add_constraint("spring", box0, box1, sim.rb_state(box0)[0], sim.rb_state(box1)[0], stiffness, damping)
    sim.rb_add_force_at_centroid(box1, (perturbation, 0, 0))
    assert sim.run(3.0)
    dC, df, _ = sim.rb_constraint_measure("spring", c, which=1)   # damper
    assert abs(-dC * damping - df) <= 1e-3
    C, f, _ = sim.rb_constraint_measure("spring", c, which=0)     # spring
    assert _within_rel(-C * stiffness, f)
    sim.close()


@pytest.mark.parametrize("mass,pert", CASES)
def test_linear_velocity(mass, pert):
    from stark_amd import sim as S
    sim = _sim(S, mass=mass)
    max_force, target_v, delay = 50.0, 3.7, 0.01
    box0, box1 = _fixed_pair(S, sim, mass, (0.1, 0.0, 0.0))
    ball = sim.rb_add_constraint("point", box0, box1, (0.05, 0.0, 0.0))
    c = sim.rb_add_constraint("linear_velocity", box0, box1, (1.0, 0.0, 0.0), target_v, max_force, delay)
    assert sim.run(1.0)
    bC, bf, _ = sim.rb_constraint_measure("point", ball)
    C, f, _ = sim.rb_constraint_measure("linear_velocity", c)
    assert _within_rel(f, -bf)
    assert _within_rel(bf, max_force)
    sim.close()


@pytest.mark.parametrize("mass,pert", CASES)
def test_angular_velocity(mass, pert):
    from stark_amd import sim as S
    sim = _sim(S, mass=mass)
    max_torque, perturbation, delay = 10.0, 1.7, 0.01
    box0, box1 = _fixed_pair(S, sim, mass, (0.1, 0.0, 0.0))
    z_lock = sim.rb_constraint_count("direction")      # attachment = point + direction(z) + direction(x): get_z_lock()
    sim.rb_add_constraint("attachment", box0, box1)
    c = sim.rb_add_constraint("angular_velocity", box0, box1, (1.0, 0.0, 0.0), perturbation, max_torque, delay)
    assert sim.run(1.0)
    bC, bf, _ = sim.rb_constraint_measure("direction", z_lock)
    C, f, _ = sim.rb_constraint_measure("angular_velocity", c)
    assert _within_rel(f, -bf)
    assert _within_rel(bf, max_torque)
    sim.close()
