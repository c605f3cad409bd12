"""Oracle restatement of the filter line search (oracle ASSUMPTIONS A5 / A6): internal consistency on CPU."""
import numpy as np
import pytest

from test_oracle_lq import perturbed_problem

NX, NU = 58, 35


@pytest.fixture(scope="module")
def step(model, oracle):
    x0, x, u, par, dt = perturbed_problem(model, 6, "walk", seed=11)
    res = oracle.sqp_iteration(dt, x0, x, u, par, threads=4, want_proj=True)
    return dict(x0=x0, x=x, u=u, par=par, dt=dt, res=res)


def test_armijo_metric_is_the_projected_cost_gradient_along_the_step(oracle, step):
    """A5, basis-free form: sum_k (g_k + H_k [0; Pe_k]) . [dx_k; du_k - Pe_k] + g_N . dx_N."""
    s = step
    N = s["u"].shape[0]
    lq = oracle.lq(s["dt"], s["x"], s["u"], s["par"], threads=4)
    am = 0.0
    for k in range(N):
        pe = np.concatenate([np.zeros(NX), s["res"]["Pe"][k]])
        dz = np.concatenate([s["res"]["dx"][k], s["res"]["du"][k]]) - pe
        am += (lq["g"][k] + lq["H"][k] @ pe) @ dz
    xdes = s["par"][N, :NX]
    am += (np.array(oracle.model.desc.Qf[:]) * (s["x"][N] - xdes)) @ s["res"]["dx"][N]
    assert s["res"]["armijo"] == pytest.approx(am, rel=1e-9, abs=1e-9)
    assert s["res"]["armijo"] < 0.0   # the QP step is a descent direction of the projected cost


def _accept(st, base, new, alpha_armijo):
    g = np.sqrt(base["dynamics_sse"] + base["equality_sse"])
    gn = np.sqrt(new["dynamics_sse"] + new["equality_sse"])
    if gn > st["g_max"]:
        return gn < (1 - st["gamma_c"]) * g, 2
    if gn < st["g_min"] and g < st["g_min"] and alpha_armijo < 0:
        return new["merit"] < base["merit"] + st["armijo_factor"] * alpha_armijo, 0
    return (new["merit"] < base["merit"] - st["gamma_c"] * g) or (gn < (1 - st["gamma_c"]) * g), 1


@pytest.mark.parametrize("override", [{}, {"gamma_c": 0.6}, {"gamma_c": 0.9, "g_max": 1e-9, "g_min": 1e-12}])
def test_linesearch_accepts_the_first_admissible_step_length(oracle, step, override):
    s = step
    st = dict(oracle.LS_DEFAULTS)
    st.update(override)
    ls = oracle.linesearch(s["dt"], s["x"], s["u"], s["res"]["dx"], s["res"]["du"], s["par"], s["res"]["armijo"], threads=4, **override)
    base = oracle.performance(s["dt"], s["x"], s["u"], s["par"], threads=4)
    alpha, tried = 1.0, 0
    while True:
        new = oracle.performance(s["dt"], s["x"] + alpha * s["res"]["dx"], s["u"] + alpha * s["res"]["du"], s["par"], threads=4)
        tried += 1
        ok, typ = _accept(st, base, new, alpha * s["res"]["armijo"])
        if ok:
            assert ls["alpha"] == alpha and ls["step_type"] == typ and ls["trials"] == tried
            assert ls["perf"]["merit"] == pytest.approx(new["merit"], rel=1e-12)
            np.testing.assert_allclose(ls["x"], s["x"] + alpha * s["res"]["dx"], rtol=0, atol=1e-15)
            break
        alpha *= st["alpha_decay"]
        dxn, dun = np.sqrt((s["res"]["dx"] ** 2).sum()), np.sqrt((s["res"]["du"] ** 2).sum())
        if (alpha * dun < st["delta_tol"] and alpha * dxn < st["delta_tol"]) or alpha < st["alpha_min"]:
            assert ls["alpha"] == 0.0 and ls["step_type"] == 3 and ls["trials"] == tried
            np.testing.assert_array_equal(ls["x"], s["x"])
            np.testing.assert_array_equal(ls["u"], s["u"])
            assert ls["perf"]["merit"] == pytest.approx(base["merit"], rel=1e-12)
            break
    assert ls["trials"] <= 14


def test_full_step_is_accepted_on_the_default_walk_problem(oracle, step):
    s = step
    ls = oracle.linesearch(s["dt"], s["x"], s["u"], s["res"]["dx"], s["res"]["du"], s["par"], s["res"]["armijo"], threads=4)
    assert ls["alpha"] > 0.0
