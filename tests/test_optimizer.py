"""The host-side mirror of the reference's plugin entry (StatusSwitchingQP.Optimizer, src/MOIwrapper.jl:131-171 dispatch
and :213-228 status table) through the C ABI: a CPU test of everything that needs no GPU and one -m gpu test."""
import numpy as np
import pytest


def test_optimizer_presolve_and_status_table_cpu(pkg):
    T = pkg.TerminationStatus
    with pytest.raises(TypeError):
        pkg.Optimizer(nonsense=1)                       # unknown Settings keyword (MOIwrapper.jl:17-31)
    opt = pkg.Optimizer(maxIter=99)
    assert opt.Settings.maxIter == 99 and opt.is_empty() and opt.result_count() == 0
    assert opt.termination_status() == T.OPTIMIZE_NOT_CALLED and opt.primal_status() == pkg.ResultStatus.NO_SOLUTION
    # mc == -20 (no bounds, no inequalities): closed-form presolve, never reaches solveQP (:133-160)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Q = pkg.QP(np.diag([2.0, 4.0]), q=np.array([1.0, 1.0]), d=np.full(2, -np.inf), A=np.zeros((0, 2)), b=np.zeros(0))
    assert Q.mc == -20 and Q.M == 0
    opt.copy_to(Q)
    opt.optimize()
    x, S, st = opt.Results
    assert np.allclose(x, [0.5, 0.25]) and (S == pkg.DN).all() and st == 1      # x = V \ q, det(V) > 0
    assert opt.termination_status() == T.OPTIMAL and opt.result_count() == 1
    opt.copy_to(Q, sense="MAX_SENSE")
    opt.optimize()
    assert opt.Results[2] == 3 and opt.termination_status() == T.INFEASIBLE_OR_UNBOUNDED
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Q2 = pkg.QP(np.eye(2), d=np.full(2, -np.inf), A=np.array([[1.0, 1.0], [1.0, -1.0]]), b=np.array([1.0, 0.0]))
    opt.copy_to(Q2)
    opt.optimize()                                                               # x = A \ b  (:142)
    assert np.allclose(opt.Results[0], [0.5, 0.5]) and opt.Results[2] == 1
    # the status table (:213-228) on every kind of Results[3], with and without the documented QP fix
    Qd = pkg.QP(np.eye(3))
    for fix in (False, True):
        o = pkg.Optimizer(qp_status_fix=fix)
        o.copy_to(Qd)
        for st, want in ((1, T.OPTIMAL), (2, T.OPTIMAL), (0, T.INFEASIBLE), (-1, T.NUMERICAL_ERROR),
                         (-7778, T.ITERATION_LIMIT), (3, T.OPTIMAL if fix else T.INFEASIBLE_OR_UNBOUNDED),
                         (17, T.OPTIMAL if fix else T.ITERATION_LIMIT)):
            o.Results = (np.zeros(3), np.zeros(3, dtype=np.int32), st)
            assert o.termination_status() == want, (fix, st)
        o.Results = (np.zeros(3), np.zeros(3, dtype=np.int32), 0)
        assert o.primal_status() == pkg.ResultStatus.INFEASIBLE_POINT


def test_moi_to_qp_mirrors_the_reference_ingest_cpu(pkg):
    """MOI2QP (MOIwrapper.jl:422-458), getConstraints (:262-349) and copy_to's LP refusal (:123-126) on hand-built term
    lists: duplicate objective terms summed, off-diagonals folded and mirrored, MAX_SENSE negation, f0; rows assigned
    per term, `>=` rows negated, bounds from VariableIndex sets, defaults d = -Inf / u = +Inf"""
    import warnings
    # f = (1/2) z'Vz + q'z + 3 with V = [[2, 1.5, 0], [1.5, 4, -1], [0, -1, 6]]: the (0,1) pair given ONCE as 1.5 (MOI's
    # convention), the (2,1) pair split over both orders, the diagonal (1,1) as two duplicate terms
    quad = [(0, 0, 2.0), (0, 1, 1.5), (1, 1, 1.0), (1, 1, 3.0), (2, 1, -0.25), (1, 2, -0.75), (2, 2, 6.0)]
    aff = [(0, 1.0), (2, -2.0), (0, 0.5)]
    cons = [("affine", [(0, 1.0), (1, 1.0), (2, 1.0)], "EqualTo", 1.0),
            ("affine", [(0, 1.0), (2, 2.0)], "GreaterThan", 0.25),
            ("affine", [(1, 1.0), (1, 3.0)], "LessThan", 0.9),          # repeated variable: the LAST coefficient stays
            ("affine", [], "LessThan", 5.0),                            # no terms: skipped with a warning
            ("variable", 0, "GreaterThan", 0.0), ("variable", 1, "Interval", (0.1, 0.8)), ("variable", 2, "LessThan", 0.7)]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        Q, f0 = pkg.moi_to_qp(3, quad, aff, 3.0, "MIN_SENSE", cons)
    assert any("redundant" in str(x.message) for x in w)
    assert np.array_equal(Q.V, [[2, 1.5, 0], [1.5, 4, -1], [0, -1, 6]]) and Q.q.tolist() == [1.5, 0, -2] and f0 == 3.0
    assert (Q.N, Q.M, Q.J, Q.mc) == (3, 1, 2, 1)
    assert np.array_equal(Q.A, [[1, 1, 1]]) and Q.b.tolist() == [1.0]
    assert np.array_equal(Q.G, [[-1, 0, -2], [0, 3, 0]]) and Q.g.tolist() == [-0.25, 0.9]
    assert Q.d.tolist() == [0.0, 0.1, -np.inf] and Q.u.tolist() == [np.inf, 0.8, 0.7]
    Qm, _ = pkg.moi_to_qp(3, quad, aff, 0.0, "MAX_SENSE", cons[:1] + cons[4:])
    assert np.array_equal(Qm.V, -Q.V) and np.array_equal(Qm.q, -Q.q) and Qm.J == 0 and Qm.mc == -70   # (a concave model)
    with pytest.raises(pkg.UnsupportedConstraint):
        pkg.get_constraints(2, [("affine", [(0, 1.0)], "Interval", (0.0, 1.0))])       # :298
    with pytest.raises(pkg.UnsupportedConstraint):
        pkg.get_constraints(2, [("variable", 0, "EqualTo", 1.0)])                      # :317
    A, b, G, g, d, u = pkg.get_constraints(2, [])
    assert A.shape == (0, 2) and G.shape == (0, 2) and (d == -np.inf).all() and (u == np.inf).all()
    opt = pkg.Optimizer()
    with pytest.raises(pkg.UnsupportedModel):                                           # norm(V, Inf) == 0 -> LP (:123-126)
        opt.copy_to_terms(2, [], [(0, 1.0)], 0.0, "MIN_SENSE", [("affine", [(0, 1.0), (1, 1.0)], "EqualTo", 1.0),
                                                                   ("variable", 0, "GreaterThan", 0.0)])
    assert opt.is_empty()
    Q2 = opt.copy_to_terms(3, quad, aff, 3.0, "MIN_SENSE", cons)
    assert opt.Problem is Q2 and opt.f0 == 3.0 and opt.Sense == "MIN_SENSE" and not opt.is_empty()


@pytest.mark.gpu
def test_optimizer_from_terms_end_to_end_on_the_gpu(pkg):
    """the reference's own fixture (test/runtests.jl:23-32) expressed as MathOptInterface-style term lists ->
    copy_to_terms (MOI2QP + getConstraints) -> optimize (solveQP on the GPU) -> the getters"""
    Vd = [[1 / 100, 1 / 80, 1 / 100], [1 / 80, 1 / 16, 1 / 40], [1 / 100, 1 / 40, 1 / 25]]
    quad = [(i, j, Vd[i][j]) for i in range(3) for j in range(i, 3)]           # upper triangle: each pair once
    cons = [("affine", [(0, 1.0), (1, 1.0), (2, 1.0)], "EqualTo", 1.0),
            ("variable", 0, "Interval", (0.0, 0.7)), ("variable", 1, "GreaterThan", 0.0), ("variable", 2, "Interval", (0.0, 0.7))]
    opt = pkg.Optimizer(qp_status_fix=True)
    Q = opt.copy_to_terms(3, quad, [], 0.25, "MIN_SENSE", cons)
    assert np.array_equal(Q.V, np.array(Vd)) and Q.mc == 1
    opt.optimize()
    z, S, it = opt.Results
    assert S.tolist() == [pkg.UP, pkg.IN, pkg.IN] and it == 2
    np.testing.assert_allclose(opt.variable_primal(), [0.7, 11 / 210, 52 / 210], rtol=1e-12)
    assert opt.termination_status() == pkg.TerminationStatus.OPTIMAL
    assert abs(opt.objective_value() - (z @ np.array(Vd) @ z / 2 + 0.25)) < 1e-15


@pytest.mark.gpu
def test_optimizer_runs_the_reference_kat_on_the_gpu(pkg):
    """MOI.optimize! -> solveQP(Q; settings) (MOIwrapper.jl:165) through the C ABI on the reference's own fixture
    (test/runtests.jl:23-32) and on a problem that needs more than 3 passes (where the reference's status table
    misreports a QP)"""
    T = pkg.TerminationStatus
    V = np.array([[1 / 100, 1 / 80, 1 / 100], [1 / 80, 1 / 16, 1 / 40], [1 / 100, 1 / 40, 1 / 25]])
    opt = pkg.Optimizer()
    opt.copy_to(pkg.QP(V, u=np.array([0.7, np.inf, 0.7])))
    opt.optimize()
    z, S, it = opt.Results
    assert S.tolist() == [pkg.UP, pkg.IN, pkg.IN] and it == 2
    assert opt.termination_status() == T.OPTIMAL and opt.primal_status() == pkg.ResultStatus.FEASIBLE_POINT
    assert abs(opt.objective_value() - (z @ V @ z / 2)) < 1e-15 and opt.solve_time_sec() > 0
    np.testing.assert_allclose(opt.variable_primal(), [0.7, 11 / 210, 52 / 210], rtol=1e-12)
    cfg = pkg.GenConfig(24, 1, 2, 48, 1e-3, 0.2, 1.05, 0.1)
    prob = pkg.generate_batch(cfg, 1, 7)
    Q = pkg.QP.inner(prob["V"][0], prob["A"][0].reshape(24, 1).T, prob["G"][0].reshape(24, 2).T, prob["q"][0], prob["b"][0],
                     prob["g"][0], prob["d"][0], prob["u"][0])
    for fix, want in ((False, T.ITERATION_LIMIT), (True, T.OPTIMAL)):
        o = pkg.Optimizer(qp_status_fix=fix)
        o.copy_to(Q)
        o.optimize()
        assert o.Results[2] > 3
        assert o.termination_status() == want
