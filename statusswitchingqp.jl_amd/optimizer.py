"""Host-side mirror of the reference's plugin entry for the hot path -- StatusSwitchingQP.Optimizer
(src/MOIwrapper.jl): the dispatch of MOI.optimize! (:131-171) and the result getters that depend on the
solver's return triple (:189-251), and the INGEST half -- MOI.copy_to (:120-128), MOI2QP (:422-458) and
getConstraints (:262-349) -- on neutral input: the model arrives as plain term lists (what a
MOI.ScalarQuadraticFunction and the constraint list hold), not as a MathOptInterface object; the
MathOptInterface plumbing itself stays Julia (julia/SSQPHip.jl).

    opt = Optimizer(maxIter=500)          # kwargs go to Settings, unknown ones are rejected (:17-31)
    opt.copy_to(Q)                        # Q: a QP (what MOI.copy_to leaves in opt.Problem, :120-128)
    opt.copy_to_terms(N, quadratic_terms, affine_terms, constant, sense, constraints)   # ... or the model as terms
    opt.optimize()                        # MOI.optimize!
    opt.termination_status(), opt.primal_status(), opt.objective_value(), opt.variable_primal()
"""
import enum
import time

import numpy as np

from .solver import solveQP
from .types import DN, QP, Settings


class TerminationStatus(enum.Enum):
    """the MOI.TerminationStatusCode values MOIwrapper.jl:213-228 can return"""
    OPTIMIZE_NOT_CALLED = 0
    OPTIMAL = 1
    INFEASIBLE = 2
    INFEASIBLE_OR_UNBOUNDED = 3
    NUMERICAL_ERROR = 4
    ITERATION_LIMIT = 5


class ResultStatus(enum.Enum):
    NO_SOLUTION = 0
    FEASIBLE_POINT = 1
    INFEASIBLE_POINT = 2


MIN_SENSE, MAX_SENSE = "MIN_SENSE", "MAX_SENSE"


class Optimizer:
    """StatusSwitchingQP.Optimizer{Float64} (MOIwrapper.jl:8-35).  `qp_status_fix`: the reference maps Results[3]
    through the LP code table although solveQP returns the ITERATION COUNT there (a QP that
    converges in 3 passes reads INFEASIBLE_OR_UNBOUNDED, in >= 4 passes ITERATION_LIMIT).  Off by default = the
    reference's behaviour, bit for bit; on = status > 0 is OPTIMAL for a QP (the documented deviation, the same switch
    as use_moi_qp_status! in julia/SSQPHip.jl)."""

    def __init__(self, ctx=None, qp_status_fix=False, **user_settings):
        self.Problem = None
        self.Settings = Settings(**user_settings)      # TypeError on an unknown keyword, like Settings{T}(; kw...)
        self.Results = None
        self.Sense = MIN_SENSE
        self.Silent = True
        self.f0 = 0.0
        self.solTime = 0.01
        self.ctx = ctx
        self.qp_status_fix = bool(qp_status_fix)

    # -- MOI.empty! / is_empty (:44-52)
    def empty(self):
        self.Problem = None
        self.Results = None
        self.Sense = MIN_SENSE

    def is_empty(self):
        return self.Problem is None

    def solver_name(self):
        return "StatusSwitchingQP"          # :59

    def result_count(self):
        return int(self.Results is not None)  # :62

    def solve_time_sec(self):
        return self.solTime                  # :63

    # -- MOI.copy_to (:120-128) leaves a QP (or an LP, out of scope here) in opt.Problem
    def copy_to(self, problem, sense=MIN_SENSE, f0=0.0):
        if not isinstance(problem, QP):
            raise TypeError("only QP models are mirrored (the LP branch, SimplexLP, is out of scope)")
        self.Sense = sense
        self.f0 = float(f0)
        self.Problem = problem
        self.Results = None

    def copy_to_terms(self, N, quadratic_terms, affine_terms, constant=0.0, sense=MIN_SENSE, constraints=()):
        """MOI.copy_to(dest, src) (:120-128) for a model given as term lists (see moi_to_qp): sense first (:121), then
        MOI2QP (:122); a model whose V is all zero is an LP (:123-126) -- the reference rebuilds it as LP(...) for
        SimplexLP, which is outside this repository's scope, so it is REFUSED here."""
        self.Sense = sense
        Q, f0 = moi_to_qp(N, quadratic_terms, affine_terms, constant, sense, constraints)
        self.f0 = f0                                        # :445
        if np.abs(Q.V).max(initial=0.0) == 0:               # norm(V, Inf) == 0  (:123; entrywise for a Matrix)
            self.Problem = None
            self.Results = None
            raise UnsupportedModel("LP model (V == 0): the reference hands it to SimplexLP (MOIwrapper.jl:123-126,167), "
                                   "which is out of scope here")
        self.Problem = Q
        self.Results = None
        return Q

    # -- MOI.optimize! (:131-171)
    def optimize(self):
        P = self.Problem
        if P is None:
            raise RuntimeError("optimize: no model (MOI.copy_to has not run)")
        if P.mc == -20:                      # pre-solve "bad" models: no bounds and no inequalities (:133-160)
            N = P.N
            if P.M > 0:
                x = _backslash(P.A, P.b)     # x = P.A \ P.b  (:142)
                self.Results = (x, np.full(N, int(DN), dtype=np.int32), 1)
            else:                            # no constraints at all (:145-153)
                x = np.linalg.solve(P.V, P.q)
                det = np.linalg.det(P.V)
                st = 1 if ((self.Sense == MIN_SENSE and det > 0) or (self.Sense == MAX_SENSE and det < 0)) else 3
                self.Results = (x, np.full(N, int(DN), dtype=np.int32), st)
            return None
        t0 = time.time()
        z, S, status = solveQP(P, settings=self.Settings, ctx=self.ctx)   # :165 -> the GPU hot path
        self.Results = (z, S, int(status))
        self.solTime = time.time() - t0
        return None

    # -- result getters (:189-251)
    def dual_status(self, result_index=1):
        return ResultStatus.FEASIBLE_POINT if result_index == 1 else ResultStatus.NO_SOLUTION   # :190-192

    def primal_status(self, result_index=1):
        if result_index != 1 or self.Results is None:
            return ResultStatus.NO_SOLUTION
        return ResultStatus.INFEASIBLE_POINT if self.Results[2] == 0 else ResultStatus.FEASIBLE_POINT   # :201-206

    def raw_status_string(self):
        return str(self.Results[2])          # :209

    def termination_status(self):
        if self.Results is None:
            return TerminationStatus.OPTIMIZE_NOT_CALLED
        st = self.Results[2]
        if self.qp_status_fix and st > 0 and self.Problem is not None and self.Problem.mc != -20:
            return TerminationStatus.OPTIMAL
        if st == 3:                          # :217-227, the LP table applied to whatever Results[3] holds
            return TerminationStatus.INFEASIBLE_OR_UNBOUNDED
        if st in (1, 2):
            return TerminationStatus.OPTIMAL
        if st == 0:
            return TerminationStatus.INFEASIBLE
        if st == -1:
            return TerminationStatus.NUMERICAL_ERROR
        return TerminationStatus.ITERATION_LIMIT

    def objective_value(self):
        x = self.Results[0]
        f = float(x @ (self.Problem.V @ x) / 2 + self.Problem.q @ x)   # :234
        return (f if self.Sense == MIN_SENSE else -f) + self.f0        # :238

    def variable_primal(self, index=None):
        x = self.Results[0]
        return x if index is None else x[index]    # :243-250 (0-based here)


class UnsupportedConstraint(ValueError):
    """MOI.UnsupportedConstraint{F,S} (:298,317,320)"""


class UnsupportedModel(ValueError):
    pass


EQUAL_TO, GREATER_THAN, LESS_THAN, INTERVAL = "EqualTo", "GreaterThan", "LessThan", "Interval"


def get_constraints(N, constraints):
    """getConstraints(P, N, T) (MOIwrapper.jl:262-349) on a neutral constraint list.  Each constraint is
        ("affine", [(i, coef), ...], set, value)       a MOI.ScalarAffineFunction row, variables 0-based
        ("variable", i, set, value)                    a MOI.VariableIndex bound
    with set in {EqualTo, GreaterThan, LessThan, Interval} and value a number (a (lower, upper) pair for Interval).
    As in the reference: a row's coefficients are ASSIGNED per term (a repeated variable keeps its LAST coefficient,
    :274-276 -- not summed, unlike the objective), a row without terms is skipped with a warning (:278-282),
    `>=` rows are negated into `<=` rows (:286-288), Interval is accepted for variables only, EqualTo for rows only;
    variables start unbounded, d = -Inf, u = +Inf (:266-267).  Returns A, b, G, g, d, u."""
    import warnings
    Ab, Gg = [], []
    d = np.full(N, -np.inf)
    u = np.full(N, np.inf)
    for con in constraints:
        kind, f, S, val = con
        if kind == "affine":
            t = np.zeros(N + 1)
            nt = 0
            for i, coef in f:
                t[i] = coef
                nt += 1
            if nt == 0:
                warnings.warn("skipping redundant rows")
                continue
            if S == EQUAL_TO:
                t[-1] = val
                Ab.append(t)
            elif S == GREATER_THAN:
                t[-1] = val
                Gg.append(-t)
            elif S == LESS_THAN:
                t[-1] = val
                Gg.append(t)
            else:
                raise UnsupportedConstraint("ScalarAffineFunction-in-" + str(S))
        elif kind == "variable":
            if S == GREATER_THAN:
                d[f] = val
            elif S == LESS_THAN:
                u[f] = val
            elif S == INTERVAL:
                d[f], u[f] = val
            else:
                raise UnsupportedConstraint("VariableIndex-in-" + str(S))
        else:
            raise UnsupportedConstraint(str(kind))
    M, J = len(Ab), len(Gg)
    A = np.array([t[:-1] for t in Ab]).reshape(M, N)
    b = np.array([t[-1] for t in Ab]).reshape(M)
    G = np.array([t[:-1] for t in Gg]).reshape(J, N)
    g = np.array([t[-1] for t in Gg]).reshape(J)
    return A, b, G, g, d, u


def moi_to_qp(N, quadratic_terms, affine_terms, constant=0.0, sense=MIN_SENSE, constraints=()):
    """MOI2QP (MOIwrapper.jl:422-458) on neutral input: the objective f = (1/2) z'Vz + q'z + f0 as the term lists of a
    MOI.ScalarQuadraticFunction -- quadratic_terms [(i, j, coef), ...], affine_terms [(i, coef), ...], 0-based.
    As in the reference: duplicate terms are SUMMED (:431-433, :442-444); then every off-diagonal pair is folded,
    V[i,j] += V[j,i] for i > j, and mirrored (:434-439) -- MathOptInterface lists an off-diagonal term once, standing for
    both V[i,j] and V[j,i]; MAX_SENSE negates V and q (:447-450); the constant goes to f0 (:445).  Returns (QP, f0)."""
    V = np.zeros((N, N))
    for i, j, coef in quadratic_terms:
        V[i, j] += coef
    for i in range(1, N):
        for j in range(i):
            V[i, j] += V[j, i]
            V[j, i] = V[i, j]
    q = np.zeros(N)
    for i, coef in affine_terms:
        q[i] += coef
    if sense == MAX_SENSE:
        V = -V
        q = -q
    A, b, G, g, d, u = get_constraints(N, constraints)
    return QP(V, A=A, G=G, q=q, b=b, g=g, d=d, u=u), float(constant)


def _backslash(A, b):
    """Julia's A \\ b for a dense A: exact solve when square, least squares otherwise (:142)"""
    A = np.asarray(A, dtype=np.float64)
    if A.shape[0] == A.shape[1]:
        return np.linalg.solve(A, b)
    return np.linalg.lstsq(A, b, rcond=None)[0]
