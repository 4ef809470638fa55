"""
Problem classes: equation strings -> (M, L, F) expressions.

Parsing follows dedalus/core/problems.py:35-108 (equations are Python expressions evaluated in
the user's namespace plus the operator names) and the IVP split of :321-364
(M.dt(X) + L.X = F).  Matrix *assembly* is not done per pencil here: see solvers.py / polyop.py.
"""

import numbers

import numpy as np

from . import operators as ops
from .field import Field, Operand


def _split_equation(eq):
    """Split 'LHS = RHS' at the top-level '=' (not '==', '<=', '>=', '!=' or keyword '=' inside calls)."""
    depth = 0
    for i, ch in enumerate(eq):
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == "=" and depth == 0:
            prev = eq[i - 1] if i > 0 else ""
            nxt = eq[i + 1] if i + 1 < len(eq) else ""
            if prev in "=<>!" or nxt == "=":
                continue
            return eq[:i].strip(), eq[i + 1:].strip()
    raise ValueError("Equation string must contain one top-level '=': %r" % eq)


class LinCtx:
    def __init__(self, variables, strict):
        self.variables = tuple(variables)
        self.strict = strict


class ProblemBase:
    def __init__(self, variables, namespace=None):
        self.variables = list(variables)
        self.dist = self.variables[0].dist
        self.equations = []
        self.namespace = {}
        # operator names (dedalus/core/operators.py parseables)
        for name in ("grad", "div", "lap", "trace", "dot", "cross", "skew", "transpose", "integ", "ave",
                     "interp", "lift", "curl", "dt"):
            self.namespace[name] = getattr(ops, name)
        self.namespace.update({"Gradient": ops.Gradient, "Divergence": ops.Divergence, "Laplacian": ops.Laplacian,
                               "Differentiate": ops.Differentiate, "Integrate": ops.Integrate,
                               "Interpolate": ops.Interpolate, "Average": ops.Average, "Lift": ops.Lift,
                               "Trace": ops.Trace, "TimeDerivative": ops.TimeDerivative,
                               "np": np, "numpy": np})
        if namespace:
            self.namespace.update(namespace)
        for v in self.variables:
            if v.name:
                self.namespace[v.name] = v

    def _parse(self, side):
        if isinstance(side, (Operand, numbers.Number)):
            return side
        return eval(side, dict(self.namespace))

    def add_equation(self, equation, condition=None):
        if isinstance(equation, str):
            lhs_s, rhs_s = _split_equation(equation)
            lhs, rhs = self._parse(lhs_s), self._parse(rhs_s)
        else:
            lhs, rhs = equation
            lhs, rhs = self._parse(lhs), self._parse(rhs)
        if not isinstance(lhs, Operand):
            raise ValueError("LHS must involve the problem variables")
        eq = self._build_equation(lhs, rhs)
        eq["string"] = equation if isinstance(equation, str) else None
        self.equations.append(eq)
        return eq

    def build_solver(self, *args, **kw):
        return self.solver_class(self, *args, **kw)


class InitialValueProblem(ProblemBase):
    """M.dt(X) + L.X = F(X, t)  (problems.py:321-364)."""

    def __init__(self, variables, time="t", namespace=None):
        super().__init__(variables, namespace)
        dist = self.dist
        if isinstance(time, Field):
            self.time = time
        else:
            self.time = Field(dist, name=time)
        self.sim_time_field = self.time
        self.namespace[self.time.name] = self.time

    @property
    def solver_class(self):
        from .solvers import InitialValueSolver
        return InitialValueSolver

    def _build_equation(self, lhs, rhs):
        ctx = LinCtx(self.variables, strict=True)
        try:
            le = lhs.lin(ctx)
        except ops.NonlinearOperatorError as e:
            raise ValueError("LHS must be linear in the problem variables: %s" % e)
        for leaf in le.leaves:
            if leaf not in self.variables:
                raise ValueError("LHS may only contain problem variables (found %r)" % (leaf,))
        L, M = le.split_dt()
        if isinstance(rhs, Operand) and rhs.has_dt():
            raise ValueError("time derivatives must be on the LHS")
        if isinstance(rhs, numbers.Number):
            F = None if rhs == 0 else ops._cast(rhs, self.dist)
        else:
            F = rhs
        if F is not None:
            if F.tensorsig != lhs.tensorsig:
                raise ValueError("LHS and RHS tensor signatures differ")
            F = ops.Convert(F, lhs.domain)     # RHS lives in the LHS bases (problems.py:339-345)
        return dict(lhs=lhs, domain=lhs.domain, tensorsig=lhs.tensorsig, ncomp=lhs.ncomp, M=M, L=L, F=F)


class LinearBoundaryValueProblem(ProblemBase):
    """L.X = F  (problems.py:LBVP); solved with the same batched pencil engine."""

    @property
    def solver_class(self):
        from .solvers import LinearBoundaryValueSolver
        return LinearBoundaryValueSolver

    def _build_equation(self, lhs, rhs):
        ctx = LinCtx(self.variables, strict=True)
        le = lhs.lin(ctx)
        L, M = le.split_dt()
        if not M.is_empty():
            raise ValueError("LBVP equations cannot contain time derivatives")
        if isinstance(rhs, numbers.Number):
            F = None if rhs == 0 else ops._cast(rhs, self.dist)
        else:
            F = rhs
        if F is not None:
            F = ops.Convert(F, lhs.domain)
        return dict(lhs=lhs, domain=lhs.domain, tensorsig=lhs.tensorsig, ncomp=lhs.ncomp, M=M, L=L, F=F)


IVP = InitialValueProblem
LBVP = LinearBoundaryValueProblem
