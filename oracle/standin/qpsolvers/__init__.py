"""Stand-in for the third-party ``qpsolvers`` package (absent from this image).

TEST INFRASTRUCTURE, used only by ``oracle/gen_golden.py`` so that the reference
package can be imported in the build container.  It is our own code: one function
with the call signature the reference uses (``qplinear.py:83-85``,
``featlinearmap.py:375-381``) that solves the equality-constrained QP exactly
instead of with OSQP/SCS.  Solver options (``solver=``, ``eps_abs=`` ...) are
accepted and ignored.
"""
import numpy as np

from oracle.aggforce_oracle import eq_qp_solve


def _dense(x):
    if x is None:
        return None
    if hasattr(x, "toarray"):
        return np.asarray(x.toarray())
    return np.asarray(x)


def solve_qp(P, q, G=None, h=None, A=None, b=None, lb=None, ub=None, solver=None,
             initvals=None, verbose=False, **kwargs):
    if G is not None or h is not None or lb is not None or ub is not None:
        raise NotImplementedError("stand-in handles equality constraints only")
    return eq_qp_solve(_dense(P), _dense(q), _dense(A), _dense(b))
