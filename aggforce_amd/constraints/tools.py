"""Host-side constraint bookkeeping (tiny, integer-only; stays on the CPU).

Mirrors the behaviour of the reference's ``constraints/tools.py`` (7-116): overlapping
constraint sets are merged into disjoint groups, and every group member points to the
group's smallest index (its anchor).  ``group_layout`` additionally produces the
device-side description of the reference's ``make_bond_constraint_matrix``
(``qp/qplinear.py:147-164``): which reduced column every atom maps to.
"""
from typing import Dict, List, Tuple

import numpy as np

from .hints import Constraints


def reduce_constraint_sets(constraints: Constraints) -> Constraints:
    """Merge constraint sets that share members into disjoint frozensets.

    Same result set as the reference (constraints/tools.py:7-77), computed with a
    union-find instead of its flood search.  Example: {{1,2},{2,3},{4,5}} ->
    {{1,2,3},{4,5}}.
    """
    parent: Dict[int, int] = {}

    def find(x: int) -> int:
        root = x
        while parent[root] != root:
            root = parent[root]
        while parent[x] != root:
            parent[x], x = root, parent[x]
        return root

    for group in constraints:
        members = list(group)
        for m in members:
            parent.setdefault(m, m)
        for m in members[1:]:
            ra, rb = find(members[0]), find(m)
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
    merged: Dict[int, set] = {}
    for m in parent:
        merged.setdefault(find(m), set()).add(m)
    return {frozenset(v) for v in merged.values()}


def constraint_lookup_dict(constraints: Constraints) -> Dict[int, int]:
    """member -> anchor (smallest member of its set); anchors themselves are absent.

    Reference: constraints/tools.py:80-116.
    """
    out: Dict[int, int] = {}
    for group in constraints:
        anchor = min(group)
        for s in group:
            if s != anchor:
                out[s] = anchor
    return out


def group_layout(n_sites: int, constraints: Constraints) -> Tuple[np.ndarray, int]:
    """Reduced-variable layout of make_bond_constraint_matrix (qp/qplinear.py:147-164).

    Returns ``(group_of_atom, n_red)``: atoms that are not a non-anchor member of a
    constraint group receive consecutive columns in atom order; every other member
    shares its anchor's column.  ``con_mat[a, group_of_atom[a]] == 1``.
    """
    lookup = constraint_lookup_dict(reduce_constraint_sets(constraints))
    for s in lookup:
        if not 0 <= s < n_sites or not 0 <= lookup[s] < n_sites:
            raise ValueError(f"constraint index {s} outside 0..{n_sites - 1}")
    goa = np.full(n_sites, -1, dtype=np.int32)
    free = np.ones(n_sites, dtype=bool)
    if lookup:
        members = np.fromiter(lookup.keys(), dtype=np.int64, count=len(lookup))
        anchors = np.fromiter(lookup.values(), dtype=np.int64, count=len(lookup))
        free[members] = False
    col = int(free.sum())
    goa[free] = np.arange(col, dtype=np.int32)
    if lookup:
        goa[members] = goa[anchors]  # (an anchor is never a non-anchor member: its column is set)
    return goa, col


def groups_csr(group_of_atom: np.ndarray, n_red: int) -> Tuple[np.ndarray, np.ndarray]:
    """CSR (grp_ptr[n_red+1], grp_atoms[N]) of the column -> atoms relation, atoms ascending."""
    order = np.argsort(group_of_atom, kind="stable").astype(np.int32)
    counts = np.bincount(group_of_atom, minlength=n_red)
    ptr = np.zeros(n_red + 1, dtype=np.int32)
    np.cumsum(counts, out=ptr[1:])
    return ptr, order


def group_lists(group_of_atom: np.ndarray, n_red: int) -> List[List[int]]:
    ptr, atoms = groups_csr(group_of_atom, n_red)
    return [atoms[ptr[g]:ptr[g + 1]].tolist() for g in range(n_red)]
