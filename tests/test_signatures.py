"""The product's public call surface against the reference's.

``tests/golden/g9_signatures.json`` was written by ``oracle/gen_signatures.py`` from the reference's own ``def``
statements (names, order, kinds, defaults of every public callable; methods and properties of every public class).
A user switching from ``aggforce`` to ``aggforce_amd`` keeps their call sites: every positional order and every
keyword name the reference accepts must be accepted here, with the same defaults.  CPU only: nothing is computed.
"""
import ast
import importlib
import inspect
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "g9_signatures.json")) as fh:
    SURFACE = json.load(fh)

_POSITIONAL = ("positional_only", "positional_or_keyword")


def _product_namespace(ref_ns: str):
    return importlib.import_module(ref_ns.replace("aggforce", "aggforce_amd", 1))


def _cases():
    for ns, entries in sorted(SURFACE.items()):
        for name, desc in sorted(entries.items()):
            yield pytest.param(ns, name, desc, id=f"{ns}.{name}")


def _ref_params(params, drop_first: bool):
    return params[1:] if drop_first and params and params[0]["name"] in ("self", "cls") else params


def _check_callable(label: str, ref_params, prod_callable, prod_module) -> None:
    sig = inspect.signature(prod_callable)
    prod = list(sig.parameters.values())
    prod_pos = [p for p in prod if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    ref_pos = [p for p in ref_params if p["kind"] in _POSITIONAL]
    # (1) positional order and names
    assert len(prod_pos) >= len(ref_pos) or any(p.kind == p.VAR_POSITIONAL for p in prod), \
        f"{label}: takes {len(prod_pos)} positional parameters, the reference {len(ref_pos)}"
    for i, rp in enumerate(ref_pos):
        if i >= len(prod_pos):
            break
        if rp["kind"] == "positional_or_keyword":
            assert prod_pos[i].name == rp["name"], \
                f"{label}: positional #{i} is '{prod_pos[i].name}', the reference's is '{rp['name']}'"
            assert prod_pos[i].kind == prod_pos[i].POSITIONAL_OR_KEYWORD, f"{label}: '{rp['name']}' is not a keyword here"
    # (2) keyword names, var-args
    by_name = {p.name: p for p in prod}
    has_var_kw = any(p.kind == p.VAR_KEYWORD for p in prod)
    for rp in ref_params:
        if rp["kind"] == "keyword_only":
            assert rp["name"] in by_name or has_var_kw, f"{label}: keyword '{rp['name']}' is not accepted"
            if rp["name"] in by_name:
                assert by_name[rp["name"]].kind in (inspect.Parameter.KEYWORD_ONLY, inspect.Parameter.POSITIONAL_OR_KEYWORD)
        if rp["kind"] == "var_positional":
            assert any(p.kind == p.VAR_POSITIONAL for p in prod), f"{label}: *{rp['name']} is not accepted"
        if rp["kind"] == "var_keyword":
            assert has_var_kw, f"{label}: **{rp['name']} is not accepted"
    # (3) defaults: what is optional in the reference is optional here, with the same value where it is a literal
    #     (or a module-level name both sides define, e.g. DEFAULT_SOLVER_OPTIONS)
    for rp in ref_params:
        if rp["default"] is None or rp["name"] not in by_name:
            continue
        pp = by_name[rp["name"]]
        assert pp.default is not inspect.Parameter.empty, f"{label}: '{rp['name']}' is required here, optional in the reference"
        try:
            want = ast.literal_eval(rp["default"])
        except (ValueError, SyntaxError):
            want = getattr(prod_module, rp["default"], None) if rp["default"].isidentifier() else None
            if want is None or rp["default"].startswith("_"):
                continue  # sentinels (_UNSET), np.float32 spelled differently, ...: being optional is the contract
        if isinstance(want, float) and want != want:
            assert pp.default != pp.default, f"{label}: default of '{rp['name']}' is {pp.default!r}, reference NaN"
        else:
            assert pp.default == want or pp.default is want, \
                f"{label}: default of '{rp['name']}' is {pp.default!r}, the reference's is {rp['default']}"
    # (4) whatever the product adds must not be required: a reference call must still bind
    ref_names = {rp["name"] for rp in ref_params}
    for p in prod:
        if p.name not in ref_names and p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY):
            if p.kind != p.KEYWORD_ONLY and prod_pos.index(p) < len(ref_pos):
                continue  # renamed positional-only parameter of the reference
            assert p.default is not inspect.Parameter.empty, f"{label}: extra parameter '{p.name}' has no default"
    # (5) the two call forms of the reference bind: all-positional and all-keyword
    sentinel = object()
    kw_required = {rp["name"]: sentinel for rp in ref_params if rp["kind"] == "keyword_only" and rp["default"] is None}
    sig.bind(*[sentinel for rp in ref_pos if rp["default"] is None], **kw_required)
    sig.bind(*[sentinel for _ in ref_pos], **kw_required)
    kw = {rp["name"]: sentinel for rp in ref_params if rp["kind"] in ("positional_or_keyword", "keyword_only")}
    po = [sentinel for rp in ref_params if rp["kind"] == "positional_only"]
    sig.bind(*po, **kw)


@pytest.mark.parametrize("ns,name,desc", list(_cases()))
def test_public_name_matches_reference(ns, name, desc) -> None:
    module = _product_namespace(ns)
    assert hasattr(module, name), f"{module.__name__} lacks the reference's public name '{name}'"
    obj = getattr(module, name)
    label = f"{module.__name__}.{name}"
    if desc["type"] == "value":
        return
    defining = inspect.getmodule(obj) or module
    if desc["type"] == "function":
        assert callable(obj), f"{label} is not callable"
        _check_callable(label, desc["params"], obj, defining)
        return
    assert inspect.isclass(obj), f"{label} is a class in the reference"
    for prop in desc["properties"]:
        assert hasattr(obj, prop) or prop in getattr(obj, "__annotations__", {}), f"{label}: no property '{prop}'"
    for mname, m in desc["methods"].items():
        if mname == "__init__":
            _check_callable(f"{label}()", _ref_params(m["params"], True), obj, defining)
            continue
        assert hasattr(obj, mname), f"{label}: no method '{mname}'"
        attr = inspect.getattr_static(obj, mname)
        if m["kind"] == "method":
            fn = attr if inspect.isfunction(attr) else getattr(obj, mname)
            params = m["params"]  # (self included on both sides)
        else:
            fn = getattr(obj, mname)  # bound classmethod / plain staticmethod
            params = _ref_params(m["params"], m["kind"] == "classmethod")
        _check_callable(f"{label}.{mname}", params, fn, defining)


def test_the_reference_module_paths_exist() -> None:
    """``defined_in`` of the fixture: the modules users import from directly keep their names (aliases where the
    product's file is named differently: jaxgausstraj -> gausstraj, jaxfeat -> gbfeat, jgauss -> gauss)."""
    seen = set()
    for entries in SURFACE.values():
        for name, desc in entries.items():
            rel = desc["defined_in"]
            if rel.endswith("__init__.py") or desc["type"] == "value":
                continue
            seen.add((rel[:-3].replace("/", "."), name))
    for mod, name in sorted(seen):
        m = importlib.import_module("aggforce_amd." + mod)
        assert hasattr(m, name), f"aggforce_amd.{mod} lacks '{name}'"


def test_jcondnormal_binds_the_reference_call_sites() -> None:
    """The literal constructor calls of qp/jgauss.py:118,237,282-286,383-387,427-430 bind (no compute: a LinearMap's
    flat_call and a LinearMap product are recognised without being called)."""
    import numpy as np

    from aggforce_amd import LinearMap
    from aggforce_amd.trajectory import JCondNormal

    cmap = LinearMap([[0, 1], [2]], n_fg_sites=4)
    a = JCondNormal(cov=0.01, premap=cmap.flat_call, seed=3)
    assert a.premap_map() is cmap and a.dtype == np.float32 and a.var == 0.01 and a.cov is None
    fmap = LinearMap(np.eye(2))
    b = JCondNormal(cov=0.3, source_postmap=(fmap @ cmap @ cmap.T), seed=None)
    assert b.source_postmap_map().standard_matrix.shape == (2, 2) and b.premap_map() is None
    c = JCondNormal(0.3, cmap.flat_call, None, 7, np.float64)  # positional order of jaxgausstraj.py:140-146
    assert c.seed == 7 and c.dtype == np.float64 and c.source_postmap_map() is None
    assert c.astype(np.float32).dtype == np.float32
    full = JCondNormal(cov=np.diag(np.full(6, 0.25)), premap=cmap.flat_call)
    assert full.dtype == np.float64 and full.var is None and full.cov.shape == (6, 6)
    with pytest.raises(ValueError):
        JCondNormal(cov=np.array([[1.0, 2.0, 0], [2.0, 1.0, 0], [0, 0, 1.0]]))  # not positive definite
    with pytest.raises(ValueError):
        full.to_SimpleCondNormal()
    with pytest.raises(ValueError):
        a.to_SimpleCondNormal()  # premap is not the identity
    assert JCondNormal(0.5).to_SimpleCondNormal().var == 0.5
    with pytest.raises(TypeError):
        JCondNormal()
