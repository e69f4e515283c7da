"""Generate tests/golden/g9_signatures.json: the PUBLIC CALL SURFACE of the reference.

TEST INFRASTRUCTURE.  Runs only in the build container, where /root/reference is mounted.  For every name a
package ``__init__`` of the reference exports (``aggforce``, ``aggforce.qp``, ``.map``, ``.trajectory``,
``.constraints``), and for the module-level helpers the path uses directly (``agg.py``, ``util.py``,
``trajectory/simplegausstraj.py``), it records

    functions:  parameter names, kinds, order and defaults
    classes:    the same for ``__init__`` and every public method, plus the public properties

read from the ``def`` statements with ``ast`` -- the modules are parsed, not imported, so the JAX-based ones
(``qp/jaxfeat.py``, ``qp/jgauss.py``, ``trajectory/jaxgausstraj.py``) are covered as well.  The fixture is interface
metadata (names, order, default values): no statement of the reference's code enters the repository.
``tests/test_signatures.py`` asserts that each product callable accepts the reference's positional order and
keyword names.

    python oracle/gen_signatures.py
"""
import ast
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PKG = "/root/reference/src/aggforce"
if not os.path.isdir(REF_PKG):
    raise SystemExit("gen_signatures.py needs the reference mounted at /root/reference")

PACKAGES = ["", "qp", "map", "trajectory", "constraints"]
# module-level names reached directly by users of the path (not re-exported by a package __init__)
EXTRA = {
    "agg.py": ["project_forces", "project_forces_grid_cv", "force_smoothness"],
    "util.py": ["trjdot", "distances"],
    "trajectory/simplegausstraj.py": ["SimpleCondNormal"],
}
# out of scope (JAX re-implementations of LinearMap and the validation helpers)
OUT_OF_SCOPE = {"jaxify_linearmap", "JLinearMap"}


def params_of(fn: ast.FunctionDef):
    a = fn.args
    out = []
    pos = [(p, "positional_only") for p in a.posonlyargs] + [(p, "positional_or_keyword") for p in a.args]
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    for (p, kind), d in zip(pos, defaults):
        out.append({"name": p.arg, "kind": kind, "default": None if d is None else ast.unparse(d)})
    if a.vararg is not None:
        out.append({"name": a.vararg.arg, "kind": "var_positional", "default": None})
    for p, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append({"name": p.arg, "kind": "keyword_only", "default": None if d is None else ast.unparse(d)})
    if a.kwarg is not None:
        out.append({"name": a.kwarg.arg, "kind": "var_keyword", "default": None})
    return out


def describe(node):
    if isinstance(node, ast.FunctionDef):
        return {"type": "function", "params": params_of(node)}
    if isinstance(node, ast.ClassDef):
        methods, props = {}, []
        for item in node.body:
            if not isinstance(item, ast.FunctionDef):
                continue
            decos = {ast.unparse(d) for d in item.decorator_list}
            if any(d.endswith(".setter") for d in decos):
                continue
            public = not item.name.startswith("_") or item.name in ("__init__", "__call__", "__getitem__",
                                                                     "__matmul__", "__rmul__", "__add__", "__len__")
            if not public:
                continue
            if "property" in decos:
                props.append(item.name)
            else:
                kind = "classmethod" if "classmethod" in decos else "staticmethod" if "staticmethod" in decos else "method"
                methods[item.name] = {"kind": kind, "params": params_of(item)}
        return {"type": "class", "bases": [ast.unparse(b) for b in node.bases], "methods": methods,
                "properties": sorted(props)}
    if isinstance(node, (ast.Assign, ast.AnnAssign)):
        return {"type": "value"}
    return None


def module_defs(relpath):
    with open(os.path.join(REF_PKG, relpath)) as fh:
        tree = ast.parse(fh.read())
    defs = {}
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            defs[node.name] = node
        elif isinstance(node, ast.Assign):
            for tgt in node.targets:
                if isinstance(tgt, ast.Name):
                    defs[tgt.id] = node
        elif isinstance(node, ast.AnnAssign) and isinstance(node.target, ast.Name):
            defs[node.target.id] = node
    return tree, defs


def imports_of(rel):
    """{name bound by a relative ImportFrom of module ``rel``: (module file, original name)} -- also the imports
    inside try/except ImportError (the optional JAX parts)."""
    tree, _ = module_defs(rel)
    here = os.path.dirname(rel)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.level >= 1 and node.module:
            base = here
            for _ in range(node.level - 1):
                base = os.path.dirname(base)
            target = os.path.join(base, *node.module.split("."))
            rel2 = target + ".py" if os.path.isfile(os.path.join(REF_PKG, target + ".py")) else os.path.join(target, "__init__.py")
            for alias in node.names:
                found[alias.asname or alias.name] = (rel2, alias.name)
    return found


def exports_of(pkg):
    return imports_of(os.path.join(pkg, "__init__.py") if pkg else "__init__.py")


def resolve(rel, name, depth=0):
    """Follow re-exports (package __init__ -> module -> module) to the defining statement."""
    _, defs = module_defs(rel)
    if name in defs:
        return rel, defs[name]
    if depth < 5:
        nxt = imports_of(rel).get(name)
        if nxt is not None:
            return resolve(nxt[0], nxt[1], depth + 1)
    return rel, None


def main():
    surface = {}
    for pkg in PACKAGES:
        entry = {}
        for public, (rel, name) in sorted(exports_of(pkg).items()):
            if public in OUT_OF_SCOPE:
                continue
            where, node = resolve(rel, name)
            d = describe(node) if node is not None else None
            if d is None:
                raise SystemExit(f"cannot find the definition of {pkg or 'aggforce'}.{public} ({rel})")
            d["defined_in"] = where
            entry[public] = d
        surface["aggforce" + ("." + pkg if pkg else "")] = entry
    for rel, names in EXTRA.items():
        _, defs = module_defs(rel)
        entry = {}
        for name in names:
            d = describe(defs[name])
            d["defined_in"] = rel
            entry[name] = d
        surface["aggforce." + rel[:-3].replace("/", ".")] = entry
    out = os.path.join(REPO, "tests", "golden", "g9_signatures.json")
    with open(out, "w") as fh:
        json.dump(surface, fh, indent=1, sort_keys=True)
    n = sum(len(v) for v in surface.values())
    print(f"wrote {out}: {n} public names in {len(surface)} namespaces")


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    main()
