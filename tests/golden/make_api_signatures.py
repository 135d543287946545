"""Writes tests/golden/api_signatures.json: for every public function / class / method of the reference modules on (and next to) the
hot path, its parameter names in order with their defaults -- DATA about the reference's API surface (no source text), read with `ast`
from the checkout at /root/reference.  tests/test_api_surface.py holds pgl_amd to it.

    python tests/golden/make_api_signatures.py            (in the build container: /root/reference must exist)
"""
import ast
import json
import os

REF = "/root/reference/pgl"
MODULES = ["graph.py", "bigraph.py", "heter_graph.py", "message.py", "math.py", "partition.py", "nn/conv.py", "nn/pool.py", "nn/gmt_pool.py",
           "nn/functional/graph_op.py", "utils/helper.py", "utils/op.py", "utils/transform.py", "utils/edge_index.py", "sampling/sage.py",
           "sampling/custom.py", "utils/data/dataloader.py", "utils/logger.py"]


def params(fn):
    a = fn.args
    names = [x.arg for x in a.args]
    defaults = [None] * (len(names) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
    return {"args": names, "defaults": defaults, "vararg": bool(a.vararg), "kwarg": bool(a.kwarg),
            "kwonly": [x.arg for x in a.kwonlyargs]}


def main():
    out = {}
    for m in MODULES:
        tree = ast.parse(open(os.path.join(REF, m)).read())
        rec = {}
        for n in tree.body:
            if isinstance(n, ast.FunctionDef) and not n.name.startswith("_"):
                rec[n.name] = params(n)
            elif isinstance(n, ast.ClassDef) and not n.name.startswith("_"):
                rec[n.name] = {"class": True, "bases": [ast.unparse(b) for b in n.bases]}
                for f in n.body:
                    if isinstance(f, ast.FunctionDef) and (not f.name.startswith("_") or f.name in ("__init__", "__call__", "__getitem__", "__len__")):
                        rec[n.name + "." + f.name] = dict(params(f), property=any(ast.unparse(d) == "property" for d in f.decorator_list))
        out[m] = rec
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "api_signatures.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print(path, sum(len(v) for v in out.values()), "entries")


if __name__ == "__main__":
    main()
