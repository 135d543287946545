"""Drop-in surface: every public function, class and method of the reference modules on (and next to) the hot path exists in pgl_amd
under the same name with the same leading parameters, in the same order -- so a positional call written against PGL binds the same way.

tests/golden/api_signatures.json is DATA read from the reference checkout with `ast` by tests/golden/make_api_signatures.py (251 entries:
parameter names, defaults, property flags; no source text).  The second test imports through the `pgl` alias by every submodule path the
reference's own programs use (`from pgl.sampling.custom import subgraph`: 15 places, `from pgl.utils.data.dataloader import ...`)."""
import ast
import importlib
import inspect
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOD = {"graph.py": ["pgl_amd.graph", "pgl_amd.distributed"], "bigraph.py": ["pgl_amd.bigraph"], "heter_graph.py": ["pgl_amd.bigraph"],
       "message.py": ["pgl_amd.message"], "math.py": ["pgl_amd.math"], "partition.py": ["pgl_amd.partition"], "nn/conv.py": ["pgl_amd.nn"],
       "nn/pool.py": ["pgl_amd.nn"], "nn/gmt_pool.py": ["pgl_amd.nn.gmt_pool"], "nn/functional/graph_op.py": ["pgl_amd.nn.functional"],
       "utils/helper.py": ["pgl_amd.utils.helper"], "utils/op.py": ["pgl_amd.utils.op"], "utils/transform.py": ["pgl_amd.utils.transform"],
       "utils/edge_index.py": ["pgl_amd.utils.edge_index"], "sampling/sage.py": ["pgl_amd.sampling"], "sampling/custom.py": ["pgl_amd.sampling"],
       "utils/data/dataloader.py": ["pgl_amd.utils.data"], "utils/logger.py": ["pgl_amd.utils.logger"]}


def _find(mods, name):
    for m in mods:
        obj = importlib.import_module(m)
        for part in name.split("."):
            if not hasattr(obj, part):
                obj = None
                break
            obj = getattr(obj, part)
        if obj is not None:
            return obj
    return None


def test_every_reference_name_exists_with_the_same_leading_parameters():
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "api_signatures.json")))
    missing, differ, n, n_defaults = [], [], 0, 0
    for mod, rec in fx.items():
        for name, sig in rec.items():
            n += 1
            obj = _find(MOD[mod], name)
            if obj is None:
                missing.append("%s: %s" % (mod, name))
                continue
            if sig.get("class"):
                continue
            if "." in name:
                owner, attr = name.split(".")
                static = inspect.getattr_static(_find(MOD[mod], owner), attr, None)
                if sig.get("property"):
                    assert isinstance(static, property) or not callable(obj), "%s: %s is a property in the reference" % (mod, name)
                    continue
                if isinstance(static, property):
                    differ.append("%s: %s is a method in the reference, a property here" % (mod, name))
                    continue
            try:
                ps = list(inspect.signature(obj).parameters.values())
            except (TypeError, ValueError):
                continue
            mine = [p.name for p in ps if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
            ref = sig["args"]
            if ref and ref[0] in ("self", "cls") and (not mine or mine[0] not in ("self", "cls")):
                ref = ref[1:]
            if mine[:len(ref)] != ref:
                differ.append("%s: %s reference %s, here %s" % (mod, name, ref, mine))
                continue
            by_name = {p.name: p for p in ps}
            for arg, dflt in zip(sig["args"], sig["defaults"]):          # the same DEFAULT VALUES: a differing default changes behaviour silently
                if arg in ("self", "cls"):
                    continue
                have = by_name[arg].default
                if dflt is None:
                    if have is not inspect.Parameter.empty:
                        differ.append("%s: %s(%s) is required in the reference, optional here" % (mod, name, arg))
                    continue
                try:
                    want = ast.literal_eval(dflt)
                except (ValueError, SyntaxError):
                    want = dflt
                if have is inspect.Parameter.empty or (have != want and str(have) != str(want)):
                    differ.append("%s: %s(%s=%s) in the reference, %r here" % (mod, name, arg, dflt, have))
                n_defaults += 1
    assert n > 240 and n_defaults > 200
    assert not missing, missing
    assert not differ, differ


def test_the_references_import_paths_resolve_through_the_alias():
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import pgl
assert pgl.dataset.CoraDataset and pgl.graph_kernel and pgl.utils and pgl.graph.Graph is pgl.Graph
from pgl.sampling.custom import subgraph
from pgl.sampling.sage import graphsage_sample, edge_hash, traverse, flat_node_and_edge
from pgl.sampling import NeighborSampler, graphsage_sample as g2
from pgl.utils.data import Dataloader, Dataset, StreamDataset
from pgl.utils.data.dataloader import Dataloader as D2
from pgl.utils.data.dataset import Dataset as DS
from pgl.utils.logger import log
from pgl.utils.helper import scatter, unique_segment, graph_send_recv, to_paddle_tensor
from pgl.utils.transform import to_undirected, add_self_loops, to_dense_batch, filter_adj
from pgl.utils import op
from pgl.graph import Graph
from pgl.bigraph import BiGraph
from pgl.message import Message
from pgl.partition import random_partition, metis_partition
from pgl.nn import functional as GF
from pgl.nn import Set2Set, GlobalAttention, SAGPool, GraphPool, GINConv, GraphMultisetTransformer
import pgl.nn as gnn, pgl.nn.functional as F
assert g2 is graphsage_sample and D2 is Dataloader and DS is Dataset and Graph is pgl.Graph
try:
    pgl.sampling.sage.HeteroNeighborSampler([], [])
except NotImplementedError:
    print("ok")
""" % (ROOT, os.path.join(ROOT, "pgl_amd", "compat"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout + r.stderr)[-3000:]
