"""Row b2 (north_star: the engine "drops into examples/gcn, gat and graphsage unchanged").

The reference's three example PROGRAMS -- byte-compiled from the reference checkout by oracle/build_ref.py into
oracle/_ref/examples/**.pyc (build outputs, never source; /root/reference does not exist on the GPU box) -- are executed
UNMODIFIED with `import pgl` / `import paddle` resolved to pgl_amd/compat: `pgl` is pgl_amd itself, `paddle` a name layer
over torch, every graph operation runs in libpglamd's HIP kernels.  Two kinds of checks:

  * the programs' own main(): on seeded stand-in datasets written in the reference's on-disk formats
    (pgl_amd.dataset.write_standin_*), they run to completion and train to far above chance;
  * the loss TRAJECTORY: the example's own model class driven by the example's own train() step from the fixture's initial
    parameters reproduces tests/golden/layers/train_{gcn,gat,sage}.npz -- trajectories the reference's code produced on the
    paddle stand-in of oracle/ (tests/golden/make_golden_layers.py).
"""
import importlib.machinery
import importlib.util
import os
import re
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EX = os.path.join(ROOT, "oracle", "_ref", "examples")
COMPAT = os.path.join(ROOT, "pgl_amd", "compat")


def _need_examples():
    if not os.path.exists(os.path.join(EX, "gcn", "train.pyc")):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import build_ref
        if build_ref.compile_examples() is None:
            pytest.skip("oracle/_ref/examples not built and the reference checkout is absent")


def test_compat_names_resolve_to_the_engine_not_to_copies():
    """`import pgl` must BE pgl_amd (one module object per name) and the paddle name layer must not reach oracle/."""
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import pgl, paddle, paddle.nn as nn, pgl_amd\n"
            "from paddle.optimizer import Adam\n"
            "from pgl.utils.logger import log\n"
            "from pgl.utils.data import Dataloader\n"
            "from pgl.sampling import graphsage_sample\n"
            "from pgl import graph_kernel\n"
            "import torch\n"
            "assert pgl is pgl_amd and pgl.nn is pgl_amd.nn and sys.modules['pgl.utils.data'] is pgl_amd.utils.data\n"
            "assert pgl.nn.GCNConv is pgl_amd.nn.GCNConv and issubclass(nn.Layer, torch.nn.Module)\n"
            "assert not any('ref_ops' in m or 'paddle_stub' in (getattr(sys.modules[m], '__file__', '') or '') for m in sys.modules)\n"
            "assert paddle.__file__.startswith(%r)\n"
            "print('ok')\n") % (ROOT, COMPAT, COMPAT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_standin_datasets_load_in_the_reference_formats(tmp_path, monkeypatch):
    import pgl_amd
    from pgl_amd import dataset as D
    monkeypatch.setenv("PGL_DATA_DIR", str(tmp_path))
    D.write_standin_cora(str(tmp_path / "cora"))
    c = D.CoraDataset()
    assert c.graph.num_nodes == 2708 and c.num_classes == 7 and c.graph.node_feat["words"].shape == (2708, 1433)
    np.testing.assert_allclose(c.graph.node_feat["words"].sum(1), 1.0, rtol=1e-5)        # row-normalised (dataset.py:213)
    e = np.asarray(c.graph.edges)
    assert len(set(map(tuple, e.tolist()))) == len(e)                                      # de-duplicated through a set
    assert (e[:, 0] == e[:, 1]).sum() == 2708                                              # one self loop per node
    assert set(map(tuple, e.tolist())) == set(map(tuple, e[:, ::-1].tolist()))             # symmetrised
    assert list(c.train_index[:3]) == [0, 1, 2] and len(c.val_index) == 300 and len(c.test_index) == 1000
    D.write_standin_citation(str(tmp_path / "citeseer"), "citeseer")
    ci = D.CitationDataset("citeseer", symmetry_edges=True)
    assert ci.graph.num_nodes == 3327 and ci.num_classes == 6 and len(ci.val_index) == 500
    D.write_standin_reddit(str(tmp_path / "reddit"), num_nodes=3000)
    r = D.RedditDataset(normalize=True, symmetry=True)
    assert r.feature.shape == (3000, 602) and r.num_classes == 41 and len(r.train_label) == len(r.train_index)
    np.testing.assert_allclose(r.feature[r.train_index].mean(0), 0.0, atol=1e-4)           # StandardScaler fitted on the training rows
    with pytest.raises(ValueError):
        monkeypatch.setenv("PGL_DATA_DIR", str(tmp_path / "nowhere"))
        D.RedditDataset()


def test_dataloader_matches_the_reference_contract():
    from pgl_amd.utils.data import Dataloader, Dataset

    class DS(Dataset):
        def __getitem__(self, i):
            return i

        def __len__(self):
            return 23
    for workers in (1, 3):
        got = list(Dataloader(DS(), batch_size=5, num_workers=workers, collate_fn=lambda b: np.array(b)))
        assert [len(b) for b in got] == [5, 5, 5, 5, 3] and np.concatenate(got).tolist() == list(range(23))
    assert len(Dataloader(DS(), batch_size=5, drop_last=True)) == 4
    np.random.seed(0)
    sh = np.concatenate(list(Dataloader(DS(), batch_size=4, shuffle=True, collate_fn=np.array)))
    assert sorted(sh.tolist()) == list(range(23)) and sh.tolist() != list(range(23))
    with pytest.raises(ValueError):
        Dataloader(DS(), num_workers=0)


# ------------------------------------------------------------------------------------------------
# the programs themselves, unchanged, on the GPU
# ------------------------------------------------------------------------------------------------
def _run_example(rel, args, data_dir, timeout=900):
    _need_examples()
    script = os.path.join(EX, rel)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, COMPAT]), PGL_DATA_DIR=str(data_dir))
    r = subprocess.run([sys.executable, script] + args, capture_output=True, text=True, cwd=os.path.dirname(script), env=env,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    return r.stdout + r.stderr


@pytest.fixture(scope="module")
def standin_data(tmp_path_factory):
    from pgl_amd import dataset as D
    d = tmp_path_factory.mktemp("pgl_data")
    D.write_standin_cora(str(d / "cora"))
    D.write_standin_reddit(str(d / "reddit"), num_nodes=12000, avg_deg=12)
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("rel,args,floor", [("gcn/train.pyc", ["--dataset", "cora", "--epoch", "60", "--runs", "1"], 0.6),
                                            ("gat/train.pyc", ["--dataset", "cora", "--epoch", "60", "--runs", "1"], 0.6)])
def test_citation_examples_run_unchanged_on_the_engine(standin_data, rel, args, floor):
    log = _run_example(rel, args, standin_data)
    m = re.search(r"Best Test Accuracy: ([0-9.]+)", log)
    assert m, log[-2000:]
    assert float(m.group(1)) > floor, log[-800:]               # 7 classes: chance is 0.14
    assert "Average Speed" in log


@pytest.mark.gpu
def test_graphsage_example_runs_unchanged_on_the_engine(standin_data):
    log = _run_example("graphsage/cpu_sample_version/train.pyc",
                       ["--epoch", "2", "--batch_size", "256", "--sample_workers", "2", "--samples", "10", "5", "--hidden_size", "64",
                        "--normalize", "--symmetry"], standin_data)
    m = re.search(r"Best Test Accuracy: ([0-9.]+)", log)
    assert m, log[-2000:]
    assert float(m.group(1)) > 0.3, log[-800:]                  # 41 classes: chance is 0.024
    assert "Num nodes 12000" in log


def _example_module(rel):
    _need_examples()
    path = os.path.join(EX, rel)
    name = "ref_example_" + rel.replace("/", "_").replace(".pyc", "")
    loader = importlib.machinery.SourcelessFileLoader(name, path)
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


@pytest.fixture
def paddle_names():
    """install the compat names in THIS process and undo their one global side effect (torch's default device) afterwards."""
    import torch
    import pgl_amd.compat
    pgl_amd.compat.install()
    import paddle
    torch.set_default_device(paddle._device())
    yield paddle
    torch.set_default_device("cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["gcn", "gat", "sage"])
def test_example_models_and_train_step_reproduce_the_reference_trajectory(tag, paddle_names):
    """examples/gcn/train.py's GCN + train(), examples/gat/train.py's GAT + train(), examples/graphsage/.../model.py's
    GraphSage: the reference's own classes and training step, executed on the engine through the compat names, against
    the trajectory the same code produced in the reference environment (fixture)."""
    import torch
    paddle = paddle_names
    import pgl
    z = np.load(os.path.join(HERE, "golden", "layers", "train_%s.npz" % tag))
    n, din, ncls = int(z["num_nodes"]), z["x"].shape[1], 5
    if tag == "sage":
        sys.path.insert(0, os.path.join(EX, "graphsage", "cpu_sample_version"))
        mod = _example_module("graphsage/cpu_sample_version/model.pyc")
        model = mod.GraphSage(din, ncls, num_layers=2, hidden_size=16, dropout=0.0)
        adam = _example_module("gcn/train.pyc").Adam
    else:
        mod = _example_module("%s/train.pyc" % tag)
        model = mod.GCN(din, ncls, num_layers=1, hidden_size=16, dropout=0.0) if tag == "gcn" else \
            mod.GAT(din, ncls, num_layers=1, feat_drop=0.0, attn_drop=0.0, num_heads=4, hidden_size=8)
        adam = mod.Adam
    sd = {}
    mine = model.state_dict()
    for k in z.files:
        if k.startswith("init::"):
            t = torch.as_tensor(z[k])
            name = k[len("init::"):]
            if name.endswith(".weight") and t.dim() == 2:
                t = t.t().contiguous()                      # Paddle's Linear keeps [in, out]
            assert tuple(t.shape) == tuple(mine[name].shape), name
            sd[name] = t
    assert set(sd) == set(mine)
    model.load_state_dict(sd)
    model = model.cuda()
    optim = adam(learning_rate=0.01, parameters=model.parameters(), weight_decay=0.0005)
    crit = paddle.nn.loss.CrossEntropyLoss()
    losses = []
    if tag == "sage":
        g = pgl.Graph(edges=z["edges"], num_nodes=n).tensor()
        xt = paddle.to_tensor(z["x"])
        idx_t, lab_t = paddle.to_tensor(z["train_idx"]), paddle.to_tensor(z["labels"][z["train_idx"]])
        for _ in range(len(z["losses"])):
            model.train()
            loss = crit(paddle.gather(model(g, xt), idx_t), lab_t)
            loss.backward()
            optim.step()
            optim.clear_grad()
            losses.append(float(loss.detach()))
        model.eval()
        logits = model(g, xt).detach().cpu().numpy()
    else:
        g = pgl.Graph(edges=z["edges"], num_nodes=n, node_feat={"words": z["x"]}).tensor()
        idx_t = paddle.to_tensor(np.expand_dims(z["train_idx"], -1))
        lab_t = paddle.to_tensor(np.expand_dims(z["labels"][z["train_idx"]], -1))
        for _ in range(len(z["losses"])):
            loss, _ = mod.train(idx_t, lab_t, model, g, crit, optim)        # the example's own training step
            losses.append(float(loss.detach()))
        model.eval()
        with paddle.no_grad():
            logits = model(g, g.node_feat["words"]).cpu().numpy()
    np.testing.assert_allclose(losses, z["losses"], rtol=2e-4)
    from gpu_common import close_rows
    close_rows(logits, z["final_logits"], rtol=2e-3, atol_row=2e-3)
