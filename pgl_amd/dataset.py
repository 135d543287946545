"""pgl.dataset (pgl/dataset.py): the three loaders the reference's gcn / gat / graphsage examples call --
CoraDataset (:163-246), CitationDataset (:60-161) and RedditDataset (:386-473) -- reading the SAME on-disk formats from
`$PGL_DATA_DIR/<name>/` (default: pgl_amd/data/<name>/, the reference keeps them in pgl/data/<name>/):

    cora/      cora.content  "<paper id> <1433 binary words> <class name>" per line;  cora.cites  "<cited> <citing>" per line
    citeseer/, pubmed/   the Planetoid pickles ind.<name>.{x,y,tx,ty,allx,ally,graph} + ind.<name>.test.index
    reddit/    reddit.npz (feats, y_train, y_val, y_test, train_index, val_index, test_index) + reddit_adj.npz (scipy sparse)

No dataset ships with this package (SURVEY section 2 marks datasets out of scope; the reference checkout itself lacks
cora.content and the Reddit files): `write_standin_*` generate SEEDED STAND-INS in exactly these formats -- planted-partition
graphs whose features correlate with the labels, so the examples train to well above chance -- for tests and offline
runs.  Loaders follow the reference's pre-processing step by step (row normalisation, symmetrisation + self loops +
de-duplication through a set, Planetoid index fix-ups, StandardScaler fitted on the training rows).
"""
import os
import pickle
import sys

import numpy as np

from .graph import Graph

__all__ = ["CitationDataset", "CoraDataset", "RedditDataset", "get_default_data_dir", "write_standin_cora",
           "write_standin_citation", "write_standin_reddit"]


def get_default_data_dir(name):
    """pgl/dataset.py:39-45, with an override for offline stand-ins."""
    root = os.environ.get("PGL_DATA_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
    return os.path.join(root, name)


def _edge_list(pairs, num_nodes, symmetry_edges, self_loop):
    """(u, v) pairs -> de-duplicated edge list in the reference's order: list(set(...)) of python int tuples."""
    all_edges = []
    for u, v in pairs:
        all_edges.append((u, v))
        if symmetry_edges:
            all_edges.append((v, u))
    if self_loop:
        for i in range(num_nodes):
            all_edges.append((i, i))
    return list(set(all_edges))


class CoraDataset(object):
    """pgl/dataset.py:163-246.  Attributes: graph (node_feat["words"]), y, num_classes, train_index, val_index, test_index."""

    def __init__(self, symmetry_edges=True, self_loop=True):
        self.path = get_default_data_dir("cora")
        self.symmetry_edges, self.self_loop = symmetry_edges, self_loop
        self._load_data()

    def _load_data(self):
        node_feature, paper_ids, y, y_dict = [], [], [], {}
        with open(os.path.join(self.path, "cora.content"), "r") as f:
            for line in f:
                line = line.strip().split()
                if not line:
                    continue
                paper_ids.append(int(line[0]))
                if line[-1] not in y_dict:
                    y_dict[line[-1]] = len(y_dict)
                y.append(y_dict[line[-1]])
                feat = np.array([int(i) for i in line[1:-1]], dtype="float32")
                node_feature.append(feat / (np.sum(feat) + 1e-15))
        paper2vid = dict((v, k) for k, v in enumerate(paper_ids))
        num_nodes = len(paper_ids)
        pairs = []
        with open(os.path.join(self.path, "cora.cites"), "r") as f:
            for line in f:
                if line.strip():
                    u, v = line.split()
                    pairs.append((paper2vid[int(u)], paper2vid[int(v)]))
        self.graph = Graph(num_nodes=num_nodes, edges=_edge_list(pairs, num_nodes, self.symmetry_edges, self.self_loop),
                           node_feat={"words": np.array(node_feature, dtype="float32")})
        perm = np.arange(0, num_nodes)
        self.train_index, self.val_index, self.test_index = perm[:140], perm[200:500], perm[500:1500]
        self.y = np.array(y, dtype="int64")
        self.num_classes = len(y_dict)


class CitationDataset(object):
    """pgl/dataset.py:60-161 (Planetoid pickles; needs networkx like the reference)."""

    def __init__(self, name, symmetry_edges=True, self_loop=True):
        self.path = get_default_data_dir(name)
        self.symmetry_edges, self.self_loop, self.name = symmetry_edges, self_loop, name
        self._load_data()

    def _load_data(self):
        import networkx as nx
        objects = []
        for part in ("x", "y", "tx", "ty", "allx", "ally", "graph"):
            with open("%s/ind.%s.%s" % (self.path, self.name, part), "rb") as f:
                objects.append(pickle.load(f, encoding="latin1") if sys.version_info > (3, 0) else pickle.load(f))
        x, y, tx, ty, allx, ally, adj = objects
        test_idx_reorder = [int(line.strip()) for line in open("%s/ind.%s.test.index" % (self.path, self.name))]
        test_idx_range = np.sort(test_idx_reorder)
        allx, tx = np.asarray(allx.todense()), np.asarray(tx.todense())
        if self.name == "citeseer":                       # isolated test nodes are missing from tx/ty: zero rows in place
            full = range(min(test_idx_reorder), max(test_idx_reorder) + 1)
            tx_ext = np.zeros((len(full), x.shape[1]), dtype="float32")
            tx_ext[test_idx_range - min(test_idx_range), :] = tx
            ty_ext = np.zeros((len(full), y.shape[1]), dtype="float32")
            ty_ext[test_idx_range - min(test_idx_range), :] = ty
            tx, ty = tx_ext, ty_ext
        features = np.vstack([allx, tx])
        features[test_idx_reorder, :] = features[test_idx_range, :]
        features = np.array(features / (np.sum(features, axis=-1, keepdims=True) + 1e-15), dtype="float32")
        onehot = np.vstack((ally, ty))
        onehot[test_idx_reorder, :] = onehot[test_idx_range, :]
        g = nx.DiGraph(nx.from_dict_of_lists(adj))
        n = g.number_of_nodes()
        self.graph = Graph(num_nodes=n, edges=_edge_list([tuple(e) for e in g.edges()], n, self.symmetry_edges, self.self_loop),
                           node_feat={"words": features})
        self.y = np.array(np.argmax(onehot, 1), dtype="int64")
        self.num_classes = onehot.shape[1]
        self.train_index = np.array(range(len(y)), dtype="int32")
        self.val_index = np.array(range(len(y), len(y) + 500), dtype="int32")
        self.test_index = np.array(test_idx_range.tolist(), dtype="int32")


class RedditDataset(object):
    """pgl/dataset.py:386-473.  Attributes: graph, feature, num_classes (41), {train,val,test}_{index,label}."""

    def __init__(self, normalize=True, symmetry=True):
        self.path = get_default_data_dir("reddit")
        if not os.path.exists(self.path):
            raise ValueError("\n Please download the dataset to \n \t%s \n before use it (reddit.npz, reddit_adj.npz), or generate "
                             "a stand-in with pgl_amd.dataset.write_standin_reddit." % self.path)
        self._load_data(normalize, symmetry)

    def _load_data(self, normalize=True, symmetry=True):
        import scipy.sparse as sp
        data = np.load(os.path.join(self.path, "reddit.npz"))
        adj = sp.load_npz(os.path.join(self.path, "reddit_adj.npz"))
        if symmetry:
            adj = adj + adj.T
        adj = adj.tocoo()
        self.train_label, self.val_label, self.test_label = data["y_train"], data["y_val"], data["y_test"]
        self.train_index, self.val_index, self.test_index = data["train_index"], data["val_index"], data["test_index"]
        feature = data["feats"].astype("float32")
        if normalize:
            from sklearn.preprocessing import StandardScaler
            scaler = StandardScaler()
            scaler.fit(feature[self.train_index])
            feature = scaler.transform(feature)
        self.graph = Graph(num_nodes=feature.shape[0], edges=np.stack([adj.row, adj.col], 1).astype(np.int64))
        self.feature = feature
        self.num_classes = 41


# ------------------------------------------------------------------------------------------------------------------
# seeded stand-ins in the reference's file formats (there is no network; the reference checkout lacks these files)
# ------------------------------------------------------------------------------------------------------------------
def _planted(rng, n, classes, avg_deg, p_in=0.8):
    """labels + undirected pairs of a planted-partition graph: a fraction p_in of every node's links stay in its class."""
    y = rng.integers(0, classes, n)
    by = [np.nonzero(y == c)[0] for c in range(classes)]
    m = n * avg_deg // 2
    u = rng.integers(0, n, m)
    same = rng.random(m) < p_in
    v = rng.integers(0, n, m)
    for c in range(classes):
        sel = same & (y[u] == c) & (len(by[c]) > 0)
        v[sel] = by[c][rng.integers(0, max(len(by[c]), 1), int(sel.sum()))]
    keep = u != v
    return y, u[keep], v[keep]


def _bag_of_words(rng, y, dim, words_per_node, classes):
    """binary rows whose active words are drawn mostly from the label's own slice of the vocabulary."""
    n = len(y)
    x = np.zeros((n, dim), dtype=np.int8)
    width = dim // classes
    for i in range(n):
        k = words_per_node
        own = rng.integers(y[i] * width, (y[i] + 1) * width, int(k * 0.6))
        other = rng.integers(0, dim, k - len(own))
        x[i, own] = 1
        x[i, other] = 1
    return x


def write_standin_cora(path, seed=0, num_nodes=2708, dim=1433, classes=7):
    """cora.content / cora.cites with Cora's sizes (2 708 papers, 1 433 words, 7 classes, 5 429 citation pairs)."""
    rng = np.random.default_rng(seed)
    os.makedirs(path, exist_ok=True)
    y, u, v = _planted(rng, num_nodes, classes, 4)
    u, v = u[:5429], v[:5429]
    x = _bag_of_words(rng, y, dim, 18, classes)
    ids = rng.permutation(np.arange(10000, 10000 + 7 * num_nodes, 7))[:num_nodes]          # sparse paper ids, as in the real file
    names = ["Case_Based", "Genetic_Algorithms", "Neural_Networks", "Probabilistic_Methods", "Reinforcement_Learning",
             "Rule_Learning", "Theory"]
    with open(os.path.join(path, "cora.content"), "w") as f:
        for i in range(num_nodes):
            f.write("%d\t%s\t%s\n" % (ids[i], "\t".join(map(str, x[i].tolist())), names[y[i] % len(names)] if classes <= len(names) else "c%d" % y[i]))
    with open(os.path.join(path, "cora.cites"), "w") as f:
        for a, b in zip(u, v):
            f.write("%d\t%d\n" % (ids[a], ids[b]))
    return path


def write_standin_citation(path, name="citeseer", seed=0, num_nodes=3327, dim=3703, classes=6, n_train=120, n_test=1000):
    """Planetoid pickles ind.<name>.* (scipy CSR features, one-hot labels, adjacency dict, test index list)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    os.makedirs(path, exist_ok=True)
    y, u, v = _planted(rng, num_nodes, classes, 3)
    x = _bag_of_words(rng, y, dim, 30, classes).astype(np.float32)
    onehot = np.eye(classes, dtype=np.int32)[y]
    n_all = num_nodes - n_test
    adj = {i: [] for i in range(num_nodes)}
    for a, b in zip(u.tolist(), v.tolist()):
        adj[a].append(b)
    test_ids = rng.permutation(np.arange(n_all, num_nodes))
    parts = {"x": sp.csr_matrix(x[:n_train]), "y": onehot[:n_train], "allx": sp.csr_matrix(x[:n_all]), "ally": onehot[:n_all],
             "tx": sp.csr_matrix(x[np.sort(test_ids)]), "ty": onehot[np.sort(test_ids)], "graph": adj}
    for k, val in parts.items():
        with open(os.path.join(path, "ind.%s.%s" % (name, k)), "wb") as f:
            pickle.dump(val, f, protocol=2)
    with open(os.path.join(path, "ind.%s.test.index" % name), "w") as f:
        f.write("\n".join(str(int(i)) for i in test_ids) + "\n")
    return path


def write_standin_reddit(path, seed=0, num_nodes=20000, dim=602, avg_deg=20, classes=41):
    """reddit.npz + reddit_adj.npz (real Reddit: 232 965 nodes, 602-d, 41 classes; the stand-in keeps d and the classes)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    os.makedirs(path, exist_ok=True)
    y, u, v = _planted(rng, num_nodes, classes, avg_deg, p_in=0.7)
    centers = rng.standard_normal((classes, dim)).astype(np.float32)
    feats = centers[y] * 0.5 + rng.standard_normal((num_nodes, dim)).astype(np.float32)
    perm = rng.permutation(num_nodes)
    a, b = int(0.66 * num_nodes), int(0.76 * num_nodes)
    tr, va, te = np.sort(perm[:a]), np.sort(perm[a:b]), np.sort(perm[b:])
    np.savez(os.path.join(path, "reddit.npz"), feats=feats, y_train=y[tr], y_val=y[va], y_test=y[te], train_index=tr,
             val_index=va, test_index=te)
    adj = sp.coo_matrix((np.ones(len(u), dtype=np.float32), (u, v)), shape=(num_nodes, num_nodes)).tocsr()
    adj.data[:] = 1.0
    sp.save_npz(os.path.join(path, "reddit_adj.npz"), adj)
    return path
