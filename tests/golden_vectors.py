"""Golden vectors re-typed from the reference's own tests / docstrings.

Each entry cites the reference file:line that asserts it.  Used by BOTH the oracle tests
(-m "not gpu") and the HIP parity tests (-m gpu): the same literals pin both sides.
"""
import numpy as np

# G1  tests/test_graph.py:341-357 (send_recv, int features) and :363-401 (send -> recv(sum))
G1_N = 5
G1_EDGES = np.array([(0, 1), (1, 2), (3, 4), (4, 1), (1, 0)], dtype=np.int64)
G1_X = np.array([[1, 2, 3, 4], [2, 3, 4, 5], [3, 4, 5, 6], [4, 5, 6, 7], [5, 6, 7, 8]])
G1_MSG = np.array([[1, 2, 3, 4], [2, 3, 4, 5], [4, 5, 6, 7], [5, 6, 7, 8], [2, 3, 4, 5]])
G1_OUT = np.array([[2, 3, 4, 5], [6, 8, 10, 12], [2, 3, 4, 5], [0, 0, 0, 0], [4, 5, 6, 7]])

# G2  tests/test_dist_graph.py:115-137 send_ue_recv(add, sum), edge feature all ones
G2_EFEAT = np.ones((5, 1))
G2_OUT = np.array([[3, 4, 5, 6], [8, 10, 12, 14], [3, 4, 5, 6], [0, 0, 0, 0], [5, 6, 7, 8]])

# G3  tests/test_math.py:35-66 segment_softmax known answer + overflow case
G3_IDS = np.array([0, 0, 1], dtype=np.int64)
G3_DATA = np.array([[1, 2, 3], [3, 2, 1], [4, 5, 6]], dtype=np.float32)
G3_OUT = np.array([[0.11920292, 0.5, 0.880797], [0.880797, 0.5, 0.11920292], [1, 1, 1]], np.float32)
G3_DATA_BIG = np.array([[1, 2, 0.003], [3, 2, 1e10], [4, 5, 6]], dtype=np.float32)
G3_OUT_BIG = np.array([[0.11920292, 0.5, 0.0], [0.880797, 0.5, 1.0], [1, 1, 1]], np.float32)

# G4  tests/test_graph_op.py:57-68 edge_softmax by dst and by src, exact equality, edge order
G4_N = 3
G4_EDGES = np.array([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)], dtype=np.int64)
G4_LOGITS = np.ones((6, ), dtype=np.float32)
G4_BY_DST = np.array([1, 0.5, 1 / 3, 0.5, 1 / 3, 1 / 3], dtype=np.float32)
G4_BY_SRC = np.array([1 / 3, 1 / 3, 1 / 3, 0.5, 0.5, 1], dtype=np.float32)

# G5  tests/test_graph.py:104-140 degrees
G5_N = 5
G5_EDGES = np.array([(0, 1), (1, 2), (3, 4)], dtype=np.int64)
G5_INDEG = np.array([0, 1, 1, 0, 1], dtype=np.int64)
G5_OUTDEG = np.array([1, 1, 0, 1, 0], dtype=np.int64)

# G6  tests/test_graph.py:77-99 neighbours
G6_N = 5
G6_EDGES = np.array([(0, 1), (0, 2), (1, 2), (3, 4)], dtype=np.int64)
G6_PRED = [set(), {0}, {0, 1}, set(), {3}]
G6_SUCC = [{1, 2}, {2}, set(), {4}, set()]

# G7  pgl/math.py:72-75,107-110,139-142,172-175 docstring answers
G7_DATA = np.array([[1, 2, 3], [3, 2, 1], [4, 5, 6]], dtype=np.float32)
G7_IDS = np.array([0, 0, 1], dtype=np.int64)
G7 = {
    "sum": np.array([[4, 4, 4], [4, 5, 6]], np.float32),
    "mean": np.array([[2, 2, 2], [4, 5, 6]], np.float32),
    "min": np.array([[1, 2, 1], [4, 5, 6]], np.float32),
    "max": np.array([[3, 2, 3], [4, 5, 6]], np.float32),
}

# G8  build_index on G1's graph keyed by dst, produced by the reference's compiled
#     graph_kernel.pyx:59-88 (and re-produced live in tests via oracle/_ref)
G8 = {
    "degree": np.array([1, 2, 1, 0, 1], np.int64),
    "sorted_v": np.array([1, 0, 4, 1, 3], np.int64),
    "sorted_u": np.array([0, 1, 1, 2, 4], np.int64),
    "sorted_eid": np.array([4, 0, 3, 1, 2], np.int64),
    "indptr": np.array([0, 1, 3, 4, 4, 5], np.int64),
}

# G9  tests/test_bigraph.py:395-401,414-507 bipartite send_recv and recv(mode="src")
G9_SRC_N, G9_DST_N = 5, 4
G9_EDGES = np.array([(0, 1), (1, 2), (3, 3), (4, 1), (1, 0)], dtype=np.int64)
G9_SRC_X = np.array([[1, 2, 3, 4], [2, 3, 4, 5], [3, 4, 5, 6], [4, 5, 6, 7], [5, 6, 7, 8]], np.float32)
G9_DST_X = np.array([[2, 3, 4, 5], [3, 4, 5, 6], [4, 5, 6, 7], [5, 6, 7, 8]], np.float32)
G9_SEND_RECV = np.array([[2, 3, 4, 5], [6, 8, 10, 12], [2, 3, 4, 5], [4, 5, 6, 7]], np.float32)
G9_DST_MSG = np.array([[3, 4, 5, 6], [4, 5, 6, 7], [5, 6, 7, 8], [3, 4, 5, 6], [2, 3, 4, 5]], np.float32)
G9_RECV_SRC = np.array([[3, 4, 5, 6], [6, 8, 10, 12], [0, 0, 0, 0], [5, 6, 7, 8], [3, 4, 5, 6]], np.float32)

# G10 legacy/tests/scatter_add_test.py:27-37 scatter(mode='add')
G10_X = np.array([[1, 2], [5, 6]], np.float32)
G10_IDX = np.array([1, 1], np.int64)
G10_UPD = np.array([[3, 4], [3, 4]], np.float32)
G10_OUT = np.array([[1, 2], [11, 14]], np.float32)

# G11 tests/test_graph.py:292-335 send gathers on the path graph 0->1->2->3
G11_N = 4
G11_EDGES = np.array([(0, 1), (1, 2), (2, 3)], dtype=np.int64)
G11_NFEAT = np.arange(4).reshape(-1, 1)
G11_EFEAT = np.arange(3).reshape(-1, 1)
G11_SRC = np.array([0, 1, 2]).reshape(-1, 1)
G11_DST = np.array([1, 2, 3]).reshape(-1, 1)
