"""Input generators shared by the golden generator and the tests (SURVEY.md section 8c/8d distributions)."""
import numpy as np


def f32exact(x):
    """Round to the nearest float32 and return as float64, so fp32 and fp64 paths see identical inputs."""
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def adjacency(kind, N, rng):
    """Three adjacency families:
    sym01    symmetric Erdos-Renyi(0.5) 0/1 with unit diagonal (pattern of tests/test_RisiContraction_18_gpu.cu:113-121)
    weighted non-negative, asymmetric, ~half zeros (Coulomb-like reduced adjacency, SMP_omega.h:556-581)
    signed   contains negative entries: pins RisiContraction_18's `A > 0` gate vs no gate in _10/_50
    """
    if kind == "sym01":
        U = (rng.uniform(0, 1, (N, N)) < 0.5).astype(np.float64)
        A = np.triu(U, 1)
        A = A + A.T + np.eye(N)
    elif kind == "weighted":
        A = rng.uniform(0, 2, (N, N)) * (rng.uniform(0, 1, (N, N)) < 0.5)
    elif kind == "signed":
        A = rng.uniform(-1, 1, (N, N))
    else:
        raise ValueError(kind)
    return f32exact(A)


def cfg_graph(N, C, seed, K=18):
    """One benchmark graph (cfg2/cfg5 distributions): P~U(-1,1), sym01 A, G~U(0,1)."""
    rng = np.random.default_rng(seed)
    P = f32exact(rng.uniform(-1, 1, (N, N, N, C)))
    A = adjacency("sym01", N, rng)
    G = f32exact(rng.uniform(0, 1, (N, N, K, C)))
    return P, A, G


# ---- molecules (data only) ------------------------------------------------------------------------------------------
def toy_molecules():
    """The four hand-built molecules of the reference's tests/test_SMP_omega.cpp:71-147 (edges + atom labels; features are
    one-hot over C,H,N,O; target = number of atoms).  Returns [(name, adj[V,V] int, feature[V,4], target)]."""
    spec = {
        "CH4": ([(0, 1), (0, 2), (0, 3), (0, 4)], "CHHHH"),
        "NH3": ([(0, 1), (0, 2), (0, 3)], "NHHH"),
        "H2O": ([(0, 1), (0, 2)], "OHH"),
        "C2H4": ([(0, 1), (0, 2), (0, 3), (3, 4), (3, 5)], "CHHCHH"),
    }
    out = []
    for name, (edges, labels) in spec.items():
        V = len(labels)
        adj = np.zeros((V, V), dtype=np.int32)
        for u, v in edges:
            adj[u, v] = adj[v, u] = 1
        feat = np.zeros((V, 4))
        for v, ch in enumerate(labels):
            feat[v, "CHNO".index(ch)] = 1.0
        out.append((name, adj, feat, float(V)))
    return out


def er_graph(V, p, nFeatures, seed):
    """Erdos-Renyi graph with random one-hot vertex labels (shape of tests/test_SMP_similarity.cu:34-53)."""
    rng = np.random.default_rng(seed)
    U = np.triu((rng.uniform(0, 1, (V, V)) < p).astype(np.int32), 1)
    adj = U + U.T
    feat = np.zeros((V, nFeatures))
    feat[np.arange(V), rng.integers(0, nFeatures, V)] = 1.0
    return adj, feat


def synthetic_molecule(seed, nV=None):
    """QM9-size synthetic molecule: nV ~ U{3..29}; ceil(9 nV / 20) heavy atoms forming a random tree
    with degree <= 4 plus up to one ring closure; the rest are H attached to heavy atoms with free valence.
    Features: one-hot over (H, C, N, O, F).  Target = nV.  Returns (adj int32 [V,V], feature [V,5], target)."""
    rng = np.random.default_rng(seed)
    if nV is None:
        nV = int(rng.integers(3, 30))
    nheavy = min(nV, max(1, -(-9 * nV // 20)))
    adj = np.zeros((nV, nV), dtype=np.int32)
    deg = np.zeros(nV, dtype=np.int64)
    for v in range(1, nheavy):
        cand = [u for u in range(v) if deg[u] < 4]
        u = int(rng.choice(cand)) if cand else 0
        adj[u, v] = adj[v, u] = 1
        deg[u] += 1
        deg[v] += 1
    if nheavy >= 4 and rng.uniform() < 0.5:  # one ring closure
        free = [u for u in range(nheavy) if deg[u] < 4]
        rng.shuffle(free)
        for i in range(len(free)):
            for j in range(i + 1, len(free)):
                u, v = free[i], free[j]
                if not adj[u, v]:
                    adj[u, v] = adj[v, u] = 1
                    deg[u] += 1
                    deg[v] += 1
                    break
            else:
                continue
            break
    for h in range(nheavy, nV):
        cand = [u for u in range(nheavy) if deg[u] < 4]
        u = int(rng.choice(cand)) if cand else int(rng.integers(0, nheavy))
        adj[u, h] = adj[h, u] = 1
        deg[u] += 1
        deg[h] += 1
    feat = np.zeros((nV, 5))
    feat[nheavy:, 0] = 1.0
    heavy_type = rng.choice([1, 1, 1, 2, 3, 4], size=nheavy)
    feat[np.arange(nheavy), heavy_type] = 1.0
    return adj, feat, float(nV)


def smp_param_count(C, F, D, L):
    return C * F * (D + 1) + L * (18 * C * C + C) + C


def smp_params(C, F, D, L, seed, scale=None):
    """Random parameters, float32-exact, sized like uniform_init's range (GraphFlow.h:1297-1306): |w| < 1/size-ish."""
    rng = np.random.default_rng(seed)
    parts = [rng.uniform(-1, 1, C * F * (D + 1)) / np.sqrt(F * (D + 1))]
    for _ in range(L):
        parts.append(rng.uniform(-1, 1, 18 * C * C) / np.sqrt(18 * C))
        parts.append(rng.uniform(-0.1, 0.1, C))
    parts.append(rng.uniform(-1, 1, C) / np.sqrt(C))
    return f32exact(np.concatenate(parts))


def edge_molecules(F=5):
    """Corner cases of the SMP driver: a single atom, a bonded pair, two disconnected fragments (Floyd-Warshall never joins
    them: receptive fields stay inside a fragment), an isolated vertex beside a triangle, a 9-clique (every field is the
    whole molecule from level 1 on) and a 12-path (fields keep growing over the levels).  [(name, adj, feature, target)]"""
    def mk(V, edges, seed):
        adj = np.zeros((V, V), dtype=np.int32)
        for u, v in edges:
            adj[u, v] = adj[v, u] = 1
        rng = np.random.default_rng(seed)
        feat = np.zeros((V, F))
        feat[np.arange(V), rng.integers(0, F, V)] = 1.0
        return adj, feat, float(V)
    out = [("atom",) + mk(1, [], 1), ("pair",) + mk(2, [(0, 1)], 2),
           ("two_fragments",) + mk(7, [(0, 1), (1, 2), (3, 4), (4, 5), (5, 6), (6, 3)], 3),
           ("isolated_vertex",) + mk(4, [(0, 1), (1, 2), (2, 0)], 4),
           ("clique9",) + mk(9, [(i, j) for i in range(9) for j in range(i + 1, 9)], 5),
           ("path12",) + mk(12, [(i, i + 1) for i in range(11)], 6)]
    return out
