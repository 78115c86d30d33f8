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
