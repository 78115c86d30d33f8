import numpy as np

REL_TOL_F32 = 1e-5  # north_star: "outputs match the GraphFlow/ CPU path within 1e-5 fp32" (relative)


def rel_err(x, ref):
    """max|x - ref| / max(||ref||_inf, 1): the criterion of SURVEY.md section 7 / BASELINE.md section 4."""
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert x.shape == ref.shape, (x.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    return float(np.abs(x - ref).max() / max(np.abs(ref).max(), 1.0))


def rel_err_slices(x, ref, axis=-2):
    """rel_err per contraction slice (axis = the K axis of Out [..., N, N, K, C]), maximum over the slices: slice k17 of
    RisiContraction_18 (A . sum_a P[a,a,a]) is ~N^2 smaller than k4 (A . sum_abc P), so one max-norm over all K slices cannot see an
    error confined to the small ones (round-2 review, weak #3)."""
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert x.shape == ref.shape, (x.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    x = np.moveaxis(x, axis, 0).reshape(x.shape[axis], -1)
    ref = np.moveaxis(ref, axis, 0).reshape(ref.shape[axis], -1)
    return float((np.abs(x - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1.0)).max())


def golden_cases(golden, prefix):
    """{tag: {field: array}} for every fixture whose tag starts with prefix."""
    out = {}
    for k, v in golden.items():
        tag, field = k.split("__")
        if tag.startswith(prefix):
            out.setdefault(tag, {})[field] = v
    return out
