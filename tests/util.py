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


def golden_cases(golden, prefix):
    """{tag: {field: array}} for every fixture whose tag starts with prefix."""
    out = {}
    for k, v in golden.items():
        tag, field = k.split("__")
        if tag.startswith(prefix):
            out.setdefault(tag, {})[field] = v
    return out
