"""N>1 path on CPU: two gloo processes exercise the sharding + the single gradient all-reduce + max-over-ranks timing."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_covers_everything_once():
    from graphflow_amd.dist import shard
    for n in (0, 1, 7, 8, 1024, 8192):
        for w in (1, 2, 3, 8):
            spans = [shard(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_two_rank_gradient_allreduce_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r)
        from graphflow_amd import dist as gd
        d = gd.init(backend="gloo")
        world, rank, _ = gd.env_world()
        lo, hi = gd.shard(10, rank, world)
        # each rank's "gradient" = sum of its shard's per-item gradients
        per_item = np.arange(10, dtype=np.float32)[:, None] * np.ones((1, 5), dtype=np.float32)
        g = torch.tensor(per_item[lo:hi].sum(axis=0))
        gd.allreduce_sum_(g, d)
        assert np.allclose(g.numpy(), per_item.sum(axis=0)), g
        t = gd.max_over_ranks(1.0 + rank, d)
        assert t == float(world)
        d.barrier()
        d.destroy_process_group()
        print("rank", rank, "ok")
    """ % ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_bench_gpus_2_spawns_two_ranks_plumbing():
    """`python bench.py --gpus 2` outside a launcher must start two ranks itself and print ONE line with n_gpus 2 (round 1
    parsed --gpus and ignored it).  --plumbing runs the real entry point's launcher, rank bookkeeping, the single exchange
    (gloo here, RCCL behind gf_dist_* on GPUs) and max-over-ranks timing without kernels."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and "plumbing" in out["metric"]


def test_bench_refuses_a_world_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--plumbing"], capture_output=True, text=True,
                       env=env, timeout=120)
    assert r.returncode != 0 and "must agree" in (r.stdout + r.stderr)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if not torch.cuda.is_available():   # asking for GPUs that are not there fails loudly instead of running fewer ranks
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode != 0 and "HIP device" in (r.stdout + r.stderr)


def _bench_plumbing(n, extra=()):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--plumbing", "--steps", "3", "--warmup", "1", *extra],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_gpus_8_plumbing_weak_and_strong():
    """The driver's 8-GPU command line end to end on CPU: eight ranks started by bench.py itself, one line from rank 0, the weak and the
    strong batch split, and the fields a multi-GPU line carries (`collective`, `scaling`)."""
    out = _bench_plumbing(8)
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["per_rank_units"] == 1024
    assert out["collective"]["gloo_ranks_seen"] == 8 and out["collective"]["expected"] == 8
    assert out["config"]["shard_of_rank0"] == [0, 1024]
    out = _bench_plumbing(8, ("--scaling", "strong"))
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["per_rank_units"] == 1024   # 8192 / 8
    out = _bench_plumbing(4, ("--scaling", "strong", "--batch", "4096"))
    assert out["n_gpus"] == 4 and out["config"]["per_rank_units"] == 1024


def test_watchdog_entry_points_exist_and_are_no_ops_without_a_communicator():
    """gf_dist_quiesce / GF_ERR_TIMEOUT are part of the ABI (include/gf_hip.h); the bounded waits themselves need RCCL and two GPUs
    (tests/test_dist_gpu.py holds the one-rank path)."""
    from graphflow_amd import _lib
    assert _lib.GF_ERR_TIMEOUT == 5
    hdr = open(os.path.join(ROOT, "include", "gf_hip.h")).read()
    assert "GF_ERR_TIMEOUT = 5" in hdr and "gf_dist_quiesce" in hdr and "GF_DIST_TIMEOUT_S" in hdr
