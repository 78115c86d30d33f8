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
