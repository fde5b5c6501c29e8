"""bench.py on a GPU box, through the plain spelling the driver uses (`python bench.py --gpus N ...`): the self-launched
multi-rank path in rehearsal mode (every rank on GPU 0, gloo collectives -- the plumbing, not a measurement), the
strong-scaling spelling of BASELINE config 4 at a reduced size, and the error a box with too few GPUs gets."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_rehearsal_of_two_ranks_through_the_plain_spelling():
    r, out = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "cora"], {"HCSPMM_BENCH_REHEARSAL": "1"})
    assert r.returncode == 0 and out is not None, r.stderr[-2000:]
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "REHEARSAL" in out["data"]
    assert out["distributed"] == {"world_size": 2, "backend": "gloo", "visible_gpus": torch.cuda.device_count(), "launcher": "self"}
    assert out["value"] > 0 and out["communication"]["bytes_received_per_rank_per_step"] == 10000 * 32 * 4
    assert out["config"]["entries_per_gpu"] > 0 and out["unit"] == "edge*dim/s"


def test_strong_scaling_spelling_describes_one_graph_at_every_world_size():
    args = ["--workload", "products", "--strong-scale", "64", "--steps", "3", "--warmup", "1", "--no-pmc", "--no-cpu-baseline"]
    r1, one = _bench(["--gpus", "1"] + args)
    assert r1.returncode == 0 and one is not None, r1.stderr[-2000:]
    r2, two = _bench(["--gpus", "2"] + args, {"HCSPMM_BENCH_REHEARSAL": "1"})
    assert r2.returncode == 0 and two is not None, r2.stderr[-2000:]
    assert one["scaling"] == two["scaling"] == "strong" and one["n_gpus"] == 1 and two["n_gpus"] == 2
    total = lambda o: round(o["value"] * o["ms_per_step"] * 1e-3 / o["config"]["dim"])  # entries all ranks processed per step
    assert total(one) == total(two) == one["config"]["entries_per_gpu"]
    assert two["config"]["nodes_per_gpu"] * 2 == one["config"]["nodes_per_gpu"]


def test_more_ranks_than_gpus_is_refused_with_the_count():
    n = torch.cuda.device_count()
    r, out = _bench(["--gpus", str(n + 1), "--steps", "2"])
    assert r.returncode == 3 and out is None
    assert "needs %d visible GPUs, found %d" % (n + 1, n) in r.stderr
