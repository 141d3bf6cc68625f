"""bench.py --gpus N without a launcher starts its own ranks (one process per GPU).  On this CPU box every rank must get
as far as fv_ctx_create and fail there loudly — no CPU fallback — and the parent must come back non-zero, quickly, with no
rank left behind."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


@pytest.mark.skipif(_has_gpu(), reason="needs a box without a GPU: the ranks are expected to fail at fv_ctx_create")
def test_bench_spawns_its_ranks_and_fails_loudly_without_a_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "no HIP device" in r.stderr, r.stderr[-2000:]
    assert "rank" in r.stderr and "stopping the other ranks" in r.stderr or r.stderr.count("no HIP device") >= 2, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no JSON line from a failed run"
    assert time.time() - t0 < 280


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr


def test_roofline_traffic_file_is_stamped_with_the_kernel_it_was_measured_on():
    t = json.load(open(os.path.join(ROOT, "profiles", "spmv_traffic.json")))
    for size, rec in t.items():
        if size == "note":
            continue
        assert {"form", "kernel", "bytes", "source"} <= set(rec), size
        assert os.path.exists(os.path.join(ROOT, rec["source"])), rec["source"]
