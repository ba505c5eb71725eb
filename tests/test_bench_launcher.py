"""bench.py --gpus N is authoritative: without a torch.distributed environment it starts the N
ranks itself; with one, WORLD_SIZE must equal N; it never prints an n_gpus: 1 line for a multi-GPU request.  CPU only:
--dry-run brings the ranks up over gloo and has them report who they are."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


@pytest.mark.timeout(300)
def test_gpus_2_spawns_two_ranks_dry_run_gloo():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--dry-run"], env=_clean_env(), capture_output=True, text=True,
                         timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["dry_run"] is True and rec["n_gpus"] == 2
    assert sorted(r["rank"] for r in rec["ranks"]) == [0, 1]
    assert all(r["world_size"] == 2 and r["master"].startswith("127.0.0.1:") for r in rec["ranks"])
    assert len({r["pid"] for r in rec["ranks"]}) == 2  # two processes, one per GPU


@pytest.mark.timeout(600)
def test_gpus_8_dry_run_gloo_line_stays_under_the_cap():
    """The shape the driver's scaling run has (8 ranks on one node), brought up here over gloo: eight processes, one line, under the
    8 KB cap of the contract line (round-5 review, items 1 and 8)."""
    import bench_line
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--backend", "gloo", "--dry-run"], env=_clean_env(), capture_output=True, text=True,
                         timeout=560, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < bench_line.MAX_LINE_BYTES
    rec = json.loads(lines[0])
    assert rec["dry_run"] is True and rec["n_gpus"] == 8 and sorted(r["rank"] for r in rec["ranks"]) == list(range(8))
    assert len({r["pid"] for r in rec["ranks"]}) == 8


@pytest.mark.timeout(120)
def test_world_size_must_equal_gpus():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=100, cwd=ROOT)
    assert out.returncode == 2 and "must agree" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]  # no JSON line that could be mistaken for a result


@pytest.mark.timeout(120)
def test_more_gpus_than_the_box_has_is_refused():
    import torch
    have = torch.cuda.device_count()
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(have + 1 if have else 2), "--steps", "1"], env=_clean_env(), capture_output=True,
                         text=True, timeout=100, cwd=ROOT)
    assert out.returncode == 2 and "refusing" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_count_parity_counts_what_assert_same_results_asserts():
    from tests.parity_util import count_parity
    free = np.uint64(0xFFFFFFFFFFFFFFFF)
    wk = np.array([[1, 2, 3], [4, 5, 6], [7, 8, free], [9, 10, 11]], dtype=np.uint64)
    wd = np.array([[0.1, 0.2, 0.3], [0.1, 0.2, 0.3], [0.5, 0.6, np.inf], [0.1, 0.2, 0.3]], dtype=np.float32)
    found = np.array([3, 3, 2, 3])
    gk, gd = wk.copy(), wd.copy()
    gk[1, 1], gk[1, 2] = 6, 5          # a swap the oracle sees as a near-tie
    gk[3, 2] = 99                      # an id the oracle puts elsewhere
    dist_of = lambda qi, key: {(1, 6): 0.2000001, (1, 5): 0.2999999, (3, 99): 0.9}[(qi, key)]
    r = count_parity(gk, gd, wk, wd, found, dist_of)
    assert r["rows"] == 4 and r["identical_rows"] == 2 and r["near_tie_positions"] == 2 and r["violations"] == 1 and r["violation_rows"] == 1
    assert r["first_violations"][0]["query"] == 3
    r = count_parity(gk, gd, wk, wd, found, dist_of, exact=True)   # integer metrics: no exception at all
    assert r["violations"] == 3
    gk2 = wk.copy()
    gk2[2, 2] = 12                     # the engine found one more than the oracle
    assert count_parity(gk2, wd, wk, wd, found, dist_of)["violations"] == 1
    gd2 = wd.copy()
    gd2[0, 0] = 0.1001
    assert count_parity(wk, gd2, wk, wd, found, dist_of)["violations"] == 1
