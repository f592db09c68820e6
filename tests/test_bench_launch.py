"""bench.py's self-launching multi-rank path: `python bench.py --gpus N` from a plain shell
must start one rank per GPU itself, relay rank 0's JSON line and propagate failures.

* CPU: the launcher with `--launch-check` (process group only, gloo) at 2 ranks; a failing rank => non-zero exit.
* GPU (-m gpu): the real bench at 2 ranks sharing the one card of the test box (UNETK_DIST_BACKEND=gloo over device
  tensors -- RCCL refuses two ranks on one device), tiny shape: one JSON line, n_ranks_seen == 2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, env=env, timeout=timeout, cwd=ROOT)


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_launcher_spawns_ranks_and_relays_rank0_line():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _line(r.stdout)
    assert out["metric"] == "launch-check" and out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["ranks"] == [0, 1]
    assert r.stdout.strip().count("\n") == 0            # ONE line on stdout; everything else went to stderr


def test_launcher_single_rank_needs_no_spawn():
    r = _run(["--gpus", "1", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_ranks_seen"] == 1


@pytest.mark.skipif(__import__("torch").cuda.device_count() > 0, reason="needs a box WITHOUT a GPU: the ranks must fail")
def test_launcher_propagates_rank_failure():
    """Without a GPU every rank dies on the `needs MI355X GPUs` assertion: the parent must exit non-zero, print no
    result line, and say which launch failed."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert '{"metric"' not in r.stdout
    assert "a rank failed" in r.stderr


def test_mismatched_world_is_refused():
    r = _run(["--gpus", "2", "--launch-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_from_a_plain_shell_gloo_rehearsal():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--size", "64", "--no-cpu-baseline"],
             {"UNETK_DIST_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _line(r.stdout)
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["dist_backend"] == "gloo"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak"
    assert out["value"] > 0 and out["rank_ms_per_step_min"] <= out["rank_ms_per_step_max"] <= out["ms_per_step"] * 1.5
    assert "roofline" in out and out["roofline"]["achieved"] > 0


@pytest.mark.gpu
def test_bench_runs_the_rccl_data_parallel_path_in_a_world_of_one():
    """the production data-parallel path -- process group on RCCL ("nccl"), GradBuckets launched from backward
    hooks, finish() joining the compute stream, 1/N in the optimiser kernel -- driven through bench.py itself on the one GPU
    of the test box; the line carries the diagnostics an 8-GPU run will be read with."""
    r = _run(["--gpus", "1", "--dp-rehearsal", "--steps", "3", "--warmup", "1", "--batch", "4", "--size", "64", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = _line(r.stdout)
    dp = out["data_parallel"]
    assert dp["backend"] == "nccl" and dp["buckets"] >= 3
    # both flat gradient buffers: the 31 037 763 trainable values + under 4 values of alignment per variable
    assert 31037763 * 4 <= dp["allreduce_bytes"] < (31037763 + 4 * 200) * 4
    assert sum(dp["bucket_bytes"]) == dp["allreduce_bytes"]
    # real overlap: buckets go out while gradients are still outstanding, in arrival order, and at most the last two (the
    # first layers' filters and norm parameters, complete only with the very last gradient) wait for the end of backward
    prog = dp["bucket_launch_progress"]
    assert all(p is not None for p in prog) and prog == sorted(prog)
    # (round 5: three buckets -- 86 MB complete when 55 % of the variables have arrived, 35 MB, and a <= 4 MB tail with gamma / beta)
    assert prog[0] < 0.6 and sum(1 for p in prog if p < 1.0) >= dp["buckets"] - 2
    assert dp["buckets_fired_in_backward"] == sum(1 for p in prog if p < 1.0)
    assert dp["allreduce_exposed_ms"] is not None and dp["allreduce_exposed_ms"] >= 0.0
    assert dp["compute_only_ms_per_step"] > 0 and 0.2 < dp["dp_efficiency_vs_compute_only"] < 1.5
