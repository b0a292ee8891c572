"""bench.py launches its own ranks: `python bench.py --gpus 2` without a launcher must end up with
world size 2 -- one process per rank over 127.0.0.1, the parent GPU-free -- and a WORLD_SIZE that contradicts --gpus
is an error, not a warning.  CPU only: --stub-backend swaps the HIP compute backend for one that exercises the phase
order and the all-reduce of droid_backends.ba_driver.ShardedBA over gloo (the reference's pattern: train.py:184-186
`mp.spawn(train, nprocs=args.gpus, ...)`)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--no-corr", "--no-cpu-baseline", "--no-extra", "--stub-backend", "--keyframes", "8", "--edges-total", "32",
          "--steps", "2", "--warmup", "1"]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_gpus_2_without_launcher_spawns_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + COMMON, env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE line, the parent relays it
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2
    assert line["config"]["stub_allreduce_sum"] == 3.0      # ranks contributed 1 + 2: the collective spanned both
    assert 0 < line["config"]["edges_local_rank0"] < line["config"]["edges_total"] == 32   # the graph was sharded
    assert line["valid"] is False


def test_single_rank_runs_in_process():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + COMMON, env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["config"]["edges_local_rank0"] == 32


def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + COMMON, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and "--gpus 2" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
