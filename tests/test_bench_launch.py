"""bench.py's launch contract, on the CPU container: `--gpus N` without a launcher starts N ranks (or refuses), a
mismatching WORLD_SIZE is refused, and the final line stays under the 4 KB the driver's capture holds.
Reference analogue of the self-started workers: syncopy/tests/conftest.py:41,60 (dd.LocalCluster(n_workers=...))."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + argv, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], SPY_BENCH_LAUNCH_TEST="1")
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                  # ONE JSON line: rank 0's
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    # max over ranks: rank 1 sleeps 2 ms per step, rank 0 only 1 ms
    assert line["ms_per_step"] >= 2.0


def test_gpus_3_spawns_three_ranks():
    p = _run(["--gpus", "3", "--steps", "2", "--warmup", "0"], SPY_BENCH_LAUNCH_TEST="1")
    assert p.returncode == 0, p.stderr
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["ranks_seen"] == 3


def test_under_a_launcher_no_second_spawn():
    # WORLD_SIZE given (torch.distributed.run's environment): the process IS a rank; one rank of one
    p = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], SPY_BENCH_LAUNCH_TEST="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert p.returncode == 0, p.stderr
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_refuses_more_gpus_than_visible():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        return
    p = _run(["--gpus", "8", "--steps", "1"])
    assert p.returncode == 2 and "{" not in p.stdout, (p.returncode, p.stdout)       # no line with another n_gpus


def test_refuses_world_size_mismatch():
    p = _run(["--gpus", "8"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert p.returncode == 2 and "{" not in p.stdout


def test_rank_failure_is_reported(tmp_path):
    # a rank that dies takes the launch down with a non-zero exit code (no hang in the barrier)
    p = _run(["--gpus", "2", "--steps", "1", "--bogus-flag"], SPY_BENCH_LAUNCH_TEST="1")
    assert p.returncode != 0


def test_final_line_stays_under_4k(tmp_path, monkeypatch, capsys):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    big = {"name": "x" * 300, "note": "y" * 3000}
    line = {"metric": "m", "value": 1.0, "unit": "trials/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 20.0,
            "roofline": {"bound": "mfma", "achieved": 1.0, "peak": 2.0, "frac": 0.5, "traffic": 3.0e10},
            "cpu_baseline": {"value": 0.5, "cores": 1}, "secondary": {"k%d" % i: {"us_per_trial": 1.0, "note": "z" * 200} for i in range(40)}}
    bench.emit(line, {"secondary": [big] * 20})
    out = capsys.readouterr().out.strip()
    assert len(out) <= 4000 and "\n" not in out
    got = json.loads(out)
    assert got["roofline"]["frac"] == 0.5 and got["cpu_baseline"]["cores"] == 1 and "secondary" not in got
    detail = json.load(open(os.path.join(str(tmp_path), got["detail"])))
    assert len(detail["secondary"]) == 20
