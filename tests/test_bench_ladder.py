"""CPU-only: bench.py's multi-GPU watchdog and launch ladder.  The attempt commands are replaced through the test hook
LJMD_BENCH_ATTEMPT_CMD_<MODE> by stand-ins -- a child that sleeps for ever (what a hung RCCL initialisation looks like
from outside), a child that dies, a child that prints a line -- so that what is exercised is the parent: per-rung
deadlines, ending the whole process tree of a hung attempt, moving down the ladder, and printing exactly ONE JSON line
whatever happened above the rung that worked.  The real rungs run on the GPU box (tests/test_gpu_sharded.py)."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

SLEEPER = [sys.executable, "-c",
           # a hung attempt with a grandchild in its own session (as torch.distributed.run's workers are)
           "import subprocess, sys, time; "
           "subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'], start_new_session=True); "
           "print('attempt started', flush=True); time.sleep(600)"]
FAILER = [sys.executable, "-c", "import sys; print('boom', file=sys.stderr); sys.exit(3)"]
SILENT = [sys.executable, "-c", "print('no json here')"]
LINE = {"metric": "md_steps_per_sec_n262144_fp64", "value": 123.0, "unit": "steps/s", "n_gpus": 2,
        "config": {"workload": "stand-in", "exchange": "peer copies"}}
PRINTER = [sys.executable, "-c", f"print('chatter'); print({json.dumps(json.dumps(LINE))})"]


def _run(env_extra, timeout=120, launcher_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE") and
           not k.startswith("LJMD_BENCH_")}
    env.update(env_extra)
    env.update(launcher_env or {})
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=timeout)
    return out, time.monotonic() - t0


def _json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]


def test_hung_first_rung_is_killed_and_the_ladder_still_prints_one_line():
    import psutil
    before = {p.pid for p in psutil.process_iter()}
    out, secs = _run({"LJMD_BENCH_DEADLINES": "3,20,20,20", "LJMD_BENCH_GRACE_S": "0",
                      "LJMD_BENCH_ATTEMPT_CMD_RANKS_RCCL": json.dumps(SLEEPER),
                      "LJMD_BENCH_ATTEMPT_CMD_MULTI_RCCL": json.dumps(FAILER),
                      "LJMD_BENCH_ATTEMPT_CMD_MULTI_COPY": json.dumps(PRINTER),
                      "LJMD_BENCH_ATTEMPT_CMD_MULTI_HOST": json.dumps(SLEEPER)})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = _json_lines(out.stdout)
    assert len(lines) == 1 and len([ln for ln in out.stdout.splitlines() if ln.strip()]) == 1   # stdout = the ONE line
    line = lines[0]
    assert line["value"] == 123.0 and line["config"]["launch_mode"] == "multi-copy"
    ladder = line["config"]["ladder"]
    assert [a["mode"] for a in ladder] == ["ranks-rccl", "multi-rccl", "multi-copy"]
    assert [a["outcome"] for a in ladder] == ["timeout", "exit 3", "ok"]
    assert secs < 60
    assert "deadline of 3 s passed" in out.stderr and "chatter" in out.stderr
    # the hung attempt's tree is gone, the grandchild in its own session included
    time.sleep(0.5)
    left = [p for p in psutil.process_iter(["pid", "cmdline"]) if p.info["pid"] not in before and
            p.info["cmdline"] and "time.sleep(600)" in " ".join(p.info["cmdline"])]
    assert not left, left


def test_a_rung_that_dies_does_not_leave_its_detached_workers_behind():
    """A rung whose launcher exits non-zero while a worker in a session of its own lives on (what torch.distributed.run
    can leave behind): once the launcher has been reaped /proc no longer links the worker to it, so the watchdog must
    have seen it while the launcher was alive (DescendantTracker) to end it before the next rung starts."""
    import psutil
    dying = [sys.executable, "-c",
             "import subprocess, sys, time; "
             "subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(601)'], start_new_session=True); "
             "time.sleep(2.0); sys.exit(3)"]
    before = {p.pid for p in psutil.process_iter()}
    out, _ = _run({"LJMD_BENCH_DEADLINES": "20", "LJMD_BENCH_LADDER": "ranks-rccl,multi-copy",
                   "LJMD_BENCH_ATTEMPT_CMD_RANKS_RCCL": json.dumps(dying),
                   "LJMD_BENCH_ATTEMPT_CMD_MULTI_COPY": json.dumps(PRINTER)})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = _json_lines(out.stdout)
    assert len(lines) == 1 and [a["outcome"] for a in lines[0]["config"]["ladder"]] == ["exit 3", "ok"]
    time.sleep(0.5)
    left = [p for p in psutil.process_iter(["pid", "cmdline"]) if p.info["pid"] not in before and
            p.info["cmdline"] and "time.sleep(601)" in " ".join(p.info["cmdline"])]
    assert not left, left


def test_every_rung_fails_exit_code_and_no_line():
    out, _ = _run({"LJMD_BENCH_DEADLINES": "20", "LJMD_BENCH_LADDER": "ranks-rccl,multi-copy",
                   "LJMD_BENCH_ATTEMPT_CMD_RANKS_RCCL": json.dumps(FAILER),
                   "LJMD_BENCH_ATTEMPT_CMD_MULTI_COPY": json.dumps(SILENT)})
    assert out.returncode == 1
    assert not _json_lines(out.stdout)
    assert "no rung delivered a bench line" in out.stderr and '"no line"' in out.stderr and '"exit 3"' in out.stderr


def test_under_a_launcher_only_rank_zero_goes_down_the_ladder():
    """What the driver does: torch.distributed.run starts one bench.py per rank.  Each is a watchdog for its own
    rank's child in the first rung; when that rung hangs, rank 1's watchdog ends its child and exits 0 (a non-zero
    exit would make the launcher tear the job down), rank 0's carries on alone and prints the line."""
    common = {"LJMD_BENCH_DEADLINES": "3,20", "LJMD_BENCH_GRACE_S": "0", "LJMD_BENCH_LADDER": "ranks-rccl,multi-copy",
              "LJMD_BENCH_ATTEMPT_CMD_RANKS_RCCL": json.dumps(SLEEPER),
              "LJMD_BENCH_ATTEMPT_CMD_MULTI_COPY": json.dumps(PRINTER)}
    out1, _ = _run(common, launcher_env={"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2"})
    assert out1.returncode == 0 and not out1.stdout.strip()
    out0, _ = _run(common, launcher_env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert out0.returncode == 0
    lines = _json_lines(out0.stdout)
    assert len(lines) == 1 and lines[0]["config"]["launch_mode"] == "multi-copy"
    assert [a["outcome"] for a in lines[0]["config"]["ladder"]] == ["timeout", "ok"]


def test_real_rungs_without_a_gpu_fail_loudly_and_quickly():
    """No stand-ins: on a box without a HIP device every real rung fails in ljmd_create / ljmd_create_multi (there is
    no CPU path), the watchdog reports each and exits non-zero without a line."""
    import ljmd_amd  # noqa: F401
    from ljmd_amd import _lib
    if _lib.load().ljmd_device_count() > 0:
        pytest.skip("a HIP device is present: the real rungs are covered by tests/test_gpu_sharded.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE") and
           not k.startswith("LJMD_BENCH_")}
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--particles", "4096", "--no-cpu-baseline", "--no-liquid"], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode != 0
    assert not _json_lines(out.stdout)
    for mode in ("ranks-rccl", "multi-rccl", "multi-copy", "multi-host"):
        assert f"rung '{mode}' gave no line" in out.stderr, mode
    assert "torch.distributed.run" in out.stderr and "--nproc-per-node=2" in out.stderr
    assert "no HIP device available" in out.stderr
