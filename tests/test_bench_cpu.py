"""bench.py's self-launching path (``python bench.py --gpus N`` without an external torchrun) on CPU:
world 2 over gloo on a stub step -- the rendezvous, barrier + max-over-ranks timing and the JSON contract."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "3", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout       # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_self_launch_world2_gloo():
    out = _run(["--gpus", "2"])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["distributed"]["world_size_seen"] == 2 and out["distributed"]["backend"] == "gloo"
    per = out["distributed"]["per_rank_ms_per_step"]
    assert len(per) == 2 and abs(max(per) - out["ms_per_step"]) < 1e-6      # MAX over ranks
    assert out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["global_batch"] == 2 * 8 and out["config"]["parallelism"] == "dp2"
    assert "configs[2]" in out["config"]["workload"]                        # default = the north-star workload
    # the run proves its own exchange: identical weights on every rank, and what the exchange costs
    d = out["distributed"]
    assert d["params_identical"] is True and len(d["weights_fingerprint"]) == 2
    assert d["no_exchange_ms_per_step"] > 0 and "exposed_allreduce_ms_per_step" in d
    # N > 1: cross teaching (BASELINE config 5, the configuration defined on 8 GPUs) is timed after the default workload,
    # with its own proof of the exchange
    c = out["others"]["cross"]
    assert c["n_gpus"] == 2 and c["distributed"]["params_identical"] is True and len(c["per_rank_ms_per_step"]) == 2


@pytest.mark.timeout(900)
def test_missing_gradient_exchange_fails_the_run():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MIS_STUB_SKIP_SYNC"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "3", "--warmup", "1",
                        "--gpus", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and "data-parallel check FAILED" in (p.stderr + p.stdout)
    # a failed multi-process run still prints ONE JSON line: the status, the ranks' stderr tail, RCCL's warnings (none over gloo)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["value"] is None and out["n_gpus"] == 2 and "exit status" in out["error"]
    assert any("data-parallel check FAILED" in l for l in out["stderr_tail"]) and out["nccl_warnings"] == []


@pytest.mark.timeout(600)
def test_single_process_default_and_world_mismatch():
    out = _run([])
    assert out["n_gpus"] == 1 and out["distributed"]["world_size_seen"] == 1
    env = {k: v for k, v in os.environ.items()}
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--gpus", "4"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert p.returncode != 0 and "must equal --gpus" in p.stderr


def test_cpu_baseline_gives_up_steps_before_its_deadline_not_the_protocol_silently():
    """bench.cpu_baseline on a tiny geometry of the 2-D workload: without a deadline the BASELINE protocol (2 warm-up + 5 timed
    steps); with a deadline that has already passed, one warm-up and the minimum of timed steps -- and the dict says which."""
    import importlib.util
    import time
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    wl = dict(bench.WORKLOADS["unet2d"], shape=(4, 1, 32, 32), labeled=2, cpu_sample=(2, 1))
    threads = torch.get_num_threads()
    try:
        full = bench.cpu_baseline("unet2d", wl, cands=(2,), also_threads=None)
        assert (full["warmup_steps"], full["timed_steps"]) == (2, 5) and len(full["s_per_step_all"]) == 5
        assert full["cores"] == 2 and full["value"] > 0 and "2 warm-up + 5 timed" in full["sample"]
        late = bench.cpu_baseline("unet2d", wl, cands=(2,), also_threads=4, deadline=time.perf_counter() - 1.0)
        assert (late["warmup_steps"], late["timed_steps"]) == (1, 3) and len(late["s_per_step_all"]) == 3
        assert late["full_batch_at_other_thread_count"] is None and "1 warm-up + 3 timed" in late["sample"]
    finally:
        torch.set_num_threads(threads)
