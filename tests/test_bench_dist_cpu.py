"""bench.py's launch + distributed path on CPU (gloo, world size 2) with a stub engine (tests/bench_stub_driver.py):
`--gpus 2` outside torchrun starts two ranks itself; the JSON line reports the real world size and one entry per rank;
rank 0's weights reach every rank; a WORLD_SIZE / --gpus mismatch is an error, not a warning."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

DRIVER = os.path.join(ROOT, "tests", "bench_stub_driver.py")
ARGS = ["--device", "cpu", "--steps", "2", "--warmup", "1", "--windows", "5", "--model", "tiny", "--gen-tokens", "5",
        "--sr", "16000", "--spec-time-step", "0.001", "--no-cpu-baseline", "--dist3-windows", "20", "--dist4-windows", "24"]


def check_dist_configs(out, world):
    """r06: under world > 1 the bench sends BASELINE configs[3] (one recording, its windows over the ranks: dist.segment_distributed) and
    configs[4] (a mixed-rate clip batch, its pooled window list over the ranks: dist.segment_batch_distributed) through the product's
    own multi-GPU entry points on every rank, and rank 0 re-decodes every rank's shard alone: the gathered token ids must be identical."""
    dc = out["dist_configs"]
    assert isinstance(dc, list) and len(dc) == 2, dc
    assert out["extra"]["dist_configs"] == dc
    c3, c4 = dc
    assert c3["entry_point"].endswith("segment_distributed") and c4["entry_point"].endswith("segment_batch_distributed")
    per = -(-20 // world)
    assert c3["windows"] == 20 and c3["windows_per_rank"] == [max(0, min(per, 20 - r * per)) for r in range(world)]
    assert c4["windows"] == 24 and sum(c4["windows_per_rank"]) == 24 and len(c4["windows_per_rank"]) == world
    for c in dc:
        assert c["scaling"] == "strong" and c["tokens_equal_to_rank0_alone"] is True and c["audio_sec_per_s"] > 0 and c["segments"] > 0


def clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


@pytest.mark.timeout(300)
def test_gpus_2_self_launches_two_ranks():
    res = subprocess.run([sys.executable, DRIVER, "--gpus", "2"] + ARGS, env=clean_env(), capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    out = last_json(res.stdout)
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["scaling"] == "weak"
    assert sorted(r["rank"] for r in out["ranks"]) == [0, 1]
    assert len({r["pid"] for r in out["ranks"]}) == 2
    assert out["collectives"] == "gloo"
    # whole-job value: 2 ranks x 5 windows x 1 s per step
    assert out["value"] == pytest.approx(2 * 5 * 1.0 * 2 / (out["ms_per_step"] * 2 / 1e3), rel=1e-6)
    assert out["config"]["windows_per_gpu"] == 5
    check_dist_configs(out, 2)


@pytest.mark.timeout(600)
def test_gpus_8_first_contact_shape():
    """The driver's 8-GPU scaling run is the first time RCCL meets more than one rank (no multi-GPU box is available to the
    builder): everything that does not need a GPU is driven here at world size 8 — port selection and the self-launch of 8 ranks,
    the WORLD_SIZE check, rank 0's weights reaching all 8 ranks, 8 x per-rank rows through the token all_gather, the 8-entry
    `ranks` list, the rank-0-only self-check beside 7 ranks waiting at the final barrier, and `rccl_ranks_seen`."""
    args = [a for a in ARGS] + ["--check-on-cpu", "--check-windows", "3"]
    res = subprocess.run([sys.executable, DRIVER, "--gpus", "8"] + args, env=clean_env(), capture_output=True, text=True, timeout=560)
    assert res.returncode == 0, res.stderr[-3000:]
    out = last_json(res.stdout)
    assert out["n_gpus"] == 8 and out["world_size"] == 8 and out["rccl_ranks_seen"] == 8 and out["scaling"] == "weak"
    assert sorted(r["rank"] for r in out["ranks"]) == list(range(8)) and len({r["pid"] for r in out["ranks"]}) == 8
    assert sorted(r["local_rank"] for r in out["ranks"]) == list(range(8))
    assert out["config"]["windows_per_gpu"] == 5 and out["config"]["parallelism"] == "clip-sharded x8"
    assert out["value"] == pytest.approx(8 * 5 * 1.0 * 2 / (out["ms_per_step"] * 2 / 1e3), rel=1e-6)
    assert out["check"]["ok"] and out["check"]["deterministic"]      # the stub decodes 15 + 0 only once rank 0's weights arrived everywhere
    check_dist_configs(out, 8)      # 20 windows over 8 ranks: 3 3 3 3 3 3 2 0 — the last rank idles through every collective


@pytest.mark.timeout(300)
def test_single_process_default_and_mismatch():
    res = subprocess.run([sys.executable, DRIVER] + ARGS, env=clean_env(), capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-3000:]
    out = last_json(res.stdout)
    assert out["n_gpus"] == 1 and out["world_size"] == 1 and len(out["ranks"]) == 1 and out["dist_configs"] is None
    env = clean_env()
    env.update(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    res = subprocess.run([sys.executable, DRIVER, "--gpus", "1"] + ARGS, env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 2 and "WORLD_SIZE" in res.stderr


@pytest.mark.timeout(300)
def test_self_check_runs_on_rank_0_without_a_collective():
    """the self-check runs on rank 0 only, so it must not enter a collective (its step used to call the
    token all_gather while the other ranks were already in the final barrier: a mismatched collective that hangs under RCCL).
    Two gloo ranks with the check enabled must finish and report it."""
    args = [a for a in ARGS] + ["--check-on-cpu", "--check-windows", "3"]
    res = subprocess.run([sys.executable, DRIVER, "--gpus", "2"] + args, env=clean_env(), capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    out = last_json(res.stdout)
    assert out["world_size"] == 2
    chk = out["check"]
    assert chk is not None and chk["ok"] and chk["deterministic"] and chk["subset_windows"] == 3
    assert chk["beams_equal_at_first_step"] and chk["first_token_equal_to_f32"] == "3/3"
