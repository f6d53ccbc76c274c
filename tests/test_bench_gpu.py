"""bench.py on the GPU with live RCCL collectives at world size 1 (WSEG_FORCE_DIST=1: a one-rank nccl group): the branch the driver's
8-GPU run takes — weight broadcast, token all_gather, the headline timing, and (r06) BASELINE configs[3] / configs[4] through the product's
own multi-GPU entry points (bench.dist_configs -> whisperseg_amd.dist.segment_distributed / segment_batch_distributed) with the real
engine, checked against rank 0 decoding every shard alone.  A multi-GPU box is not available to the builder: everything that can be
exercised with one GPU is (the same code under gloo at world 2 / 8: tests/test_bench_dist_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_bench_dist_configs_with_live_rccl_at_world_1(gpu_lib):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WSEG_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29300 + os.getpid() % 200), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--model", "base", "--windows", "16", "--steps", "1", "--warmup", "1",
           "--gen-tokens", "8", "--no-cpu-baseline", "--no-extra", "--no-roofline", "--dist3-windows", "21", "--dist4-windows", "24"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=850)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["world_size"] == 1 and out["collectives"].startswith("RCCL") and out["rccl_ranks_seen"] == 1
    assert out["check"]["ok"] and out["dtype"] == "f16x3"      # the default mode: int16 operand rows go through the byte-wise weight broadcast
    c3, c4 = out["dist_configs"]
    assert c3["entry_point"].endswith("segment_distributed") and c3["windows"] == 21 and c3["windows_per_rank"] == [21]
    assert c4["entry_point"].endswith("segment_batch_distributed") and c4["windows"] == 24 and c4["windows_per_rank"] == [24]
    for c in (c3, c4):
        assert c["tokens_equal_to_rank0_alone"] is True and c["audio_sec_per_s"] > 0 and c["scaling"] == "strong"
