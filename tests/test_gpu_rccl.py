"""RCCL on the MI355X before the driver's 8-GPU run needs it (trainer.py:215-222): the product's collectives
(irbpp_amd/distributed.py) through a real single-rank `nccl` communicator, and bench.py's timed loop with the process
group forced at world size 1 (barrier, agreement on the timed blocks, the final all-reduce of the episode totals)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch
from irbpp_amd import distributed as D
rank, world, local = D.init_from_env("nccl", force=True)
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
import torch.distributed as dist
t = torch.tensor([3.0, 1.5, 12.0, 7.25], dtype=torch.float64, device=dev)
out = D.reduce_totals(t.clone())
D.barrier(dev)
mx = D.max_over_ranks(2.5, dev)
names = D.gather_strings("rank %%d" %% rank)
maps = open("/proc/self/maps").read()
libs = sorted({l.split()[-1] for l in maps.splitlines() if "rccl" in l.lower() or "nccl" in l.lower()})
print(json.dumps({"backend": dist.get_backend(), "world": dist.get_world_size(), "totals": out.cpu().tolist(), "max": mx,
                  "names": names, "libs": libs}))
dist.destroy_process_group()
""" % ROOT


def _env():
    import socket
    with socket.socket() as sk:                      # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_collectives_run_through_a_single_rank_rccl_communicator():
    res = subprocess.run([sys.executable, "-c", _SCRIPT], env=_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])    # (librccl prints its path on stdout too)
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["totals"] == [3.0, 1.5, 12.0, 7.25] and out["max"] == 2.5 and out["names"] == ["rank 0"]
    assert any("rccl" in l.lower() for l in out["libs"]), out["libs"]       # librccl is what `nccl` loads on ROCm
    log = os.path.join(ROOT, "gpurun_out", "rccl_single_rank.json")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    json.dump(out, open(log, "w"), indent=1)


def test_bench_timed_loop_under_a_forced_rccl_process_group():
    env = _env()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-process-group", "--no-extra", "--no-cpu-baseline",
                          "--bins", "2048", "--prefill", "20", "--warmup", "5", "--steps", "10", "--min-seconds", "0.2"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])             # bench.py keeps its JSON line the last one on stdout
    assert line["ranks"]["process_group"] is True and line["ranks"]["backend"] == "nccl" and line["n_gpus"] == 1
    assert line["value"] > 1e6 and line["episodes"]["finished_since_reset"] >= 0
