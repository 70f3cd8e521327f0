"""bench.py's own launch paths on the GPU box, before the driver runs it at N = 1, 2, 4, 8: the N > 1 re-launch under
torch.distributed.run (two ranks sharing the one GPU of the test box, collectives over gloo) and the RCCL path of the
script itself (backend "nccl": init_process_group with device_id, barrier, MAX all-reduce of the clock, summary
all-reduce, destroy) with a world of one rank.  Run with `-m gpu`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--no-extras", "--inner", "300"]


def _bench(cmd, env_extra, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}: {r.stdout[-1000:]}"
    return json.loads(lines[0])


def test_bench_two_ranks_share_the_gpu_over_gloo():
    """`python bench.py --gpus 2`: the script's own re-launch under torch.distributed.run, one process per rank, both on
    cuda:0 (PTRWM_BENCH_BACKEND=gloo: the rehearsal mode for a box with fewer GPUs than ranks).  One JSON line from rank
    0, weak scaling (each rank its own 65 536 ladders with its own global ladder ids), the whole-job value within 2x of
    the one-rank reading on the same GPU (two processes time-share it)."""
    one = _bench([sys.executable, "bench.py", "--gpus", "1"] + ARGS, {})
    two = _bench([sys.executable, "bench.py", "--gpus", "2"] + ARGS, {"PTRWM_BENCH_BACKEND": "gloo", "MASTER_PORT": "29641"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["sharding"].startswith("2 x 65536 independent ladders")
    assert two["config"]["collective_backend"] == "gloo" and one["config"]["collective_backend"] is None
    assert two["summary"]["replicas"] == 2 * one["summary"]["replicas"] == 2 * 65536
    assert 0.5 < two["value"] / one["value"] < 2.0, (one["value"], two["value"])
    # the ranks sample different ladders (global ids 65536.. on rank 1): same statistics, not the same numbers
    assert two["summary"]["acceptance_rate_cold"] == pytest.approx(one["summary"]["acceptance_rate_cold"], rel=2e-2)
    assert two["summary"]["acceptance_rate_cold"] != one["summary"]["acceptance_rate_cold"]
    # the line diagnoses itself: one record per rank (device identity, its own launch times), the world size it saw
    assert one["config"]["world_size"] == 1 and two["config"]["world_size"] == 2
    assert [r["rank"] for r in two["ranks"]] == [0, 1] and len(one["ranks"]) == 1
    for r in two["ranks"]:
        assert r["pci"] and r["device_name"] and 0 < r["kernel_ms_min"] <= r["kernel_ms_median"] <= r["kernel_ms_max"]
        assert r["wall_s"] > 0
    assert two["distinct_devices"] == 1  # the rehearsal mode: both ranks on the test box's one GPU (RCCL would refuse)


def test_bench_under_torchrun_over_rccl():
    """The exact launch line the driver uses for N > 1, at one rank: python -m torch.distributed.run --nnodes=1
    --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P bench.py --gpus 1 ...  Backend "nccl" (RCCL):
    init_process_group(device_id=...), barriers, the MAX all-reduce of the clock and the summary all-reduce on device
    tensors, destroy_process_group - all executed by bench.py itself."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29643", "bench.py", "--gpus", "1"] + ARGS
    out = _bench(cmd, {})
    assert out["n_gpus"] == 1 and out["config"]["collective_backend"] == "nccl"
    assert out["summary"]["replicas"] == 65536 and out["value"] > 1e9
    assert out["roofline"]["kernel_ms"] > 0 and out["ms_per_step"] >= out["roofline"]["kernel_ms"] * 0.99
    assert out["config"]["world_size"] == 1 and out["distinct_devices"] == 1 and len(out["ranks"]) == 1
    r0 = out["ranks"][0]
    assert r0["rank"] == 0 and r0["pci"] and r0["kernel_ms_mean"] == pytest.approx(out["roofline"]["kernel_ms"], rel=1e-6)
