"""GPU (-m gpu): the N > 1 code of bench.py, rehearsed on the one GPU a box has — a regression guard until a multi-GPU node runs it.

`LFI_BENCH_REHEARSE=1` puts both ranks on cuda:0 and uses gloo for the process group (RCCL refuses two ranks on one device), so what
executes is everything but the RCCL transport: torch.distributed.run's rendezvous, the per-rank view ranges / row bands, the all-gather
distribution, the barrier + max-over-ranks timing and the JSON line.  The ranks are fresh child processes (the launcher, then one
process per rank); the line says that it is a rehearsal.  Real N > 1 runs are the driver's (SCALE)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench_two_ranks(*extra):
    env = dict(os.environ, LFI_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--prewarm-ms", "0", "--no-cpu-baseline", "--no-also", *extra]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]            # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_rank_rehearsal_config4_allgather(gpu):
    """BASELINE config 4 strong-scaled over two ranks: one 256-view trajectory @4K, 128 views per rank, the grid all-gathered from 1/G
    per rank."""
    line = _bench_two_ranks("--config", "4", "--distribute", "allgather")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 2
    cfg = line["config"]
    assert cfg["baseline_config"] == 4 and cfg["views_per_gpu"] == 128 and cfg["images"] == 64
    assert "gloo rehearsal" in cfg["parallelism"] and "all-gathered" in cfg["parallelism"]
    assert cfg["view_ranges"] == [[0, 128], [128, 256]]
    assert cfg["multi_gpu_hardware_runs"].startswith("none")
    assert line["value"] > 0 and line["unit"] == "views/s" and line["ms_per_step"] > 0
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1.0


def test_two_rank_rehearsal_row_bands(gpu):
    """One config-2 render split into two row bands: no collective, each rank holds band + halo rows of every image."""
    line = _bench_two_ranks("--shard", "rows")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    cfg = line["config"]
    assert "rows sharded over 2 GPU(s)" in cfg["parallelism"] and "no collective" in cfg["parallelism"]
    assert cfg["row_bands"] == [[0, 540], [540, 1080]]
    assert line["value"] > 0


def test_two_rank_rehearsal_default_run(gpu):
    """The contract's own command (`bench.py --gpus 2 --steps K --warmup W`, what the driver's SCALE run launches): BASELINE config 2 per
    rank, weak-scaled, the grid broadcast once from rank 0."""
    line = _bench_two_ranks()
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2 and line["warmup"] == 1
    cfg = line["config"]
    assert cfg["baseline_config"] == 2 and cfg["views_per_gpu"] == 64 and cfg["images"] == 64
    assert "broadcast" in cfg["parallelism"] and "gloo rehearsal" in cfg["parallelism"]
    assert cfg["view_ranges"] == [[0, 64], [64, 128]]
    assert line["metric"].startswith("novel views/sec") and line["unit"] == "views/s" and line["higher_is_better"] is True
    assert line["value"] > 0 and abs(line["value"] - 128 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]   # both ranks' views
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1.0 and line["vs_baseline"] is None


def test_more_ranks_than_gpus_fail_fast(gpu):
    """`bench.py --gpus N` on a node with fewer GPUs (no rehearsal mode): every rank stops in set-up with a message that names the GPU
    count, instead of dying inside set_device / the RCCL initialisation (round 4)."""
    import torch
    n = torch.cuda.device_count()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE=str(n + 1), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.pop("LFI_BENCH_REHEARSE", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert res.returncode == 3, (res.returncode, res.stderr[-1000:])
    assert f"needs {n + 1} visible GPUs, this node shows {n}" in res.stderr
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]
