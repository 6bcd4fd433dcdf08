"""N>1 on a one-GPU box (-m gpu): two ranks launched exactly as the driver launches bench.py
(`python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 ...`) share cuda:0 and talk over gloo
(FS_DIST_BACKEND=gloo FS_SHARE_GPU=1; RCCL refuses two ranks on one device).  What runs is the real sharded code:
bench.py's N>1 branch and the process-group paths of the decoder and the cost volume, on the HIP kernels."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(script_args, timeout=600):
    env = dict(os.environ, FS_DIST_BACKEND="gloo", FS_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    return p.stdout


def test_sharded_decoder_and_cost_volume_two_ranks(hip_device):
    out = _torchrun([os.path.join(ROOT, "tests", "dist_worker.py")])
    line = [l for l in out.splitlines() if l.startswith("DIST_WORKER_RESULT ")][-1]
    for res in json.loads(line[len("DIST_WORKER_RESULT "):]):
        assert res["decoder_color_equal"] and res["decoder_depth_equal"], res      # gathered images == single-process render
        assert max(res["decoder_grad_err"].values()) < 2e-4, res                  # float-atomic backward: the usual bar
        assert res["cv_rows_equal"], res
        assert res["cv_feat_grad_err"] < 1e-4, res


def test_bench_two_ranks_chunked_gradient_exchange(hip_device):
    """bench.py --gpus 2 --mode train --grad-exchange chunked: the per-Gaussian backward pass runs in row chunks
    (fs_raster_backward_views_rows) with one reduce-scatter per chunk on a side stream; same JSON contract as the other exchanges."""
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--views", "3",
                     "--workload", "c1_256x256_plumbing", "--mode", "train", "--grad-exchange", "chunked", "--grad-chunks", "3"])
    lines = [l for l in out.splitlines() if l.startswith("{")]
    d = json.loads(lines[-2])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["parallelism"] == "view-sharded x2 + chunked(gaussian grads)", d["config"]
    per = d["multi_gpu"]["per_rank"]
    assert len(per) == 2 and all(0 <= r["grad_exchange_ms"] < r["step_ms"] for r in per), per


@pytest.mark.parametrize("mode", ["fwd", "train"])
def test_bench_two_ranks_runs_the_sharded_branch(hip_device, mode):
    """bench.py --gpus 2 end to end (small workload): NCCL-free init, view sharding, AsyncViewGather on a side stream
    (fwd) / reduce-scatter gradient exchange (train), max-over-ranks timing, one JSON line from rank 0."""
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--views", "3",
                     "--workload", "c1_256x256_plumbing", "--mode", mode])
    lines = [l for l in out.splitlines() if l.startswith("{")]
    d, compact = json.loads(lines[-2]), json.loads(lines[-1])      # the full line, then the compact one the driver parses
    assert len(lines[-1]) <= 4096 and compact["value"] == pytest.approx(d["value"], rel=1e-4) and compact["n_gpus"] == 2
    assert "roofline" in compact and set(compact["multi_gpu"]) == set(d["multi_gpu"]["per_rank"][0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    want = "reduce_scatter(gaussian grads)" if mode == "train" else "all_gather(color)"
    assert d["config"]["parallelism"] == f"view-sharded x2 + {want}", d["config"]
    assert d["roofline"]["launches"] == 2 * 3 * len(d["timed_regions_ms"])     # rank 0's own views of every timed region
    if mode == "fwd":
        # the N > 1 diagnostics: per rank one step on the render stream, the all-gather on the side stream, and the part of
        # it the rendering did not hide (the render stream's wait)
        per = d["multi_gpu"]["per_rank"]
        assert len(per) == 2 and all(r["step_ms"] > 0 and r["gather_ms"] > 0 and 0 <= r["exposed_gather_ms"] <= r["step_ms"] + r["gather_ms"]
                                     for r in per), per
    else:   # training: the gradient exchange alone, per rank (it runs on the render stream: all of it is exposed)
        per = d["multi_gpu"]["per_rank"]
        assert len(per) == 2 and all(0 < r["grad_exchange_ms"] < r["step_ms"] for r in per), per


def test_bench_gpus_2_launches_itself(hip_device):
    """`python bench.py --gpus 2` with no launcher environment (the driver's command shape) re-launches itself under
    torch.distributed.run with two ranks: same two lines from rank 0, exit status 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(FS_DIST_BACKEND="gloo", FS_SHARE_GPU="1", OMP_NUM_THREADS="8")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--views", "2",
                        "--workload", "c1_256x256_plumbing"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    compact = json.loads(lines[-1])
    assert compact["n_gpus"] == 2 and compact["value"] > 0 and compact["config"]["parallelism"].startswith("view-sharded x2")
    assert len(lines[-1]) <= 4096 and json.loads(lines[-2])["n_gpus"] == 2
