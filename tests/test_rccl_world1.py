"""The RCCL branch on a one-GPU box (-m gpu): `torch.distributed.run --nproc-per-node 1` with backend "nccl"
(= RCCL on ROCm).  World size 1 is the only RCCL configuration one MI355X allows (RCCL refuses two ranks per device,
tests/test_multi_rank_one_gpu.py covers world 2 over gloo): here every collective of freesplat_amd.view_sharding --
all_gather_into_tensor / reduce_scatter_tensor / all_reduce on DEVICE tensors, the side-stream AsyncViewGather with its
record_stream logic --, the decoder's sharded path and the sharded cost volume run through RCCL itself, and bench.py
takes its N>1 code path (`--single-rank-collectives`).  Still unmeasured at N>1."""
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


def _torchrun(script_args, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="8", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("FS_DIST_BACKEND", None)
    env.pop("FS_SHARE_GPU", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    return p.stdout


def test_collectives_decoder_and_cost_volume_on_rccl(hip_device):
    out = _torchrun([os.path.join(ROOT, "tests", "rccl_worker.py")])
    line = [l for l in out.splitlines() if l.startswith("RCCL_WORKER_RESULT ")][-1]
    res = json.loads(line[len("RCCL_WORKER_RESULT "):])
    assert res["backend"] == "nccl", res
    for k in ("gather_views", "async_view_gather", "reduce_scatter", "reduce_scatter_bucket_reuse", "all_reduce", "grad_exchange",
              "gather_views_autograd_grad", "gather_features_autograd_grad", "decoder_color_equal", "decoder_depth_equal",
              "decoder_color_only_equal", "replica_check_ran", "cv_rows_equal"):
        assert res[k] is True, (k, res)
    assert max(res["decoder_grad_err"].values()) < 2e-4, res        # float-atomic backward: the usual bar
    assert res["cv_feat_grad_err"] < 1e-4, res


@pytest.mark.parametrize("mode,extra", [("fwd", []), ("train", []), ("fwd", ["--gather-dtype", "uint8", "--gather-depth"])])
def test_bench_single_rank_takes_the_rccl_branch(hip_device, mode, extra):
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--views", "3",
                     "--workload", "c1_256x256_plumbing", "--mode", mode, "--sections", "raster", "--single-rank-collectives",
                     "--no-cpu-baseline", "--min-time", "0"] + extra)
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-2])      # (the full line; the compact one follows it)
    assert d["n_gpus"] == 1 and d["value"] > 0
    want = "reduce_scatter(gaussian grads)" if mode == "train" else ("all_gather(color,depth)[uint8]" if extra else "all_gather(color)")
    assert d["config"]["parallelism"] == f"view-sharded x1 + {want}", d["config"]
    if mode == "fwd":   # the N > 1 diagnostics, on RCCL: step / gather / exposed-gather times of the (one) rank
        per = d["multi_gpu"]["per_rank"]
        assert len(per) == 1 and per[0]["step_ms"] > 0 and per[0]["gather_ms"] > 0 and per[0]["exposed_gather_ms"] >= 0, per
        bytes_per = 3 * 256 * 256 * (4 if extra else 3) * (1 if extra else 4)
        assert d["multi_gpu"]["gather_bytes_per_rank_per_step"] == bytes_per, d["multi_gpu"]
    else:
        per = d["multi_gpu"]["per_rank"]
        assert len(per) == 1 and 0 < per[0]["grad_exchange_ms"] < per[0]["step_ms"], per
