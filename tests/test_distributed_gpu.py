"""Two ranks of bench.py on ONE GPU (gloo for the collectives): the N > 1 code path of the step -- sharded batch, SyncBatchNorm
statistics exchange through codeps_bn_sync_combine, flat-gradient all-reduce, the rank-collective roofline pass -- has to run to
completion and report a finite rate.  (RCCL itself needs one GPU per rank; the driver exercises it at round end.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_complete_with_default_flags():
    env = dict(os.environ, CODEPS_ALL_RANKS_ON_GPU0="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo",
           "--encoder", "18", "--height", "128", "--width", "256", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["launch"] == "eager"
    assert d["value"] > 0 and d["value"] == d["value"]
    assert d["roofline"]["achieved"] > 0
