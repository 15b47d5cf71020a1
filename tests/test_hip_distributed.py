"""GPU, two ranks on the box's single card (gloo transport): the data-parallel training step and tools.training.

The reference is single-device (tools.py:266); SURVEY s.8e defines the partition.  What is pinned here: replicas start
from rank 0's state, the reduced gradient equals the sum of the ranks' independently computed gradients, weights stay
identical across ranks over eager and hipGraph-replayed steps, rank 0 writes the one checkpoint everybody loads."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(case, tmp, nproc=2, timeout=900):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dist_gpu_worker.py"), case, str(tmp)]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, p.stdout[-4000:]
    return json.loads(line[-1][7:])


def test_two_rank_training_step(tmp_path):
    r = _run("step", tmp_path)
    assert r["broadcast_params_equal"] and r["broadcast_buffers_equal"], r
    assert r["allreduce_rel_err"] < 1e-6, r               # same deterministic kernels; only the summation of 2 ranks differs
    assert r["buckets"] >= 2, r                           # several buckets went out during backward
    assert r["weights_equal_after_step1"] and r["weights_equal_after_graph_steps"] and r["momentum_equal"], r
    assert r["graph_segments"] >= 3 and r["graph_allreduces"] == r["buckets"], r
    assert r["graph_vs_eager_max_abs"] == 0.0, r          # a replayed step is bit-identical to the eager one
    assert r["finite"], r


def test_two_rank_self_check(tmp_path):
    """The start-up self-check of a multi-rank job (bench.py, Trainer.fit): passes on a healthy job without moving its state,
    and a replay that differs from the eager step on one rank switches every rank to eager launches."""
    r = _run("selfcheck", tmp_path)
    h = r["healthy"]
    assert h["graph_equals_eager"] and h["replicas_agree_eager"] and h["replicas_agree_replay"], r
    assert h["launch_mode"] == "hipGraph segments" and h["graph_segments"] >= 3, r
    assert r["state_restored"] and r["plan_kept"] and r["weights_equal_after_steps"], r
    f = r["faulty"]
    assert f["replicas_agree_eager"] and not f["graph_equals_eager"] and not f["replicas_agree_replay"], r
    assert f["launch_mode"].startswith("eager (self-check") and not r["faulty_use_graph"] and r["faulty_plans"] == 0, r
    assert r["faulty_weights_equal"], r
    o = r["one_sided"]                      # capture failed on rank 1 only: a collective decision, nobody replays
    assert not o["captured_on_every_rank"] and not o["graph_equals_eager"] and not o["replicas_agree_replay"], r
    assert o["launch_mode"].startswith("eager (self-check") and not r["one_sided_use_graph"] and r["one_sided_weights_equal"], r


def test_two_rank_tools_training(tmp_path):
    r = _run("fit", tmp_path)
    assert r["fit_params_equal"] and r["fit_bank_equal"], r
    assert r["ckpt_exists"] and r["ckpt_loads"] and r["hist_ok"], r


def test_graph_segments_with_rccl_allreduces_one_rank(tmp_path):
    """hipGraph segments interleaved with real RCCL all-reduces (a one-rank communicator: the gloo cases above cover two
    ranks but not ProcessGroupNCCL's streams and events)."""
    r = _run("rccl1", tmp_path, nproc=1)
    assert r["backend"] == "nccl", r
    assert r["buckets"] >= 2 and r["graph_segments"] >= 3 and r["graph_allreduces"] == r["buckets"], r
    assert r["eager_max_abs"] == 0.0 and r["replay_max_abs"] == 0.0 and r["momentum_max_abs"] == 0.0, r
    assert r["losses_equal"] and r["finite"], r
