"""Data parallelism with world size 2 on the ONE GPU of the test box, through the product's own arena, GradBuckets and fused
Adam (asr/train_asr.py:67-71,236-243): two fresh child processes (tests/dp_worker.py), each a rank on cuda:0, joined
by a gloo group -- RCCL needs one GPU per rank, everything above the collective backend is the 8-GPU code path.

The children assert: the rank dropout seed is applied before the first forward; the bucketed asynchronous all-reduce leaves
the SUM of the ranks' own gradients in every arena; after three train_step updates the parameters are bit-identical on
both ranks; the same after two train_group updates (two stacked micro-batches per rank and step) of a d = 256 model.  This process makes no GPU call itself (it sorts first in the suite for that reason) and only waits."""
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


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu(tmp_path):
    out = str(tmp_path / "dp2.json")
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EMOASR_CPU_THREADS="2")
    # each child's output goes to a file of its own: two pipes drained one after the other would deadlock as soon as rank 1
    # filled its 64 KB pipe buffer (verbose hipcc lines on a cold build) while rank 0 sat in a collective waiting for it
    files = [open(str(tmp_path / f"rank{r}.log"), "w") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, "-m", "tests.dp_worker", str(r), "2", port, out], cwd=ROOT, env=env,
                              stdout=files[r], stderr=subprocess.STDOUT) for r in range(2)]
    try:
        for p in procs:
            p.wait(timeout=540)
    except subprocess.TimeoutExpired:
        for q in procs:
            q.kill()
        raise
    finally:
        for f in files:
            f.close()
    logs = [open(str(tmp_path / f"rank{r}.log")).read() for r in range(2)]
    if any("no GPU" in (log or "") and p.returncode != 0 for p, log in zip(procs, logs)):
        pytest.skip("no GPU")
    for r, (p, log) in enumerate(zip(procs, logs)):
        assert p.returncode == 0, f"rank {r} failed:\n{log[-3000:]}"
    res = json.load(open(out))
    assert res["ok"] and res["step"] == 3 and res["async_ranges"] >= 1 and res["stacked_step"] == 2, res
    print("two ranks on one GPU:", res)
