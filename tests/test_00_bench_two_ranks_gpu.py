"""bench.py --gpus 2 as the driver launches it (one process per rank, RANK / WORLD_SIZE / MASTER_* in the environment), rehearsed
on the ONE GPU of the test box: EMOASR_BENCH_ONE_GPU=1 puts both ranks on cuda:0 and EMOASR_DIST_BACKEND=gloo replaces RCCL (which
needs one GPU per rank); everything above the collective backend -- per-rank batches and dropout seeds, the stacked pass, the
bucketed asynchronous all-reduce released layer by layer, 1 / world in the fused Adam, the max-over-ranks timing, the JSON line
-- is the code the 8-GPU run executes (asr/train_asr.py:67-71,236-243 replaced by data parallelism).

The line must validate itself: `dp.nranks` = 2, both ranks' frame counts, and bit-identical parameters on both ranks after the
steps.  This process makes no GPU call (it sorts first in the suite for that reason) and never re-executes itself."""
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


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu(tmp_path):
    port = str(_free_port())
    procs, files = [], []
    for r in range(2):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EMOASR_CPU_THREADS="2", EMOASR_BENCH_ONE_GPU="1",
                   EMOASR_DIST_BACKEND="gloo", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port)
        out = open(str(tmp_path / f"rank{r}.out"), "w")
        err = open(str(tmp_path / f"rank{r}.err"), "w")
        files += [out, err]
        procs.append(subprocess.Popen([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "2", "--no-decode",
                                       "--no-cpu-baseline"], cwd=ROOT, env=env, stdout=out, stderr=err))
    try:
        for p in procs:
            p.wait(timeout=840)
    except subprocess.TimeoutExpired:
        for q in procs:
            q.kill()
        raise
    finally:
        for f in files:
            f.close()
    errs = [open(str(tmp_path / f"rank{r}.err")).read() for r in range(2)]
    if any(p.returncode != 0 and ("No HIP GPUs" in e or "no GPU" in e) for p, e in zip(procs, errs)):
        pytest.skip("no GPU")
    for r, (p, e) in enumerate(zip(procs, errs)):
        assert p.returncode == 0, f"rank {r} failed:\n{e[-3000:]}"
    lines = [ln for ln in open(str(tmp_path / "rank0.out")).read().splitlines() if ln.startswith("{")]
    # rank 0 prints ONE JSON line, rank 1 none (gloo itself may print a connection notice)
    assert len(lines) == 1 and not [ln for ln in open(str(tmp_path / "rank1.out")).read().splitlines() if ln.startswith("{")]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["scaling"] == "weak" and res["value"] > 0
    dp = res["dp"]
    assert dp["nranks"] == 2 and dp["backend"] == "gloo" and dp["one_gpu_rehearsal"]
    assert len(dp["frames_per_rank"]) == 2 and all(v > 0 for v in dp["frames_per_rank"])
    assert dp["frames_per_rank"][0] != dp["frames_per_rank"][1]          # every rank packs its own batches
    assert abs(sum(dp["frames_per_rank"]) - res["value"] * res["ms_per_step"] * 2 / 1000.0) < 1e-3 * sum(dp["frames_per_rank"])
    assert dp["params_identical_across_ranks"], dp
    assert dp["overlapped_allreduce"]
    print("bench.py --gpus 2 on one GPU:", {k: dp[k] for k in ("nranks", "frames_per_rank", "params_identical_across_ranks")},
          "value", res["value"])
