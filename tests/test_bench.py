"""bench.py keeps its contract (GPU): one JSON line with the agreed keys on one rank, and the N-rank code path
(cstone_hip_domain_mr_sync under torch.distributed.run) runs end to end -- rehearsed here with two gloo ranks that share
the one GPU of the test box (numbers meaningless, control flow identical to the RCCL run)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"]


args_key_passes = 8  # one standalone sort of 64-bit keys


def _json_line(out):
    # the bench's contract: ONE JSON line on stdout and nothing else (banners of libraries it loads go to stderr)
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_n_starts_n_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (a torch.distributed.run child of a
    parent that never touches the GPU): --launch-only lets each rank report its RANK / WORLD_SIZE and leave"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-only"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    import re

    lines = [json.loads(m.group(0)) for m in re.finditer(r"\{[^{}]*\}", r.stdout)]
    assert sorted(j["rank"] for j in lines) == [0, 1]
    assert all(j["world_size"] == 2 and j["launch_only"] and j["gpus_requested"] == 2 for j in lines)
    assert sorted(j["local_rank"] for j in lines) == [0, 1] and all(j["master"].startswith("127.0.0.1:") for j in lines)


def test_bench_failed_rank_fails_the_launcher():
    """a rank that dies makes the parent exit non-zero (an unknown --dist is refused by every rank's parser)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-only", "--path", "mr",
                        "--key-bits", "sixty-four"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0


@pytest.mark.gpu
def test_bench_single_rank_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--particles", "2e6", "--steps", "2", "--warmup", "1",
           "--cpu-sample", "2e5", "--neighbor-targets", "1e5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    for k in KEYS + ["cpu_baseline"]:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["vs_baseline"] is None
    assert j["value"] > 0 and abs(j["value"] - 2e6 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    pl = j["extras"]["plummer"]  # BASELINE configs[2] next to the headline number: the reference's cloud, drifting
    assert pl["value"] > 0 and pl["focus_leaves"] > 0 and pl["find_neighbors"]["mean_neighbors"] > 10
    assert pl["syncs"]["syncs"] == 2 and pl["zero_motion"]["value"] > 0 and "box_redos" in pl["syncs"]
    assert max(abs(v) for v in pl["box"]) > 20  # no clamping shell: R < 100 * 3 pi / 16
    mr = j["extras"]["mr_path_world_of_one"]  # the first point of the scaling curve through the N-GPU code path
    assert mr["rccl_ranks"] == 1 and mr["value"] > 0 and mr["invariants_ok"] is True
    assert j["config"]["path"] == "single" and j["config"]["dist"] == "uniform"
    assert j["extras"]["encode_sort_tree_1e7"]["leaves"] > 0  # BASELINE configs[1]
    assert j["extras"]["one_stream"]["value"] > 0  # the headline loop without the second stream
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal():
    env = dict(os.environ, CSTONE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29790", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles", "2e6",
           "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    for k in KEYS:
        assert k in j, k
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and "cpu_baseline" not in j
    assert 0.0 < j["roofline"]["frac"] < 1.0 and j["roofline"]["launches"] == args_key_passes
    assert j["config"]["particles_per_gpu"] == 1000000 and j["config"]["invariants_ok"] is True
    assert j["config"]["orchestration"].startswith("libcstone_hip")
    # both exchanges moved data: particles changed owner and halos were served
    ex = j["config"]["rank0_exchange"]
    assert ex["halos"] > 0 and j["config"]["rank0_assigned"] > 0


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher():
    """the driver's command line, `python bench.py --gpus 2 ...` (no torchrun): two ranks are started, ONE line comes back
    with n_gpus 2 (gloo collectives: the two ranks share the one GPU of the test box)"""
    env = dict(os.environ, CSTONE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles", "2e6", "--steps", "2",
           "--warmup", "1", "--dist", "clustered"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["self_launched"] is True and j["config"]["path"] == "mr"
    assert j["config"]["dist"] == "clustered" and j["config"]["invariants_ok"] is True
    assert j["config"]["rank0_exchange"]["halos"] > 0


def _num_gpus():
    try:
        import torch

        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_num_gpus() < 2, reason="needs two GPUs: the RCCL path with more than one rank")
def test_bench_two_ranks_rccl():
    """the same run over RCCL, one rank per GPU (skipped on the one-GPU test boxes)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CSTONE_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29791", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles", "4e6",
           "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["rank0_exchange"]["halos"] > 0 and 0.0 < j["roofline"]["frac"] < 1.0
    assert j["config"]["transport"].startswith("RCCL inside libcstone_hip") and j["config"]["invariants_ok"] is True


@pytest.mark.gpu
def test_bench_native_rccl_world_of_one():
    """the multi-rank domain over RCCL served from C++ inside the library (csrc/comm_rccl.hip), rehearsed with a world of
    ONE rank on the one-GPU box: the same entry points, callbacks and stream ordering as the N-GPU run (all-reduce,
    all-gather and the grouped send/recv with no peer)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CSTONE_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--particles", "2e6", "--steps", "2", "--warmup", "1",
           "--path", "mr", "--dist", "plummer"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j["config"]["transport"].startswith("RCCL inside libcstone_hip"), j["config"]["transport"]
    assert j["config"]["invariants_ok"] is True and j["config"]["rank0_assigned"] == 2000000
    assert j["config"]["orchestration"].startswith("libcstone_hip")
    assert j["config"]["rccl_ranks"] == 1 and j["config"]["path"] == "mr" and j["config"]["dist"] == "plummer"
