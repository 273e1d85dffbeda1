"""bench.py is what the driver launches under torch.distributed.run on 1/2/4/8 GPUs. These tests execute the script's own
step / reduce / resolve / timing / JSON code: on CPU over gloo with the oracle injected as the renderer (world 1, 2),
and on the GPU box as the driver would call it (marked gpu)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "bench_oracle_backend.py")


def _run(cmd, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def _check_contract(j, world, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "counters"):
        assert k in j, k
    assert j["n_gpus"] == world and j["steps"] == steps and j["warmup"] == warmup
    assert j["unit"] == "Mrays/s" and j["higher_is_better"] is True and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and j["ms_per_step"] > 0
    c = j["counters"]
    assert c["rays"] == c["rays_closest"] + c["rays_shadow_walked"] + c["rays_shadow_shortcut"]


@pytest.mark.parametrize("split", ["replicate", "strata"])
def test_bench_script_world2_gloo_oracle_backend(split):
    port = 29611 if split == "replicate" else 29612
    j = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), WORKER, "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cornell_tiny",
              "--split", split, "--no-kernel-timing", "--no-cpu-baseline"])
    _check_contract(j, 2, 2, 1)
    paths_per_rank = 96 * 54 * 16 // (2 if split == "strata" else 1)
    assert j["scaling"] == ("strong" if split == "strata" else "weak")
    # whole-job aggregate: both ranks' paths over the max-over-ranks time
    assert abs(j["mpaths_per_s"] * j["ms_per_step"] * 1e3 - 2 * paths_per_rank) < 1e-3 * 2 * paths_per_rank
    assert "reduce_ms_per_step" in j
    if split == "replicate":
        assert j["frame_properties_ok"] is True  # rank 0's own frame, un-reduced


def test_bench_script_world1_oracle_backend_checks_its_frame():
    j = _run([sys.executable, WORKER, "--steps", "1", "--warmup", "0", "--workload", "cornell_tiny", "--configs", "none",
              "--no-kernel-timing"])
    _check_contract(j, 1, 1, 0)
    assert j["frame_properties_ok"] is True
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["gpu_band_bit_exact"] is True
    assert j["scaling"] == "weak"


def test_bench_script_refuses_a_world_size_mismatch():
    p = subprocess.run([sys.executable, WORKER, "--gpus", "2"], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr


@pytest.mark.gpu
def test_bench_script_on_the_gpu_verifies_the_frame_it_times():
    """The driver's command with fewer steps: the headline config (cfg3: the atrium at 1080p, 256 spp) plus cfg2 and cfg1. The
    timed frames must equal the committed oracle digests (cfg3: a 32-row band at its own sample count; cfg2: the whole frame
    with its counters), the oracle band timed beside them must equal the GPU frame, and no roofline fraction may pass 1."""
    j = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--configs", "cornell1080,cube512"], timeout=900)
    _check_contract(j, 1, 1, 0)
    assert "cfg3" in j["config"]["workload"] and j["config"]["name"] == "atrium1080" and j["config"]["sample_count"] == 256
    assert j["frame_verified"] is True and j["frame_properties_ok"] is True and j["frame_verified_rows"][1] - j["frame_verified_rows"][0] >= 32
    cb = j["cpu_baseline"]
    assert cb["gpu_band_bit_exact"] is True and cb["kind"] == "port" and cb["lavapipe"].startswith("unavailable (")
    assert cb["cores"] == cb["threads"] <= cb["host"]["affinity_cores"] and cb["single_thread"]["value"] > 0
    r = j["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] <= 1 and r["unit"] == "GB/s"
    cornell, cube = j["configs"]
    assert cornell["name"] == "cornell1080" and cornell["frame_verified"] is True and cornell["cpu_baseline"]["gpu_band_bit_exact"] is True
    assert cornell["roofline"]["bound"] == "valu" and 0 < cornell["roofline"]["frac"] <= 1 and cornell["roofline"]["served_from"] == "lds"
    assert cube["name"] == "cube512" and cube["frame_properties_ok"] is True and cube["cpu_baseline"]["gpu_band_bit_exact"] is True
    assert 0 < cube["roofline"]["frac"] <= 1


@pytest.mark.gpu
def test_bench_script_rccl_leg_on_a_one_rank_group():
    """A one-GPU box cannot launch N > 1, so the script's RCCL calls (process group on the device, the accumulator reduce on
    the render stream, barrier, the MAX / SUM all-reduces of the timing) run here on a one-rank `nccl` group; the reduced
    frame of one rank is that rank's frame, so the committed cfg2 digest still has to match."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               PT_BENCH_FORCE_COLLECTIVES="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cornell1080", "--steps", "2", "--warmup", "1",
                        "--configs", "none", "--no-cpu-baseline", "--no-fast-arith"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    (line,) = [l for l in p.stdout.splitlines() if l.startswith("{")]
    j = json.loads(line)
    _check_contract(j, 1, 2, 1)
    assert j["reduce_ms_per_step"] > 0  # the reduce ran on the device and was timed by events on the render stream
    assert j["frame_verified"] is True and j["frame_properties_ok"] is True
