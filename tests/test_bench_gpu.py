"""bench.py's own control flow on the GPU box: the multi-rank branch (process group, barrier + MAX-over-ranks timing, the final all_gather) is
executed here before the driver's 8-GPU node does it -- two ranks that share GPU 0 over gloo (GCDM_BENCH_SINGLE_GPU_TEST=1: a test hook, never
used for reported numbers) -- and the N = 1 line under torchrun equals what a plain `python bench.py` prints (reference analogue:
src/mol_gen_sample.py:108-112 picks one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
QUICK = ["--steps", "10", "--warmup", "2", "--batch", "128", "--no-cpu-baseline", "--no-other-configs", "--no-extras", "--no-full-sample", "--no-fp32-timing"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, env_extra=None, timeout=280):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.update(env_extra or {})
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"bench.py must print ONE JSON line, got {len(lines)}:\n{p.stdout[-2000:]}"
    return json.loads(lines[0])


def _check_line(r, n_gpus, B):
    assert r["n_gpus"] == n_gpus and r["config"]["parallelism"] == f"shard{n_gpus}" and r["scaling"] == "weak"
    assert r["steps"] == 10 and r["warmup"] == 2 and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert r["config"]["outputs_finite"] is True and r["config"]["flags"] == 0
    assert r["config"]["molecules_per_gpu"] == B
    # value = the units ALL ranks processed / the slowest rank's time: n_gpus x B molecules per 1001 network evaluations of ms_per_step each
    want = n_gpus * B / (1001 * r["ms_per_step"] * 1e-3)
    assert abs(r["value"] / want - 1) < 1e-9
    assert 0.0 < r["roofline"]["frac"] < 1.0 and r["roofline"]["avg_launch_ms"] > 0 and r["roofline"]["node_kernel"]["avg_launch_ms"] > 0
    # the shipped kernel's own cycles per tile (end-of-tile stamp): the box-independent figure of the line
    assert 3.0e4 < r["roofline"]["tile_cycles"] < 1.0e5 and r["roofline"]["tiles_per_launch"] == -(-B * 19 * 19 // 64)
    # the clock the line ran at (round 6): sclk / socket power sampled while the windows ran, the roofline fraction at THAT clock, and the clock the
    # kernel's own cycle stamps imply -- what separates 3 % of kernel from 4 % of box
    rf = r["roofline"]
    for key in ("sclk_mhz", "power_w", "frac_at_measured_clock", "sclk_mhz_from_cycles", "clock"):
        assert key in rf, key
    assert rf["clock"]["source"] is not None, "no clock source answered on this box (amdsmi / hwmon / rocm-smi)"
    assert 400.0 < rf["sclk_mhz"] <= 2500.0 and 100.0 < rf["power_w"] < 2000.0
    assert 400.0 < rf["sclk_mhz_from_cycles"] <= 2600.0
    assert rf["frac"] <= rf["frac_at_measured_clock"] < 1.0          # the box never runs above the nominal 2.4 GHz the peak is quoted at
    # the fused layer launch (round 6) is reported beside the roofline object, which stays about the edge kernel by itself
    fz = rf["fused_layer"]
    assert fz["active"] is (B * 361 > 64 * 256) and fz["avg_launch_ms"] > 0 and 0.0 < fz["frac"] < 1.0
    assert set(fz["one_handle_ms_per_step"]) == {"two_launches_per_layer", "fused"} == set(fz["one_handle_sclk_mhz"])        # alternating windows, each form with its clock
    if n_gpus > 1:
        pr = r["per_rank"]
        assert len(pr["ms_per_step"]) == n_gpus == len(pr["sclk_mhz"]) == len(pr["power_w"])
        assert abs(max(pr["ms_per_step"]) / r["ms_per_step"] - 1) < 1e-3          # the line reports the slowest rank


def test_two_ranks_run_the_multi_rank_branch():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "bench.py", "--gpus", "2"] + QUICK
    r = _run(cmd, {"GCDM_BENCH_SINGLE_GPU_TEST": "1"})
    _check_line(r, 2, 128)
    assert r["config"]["final_gather_ms"] > 0.0           # the one collective of the path: all_gather of the final samples, outside the timed region
    assert "cpu_baseline" not in r                         # rank 0 at N = 1 only


def test_one_rank_under_torchrun_prints_the_n1_line():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "bench.py", "--gpus", "1"] + QUICK
    a = _run(cmd)
    b = _run([sys.executable, "bench.py"] + QUICK)
    for r in (a, b):
        _check_line(r, 1, 128)
        assert r["config"]["final_gather_ms"] == 0.0 or r["n_gpus"] == 1
    assert set(a) == set(b) and set(a["config"]) == set(b["config"]) and set(a["roofline"]) == set(b["roofline"])
    assert a["metric"] == b["metric"] and a["unit"] == b["unit"] and a["config"]["workload"] == b["config"]["workload"]
    assert abs(a["ms_per_step"] / b["ms_per_step"] - 1) < 0.25                 # same work, same box (clock noise only)


def test_plain_command_with_gpus_2_launches_its_own_ranks_and_measures_configs4():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment -- the shape of the driver's 1-GPU command -- must start its two ranks itself
    (torch.distributed.run on 127.0.0.1) and still print ONE JSON line; at N > 1 that line carries the BASELINE.json configs[4] leg (GEOM-Drugs,
    256 molecules per GPU, sharded) next to the QM9 weak-scaling headline."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", GCDM_BENCH_SINGLE_GPU_TEST="1")
    args = [a for a in QUICK if a != "--no-other-configs"]
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    _check_line(r, 2, 128)
    leg = r["other_configs"]["configs[4] geom x2"]
    assert leg["n_gpus"] == 2 and leg["parallelism"] == "shard2" and leg["molecules"] == 512 and leg["gathered_rows"] == 2 * 256 * 44
    assert leg["final_gather_ms"] > 0.0 and leg["outputs_finite"] is True and leg["flags"] == 0
    assert abs(leg["value"] / (512 / (1001 * leg["ms_per_step"] * 1e-3)) - 1) < 1e-9


def test_rccl_communicator_and_all_gather_with_world_size_one():
    """The "nccl" backend (= RCCL on ROCm) itself: communicator creation, barrier and the all_gather of the final samples, with world size 1 on
    this box's GPU -- so that the first RCCL call of this code base is not made on the driver's 8-GPU node (the two-rank tests above share
    one GPU and therefore have to use gloo)."""
    code = (
        "import os, torch, torch.distributed as dist\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "x = torch.arange(19 * 9, dtype=torch.float32, device='cuda').view(19, 9)\n"
        "bufs = [torch.empty_like(x)]\n"
        "dist.barrier(); dist.all_gather(bufs, x)\n"
        "t = torch.tensor([3.5], dtype=torch.float64, device='cuda'); dist.all_reduce(t, op=dist.ReduceOp.MAX)\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(bufs[0], x) and float(t.item()) == 3.5 and dist.get_backend() == 'nccl'\n"
        "dist.destroy_process_group(); print('RCCL_OK')\n")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stderr[-3000:]
