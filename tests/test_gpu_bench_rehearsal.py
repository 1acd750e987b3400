"""bench.py's multi-rank code paths, rehearsed on ONE GPU under pytest: the driver's 8-GPU run must not
be the first execution of the self-spawning launcher, the process group, the rank-sharded sections and the native RCCL
communicator set-up.

  gloo, 2 ranks   `VG_BENCH_BACKEND=gloo python bench.py --gpus 2 ...`: bench.py re-executes itself through
                  torch.distributed.run, two ranks share the one device (RCCL refuses that, gloo does not), every secondary
                  section runs with its collectives, and the sharded solves end at the one-rank optimum.
  RCCL, 1 rank    `VG_BENCH_FORCE_DIST=1` under torch.distributed.run with one process: init_process_group("nccl"),
                  vg_comm_unique_id -> broadcast -> vg_comm_create (ncclCommInitRank) in the helper thread, the gloo side-group
                  agreement, the all-reduces of the sections through the native communicator.
Sizes are cut down (the code paths, not the numbers, are under test)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--images", "500", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0.2", "--sharded-images", "600",
         "--sharded-solve-images", "1500"]
SECTIONS = ("roofline", "jtj", "sharded_mei", "sharded_solve", "solve", "pcie_inclusive", "config3_stereo", "config5_rig", "eucm_100k",
            "emit_sweep", "calib_e2e", "pose_init")


def run(cmd, env_extra):
    env = dict(os.environ)
    env.update({"HSA_ENABLE_IPC_MODE_LEGACY": "0", "VG_BENCH_CONFIG_IMAGES": "60", "VG_BENCH_SECONDARY_TIMEOUT": "600"})
    env.update(env_extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py must print ONE JSON line (rank 0): %r" % r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


def check_sections(line):
    for k in SECTIONS:
        assert k in line, "section %s missing" % k
        assert "error" not in line[k], (k, line[k])
    # round 5: the product entry point end to end, the pose-initialisation kernel, the host-memory route with its bus ceiling
    assert set(line["calib_e2e"]["phases"]) >= {"parse_json_s", "refine_total_s", "solve_s", "residual_format_s"} and line["calib_e2e"]["cli_wall_s"] > 0
    assert line["pose_init"]["kernel_ms"] > 0 and line["pose_init"]["converged"] > 0 and line["pose_init"]["roofline"]["frac"] > 0
    pc = line["pcie_inclusive"]
    assert pc["repetitions"] >= 20 and pc["d2h_ceiling_same_box"]["median_GBps"] > 0
    assert pc["pinned_destination"]["median_ms"] > 0 and pc["pageable_destination"]["median_ms"] > 0
    assert len(line["emit_sweep"]) >= 1 and all(r["frac"] > 0 for r in line["emit_sweep"])
    assert "secondary_sections" not in line, line.get("secondary_sections")
    for k in ("config3_stereo", "config5_rig"):
        for part in ("emit", "jtj"):
            rf = line[k][part]["roofline"]
            assert rf["frac"] > 0 and rf["avg_launch_ms"] > 0 and rf["kernel"].startswith("vg_")
        assert line[k]["solve"]["iterations"] >= 1
    assert line["eucm_100k"]["roofline"]["bound"] == "hbm"


@pytest.fixture(scope="module")
def one_rank():
    line, _ = run([sys.executable, "bench.py", "--gpus", "1"] + SMALL, {})
    assert line["n_gpus"] == 1
    check_sections(line)
    assert "cpu_baseline" in line and line["cpu_baseline"]["kind"] == "port"
    return line


@pytest.mark.parametrize("n_ranks", [2, 8])
def test_ranks_over_gloo_on_one_gpu_run_every_section_and_reach_the_one_rank_optimum(one_rank, n_ranks):
    """2 ranks, and the 8 of the driver's scaling run (the shard arithmetic, every collective and the deadline thread with the
    world size the real run has)"""
    line, err = run([sys.executable, "bench.py", "--gpus", str(n_ranks)] + SMALL, {"VG_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == n_ranks and line["scaling"] == "weak"
    check_sections(line)
    assert line["jtj"]["allreduce"] is True and "gloo" in line["jtj"]["collective"]
    assert line["sharded_mei"]["n_ranks"] == n_ranks and line["sharded_mei"]["images_this_rank"] == 600 // n_ranks
    # the multi-rank sections explain themselves: isolated all-reduce latency at the exact message sizes and
    # the compute / collective split of an iteration
    sm = line["sharded_mei"]
    assert sm["per_iteration"]["compute_ms"] > 0 and sm["per_iteration"]["message_doubles"] == 17 * 17
    assert sm["allreduce_us"]["289"]["median_us"] > 0
    assert line["solve"]["allreduce_us"] and set(line["solve"]["messages_doubles"]) == {"evaluation", "schur"}
    for key in ("mei_10k", "eucm_100k", "mei_100k"):
        two, one = line["sharded_solve"][key], one_rank["sharded_solve"][key]
        assert two["n_ranks"] == n_ranks and two["collectives_per_iteration"] == 2
        msg = two["messages_doubles"]
        assert two["allreduce_us"][str(msg["evaluation"])]["median_us"] > 0 and two["allreduce_us"][str(msg["schur"])]["median_us"] > 0
        assert two["per_iteration"]["collective_ms"] > 0 and two["per_iteration"]["compute_ms"] >= 0
        # the same problem split over two ranks ends at the same optimum (summation order differs: 1e-9 on the cost)
        assert abs(two["final_cost"] - one["final_cost"]) <= 1e-9 * abs(one["final_cost"]), (key, two["final_cost"], one["final_cost"])
        assert abs(two["max_rel_intrinsics_error_vs_generating"] - one["max_rel_intrinsics_error_vs_generating"]) <= 1e-6
    # whole-job value: both ranks' observations over the max of the ranks' times
    assert line["value"] > 0 and line["config"]["images_per_gpu"] == 500


def test_one_rank_through_rccl_takes_the_native_communicator_path(one_rank):
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "1"] + SMALL
    line, err = run(cmd, {"VG_BENCH_FORCE_DIST": "1"})
    assert line["n_gpus"] == 1
    check_sections(line)
    # the native communicator was created (ncclCommInitRank through vg_comm_create) and agreed on over the gloo side group
    assert "native RCCL communicator: world size 1" in err, err[-2000:]
    assert line["jtj"]["allreduce"] is True and "vg_comm_allreduce_sum" in line["jtj"]["collective"]
    assert line["sharded_mei"]["rccl_world_size"] == 1
    for key in ("jtj", "solve", "config3_stereo", "config5_rig", "eucm_100k", "calib_e2e", "pose_init"):
        assert line[key]["rccl_world_size"] == 1, key
    assert line["sharded_mei"]["allreduce_us"]["289"]["median_us"] > 0      # ncclAllReduce of the 17 x 17 block, isolated
    for key in ("mei_10k", "eucm_100k", "mei_100k"):
        assert line["sharded_solve"][key]["rccl_world_size"] == 1
        a, b = line["sharded_solve"][key], one_rank["sharded_solve"][key]
        assert abs(a["final_cost"] - b["final_cost"]) <= 1e-9 * abs(b["final_cost"])
