"""bench.py on the GPU box: the launcher of `--gpus N` (no torchrun around it) and the chain rates of the driver's line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_line(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line, nothing else on stdout
    return json.loads(lines[0])


def test_gpus_2_run_plainly_starts_two_ranks_and_counts_both():
    """`python bench.py --gpus 2` the way the driver runs `--gpus 1`: no WORLD_SIZE in the environment.  Both ranks are
    pinned to the one device of this box (AFX_BENCH_DEVICE): the line must say two ranks, name them, and count the
    frames of both -- not measure one GPU and call it n_gpus 1."""
    common = ("--buffers", "16", "--steps", "3", "--no-cpu-baseline", "--no-single")
    one = _bench_line("--gpus", "1", *common)
    two = _bench_line("--gpus", "2", *common, env={"AFX_BENCH_DEVICE": "0"})
    assert one["n_gpus"] == 1 and one["ranks_seen"] == [0]
    assert two["n_gpus"] == 2 and two["ranks_seen"] == [0, 1]
    assert [r["device"] for r in two["ranks"]] == [0, 0]
    per_rank = one["config"]["frames_per_gpu_per_step"]
    assert per_rank == 16 * 10000
    assert [r["frames"] for r in two["ranks"]] == [per_rank, per_rank]
    # value = frames of all ranks x steps / max-over-ranks time
    assert abs(two["value"] - 2 * per_rank * two["steps"] / (two["ms_per_step"] * 1e-3 * two["steps"])) <= 1e-6 * two["value"]
    assert all(r["ms_per_step"] <= two["ms_per_step"] * (1 + 1e-9) for r in two["ranks"])
    assert two["config"]["parity_spot_check"]["passed"] and two["config"]["parity_spot_check"]["worst_over_ceiling"] <= 1.0
    # the clock of the timed launches is sampled in the run itself (tools/clock_probe), per rank
    # (two ranks share the one device here: a probe that could not get a hardware queue of its own reports no clock --
    # the single-rank line below demands one)
    for clk in [two["roofline"]["clock_ghz_in_run"]] + [r["clock_ghz_in_run"] for r in two["ranks"]]:
        assert clk is None or 1.0 < clk < 2.6, clk
    # after the replicas ONE process drove both shards through the C++ file-sharding crawler (file i -> shard i mod 2,
    # Crawler.cpp:706-728): BASELINE configs[3]'s 12 500 files per GPU, every content on every shard with the same digest
    sc = two["config"]["sharded_crawl"]
    assert "error" not in sc, sc
    assert sc["devices"] == [0, 0] and sc["files"] == 25000 and sc["failed"] == 0
    assert sc["files_per_device"] == [12500, 12500]
    assert sc["row_digests"]["identical_per_content"] and sc["row_digests"]["files_per_device"] == [126, 126]
    from afec_amd import hostlib
    assert sc["workers_per_device"] == hostlib.workers_per_device_for(2) and sc["cpu_quota"] == hostlib.usable_host_cpus()
    assert sc["files_per_s"] > 0 and len(sc["upload_GB_per_s_per_device"]) == 2 and sc["busy_host_cpus"] > 0


def test_a_rank_without_a_device_of_its_own_fails_the_job():
    """one GPU on this box: rank 1 of a plain --gpus 2 has no device 1 and must say so; the job fails, no line"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices: rank 1 has its own")
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "AFX_BENCH_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--buffers", "4", "--steps", "2",
                        "--no-cpu-baseline", "--no-single"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0
    assert b"LOCAL_RANK=1 of 2 cannot use HIP device 1" in r.stderr
    assert not any(l.startswith(b"{") for l in r.stdout.splitlines())


@pytest.mark.parametrize("workload,n_files,mask_name,batch_files",
                         [("c3", 96, "frame", 0), ("c4", 160, "frame", 0), ("c3", 96, "all", 0), ("c4", 160, "frame", 48)])
def test_chain_rate_objects_of_the_line(workload, n_files, mask_name, batch_files):
    """config.c3_frames_per_s / c4_share_frames_per_s / c3_spectral_set_frames_per_s / c4_share_at_crawler_shape: a small
    batch of the same files through bench.chain_rate -- rate, roofline fraction on SURVEY's 5 080 B per frame, the clock
    sampled during the timed launches, and the three-file parity spot check inside bar and ceiling"""
    sys.path.insert(0, ROOT)
    import afec_amd as afx
    import bench
    plan = afx.Plan(max_analysis_ms=0, frame_kernel=afx.FRAME_KERNEL_WAVE64 if batch_files else afx.FRAME_KERNEL_AUTO)
    out = bench.chain_rate(plan, workload, n_files, 4321, mask_name=mask_name, batch_files=batch_files, in_flight=3)
    plan.close()
    assert out["files"] == n_files and out["frames"] > 30 * n_files and out["frames_per_s"] > 0
    assert abs(out["frac"] - out["frames_per_s"] * 5080 / 8e12) <= 1e-12
    assert out["clock_ghz_in_run"] is None or 1.0 < out["clock_ghz_in_run"] < 2.6
    if batch_files:
        assert out["batches"].startswith("4 batches of <= 48 files, 3 in flight") and out["frame_kernel"] == "wave64"
    spot = out["parity_spot_check"]
    assert spot.get("passed"), spot
    assert spot["files"] == sorted({0, n_files // 2, n_files - 1}) and spot["worst_over_ceiling"] <= 1.0
    if mask_name == "all":
        assert "f0" not in out["descriptors"].split("BASELINE")[0]


def test_the_default_line_one_gpu_sharded_crawl_agrees_with_the_end_to_end_driver():
    """The driver's own command (minus the CPU baseline): config.sharded_crawl at N = 1 is the same C++ driver over one
    device and must agree with config.end_to_end_host_driver; the roofline carries the clock of this very run and the
    VALU fraction at that clock and at the nominal one; the two round-6 chain objects are there with their spot checks."""
    line = _bench_line("--no-cpu-baseline")
    roof, cfg = line["roofline"], line["config"]
    assert 1.0 < roof["clock_ghz_in_run"] < 2.6 and roof["clock_probe"]["ended_by"] == "stop"
    if roof["valu"] is not None:      # counters are quoted for the build they were measured on only
        assert roof["valu"]["clock_ghz"] == roof["clock_ghz_in_run"]
        assert roof["valu"]["frac_at_nominal_clock"] < roof["valu"]["frac"] <= 1.0
    sc, e2e = cfg["sharded_crawl"], cfg["end_to_end_host_driver"]
    assert "error" not in sc and "error" not in e2e, (sc, e2e)
    assert sc["files_per_device"] == [12500] and sc["row_digests"]["identical_per_content"]
    assert sc["workers_per_device"] == e2e["workers"] == 5 or sc["cpu_quota"] < 5
    # (best of eight crawls each, minutes apart in one process: 0.4 % ... 6 % apart over the boxes of round 6)
    assert abs(sc["files_per_s"] / e2e["files_per_s"] - 1.0) < 0.15, (sc["files_per_s"], e2e["files_per_s"])
    for key in ("c3_frames_per_s", "c3_spectral_set_frames_per_s", "c4_share_frames_per_s", "c4_share_at_crawler_shape"):
        assert cfg[key]["parity_spot_check"]["passed"], (key, cfg[key]["parity_spot_check"])
    assert cfg["c4_share_at_crawler_shape"]["frame_kernel"] == "wave64"
    assert "mt19937" in cfg["workload"]
    # the driver's default run has to finish within minutes: every secondary object of the line included (the CPU
    # baseline, not run here, adds its bounded ~20 s; profiles/r06/bench_default.json)
    assert line["run_seconds"] < 150, line["run_seconds"]
