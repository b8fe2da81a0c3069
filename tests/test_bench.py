"""bench.py on the GPU box: the single-GPU line carries what the contract asks
for, and the N > 1 path (one frame in row bands, stitched on rank 0) runs its
rank logic end to end -- two ranks sharing the box's one device over gloo,
which stages the transfers through host memory -- and reproduces the golden
hash of the whole frame."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gpu = pytest.mark.gpu


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


@gpu
def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup",
                        "2", "--settle-ms", "50", "--no-cpu"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["parity"]["ok"]
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["sustained"]["seconds"] >= 1.0
    assert d["config"]["exec_path"].startswith("exact-2x")
    # one clock for ms_per_step and the roofline fraction
    assert abs(d["roofline"]["kernel_ms"] - d["ms_per_step"]) < 1e-3
    assert 0 < d["roofline"]["frac_sustained"] < 1
    # every other BASELINE.json config rides on the same line, golden-equal
    cf = {c["config"]: c for c in d["configs"]}
    assert sorted(cf) == ["cfg1", "cfg2", "cfg4", "cfg5"]
    for name, c in cf.items():
        assert c["parity_ok"] and c["fnv1a64"] == c["golden"], c
        assert 0 < c["roofline"]["frac"] < 1 and c["ms_per_step"] > 0, c
    assert d["configs_parity_ok"]


@gpu
@pytest.mark.parametrize("world", [2, 8])
def test_bench_two_ranks_band_pipeline_gloo_dry_run(world):
    """The driver's N > 1 command as far as one GPU can take it: `world` ranks
    (2, and the 8 of BASELINE.json configs[2]) share the box's GPU over gloo --
    band arithmetic, window-only source distribution, the stitch, the line."""
    port = 29700 + os.getpid() % 200 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", str(world),
           "--steps", "3", "--warmup", "1", "--settle-ms", "0", "--backend",
           "gloo", "--no-cpu"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["n_gpus"] == world and d["scaling"] == "strong"
    assert d["config"]["sharding"].startswith("row bands of ONE frame")
    assert d["parity"]["ok"], d["parity"]
    s = d["sharded"]
    assert "error" not in s, s
    assert s["gather_only"]["stitched_ok"]
    assert s["compute_only"]["value"] > 0 and s["independent_frames"]["value"] > 0
    assert d["value_distributed"] == s["compute_only"]["value"]
    assert d["value_independent_frames"] == s["independent_frames"]["value"]
    # rank 0 was given its band's source rows only, the rest NaN
    w = d["config"]["source_rows_on_this_rank"]
    assert w["first_row"] == 0 and w["of"] == 2160, w
    assert w["rows"] < 2160 // world + 120, w


@gpu
def test_bench_band_pipeline_failure_is_not_a_success_line():
    """If the stitch fails, `value` must not silently become something else:
    null value, an error field, exit code 3."""
    env = dict(os.environ, AVIRHIP_BENCH_FAIL_PIPELINE="1",
               MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29300 + os.getpid() % 150))
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup",
                        "1", "--settle-ms", "0", "--no-cpu", "--pg-smoke",
                        "--backend", "gloo"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    d = _last_json(r.stdout)
    assert d["value"] is None and "band pipeline failed" in d["error"]
    assert d["value_independent_frames"] > 0
    assert d["config"]["sharding"].startswith("FAILED")


@gpu
def test_bench_one_rank_process_group_on_rccl():
    """The N > 1 code path (process group on the nccl backend = RCCL, barriers,
    the max-over-ranks all-reduce, the band pipeline and the sharded extras)
    with the one rank this box can host."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29500 + os.getpid() % 150))
    r = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup",
                        "2", "--settle-ms", "50", "--no-cpu", "--pg-smoke",
                        "--backend", "nccl"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["parity"]["ok"], d
    assert "row bands of ONE frame" in d["config"]["sharding"] or \
        d["config"]["sharding"].startswith("one GPU"), d["config"]
    assert "error" not in d["sharded"], d["sharded"]
    assert d["sharded"]["gather_only"]["stitched_ok"]


@gpu
def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no torchrun around it (how the driver
    runs the N = 1 line): the script must start the two ranks itself and the
    line must say so -- ranks_seen counted by an all-reduce, one device record
    per rank."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps",
                        "3", "--warmup", "1", "--settle-ms", "0", "--backend",
                        "gloo", "--no-cpu"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2, d
    assert [x["rank"] for x in d["devices"]] == [0, 1], d["devices"]
    assert d["parity"]["ok"], d["parity"]


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    """WORLD_SIZE from the launcher and --gpus must agree; the check runs
    before anything touches a device, so it is a CPU test."""
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps",
                        "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE=4" in r.stderr, r.stderr[-500:]
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    # bare --gpus 1 with WORLD_SIZE=1 from a launcher is fine up to the device
    # check; --gpus 3 under WORLD_SIZE=1 is not
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "3", "--steps",
                        "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 3" in r.stderr


def test_bench_nccl_refuses_more_ranks_than_devices():
    """On the RCCL backend N ranks need N devices (this container has none;
    the GPU box has one): the run ends with a message, not with n_gpus: 1."""
    env = dict(os.environ, WORLD_SIZE="16", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "16", "--steps",
                        "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "needs 16 devices" in r.stderr, r.stderr[-500:]
