"""bench.py --gpus N must start its own ranks: run as plain `python bench.py --gpus 2 ...`, i.e.
without a launcher environment, it spawns torch.distributed.run on itself before any GPU call, the ranks rendezvous on
127.0.0.1, and the parent relays exactly one JSON line (last on stdout) plus the exit code.  `--dry-run` swaps RCCL for
gloo and the kernels for a row-id exchange so the whole control path runs here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("gpus,transport", [(2, "allgather"), (2, "needed"), (4, "needed"), (8, "allgather"), (8, "needed")])
def test_self_launch_gloo_dry_run(gpus, transport):
    """2, 4 and 8 real ranks at the BENCH geometry (128 frames per GPU: two half-batches, 320 items per rank and half; at 8 ranks
    5 x 512 items per half)."""
    p = _run("--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--dry-run", "--transport", transport)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    rec = json.loads(lines[-1])                      # the record is the LAST stdout line
    assert sum(ln.startswith('{"metric"') for ln in lines) == 1
    assert rec["n_gpus"] == gpus and rec["ranks_seen"] == gpus and rec["dry_run"] and rec["exchange_ok"]
    assert rec["items_per_rank"] == 5 * 64 and rec["transport"] == transport and rec["frames_per_step"] == 128 * gpus
    assert rec["layout"] == "spread" and rec["active_ranks"] == gpus
    # the strong-scaling geometry (128 frames in total: 320 items per half-batch over the ranks) and the rows of the on-hardware
    # sharded == unsharded self-check (8 frames x the agents a rank owns of them), weak and strong
    assert rec["strong_items_per_rank"] == 320 // gpus
    assert len(rec["shard_check_pairs"]) == 2 and all(8 <= n <= 40 for n in rec["shard_check_pairs"])


@pytest.mark.parametrize("gpus,transport", [(5, "allgather"), (8, "allgather"), (8, "needed")])
def test_agent_per_gpu_layout_dry_run(gpus, transport):
    """SURVEY 8(e) 'literal one-agent-per-GPU layout -- report both': --layout agent-per-gpu runs the five agents on ranks 0..4 (their
    own sub-group at N = 8; ranks 5..7 idle through the barriers), 128 frames per active GPU."""
    p = _run("--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--dry-run", "--transport", transport, "--layout", "agent-per-gpu")
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert rec["n_gpus"] == gpus and rec["ranks_seen"] == gpus and rec["exchange_ok"]
    assert rec["layout"] == "agent-per-gpu" and rec["active_ranks"] == 5 and rec["frames_per_step"] == 128 * 5
    assert rec["items_per_rank"] == 320                 # rank a owns agent a: 320 frames of it per half-batch


def test_agent_per_gpu_needs_five_ranks():
    p = _run("--gpus", "2", "--dry-run", "--layout", "agent-per-gpu")
    assert p.returncode != 0 and "needs --gpus >= 5" in p.stderr


def test_n1_dry_run_needs_no_launcher():
    p = _run("--gpus", "1", "--dry-run", "--steps", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.splitlines()[-1])["ranks_seen"] == 1


def test_ranks_fail_loudly_without_a_gpu():
    """Without --dry-run on a box without (enough) GPUs nothing runs: --gpus N > device_count() is refused by the parent with the two numbers, a single
    rank stops at the 'needs the MI355X' check; a non-zero exit code and no record either way (never a CPU fallback number)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: the real run is the driver's")
    p = _run("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert p.returncode != 0
    assert "--gpus 2 requested, this node exposes 0 GPUs" in p.stderr          # (round 5) the parent answers before it starts any rank
    assert not any(ln.startswith('{"metric"') for ln in p.stdout.splitlines())
    p = _run("--gpus", "1", "--steps", "2", "--warmup", "1")
    assert p.returncode != 0 and "needs the MI355X" in p.stderr and not any(ln.startswith('{"metric"') for ln in p.stdout.splitlines())


def test_world_size_mismatch_is_refused():
    p = _run("--gpus", "2", "--dry-run", env_extra={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=4 but --gpus 2" in p.stderr


def test_shard_check_plan_covers_owned_rows_at_every_world():
    """bench.py::shard_check_plan (host logic of `sharded_equals_unsharded`): at every world size and both geometries the frames start at the
    rank's first item, every pair points at the right global row, and nothing the rank does not own is compared."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "v2x-sim_amd"))
    import bench
    from v2x_sim_amd.parallel import AgentShard
    for world in (1, 2, 4, 5, 8):
        for bh in (64 * world, 64):
            if (5 * bh) % world:
                continue
            covered = 0
            for r in range(world):
                rows = AgentShard(5, bh, r, world).rows
                frames, pairs = bench.shard_check_plan(rows, bh)
                assert len(frames) == min(8, bh) and frames[0] == rows[0] % bh and len(set(frames)) == len(frames)
                assert pairs and len(set(p[0] for p in pairs)) == len(pairs)
                for ri, gi in pairs:
                    a, k = divmod(ri, len(frames))
                    assert rows[gi] == a * bh + frames[k]
                covered += len(pairs)
            assert covered >= 8 * world or world == 1


# ---- the watchdog around the ranks of an N > 1 run (the first N > 1 run on hardware is the driver's) ---------------------
def _record(p):
    return json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])


def test_ranks_of_a_multi_gpu_run_are_supervised():
    """Every rank process the launcher starts is a supervisor that runs the real rank as a child (bench.py: supervise); the record says so.
    N = 1 stays a plain process."""
    p = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    assert _record(p)["launch"] == {"supervised": True, "attempt": 1, "graph_mode": 1, "watchdog_s": 240.0}
    p = _run("--gpus", "1", "--steps", "2", "--dry-run")
    assert p.returncode == 0 and _record(p)["launch"]["supervised"] is False


def test_watchdog_relaunches_a_stalled_first_attempt_in_the_one_stream_order():
    """A collective that never completes (rank 1 never issues its first exchange: V2X_BENCH_SIMULATE_HANG) stalls every rank.  Rank 0's supervisor
    sees no first step within the watchdog time of `ranks-ready`, raises the node-local flag, every supervisor kills its child and starts a
    FRESH child process with --graph 4 on a new rendezvous; the record comes from attempt 2 and no process is left behind."""
    p = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", env_extra={"V2X_BENCH_SIMULATE_HANG": "1", "V2X_BENCH_WATCHDOG_S": "4"})
    assert p.returncode == 0, p.stderr[-2000:]
    rec = _record(p)
    assert rec["launch"]["attempt"] == 2 and rec["launch"]["graph_mode"] == 4 and rec["exchange_ok"] and rec["ranks_seen"] == 2
    assert p.stderr.count("attempt 1 stalled") == 2 and "relaunching with --graph 4" in p.stderr
    assert sum(ln.startswith('{"metric"') for ln in p.stdout.splitlines()) == 1
    left = subprocess.run(["pgrep", "-f", "bench.py --gpus 2 --steps 2 --warmup 1 --dry-run"], capture_output=True, text=True).stdout.split()
    assert not left, left


@pytest.mark.parametrize("transport", ["allgather", "needed"])
def test_a_late_rank_delays_but_cannot_deadlock_the_two_stream_host_order(transport):
    """bench.py's default step issues exchange A and exchange B asynchronously before waiting for either, in the same host order on every rank.
    With one rank 3 s late for its first exchange the step completes (first attempt, no relaunch) with the right rows -- for the all-gather and
    for the grouped point-to-point transport."""
    p = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--transport", transport,
             env_extra={"V2X_BENCH_DELAY_RANK": "1", "V2X_BENCH_DELAY_S": "3", "V2X_BENCH_WATCHDOG_S": "60"})
    assert p.returncode == 0, p.stderr[-2000:]
    rec = _record(p)
    assert rec["exchange_ok"] and rec["launch"]["attempt"] == 1 and rec["elapsed_s"] >= 2.5


def test_live_traffic_measurement_is_optional_and_fails_soft(monkeypatch, tmp_path):
    """bench.py re-measures the dominant kernel's HBM traffic with two `rocprofv3 --pmc` child passes from a process that never touches the GPU (round 6:
    the N = 1 orchestrator, AFTER the timed run's child); switched off, under a profiler, without rocprofv3 or when a pass fails it says why and the
    committed profiles/ value is quoted -- it never raises."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = bench.parse(["--no-live-traffic"])
    assert bench.live_traffic_table(args) == (None, "switched off", 0.0) and bench.live_traffic_wanted(args) == (False, "switched off")
    args = bench.parse([])
    monkeypatch.setenv("V2X_BENCH_LIVE_TRAFFIC", "0")
    assert bench.live_traffic_table(args)[0] is None
    monkeypatch.delenv("V2X_BENCH_LIVE_TRAFFIC")
    monkeypatch.setenv("ROCPROFILER_FAKE", "1")
    assert bench.live_traffic_table(args) == (None, "this run is itself under a profiler", 0.0)
    monkeypatch.delenv("ROCPROFILER_FAKE")
    # a "rocprofv3" that fails at once: the pass is reported as failed, nothing is raised, no temporary directory is left behind
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/sh\nexit 7\n")
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    before = set(os.listdir("/tmp"))
    table, note, seconds = bench.live_traffic_table(args)
    assert table is None and "FETCH_SIZE pass failed (exit code 7)" in note and seconds >= 0.0
    assert not [d for d in set(os.listdir("/tmp")) - before if d.startswith("v2x_pmc_")]
    # ... and one that writes the csv rocprofv3 writes: parsed per kernel, KiB -> bytes, FETCH_SIZE doubled
    fake.write_text("""#!/bin/sh
out=""; c=""
while [ $# -gt 0 ]; do
    case "$1" in --pmc) c=$2; shift;; -d) out=$2; shift;; --) break;; esac
    shift
done
if [ "$c" = FETCH_SIZE ]; then v=1000; else v=300; fi
mkdir -p "$out/host"; f="$out/host/p_counter_collection.csv"
echo '"Kernel_Name","Counter_Name","Counter_Value"' > "$f"
echo "\\"void conv3x3_stream8g_kernel<96, 2, false>(StreamArgs)\\",\\"$c\\",$v" >> "$f"
echo "\\"void conv3x3_stream8g_kernel<96, 2, false>(StreamArgs)\\",\\"$c\\",$v" >> "$f"
mkdir -p "$out/helper"; echo '"Kernel_Name","Counter_Name","Counter_Value"' > "$out/helper/q_counter_collection.csv"
""")
    table, note, seconds = bench.live_traffic_table(args)     # (the LARGEST csv is the bench child's: the header-only one of a helper process is ignored)
    assert note == "measured in this run" and table == {"conv3x3_stream8g_kernel<96, 2, false>": (1024.0 * 1000, 1024.0 * 300, 2)}     # RAW FETCH_SIZE bytes
    # the merge into a roofline record: measured value in (FETCH_SIZE x the kernel's CALIBRATED factor, round 6), committed value kept beside it, the passes' duration recorded
    roof = bench.apply_live_traffic({"kernel": "conv3x3_stream8g_kernel<96, 2, false>", "traffic": 7.0, "traffic_live": "x"}, (table, note, 12.34), "AFTER the timed run")
    ff = bench.FETCH_FACTOR["conv3x3_stream8g_kernel<96, 2, false>"]
    assert 1.45 < ff < 1.6 and roof["fetch_factor"] == ff and roof["fetch_factor_calibrated"] is True
    assert roof["traffic"] == ff * 1024.0 * 1000 + 1024.0 * 300 and roof["traffic_uniform_x2"] == 2.0 * 1024.0 * 1000 + 1024.0 * 300
    assert roof["traffic_committed"] == 7.0 and roof["pmc_passes_s"] == 12.3
    other = bench.apply_live_traffic({"kernel": "k2", "traffic": 1.0}, ({"k2": (100.0, 10.0, 1)}, note, 1.0), "after")
    assert other["traffic"] == 2.0 * 100.0 + 10.0 and other["fetch_factor_calibrated"] is False        # no entry: the guide's x 2
    assert "AFTER the timed run" in roof["traffic_source"] and "traffic_live" not in roof
    roof = bench.apply_live_traffic({"kernel": "other", "traffic": 7.0}, (table, note, 1.0), "after")
    assert roof["traffic"] == 7.0 and "kernel not in the counter passes" in roof["traffic_live"]
    # kernel names: the ARGUMENT list is cut, not the first parenthesis
    assert bench.strip_kernel_args("void (anonymous namespace)::k<1, 2>(Args, (anonymous namespace)::T)") == "(anonymous namespace)::k<1, 2>"
    assert bench.strip_kernel_args("conv3x3_tail_kernel(TailArgs) [clone .kd]") == "conv3x3_tail_kernel(TailArgs) [clone .kd]"


def test_n1_orchestrator_runs_the_bench_child_first_and_merges_the_counter_passes(tmp_path):
    """Round 6: at N = 1 the process the driver starts stays off the GPU; it runs the benchmark as a child FIRST and the two
    rocprofv3 --pmc passes AFTER it, then merges.  Here with a stand-in child (V2X_BENCH_INNER=1 makes bench.py the child: on a CPU box it exits 3
    "needs the MI355X") -- the orchestrator must relay that exit code and print no record."""
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/sh\necho pmc-pass-ran >> %s\nexit 7\n" % (tmp_path / "calls"))
    fake.chmod(0o755)
    env = {"V2X_BENCH_LIVE_TRAFFIC": "1", "V2X_BENCH_FORCE_ORCHESTRATE": "1", "PATH": str(tmp_path) + os.pathsep + os.environ["PATH"]}
    p = _run("--steps", "1", "--warmup", "0", env_extra=env)
    assert p.returncode == 3 and "needs the MI355X" in p.stderr and '{"metric"' not in p.stdout
    assert not (tmp_path / "calls").exists()        # no record from the child -> no counter pass is started


def test_summary_and_training_fractions():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    tr = {"maps_per_step": 10, "FaFNet": {"fp32 PyTorch-ROCm graph (MIOpen) ms": 48.5, "bf16 NHWC graph on the HIP kernels ms": 5.8, "the same as one replayed hipGraph ms": 5.17},
          "maps_40": {"FaFNet bf16 NHWC graph on the HIP kernels ms": 12.8}}
    fr = bench.training_fractions(tr)
    assert abs(fr["FaFNet_10_maps"] - 3 * 31.16 * 10 / 5.17 / 2500.0) < 1e-6 and abs(fr["FaFNet_40_maps"] - 3 * 31.16 * 40 / 12.8 / 2500.0) < 1e-6
    tr["frac_of_mfma_peak"] = fr
    sm = bench.build_summary(6400.0, 20.0, {"b1_ms": 0.5, "b1_frames_per_s": 2000.0, "b8_ms": 1.5, "b8_frames_per_s": 5300.0, "b32_ms": 5.0, "b32_frames_per_s": 6400.0},
                             {"configs": {"0n lowerbound network": {"frames_per_s": 7000.0}, "4 V2VNet segmentation": {"frames_per_s": 6900.0}}}, tr,
                             {"fp32_frames_per_s": 100.0, "bf16_autocast_channels_last_frames_per_s": 200.0, "value": 200.0}, {"value": 2.0}, {"frac": 0.58}, 165.2, 1)
    assert sm["latency_b8_frames_per_s"] == 5300.0 and sm["config4_v2vnet_seg_frames_per_s"] == 6900.0 and sm["speedup_vs_gpu_stock"] == 32.0
    assert sm["speedup_vs_cpu_baseline"] == 3200.0 and all(not isinstance(v, (dict, list)) for v in sm.values())
    assert abs(sm["whole_step_frac"] - 165.2e9 * 6400.0 / 2.5e15) < 1e-9


def test_n1_orchestrator_merges_the_counter_passes_into_the_childs_record(tmp_path):
    """The whole N = 1 path of round 6 on a CPU box: a canned child record (test hook V2X_BENCH_FAKE_RECORD) + a stand-in rocprofv3 that writes the csv the real one
    writes -> ONE JSON line whose roofline.traffic is the counters' value with the kernel's CALIBRATED fetch factor, the uniform-x2 value and the committed value beside
    it, `pmc_passes_s` recorded, and `summary` still the LAST key of the line (the driver keeps the tail of stdout)."""
    import json
    kern = "conv3x3_stream8g_kernel<96, 2, false>"
    child = {"metric": "BEV frames/sec, V2VNet 5-agent detection (256x256 BEV)", "value": 6500.0, "unit": "frames/s", "n_gpus": 1, "steps": 20, "warmup": 5,
             "ms_per_step": 19.7, "roofline": {"bound": "mfma", "achieved": 1400.0, "peak": 2500.0, "frac": 0.56, "traffic": 7.0, "kernel": kern,
                                               "traffic_live": "not re-measured: deferred to the parent process (after the timed run)"},
             "cpu_baseline": {"value": 2.0}, "summary": {"headline_frames_per_s": 6500.0}}
    rec_path = tmp_path / "child.json"
    rec_path.write_text(json.dumps(child))
    fake = tmp_path / "rocprofv3"
    fake.write_text("""#!/bin/sh
out=""; c=""
while [ $# -gt 0 ]; do
    case "$1" in --pmc) c=$2; shift;; -d) out=$2; shift;; --) break;; esac
    shift
done
if [ "$c" = FETCH_SIZE ]; then v=600000; else v=160000; fi
mkdir -p "$out/host"; f="$out/host/p_counter_collection.csv"
echo '"Kernel_Name","Counter_Name","Counter_Value"' > "$f"
echo "\\"void %s(StreamArgs)\\",\\"$c\\",$v" >> "$f"
""" % kern)
    fake.chmod(0o755)
    env = {"V2X_BENCH_FORCE_ORCHESTRATE": "1", "V2X_BENCH_FAKE_RECORD": str(rec_path), "V2X_BENCH_LIVE_TRAFFIC": "1", "PATH": str(tmp_path) + os.pathsep + os.environ["PATH"]}
    p = _run("--steps", "20", "--warmup", "5", env_extra=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    roof = rec["roofline"]
    raw, wr = 1024.0 * 600000, 1024.0 * 160000
    assert abs(roof["traffic"] - (roof["fetch_factor"] * raw + wr)) < 1.0 and 1.45 < roof["fetch_factor"] < 1.6 and roof["fetch_factor_calibrated"] is True
    assert abs(roof["traffic_uniform_x2"] - (2.0 * raw + wr)) < 1.0 and roof["traffic_committed"] == 7.0 and "traffic_live" not in roof
    assert "AFTER the timed run" in roof["traffic_source"] and roof["pmc_passes_s"] >= 0.0
    assert list(rec)[-1] == "summary" and lines[0].rstrip().endswith('"summary": {"headline_frames_per_s": 6500.0}}')
    assert rec["value"] == 6500.0 and rec["cpu_baseline"] == {"value": 2.0}
