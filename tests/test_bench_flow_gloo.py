"""bench.py's multi-rank control flow on 2 CPU ranks (tests/bench_dryrun.py: emulator kernels, gloo, reduced-depth net): the
sequence of collectives -- broadcast, per-step gradient all-reduce, barriers, MAX-reduce of the time, the loss reduce, and
the rank-0-ONLY instrumented roofline step that must not enter a collective -- completes, and rank 0 prints ONE JSON line
with the contract's keys.  (What the driver launches at round end with --gpus 2/4/8 over RCCL.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_world2_control_flow(emu_lib):
    env = dict(os.environ, HIPEMU_THREADS="4", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "tests", "bench_dryrun.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch", "1", "--size", "32", "--dtype", "bf16", "--extras", "3dpart"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                   # rank 0 only
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in rec, k
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["global_batch_slices"] == 2 and rec["config"]["parallelism"] == "dp2"
    assert "cpu_baseline" not in rec                                # rank 0 at N=1 only
    assert "DRY RUN" in rec["data"]
    assert rec["roofline"]["launches_per_step"] > 0
    # the 3D half of the metric rides in the same line: timed by the same function, data parallel as well
    ex = rec["config"]["extra_workloads"]
    assert len(ex) == 1 and ex[0]["workload"].startswith("denseunet_3d") and ex[0]["value"] > 0
    assert ex[0]["global_batch_slices"] == 16 and ex[0]["roofline"]["launches_per_step"] > 0


def test_bench_world2_shard3d_control_flow(emu_lib):
    """--config shard3d with 2 ranks: the depth-sharded step always contains collectives (halo exchange, sync-BN), so
    the rank-0-only instrumented roofline step must be skipped -- otherwise rank 0 blocks on sends that the other
    rank, already in the final barrier, never matches."""
    env = dict(os.environ, HIPEMU_THREADS="4", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29549", os.path.join(ROOT, "tests", "bench_dryrun.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "1", "--config", "shard3d", "--size", "32", "--cols", "16", "--dtype", "f32"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["config"]["global_batch_slices"] == 16 and "roofline" not in rec


def _run_guarded(port, timeout_s):
    env = dict(os.environ, HIPEMU_THREADS="4", OMP_NUM_THREADS="2", HDU_BENCH_SHARD3D="force",
               HDU_BENCH_SHARD3D_TIMEOUT=timeout_s)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(ROOT, "tests", "bench_dryrun.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "1", "--batch", "1", "--size", "32", "--dtype", "f32", "--extras", "none", "--no-roofline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, lines


def test_bench_world2_guarded_shard3d_extra(emu_lib):
    """under N > 1 the default bench also times ONE volume depth-sharded over the ranks (BASELINE configs[4], strong scaling)
    behind a watchdog: the record rides in the data-parallel line ..."""
    out, lines = _run_guarded("29551", "600")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    ex = rec["config"]["extra_workloads"]
    assert rec["config"]["parallelism"] == "dp2" and len(ex) == 1 and ex[0]["parallelism"] == "depth-shard2"
    assert ex[0]["value"] > 0 and ex[0]["global_batch_slices"] == 16 and "error" not in ex[0]
    assert rec["config"]["collectives"]["ranks"] == 2
    # what one sharded step exchanges is counted and reported (round 6): sync-BN / gradient all-reduces and neighbour (halo) exchanges
    cps = ex[0]["collectives_per_step"]
    assert cps["allreduce"] > 10 and cps["neighbour_exchange"] > 4 and cps["neighbour_mb"] > 0 and cps["hidden"] == 0


def test_bench_world2_guarded_shard3d_watchdog(emu_lib):
    """... and a depth-sharded step that does not complete in time (here: a watchdog of 10 ms) costs that record only: rank 0
    still prints the ONE line with the data-parallel result and every process exits cleanly"""
    out, lines = _run_guarded("29553", "0.01")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["value"] > 0 and rec["n_gpus"] == 2
    ex = rec["config"]["extra_workloads"]
    assert len(ex) == 1 and "watchdog" in ex[0]["error"]


def test_bench_dryrun_float32_split_contraction_extra(emu_lib):
    """the `2d:f32x3` extra workload (float32 storage, split-bf16 contraction switched on for that workload only) through
    bench.py's own control flow on CPU: both float32 lines are reported, labelled, and the mode is switched back afterwards"""
    env = dict(os.environ, HIPEMU_THREADS="4", OMP_NUM_THREADS="2")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_dryrun.py"), "--steps", "1", "--warmup", "1", "--batch", "1", "--size", "32",
           "--extras", "2d:f32x3,2d:f32"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    ex = rec["config"]["extra_workloads"]
    assert [e["dtype"] for e in ex] == ["f32x3", "f32"] and all(e["value"] > 0 for e in ex)
