"""N>1 path on CPU: world_size-2 `gloo` run of the data-parallel step (one process per rank, launched exactly as
bench.py is launched on the GPU node) over the emulator build of the kernels."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_data_parallel_world2_gloo(emu_lib):
    env = dict(os.environ, HIPEMU_THREADS="4", OMP_NUM_THREADS="2", HDU_DP_BUCKETS="0.3,0.6,0.9")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "tests", "dp_worker.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "DP_OK" in out.stdout
