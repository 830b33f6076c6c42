"""Depth sharding of one volume (halo exchange, halo-gradient reduce, sync-BN, summed gradient) on 2 CPU ranks (gloo)
over the emulator build of the kernels: the sharded training step must reproduce the unsharded one."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_depth_shard_world2_gloo(emu_lib):
    env = dict(os.environ, HIPEMU_THREADS="4", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tests", "shard_worker.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-5000:]
    assert "SHARD_OK" in out.stdout
