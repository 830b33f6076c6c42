"""Depth sharding of one volume (halo exchange, halo-gradient reduce, sync-BN, summed gradient) on CPU ranks (gloo) over the
emulator build of the kernels: the sharded training step must reproduce the unsharded one."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("halo_wide", ["", "4", "8"], ids=["default-kernels", "halo-wide-16x96", "halo-wide-16x64p"])
def test_depth_shard_world4_gloo(emu_lib, halo_wide):
    """4 ranks x 4 depth planes of the stand-alone 3D net: ranks 0 and 3 are edge shards (zero padding on one side),
    ranks 1 and 2 INTERIOR shards -- two depth neighbours each, halos received from and halo gradients returned to both
    sides.  (Replaces the world-2 run of round 1, which only had edge shards.)
    Round 5: run again with the halo-tile forward / data-gradient kernel and the stem kernels FORCED onto every layer they cover
    (HDU_HALO_WIDE: at 32 x 32 the size thresholds would keep them out) -- their depth-valid / cropped forms over stored halo planes,
    in interior and edge shards, against the unsharded net built with the same kernels."""
    env = dict(os.environ, HIPEMU_THREADS="2", OMP_NUM_THREADS="1", SHARD_TEST_DL="4", SHARD_TEST_H="32")
    if halo_wide:
        env["HDU_HALO_WIDE"] = halo_wide
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
           "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "tests", "shard_worker.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=2400)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-5000:]
    assert "SHARD_OK" in out.stdout and out.stdout.count("rank ") >= 4


@pytest.mark.parametrize("net,port,comm", [("end2end", "29551", ""), ("3dpart", "29553", ""), ("end2end", "29555", "double")],
                         ids=["end2end", "3dpart", "end2end-comm-object"])
def test_depth_sharded_hybrid_world2_gloo(emu_lib, net, port, comm):
    """SURVEY.md section 8e, third row: the HYBRID nets on one volume split over 2 ranks -- each rank runs the 2D branch on
    its own slices (one raw CT plane exchanged with each depth neighbour for the 2.5D slabs, denseunet3d.py:399-409), the
    3D net with halo exchange / sync-BN, the HFF add + `fianl_conv` with a halo, loss.py's slices 1:7 split over the
    ranks; the sharded training step reproduces the unsharded one (logits, loss, all-reduced gradient, weights, moving
    statistics).  end2end also returns the stem's halo gradients (stride-2 7x7x7 data gradient) to the 2D branch."""
    # Default split-K: a shard's launches take their tile-shape / split decisions for the WHOLE layer
    # (hdu_conv_desc.layer_rows = the unsharded layer's pixel count), so every output is summed over the same K partition as in the unsharded run.
    # (Round 2 had to force HDU_SPLITK=1 here: the split count followed the shard's own tile count and the different float32
    # summation order alone moved the hybrid's logits -- 250 x logits2d feed the 3D stem -- by 3e-4 of max|logit|.)
    # "end2end-comm-object": the same run with ShardInfo.comm set (a gloo-backed double of h-denseunet_amd.comm.Comm): every
    # halo exchange, halo-gradient return, CT-plane exchange, sync-BN table and the flat gradient go through the
    # `sh.comm.sendrecv / allreduce_` call sites that HDU_COMM=rccl_abi uses on hardware
    env = dict(os.environ, HIPEMU_THREADS="4", OMP_NUM_THREADS="2", SHARD_TEST_DL="4", SHARD_TEST_H="32", SHARD_TEST_NET=net,
               SHARD_TEST_COMM=comm)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(ROOT, "tests", "shard_worker.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=2400)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-5000:]
    assert "SHARD_OK" in out.stdout
