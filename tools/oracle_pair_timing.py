"""developer: serial vs two-thread timing of the parity tests' oracle pair on the current host (no GPU work)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_utils as U
import test_gpu_parity_bf16 as T
kind, variant, b, size, cols = "hybrid", "end2end", 1, 224, 12
P0 = U.R.ParamStore(seed=3, dtype=torch.float32, perturb=True)
fwd = U.oracle_forward_fn(kind, variant, T.FULL2D, T.FULL3D)
x, y = U.synthetic_batch(kind, b, size, cols)
with torch.no_grad():
    fwd(P0, torch.tensor(x))
W = P0.numpy()
xt, yt = torch.tensor(x), torch.tensor(y)
print("cpu_count", os.cpu_count(), "torch threads", torch.get_num_threads(), flush=True)
for mode in ("1", "0"):
    os.environ["HDU_PARITY_SERIAL_ORACLE"] = mode
    t = time.time()
    T.oracle_pair(W, kind, variant, b, size, cols, T.FULL2D, T.FULL3D, xt, yt)
    print("serial" if mode == "1" else "two threads, half the cores each", round(time.time() - t, 1), "s", flush=True)
for nt in (32, 64):
    torch.set_num_threads(nt)
    os.environ["HDU_PARITY_SERIAL_ORACLE"] = "1"
    t = time.time()
    T.oracle_pair(W, kind, variant, b, size, cols, T.FULL2D, T.FULL3D, xt, yt)
    print("serial with", nt, "threads", round(time.time() - t, 1), "s", flush=True)
