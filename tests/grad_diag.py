"""Diagnostic, TEST INFRASTRUCTURE (lives under tests/ because it imports oracle/): per-tensor gradient error of one f32
training step vs the oracle.  usage: python tests/grad_diag.py <kind> <variant> <b> <size> <cols|0> [emu]"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import parity_utils as U
kind, variant, b, size, cols = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) or None
if len(sys.argv) > 6:
    import emu_bind
    emu_bind.use_emulator()
nb2, nb3 = ((6, 12, 36, 24), (3, 4, 12, 8)) if size >= 128 else ((2, 2, 2, 2), (1, 1, 2, 1))
m, P, fwd = U.build_pair(kind, variant, b, size, cols, "f32", nb2, nb3, odtype=torch.float32)
m.ctx.dropout_enabled = False
x, y = U.synthetic_batch(kind, b, size, cols)
ka = U.pkg("keras_api")
m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
ref_loss, ref_grads, ref_logits = U.R.train_step(P, fwd, U.loss_fn_for(kind), torch.tensor(x), torch.tensor(y), {})
loss = m.train_on_batch(x, y)
print("loss", loss, ref_loss, "logits err", np.abs(m._download_logits().cpu().numpy() - ref_logits.numpy()).max())
gg = m.get_grads_dict()
rms_max = max(float(np.sqrt((g.numpy().astype(np.float64) ** 2).mean())) for g in ref_grads.values())
rows = []
for (name, i), g in ref_grads.items():
    a, r = gg[name][i].astype(np.float64), g.numpy().astype(np.float64)
    den = max(np.linalg.norm(r), 1e-3 * rms_max * np.sqrt(r.size))
    rows.append((np.linalg.norm(a - r) / den, name, i, np.linalg.norm(r), np.linalg.norm(a), r.size))
order = {n: k for k, n in enumerate(m.layer_names())}
for r in sorted(rows, reverse=True)[:25]:
    print("%.4f %-22s %d  |ref| %.3e |got| %.3e n=%d  layer#%d" % (r + (order[r[1]],)))
print("median %.5f" % np.median([r[0] for r in rows]))
