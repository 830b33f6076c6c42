"""bench.py's record helpers (host logic, no GPU): the time-weighted whole-step roofline, the parity field read from the committed
figures, the 3D dense-block layer filter."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_step_roofline_weights_every_kernel_by_its_time():
    bench = importlib.import_module("bench")
    # kernel -> [launches, ms, flops, bytes]: an MFMA-bound kernel at half its roof for 2 ms, an HBM-bound one at a quarter for 1 ms,
    # an unmodelled one for 1 ms
    agg = {"mfma": [1, 2.0, 0.5 * 2500e12 * 2e-3, 1e6],
           "hbm": [4, 1.0, 0.0, 0.25 * 8000e9 * 1e-3],
           "fold": [10, 1.0, 0.0, None]}
    r = bench.step_roofline(agg, "bf16")
    assert abs(r["time_weighted_frac"] - (2.0 * 0.5 + 1.0 * 0.25 + 0.0) / 4.0) < 1e-3
    assert r["mfma_bound_time_share"] == 0.5 and r["hbm_bound_time_share"] == 0.25 and r["unmodelled_time_share"] == 0.25


def test_parity_field_reads_the_committed_figures():
    bench = importlib.import_module("bench")
    rec = bench.parity_of_timed_mode("2d", "bf16")
    assert rec is not None and rec["dtype"] == "bf16" and rec["case"].endswith("/mid")      # the benchmarked batch is the last mid case
    assert len(rec["dice_deficit_per_class"]) == 3 and 0 < rec["logit_max_abs_err"] < 1.0
    hyb = bench.parity_of_timed_mode("end2end", "bf16")
    assert hyb["case"].endswith("/trained") and "max_dice_deficit_by_case" in hyb
    f32 = bench.parity_of_timed_mode("2d", "f32")
    assert f32["dtype"] == "f32" and f32["logit_max_abs_err"] <= 1e-4 and max(f32["dice_deficit_per_class"]) <= 1e-3


def test_dense_block_layer_filter():
    bench = importlib.import_module("bench")
    f = bench._is_3d_dense_block_conv
    assert f("3dconv2_1_x1") and f("3dconv5_8_x2")
    assert not f("3dconv_up3") and not f("conv2_1_x1") and not f("3dconv3_blk") and not f("3dconv1")
