"""Drop-in for lib/funcs.py:4-51 `predict_tumor_inwindow`: z-sliding-window inference with score averaging
(SURVEY.md section 8f, row N1).  Same arguments and return values; differences in mechanism only:
 * the CT volume, the accumulated `score` and `score_num` stay resident in HBM for the whole sweep (the reference
   round-trips every window through numpy and grows the TF graph with K.softmax/K.eval nodes per iteration,
   lib/funcs.py:31-32);
 * each window is copied device-to-device into the model's input buffer and run with Model.predict's launch list
   (learning_phase 0).
The 3-class softmax and the `score +=` of a window are ONE launch (hdu_softmax_accumulate) straight off the logits; how many
windows cover a slice (`score_num`) is a function of the window starts alone and is counted on the host."""
import numpy as np
import torch

from . import ops


def predict_tumor_inwindow(model, imgs_test, num, mini, maxi, args):
    batch = args.b
    img_deps, img_rows, img_cols = args.input_size, args.input_size, args.input_cols
    if batch != 1 or model.kind != "hybrid":
        raise ValueError("predict_tumor_inwindow drives the hybrid nets with b=1 (test.py:27-29)")
    window_cols = img_cols // 4                       # lib/funcs.py:12 (py2 integer division)
    x, y, z = imgs_test.shape[:3]
    if x < img_deps or y < img_rows or z < img_cols:
        raise ValueError("volume smaller than the network window")
    right_cols = int(min(z, maxi[2] + 10) - img_cols)
    left_cols = max(0, min(mini[2] - 5, right_cols))
    dev = model.ctx.dev
    # depth-major resident copy of the cropped volume: [z][deps][rows]
    vol = torch.as_tensor(np.ascontiguousarray(np.asarray(imgs_test[:img_deps, :img_rows, :], np.float32).transpose(2, 0, 1))).to(dev)
    if not 1 <= num <= 3:
        raise ValueError("num: 1..3 of the 3 class scores (test.py passes 3)")
    score = torch.zeros((z, img_deps, img_rows, num), dtype=torch.float32, device=dev)
    score_num = np.zeros((z, 1, 1, 1), np.float32)
    plane = img_deps * img_rows
    ctx = model.ctx
    a = model.logits.act
    for cols in range(left_cols, right_cols + window_cols, window_cols):
        c0 = z - img_cols if cols > z - img_cols else cols        # lib/funcs.py:26-28: last window is clamped
        model.vol.copy_(vol[c0:c0 + img_cols].reshape(-1))
        ctx.learning_phase = 0
        try:
            ctx.prep_weights()
            ctx.run_forward()
        finally:
            ctx.learning_phase = 1
        # first / last slice of each window dropped (lib/funcs.py:33): planes 1 .. img_cols-2 of the logits onto planes c0+1 ..
        ops.softmax_accumulate(a, plane, (img_cols - 2) * plane, num, score[c0 + 1:c0 + img_cols - 1].reshape(-1))
        score_num[c0 + 1:c0 + img_cols - 1] += 1
    score = score.cpu().numpy() / (score_num + np.float32(1e-4))        # lib/funcs.py:36
    out = np.zeros((x, y, z, num), np.float32)
    out[:img_deps, :img_rows] = score.transpose(1, 2, 0, 3)
    return out[:, :, :, num - 2], out[:, :, :, num - 1]


def liver_window_from_mask(mask):
    """test.py:57-62: bounding box (mini, maxi) of the dilated liver mask (labels 1 and 2 merged) that
    `predict_tumor_inwindow` sweeps; returns (dilated mask, mini, maxi)."""
    from scipy import ndimage
    m = np.array(mask, copy=True)
    m[m == 2] = 1
    m = ndimage.binary_dilation(m, iterations=1).astype(m.dtype)
    index = np.where(m == 1)
    if index[0].size == 0:
        raise ValueError("empty liver mask")
    return m, np.min(index, axis=-1), np.max(index, axis=-1)


def _largest_component(binary):
    """skimage.measure.label(..., return_num=True) + regionprops areas + `box.index(max(box)) + 1` (test.py:84-92):
    full connectivity (skimage's default for label is connectivity = ndim), first label wins a tie"""
    from scipy import ndimage
    lab, num = ndimage.label(binary, structure=np.ones((3,) * binary.ndim, dtype=bool))
    if num == 0:
        raise ValueError("no foreground component (the reference raises on max([]) here, test.py:90)")
    areas = np.bincount(lab.ravel(), minlength=num + 1)[1:]
    keep = int(np.argmax(areas)) + 1          # argmax returns the FIRST maximum, like list.index(max(...))
    return (lab == keep).astype(lab.dtype)


def segment_liver_tumor(score1, score2, mask, thres_liver=0.5, thres_tumor=0.8):
    """Host-side post-processing of test.py:70-112: threshold the averaged scores,
    keep the largest liver component, restrict tumours to the (dilated, largest, hole-filled) coarse liver mask, fill
    holes, and return the uint8 label volume {0 background, 1 liver, 2 tumour} that the reference saves as NIfTI.
    `mask` is the ALREADY dilated coarse liver mask of liver_window_from_mask (test.py dilates it a second time
    before labelling, :95 -- reproduced).  scipy.ndimage replaces skimage.measure (same connectivity and tie rule)."""
    from scipy import ndimage
    result1 = np.array(score1, dtype=np.float64, copy=True)
    result2 = np.array(score2, dtype=np.float64, copy=True)
    result1 = (result1 >= thres_liver).astype(np.float64)
    result2 = (result2 >= thres_tumor).astype(np.float64)
    result1[result2 == 1] = 1
    segmask = result2
    liver_res = _largest_component(result1)
    m = ndimage.binary_dilation(mask, iterations=1).astype(np.asarray(mask).dtype)
    liver_labels = _largest_component(m)
    liver_labels = ndimage.binary_fill_holes(liver_labels).astype(int)
    segmask = segmask * liver_labels
    segmask = ndimage.binary_fill_holes(segmask).astype(int).astype(np.uint8)
    liver_res = ndimage.binary_fill_holes(liver_res.astype(np.uint8)).astype(int)
    liver_res[segmask == 1] = 2
    return liver_res.astype(np.uint8)
