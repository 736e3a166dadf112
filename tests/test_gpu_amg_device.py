"""GPU: the box NMS of the mask generator's device path (csrc/amg_device.hip, C-ABI saber_k_box_nms) against torchvision.ops.nms's definition
restated in numpy (stable descending score order, greedy, suppress box IoU > threshold, every step in fp32 in the same order): random boxes
in clusters (partial suppression), tied scores (stability), degenerate boxes, thresholds from 0 to 1, up to 3 072 boxes (one crop of the
default generator).  The end-to-end equality of the device path with the host path is tests/test_gpu_graphs.py."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def nms_ref(boxes, scores, thr):
    boxes = boxes.astype(np.float32); scores = scores.astype(np.float32)
    order = np.argsort(-scores, kind="stable")
    removed = np.zeros(len(boxes), bool)
    keep = []
    f = np.float32
    for a_, i in enumerate(order):
        if removed[a_]:
            continue
        keep.append(int(i))
        bi = boxes[i]
        rest = order[a_ + 1:]
        bj = boxes[rest]
        iarea = f(bi[2] - bi[0]) * f(bi[3] - bi[1])
        jarea = (bj[:, 2] - bj[:, 0]).astype(f) * (bj[:, 3] - bj[:, 1]).astype(f)
        w = np.maximum(f(0), np.minimum(bi[2], bj[:, 2]) - np.maximum(bi[0], bj[:, 0])).astype(f)
        h = np.maximum(f(0), np.minimum(bi[3], bj[:, 3]) - np.maximum(bi[1], bj[:, 1])).astype(f)
        inter = (w * h).astype(f)
        with np.errstate(invalid="ignore", divide="ignore"):
            ovr = inter / ((iarea + jarea).astype(f) - inter).astype(f)
        removed[a_ + 1:] |= ovr > f(thr)
    return keep


def test_box_nms_kernel_against_the_definition(gpu_lib):
    lib = gpu_lib
    p = lambda t: C.c_void_p(t.data_ptr())
    rng = np.random.default_rng(0)
    for n, clusters in ((1, 1), (7, 2), (200, 12), (1000, 40), (3072, 150), (3072, 3)):
        centres = rng.uniform(100, 900, (clusters, 2))
        sizes = rng.uniform(20, 300, (clusters, 2))
        which = rng.integers(0, clusters, n)
        c = centres[which] + rng.normal(0, 6, (n, 2))
        wh = np.maximum(1.0, sizes[which] + rng.normal(0, 8, (n, 2)))
        boxes = np.round(np.concatenate([c - wh / 2, c + wh / 2], 1)).astype(np.float32)        # integer coordinates like batched_mask_to_box
        if n >= 200:
            boxes[5] = boxes[4]                                         # exact duplicates
            boxes[9, 2:] = boxes[9, :2]                                 # zero-area box
        scores = rng.uniform(0.5, 1.0, n).astype(np.float32)
        scores[rng.integers(0, n, n // 3)] = np.float32(0.75)          # many exact ties: the order must be stable
        bd, sd = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
        scratch = torch.zeros(max(1, n) * 64, dtype=torch.uint8, device="cuda")
        keep = torch.zeros(max(1, n), dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        for thr in (0.0, 0.3, 0.7, 0.95, 1.0):
            assert lib.saber_k_box_nms(p(bd), p(sd), n, thr, p(scratch), p(keep), p(cnt), None) == 0, lib.saber_k_last_error()
            torch.cuda.synchronize()
            got = keep.cpu().numpy()[:int(cnt.item())].tolist()
            want = nms_ref(boxes, scores, thr)
            assert got == want, f"n={n} clusters={clusters} thr={thr}: {len(got)} vs {len(want)} kept"
        print(f"n={n}, {clusters} clusters: kept at IoU > 0.7: {len(nms_ref(boxes, scores, 0.7))}")
