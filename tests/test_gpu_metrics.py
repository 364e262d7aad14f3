"""GPU parity of Metrics/{Region,ConfusionMatrix}_based_metrics against the reference's answers
(tests/golden/metrics.npz, produced by importing the reference) -- bit-exact for integer masks."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from retinal_oct_image_segmentation_via_deep_learning_amd.Metrics import ConfusionMatrix_based_metrics as cm
    from retinal_oct_image_segmentation_via_deep_learning_amd.Metrics import Region_based_metrics as rg
    from retinal_oct_image_segmentation_via_deep_learning_amd.Metrics import PixelError_based_metrics as px
    from retinal_oct_image_segmentation_via_deep_learning_amd.Metrics import Biomarker_based_metrics as bio
    return {"pixel.mean_squared_error": px.mean_squared_error, "pixel.root_mean_squared_error": px.root_mean_squared_error,
            "bio.thickness_difference": bio.thickness_difference, "bio.vascularity_index": bio.vascularity_index,
            "region.dice_coefficient": rg.dice_coefficient, "region.iou_score": rg.iou_score,
            "region.precision": rg.precision, "region.recall": rg.recall, "cm.accuracy": cm.accuracy,
            "cm.sensitivity": cm.sensitivity, "cm.precision": cm.precision, "cm.specificity": cm.specificity}


def test_known_answers_all_dtypes(golden_dir, mods):
    z = np.load(os.path.join(golden_dir, "metrics.npz"))
    cases = sorted({k.split("/")[1] for k in z.files if k.startswith("in/")})
    for c in cases:
        yt, yp = z[f"in/{c}/y_true"], z[f"in/{c}/y_pred"]
        for fname, fn in mods.items():
            key = f"out/{c}/{fname}"
            if key not in z.files:
                continue
            got = fn(yt, yp)
            if fname.startswith(("pixel.", "bio.")):
                # fp64 / wrapped-uint64 means: the reference's pairwise float summation order is not
                # reproduced bit for bit (values up to 1.8e19 in the wrapped uint8 case)
                np.testing.assert_allclose(float(got), float(z[key]), rtol=1e-12 if yt.dtype != np.float32 else 1e-5,
                                           atol=1e-15, err_msg=key)
            elif yt.dtype == np.float32:
                assert isinstance(got, np.float32)
                np.testing.assert_allclose(got, z[key], rtol=1e-5, err_msg=key)  # reference sums in fp32
            else:
                assert float(got) == float(z[key]), (key, got, float(z[key]))  # bit-exact


def test_seeded_full_size_and_device_tensors(golden_dir, mods):
    z = np.load(os.path.join(golden_dir, "metrics.npz"))
    rng = np.random.default_rng(1234)
    a = (rng.random((32, 512, 1024)) < 0.3).astype(np.uint8)
    b = (rng.random((32, 512, 1024)) < 0.3).astype(np.uint8)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    for fname, fn in mods.items():
        if fname.startswith(("pixel.", "bio.")):
            np.testing.assert_allclose(float(fn(ta, tb)), float(z[f"out/seeded_32x512x1024/{fname}"]), rtol=1e-12)
        else:
            assert float(fn(ta, tb)) == float(z[f"out/seeded_32x512x1024/{fname}"]), fname
    from retinal_oct_image_segmentation_via_deep_learning_amd.Metrics._counts import confusion_sums
    s, n, _ = confusion_sums(ta, tb)
    assert [s[0], s[1], s[2], n] == z["seeded_counts"].tolist()
    assert s[0] + s[3] + s[4] + s[5] == n  # tp + tn + fp + fn partitions the pixels
    # bool tensors and class-map derived masks
    assert float(mods["region.dice_coefficient"](ta.bool(), tb.bool())) == float(z["out/seeded_32x512x1024/region.dice_coefficient"])


def test_uint8_wraparound_matches_numpy(mods):
    """numpy evaluates t*p and 1-t in uint8: 255-valued masks wrap.  The kernel reproduces that."""
    from oracle import ref_cpu as O
    rng = np.random.default_rng(3)
    a = (rng.random((7, 33)) < 0.5).astype(np.uint8) * 255
    b = (rng.random((7, 33)) < 0.5).astype(np.uint8) * 255
    for fname, fn in mods.items():
        np.testing.assert_allclose(float(fn(a, b)), float(O.METRIC_FUNCS[fname](a, b)), rtol=1e-12, err_msg=fname)
    e = np.zeros((0,), dtype=np.uint8)
    assert float(mods["region.dice_coefficient"](e, e)) == 0.0
