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


@pytest.mark.parametrize("dtype", [np.uint8, np.int8, np.int16, np.uint16, np.int32, np.int64, np.float32, np.float64])
@pytest.mark.parametrize("shape", [(7, 64), (5, 3, 48), (9, 33), (1, 16), (4096, 16)])
def test_column_difference_and_squared_error_paths(mods, dtype, shape):
    """16-byte-vector (cols % (16/size) == 0) and scalar column kernels, row counts around the 4-row unroll, every
    element type: the Biomarker / PixelError answers equal the numpy restatement (uint inputs wrap like numpy)."""
    from oracle import ref_cpu as O
    rng = np.random.default_rng(11)
    hi = 2 if np.dtype(dtype).kind != "f" else 1
    a = (rng.integers(0, hi + 1, shape) if np.dtype(dtype).kind != "f" else rng.random(shape)).astype(dtype)
    b = (rng.integers(0, hi + 1, shape) if np.dtype(dtype).kind != "f" else rng.random(shape)).astype(dtype)
    for fname in ("bio.thickness_difference", "pixel.mean_squared_error", "pixel.root_mean_squared_error"):
        np.testing.assert_allclose(float(mods[fname](a, b)), float(O.METRIC_FUNCS[fname](a, b)),
                                   rtol=1e-12 if dtype != np.float32 else 1e-5, err_msg=fname)


def test_one_pass_evaluate_and_device_tensor_cache(golden_dir, mods):
    """dice + iou + precision + recall + accuracy + ... on one pair: ONE confusion pass, not one per metric"""
    from retinal_oct_image_segmentation_via_deep_learning_amd import Metrics
    from retinal_oct_image_segmentation_via_deep_learning_amd.Metrics import _counts
    z = np.load(os.path.join(golden_dir, "metrics.npz"))
    yt, yp = z["in/ragged_u8/y_true"], z["in/ragged_u8/y_pred"]
    n0 = _counts.launch_count[0]
    res = Metrics.evaluate(yt, yp)
    assert _counts.launch_count[0] == n0 + 1
    for name, key in (("dice_coefficient", "region.dice_coefficient"), ("iou_score", "region.iou_score"),
                      ("precision", "region.precision"), ("recall", "region.recall"), ("accuracy", "cm.accuracy"),
                      ("sensitivity", "cm.sensitivity"), ("cm_precision", "cm.precision"), ("specificity", "cm.specificity")):
        assert res[name] == z[f"out/ragged_u8/{key}"], name          # bit-exact, like the individual functions
    # the individual functions on the same DEVICE tensors share the pass
    dt, dp = torch.from_numpy(yt).cuda(), torch.from_numpy(yp).cuda()
    n0 = _counts.launch_count[0]
    vals = [mods[k](dt, dp) for k in ("region.dice_coefficient", "region.iou_score", "region.precision", "region.recall",
                                      "cm.accuracy", "cm.sensitivity", "cm.precision", "cm.specificity")]
    assert _counts.launch_count[0] == n0 + 1
    assert vals[0] == z["out/ragged_u8/region.dice_coefficient"] and vals[7] == z["out/ragged_u8/cm.specificity"]
    dp[0, 0, 0] ^= 1                                                   # in-place change: the cache must not answer
    d2 = mods["region.dice_coefficient"](dt, dp)
    assert _counts.launch_count[0] == n0 + 2
    ref = yp.copy(); ref[0, 0, 0] ^= 1
    from oracle import ref_cpu
    assert d2 == ref_cpu.dice_coefficient(yt, ref)
    # a NEW tensor object never hits, even if the allocator hands it the freed tensor's address
    del dp
    dp2 = torch.from_numpy(yp).cuda()
    mods["region.iou_score"](dt, dp2)
    assert _counts.launch_count[0] == n0 + 3


@pytest.mark.parametrize("dtype,C", [(np.uint8, 9), (np.int64, 9), (np.int32, 9), (np.int16, 9), (np.uint8, 3), (np.uint8, 8),
                                     (np.int64, 4), (np.uint8, 16), (np.int64, 16)])
def test_per_class_counts_one_pass_match_one_vs_rest_reference(dtype, C):
    """class maps in, [C][6] counts out: equal to the reference formulas on (y == c) masks for every class (class
    counts on either side of the kernel's 4 / 8 / 16 instances; labels beyond C are present)"""
    from oracle import ref_cpu
    from retinal_oct_image_segmentation_via_deep_learning_amd import Metrics
    from retinal_oct_image_segmentation_via_deep_learning_amd.Metrics import _counts
    rng = np.random.default_rng(5)
    yt = rng.integers(0, C, (3, 37, 53)).astype(dtype)
    yp = np.where(rng.random(yt.shape) < 0.7, yt, rng.integers(0, C + 2, yt.shape)).astype(dtype)   # some labels out of range
    n0 = _counts.launch_count[0]
    res = Metrics.evaluate(yt, yp, classes=C)
    assert _counts.launch_count[0] == n0 + 1
    for c in range(C):
        a, b = (yt == c).astype(np.int64), (yp == c).astype(np.int64)
        cs = ref_cpu.confusion_sums(a, b)
        assert res["counts"][c].tolist() == [cs["tp"], cs["t"], cs["p"], cs["tn"], cs["fp"], cs["fn"]]
        assert res["dice_coefficient"][c] == ref_cpu.dice_coefficient(a, b)
        assert res["iou_score"][c] == ref_cpu.iou_score(a, b)
        assert res["specificity"][c] == ref_cpu.specificity(a, b)
        assert res["accuracy"][c] == ref_cpu.accuracy(a, b)
    # device class maps straight from the model's predict(): int64
    dres = Metrics.evaluate(torch.from_numpy(yt.astype(np.int64)).cuda(), torch.from_numpy(yp.astype(np.int64)).cuda(), classes=C)
    assert np.array_equal(dres["counts"], res["counts"])
    with pytest.raises(TypeError):
        Metrics.evaluate(yt.astype(np.float32), yp.astype(np.float32), classes=C)


def test_full_size_pair_metrics_and_achieved_bandwidth(golden_dir):
    """32 x 512 x 1024 uint8 pair: answers equal the reference's; the kernel's achieved HBM rate
    is measured with HIP events on resident tensors (33.5 MB per pass; reported, loosely bounded)."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import Metrics, _lib as L
    z = np.load(os.path.join(golden_dir, "metrics.npz"))
    rng = np.random.default_rng(1234)
    a = torch.from_numpy((rng.random((32, 512, 1024)) < 0.3).astype(np.uint8)).cuda()
    b = torch.from_numpy((rng.random((32, 512, 1024)) < 0.3).astype(np.uint8)).cuda()
    res = Metrics.evaluate(a, b)
    assert res["dice_coefficient"] == z["out/seeded_32x512x1024/region.dice_coefficient"]
    assert res["iou_score"] == z["out/seeded_32x512x1024/region.iou_score"]
    assert res["accuracy"] == z["out/seeded_32x512x1024/cm.accuracy"]
    oi, of = torch.empty(6, dtype=torch.int64, device="cuda"), torch.empty(6, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        L.check(L.lib().oct_confusion_counts(a.data_ptr(), b.data_ptr(), 0, a.numel(), oi.data_ptr(), of.data_ptr(), st))
    e0.record()
    for _ in range(20):
        L.check(L.lib().oct_confusion_counts(a.data_ptr(), b.data_ptr(), 0, a.numel(), oi.data_ptr(), of.data_ptr(), st))
    e1.record()
    torch.cuda.synchronize()
    gbs = 2 * a.numel() / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9
    print(f"confusion_counts: {gbs:.0f} GB/s on 2 x 16.8 M uint8 (33.5 MB per pass, incl. the 6-word zero-fill launch)")
    assert gbs > 500.0
