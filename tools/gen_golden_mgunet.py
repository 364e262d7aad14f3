#!/usr/bin/env python3
"""Golden vectors for MGU-Net (SURVEY.md §8 a9 / the §8(b) constructor list), made by IMPORTING
  /root/reference/SOTAS/Layers_Segment/MGUNet_2021.py   GloRe_Unit (:110-148), MGR_Module (:150-194), MGUNet (:197-252), MGUNet_2 (:255-309)
in the build container.  Nothing of its source is copied; the fixtures hold inputs and outputs (float64 runs on
fp32-representable weights / inputs), the networks a seeded recipe + checksums of the reference's tensors (oracle/cases.py).
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gen_golden_blocks as G  # noqa: E402  (imports the reference module as G.ref_mg, sets up OUT)

ref = G.ref_mg


def main():
    G.torch.set_num_threads(8)
    G.block_case("blk_basconv1x1", lambda: ref.Basconv(8, 4, kernel_size=1, padding=0), [(2, 8, 6, 10)], 600)
    G.block_case("blk_glore", lambda: ref.GloRe_Unit(8, 4), [(2, 8, 6, 10)], 610)
    # 11 x 17: none of the 2 / 3 / 5 poolings divides it (torch's floor mode), the 5 x 5 branch ends at 2 x 3
    G.block_case("blk_mgr", lambda: ref.MGR_Module(8, 16), [(2, 8, 11, 17)], 620)
    G.net_case("mgunet2_c3_2x48x64", lambda ci, nc: ref.MGUNet_2(ci, nc, feature_scale=16), 700, 2, 1, 3, 48, 64,
               thresh=1e-5, full_weights=False)
    G.net_case("mgunet_c2_2x160x192", lambda ci, nc: ref.MGUNet(ci, nc, feature_scale=16), 800, 2, 1, 2, 160, 192,
               thresh=3e-6, full_weights=False, compact=True)
    # API facts: default-argument parameter counts, the failure of an input whose bottleneck is smaller than the 5 x 5 pool
    rec = {"mgunet2_default_params": G.np.array(sum(p.numel() for p in ref.MGUNet_2().parameters())),
           "mgunet_default_params": G.np.array(sum(p.numel() for p in ref.MGUNet().parameters())),
           "mgunet2_keys": G.np.array(list(ref.MGUNet_2(1, 3, feature_scale=16).state_dict().keys()))}
    try:
        ref.MGUNet_2(1, 3, feature_scale=16)(G.torch.zeros(2, 1, 32, 32))     # bottleneck 4 x 4 < 5
        rec["small_msg"] = G.np.array("")
    except RuntimeError as e:
        rec["small_msg"] = G.np.array(str(e))
    try:
        ref.MGUNet_2(1, 3, feature_scale=16)(G.torch.zeros(1, 1, 48, 64))     # one image: the 5 x 5 branch is 1 x 1 -> train-mode BN refuses
        rec["single_msg"] = G.np.array("")
    except ValueError as e:
        rec["single_msg"] = G.np.array(str(e))
    G.np.savez_compressed(os.path.join(G.OUT, "mgunet_api.npz"), **rec)
    print("mgunet_api:", int(rec["mgunet2_default_params"]), int(rec["mgunet_default_params"]), "|", str(rec["small_msg"])[:90])


if __name__ == "__main__":
    main()
