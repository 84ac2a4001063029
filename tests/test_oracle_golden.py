"""CPU: pin oracle/mvster_oracle.py against the fixtures produced by the reference itself."""
import pytest
import torch

from oracle import mvster_oracle as orc
from tests.util import golden, golden_names, stage_case

TIGHT = dict(rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", golden_names("warp_"))
def test_warp_matches_reference(name):
    z = golden(name)
    src = z["src"].clone().requires_grad_("gout" in z)
    out = orc.homo_warping(src, z["src_proj"], z["ref_proj"], z["depth"])
    torch.testing.assert_close(out, z["out"], **TIGHT)
    if "gout" in z:
        (out * z["gout"]).sum().backward()
        torch.testing.assert_close(src.grad, z["gsrc"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", golden_names("stage_"))
def test_stage_matches_reference(name):
    cfg, z = stage_case(name)
    feats = [z["feat%d" % i].clone().requires_grad_(True) for i in range(cfg["N"])]
    seen = {}

    def regnet(vol):
        seen["vol"] = vol
        return (vol * z["chan_w"].view(1, -1, 1, 1, 1)).sum(1)

    out = orc.stage_forward(feats, z["proj"], z["hypo"], regnet, cfg["training"], group_cor=cfg["group_cor"],
                            group_cor_dim=cfg["G"], split_itv=cfg["split_itv"], inverse_depth=True, mono=True,
                            attn_fuse_d=cfg["attn_fuse_d"], attn_temp=cfg["temp"])
    torch.testing.assert_close(seen["vol"], z["cor_feats"], **TIGHT)
    # stub logits = 12x-weighted channel sums: 1-ulp differences of the volume are amplified
    torch.testing.assert_close(out["attn_weight"], z["attn_weight"], rtol=2e-5, atol=5e-6)
    assert torch.equal(out["depth"], z["depth"])            # gather of identical hypotheses: exact
    torch.testing.assert_close(out["inverse_min_depth"], z["inv_min"], rtol=0, atol=0)
    torch.testing.assert_close(out["inverse_max_depth"], z["inv_max"], rtol=0, atol=0)
    if cfg["training"]:
        assert out["photometric_confidence"].ndim == 0 and float(out["photometric_confidence"]) == 0.0
    else:
        # fully out-of-view pixels have all-zero logits: the reference yields 0/0 = NaN there
        torch.testing.assert_close(out["photometric_confidence"], z["conf"], rtol=2e-5, atol=5e-6, equal_nan=True)
    assert out["mono_feat"] is feats[0]
    assert out["hypo_depth"] is z["hypo"]
    (out["attn_weight"] * z["loss_w"]).sum().backward()
    for i, f in enumerate(feats):
        ref = z["gfeat%d" % i]
        torch.testing.assert_close(f.grad, ref, rtol=1e-4, atol=2e-5 * float(ref.abs().max()))


@pytest.mark.parametrize("name", golden_names("range_init"))
def test_init_ranges(name):
    z = golden(name)
    dv = z["depth_values"]
    B, D, H, W = z["inverse"].shape
    assert torch.equal(orc.init_inverse_range(dv, D, dv.device, dv.dtype, H, W), z["inverse"])
    assert torch.equal(orc.init_range(dv, D, dv.device, dv.dtype, H, W), z["linear"])
    # plane 0 of the inverse variant is the FAR plane
    assert torch.all(z["inverse"][:, 0] > z["inverse"][:, -1])


@pytest.mark.parametrize("name", golden_names("range_sched"))
def test_schedule_inverse_range(name):
    z = golden(name)
    B, D, H, W = z["out"].shape
    assert torch.equal(orc.schedule_inverse_range(z["inv_min"], z["inv_max"], D, H, W), z["out"])


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_argmax_ties_first_index(mode):
    z = golden("ties_" + mode)
    out = orc.regress_depth(z["logits"], z["hypo"], training=(mode == "train"), inverse_depth=True, split_itv=0.5)
    assert torch.equal(out["argmax_index"], z["index"])
    assert torch.equal(out["depth"], z["depth"])
    assert torch.equal(out["attn_weight"], z["attn_weight"])
    assert int(z["index"][0, 0, 0]) == 0 and int(z["index"][0, 0, 1]) == 3 and int(z["index"][0, 0, 2]) == 1
    if mode == "eval":
        torch.testing.assert_close(out["photometric_confidence"], z["conf"], rtol=0, atol=0)
    torch.testing.assert_close(out["inverse_min_depth"], z["inv_min"], rtol=0, atol=0)
