"""CPU tier: BASELINE configs[4] networks (MonoViT = MPViT-small encoder + HR decoder) against vectors
captured from the live reference (`tools/make_golden_vit.py`): state-dict layout incl. the alias keys,
eval / train forward, stochastic depth draw order, parameter gradients."""
import torch

import vit_checks


def test_state_dict_layout_matches_reference():
    vit_checks.check_state_dict_layout()


def test_forward_and_gradients_match_reference_cpu():
    torch.set_num_threads(4)
    vit_checks.check_forward_and_gradients("cpu", rel_eval=1e-4, rel_train=1e-4, rel_grad=2e-3)


def test_full_size_forward_matches_reference_cpu():
    torch.set_num_threads(4)
    vit_checks.check_full_size("cpu", rel=2e-3)


def test_checkpoint_roundtrip_with_alias_keys(tmp_path):
    """`encoder.pth` / `depth.pth` written from one instance load strictly into another (the shared
    position-encoding modules appear under several names; all of them must be present)."""
    from baseboostdepth_amd import networksvit
    torch.manual_seed(0)
    a, b = networksvit.mpvit_small(checkpoint=None), networksvit.mpvit_small(checkpoint=None)
    torch.save(a.state_dict(), tmp_path / "encoder.pth")
    b.load_state_dict(torch.load(tmp_path / "encoder.pth"))
    blk = b.mhca_stages[1].mhca_blks[0]
    assert blk.MHCA_layers[2].cpe is blk.cpe and blk.MHCA_layers[1].factoratt_crpe.crpe is blk.crpe
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k
    # an ImageNet-style checkpoint ({'model': ...} with a classifier head MonoViT lacks) loads non-strictly
    sd = dict(a.state_dict())
    sd["cls_head.cls.weight"] = torch.zeros(1000, 288)
    torch.save({"model": sd}, tmp_path / "mpvit_small.pth")
    c = networksvit.mpvit_small(checkpoint=str(tmp_path / "mpvit_small.pth"))
    assert torch.equal(c.stem[0].conv.weight, a.stem[0].conv.weight)
