"""Import shim for the upstream reference (this container only).

`/root/reference` is pure Python but imports packages this image lacks (cv2, wandb,
skimage, torchvision).  The hot path never touches them, so empty module stubs are
enough to import `trainer`, `layers` and `networks` unmodified.

Nothing here travels to the GPU box as a dependency: it is used only by
`tools/make_golden.py` (fixture generation) and by the `-m "not gpu"` tests that
cross-check `oracle/` against the live reference when `/root/reference` exists.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("BBD_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "trainer.py"))


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def install_stubs():
    """Put placeholder modules in sys.modules for the reference's unused imports."""
    if "cv2" not in sys.modules:
        _stub("cv2")
    if "wandb" not in sys.modules:
        _stub("wandb")
    if "skimage" not in sys.modules:
        sk = _stub("skimage")
        sk.transform = _stub("skimage.transform")
    try:
        import torchvision  # noqa: F401
    except Exception:
        class _ResNet:  # placeholder base class for the reference encoder module
            pass

        tv = _stub("torchvision")
        models = _stub("torchvision.models", ResNet=_ResNet)
        models.resnet = _stub("torchvision.models.resnet")
        tv.models = models
        tr = _stub("torchvision.transforms",
                   InterpolationMode=types.SimpleNamespace(LANCZOS=1))
        tv.transforms = tr


def install_vit_stubs():
    """Placeholders for what `networksvit/mpvit.py:18-32` imports and this image lacks: timm 0.6.12
    (`DropPath`, `trunc_normal_`, the ImageNet mean/std constants), mmcv-full 1.4.0 (`build_norm_layer`,
    `load_checkpoint`, `load_state_dict`) and mmseg 0.19 (`get_root_logger`, the `BACKBONES` registry).
    `DropPath` restates timm's published stochastic-depth rule (per-sample Bernoulli(keep) mask divided by
    keep, identity in eval mode); `build_norm_layer(dict(type="BN"), c)` is mmcv's `("bn", BatchNorm2d(c))`.
    The ImageNet checkpoint the reference loads unconditionally (`./ckpt/mpvit_small.pth`, mpvit.py:815,
    git-ignored upstream) does not exist: `torch.load` of that path is answered with an empty state dict."""
    import torch
    import torch.nn as nn

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                mask.div_(keep)
            return x * mask

    def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

    def build_norm_layer(cfg, num_features, postfix=""):
        assert cfg.get("type") == "BN"
        layer = nn.BatchNorm2d(num_features, eps=cfg.get("eps", 1e-5))
        for p in layer.parameters():
            p.requires_grad = cfg.get("requires_grad", True)
        return "bn" + str(postfix), layer

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    timm = _stub("timm")
    timm.data = _stub("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    timm.models = _stub("timm.models")
    timm.models.layers = _stub("timm.models.layers", DropPath=DropPath, trunc_normal_=trunc_normal_)
    mmcv = _stub("mmcv")
    mmcv.runner = _stub("mmcv.runner", load_checkpoint=lambda *a, **k: None, load_state_dict=lambda *a, **k: None)
    mmcv.cnn = _stub("mmcv.cnn", build_norm_layer=build_norm_layer)
    mmseg = _stub("mmseg")
    mmseg.utils = _stub("mmseg.utils", get_root_logger=lambda *a, **k: None)
    mmseg.models = _stub("mmseg.models")
    mmseg.models.builder = _stub("mmseg.models.builder", BACKBONES=_Registry())


def import_reference_vit():
    """The reference's `networksvit` package (MonoViT: MPViT encoder + HR-Depth decoder)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    install_stubs()
    install_vit_stubs()
    import torch
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    real_load = torch.load

    def fake_load(path, *a, **k):
        if isinstance(path, str) and path.startswith("./ckpt/"):
            return {"model": {}}
        return real_load(path, *a, **k)
    torch.load = fake_load
    try:
        import networksvit
    finally:
        torch.load = real_load
    networksvit._bbd_fake_load = fake_load
    return networksvit


def import_reference():
    """Returns (trainer_module, layers_module, networks_module) of the reference."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    install_stubs()
    # The reference's trainer pins thread-count env vars at import; keep ours.
    saved = {k: os.environ.get(k) for k in
             ("MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS", "OMP_NUM_THREADS")}
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    try:
        import matplotlib
        matplotlib.use("Agg")
        import trainer as ref_trainer
        import layers as ref_layers
        import networks as ref_networks
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return ref_trainer, ref_layers, ref_networks
