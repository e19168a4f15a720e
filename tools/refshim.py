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
