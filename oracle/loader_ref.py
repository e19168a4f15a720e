"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference loader's per-item image work and collation, built on Pillow itself
(the third-party library that carries the arithmetic; installed here: Pillow 12.2, the reference's
environment.yml pins pillow 8.x - same 8-bit resample / blend / HSV code):

  * `resize_lanczos`, `pyramid`      - torchvision `Resize(size, LANCZOS)` on PIL images = `img.resize`
                                       (mono_dataset.py:70-74, :186-191)
  * `color_jitter`                   - torchvision 0.9 `ColorJitter.forward` on PIL images
                                       (functional_pil.adjust_brightness/contrast/saturation/hue);
                                       torchvision is ABSENT here, so this glue (which Pillow calls, in
                                       which order) is restated from its published source: PARITY
                                       UNPINNED for the glue, pinned for the Pillow arithmetic below it
  * `to_tensor`                      - torchvision ToTensor for uint8 RGB: HWC -> CHW, `.div(255)`
  * `preprocess_item`, `collate`     - mono_dataset.py:186-205 and trainer.py:867-886
"""
import numpy as np
import torch
from PIL import Image, ImageEnhance

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3


def resize_lanczos(img_u8, out_h, out_w, flip=False):
    im = Image.fromarray(img_u8, "RGB")
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)       # kitti_dataset.py:58-59
    return np.array(im.resize((out_w, out_h), Image.LANCZOS))


def pyramid(level0_u8, num_scales):
    out = [level0_u8]
    for _ in range(1, num_scales):
        h, w = out[-1].shape[:2]
        out.append(resize_lanczos(out[-1], h // 2, w // 2))
    return out


def color_jitter(img_u8, sequence):
    """sequence: [(op, factor)] in application order."""
    im = Image.fromarray(img_u8, "RGB")
    for op, f in sequence:
        if op == BRIGHTNESS:
            im = ImageEnhance.Brightness(im).enhance(f)
        elif op == CONTRAST:
            im = ImageEnhance.Contrast(im).enhance(f)
        elif op == SATURATION:
            im = ImageEnhance.Color(im).enhance(f)
        elif op == HUE:
            h, s, v = im.convert("HSV").split()
            np_h = np.array(h, dtype=np.uint8)
            np_h = (np_h.astype(np.int64) + (int(f * 255) & 0xFF)).astype(np.uint8)   # uint8 wrap-around add
            im = Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")
    return np.array(im)


def to_tensor(img_u8):
    return torch.from_numpy(np.ascontiguousarray(img_u8.transpose(2, 0, 1))).float().div(255)
