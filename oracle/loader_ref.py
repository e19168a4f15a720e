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


def preprocess_item(recipe, scales, H, W):
    """What the reference's `MonoDataset.__getitem__` returns (mono_dataset.py:76-146 + preprocess
    :186-205) for the frames / flip / jitter draws recorded in `recipe` (see baseboostdepth_amd/
    datasets.py).  Only the keys `custom_collate` can select are produced."""
    inputs = {}
    for f, img in recipe["images"].items():
        levels = pyramid(resize_lanczos(img, H, W, recipe["flip"]), max(scales) + 1)
        if f == 0 or f == "s":
            for s in scales:
                inputs[("color", f, s)] = to_tensor(levels[s])
        if f != "s":
            inputs[("color", f, 0)] = to_tensor(levels[0])
            inputs[("color_aug", f, 0)] = to_tensor(color_jitter(levels[0], recipe["jitter"].get(f, [])))
    if "K" in recipe:
        inputs[("K", 0)] = torch.from_numpy(recipe["K"])
        inputs[("inv_K", 0)] = torch.from_numpy(recipe["inv_K"])
    inputs["stereo_T"] = torch.from_numpy(recipe["stereo_T"])
    inputs["frames"] = recipe["frames"]
    inputs["cutt_off"] = recipe["cutt_off"]
    inputs["to_use"] = recipe["to_use"]
    return inputs


def select_frames_ref(line, epoch, trimin, rand, is_train, draws, exists):
    """Frame-set selection of mono_dataset.py:58-63, :77-106 for one split line.  `draws` is an object
    with `.random()` / `.randint(a, b)` (the reference uses the `random` module); `exists(offset)`
    says whether frame_index+offset is on disk.  Returns (do_color_aug, do_flip, frame_idxs)."""
    if epoch < 10:
        to_use = 2 if trimin else 1
        cutt_off = 0.1 + (0.04 * epoch)
    else:
        to_use = 7 if trimin else 5
        cutt_off = (0.15 * epoch) - 0.9
    do_color_aug = is_train and draws.random() > 0.5
    do_flip = is_train and draws.random() > 0.5
    baseline = line.split()[-1] if (rand and is_train) else 0
    if is_train:
        if rand:
            frame_idxs = sorted([i for i in range(-to_use, to_use + 1) if (abs(i) * float(baseline)) <= cutt_off], key=abs)
            if max(frame_idxs) < 3:
                frame_idxs.append("s")
        else:
            frame_idxs = [0, 1, -1, "s"]
    else:
        frame_idxs = [0]
    if is_train:
        mini = draws.randint(1, 6) if draws.random() > 0.7 else 0
        limit_pos = max([i for i in range(1, 8 - mini) if exists(i)])
        limit_neg = max([abs(i) for i in range(-1, -8 + mini, -1) if exists(i)])
        limit = min([limit_pos, limit_neg])
    else:
        limit = 7
    frame_idxs[:] = [x for x in frame_idxs if x != "s" and abs(x) <= abs(limit)]
    if max(frame_idxs) < 3:
        frame_idxs.append("s")
    return do_color_aug, do_flip, frame_idxs
