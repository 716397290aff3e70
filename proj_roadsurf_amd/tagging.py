"""Tagged sample images: the PNG previews the object-detector scripts leave in ``sample_tagged_img_subfolder``
(``sample_detection_images`` for make_detections.py, R:config/config_obj_detec.yaml:77; ``sample_training_images`` for
train_model.py, R:config/config_obj_detec.yaml:65).  The reference draws them with detectron2's ``Visualizer``
([EXT d2: utils/visualizer.py] ``draw_instance_predictions`` / ``draw_dataset_dict``, matplotlib); these are a plain PIL
rendering of the same content -- translucent mask / polygon fill in a per-class colour, the box, and a "<class> <score>%" label --
not a pixel copy of the Visualizer's styling.  Host only; nothing here is on the detection path."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

PALETTE = [(230, 60, 60), (60, 130, 230), (60, 190, 90), (240, 180, 40), (170, 80, 200), (40, 200, 200), (240, 120, 40), (150, 150, 150)]


def class_colour(k: int):
    return PALETTE[int(k) % len(PALETTE)]


def _label(draw, xy, text, colour):
    x, y = float(xy[0]), float(xy[1])
    try:
        l, t, r, b = draw.textbbox((x, y), text)
    except AttributeError:                                  # very old Pillow
        w, h = draw.textsize(text)
        l, t, r, b = x, y, x + w, y + h
    draw.rectangle([l - 1, t - 1, r + 1, b + 1], fill=(0, 0, 0))
    draw.text((x, y), text, fill=colour)


def draw_instances(image_rgb: np.ndarray, boxes: np.ndarray, classes: np.ndarray, scores: Optional[np.ndarray] = None,
                   masks: Optional[np.ndarray] = None, class_names: Optional[Sequence[str]] = None, alpha: float = 0.4):
    """image_rgb (H,W,3) uint8; boxes (n,4) x1,y1,x2,y2 in image pixels; masks (n,H,W) bool or None.  Returns a PIL image."""
    from PIL import Image, ImageDraw
    img = np.ascontiguousarray(image_rgb[:, :, :3]).astype(np.float32)
    n = len(boxes)
    order = np.argsort(scores) if scores is not None and n else np.arange(n)        # best score drawn last (on top)
    if masks is not None:
        for i in order:
            m = np.asarray(masks[i], bool)
            if m.shape == img.shape[:2] and m.any():
                img[m] = (1.0 - alpha) * img[m] + alpha * np.array(class_colour(classes[i]), np.float32)
    out = Image.fromarray(np.clip(img + 0.5, 0, 255).astype(np.uint8))
    d = ImageDraw.Draw(out)
    for i in order:
        c = class_colour(classes[i])
        x1, y1, x2, y2 = [float(v) for v in boxes[i]]
        d.rectangle([x1, y1, max(x2, x1), max(y2, y1)], outline=c)
        name = class_names[int(classes[i])] if class_names is not None and int(classes[i]) < len(class_names) else str(int(classes[i]))
        _label(d, (x1 + 1, y1 + 1), name if scores is None else f"{name} {100.0 * float(scores[i]):.0f}%", c)
    return out


def draw_annotations(image_rgb: np.ndarray, annotations: Sequence[dict], class_of_category: dict,
                     class_names: Optional[Sequence[str]] = None, alpha: float = 0.4):
    """Ground truth of one COCO image: ``annotations`` with ``bbox`` [x,y,w,h], ``category_id`` and polygon ``segmentation``."""
    from PIL import Image, ImageDraw
    base = Image.fromarray(np.ascontiguousarray(image_rgb[:, :, :3])).convert("RGBA")
    over = Image.new("RGBA", base.size, (0, 0, 0, 0))
    d = ImageDraw.Draw(over)
    a = int(round(255 * alpha))
    for an in annotations:
        k = class_of_category.get(an.get("category_id"), 0)
        c = class_colour(k)
        seg = an.get("segmentation")
        if isinstance(seg, list):
            for poly in seg:
                if len(poly) >= 6:
                    d.polygon([(float(poly[j]), float(poly[j + 1])) for j in range(0, len(poly) - 1, 2)], fill=c + (a,), outline=c + (255,))
    out = Image.alpha_composite(base, over).convert("RGB")
    d = ImageDraw.Draw(out)
    for an in annotations:
        k = class_of_category.get(an.get("category_id"), 0)
        c = class_colour(k)
        if an.get("bbox") is not None:
            x, y, w, h = [float(v) for v in an["bbox"]]
            d.rectangle([x, y, x + w, y + h], outline=c)
            name = class_names[k] if class_names is not None and k < len(class_names) else str(k)
            _label(d, (x + 1, y + 1), name, c)
    return out
