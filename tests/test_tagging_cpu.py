"""Tagged sample images (proj_roadsurf_amd/tagging.py): host-only PIL rendering of predictions / ground truth for the
``sample_tagged_img_subfolder`` artefacts of the reference's config (R:config/config_obj_detec.yaml:65,77)."""
import numpy as np

from proj_roadsurf_amd.tagging import class_colour, draw_annotations, draw_instances


def test_draw_instances_marks_mask_box_and_keeps_the_rest():
    img = np.full((64, 80, 3), 200, np.uint8)
    masks = np.zeros((2, 64, 80), bool)
    masks[0, 10:30, 10:40] = True
    masks[1, 40:60, 50:70] = True
    boxes = np.array([[10, 10, 40, 30], [50, 40, 70, 60]], np.float32)
    out = np.asarray(draw_instances(img, boxes, np.array([0, 1]), np.array([0.9, 0.5]), masks, ["artificial", "natural"]))
    assert out.shape == (64, 80, 3) and out.dtype == np.uint8
    c0 = np.array(class_colour(0), np.float32)
    inside = out[25, 35].astype(np.float32)                      # mask interior away from label and outline
    assert np.allclose(inside, 0.6 * 200 + 0.4 * c0, atol=1.5)
    assert tuple(out[30, 10]) == class_colour(0) or tuple(out[10, 25]) == class_colour(0)    # box outline in the class colour
    assert tuple(out[5, 75]) == (200, 200, 200)                  # untouched background
    # no instances: the image comes back unchanged
    assert np.array_equal(np.asarray(draw_instances(img, np.zeros((0, 4)), np.zeros(0, int))), img)


def test_draw_annotations_fills_polygons():
    img = np.zeros((50, 50, 3), np.uint8)
    anns = [{"bbox": [10, 10, 20, 20], "category_id": 7, "segmentation": [[10, 10, 30, 10, 30, 30, 10, 30]]}]
    out = np.asarray(draw_annotations(img, anns, {7: 1}, ["a", "b"]))
    c = np.array(class_colour(1), np.float32)
    assert np.allclose(out[25, 25].astype(np.float32), 0.4 * c, atol=2.0)
    assert tuple(out[45, 45]) == (0, 0, 0)
