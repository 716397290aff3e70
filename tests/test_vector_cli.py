"""Host tail of the detector step: mask polygonisation, RDP, GeoPackage writer (CPU) and the
``make_detections`` CLI end to end on a small synthetic tileset (GPU)."""
import json
import os
import sys
import sqlite3

import numpy as np
import pytest
from scipy import ndimage

from proj_roadsurf_amd.gpkg import read_gpkg, write_gpkg
from oracle.host_tail_oracle import instances_to_features as oracle_features, mask_to_polygons, rdp, ring_area
from proj_roadsurf_amd.vectorize import instances_to_features, vectorize_masks_native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _area(polys):
    return -sum(sum(ring_area(r) for r in p) for p in polys)      # rasterio's ring direction: exteriors negative, holes positive


def test_polygonize_rect_with_hole():
    m = np.zeros((6, 7), bool)
    m[1:5, 1:6] = True
    m[2:4, 3] = False
    P = mask_to_polygons(m)
    assert len(P) == 1 and len(P[0]) == 2
    assert P[0][0][0] == P[0][0][-1] and len(P[0][0]) == 5          # closed rectangle, corner vertices only
    assert ring_area(P[0][0]) == -20 and ring_area(P[0][1]) == 2 and _area(P) == m.sum()


def test_polygonize_follows_the_documented_rasterio_conventions():
    """rasterio.features.shapes is absent (parity unpinned); what its documentation states is pinned here:
    * ring direction and start: the example of topics/features -- one pixel at column 71, row 6 ->
      [(71, 6), (71, 7), (72, 7), (72, 6), (71, 6)] (recalled from the documentation, like the detectron2 vectors of test_train_oracle);
    * connectivity 4 by default: foreground pixels touching only at a corner are separate polygons (next test);
    * an island inside a hole is its own polygon, not a ring of the enclosing one (GeoJSON polygon = exterior + its holes only);
    * the native vectoriser (rs_vectorize_masks) returns the same vertices in the same order."""
    m = np.zeros((10, 80), bool)
    m[6, 71] = True
    P = mask_to_polygons(m)
    assert P == [[[(71.0, 6.0), (71.0, 7.0), (72.0, 7.0), (72.0, 6.0), (71.0, 6.0)]]]
    _native_lib()
    got = vectorize_masks_native(np.packbits(m[None], axis=2, bitorder="little"), 10, 80, 0.0, 1)
    assert got == [[[[(71.0, 6.0), (71.0, 7.0), (72.0, 7.0), (72.0, 6.0), (71.0, 6.0)]]]]
    # a 5x5 frame (hole 3x3) with a 1-pixel island in the middle of the hole: two polygons, the frame with ONE hole
    f = np.zeros((7, 7), bool)
    f[1:6, 1:6] = True
    f[2:5, 2:5] = False
    f[3, 3] = True
    P = mask_to_polygons(f)
    assert sorted(len(p) for p in P) == [1, 2]
    frame = next(p for p in P if len(p) == 2)
    island = next(p for p in P if len(p) == 1)
    assert -ring_area(frame[0]) == 25 and ring_area(frame[1]) == 9 and -ring_area(island[0]) == 1
    assert frame[0][0] == (1.0, 1.0) and frame[0][1] == (1.0, 6.0)          # exterior: top-left corner first, then down
    # a background pixel enclosed except for a CORNER contact with the outside: with 4-connected foreground the background is
    # 8-connected, so that pixel belongs to the outside -- no hole, ONE exterior ring that passes twice through the shared vertex
    # (our reading of connectivity 4; GDAL's handling of such pinch points is not documented: parity unpinned)
    g = np.ones((4, 4), bool)
    g[1, 1] = False
    g[0, 0] = False
    P = mask_to_polygons(g)
    assert len(P) == 1 and len(P[0]) == 1 and _area(P) == 14 and P[0][0].count((1.0, 1.0)) == 2
    assert vectorize_masks_native(np.packbits(np.stack([f, np.pad(g, ((0, 3), (0, 3)))]), axis=2, bitorder="little"), 7, 7, 0.0, 1) == \
        [[[[(float(x), float(y)) for x, y in r] for r in p] for p in mask_to_polygons(f)],
         [[[(float(x), float(y)) for x, y in r] for r in p] for p in mask_to_polygons(np.pad(g, ((0, 3), (0, 3))))]]


def test_polygonize_diagonal_pixels_are_separate_regions():
    m = np.zeros((3, 3), bool)
    m[0, 0] = m[1, 1] = m[2, 2] = True
    assert len(mask_to_polygons(m)) == 3                              # 4-connectivity (rasterio default)


def test_polygonize_random_masks_area_and_components():
    rng = np.random.default_rng(1)
    for _ in range(100):
        m = rng.random((15, 17)) > rng.uniform(0.3, 0.7)
        P = mask_to_polygons(m)
        assert abs(_area(P) - m.sum()) < 1e-9
        assert len(P) == ndimage.label(m)[1]
        holes = ndimage.label(~np.pad(m, 1))[1] - 1                   # background regions not touching the border... (8-conn for holes)
        assert sum(len(p) - 1 for p in P) >= 0 and holes >= 0
    assert mask_to_polygons(np.zeros((4, 4), bool)) == []


def test_rdp_known_answers():
    line = [(0, 0), (1, 0.1), (2, -0.1), (3, 5), (4, 6), (5, 7), (6, 8.1), (7, 9), (8, 9), (9, 9)]
    assert rdp(line, 1.0) == [(0.0, 0.0), (2.0, -0.1), (3.0, 5.0), (7.0, 9.0), (9.0, 9.0)]
    assert rdp(line, 0.0) == [tuple(map(float, p)) for p in line]
    sq = [(0, 0), (4, 0), (4, 4), (0, 4), (0, 0)]                      # closed ring: chord degenerate -> distance to start
    assert rdp(sq, 0.75) == [tuple(map(float, p)) for p in sq]
    stair = [(0, 0), (1, 0), (1, 1), (2, 1), (2, 2), (3, 2), (3, 3)]    # 0.5-px staircase collapses at eps 0.75
    assert rdp(stair, 0.75) == [(0.0, 0.0), (3.0, 3.0)]


class _Inst:
    image_size = (8, 8)

    def __init__(self):
        self.pred_masks = np.zeros((1, 8, 8), bool)
        self.pred_masks[0, 2:6, 1:7] = True
        self.scores = np.array([0.9], np.float32)
        self.pred_classes = np.array([1])
        self.pred_boxes = np.array([[1, 2, 7, 6]], np.float32)

    def __len__(self):
        return 1

    def has(self, k):
        return True


def _native_lib():
    from proj_roadsurf_amd.engine import LIB_PATH
    if not os.path.exists(LIB_PATH):
        import __graft_entry__ as g
        g.build()


def _restated(masks, eps):
    out = []
    for m in masks:
        polys = []
        for poly in mask_to_polygons(m):
            rings = []
            for r in poly:
                rr = rdp(r, eps) if eps > 0 else list(r)
                if len(rr) < 4:
                    rr = list(r)
                rings.append([(float(x), float(y)) for x, y in rr])
            polys.append(rings)
        out.append(polys)
    return out


@pytest.mark.parametrize("shape,eps", [((64, 64), 0.0), ((64, 64), 0.75), ((50, 77), 0.75), ((128, 100), 2.0)])
def test_native_vectorizer_equals_restatement_vertex_for_vertex(shape, eps):
    """rs_vectorize_masks (C++, multi-threaded, on the engine's bit-packed masks) == mask_to_polygons + rdp of this
    module: same polygons, same ring order, same start vertices, identical float64 coordinates.  Cases: smooth
    blobs, salt-and-pepper noise (many holes and diagonal contacts), empty and full masks, widths that are not a
    multiple of 8."""
    _native_lib()
    h, w = shape
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    masks = []
    for i in range(6):                                    # smooth blobs with holes
        f = ndimage.gaussian_filter(rng.random((h, w)), 2.5 + i)
        masks.append(f > np.quantile(f, 0.55))
    for dens in (0.3, 0.5, 0.7):                          # noise: diagonal touches, 1-pixel holes/islands
        masks.append(rng.random((h, w)) < dens)
    chk = (np.add.outer(np.arange(h), np.arange(w)) % 2).astype(bool)
    masks += [chk, ~chk, np.zeros((h, w), bool), np.ones((h, w), bool)]
    ring = np.zeros((h, w), bool); ring[5:40, 5:45] = True; ring[10:30, 10:40] = False; ring[15:25, 15:35] = True
    masks.append(ring)                                    # island inside a hole
    masks = np.stack(masks)
    packed = np.packbits(masks, axis=2, bitorder="little")
    want = _restated(masks, eps)
    for threads in (1, 4):
        got = vectorize_masks_native(packed, h, w, eps, threads)
        assert len(got) == len(want)
        for gi, wi in zip(got, want):
            assert gi == wi


def test_native_vectorizer_in_features_path():
    _native_lib()
    from proj_roadsurf_amd.engine import Instances
    rng = np.random.default_rng(5)
    h, w, n = 96, 120, 7
    masks = np.stack([ndimage.gaussian_filter(rng.random((h, w)), 4) > 0.5 for _ in range(n)])
    inst = Instances((h, w), rng.random((n, 4)).astype(np.float32), rng.random(n).astype(np.float32), np.arange(n) % 2,
                     np.packbits(masks, axis=2, bitorder="little"), None)
    a = instances_to_features(inst, "t.tif", (0.0, 0.0, 120.0, 96.0), True, 0.75)
    b = oracle_features(inst, "t.tif", (0.0, 0.0, 120.0, 96.0), True, 0.75)
    assert a == b and len(a) >= n


def test_native_gpkg_rows_equal_python_writer(tmp_path):
    """instances_to_gpkg_rows (vectorise + RDP + georeference + GeoPackage blob, all C++) produces byte for byte the blobs that
    gpkg_geom builds from instances_to_features, and GpkgWriter's file reads back as the same features."""
    _native_lib()
    from proj_roadsurf_amd.engine import Instances
    from proj_roadsurf_amd.gpkg import GpkgWriter, gpkg_geom
    from proj_roadsurf_amd.vectorize import instances_to_gpkg_rows
    rng = np.random.default_rng(11)
    h, w, n = 96, 120, 9
    masks = np.stack([ndimage.gaussian_filter(rng.random((h, w)), 3 + i % 3) > 0.5 for i in range(n)])
    masks[3] = False                                               # an empty mask: no polygon, no row
    inst = Instances((h, w), rng.random((n, 4)).astype(np.float32), rng.random(n).astype(np.float32), np.arange(n) % 2,
                     np.packbits(masks, axis=2, bitorder="little"), None)
    for extent, srs in [((2600000.0, 1200000.0, 2600104.6, 1200083.7), 2056), (None, -1)]:
        feats = oracle_features(inst, "18_1_2.tif", extent, True, 0.75)
        rows, bbox = instances_to_gpkg_rows(inst, "18_1_2.tif", extent, True, 0.75, srs_id=srs)
        assert len(rows) == len(feats) > n - 1
        for f, (blob, score, cls, image) in zip(feats, rows):
            assert blob == gpkg_geom(f["geometry"]["coordinates"], srs)
            assert (score, cls, image) == (f["properties"]["score"], f["properties"]["det_class"], "18_1_2.tif")
        xs = [p[0] for f in feats for r in f["geometry"]["coordinates"] for p in r]
        ys = [p[1] for f in feats for r in f["geometry"]["coordinates"] for p in r]
        assert bbox == [min(xs), min(ys), max(xs), max(ys)]
    p = str(tmp_path / "rows.gpkg")
    gw = GpkgWriter(p, table="t", epsg=2056)
    gw.add_rows(rows[:0], None)
    rows2, bbox2 = instances_to_gpkg_rows(inst, "18_1_2.tif", (2600000.0, 1200000.0, 2600104.6, 1200083.7), True, 0.75, srs_id=2056)
    gw.add_rows(rows2, bbox2)
    assert gw.close() == len(rows2)
    back = read_gpkg(p, "t")
    want = oracle_features(inst, "18_1_2.tif", (2600000.0, 1200000.0, 2600104.6, 1200083.7), True, 0.75)
    assert [b["geometry"]["coordinates"] for b in back] == [f["geometry"]["coordinates"] for f in want] and back[0]["srs_id"] == 2056
    empty = Instances((h, w), np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int64), np.zeros((0, h, (w + 7) // 8), np.uint8), None)
    assert instances_to_gpkg_rows(empty, "x.tif") == ([], None)


def test_features_georeference_and_gpkg_roundtrip(tmp_path):
    feats = instances_to_features(_Inst(), "18_1_2.tif", extent=(1000.0, 2000.0, 1080.0, 2080.0), rdp_enabled=True, rdp_epsilon=0.75)
    assert len(feats) == 1
    ring = feats[0]["geometry"]["coordinates"][0]
    xs, ys = [p[0] for p in ring], [p[1] for p in ring]
    assert (min(xs), max(xs)) == (1010.0, 1070.0) and (min(ys), max(ys)) == (2020.0, 2060.0)      # 10 CRS units per pixel, y flipped
    assert feats[0]["properties"] == {"score": pytest.approx(0.9), "det_class": 1, "image": "18_1_2.tif"}
    p = str(tmp_path / "val_detections_at_0dot05_threshold.gpkg")
    assert write_gpkg(p, feats, table="val_detections", epsg=3857) == 1
    back = read_gpkg(p, "val_detections")
    assert back[0]["geometry"]["coordinates"] == feats[0]["geometry"]["coordinates"] and back[0]["srs_id"] == 3857
    con = sqlite3.connect(p)
    assert con.execute("PRAGMA application_id").fetchone()[0] == 0x47504B47
    assert con.execute("SELECT geometry_type_name, srs_id FROM gpkg_geometry_columns").fetchone() == ("POLYGON", 3857)
    assert con.execute("SELECT data_type FROM gpkg_contents").fetchone()[0] == "features"
    con.close()


def test_gpkg_shards_appended_with_attach_equal_one_writer(tmp_path):
    """Per-rank GeoPackage shards merged on the host: rows of three shards appended in rank order with SQLite ATTACH == the same rows
    written by one writer, row for row (fid order), bounding box and feature count included; an empty shard in the middle changes nothing."""
    from proj_roadsurf_amd.gpkg import GpkgWriter, gpkg_geom
    rng = np.random.default_rng(3)
    tab = "val_detections_at_0dot05_threshold"

    def rows_of(k, n):
        out, bx = [], [np.inf, np.inf, -np.inf, -np.inf]
        for i in range(n):
            x, y = float(rng.uniform(0, 5000)), float(rng.uniform(0, 5000))
            ring = [[x, y], [x + 3.0, y], [x + 3.0, y + 2.0], [x, y + 2.0], [x, y]]
            out.append((gpkg_geom([ring], 3857), float(rng.uniform(0.05, 1.0)), int(rng.integers(0, 2)), f"18_{k}_{i}.tif"))
            bx = [min(bx[0], x), min(bx[1], y), max(bx[2], x + 3.0), max(bx[3], y + 2.0)]
        return out, (bx if n else None)

    shards = [rows_of(k, n) for k, n in enumerate((7, 0, 12, 5))]
    one = GpkgWriter(str(tmp_path / "one.gpkg"), table=tab, epsg=3857)
    for rows, bx in shards:
        one.add_rows(rows, bx)
    assert one.close() == 24
    paths = []
    for k, (rows, bx) in enumerate(shards):
        pth = str(tmp_path / ("merged.gpkg" if k == 0 else f"merged.rank{k}.gpkg"))
        w = GpkgWriter(pth, table=tab, epsg=3857)
        w.add_rows(rows, bx)
        paths.append((pth, w))
    for pth, w in paths[1:]:
        w.close()
    head = paths[0][1]
    assert [head.append_shard(pth) for pth, _ in paths[1:]] == [0, 12, 5]
    assert head.close() == 24
    q = f'SELECT fid, geom, score, det_class, image FROM "{tab}" ORDER BY fid'
    a, b = sqlite3.connect(str(tmp_path / "one.gpkg")), sqlite3.connect(str(tmp_path / "merged.gpkg"))
    try:
        assert a.execute(q).fetchall() == b.execute(q).fetchall()
        cq = "SELECT table_name, data_type, min_x, min_y, max_x, max_y, srs_id FROM gpkg_contents"
        assert a.execute(cq).fetchall() == b.execute(cq).fetchall()
    finally:
        a.close(); b.close()
    with pytest.raises(ValueError):                      # a shard of another SRS is refused, not silently mixed in
        w2 = GpkgWriter(str(tmp_path / "other.gpkg"), table=tab, epsg=2056)
        w2.close()
        w3 = GpkgWriter(str(tmp_path / "m2.gpkg"), table=tab, epsg=3857)
        w3.append_shard(str(tmp_path / "other.gpkg"))


def _cli_dataset(tmp_path, n_tiles, tile=128):
    """Working directory of the reference CLI (R:config/config_obj_detec.yaml:74-90) with n synthetic TIFF tiles; returns (config path, wd)."""
    from PIL import Image
    import yaml
    from tests.util import synthetic_tiles
    wd = tmp_path / "outputs" / "obj_detector"
    (wd / "val-images").mkdir(parents=True)
    tiles = synthetic_tiles(n_tiles, tile, tile, 3, seed=9)
    images, meta = [], {}
    for i in range(n_tiles):
        fn = f"val-images/18_{100 + i}_200.tif"
        Image.fromarray(tiles[i][:, :, ::-1]).save(str(wd / fn))           # tiles are BGR; files hold RGB
        images.append({"id": i, "file_name": fn, "width": tile, "height": tile})
        meta[fn] = {"extent": [1000.0 * i, 0.0, 1000.0 * i + 52.0, 52.0], "crs": "EPSG:3857"}
    cats = [{"id": 1, "name": "artificial"}, {"id": 2, "name": "natural"}]
    json.dump({"images": images, "annotations": [], "categories": cats}, open(wd / "COCO_val.json", "w"))
    json.dump(meta, open(wd / "img_metadata.json", "w"))
    d2 = {"INPUT": {"FORMAT": "RGB", "MIN_SIZE_TEST": 192, "MAX_SIZE_TEST": 320},
          "MODEL": {"RPN": {"PRE_NMS_TOPK_TEST": 200, "POST_NMS_TOPK_TEST": 200}, "ROI_HEADS": {"NUM_CLASSES": 1}},
          "TEST": {"DETECTIONS_PER_IMAGE": 20}}
    yaml.safe_dump(d2, open(tmp_path / "d2.yaml", "w"))
    cfg = {"make_detections.py": {"working_directory": str(wd), "log_subfolder": "logs", "image_metadata_json": "img_metadata.json", "COCO_files": {"val": "COCO_val.json"},
                                  "detectron2_config_file": str(tmp_path / "d2.yaml"), "model_weights": {"pth_file": "logs/model_0005999.pth"},
                                  "rdp_simplification": {"enabled": True, "epsilon": 0.75}, "score_lower_threshold": 0.05}}
    yaml.safe_dump(cfg, open(tmp_path / "config.yaml", "w"))
    return str(tmp_path / "config.yaml"), wd


@pytest.mark.gpu
def test_make_detections_two_ranks_write_the_one_rank_geopackage(gpu_required, tmp_path):
    """BASELINE configs[2] in small: the CLI itself under `torch.distributed.run --nproc-per-node 2` (both ranks on the one GPU of the box, gloo rendezvous) on
    41 TIFF tiles -- rank 0 takes 21, rank 1 takes 20, a ragged last batch on each.  Each rank writes its own GeoPackage shard and rank 0 appends the other's
    with SQLite ATTACH: the merged file equals the one-rank run row for row (geometry blob, score, class, image, order), and so does the GeoJSON."""
    import subprocess
    cfg, wd = _cli_dataset(tmp_path, 41)
    name = "val_detections_at_0dot05_threshold"
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), RS_DIST_BACKEND="gloo")
    common = ["-m", "proj_roadsurf_amd.make_detections", cfg, "--synthetic-weights", "--batch", "4", "--geojson", "--tagged-samples", "0"]
    r1 = subprocess.run([sys.executable] + common, env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-3000:]
    os.rename(wd / f"{name}.gpkg", wd / "one_rank.gpkg")
    os.rename(wd / f"{name}.geojson", wd / "one_rank.geojson")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631"] + common,
                        env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert not [f for f in os.listdir(wd) if ".rank" in f], "shard files left behind"
    q = f'SELECT fid, geom, score, det_class, image FROM "{name}" ORDER BY fid'
    a, b = sqlite3.connect(str(wd / "one_rank.gpkg")), sqlite3.connect(str(wd / f"{name}.gpkg"))
    try:
        ra, rb = a.execute(q).fetchall(), b.execute(q).fetchall()
        assert len(ra) > 41 and ra == rb, f"{len(ra)} rows against {len(rb)}"
        cq = "SELECT min_x, min_y, max_x, max_y, srs_id FROM gpkg_contents"
        assert a.execute(cq).fetchall() == b.execute(cq).fetchall()
    finally:
        a.close(); b.close()
    assert json.load(open(wd / "one_rank.geojson")) == json.load(open(wd / f"{name}.geojson"))
    assert "rank 1 of 2 ranks on device 0" in r2.stderr


@pytest.mark.gpu
def test_make_detections_cli_end_to_end(gpu_required, tmp_path):
    """Same argv and YAML keys as the reference CLI (R:config/config_obj_detec.yaml:74-90), synthetic tiles and weights."""
    from PIL import Image
    import yaml
    from proj_roadsurf_amd import make_detections
    from proj_roadsurf_amd.spec import EngineSpec
    from tests.util import synthetic_tiles

    wd = tmp_path / "outputs" / "obj_detector"
    (wd / "val-images").mkdir(parents=True)
    tiles = synthetic_tiles(5, 128, 128, 3, seed=9)
    images, meta = [], {}
    for i in range(5):
        fn = f"val-images/18_{100 + i}_200.tif"
        Image.fromarray(tiles[i][:, :, ::-1]).save(str(wd / fn))           # tiles are BGR; files hold RGB
        images.append({"id": i, "file_name": fn, "width": 128, "height": 128})
        meta[fn] = {"extent": [1000.0 * i, 0.0, 1000.0 * i + 52.0, 52.0], "crs": "EPSG:3857"}
    cats = [{"id": 1, "name": "artificial"}, {"id": 2, "name": "natural"}]
    json.dump({"images": images, "annotations": [], "categories": cats}, open(wd / "COCO_val.json", "w"))
    json.dump(meta, open(wd / "img_metadata.json", "w"))
    # a small detectron2 YAML (the reference's keys, smaller sizes so the test is quick)
    d2 = {"INPUT": {"FORMAT": "RGB", "MIN_SIZE_TEST": 192, "MAX_SIZE_TEST": 320},
          "MODEL": {"RPN": {"PRE_NMS_TOPK_TEST": 200, "POST_NMS_TOPK_TEST": 200}, "ROI_HEADS": {"NUM_CLASSES": 1}},
          "TEST": {"DETECTIONS_PER_IMAGE": 20}}
    yaml.safe_dump(d2, open(tmp_path / "d2.yaml", "w"))
    cfg = {"make_detections.py": {"working_directory": str(wd), "log_subfolder": "logs", "sample_tagged_img_subfolder": "sample_detection_images",
                                  "image_metadata_json": "img_metadata.json", "COCO_files": {"val": "COCO_val.json"},
                                  "detectron2_config_file": str(tmp_path / "d2.yaml"), "model_weights": {"pth_file": "logs/model_0005999.pth"},
                                  "rdp_simplification": {"enabled": True, "epsilon": 0.75}, "score_lower_threshold": 0.05}}
    yaml.safe_dump(cfg, open(tmp_path / "config.yaml", "w"))
    cwd = os.getcwd()
    try:
        # the same job in the reference's arithmetic first (--precision fp32: fp32 activations / weights on the fp32 matrix cores) ...
        assert make_detections.main([str(tmp_path / "config.yaml"), "--synthetic-weights", "--batch", "2", "--precision", "fp32"]) == 0
        os.chdir(cwd)
        ref_feats = read_gpkg(str(wd / "val_detections_at_0dot05_threshold.gpkg"), "val_detections_at_0dot05_threshold")
        assert len(ref_feats) > 0
        # ... the same job with split operands (the reference's fp32 results on the fp16 matrix cores): the detections of the fp32 run, feature for feature
        assert make_detections.main([str(tmp_path / "config.yaml"), "--synthetic-weights", "--batch", "2", "--precision", "split"]) == 0
        os.chdir(cwd)
        split_feats = read_gpkg(str(wd / "val_detections_at_0dot05_threshold.gpkg"), "val_detections_at_0dot05_threshold")
        assert len(split_feats) == len(ref_feats)
        for a, b in zip(split_feats, ref_feats):
            assert a["properties"]["image"] == b["properties"]["image"] and a["properties"]["det_class"] == b["properties"]["det_class"]
            assert abs(a["properties"]["score"] - b["properties"]["score"]) <= 1e-4
        same = sum(a["geometry"]["coordinates"] == b["geometry"]["coordinates"] for a, b in zip(split_feats, ref_feats))
        assert same >= 0.98 * len(ref_feats), f"{same} of {len(ref_feats)} polygons identical"
        # ... then the production mode, which overwrites the outputs
        assert make_detections.main([str(tmp_path / "config.yaml"), "--synthetic-weights", "--batch", "2", "--geojson"]) == 0
    finally:
        os.chdir(cwd)
    out = wd / "val_detections_at_0dot05_threshold.gpkg"
    assert out.exists() and (wd / "logs").is_dir()
    feats = read_gpkg(str(out), "val_detections_at_0dot05_threshold")
    assert len(feats) > 0
    assert 0.5 * len(ref_feats) <= len(feats) <= 2.0 * len(ref_feats)        # random weights: the two precisions agree statistically only
    for f in feats:
        assert 0.05 < f["properties"]["score"] <= 1.0 and f["properties"]["det_class"] in (0, 1)
        i = int(f["properties"]["image"].split("_")[1]) - 100
        xs = [p[0] for p in f["geometry"]["coordinates"][0]]
        assert 1000.0 * i - 1e-6 <= min(xs) and max(xs) <= 1000.0 * i + 52.0 + 1e-6       # georeferenced into its own tile
    gj = json.load(open(wd / "val_detections_at_0dot05_threshold.geojson"))
    assert len(gj["features"]) == len(feats)
    # tagged previews of the first images (sample_tagged_img_subfolder, R:config/config_obj_detec.yaml:77)
    pngs = sorted(os.listdir(wd / "sample_detection_images"))
    assert pngs and all(p.startswith("val_det_") and p.endswith(".png") for p in pngs)
    from PIL import Image
    assert Image.open(wd / "sample_detection_images" / pngs[0]).size == (128, 128) or Image.open(wd / "sample_detection_images" / pngs[0]).size[0] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("coco_sizes", ["true", "absent", "wrong"])
def test_make_detections_cli_with_one_odd_sized_tile(gpu_required, tmp_path, coco_sizes):
    """A dataset with one tile of another shape at the DEFAULT --decode-procs: the reference reads tile by tile (cv2.imread) and
    copes with any mix.  The process decoder serves one shape per dataset, so the CLI decides up front from the COCO sizes
    ("true"), from the file headers when the COCO entries carry none ("absent"), and falls back to thread decoding when a worker
    meets a tile the COCO sizes did not announce ("wrong").  Every tile must come out in all three cases."""
    from PIL import Image
    import yaml
    from proj_roadsurf_amd import make_detections
    from tests.util import synthetic_tiles

    wd = tmp_path / "outputs" / "obj_detector"
    (wd / "val-images").mkdir(parents=True)
    tiles = synthetic_tiles(6, 128, 128, 3, seed=9)
    images = []
    for i in range(6):
        fn = f"val-images/18_{100 + i}_200.tif"
        t = tiles[i][:96] if i == 3 else tiles[i]                          # tile 3 is 96 x 128
        Image.fromarray(t[:, :, ::-1]).save(str(wd / fn))
        e = {"id": i, "file_name": fn}
        if coco_sizes == "true":
            e.update(width=128, height=int(t.shape[0]))
        elif coco_sizes == "wrong":
            e.update(width=128, height=128)
        images.append(e)
    json.dump({"images": images, "annotations": [], "categories": [{"id": 1, "name": "artificial"}, {"id": 2, "name": "natural"}]},
              open(wd / "COCO_val.json", "w"))
    d2 = {"INPUT": {"FORMAT": "RGB", "MIN_SIZE_TEST": 192, "MAX_SIZE_TEST": 320},
          "MODEL": {"RPN": {"PRE_NMS_TOPK_TEST": 200, "POST_NMS_TOPK_TEST": 200}, "ROI_HEADS": {"NUM_CLASSES": 1}},
          "TEST": {"DETECTIONS_PER_IMAGE": 20}}
    yaml.safe_dump(d2, open(tmp_path / "d2.yaml", "w"))
    cfg = {"make_detections.py": {"working_directory": str(wd), "log_subfolder": "logs",
                                  "COCO_files": {"val": "COCO_val.json"}, "detectron2_config_file": str(tmp_path / "d2.yaml"),
                                  "model_weights": {"pth_file": "logs/model_0005999.pth"}, "score_lower_threshold": 0.05}}
    yaml.safe_dump(cfg, open(tmp_path / "config.yaml", "w"))
    cwd = os.getcwd()
    try:
        assert make_detections.main([str(tmp_path / "config.yaml"), "--synthetic-weights", "--batch", "2", "--tagged-samples", "0"]) == 0
    finally:
        os.chdir(cwd)
    feats = read_gpkg(str(wd / "val_detections_at_0dot05_threshold.gpkg"), "val_detections_at_0dot05_threshold")
    seen = {f["properties"]["image"] for f in feats}
    assert seen == {f"18_{100 + i}_200.tif" for i in range(6)}, seen
    odd = [f for f in feats if f["properties"]["image"] == "18_103_200.tif"]
    assert max(p[1] for f in odd for p in f["geometry"]["coordinates"][0]) <= 96.0          # pixel coordinates of the 96-row tile
    from proj_roadsurf_amd.engine import _REGISTERED_HOST
    assert not _REGISTERED_HOST                                            # the slab was unpinned before its memory went away


def test_cropped_vectoriser_equals_full_canvas(lib_path_ok=None):
    """Masks as crops of their boxes (rs_mask_crops, what the streaming host interface copies back) give the same polygons,
    vertex for vertex and byte for byte in the GeoPackage blobs, as the full canvases: tracing, RDP and hole assignment only
    use differences of integer coordinates."""
    import numpy as np

    from proj_roadsurf_amd.engine import Instances
    from proj_roadsurf_amd.vectorize import instances_to_gpkg_rows
    rng = np.random.default_rng(3)
    h, w = 96, 101                       # partial last byte
    n = 6
    masks = np.zeros((n, h, w), bool)
    rects = np.zeros((n, 4), np.int32)
    for i in range(n):
        x0, y0 = int(rng.integers(0, 60)), int(rng.integers(0, 50))
        bw, bh = int(rng.integers(8, 40)), int(rng.integers(8, 40))
        blob = rng.random((bh, bw)) > 0.35
        blob[bh // 3: bh // 3 + 3, bw // 3: bw // 3 + 3] = False          # holes
        masks[i, y0:y0 + bh, x0:x0 + bw] = blob
        if i == n - 1:
            masks[i] = False; masks[i, :, w - 3:] = True                  # touches the partial last byte
            x0, y0, bw, bh = w - 3, 0, 3, h
        xb0, xb1 = x0 >> 3, min(w - 1, x0 + bw) >> 3
        rects[i] = (xb0, max(0, y0 - 1), xb1 - xb0 + 1, min(h, y0 + bh + 1) - max(0, y0 - 1))
    packed = np.packbits(masks, axis=2, bitorder="little")
    data, offs = [], []
    for i in range(n):
        xb0, yy, wb, rows = (int(v) for v in rects[i])
        offs.append(sum(len(d) for d in data))
        data.append(packed[i, yy:yy + rows, xb0:xb0 + wb].reshape(-1))
    boxes = np.zeros((n, 4), np.float32); scores = np.linspace(0.9, 0.5, n).astype(np.float32); classes = np.arange(n) % 2
    full = Instances((h, w), boxes, scores, classes, packed, None)
    crop = Instances((h, w), boxes, scores, classes, None, None, crops=(rects, np.array(offs, np.uint32), np.concatenate(data)))
    assert np.array_equal(crop._packed, packed) and np.array_equal(crop.pred_masks, masks)
    for eps in (0.0, 0.75):
        a = instances_to_gpkg_rows(full, "t.tif", (10.0, 20.0, 110.0, 120.0), True, eps, srs_id=2056, threads=2)
        b = instances_to_gpkg_rows(crop, "t.tif", (10.0, 20.0, 110.0, 120.0), True, eps, srs_id=2056, threads=2)
        assert a[1] == b[1] and len(a[0]) == len(b[0]) > n
        assert all(x == y for x, y in zip(a[0], b[0]))
