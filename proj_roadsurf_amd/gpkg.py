"""Minimal GeoPackage (OGC 12-128r15) writer/reader for polygon features, on the stdlib ``sqlite3``.

The reference's detector step writes ``<dataset>_detections_at_<thr>_threshold.gpkg`` with columns
``score``, ``det_class``, ``geometry`` (R:config/config_obj_detec.yaml:100-103,119-122; consumers:
R:scripts/road_segmentation/determine_class.py:22-25, final_metrics.py:214-221) through geopandas, which is
not available offline.  This writes the same table with the standard GeoPackage binary geometry
(GP header + little-endian WKB Polygon), so ``geopandas.read_file`` / GDAL can open it.
"""
from __future__ import annotations

import sqlite3
import struct
from typing import Iterable, List, Optional, Sequence, Tuple

_SRS = {
    3857: ("WGS 84 / Pseudo-Mercator", 'PROJCS["WGS 84 / Pseudo-Mercator",GEOGCS["WGS 84",DATUM["WGS_1984",SPHEROID["WGS 84",6378137,298.257223563]],PRIMEM["Greenwich",0],UNIT["degree",0.0174532925199433]],PROJECTION["Mercator_1SP"],PARAMETER["central_meridian",0],PARAMETER["scale_factor",1],PARAMETER["false_easting",0],PARAMETER["false_northing",0],UNIT["metre",1],AUTHORITY["EPSG","3857"]]'),
    2056: ("CH1903+ / LV95", 'PROJCS["CH1903+ / LV95",GEOGCS["CH1903+",DATUM["CH1903+",SPHEROID["Bessel 1841",6377397.155,299.1528128]],PRIMEM["Greenwich",0],UNIT["degree",0.0174532925199433]],PROJECTION["Hotine_Oblique_Mercator_Azimuth_Center"],PARAMETER["latitude_of_center",46.9524055555556],PARAMETER["longitude_of_center",7.43958333333333],PARAMETER["azimuth",90],PARAMETER["rectified_grid_angle",90],PARAMETER["scale_factor",1],PARAMETER["false_easting",2600000],PARAMETER["false_northing",1200000],UNIT["metre",1],AUTHORITY["EPSG","2056"]]'),
    4326: ("WGS 84", 'GEOGCS["WGS 84",DATUM["WGS_1984",SPHEROID["WGS 84",6378137,298.257223563]],PRIMEM["Greenwich",0],UNIT["degree",0.0174532925199433],AUTHORITY["EPSG","4326"]]'),
}


def polygon_wkb(rings: Sequence[Sequence[Sequence[float]]]) -> bytes:
    out = [struct.pack("<BII", 1, 3, len(rings))]
    for r in rings:
        out.append(struct.pack("<I", len(r)))
        for x, y in r:
            out.append(struct.pack("<dd", float(x), float(y)))
    return b"".join(out)


def gpkg_geom(rings: Sequence[Sequence[Sequence[float]]], srs_id: int) -> bytes:
    xs = [p[0] for r in rings for p in r]
    ys = [p[1] for r in rings for p in r]
    # magic 'GP', version 0, flags: little-endian (bit0) + envelope type 1 = [minx,maxx,miny,maxy] (bits 1-3)
    hdr = struct.pack("<2sBBi4d", b"GP", 0, 0b00000011, srs_id, min(xs), max(xs), min(ys), max(ys))
    return hdr + polygon_wkb(rings)


def _create_schema(cur: sqlite3.Cursor, table: str, srs_id: int) -> None:
    """The GeoPackage bookkeeping tables (OGC 12-128r15 1.1.2, 1.1.3, 2.1.5) and the feature table; one definition for both writers."""
    cur.execute("PRAGMA application_id = 1196444487")    # 'GPKG'
    cur.execute("PRAGMA user_version = 10200")
    cur.executescript("""
        DROP TABLE IF EXISTS gpkg_spatial_ref_sys; DROP TABLE IF EXISTS gpkg_contents; DROP TABLE IF EXISTS gpkg_geometry_columns;
        CREATE TABLE gpkg_spatial_ref_sys (srs_name TEXT NOT NULL, srs_id INTEGER NOT NULL PRIMARY KEY, organization TEXT NOT NULL,
            organization_coordsys_id INTEGER NOT NULL, definition TEXT NOT NULL, description TEXT);
        CREATE TABLE gpkg_contents (table_name TEXT NOT NULL PRIMARY KEY, data_type TEXT NOT NULL, identifier TEXT UNIQUE, description TEXT DEFAULT '',
            last_change DATETIME NOT NULL DEFAULT (strftime('%Y-%m-%dT%H:%M:%fZ','now')), min_x DOUBLE, min_y DOUBLE, max_x DOUBLE, max_y DOUBLE, srs_id INTEGER);
        CREATE TABLE gpkg_geometry_columns (table_name TEXT NOT NULL, column_name TEXT NOT NULL, geometry_type_name TEXT NOT NULL, srs_id INTEGER NOT NULL,
            z TINYINT NOT NULL, m TINYINT NOT NULL, CONSTRAINT pk_geom_cols PRIMARY KEY (table_name, column_name));
    """)
    cur.execute("INSERT INTO gpkg_spatial_ref_sys VALUES ('Undefined cartesian SRS', -1, 'NONE', -1, 'undefined', 'undefined cartesian coordinate reference system')")
    cur.execute("INSERT INTO gpkg_spatial_ref_sys VALUES ('Undefined geographic SRS', 0, 'NONE', 0, 'undefined', 'undefined geographic coordinate reference system')")
    name4326, def4326 = _SRS[4326]
    cur.execute("INSERT INTO gpkg_spatial_ref_sys VALUES (?, 4326, 'EPSG', 4326, ?, NULL)", (name4326, def4326))
    if srs_id not in (-1, 0, 4326):
        name, definition = _SRS.get(srs_id, (f"EPSG:{srs_id}", "undefined"))
        cur.execute("INSERT INTO gpkg_spatial_ref_sys VALUES (?, ?, 'EPSG', ?, ?, NULL)", (name, srs_id, srs_id, definition))
    cur.execute(f'DROP TABLE IF EXISTS "{table}"')
    cur.execute(f'CREATE TABLE "{table}" (fid INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, geom BLOB, score REAL, det_class INTEGER, image TEXT)')


def _finish_schema(con: sqlite3.Connection, table: str, srs_id: int, bx: Sequence[Optional[float]]) -> None:
    con.execute("INSERT INTO gpkg_contents (table_name, data_type, identifier, min_x, min_y, max_x, max_y, srs_id) VALUES (?, 'features', ?, ?, ?, ?, ?, ?)",
                (table, table, bx[0], bx[1], bx[2], bx[3], srs_id))
    con.execute("INSERT INTO gpkg_geometry_columns VALUES (?, 'geom', 'POLYGON', ?, 0, 0)", (table, srs_id))


def write_gpkg(path: str, features: Iterable[dict], table: str = "detections", epsg: Optional[int] = None) -> int:
    """Write GeoJSON-like polygon features (``properties``: score, det_class, image).  Returns the row count."""
    srs_id = int(epsg) if epsg else -1
    con = sqlite3.connect(path)
    try:
        cur = con.cursor()
        _create_schema(cur, table, srs_id)
        n = 0
        bx = [float("inf"), float("inf"), float("-inf"), float("-inf")]
        for f in features:
            rings = f["geometry"]["coordinates"]
            pr = f.get("properties", {})
            cur.execute(f'INSERT INTO "{table}" (geom, score, det_class, image) VALUES (?, ?, ?, ?)',
                        (gpkg_geom(rings, srs_id), pr.get("score"), pr.get("det_class"), pr.get("image")))
            for r in rings:
                for x, y in r:
                    bx[0] = min(bx[0], x); bx[1] = min(bx[1], y); bx[2] = max(bx[2], x); bx[3] = max(bx[3], y)
            n += 1
        _finish_schema(con, table, srs_id, bx if n else [None, None, None, None])
        con.commit()
        return n
    finally:
        con.close()


class GpkgWriter:
    """Streaming form of ``write_gpkg`` for rows that already carry the GeoPackage geometry blob (``rs_vec_gpkg_blobs``):
    ``add_rows([(blob, score, det_class, image), ...], bbox)`` per batch, ``close()`` writes the contents/geometry metadata.
    ``append_shard(path)`` copies the feature rows of another rank's GeoPackage of the same table behind the rows written so far
    (SQLite ``ATTACH``: the rows never pass through Python) -- the host-side merge of per-rank shards, SURVEY.md section 8e."""

    def __init__(self, path: str, table: str = "detections", epsg: Optional[int] = None):
        self.table, self.srs_id = table, int(epsg) if epsg else -1
        self.con = sqlite3.connect(path)
        self.n = 0
        self.bx = [float("inf"), float("inf"), float("-inf"), float("-inf")]
        cur = self.con.cursor()
        cur.execute("PRAGMA journal_mode = OFF")
        cur.execute("PRAGMA synchronous = OFF")
        _create_schema(cur, table, self.srs_id)

    def add_rows(self, rows: Sequence[Tuple[bytes, float, int, str]], bbox: Optional[Sequence[float]]) -> None:
        if not rows:
            return
        self.con.executemany(f'INSERT INTO "{self.table}" (geom, score, det_class, image) VALUES (?, ?, ?, ?)', rows)
        self.n += len(rows)
        if bbox is not None:
            self.bx = [min(self.bx[0], bbox[0]), min(self.bx[1], bbox[1]), max(self.bx[2], bbox[2]), max(self.bx[3], bbox[3])]

    def append_shard(self, path: str) -> int:
        """Rows of ``path``'s table (same name and SRS, written by another GpkgWriter) appended in their fid order; returns their count."""
        self.con.commit()                                   # ATTACH is not allowed inside a transaction
        self.con.execute("ATTACH DATABASE ? AS shard", (path,))
        try:
            srs = self.con.execute("SELECT srs_id, min_x, min_y, max_x, max_y FROM shard.gpkg_contents WHERE table_name = ?", (self.table,)).fetchone()
            if srs is None:
                raise ValueError(f"{path}: no feature table {self.table!r}")
            if int(srs[0]) != self.srs_id:
                raise ValueError(f"{path}: SRS {srs[0]} differs from {self.srs_id}")
            cur = self.con.execute(f'INSERT INTO main."{self.table}" (geom, score, det_class, image) '
                                   f'SELECT geom, score, det_class, image FROM shard."{self.table}" ORDER BY fid')
            k = cur.rowcount
            self.con.commit()
        finally:
            self.con.execute("DETACH DATABASE shard")
        self.n += k
        if k and srs[1] is not None:
            self.bx = [min(self.bx[0], srs[1]), min(self.bx[1], srs[2]), max(self.bx[2], srs[3]), max(self.bx[3], srs[4])]
        return k

    def close(self) -> int:
        _finish_schema(self.con, self.table, self.srs_id, self.bx if self.n else [None, None, None, None])
        self.con.commit()
        self.con.close()
        return self.n


def read_gpkg(path: str, table: str = "detections") -> List[dict]:
    """Read back what ``write_gpkg`` wrote (used by the tests; a reader for consumers without geopandas)."""
    con = sqlite3.connect(path)
    try:
        rows = con.execute(f'SELECT geom, score, det_class, image FROM "{table}" ORDER BY fid').fetchall()
    finally:
        con.close()
    out = []
    for blob, score, cls, image in rows:
        magic, _ver, flags, srs = struct.unpack_from("<2sBBi", blob, 0)
        assert magic == b"GP" and flags & 1
        env = {0: 0, 1: 32, 2: 48, 3: 48, 4: 64}[(flags >> 1) & 7]
        off = 8 + env
        bo, typ, nr = struct.unpack_from("<BII", blob, off)
        assert bo == 1 and typ == 3
        off += 9
        rings = []
        for _ in range(nr):
            (npt,) = struct.unpack_from("<I", blob, off)
            off += 4
            pts = [list(struct.unpack_from("<dd", blob, off + 16 * i)) for i in range(npt)]
            off += 16 * npt
            rings.append(pts)
        out.append({"type": "Feature", "geometry": {"type": "Polygon", "coordinates": rings},
                    "properties": {"score": score, "det_class": cls, "image": image}, "srs_id": srs})
    return out
