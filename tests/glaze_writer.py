"""Test utility: writes `.glaze` V1 files (python lzma / xxhash / PIL) so that the C++ reader can be exercised on
arbitrary content, the way the reference's parser tests round-trip through its own Serializer
(lib/src/parser/v1.rs:1400-1748).  Byte layout: SURVEY Appendix B (lib/src/parser/mod.rs:12-13, v1.rs:21-37,
:135-195, :426-449, :459-609, record encoders :613-1061).
"""
import io
import lzma
import struct

import numpy as np
import xxhash

SEED = 0x368262AAA1DEB64D
IDS = {"vertex": 0, "mesh": 1, "camera": 2, "texture": 3, "material": 4, "transform": 5, "instance": 6, "light": 7, "meta": 250}


def _h(b):
    return struct.pack("<Q", xxhash.xxh64(b, seed=SEED).intdigest())


def _xz(payload, preset=9, check=lzma.CHECK_CRC64):
    return lzma.compress(payload, format=lzma.FORMAT_XZ, check=check, preset=preset)


def _dynamic(items):
    out = struct.pack("<H", len(items))
    for it in items:
        out += struct.pack("<I", len(it)) + it
    return out


def png_bytes(arr):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr, "L" if arr.ndim == 2 else "RGBA").save(buf, format="PNG")
    return buf.getvalue()


def encode_material(m):
    e = m.get("emissive") or (0, 0, 0)
    return (bytes([m["mtype"], m["metal"]]) + bytes(m["diffuse_mul"]) + bytes(e) +
            struct.pack("<4f", m["ior"], m["roughness_mul"], m["metalness_mul"], m["anisotropy"]) +
            struct.pack("<5H", m["diffuse"], m["roughness"], m["metalness"], m["normal"], m["opacity"]) + m["name"].encode("utf8"))


def encode_light(l):
    return (bytes([l["ltype"]]) + struct.pack("<3f", *l["position"]) + struct.pack("<3f", *l["direction"]) +
            struct.pack("<I", l["resource_id"]) + struct.pack("<4f", l["intensity"], l["yaw"], l["pitch"], l["roll"]) +
            np.asarray(l["color"], "<f4").tobytes() + l["name"].encode("utf8"))


def encode_camera(c):
    return bytes([c["type"]]) + struct.pack("<12f", *c["position"], *c["target"], *c["up"], c["fovx_or_scale"], c["near"], c["far"])


def encode_texture(t):
    name = t["name"].encode("utf8")
    out = bytes([t["format"], len(name)]) + name + bytes([len(t["levels"])])
    for lvl in t["levels"]:
        p = png_bytes(lvl)
        out += struct.pack("<I", len(p)) + p
    return out


def write_glaze(path, vertices=None, meshes=None, cameras=None, textures=None, materials=None, transforms=None, instances=None,
                lights=None, meta=None, preset=6, check=lzma.CHECK_CRC64, extra_chunks=None, version=1):
    """All arguments optional (a chunk that would be empty is not written, like OffsetsTable::set_offset, v1.rs:188-194)."""
    chunks = []
    if vertices is not None and len(vertices):
        chunks.append(("vertex", _xz(np.asarray(vertices, "<f4").reshape(-1, 8).tobytes(), preset, check)))
    if meshes:
        items = [struct.pack("<HIH", m["id"], len(m["indices"]), m["material"]) + np.asarray(m["indices"], "<u4").tobytes() for m in meshes]
        chunks.append(("mesh", _xz(_dynamic(items), preset, check)))
    if cameras:
        chunks.append(("camera", _xz(b"".join(encode_camera(c) for c in cameras), preset, check)))
    if textures:
        chunks.append(("texture", _dynamic([encode_texture(t) for t in textures])))      # not xz: PNG inside
    if materials:
        chunks.append(("material", _xz(_dynamic([encode_material(m) for m in materials]), preset, check)))
    if transforms is not None and len(transforms):
        chunks.append(("transform", _xz(np.asarray(transforms, "<f4").reshape(-1, 16).tobytes(), preset, check)))
    if instances is not None and len(instances):
        chunks.append(("instance", _xz(np.asarray(instances, "<u2").reshape(-1, 2).tobytes(), preset, check)))
    if lights:
        chunks.append(("light", _xz(_dynamic([encode_light(l) for l in lights]), preset, check)))
    if meta:
        chunks.append(("meta", _xz(struct.pack("<5f", *meta["scene_centre"], meta["scene_radius"], meta["exposure"]), preset, check)))
    bodies = [(IDS[name], _h(body) + body) for name, body in chunks]
    for cid, body in (extra_chunks or []):
        bodies.append((cid, _h(body) + body))
    n = len(bodies)
    table_len = 8 + 1 + 17 * n
    off = 16 + table_len
    table = bytes([n])
    for cid, body in bodies:
        table += struct.pack("<BQQ", cid, off, len(body))
        off += len(body)
    data = b"glaze" + bytes([version]) + bytes(10) + _h(table) + table + b"".join(b for _, b in bodies)
    with open(path, "wb") as f:
        f.write(data)
    return data
