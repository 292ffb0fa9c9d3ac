"""Built-in scenes for the BASELINE.json workloads that have no file in the reference repository.

* cube_scene()   -- config 2: what the reference converter makes of resources/cube.obj
                    (converter/src/main.rs:330-372, :396-406, :545-605; counts pinned by its test
                    at :740-746: 24 vertices, 1 mesh/transform/instance/camera, 3 materials,
                    2 textures) plus the build-added omni light (the file has no light
                    and the reference would render black).
* atrium_scene() -- configs 4/5: synthetic stand-in for Sponza, which is not in the reference
                    repository (README.md:59-61).  Deterministic for a given seed.
"""
import numpy as np

from . import abi
from .scene_desc import (INSTANCE_DTYPE, MESH_DTYPE, VERTEX_DTYPE, SceneDesc, make_camera, make_light, make_material,
                         make_meta)


def checker_texture(size=512, squares=8, lo=40, hi=230):
    """Procedural stand-in for resources/checker.jpg (512x512, materials/texture.rs:313-316)."""
    y, x = np.mgrid[0:size, 0:size]
    c = (((x * squares) // size + (y * squares) // size) % 2).astype(np.uint8)
    g = np.where(c == 0, lo, hi).astype(np.uint8)
    return np.stack([g, g, g, np.full_like(g, 255)], axis=-1)


# resources/cube.obj restated as data: 8 corner positions, 14 uvs, 6 normals, 6 quads of (v/vt/vn), 1-based
_CUBE_V = [(1, 1, -1), (1, -1, -1), (1, 1, 1), (1, -1, 1), (-1, 1, -1), (-1, -1, -1), (-1, 1, 1), (-1, -1, 1)]
_CUBE_VT = [(0.625, 0.5), (0.875, 0.5), (0.875, 0.75), (0.625, 0.75), (0.375, 0.75), (0.625, 1.0), (0.375, 1.0),
            (0.375, 0.0), (0.625, 0.0), (0.625, 0.25), (0.375, 0.25), (0.125, 0.5), (0.375, 0.5), (0.125, 0.75)]
_CUBE_VN = [(0, 1, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0), (1, 0, 0), (0, 0, -1)]
_CUBE_F = [[(1, 1, 1), (5, 2, 1), (7, 3, 1), (3, 4, 1)], [(4, 5, 2), (3, 4, 2), (7, 6, 2), (8, 7, 2)],
           [(8, 8, 3), (7, 9, 3), (5, 10, 3), (6, 11, 3)], [(6, 12, 4), (2, 13, 4), (4, 5, 4), (8, 14, 4)],
           [(2, 13, 5), (1, 1, 5), (3, 4, 5), (4, 5, 5)], [(6, 11, 6), (5, 10, 6), (1, 1, 6), (2, 13, 6)]]


def cube_scene(light=True, material_type=abi.MAT_LAMBERT):
    verts, index_of, indices = [], {}, []
    for quad in _CUBE_F:
        for tri in ((0, 1, 2), (0, 2, 3)):          # assimp's fan triangulation of a quad
            for k in tri:
                v, vt, vn = quad[k]
                uv = _CUBE_VT[vt - 1]
                key = (_CUBE_V[v - 1], _CUBE_VN[vn - 1], (uv[0], 1.0 - uv[1]))   # converter/src/main.rs:354 flips v
                if key not in index_of:
                    index_of[key] = len(verts)
                    verts.append(key)
                indices.append(index_of[key])
    vertices = np.zeros(len(verts), VERTEX_DTYPE)
    for i, (p, n, t) in enumerate(verts):
        vertices[i] = (p, n, t)
    assert len(verts) == 24 and len(indices) == 36
    meshes = np.array([(0, 2, 0, 36)], MESH_DTYPE)    # material index + 1 (converter/src/main.rs:368)
    instances = np.array([(0, 0)], INSTANCE_DTYPE)
    materials = [make_material("default"),
                 make_material("DefaultMaterial"),
                 make_material("Material", mtype=material_type, diffuse_mul=(204, 204, 204), diffuse=1, ior=1.45)]
    textures = [(abi.TEX_RGBA_SRGB, np.full((1, 1, 4), 255, np.uint8), "default"),
                (abi.TEX_RGBA_SRGB, checker_texture(), "checker")]
    lights = [make_light(abi.LIGHT_OMNI, "build-added omni", position=(0.0, 0.5, 0.0), intensity=1.0)] if light else []
    camera = make_camera(position=(0, 0, 0), target=(0, 0, 100), up=(0, 1, 0), fovx=np.float32(np.radians(np.float32(90.0))),
                         near=1e-3, far=100.0)
    meta = make_meta(centre=(0, 0, 0), radius=float(np.float32(np.sqrt(3.0))), exposure=1.0)
    return SceneDesc(vertices, np.array(indices, np.uint32), meshes, None, instances, materials, lights, textures, camera, meta)


# ------------------------------------------------------------------------------------------------
# synthetic atrium
# ------------------------------------------------------------------------------------------------
class _Builder:
    def __init__(self):
        self.v, self.i, self.meshes = [], [], []
        self.nv = 0
        self.ni = 0

    def add(self, pos, nrm, uv, tri, material):
        pos = np.asarray(pos, np.float32).reshape(-1, 3)
        nrm = np.asarray(nrm, np.float32).reshape(-1, 3)
        uv = np.asarray(uv, np.float32).reshape(-1, 2)
        tri = np.asarray(tri, np.uint32).reshape(-1)
        block = np.zeros(pos.shape[0], VERTEX_DTYPE)
        block["vv"], block["vn"], block["vt"] = pos, nrm, uv
        self.v.append(block)
        self.i.append(tri + self.nv)
        self.meshes.append((len(self.meshes), material, self.ni, tri.size))
        self.nv += pos.shape[0]
        self.ni += tri.size

    def grid(self, origin, du, dv, nu, nv, material, uv_scale=1.0, displace=None):
        """(nu x nv)-cell grid spanning origin + s*du + t*dv; normal = normalize(du x dv)."""
        s, t = np.meshgrid(np.linspace(0, 1, nu + 1, dtype=np.float32), np.linspace(0, 1, nv + 1, dtype=np.float32), indexing="ij")
        origin, du, dv = (np.asarray(a, np.float32) for a in (origin, du, dv))
        pos = origin + s[..., None] * du + t[..., None] * dv
        n = np.cross(du, dv)
        n = n / np.linalg.norm(n)
        nrm = np.broadcast_to(n, pos.shape).copy()
        if displace is not None:
            pos, nrm = displace(pos, nrm, s, t)
        uv = np.stack([s * uv_scale, t * uv_scale], axis=-1)
        idx = np.arange((nu + 1) * (nv + 1), dtype=np.uint32).reshape(nu + 1, nv + 1)
        a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        tri = np.stack([a, b, c, a, c, d], axis=-1)
        self.add(pos, nrm, uv, tri, material)

    def column(self, base, radius, height, segs, rings, material):
        th = np.linspace(0, 2 * np.pi, segs + 1, dtype=np.float32)
        hh = np.linspace(0, 1, rings + 1, dtype=np.float32)
        T, Hh = np.meshgrid(th, hh, indexing="ij")
        flute = 1.0 + 0.04 * np.cos(T * 12.0)                       # fluted shaft
        entasis = 1.0 - 0.15 * Hh ** 2
        r = radius * flute * entasis
        pos = np.stack([base[0] + r * np.cos(T), base[1] + height * Hh, base[2] + r * np.sin(T)], axis=-1)
        nrm = np.stack([np.cos(T), np.zeros_like(T), np.sin(T)], axis=-1)
        uv = np.stack([T / (2 * np.pi) * 2.0, Hh * 4.0], axis=-1)
        idx = np.arange((segs + 1) * (rings + 1), dtype=np.uint32).reshape(segs + 1, rings + 1)
        a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        tri = np.stack([a, c, b, a, d, c], axis=-1)
        self.add(pos, nrm, uv, tri, material)

    def arch(self, p0, p1, y, rise, depth, segs, across, material):
        """Half-cylinder vault between two column tops p0 -> p1 (same y), extruded `depth` along z or x."""
        p0, p1 = np.asarray(p0, np.float32), np.asarray(p1, np.float32)
        axis = p1 - p0
        span = np.linalg.norm(axis)
        axis = axis / span
        side = np.array([axis[2], 0, -axis[0]], np.float32)
        a = np.linspace(0, np.pi, segs + 1, dtype=np.float32)
        w = np.linspace(-0.5, 0.5, across + 1, dtype=np.float32)
        A, Wd = np.meshgrid(a, w, indexing="ij")
        along = (1 - np.cos(A)) * 0.5 * span
        up = np.sin(A) * rise
        pos = p0 + along[..., None] * axis + Wd[..., None] * depth * side
        pos[..., 1] = y + up
        nrm = -(np.cos(A)[..., None] * (-axis) + np.sin(A)[..., None] * np.array([0, 1, 0], np.float32))
        uv = np.stack([A / np.pi * 2.0, Wd + 0.5], axis=-1)
        idx = np.arange((segs + 1) * (across + 1), dtype=np.uint32).reshape(segs + 1, across + 1)
        q0, q1, q2, q3 = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        tri = np.stack([q0, q1, q2, q0, q2, q3], axis=-1)
        self.add(pos, nrm, uv, tri, material)

    def finish(self):
        return np.concatenate(self.v), np.concatenate(self.i), np.array(self.meshes, MESH_DTYPE)


def _procedural_texture(rng, size, kind, base):
    y, x = np.mgrid[0:size, 0:size].astype(np.float32) / size
    if kind == "brick":
        row = np.floor(y * 16)
        xx = x * 8 + 0.5 * (row % 2)
        mortar = ((xx % 1.0) < 0.06) | (((y * 16) % 1.0) < 0.12)
        tone = 0.8 + 0.2 * rng.random((16, 9)).astype(np.float32)[row.astype(int) % 16, np.floor(xx).astype(int) % 9]
        img = np.where(mortar[..., None], np.float32(0.75), tone[..., None] * np.asarray(base, np.float32) / 255.0)
    elif kind == "tiles":
        c = ((np.floor(x * 12) + np.floor(y * 12)) % 2)
        img = (0.55 + 0.45 * c)[..., None] * np.asarray(base, np.float32) / 255.0
    elif kind == "cloth":
        w = 0.75 + 0.25 * np.sin(x * 180.0) * np.sin(y * 180.0)
        stripes = (np.floor(x * 10) % 2) * 0.35 + 0.65
        img = (w * stripes)[..., None] * np.asarray(base, np.float32) / 255.0
    else:   # "stone": value noise
        n = rng.random((size // 16 + 1, size // 16 + 1)).astype(np.float32)
        n = np.kron(n, np.ones((16, 16), np.float32))[:size, :size]
        img = (0.7 + 0.3 * n)[..., None] * np.asarray(base, np.float32) / 255.0
    rgb = np.clip(img * 255.0 + 0.5, 0, 255).astype(np.uint8)
    return np.concatenate([rgb, np.full((size, size, 1), 255, np.uint8)], axis=-1)


def _sky_texture(width=2048, height=1024):
    """Gradient sky with a soft sun disc, sRGB RGBA8 (equirectangular: v = theta/pi, u = phi/2pi)."""
    v, u = np.mgrid[0:height, 0:width].astype(np.float32)
    v = (v + 0.5) / height
    u = (u + 0.5) / width
    horizon = np.exp(-((v - 0.5) * 6.0) ** 2)
    top = np.clip(1.0 - v * 2.0, 0, 1)
    r = 0.35 + 0.45 * horizon + 0.05 * top
    g = 0.50 + 0.35 * horizon + 0.10 * top
    b = 0.80 + 0.15 * horizon + 0.15 * top
    ground = v > 0.52
    r, g, b = (np.where(ground, 0.25, c) for c in (r, g, b))
    sun = np.exp(-(((u - 0.30) * 2.0) ** 2 + ((v - 0.22) * 1.0) ** 2) * 900.0)
    rgb = np.stack([np.clip(r + sun, 0, 1), np.clip(g + sun, 0, 1), np.clip(b + 0.8 * sun, 0, 1)], axis=-1)
    out = np.concatenate([np.clip(rgb * 255 + 0.5, 0, 255).astype(np.uint8), np.full((height, width, 1), 255, np.uint8)], axis=-1)
    return out


def _height_normal_map(rng, size, cells, strength):
    """RGBA8 UNORM tangent-space normal map of a smooth value-noise height field (`cells` x `cells` control points, tiling)."""
    n = rng.random((cells, cells)).astype(np.float32)
    n = np.concatenate([n, n[:1]], 0)
    n = np.concatenate([n, n[:, :1]], 1)
    t = (np.arange(size, dtype=np.float32) + 0.5) / size * cells
    i = np.floor(t).astype(int)
    f = t - i
    f = f * f * (3.0 - 2.0 * f)
    rows = n[i][:, :] * (1.0 - f)[:, None] + n[i + 1][:, :] * f[:, None]                 # (size, cells + 1)
    h = rows[:, i] * (1.0 - f)[None, :] + rows[:, i + 1] * f[None, :]                     # (size, size)
    dx = (np.roll(h, -1, 1) - np.roll(h, 1, 1)) * (0.5 * size / cells) * strength
    dy = (np.roll(h, -1, 0) - np.roll(h, 1, 0)) * (0.5 * size / cells) * strength
    nrm = np.stack([-dx, -dy, np.ones_like(h)], -1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    rgb = np.clip((nrm * 0.5 + 0.5) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    return np.concatenate([rgb, np.full((size, size, 1), 255, np.uint8)], axis=-1)


def _opacity_map(rng, size, kind):
    """Gray opacity map (the any-hit shader keeps a candidate when the red channel is >= 0.5, raytrace_hit.rahit:24-39): `lace` -- a
    woven cloth with a regular grid of holes; `leaves` -- a card of overlapping leaf blobs on a transparent ground."""
    y, x = np.mgrid[0:size, 0:size].astype(np.float32) / size
    if kind == "lace":
        u, v = (x * 24.0) % 1.0, (y * 24.0) % 1.0
        hole = ((u - 0.5) ** 2 + (v - 0.5) ** 2) < 0.11
        a = np.where(hole, 0.0, 1.0)
    else:
        a = np.zeros((size, size), np.float32)
        for _ in range(70):
            cx, cy = rng.random(), 0.08 + 0.9 * rng.random()
            ang, lx, ly = rng.random() * np.pi, 0.03 + 0.05 * rng.random(), 0.012 + 0.02 * rng.random()
            dxw = (x - cx + 0.5) % 1.0 - 0.5            # wraps in u, so the card tiles around a plant
            dx = dxw * np.cos(ang) + (y - cy) * np.sin(ang)
            dy = -dxw * np.sin(ang) + (y - cy) * np.cos(ang)
            a = np.maximum(a, ((dx / lx) ** 2 + (dy / ly) ** 2 < 1.0).astype(np.float32))
        a = np.maximum(a, ((np.abs(x - 0.5) < 0.012) & (y > 0.3)).astype(np.float32))      # a stem
    return np.clip(a * 255.0 + 0.5, 0, 255).astype(np.uint8)


def _roughness_map(rng, size):
    """Gray roughness map: worn patches (value noise in 32-texel blocks) over a fine grain."""
    n = rng.random((size // 32 + 1, size // 32 + 1)).astype(np.float32)
    n = np.kron(n, np.ones((32, 32), np.float32))[:size, :size]
    g = rng.random((size, size)).astype(np.float32)
    return np.clip((0.35 + 0.5 * n + 0.15 * g) * 255.0 + 0.5, 0, 255).astype(np.uint8)


N_ATRIUM_MATERIALS = 25     # scene materials of the atrium, each with its own texture; material 0 is the format's default


def atrium_scene(seed=1, detail=0.564, texture_size=1024, sky_size=(2048, 1024), sponza_like=False):
    """Sponza-like atrium: floor, four walls with window openings, two storeys of colonnades
    (fluted columns + vaulted arches), draped cloth between columns, roof beams; open to the sky.
    The default detail gives 262 267 triangles -- Sponza's count -- in 25 Lambert / Uber materials, each with its
    own `texture_size`^2 procedural sRGB texture (1024^2: 100 MB of texels), a sun and a 2048 x 1024 sky.

    sponza_like=True adds the content classes real Sponza exercises and this stand-in otherwise lacks (reference README.md:59-61):
    opacity-mapped geometry -- the draped cloths become lace, foliage cards stand between the ground-floor columns and vines hang from
    the gallery (candidates on them go through the any-hit alpha test, raytrace_hit.rahit:24-39, acceleration.rs:136-141) -- normal
    maps on the stone materials (raytrace_hit.rchit:53-60) and roughness maps on the Uber materials."""
    # (tools may pass a subset of {"opacity", "normal", "roughness"} to see what each class costs)
    classes = {"opacity", "normal", "roughness"} if sponza_like is True else set(sponza_like or ())
    sponza_like = bool(classes)
    rng = np.random.default_rng(seed)
    B = _Builder()
    L, Wd, H = 36.0, 16.0, 14.0            # length (x), width (z), height (y)
    d = float(detail)
    k = lambda n: max(2, int(round(n * np.sqrt(d))))
    # materials: 0 default, then 25 scene materials, every one with a texture of its own (texture id = material id)
    # by role: 0 short walls, 1 long walls, 2 floor, 3 gallery floor, 4-6 drapes, 7 gallery underside, 8-11 ground-floor columns, 12 their
    # capitals, 13 / 14 arches of the two storeys, 15-18 upper columns, 19 their capitals, 20 roof beams, 21-24 braziers
    stone, brick, tiles, cloth = "stone", "brick", "tiles", "cloth"
    kinds = [(stone, (200, 190, 170)), (brick, (170, 90, 70)), (tiles, (210, 200, 180)), (stone, (150, 150, 155)),
             (cloth, (190, 40, 40)), (cloth, (40, 80, 170)), (cloth, (60, 150, 70)), (stone, (120, 100, 80)),
             (stone, (205, 195, 180)), (stone, (190, 185, 170)), (stone, (210, 200, 175)), (stone, (185, 175, 165)), (stone, (225, 215, 200)),
             (brick, (180, 120, 90)), (brick, (160, 110, 95)),
             (stone, (195, 190, 185)), (stone, (180, 180, 175)), (stone, (200, 195, 190)), (stone, (175, 170, 170)), (stone, (215, 210, 200)),
             (stone, (110, 80, 55)), (tiles, (150, 140, 120)), (brick, (140, 70, 60)), (tiles, (120, 130, 140)), (stone, (90, 90, 95))]
    assert len(kinds) == N_ATRIUM_MATERIALS
    textures = [(abi.TEX_RGBA_SRGB, np.full((1, 1, 4), 255, np.uint8), "default")]
    material_kind = []
    for i in range(N_ATRIUM_MATERIALS):
        kind, tint = kinds[i]
        textures.append((abi.TEX_RGBA_SRGB, _procedural_texture(rng, texture_size, kind, tint), "%s%d" % (kind, i)))
        material_kind.append(kind)
    sky_tex_id = len(textures)
    textures.append((abi.TEX_RGBA_SRGB, _sky_texture(*sky_size), "sky"))
    extra = {}
    if sponza_like:
        def add_texture(fmt, pixels, name):
            textures.append((fmt, pixels, name))
            return len(textures) - 1
        extra["lace"] = add_texture(abi.TEX_GRAY, _opacity_map(rng, texture_size, "lace"), "lace")
        extra["leaves"] = add_texture(abi.TEX_GRAY, _opacity_map(rng, texture_size, "leaves"), "leaves")
        extra["normal"] = [add_texture(abi.TEX_RGBA_NORM, _height_normal_map(rng, texture_size, 48 + 16 * j, 0.6 + 0.2 * j), "stone_normal%d" % j) for j in range(3)]
        extra["rough"] = [add_texture(abi.TEX_GRAY, _roughness_map(rng, texture_size), "rough%d" % j) for j in range(2)]
    materials = [make_material("default")]
    for i in range(N_ATRIUM_MATERIALS):
        tex = 1 + i
        maps = {}
        if "normal" in classes and material_kind[i] in ("stone", "brick"):
            maps["normal"] = extra["normal"][i % 3]
        if "opacity" in classes and material_kind[i] == "cloth":
            maps["opacity"] = extra["lace"]
        if i % 5 == 3:
            if "roughness" in classes:
                maps["roughness"] = extra["rough"][(i // 5) % 2]
            materials.append(make_material("uber%d" % i, mtype=abi.MAT_UBER, diffuse=tex, roughness_mul=0.35 + 0.02 * i,
                                           metalness_mul=0.0, diffuse_mul=(230, 230, 230), **maps))
        else:
            materials.append(make_material("lambert%d" % i, mtype=abi.MAT_LAMBERT, diffuse=tex,
                                           diffuse_mul=(255 - 3 * i, 250 - 2 * i, 245 - 2 * i), **maps))
    if sponza_like:
        materials.append(make_material("foliage", mtype=abi.MAT_LAMBERT, diffuse=1 + 6, diffuse_mul=(150, 235, 140), opacity=extra["leaves"] if "opacity" in classes else 0))   # (the green cloth's texture)
    M = lambda i: 1 + (i % N_ATRIUM_MATERIALS)
    FOLIAGE = N_ATRIUM_MATERIALS + 1

    def bumpy(amp, freq):
        def f(pos, nrm, s, t):
            h = amp * (np.sin(s * freq * 6.28) * np.sin(t * freq * 6.28))
            return pos + nrm * h[..., None], nrm
        return f

    # floor (tiles) and surrounding ground
    B.grid((-L / 2, 0, -Wd / 2), (0, 0, Wd), (L, 0, 0), k(120), k(260), M(2), uv_scale=10.0, displace=bumpy(0.004, 40))
    # long walls (brick) with relief, two storeys; short walls (stone)
    for z, sgn in ((-Wd / 2, 1), (Wd / 2, -1)):
        o = (-L / 2, 0, z) if sgn > 0 else (L / 2, 0, z)
        B.grid(o, (sgn * L, 0, 0), (0, H, 0), k(260), k(100), M(1), uv_scale=6.0, displace=bumpy(0.02, 30))
    for x, sgn in ((-L / 2, -1), (L / 2, 1)):
        o = (x, 0, Wd / 2) if sgn < 0 else (x, 0, -Wd / 2)
        B.grid(o, (0, 0, -Wd) if sgn < 0 else (0, 0, Wd), (0, H, 0), k(120), k(100), M(0), uv_scale=4.0, displace=bumpy(0.02, 20))
    # gallery floors (second storey walkways along the long walls)
    for z0, z1 in ((-Wd / 2, -Wd / 2 + 3.0), (Wd / 2 - 3.0, Wd / 2)):
        B.grid((-L / 2, 6.0, z0), (0, 0, z1 - z0), (L, 0, 0), k(24), k(200), M(3), uv_scale=8.0)      # top
        B.grid((-L / 2, 5.7, z0), (L, 0, 0), (0, 0, z1 - z0), k(200), k(24), M(7), uv_scale=8.0)      # underside
    # colonnades: two rows x two storeys
    ncol = 12
    xs = np.linspace(-L / 2 + 2.0, L / 2 - 2.0, ncol)
    for zi, z in enumerate((-Wd / 2 + 3.0, Wd / 2 - 3.0)):
        for storey, (y0, hc, rad) in enumerate(((0.0, 5.0, 0.38), (6.0, 4.2, 0.30))):
            for ci, x in enumerate(xs):
                B.column((x, y0, z), rad, hc, k(40), k(36), M((8 if storey == 0 else 15) + (ci + zi) % 4))
                # capital + base as short wide rings
                B.column((x, y0 + hc, z), rad * 1.5, 0.35, k(24), 2, M(12 if storey == 0 else 19))
                B.column((x, y0 - 0.0, z), rad * 1.4, 0.25, k(24), 2, M(12 if storey == 0 else 19))
            for ci in range(ncol - 1):
                B.arch((xs[ci], 0, z), (xs[ci + 1], 0, z), y0 + hc + 0.35, 0.9 if storey == 0 else 0.7, 0.8, k(28), k(6),
                       M(13 + storey))
    # draped cloth between first-storey columns of one row (catenary + ripples)
    for ci in range(0, ncol - 1, 2):
        x0, x1 = xs[ci], xs[ci + 1]
        z = -Wd / 2 + 3.0 + 0.5

        def drape(pos, nrm, s, t, x0=x0, x1=x1):
            sag = 1.2 * (1 - (2 * s - 1) ** 2) * (0.3 + 0.7 * t)
            pos = pos.copy()
            pos[..., 1] -= sag
            pos[..., 2] += 0.25 * np.sin(s * 18.0 + t * 5.0) * t
            gx = np.gradient(pos, axis=0)
            gy = np.gradient(pos, axis=1)
            n = np.cross(gx, gy)
            n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-8)
            return pos, n.astype(np.float32)

        B.grid((x0, 5.4, z), (x1 - x0, 0, 0), (0, -3.2, 0), k(56), k(56), M(4 + ci % 3), uv_scale=2.0, displace=drape)
    # roof beams across the opening
    for x in np.linspace(-L / 2 + 3, L / 2 - 3, 9):
        B.grid((x - 0.2, H - 0.6, -Wd / 2), (0.4, 0, 0), (0, 0, Wd), 2, k(60), M(20))
        B.grid((x - 0.2, H - 0.6, -Wd / 2), (0, 0, Wd), (0, 0.6, 0), k(60), 2, M(20))
        B.grid((x + 0.2, H - 0.6, Wd / 2), (0, 0, -Wd), (0, 0.6, 0), k(60), 2, M(20))
    # a brazier in the middle of the floor: an open cone of single triangles -- they bring the count to Sponza's 262 267 and give the
    # hierarchy leaves of one triangle next to the paired ones (materials 21 ... 24 are its rings' and the braziers' along the walls)
    def cone(base, radius, height, segs, material):
        th = np.linspace(0, 2 * np.pi, segs + 1, dtype=np.float32)
        ring = np.stack([base[0] + radius * np.cos(th), np.full_like(th, base[1]), base[2] + radius * np.sin(th)], -1)
        pos = np.concatenate([ring, np.array([[base[0], base[1] + height, base[2]]], np.float32)])
        out = np.stack([np.cos(th), np.full_like(th, radius / height), np.sin(th)], -1)
        nrm = np.concatenate([out / np.linalg.norm(out, axis=-1, keepdims=True), np.array([[0, 1, 0]], np.float32)])
        uv = np.concatenate([np.stack([th / (2 * np.pi) * 3.0, np.zeros_like(th)], -1), np.array([[1.5, 1.0]], np.float32)])
        apex = segs + 1
        tri = np.stack([np.arange(segs, dtype=np.uint32), np.full(segs, apex, np.uint32), np.arange(1, segs + 1, dtype=np.uint32)], -1)
        B.add(pos, nrm, uv, tri, material)

    if detail == 0.564:
        need = 262267 - B.ni // 3
        segs = [need // 4 + (1 if j < need % 4 else 0) for j in range(4)]
        for j, (cx, cz) in enumerate(((0.0, 0.0), (-9.0, 2.5), (9.0, -2.5), (4.0, 3.0))):
            cone((cx, 0.0, cz), 0.45, 1.1, segs[j], M(21 + j))
    else:
        for j, (cx, cz) in enumerate(((0.0, 0.0), (-9.0, 2.5), (9.0, -2.5), (4.0, 3.0))):
            cone((cx, 0.0, cz), 0.45, 1.1, k(32), M(21 + j))
    if sponza_like:
        # foliage: three crossed cards per planter between the ground-floor columns of both rows, vines hanging from the gallery's edge
        def card(centre, half_w, height, yaw, material):
            c, s_ = np.cos(yaw), np.sin(yaw)
            ax = np.array([c, 0, s_], np.float32) * half_w
            p0 = np.asarray(centre, np.float32)
            pos = np.stack([p0 - ax, p0 + ax, p0 + ax + (0, height, 0), p0 - ax + (0, height, 0)]).astype(np.float32)
            n = np.array([-s_, 0, c], np.float32)
            uv = np.array([[0, 1], [1, 1], [1, 0], [0, 0]], np.float32)
            B.add(pos, np.broadcast_to(n, pos.shape).copy(), uv, np.array([0, 1, 2, 0, 2, 3], np.uint32), material)
        for zi, z in enumerate((-Wd / 2 + 3.0, Wd / 2 - 3.0)):
            for ci in range(ncol - 1):
                cx = 0.5 * (xs[ci] + xs[ci + 1])
                for j in range(3):
                    card((cx, 0.0, z + (0.9 if zi == 0 else -0.9)), 0.8, 1.9, j * np.pi / 3 + 0.2 * ci, FOLIAGE)
                for j in range(2):   # vines: narrow tall cards in front of the arches
                    card((cx + (j - 0.5) * 1.1, 2.2, z + (0.45 if zi == 0 else -0.45)), 0.35, 3.4, 0.15 * (ci + j), FOLIAGE)
    vertices, indices, meshes = B.finish()
    instances = np.array([(m, 0) for m in range(meshes.shape[0])], INSTANCE_DTYPE)
    lights = [make_light(abi.LIGHT_SUN, "sun", direction=(-0.35, -0.85, 0.25), intensity=2.5),
              make_light(abi.LIGHT_SKY, "sky", resource_id=sky_tex_id, intensity=1.0, yaw=0.0, pitch=90.0, roll=0.0)]
    camera = make_camera(position=(-L / 2 + 4.0, 2.2, 0.6), target=(L / 2, 4.5, -0.8), up=(0, 1, 0),
                         fovx=np.float32(np.radians(np.float32(75.0))), near=1e-2, far=200.0)
    radius = float(np.float32(0.5 * np.sqrt(L * L + Wd * Wd + H * H)))
    meta = make_meta(centre=(0, H / 2, 0), radius=radius, exposure=1.0)
    return SceneDesc(vertices, indices, meshes, None, instances, materials, lights, textures, camera, meta)


# ------------------------------------------------------------------------------------------------
# instanced forest of columns
# ------------------------------------------------------------------------------------------------
def forest_scene(n, seed=3):
    """One fluted column (6 144 triangles) instanced n times under random rotations, scales and placements on a flat 40 m ground
    grid (8 192 triangles): the kind of scene a two-level structure is for (1 + n instances of 2 meshes)."""
    rng = np.random.default_rng(seed)
    B = _Builder()
    B.grid((-20, 0, -20), (0, 0, 40), (40, 0, 0), 64, 64, 1, uv_scale=20.0)
    B.column((0, 0, 0), 0.25, 3.0, 48, 64, 2)
    vertices, indices, meshes = B.finish()
    mats = [np.eye(4)]
    for i in range(n):
        a = rng.uniform(0, 2 * np.pi)
        s = rng.uniform(0.5, 1.5, 3)
        m = np.eye(4)
        m[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ np.diag(s)
        m[:3, 3] = (rng.uniform(-19, 19), 0.0, rng.uniform(-19, 19))
        mats.append(m)
    transforms = np.stack([np.asarray(m, np.float32).T.reshape(16) for m in mats])
    instances = np.array([(0, 0)] + [(1, 1 + i) for i in range(n)], INSTANCE_DTYPE)
    materials = [make_material("default"), make_material("ground", mtype=abi.MAT_LAMBERT, diffuse=1, diffuse_mul=(200, 200, 200)),
                 make_material("column", mtype=abi.MAT_UBER, diffuse_mul=(220, 210, 190), roughness_mul=0.4)]
    textures = [(abi.TEX_RGBA_SRGB, np.full((1, 1, 4), 255, np.uint8), "default"), (abi.TEX_RGBA_SRGB, checker_texture(), "checker")]
    lights = [make_light(abi.LIGHT_SUN, "sun", direction=(-0.35, -0.85, 0.25), intensity=2.5),
              make_light(abi.LIGHT_OMNI, "omni", position=(0.0, 6.0, 0.0), intensity=40.0)]
    camera = make_camera(position=(-18, 2.0, -18), target=(10, 1.0, 10), up=(0, 1, 0), fovx=np.float32(np.radians(np.float32(70.0))), near=1e-2, far=200.0)
    meta = make_meta(centre=(0, 2, 0), radius=30.0, exposure=1.0)
    return SceneDesc(vertices, indices, meshes, transforms, instances, materials, lights, textures, camera, meta)
