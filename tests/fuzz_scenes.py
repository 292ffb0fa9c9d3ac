"""Random scenes nobody designed, for the parity fuzz runs (tests/test_oracle_fuzz.py, tests/test_gpu_fuzz.py, tools/gpu_fuzz_parity.py):
random meshes (grids, boxes, triangle soups with degenerate and duplicate triangles), instances under random transforms (rotation,
non-uniform and mirroring scales; the same mesh under the same transform twice happens -- coincident triangles, exact ties), every
material family with random parameters and texture bindings (odd-sized sRGB / grey / normal textures, opacity maps), every light
type, perspective and orthographic cameras, and a random way to render each (size, samples, depth, seed, integrator, launch mode,
acceleration-structure levels, chains).  random_scene(seed) is deterministic.
"""
import numpy as np

import glaze_amd
from glaze_amd import abi
from glaze_amd.scene_desc import INSTANCE_DTYPE, MESH_DTYPE, VERTEX_DTYPE, SceneDesc, make_camera, make_light, make_material, make_meta


def col_major(m):
    return np.asarray(m, np.float32).T.reshape(16)


def rot(axis, deg):
    a = np.radians(deg)
    c, s = np.cos(a), np.sin(a)
    m = np.eye(4)
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    m[i, i], m[i, j], m[j, i], m[j, j] = c, -s, s, c
    return m


def random_transform(rng, spread):
    t = np.eye(4)
    t[:3, 3] = rng.uniform(-spread, spread, 3)
    s = np.diag(list(rng.uniform(0.2, 1.2, 3) * rng.choice([1.0, 1.0, 1.0, -1.0], 3)) + [1.0])
    return t @ rot(int(rng.integers(3)), rng.uniform(0, 360)) @ rot(int(rng.integers(3)), rng.uniform(0, 360)) @ s


def unit(v):
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    return v / np.where(n > 0, n, 1.0)


def mesh_grid(rng, large=False):
    nu, nv = int(rng.integers(1, 49 if large else 7)), int(rng.integers(1, 49 if large else 7))
    s, t = np.meshgrid(np.linspace(-0.5, 0.5, nu + 1), np.linspace(-0.5, 0.5, nv + 1), indexing="ij")
    pos = np.stack([s, 0.15 * rng.standard_normal(s.shape) * rng.choice([0.0, 1.0]), t], -1).reshape(-1, 3)
    nrm = unit(np.array([0.0, 1.0, 0.0]) + 0.3 * rng.standard_normal(pos.shape))
    uv = np.stack([s, t], -1).reshape(-1, 2) * rng.uniform(0.5, 3.0) + rng.uniform(-1, 1, 2)
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    tri = np.stack([a, b, c, a, c, d], -1).reshape(-1)
    return pos, nrm, uv, tri


def mesh_box(rng, inward=False, size=0.5):
    c = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float64) * size
    faces = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    pos, nrm, uv, tri = [], [], [], []
    for f in faces:
        p = c[list(f)]
        n = unit(np.cross(p[1] - p[0], p[2] - p[0]))
        if inward:
            n = -n
        base = len(pos)
        pos += list(p)
        nrm += [n] * 4
        uv += [(0, 0), (1, 0), (1, 1), (0, 1)]
        tri += [base, base + 1, base + 2, base, base + 2, base + 3]
    return np.array(pos), np.array(nrm), np.array(uv, np.float64), np.array(tri)


def mesh_soup(rng, large=False):
    k = int(rng.integers(1, 400 if large else 24))
    pos = rng.uniform(-0.5, 0.5, (3 * k, 3))
    pos[1::3] = pos[0::3] + rng.uniform(-0.3, 0.3, (k, 3))
    pos[2::3] = pos[0::3] + rng.uniform(-0.3, 0.3, (k, 3))
    tri = np.arange(3 * k)
    if k > 2 and rng.random() < 0.5:
        tri[3:6] = tri[0:3]                      # a duplicate triangle
    if k > 3 and rng.random() < 0.5:
        pos[10] = pos[9]                         # a degenerate one
    nrm = unit(rng.standard_normal(pos.shape))
    return pos, nrm, rng.uniform(-2, 2, (3 * k, 2)), tri


def random_texture(rng, fmt):
    h, w = (int(rng.integers(1, 70)), int(rng.integers(1, 70))) if rng.random() < 0.7 else (int(2 ** rng.integers(0, 7)), int(2 ** rng.integers(0, 7)))
    if fmt == abi.TEX_GRAY:
        if rng.random() < 0.5:
            y, x = np.mgrid[0:h, 0:w]
            return np.where(((x // 3 + y // 2) % 2) == 0, 255, 0).astype(np.uint8)
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    if fmt == abi.TEX_RGBA_NORM:
        px[..., :2] = rng.integers(90, 166, (h, w, 2), dtype=np.uint8)
        px[..., 2] = 255
    px[..., 3] = 255
    return px


LARGE = 1000000     # seeds from here on: more and bigger meshes (thousands of triangles), bigger images


def random_scene(seed):
    rng = np.random.default_rng(seed)
    large = seed >= LARGE
    # textures: 0 = the white default
    textures = [(abi.TEX_RGBA_SRGB, np.full((1, 1, 4), 255, np.uint8), "default")]
    by_fmt = {abi.TEX_RGBA_SRGB: [0], abi.TEX_GRAY: [], abi.TEX_RGBA_NORM: []}
    for _ in range(int(rng.integers(0, 6))):
        fmt = int(rng.choice([abi.TEX_RGBA_SRGB, abi.TEX_RGBA_SRGB, abi.TEX_GRAY, abi.TEX_RGBA_NORM]))
        by_fmt[fmt].append(len(textures))
        textures.append((fmt, random_texture(rng, fmt), "t%d" % len(textures)))

    def pick(fmt, p):
        return int(rng.choice(by_fmt[fmt])) if by_fmt[fmt] and rng.random() < p else 0

    materials = [make_material("default")]
    for k in range(int(rng.integers(1, 6))):
        emissive = tuple(int(v) for v in rng.integers(0, 256, 3)) if rng.random() < 0.2 else None
        materials.append(make_material(
            "m%d" % k, mtype=int(rng.integers(0, 7)), metal=int(rng.integers(0, 29)), diffuse_mul=tuple(int(v) for v in rng.integers(0, 256, 3)),
            emissive=emissive, ior=float(rng.uniform(1.05, 2.2)), roughness_mul=float(rng.uniform(0, 1)), metalness_mul=float(rng.uniform(0, 1)),
            anisotropy=float(rng.uniform(-0.9, 0.9)) if rng.random() < 0.5 else 0.0, diffuse=pick(abi.TEX_RGBA_SRGB, 0.6),
            roughness=pick(abi.TEX_GRAY, 0.4), metalness=pick(abi.TEX_GRAY, 0.4), normal=pick(abi.TEX_RGBA_NORM, 0.5), opacity=pick(abi.TEX_GRAY, 0.3)))
    # meshes: optionally a room around everything (inward box), then random objects
    parts, meshes = [], []
    nv = ni = 0

    def add(pos, nrm, uv, tri, material):
        nonlocal nv, ni
        block = np.zeros(len(pos), VERTEX_DTYPE)
        block["vv"], block["vn"], block["vt"] = pos, nrm, uv
        parts.append((block, np.asarray(tri, np.uint32) + nv))
        meshes.append((len(meshes), material, ni, len(tri)))
        nv += len(pos)
        ni += len(tri)

    room = rng.random() < 0.7
    if room:
        add(*mesh_box(rng, inward=True, size=3.0), int(rng.integers(0, len(materials))))
    for _ in range(int(rng.integers(1, 13 if large else 5))):
        kind = int(rng.integers(3))
        add(*(mesh_grid(rng, large) if kind == 0 else mesh_box(rng) if kind == 1 else mesh_soup(rng, large)), int(rng.integers(0, len(materials))))
    transforms = [np.eye(4)] + [random_transform(rng, 1.5) for _ in range(int(rng.integers(1, 6)))]
    instances = [(0, 0)] if room else []
    for m in range(1 if room else 0, len(meshes)):
        for _ in range(int(rng.integers(1, 4))):
            instances.append((m, int(rng.integers(0, len(transforms)))))
    lights = []
    for _ in range(int(rng.integers(1, 4)) if rng.random() < 0.93 else 0):
        lt = int(rng.integers(4))
        if lt == abi.LIGHT_OMNI:
            lights.append(make_light(abi.LIGHT_OMNI, "omni", position=tuple(rng.uniform(-2, 2, 3)), intensity=float(rng.uniform(0.2, 3))))
        elif lt == abi.LIGHT_SUN:
            lights.append(make_light(abi.LIGHT_SUN, "sun", direction=tuple(unit(rng.standard_normal(3))), intensity=float(rng.uniform(0.2, 2))))
        elif lt == abi.LIGHT_AREA:
            lights.append(make_light(abi.LIGHT_AREA, "area", resource_id=int(rng.integers(0, len(materials))), intensity=float(rng.uniform(0.2, 2))))
        elif not any(l.ltype == abi.LIGHT_SKY for l in lights):
            lights.append(make_light(abi.LIGHT_SKY, "sky", resource_id=int(rng.choice(by_fmt[abi.TEX_RGBA_SRGB])), intensity=float(rng.uniform(0.2, 2)),
                                     yaw=float(rng.uniform(0, 360)), pitch=float(rng.uniform(-90, 90)), roll=float(rng.uniform(0, 360))))
    eye = rng.uniform(-1.8, 1.8, 3)
    ortho = rng.random() < 0.25
    camera = make_camera(position=tuple(eye), target=tuple(rng.uniform(-0.5, 0.5, 3)), up=(0, 1, 0), fovx=np.float32(np.radians(rng.uniform(30, 110))),
                         near=1e-3, far=100.0, orthographic=ortho, scale=float(rng.uniform(1, 4)))
    vertices = np.concatenate([p[0] for p in parts])
    indices = np.concatenate([p[1] for p in parts])
    desc = SceneDesc(vertices, indices, np.array(meshes, MESH_DTYPE), np.stack([col_major(t) for t in transforms]), np.array(instances, INSTANCE_DTYPE),
                     materials, lights, textures, camera, make_meta(centre=(0, 0, 0), radius=6.0, exposure=float(rng.uniform(0.5, 2))))
    run = dict(w=int(rng.integers(9, 200 if large else 90)), h=int(rng.integers(9, 160 if large else 90)), spp=int(rng.integers(2, 6)), depth=int(rng.integers(1, 9)), seed=int(rng.integers(0, 1000)),
               integrator=glaze_amd.Integrator.PATH_TRACE if rng.random() < 0.8 else glaze_amd.Integrator.DIRECT, mode=str(rng.choice(["two_kernels", "path", "auto"])),
               levels=str(rng.choice(["auto", "flat", "two_level"])), chains=int(rng.integers(0, 4)))
    # more ways to render it, from a stream of their own (the scenes of the seeds that found something stay what they were)
    more = np.random.default_rng(7000000 + seed)
    run["builder"] = str(more.choice(["auto", "auto", "lbvh", "ploc", "sah_host"]))
    run["lod"] = int(more.choice([0, 0, 0, 1, 2]))
    run["exposure_after"] = int(more.integers(1, 6)) if more.random() < 0.3 else 0      # launches after which the exposure changes (0: never)
    run["exposure"] = float(more.uniform(0.3, 3.0))
    run["camera_after"] = int(more.integers(1, 6)) if more.random() < 0.2 else 0        # launches after which update_camera replaces the camera (0: never)
    run["camera"] = make_camera(position=tuple(more.uniform(-1.8, 1.8, 3)), target=tuple(more.uniform(-0.5, 0.5, 3)), up=(0, 1, 0),
                                fovx=np.float32(np.radians(more.uniform(30, 110))), near=1e-3, far=100.0, orthographic=bool(more.random() < 0.3),
                                scale=float(more.uniform(1, 4)))
    world = int(more.integers(2, 6))
    run["partition"] = (int(more.integers(0, world)), world) if more.random() < 0.25 else None   # render rank r's tiles of a world of n only
    run["devices"] = int(more.integers(2, 4)) if (run["partition"] is None and more.random() < 0.12) else 1   # the one GPU named n times (loop-back set_devices)
    run["via_file"] = bool(more.random() < 0.1)          # the HIP side reads the scene from a `.glaze` file (Serializer -> parse -> RayTraceScene.new)
    run["stored_mips"] = int(more.integers(2, 12)) if more.random() < 0.5 else 0   # ... whose textures carry that many mip levels (0: level 0 only)
    run["node_width"] = int(more.choice([0, 0, 8]))      # the two-kernel mode's traversal over the hierarchy's 8-wide nodes (k_trace8) a third of the time
    return desc, run


def render_oracle(desc, run):
    """the oracle's half of render_both alone (for a child process that runs with ORC_DEBUG_PIXEL set)"""
    from oracle.pyoracle import OracleRenderer, OracleScene
    o = OracleRenderer(OracleScene(desc), run["w"], run["h"])
    o.set_threads(1)
    o.set_integrator(run["integrator"].value)
    o.set_depth(run["depth"])
    o.set_seed(run["seed"])
    if run.get("lod", 0):
        o.set_texture_lod(run["lod"])
    o.restart()
    if run.get("partition"):
        rank, world = run["partition"]
        n_tiles = ((run["w"] + 63) // 64) * ((run["h"] + 63) // 64)
        own = [t for t in range(n_tiles) if t % world == rank]
        if own:
            o.set_tiles(own)
            o.restart()
    launches = run["spp"] * o.steps_per_sample()
    events = sorted([(min(run.get("exposure_after", 0), launches), "exposure"), (min(run.get("camera_after", 0), launches), "camera")])
    done = 0
    for at, what in events:
        if at == 0:
            continue
        o.step(at - done)
        done = at
        if what == "exposure":
            o.set_exposure(run["exposure"])
        else:
            o.update_camera(run["camera"])
    o.step(launches - done)
    return o


def render_both(desc, run, levels=None, mode=None):
    """the scene through the HIP path and through the oracle, the way `run` says (levels / mode override it): (renderer, oracle renderer)"""
    from oracle.pyoracle import OracleRenderer, OracleScene
    inst = glaze_amd.RayTraceInstance.new()
    inst.set_as_levels(levels or run["levels"])
    inst.set_bvh_builder(run.get("builder", "auto"))
    if run.get("via_file"):
        import os
        import tempfile
        from glaze_amd.scene_desc import save_scene
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "fuzz.glaze")
            stored = desc
            if run.get("stored_mips"):
                # never the COMPLETE chain: a texture that brings all its levels is rendered from them (the converter's Catmull-Rom levels,
                # materials/texture.rs:196-221; tests/test_gpu_texture_lod.py), which the oracle, given the description, cannot know;
                # an incomplete chain is dropped and regenerated, and that path is what this exercises
                def some(t):
                    full = 1 + int(np.floor(np.log2(max(t[1].shape[0], t[1].shape[1]))))
                    return max(1, min(run["stored_mips"], full - 1))
                stored = desc.copy()
                stored.textures = [(t[0], t[1], t[2], some(t)) for t in desc.textures]
            save_scene(stored, path)
            scene = glaze_amd.RayTraceScene.new(inst, glaze_amd.parse(path))
    else:
        scene = glaze_amd.RayTraceScene.from_desc(inst, desc)
    r = glaze_amd.RayTraceRenderer.new(inst, scene, run["w"], run["h"])
    o = OracleRenderer(OracleScene(desc), run["w"], run["h"])
    if run.get("devices", 1) > 1:
        import os
        before = os.environ.get("GLAZE_MULTI_LOOPBACK")
        os.environ["GLAZE_MULTI_LOOPBACK"] = "1"
        try:
            r.set_devices([inst.device] * run["devices"])
        finally:
            if before is None:
                del os.environ["GLAZE_MULTI_LOOPBACK"]
            else:
                os.environ["GLAZE_MULTI_LOOPBACK"] = before
    r.set_launch_mode(mode or run["mode"])
    r.set_node_width(run.get("node_width", 0))
    r.set_chains(run["chains"])
    for x in (r, o):
        x.set_integrator(run["integrator"] if x is r else run["integrator"].value)
        x.set_depth(run["depth"])
        x.set_seed(run["seed"])
        if run.get("lod", 0):
            x.set_texture_lod(run["lod"])
        x.restart()
    if run.get("partition"):
        rank, world = run["partition"]
        n_tiles = ((run["w"] + 63) // 64) * ((run["h"] + 63) // 64)
        own = [t for t in range(n_tiles) if t % world == rank]
        if own:                                  # (a rank without a tile has nothing to compare: the whole frame then)
            r.set_partition(rank, world)
            o.set_tiles(own)
            r.restart()
            o.restart()
    launches = run["spp"] * r.steps_per_sample()
    events = sorted([(min(run.get("exposure_after", 0), launches), "exposure"), (min(run.get("camera_after", 0), launches), "camera")])
    for x in (r, o):
        done = 0
        for at, what in events:
            if at == 0:
                continue
            x.step(at - done)
            done = at
            if what == "exposure":
                x.set_exposure(run["exposure"])
            else:
                x.update_camera(run["camera"])
        x.step(launches - done)
    return r, o
