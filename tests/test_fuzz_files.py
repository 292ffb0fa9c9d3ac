"""The scenes of tests/fuzz_scenes.py through the file: SceneDesc -> Serializer -> `.glaze` -> parse (the C++ reader) and -> oracle/glaze_v1.py
(the independent python reader): every chunk must come back as it went in -- odd-sized textures of all three formats, every material
family and light type, mirrored transforms, instances that share meshes -- and the oracle must render the file exactly as it renders
the description."""
import ctypes as C

import numpy as np

from fuzz_scenes import LARGE, random_scene
from glaze_amd import abi, parse
from glaze_amd.scene_desc import save_scene
from helpers import desc_from_oracle_parse
from oracle.pyoracle import OracleRenderer, OracleScene


def raw(s):
    return bytes((C.c_char * C.sizeof(s)).from_address(C.addressof(s)))


def test_random_scenes_survive_the_file(tmp_path):
    for seed in list(range(40)) + [LARGE, LARGE + 1]:
        desc, run = random_scene(seed)
        path = str(tmp_path / ("fuzz_%d.glaze" % seed))
        save_scene(desc, path)
        p = parse(path)
        assert np.array_equal(p.vertices().view(np.uint32), np.stack([desc.vertices["vv"][:, 0], desc.vertices["vv"][:, 1], desc.vertices["vv"][:, 2],
                                                                      desc.vertices["vn"][:, 0], desc.vertices["vn"][:, 1], desc.vertices["vn"][:, 2],
                                                                      desc.vertices["vt"][:, 0], desc.vertices["vt"][:, 1]], 1).view(np.uint32)), seed
        meshes = p.meshes()
        assert len(meshes) == len(desc.meshes)
        for m, d in zip(meshes, desc.meshes):
            assert (m["id"], m["material"]) == (int(d["id"]), int(d["material"]))
            assert np.array_equal(m["indices"], desc.indices[int(d["index_offset"]):int(d["index_offset"]) + int(d["index_count"])])
        assert np.array_equal(p.transforms().view(np.uint32), desc.transforms.view(np.uint32))
        assert np.array_equal(p.instances(), np.stack([desc.instances["mesh_id"], desc.instances["transform_id"]], 1))
        assert [raw(m) for m in p.materials()] == [raw(m) for m in desc.materials], seed
        assert [raw(l) for l in p.lights()] == [raw(l) for l in desc.lights], seed
        assert raw(p.cameras()[0]) == raw(desc.camera) and raw(p.meta()) == raw(desc.meta)
        tex = p.textures()
        assert len(tex) == len(desc.textures)
        for (fmt, px, name, _), t in zip(tex, desc.textures):
            assert fmt == t[0] and name == t[2] and np.array_equal(px, t[1]), (seed, name)
        # the independent reader, and what the oracle makes of the file
        again = desc_from_oracle_parse(path)
        images = []
        for dsc in (desc, again):
            o = OracleRenderer(OracleScene(dsc), run["w"], run["h"])
            o.set_integrator(run["integrator"].value)
            o.set_depth(run["depth"])
            o.set_seed(run["seed"])
            o.draw(min(run["spp"], 2))
            images.append(np.nan_to_num(o.read_hdr(), nan=-1.0).view(np.uint32))
        assert np.array_equal(images[0], images[1]), seed
