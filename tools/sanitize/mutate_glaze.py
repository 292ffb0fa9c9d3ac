"""Writes N mutated copies of a few random scenes' `.glaze` files into DIR: bytes inside one chunk's body changed, truncations and
offset-table damage in a tenth of them, the chunk's XXH64 prefix recomputed so that the damage reaches the decoders.
    python tools/sanitize/mutate_glaze.py SEED N DIR"""
import os
import random
import struct
import sys

import xxhash

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_scenes as fz
from glaze_amd.scene_desc import save_scene

HASHER_SEED = 0x368262AAA1DEB64D
seed, n, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
base = []
for s in (1, 5, 8, 13, fz.LARGE + 2):
    desc, _ = fz.random_scene(s)
    p = os.path.join(out, "base.tmp")
    save_scene(desc, p)
    base.append(open(p, "rb").read())
    os.remove(p)
rng = random.Random(seed)
for it in range(n):
    data = bytearray(rng.choice(base))
    count = data[24]
    chunks = [struct.unpack_from("<BQQ", data, 25 + 17 * i) for i in range(count)]
    cid, off, ln = rng.choice(chunks)
    body0, body1 = off + 8, off + ln
    for _ in range(rng.choice([1, 1, 2, 3, 8])):
        i = rng.randrange(body0, body1)
        r = rng.random()
        if r < 0.6:
            data[i] = rng.randrange(256)
        elif r < 0.8:
            data[i] ^= 1 << rng.randrange(8)
        else:
            j = min(body1, i + rng.randrange(1, 64))
            data[i:j] = bytes([rng.choice([0, 255])]) * (j - i)
    struct.pack_into("<Q", data, off, xxhash.xxh64(bytes(data[body0:body1]), seed=HASHER_SEED).intdigest())
    if rng.random() < 0.1:
        data = data[:rng.randrange(16, len(data))]
    open(os.path.join(out, "m%05d.glaze" % it), "wb").write(data)
