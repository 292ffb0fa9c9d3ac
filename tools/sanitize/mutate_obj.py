"""Writes N damaged Wavefront files (with their .mtl) into DIR: the cube of tests/golden with lines dropped, doubled, shuffled, tokens
replaced by junk, indices out of range / negative / zero, huge and non-finite numbers, missing material files.
    python tools/sanitize/mutate_obj.py SEED N DIR"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
seed, n, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
obj = open(os.path.join(ROOT, "tests", "golden", "cube.obj")).read().splitlines()
mtl = open(os.path.join(ROOT, "tests", "golden", "cube.mtl")).read().splitlines()
rng = random.Random(seed)
JUNK = ["", "nan", "inf", "-inf", "1e999", "-1e-999", "0", "-1", "99999999999", "4294967296", "1/", "/1", "1//", "//", "1/2/3/4", "a", "#", "\t", "0x10", "1.5.2",
        "-0", "1/0/0", "-1/-1/-1", "2147483648", "v", "f", "usemtl", "mtllib", "\x00", "é"]


def damage(lines):
    lines = list(lines)
    for _ in range(rng.randint(1, 6)):
        if not lines:
            break
        i = rng.randrange(len(lines))
        r = rng.random()
        if r < 0.2:
            del lines[i]
        elif r < 0.35:
            lines.insert(i, lines[rng.randrange(len(lines))])
        elif r < 0.45:
            rng.shuffle(lines)
        else:
            tok = lines[i].split(" ")
            j = rng.randrange(len(tok))
            tok[j] = rng.choice(JUNK) if rng.random() < 0.7 else tok[j] * rng.randint(2, 40)
            if rng.random() < 0.2:
                tok += [rng.choice(JUNK)] * rng.randint(1, 5)
            lines[i] = " ".join(tok)
    return lines


for it in range(n):
    base = os.path.join(out, "m%05d" % it)
    o = damage(obj)
    if rng.random() < 0.8:
        o = [l.replace("cube.mtl", os.path.basename(base) + ".mtl") for l in o]
        if rng.random() < 0.9:
            open(base + ".mtl", "w", errors="ignore").write("\n".join(damage(mtl) if rng.random() < 0.7 else mtl) + "\n")
    data = ("\n".join(o) + ("\n" if rng.random() < 0.9 else "")).encode("utf8", "ignore")
    if rng.random() < 0.1:
        data = data[:rng.randrange(len(data) + 1)]
    open(base + ".obj", "wb").write(data)
