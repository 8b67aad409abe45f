#!/usr/bin/env python3
"""Writes the golden fixtures under tests/golden/ from the CPU oracle.

The reference (Rust) cannot be built here, so the vectors come from the oracle,
which is itself pinned by the reference's known-answer tests (oracle/kat_tests.cpp).
Run from the repository root after `make -C oracle`:  python tests/golden/make_golden.py

Outputs (numpy .npz, no pickles):
  scene_<name>.npz   small render of each shipped scene: rgb64, hit_leaf, ray counts
  allprims.json      a scene touching every primitive type, every JSON pattern, nested and
                     transformed groups, a plane inside a group, nested glass (containers)
  allprims.npz       its small render plus colours / leaves / distances of seeded random rays
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle as orc  # noqa: E402
from scenegen import matmul, rot_x, rot_y, rot_z, scaling, translation  # noqa: E402

SIZES = {"example": (64, 36), "egg": (64, 36), "glass": (64, 36), "pattern": (64, 36), "teapot": (48, 27)}


def allprims_scene():
    def m(*ms):
        out = ms[0]
        for x in ms[1:]:
            out = matmul(out, x)
        return out

    def mat(color, **kw):
        d = {"color": list(color)}
        d.update(kw)
        return d

    shapes = [
        {"ty": "plane", "transform": translation(0, -1, 0),
         "material": mat((1, 1, 1), reflective=0.2,
                         pattern={"ty": "checker", "primary_color": [0.9, 0.9, 0.9], "secondary_color": [0.2, 0.2, 0.3],
                                  "transform": scaling(1.5, 1.5, 1.5)})},
        {"ty": "cube", "transform": m(translation(-4, 0, 2), rot_y(0.6), scaling(0.8, 1.0, 0.6)),
         "material": mat((1, 0.3, 0.2), pattern={"ty": "stripe", "primary_color": [1, 0.3, 0.2],
                                                "secondary_color": [0.2, 0.3, 1], "transform": scaling(0.25, 1, 1)})},
        {"ty": "cylinder", "transform": m(translation(7, 0, 9), scaling(0.4, 1, 0.4)),
         "material": mat((0.3, 0.8, 0.3))},
        {"ty": "bounded_cylinder", "transform": m(translation(-1.8, 0, 1.5), rot_z(0.4), scaling(0.5, 1.0, 0.5)),
         "material": mat((0.8, 0.8, 0.1), pattern={"ty": "ring", "transform": scaling(0.2, 0.2, 0.2)})},
        {"ty": "capped_cylinder", "transform": m(translation(0.2, -0.2, 3.5), rot_x(0.5), scaling(0.6, 0.8, 0.6)),
         "material": mat((0.1, 0.6, 0.9), pattern={"ty": "gradient", "primary_color": [0.1, 0.6, 0.9],
                                                  "secondary_color": [0.9, 0.2, 0.6]})},
        {"ty": "bounded_cone", "transform": m(translation(2.2, -1, 1.0), scaling(0.9, 1.6, 0.9)),
         "material": mat((0.9, 0.5, 0.1))},
        {"ty": "bounded_dn_cone", "transform": m(translation(4.2, 0.2, 3.0), rot_x(-0.3), scaling(0.7, 1.0, 0.7)),
         "material": mat((0.6, 0.2, 0.8), pattern={"ty": "point"})},
        {"ty": "capped_cone", "transform": m(translation(-3.0, -1, -1.5), scaling(0.8, 1.2, 0.8)),
         "material": mat((0.2, 0.9, 0.7), pattern={"ty": "identity", "primary_color": [0.2, 0.9, 0.7]})},
        {"ty": "capped_dn_cone", "transform": m(translation(5.5, 0.0, -0.5), rot_z(0.8), scaling(0.5, 0.9, 0.5)),
         "material": mat((0.9, 0.9, 0.9), pattern={"ty": "null"}, ambient=0.3)},
        # glass sphere with a smaller air bubble inside: exercises the container logic behind n1 / n2
        {"ty": "sphere", "transform": m(translation(0.5, 0.5, -1.0), scaling(1.2, 1.2, 1.2)),
         "material": mat((0.1, 0.1, 0.15), ambient=0.0, diffuse=0.1, specular=1.0, shininess=300.0,
                         reflective=0.8, transparency=0.9, refractive_index=1.5)},
        {"ty": "sphere", "transform": m(translation(0.5, 0.5, -1.0), scaling(0.5, 0.5, 0.5)),
         "material": mat((1, 1, 1), ambient=0.0, diffuse=0.0, specular=0.5, transparency=1.0, refractive_index=1.0003)},
        # transformed group holding a transformed group, a plane and an empty shape
        {"ty": "group", "transform": m(translation(-1.0, 1.8, 4.0), rot_y(0.3)),
         "children": [
             {"ty": "sphere", "transform": m(translation(-1.2, 0, 0), scaling(0.5, 0.5, 0.5)),
              "material": mat((1, 0.2, 0.2), reflective=0.3)},
             {"ty": "group", "transform": m(translation(1.2, 0, 0), scaling(0.6, 0.6, 0.6)),
              "children": [
                  {"ty": "cube", "transform": rot_z(0.4), "material": mat((0.2, 1, 0.2))},
                  {"ty": "sphere", "transform": translation(0, 1.6, 0), "material": mat((0.2, 0.2, 1))},
              ]},
             {"ty": "empty"},
         ]},
        {"ty": "group", "children": []},
    ]
    return {
        "canvas_width": 96, "canvas_height": 54, "field_of_view": 1.1,
        "camera_from": [1.0, 2.5, -9.0], "camera_to": [0.5, 0.3, 0.0], "camera_up": [0.0, 1.0, 0.0],
        "light": {"intensity": [1.0, 0.95, 0.9], "position": [-6.0, 9.0, -8.0, 1.0]},
        "shapes": shapes,
    }


def random_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-7.0, 7.0, (n, 3))
    o[:, 1] = rng.uniform(-0.5, 6.0, n)
    target = rng.uniform(-4.0, 4.0, (n, 3))
    target[:, 1] = rng.uniform(-1.0, 2.0, n)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # a few axis-parallel rays: the epsilon branches of cubes, cylinders, planes and group gates
    o[:6] = [[0.5, 0.5, -8], [0.5, 8, -1], [-8, 0.5, -1], [-4, 0.2, -8], [7, 0.3, -8], [2.2, 6, 1.0]]
    d[:6] = [[0, 0, 1], [0, -1, 0], [1, 0, 0], [0, 0, 1], [0, 0, 1], [0, -1, 0]]
    return np.concatenate([o, d], axis=1)


def main():
    for name, size in SIZES.items():
        s = orc.OracleScene.from_file(os.path.join(ROOT, "scenes", name + ".json"), base_dir=ROOT)
        s.set_canvas(*size)
        r = s.render(jobs=8)
        np.savez_compressed(os.path.join(HERE, "scene_%s.npz" % name), rgb64=r["rgb64"], hit_leaf=r["hit_leaf"],
                            counts=np.array(r["counts"], np.uint64), size=np.array(size))
        print(name, size, r["counts"])
    scene = allprims_scene()
    with open(os.path.join(HERE, "allprims.json"), "w") as f:
        json.dump(scene, f, indent=1)
        f.write("\n")
    s = orc.OracleScene(json.dumps(scene), ROOT)
    r = s.render(jobs=8)
    rays = random_rays(600, 11)
    rgb, leaf, counts = s.color_at(rays)
    t, leaf2 = s.intersect_hit(rays)
    assert np.array_equal(leaf, leaf2)
    np.savez_compressed(os.path.join(HERE, "allprims.npz"), rgb64=r["rgb64"], hit_leaf=r["hit_leaf"],
                        counts=np.array(r["counts"], np.uint64), rays=rays, ray_rgb=rgb, ray_leaf=leaf, ray_t=t,
                        ray_counts=np.array(counts, np.uint64))
    print("allprims", r["counts"], "rays", counts, "leaf histogram", np.bincount(leaf[leaf >= 0]))


if __name__ == "__main__":
    main()
