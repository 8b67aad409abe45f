#!/usr/bin/env python3
"""Deterministic scene and model generators for the render path.

Writes
  * the five scene descriptions the reference ships (same numbers, own
    formatting) into ``scenes/``,
  * ``models/teapot.obj`` -- the reference's ``scenes/teapot.json`` names this
    file but the reference does not ship it, so a teapot is tessellated here
    from Newell's profile curves (body, lid, bottom: surfaces of revolution;
    handle and spout: tubes swept between Newell's boundary curves),
  * the two synthetic many-primitive configurations (a height-field triangle
    mesh and a field of transformed spheres / cylinders / cones), at any size.

Everything is a pure function of its arguments (splitmix64 streams), so the
files can be regenerated bit-identically anywhere.
"""
import argparse
import json
import math
import os

MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * ((self.next() >> 11) * (1.0 / (1 << 53)))


IDENTITY = [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]


def translation(x, y, z):
    return [1.0, 0.0, 0.0, x, 0.0, 1.0, 0.0, y, 0.0, 0.0, 1.0, z, 0.0, 0.0, 0.0, 1.0]


def scaling(x, y, z):
    return [x, 0.0, 0.0, 0.0, 0.0, y, 0.0, 0.0, 0.0, 0.0, z, 0.0, 0.0, 0.0, 0.0, 1.0]


def matmul(a, b):
    return [sum(a[r * 4 + k] * b[k * 4 + c] for k in range(4)) for r in range(4) for c in range(4)]


def rot_x(r):
    c, s = math.cos(r), math.sin(r)
    return [1.0, 0.0, 0.0, 0.0, 0.0, c, -s, 0.0, 0.0, s, c, 0.0, 0.0, 0.0, 0.0, 1.0]


def rot_y(r):
    c, s = math.cos(r), math.sin(r)
    return [c, 0.0, s, 0.0, 0.0, 1.0, 0.0, 0.0, -s, 0.0, c, 0.0, 0.0, 0.0, 0.0, 1.0]


def rot_z(r):
    c, s = math.cos(r), math.sin(r)
    return [c, -s, 0.0, 0.0, s, c, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]


def base_scene(width, height, cam_from, light_intensity, light_pos, shapes, cam_to=(0.0, 1.0, 0.0), fov=1.0472):
    return {
        "canvas_width": width,
        "canvas_height": height,
        "field_of_view": fov,
        "camera_from": list(cam_from),
        "camera_to": list(cam_to),
        "camera_up": [0.0, 1.0, 0.0],
        "light": {"intensity": list(light_intensity), "position": list(light_pos)},
        "shapes": shapes,
    }


WARM = (0.85, 0.8, 0.65)
LIGHT_POS = (-10.0, 10.0, -10.0, 1.0)


def reference_scenes():
    """The five scenes under the reference's scenes/ directory, value for value."""
    floor_m2 = {"ty": "plane", "transform": translation(0.0, -2.0, 0.0)}
    out = {}
    out["example"] = base_scene(1920, 1080, (0.0, 1.5, -10.0), WARM, LIGHT_POS,
                                [{"ty": "sphere", "transform": IDENTITY}, floor_m2])
    out["egg"] = base_scene(1920, 1080, (0.0, 1.5, -5.0), WARM, LIGHT_POS, [
        {"ty": "sphere", "transform": IDENTITY, "material": {"color": [1.0, 1.0, 0.0]}},
        {"ty": "plane", "transform": IDENTITY},
    ])
    out["glass"] = base_scene(3840, 2160, (0.0, 1.5, -18.0), (1.0, 1.0, 1.0), (0.0, 10.0, -20.0, 1.0), [
        {"ty": "sphere",
         "transform": [1.5, 0.0, 0.0, 0.0, 0.0, 1.5, 0.0, 1.0, 0.0, 0.0, 1.5, 0.0, 0.0, 0.0, 0.0, 1.0],
         "material": {"color": [0.4, 0.4, 0.6], "ambient": 0.0, "diffuse": 0.2, "specular": 1.0,
                      "refractive_index": 1.5, "reflective": 0.7, "transparency": 0.7}},
        {"ty": "cube", "transform": translation(-5.0, 0.0, 10.0), "material": {"color": [1.0, 0.0, 0.0]}},
        {"ty": "cube", "transform": translation(4.0, 4.0, 4.0), "material": {"color": [1.0, 1.0, 0.0]}},
        {"ty": "plane", "transform": translation(0.0, -2.0, 0.0),
         "material": {"color": [1.0, 1.0, 1.0],
                      "pattern": {"ty": "checker", "primary_color": [0.0, 1.0, 0.0],
                                  "secondary_color": [0.0, 0.0, 1.0]}}},
    ])
    out["pattern"] = base_scene(960, 540, (0.0, 1.5, -10.0), WARM, LIGHT_POS, [
        {"ty": "cube",
         "transform": [0.7071, 0.0, 0.7071, 0.0, 0.0, 1.0, 0.0, 2.0, -0.7071, 0.0, 0.7071, 0.0, 0.0, 0.0, 0.0, 1.0]},
        {"ty": "plane",
         "material": {"color": [1.0, 1.0, 1.0],
                      "pattern": {"ty": "checker", "primary_color": [1.0, 0.0, 0.0],
                                  "secondary_color": [0.0, 0.0, 1.0]}}},
    ])
    out["teapot"] = base_scene(960, 540, (0.0, 1.5, -10.0), WARM, LIGHT_POS, [
        {"ty": "model", "path": "models/teapot.obj", "transform": IDENTITY},
        floor_m2,
    ])
    return out


# ------------------------------------------------------------------ teapot
def bezier(p, t):
    u = 1.0 - t
    b = (u * u * u, 3 * u * u * t, 3 * u * t * t, t * t * t)
    return tuple(sum(b[i] * p[i][k] for i in range(4)) for k in range(len(p[0])))


class Mesh:
    def __init__(self):
        self.v, self.faces = [], []  # faces: (group, a, b, c), zero-based

    def add_vertex(self, p):
        self.v.append(p)
        return len(self.v) - 1

    def tri(self, group, a, b, c):
        pa, pb, pc = self.v[a], self.v[b], self.v[c]
        e1 = [pb[i] - pa[i] for i in range(3)]
        e2 = [pc[i] - pa[i] for i in range(3)]
        n = (e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0])
        if n[0] * n[0] + n[1] * n[1] + n[2] * n[2] > 1e-18:
            self.faces.append((group, a, b, c))


def grid_surface(mesh, group, fn, nu, nv, wrap_v):
    """Tessellates fn(u, v) -> (x, y, z) on an nu x nv grid; rows that collapse to one point are shared."""
    rows = []
    for i in range(nu + 1):
        u = i / nu
        pts = [fn(u, j / nv) for j in range(nv + (0 if wrap_v else 1))]
        if all(math.dist(pts[0], q) < 1e-12 for q in pts):
            idx = mesh.add_vertex(pts[0])
            rows.append([idx] * len(pts))
        else:
            rows.append([mesh.add_vertex(q) for q in pts])
    cols = nv
    for i in range(nu):
        for j in range(cols):
            j2 = (j + 1) % len(rows[i]) if wrap_v else j + 1
            a, b, c, d = rows[i][j], rows[i + 1][j], rows[i + 1][j2], rows[i][j2]
            mesh.tri(group, a, b, d)  # zero-area triangles (collapsed rows) are dropped by Mesh.tri
            mesh.tri(group, b, c, d)


def revolve(mesh, group, profile, nu, nv):
    """profile: four (radius, height) Bezier control points; y is up."""
    def fn(u, v):
        r, h = bezier(profile, u)
        a = 2.0 * math.pi * v
        return (r * math.cos(a), h, r * math.sin(a))
    grid_surface(mesh, group, fn, nu, nv, True)


def tube(mesh, group, outer, inner, half_width, nu, nv):
    """Tube between two planar curves (x, height): elliptical section, half_width(u) across."""
    def curve(segs, u):
        s = min(int(u * len(segs)), len(segs) - 1)
        return bezier(segs[s], u * len(segs) - s)

    def fn(u, v):
        (xo, ho), (xi, hi) = curve(outer, u), curve(inner, u)
        cx, ch = 0.5 * (xo + xi), 0.5 * (ho + hi)
        ax, ah = 0.5 * (xo - xi), 0.5 * (ho - hi)
        a = 2.0 * math.pi * v
        return (cx + ax * math.cos(a), ch + ah * math.cos(a), half_width(u) * math.sin(a))
    grid_surface(mesh, group, fn, nu, nv, True)


def teapot_mesh(detail=16):
    m = Mesh()
    around = 4 * detail
    revolve(m, "lid", [(0.0, 3.15), (0.8, 3.15), (0.0, 2.85), (0.2, 2.7)], detail, around)
    revolve(m, "lid", [(0.2, 2.7), (0.4, 2.55), (1.3, 2.55), (1.3, 2.4)], detail, around)
    revolve(m, "rim", [(1.4, 2.4), (1.3375, 2.53125), (1.4375, 2.53125), (1.5, 2.4)], detail // 2, around)
    revolve(m, "body", [(1.5, 2.4), (1.75, 1.875), (2.0, 1.35), (2.0, 0.9)], detail, around)
    revolve(m, "body", [(2.0, 0.9), (2.0, 0.45), (1.5, 0.225), (1.5, 0.15)], detail, around)
    revolve(m, "bottom", [(1.5, 0.15), (1.5, 0.075), (1.425, 0.0), (0.0, 0.0)], detail // 2, around)
    tube(m, "handle",
         [[(-1.5, 2.25), (-2.5, 2.25), (-3.0, 2.25), (-3.0, 1.8)], [(-3.0, 1.8), (-3.0, 1.35), (-2.65, 0.9375), (-1.9, 0.6)]],
         [[(-1.6, 2.025), (-2.3, 2.025), (-2.7, 2.025), (-2.7, 1.8)], [(-2.7, 1.8), (-2.7, 1.575), (-2.5, 1.125), (-2.0, 0.9)]],
         lambda u: 0.3, 2 * detail, detail)
    tube(m, "spout",
         [[(1.7, 0.6), (3.1, 0.825), (2.4, 2.025), (2.7, 2.4)], [(2.7, 2.4), (2.8, 2.475), (2.9, 2.475), (2.8, 2.4)]],
         [[(1.7, 1.425), (2.6, 1.425), (2.3, 2.1), (2.7, 2.4 - 1e-3)], [(2.7, 2.4 - 1e-3), (3.3, 2.4), (3.525, 2.49375), (3.2, 2.4)]],
         lambda u: 0.66 - 0.41 * min(1.0, 2.0 * u) if u < 0.5 else 0.25 - 0.1 * (2.0 * u - 1.0), 2 * detail, detail)
    return m


def write_obj(mesh, path, smooth=True):
    nv = len(mesh.v)
    normals = [[0.0, 0.0, 0.0] for _ in range(nv)]
    for (_, a, b, c) in mesh.faces:
        pa, pb, pc = mesh.v[a], mesh.v[b], mesh.v[c]
        e1 = [pb[i] - pa[i] for i in range(3)]
        e2 = [pc[i] - pa[i] for i in range(3)]
        n = (e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0])
        for k in (a, b, c):
            for i in range(3):
                normals[k][i] += n[i]
    lines = ["# generated by tools/scenegen.py (teapot from Newell's profile curves)"]
    for p in mesh.v:
        lines.append("v %r %r %r" % (p[0], p[1], p[2]))
    if smooth:
        for n in normals:
            l = math.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) or 1.0
            lines.append("vn %r %r %r" % (n[0] / l, n[1] / l, n[2] / l))
    current = None
    for (g, a, b, c) in mesh.faces:
        if g != current:
            lines.append("g %s" % g)
            current = g
        if smooth:
            lines.append("f %d//%d %d//%d %d//%d" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1))
        else:
            lines.append("f %d %d %d" % (a + 1, b + 1, c + 1))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return len(mesh.faces)


# ------------------------------------------------------- synthetic configs
def heightfield(out_dir, cells=708, width=3840, height=2160, seed=0x5EED0001, name="heightfield"):
    """cells x cells grid of flat triangle pairs over [-50, 50]^2, heights +-2.5 (5 % of the extent)."""
    rng = SplitMix64(seed)
    n = cells + 1
    extent, amp = 100.0, 2.5
    obj_path = os.path.join(out_dir, name + ".obj")
    with open(obj_path, "w") as f:
        f.write("# generated by tools/scenegen.py heightfield cells=%d seed=%#x\n" % (cells, seed))
        buf = []
        for j in range(n):
            z = -0.5 * extent + extent * j / cells
            for i in range(n):
                x = -0.5 * extent + extent * i / cells
                buf.append("v %r %r %r" % (x, rng.uniform(-amp, amp), z))
        f.write("\n".join(buf) + "\n")
        buf = []
        for j in range(cells):
            for i in range(cells):
                a = j * n + i + 1
                b, c, d = a + 1, a + n, a + n + 1
                buf.append("f %d %d %d" % (a, c, b))
                buf.append("f %d %d %d" % (b, c, d))
        f.write("\n".join(buf) + "\n")
    scene = base_scene(width, height, (0.0, 45.0, -85.0), (1.0, 1.0, 1.0), (-40.0, 80.0, -60.0, 1.0), [
        {"ty": "model", "path": obj_path, "transform": IDENTITY},
        {"ty": "plane", "transform": translation(0.0, -4.0, 0.0), "material": {"color": [0.6, 0.7, 0.6]}},
    ], cam_to=(0.0, 0.0, 0.0))
    path = os.path.join(out_dir, name + ".json")
    with open(path, "w") as f:
        json.dump(scene, f)
    return path, 2 * cells * cells


def sweep_camera(frame, frames=60):
    a = 2.0 * math.pi * frame / frames
    return (90.0 * math.cos(a), 30.0, 90.0 * math.sin(a)), (0.0, 5.0, 0.0)


def primitives(out_dir, count=100000, width=7680, height=4320, seed=0x5EED0002, frame=0, name="primitives"):
    """count top-level primitives: 50 % sphere, 25 % capped_cylinder, 25 % bounded_cone, plus a floor."""
    rng = SplitMix64(seed)
    shapes = []
    for _ in range(count):
        k = rng.uniform()
        ty = "sphere" if k < 0.5 else ("capped_cylinder" if k < 0.75 else "bounded_cone")
        cx, cy, cz = rng.uniform(-60.0, 60.0), rng.uniform(0.0, 25.0), rng.uniform(-60.0, 60.0)
        sx, sy, sz = rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0)
        ax, ay, az = (rng.uniform(0.0, 2.0 * math.pi) for _ in range(3))
        m = matmul(translation(cx, cy, cz), matmul(matmul(rot_x(ax), matmul(rot_y(ay), rot_z(az))), scaling(sx, sy, sz)))
        mat = {"color": [rng.uniform(), rng.uniform(), rng.uniform()]}
        if rng.uniform() < 0.1:
            mat["reflective"] = 0.3
        shapes.append({"ty": ty, "transform": m, "material": mat})
    shapes.append({"ty": "plane", "transform": translation(0.0, -1.0, 0.0), "material": {"color": [0.8, 0.8, 0.8]}})
    cam_from, cam_to = sweep_camera(frame)
    scene = base_scene(width, height, cam_from, (1.0, 1.0, 1.0), (-80.0, 120.0, -60.0, 1.0), shapes, cam_to=cam_to)
    path = os.path.join(out_dir, name + ".json")
    with open(path, "w") as f:
        json.dump(scene, f)
    return path, count + 1


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--root", default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ap.add_argument("--heightfield", type=int, metavar="CELLS", help="also write gen/heightfield.{json,obj}")
    ap.add_argument("--primitives", type=int, metavar="COUNT", help="also write gen/primitives.json")
    args = ap.parse_args()
    os.makedirs(os.path.join(args.root, "scenes"), exist_ok=True)
    os.makedirs(os.path.join(args.root, "models"), exist_ok=True)
    for name, scene in reference_scenes().items():
        with open(os.path.join(args.root, "scenes", name + ".json"), "w") as f:
            json.dump(scene, f, indent=1)
            f.write("\n")
    tris = write_obj(teapot_mesh(), os.path.join(args.root, "models", "teapot.obj"))
    print("teapot: %d triangles" % tris)
    gen = os.path.join(args.root, "gen")
    if args.heightfield:
        os.makedirs(gen, exist_ok=True)
        print(heightfield(gen, args.heightfield))
    if args.primitives:
        os.makedirs(gen, exist_ok=True)
        print(primitives(gen, args.primitives))


if __name__ == "__main__":
    main()
