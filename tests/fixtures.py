"""Shared helpers for the tests: fixture loading (tests/golden/*.npz are DATA captured from the compiled
reference by tools/gen_golden.py)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

FLAT_KEYS = ("root", "node_axis", "node_split", "node_left", "node_right", "items", "batch_recs", "batch_mats",
             "tri_recs", "tri_mats", "solid_recs", "solid_types", "solid_mats", "materials", "aabb_start", "aabb_end")
PARAM_KEYS = ("fov", "shadows", "camera_light", "max_reflect_depth", "bg_gradient_axis", "ambient", "bg1", "bg2",
              "bg3", "point_light_pos", "point_light_color", "global_light_dir", "global_light_color")
RGBX8 = [(8, 1, 0, 0), (8, 0, 1, 0), (8, 0, 0, 1), (8, 0, 0, 0)]
RGB16 = [(16, 1, 0, 0), (16, 0, 1, 0), (16, 0, 0, 1)]
RGBF32 = [(32, 1, 0, 0, 0, True), (32, 0, 1, 0, 0, True), (32, 0, 0, 1, 0, True)]

BOX_FIXTURES = ["box_n3_1920x1080", "box_n6_1920x1080", "box_n10_4096x4096", "box_n6_640x480_generic",
                "box_n5_320x200", "box_n8_320x200", "box_n12_320x200"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def flat_of(g, opaque=False):
    flat = {k: g[k] for k in FLAT_KEYS}
    flat["batch_size"] = 4
    if opaque:
        m = np.array(g["materials"], np.float32).copy()
        m[:, 6] = 1.0
        flat["materials"] = m
    return flat


def params_of(g, prefix=""):
    return {k: g[prefix + k] for k in PARAM_KEYS}


def known_answer():
    with open(os.path.join(GOLDEN, "kdtree_known_answer.json")) as f:
        return json.load(f)


def known_answer_flat(ka):
    """Flatten the hand-built scene of the reference's test_kdtree into the nt_scene_desc layout."""
    n = ka["dimension"]
    recs = []
    for t in ka["triangles"]:
        fn = np.asarray(t["face_normal"], np.float32)
        p1 = np.asarray(t["p1"], np.float32)
        d = np.float32(0)
        acc = np.float32(fn[0] * p1[0])
        for k in range(1, n):
            acc = np.float32(acc + np.float32(fn[k] * p1[k]))
        d = -acc
        rec = [d] + list(fn) + list(p1)
        for e in t["edge_normals"]:
            rec += list(np.asarray(e, np.float32))
        recs.append(rec)
    nodes, items = [], []

    def add(node):
        if node is None:
            return -1
        idx = len(nodes)
        nodes.append(None)
        if "leaf" in node:
            start = len(items)
            items.extend((i << 2) | 1 for i in node["leaf"])
            nodes[idx] = (-1, 0.0, start, len(node["leaf"]))
        else:
            b = node["branch"]
            l = add(b["left"])
            r = add(b["right"])
            nodes[idx] = (b["axis"], b["split"], l, r)
        return idx

    root = add(ka["tree"])
    nd = np.asarray(nodes, np.float64)
    rl = n * n + n + 1
    return dict(root=root, node_axis=nd[:, 0].astype(np.int32), node_split=nd[:, 1].astype(np.float32),
                node_left=nd[:, 2].astype(np.int32), node_right=nd[:, 3].astype(np.int32),
                items=np.asarray(items, np.int32), batch_recs=np.zeros((0, 4, rl), np.float32),
                batch_mats=np.zeros((0, 4), np.int32), tri_recs=np.asarray(recs, np.float32),
                tri_mats=np.zeros(len(recs), np.int32), solid_recs=np.zeros((0, 2 * n * n + n), np.float32),
                solid_types=np.zeros(0, np.int32), solid_mats=np.zeros(0, np.int32),
                materials=np.asarray([[1, 1, 1, 1, 1, 1, 1, 0, 1, 8]], np.float32),
                aabb_start=np.asarray(ka["aabb"]["start"], np.float32), aabb_end=np.asarray(ka["aabb"]["end"], np.float32),
                batch_size=4)
