"""Extracts the triangle soups of the reference's two sample meshes (Content/models/teapot.dae, sphere.dae) into
tests/golden/*.npz, so that tests and benchmarks can use them on machines that do not have the reference tree.

The arrays are DATA (vertex positions and normals per triangle corner), produced the way the reference's loader
hands meshes to the tracer (ModelLoader.cs:16-64: Assimp import with PreTransformVertices, then one Tri per face
with the corner normals re-normalised): COLLADA <triangles> with VERTEX/NORMAL inputs, the instancing node's
<matrix>, and Assimp's Z_UP -> Y_UP root rotation (x, y, z) -> (x, z, -y).  Assimp itself is not available, so the
float rounding of its matrix products is not reproduced bit for bit (parity unpinned for the importer): transforms
are evaluated in double and rounded once.

Run from the repository root:  python tests/golden/make_mesh_fixtures.py [/root/reference]
"""
import os
import re
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NS = {"c": "http://www.collada.org/2005/11/COLLADASchema"}


def load_dae(path):
    root = ET.parse(path).getroot()
    up = root.find("c:asset/c:up_axis", NS)
    up = up.text.strip() if up is not None else "Y_UP"
    geoms = {}
    for g in root.findall("c:library_geometries/c:geometry", NS):
        mesh = g.find("c:mesh", NS)
        sources = {}
        for s in mesh.findall("c:source", NS):
            fa = s.find("c:float_array", NS)
            stride = int(s.find("c:technique_common/c:accessor", NS).get("stride"))
            sources["#" + s.get("id")] = np.array(fa.text.split(), dtype=np.float64).reshape(-1, stride)
        verts = mesh.find("c:vertices", NS)
        vmap = {"#" + verts.get("id"): verts.find("c:input[@semantic='POSITION']", NS).get("source")}
        pos_all, nrm_all = [], []
        for tris in mesh.findall("c:triangles", NS):
            inputs = {i.get("semantic"): (int(i.get("offset")), i.get("source")) for i in tris.findall("c:input", NS)}
            n_off = max(o for o, _ in inputs.values()) + 1
            idx = np.array(tris.find("c:p", NS).text.split(), dtype=np.int64).reshape(-1, 3, n_off)
            voff, vsrc = inputs["VERTEX"]
            noff, nsrc = inputs["NORMAL"]
            pos_all.append(sources[vmap[vsrc]][idx[:, :, voff]])
            nrm_all.append(sources[nsrc][idx[:, :, noff]])
        geoms["#" + g.get("id")] = (np.concatenate(pos_all), np.concatenate(nrm_all))
    out_p, out_n = [], []
    for node in root.findall("c:library_visual_scenes/c:visual_scene/c:node", NS):
        inst = node.find("c:instance_geometry", NS)
        if inst is None:
            continue
        m = node.find("c:matrix", NS)
        M = np.array(m.text.split(), dtype=np.float64).reshape(4, 4) if m is not None else np.eye(4)
        if up == "Z_UP":
            R = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
            M = R @ M
        p, n = geoms[inst.get("url")]
        p = p @ M[:3, :3].T + M[:3, 3]
        n = n @ np.linalg.inv(M[:3, :3])          # inverse transpose applied to row vectors
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        out_p.append(p)
        out_n.append(n)
    p = np.concatenate(out_p).astype(np.float32).reshape(-1, 9)
    n = np.concatenate(out_n).astype(np.float32).reshape(-1, 9)
    return p, n


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    for name in ("teapot", "sphere"):
        p, n = load_dae(os.path.join(ref, "Content", "models", name + ".dae"))
        np.savez_compressed(os.path.join(HERE, name + "_dae.npz"), pos9=p, nrm9=n)
        print(name, p.shape, "bounds", p.reshape(-1, 3).min(0), p.reshape(-1, 3).max(0))
