"""Generate icon_amd/data/synth_body_6890.npz (needs scipy; run once, result is committed).

The mesh is the SURVEY.md §8(d) synthetic body: watertight, genus-0, V=6,890 / F=13,776."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icon_amd import synth

if __name__ == "__main__":
    verts, faces = synth.generate_body_mesh()
    os.makedirs(os.path.dirname(synth.body_mesh_path()), exist_ok=True)
    np.savez_compressed(synth.body_mesh_path(), verts=verts, faces=faces.astype(np.int32))
    # sanity: closed 2-manifold, consistent winding
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    key = e[:, 0] * len(verts) + e[:, 1]
    rkey = e[:, 1] * len(verts) + e[:, 0]
    assert len(np.unique(key)) == len(key) and set(key) == set(rkey), "not a closed oriented manifold"
    v = verts.astype(np.float64)
    vol = (np.cross(v[faces[:, 0]], v[faces[:, 1]]) * v[faces[:, 2]]).sum() / 6
    el = np.linalg.norm(v[e[:, 0]] - v[e[:, 1]], axis=1)
    print("V,F", verts.shape, faces.shape, "volume", vol, "bbox", verts.min(0), verts.max(0))
    print("edge len min/mean/max", el.min(), el.mean(), el.max())
