#!/usr/bin/env python3
"""How often tier 1 of the pixel selection leaves a lane / a wave-voxel undecided: an experiment build (-DDMI_T1_DEBUG) adds
those counts to the per-view hit counters; the difference to the shipped library's counters is the answer."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import upload_scene
from cudadepthmapintegration_amd import build as _build, capi, scene
from tools.gpu_exp import load_lib

def run(path, kind, n, cells, W, H):
    load_lib(path)
    grid = scene.default_grid(cells); ray = scene.default_ray_potential(grid)
    c = capi.FusionContext(grid, ray, grid_dtype="f32", count_hits=True)
    upload_scene(c, scene, kind, n, W, H, float(max(grid.spacing)))
    c.fuse(); c.synchronize()
    _, mh = c.download_hits()
    hist = c.brick_class_histogram()
    c.close()
    return np.array(mh, dtype=np.int64), hist

if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "dense"
    n, cells, W, H = 64, 256, 1280, 720
    a, hist = run(os.path.join(_build.CSRC, "libdmi_hip.so"), kind, n, cells, W, H)
    b, _ = run(os.path.join(_build.CSRC, "libdmi_hip_exp_t1dbg.so"), kind, n, cells, W, H)
    d = (b - a).sum()
    wv = d // 65536
    tk = 8
    print(json.dumps({"scene": kind, "mixed_pairs": hist["mixed"], "wave_voxels_in_mixed": hist["mixed"] * tk, "wave_voxels_undecided~": int(wv),
                      "fraction": float(wv) / (hist["mixed"] * tk), "lanes~": int(d - wv * 65536), "per_view": ((b - a) // 65536)[:16].tolist()}))
