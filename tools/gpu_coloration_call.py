import json, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from cudadepthmapintegration_amd import capi, scene
pcie = capi.pcie_probe(0)
out = bench.coloration_probe(scene, capi, 2_000_000, 1280, 720, pcie=pcie)
for k in ("random_vertices", "mesh_ordered_vertices", "random_vertices_reordered_on_device"):
    d = out[k]
    print(k, d["seconds_of_five_calls"], "kernel_ms", round(d["kernel_ms"], 3), "call_ms", round(d["seconds"] * 1e3, 3), "floor_ms", round(d["pcie_floor_s"] * 1e3, 3), "ratio", round(d["seconds_over_floor_plus_kernels"], 3))
print("pcie", pcie)
