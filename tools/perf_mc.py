"""Meshing workload alone (marching cubes of 32 level grids at vox_res = 100): python tools/perf_mc.py"""
import importlib.util, json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("w", os.path.join(ROOT, "tools", "workloads.py"))
w = importlib.util.module_from_spec(spec)
spec.loader.exec_module(w)
print(json.dumps(w.marching_cubes_100()))
