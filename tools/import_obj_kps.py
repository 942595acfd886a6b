#!/usr/bin/env python3
"""Pack the per-object keypoint fixtures of the reference's dataset config into one npz.

Reads (in the build container only; /root/reference does not exist on the GPU box):
  pvn3d/datasets/linemod/lm_obj_kps/<obj>/{farthest.txt,corners.txt}   (13 objects)
  pvn3d/datasets/ycb/ycb_object_kps/<obj>/{farthest.txt,corners.txt}   (21 objects)
  pvn3d/datasets/ycb/dataset_config/{classes.txt,radius.txt}
  pvn3d/datasets/linemod/dataset_config/models_info.yml                (object diameters, mm)
which are what Basic_Utils.get_kps / get_ctr (pvn3d/lib/utils/basic_utils.py:541-595) and
Config.ycb_r_lst (pvn3d/common.py:80) load.  Output: pvn3d_amd/data/obj_kps.npz
"""
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pvn3d_amd", "data", "obj_kps.npz")

LM_OBJ = {'ape': 1, 'benchvise': 2, 'cam': 4, 'can': 5, 'cat': 6, 'driller': 8, 'duck': 9,
          'eggbox': 10, 'glue': 11, 'holepuncher': 12, 'iron': 13, 'lamp': 14, 'phone': 15}

out = {}
lm_dir = os.path.join(REF, "pvn3d/datasets/linemod/lm_obj_kps")
for name in LM_OBJ:
    out["lm/%s/farthest" % name] = np.loadtxt(os.path.join(lm_dir, name, "farthest.txt"), dtype=np.float32)
    out["lm/%s/corners" % name] = np.loadtxt(os.path.join(lm_dir, name, "corners.txt"), dtype=np.float32)
ycb_dir = os.path.join(REF, "pvn3d/datasets/ycb/ycb_object_kps")
with open(os.path.join(REF, "pvn3d/datasets/ycb/dataset_config/classes.txt")) as f:
    ycb_cls = [l.strip() for l in f.readlines() if l.strip()]
for name in ycb_cls:
    out["ycb/%s/farthest" % name] = np.loadtxt(os.path.join(ycb_dir, name, "farthest.txt"), dtype=np.float32)
    out["ycb/%s/corners" % name] = np.loadtxt(os.path.join(ycb_dir, name, "corners.txt"), dtype=np.float32)
out["ycb_classes"] = np.array(ycb_cls)
out["ycb_radius"] = np.loadtxt(os.path.join(REF, "pvn3d/datasets/ycb/dataset_config/radius.txt")).astype(np.float64)
import yaml
with open(os.path.join(REF, "pvn3d/datasets/linemod/dataset_config/models_info.yml")) as f:
    info = yaml.safe_load(f)                       # Config.lm_r_lst (pvn3d/common.py:131-133)
out["lm_diameter_ids"] = np.array(sorted(info), dtype=np.int32)
out["lm_diameter_mm"] = np.array([info[k]["diameter"] for k in sorted(info)], dtype=np.float64)
out["lm_names"] = np.array(list(LM_OBJ.keys()))
out["lm_ids"] = np.array(list(LM_OBJ.values()), dtype=np.int32)
np.savez_compressed(OUT, **out)
print("wrote", os.path.abspath(OUT), len(out), "arrays")
