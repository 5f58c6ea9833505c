#!/usr/bin/env python3
"""BASELINE configs[2] alone (4 x 1024^2, LAMA_PREC_F16 beside the fp32-accurate split): bench.configs2_leg without the rest of bench.py.
usage: [LAMA_HIP_LIB=other/liblama_hip.so] python tools/fp16_leg.py   -- one JSON line (same-box A/B of two builds of the library)"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lama_amd import _lib as L
device = torch.device('cuda', 0)
torch.cuda.set_device(device)
lib = L.get_lib()
model = bench.build_model(device, L.PREC_F16X3)
print(json.dumps(bench.configs2_leg(model, device, lib, argparse.Namespace(no_graph=False))), flush=True)
