#!/usr/bin/env python3
"""How many distinct side-by-side streams does vr_stream() find on this box?  (with and without torch initialised first)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.init()
    torch.zeros(4, device="cuda")
from volumerendering_amd import capi
with capi.Context(256, 256, 0) as ctx:
    print([hex(ctx.stream(i)) for i in range(4)])
