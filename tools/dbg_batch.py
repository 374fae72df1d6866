import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from volumerendering_amd import capi, host, workloads as wl
wk = sys.argv[1] if len(sys.argv) > 1 else "C2"
n, W, H, vname = wl.WORKLOADS[wk]
app = host.Application(W, H, 0)
variant, vols = wl.build_scene(app, wk, "default")
ctx = app.context()
fl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx.set_kernel_flavour(fl)
ref = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
outs = [torch.full((H, W, 4), -7.0, dtype=torch.float32, device="cuda") for _ in range(4)]
s = ctx.stream(0)
for _ in range(6):
    ctx.render_async(variant, ref.data_ptr(), s)
torch.cuda.synchronize()
u = app.uniforms()
for nf in (1, 2, 3, 4):
    for o in outs:
        o.fill_(-7.0)
    torch.cuda.synchronize()
    for _ in range(6):
        ctx.render_batch_async(variant, [u] * nf, [o.data_ptr() for o in outs[:nf]], s)
    torch.cuda.synchronize()
    print("ran flavour", ctx.last_kernel_flavour())
    for f in range(nf):
        d = (outs[f].view(torch.int32) != ref.view(torch.int32)).any(dim=2)
        nd = int(d.sum().item())
        msg = ""
        if nd:
            ys, xs = torch.nonzero(d, as_tuple=True)
            unwritten = int((outs[f][ys, xs, 0] == -7.0).sum().item())
            msg = f" first at (x={int(xs[0])}, y={int(ys[0])}) x range {int(xs.min())}..{int(xs.max())} y range {int(ys.min())}..{int(ys.max())} unwritten {unwritten}"
        print(f"n_frames {nf} frame {f}: {nd} differing pixels{msg}")
