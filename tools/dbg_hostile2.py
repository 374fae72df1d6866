import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import host_ref as hr, vrtest as vt
import test_parity_gpu as T
from volumerendering_amd import capi
f32 = np.float32
W, H, n = 64, 48, 16
step, count = hr.stepping_params(n, n, n)
u = hr.make_uniforms(W, H, steps_count=count, step_size=step)
tf = T.zero_prefix_tf(32, 3)
with capi.Context(W, H, 0) as ctx:
    for pos in [(8, 8, 8), (3, 3, 3), (4, 4, 4), (13, 3, 3), (3, 13, 13), (12, 12, 12), (15, 15, 15), (0, 0, 0), (7, 9, 2)]:
        for val in (np.nan, 0.9):
            v = np.zeros((n, n, n, 4), dtype=f32)
            v[pos[0], pos[1], pos[2], 3] = val
            res = {}
            for fl in (1, 5):
                ctx.set_kernel_flavour(fl)
                frag, _, ns = vt.gpu_render(ctx, capi.BASIC, u, [v], [tf])
                res[fl] = (frag, ns, ctx.counters())
            a, b = res[1][0], res[5][0]
            same = np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(vt.bits(a)[np.isfinite(a)], vt.bits(b)[np.isfinite(a)])
            print(f"voxel {pos} = {val}: plain nan px {int(np.isnan(a).any(axis=2).sum())} nonzero px {int((a != 0).any(axis=2).sum())}; skipping nan px {int(np.isnan(b).any(axis=2).sum())} nonzero {int((b != 0).any(axis=2).sum())}; same {same}; counters plain {res[1][2]} skip {res[5][2]}")
