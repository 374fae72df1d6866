#!/usr/bin/env python3
"""Static report on the device code of one kernel: registers, spills, and -- per basic block that contains vector-memory loads or
waits -- the loads and the s_waitcnt vmcnt values, so that a change that makes the compiler drain the software pipeline
(`vmcnt(0)` inside the pipelined loop of march_p2_kernel) is seen without a GPU.

    hipcc <flags of build.py> --cuda-device-only -S -o /tmp/vr_api.s volumerendering_amd/csrc/vr_api.hip
    python tools/isa_report.py /tmp/vr_api.s march_p2_kernelILi1ELb1E
"""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    names = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l]
    for start in names:
        name = lines[start].split(":")[0]
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = lines[start:end]
        meta = {}
        for l in lines:
            m = re.match(r"\s*\.set " + re.escape(name) + r"\.(\w+), (\d+)", l)
            if m:
                meta[m.group(1)] = int(m.group(2))
        spills = sum(1 for l in body if "scratch_store" in l or "scratch_load" in l)
        n_inst = sum(1 for l in body if re.match(r"\s+[sv]_|\s+buffer_|\s+global_|\s+ds_|\s+flat_|\s+scratch_", l))
        print(f"== {name}\n   vgpr {meta.get('num_vgpr')} sgpr {meta.get('numbered_sgpr')} scratch-insts {spills} insts {n_inst}")
        # blocks
        blk, cur = [], ["<entry>", []]
        for l in body[1:]:
            if re.match(r"^\.LBB\d+_\d+:", l):
                blk.append(cur)
                cur = [l.split(":")[0], []]
            else:
                cur[1].append(l)
        blk.append(cur)
        for label, ls in blk:
            loads = [l for l in ls if re.search(r"buffer_load|global_load", l)]
            waits = [re.search(r"vmcnt\((\d+)\)", l).group(1) for l in ls if "vmcnt(" in l]
            valu = sum(1 for l in ls if re.match(r"\s+v_", l))
            salu = sum(1 for l in ls if re.match(r"\s+s_", l))
            if len(loads) >= 4 or (waits and valu > 40):
                print(f"   {label:12s} valu {valu:4d} salu {salu:4d} vmem-loads {len(loads):3d} vmcnt waits: {' '.join(waits)}")


if __name__ == "__main__":
    main()
