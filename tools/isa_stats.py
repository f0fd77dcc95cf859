"""Compile a .hip file of csrc/ to gfx950 assembly and summarise kernels: registers, LDS, occupancy,
and the instruction mix of the hottest loop (the innermost loop with the most instructions).
Builder tool; runs in the CPU container (hipcc cross-compiles).

    python tools/isa_stats.py nb_tree.hip walk_cells_kernelILi8ELb0ELi0 [--dump]
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
dump = "--dump" in sys.argv
path = os.path.join(ROOT, "wgpu_n_body_amd", "csrc", src)
out = "/tmp/" + os.path.splitext(src)[0] + ".s"
if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(path):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                    f"-I{ROOT}/include", f"-I{ROOT}/wgpu_n_body_amd/csrc", "--cuda-device-only", "-S",
                    path, "-o", out], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()


def cls(op):
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op in ("v_sqrt_f32_e32", "v_rcp_f32_e32", "v_rsq_f32_e32", "v_sqrt_f32_e64", "v_rcp_f32_e64"):
        return "valu_trans"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "valu_lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


for m in re.finditer(r"^(_Z\S+):\s*; @\S+\n", s, re.M):
    name = m.group(1)
    if pat not in name:
        continue
    end = s.index(".Lfunc_end", m.end())
    body = s[m.end():end]
    meta = s[end:end + 4000]
    def g(k):
        r = re.search(r"\." + k + r", (\d+)", meta)
        return r.group(1) if r else "?"
    kd = s[s.index(".amdhsa_kernel " + name):]
    kd = kd[:kd.index(".end_amdhsa_kernel")]
    lds = re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", kd)
    lines = body.split("\n")
    insts = []  # (label or None, op)
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith(".") and not t.endswith(":"):
            continue
        if t.endswith(":") or re.match(r"^\.LBB\S+:", t):
            insts.append((t.split(":")[0], None))
            continue
        insts.append((None, t.split()[0]))
    print(f"{name[:110]}\n  vgpr {g('num_vgpr')} sgpr {g('numbered_sgpr')} scratch {g('private_seg_size')} "
          f"lds {lds.group(1) if lds else '?'}  instructions {sum(1 for l, o in insts if o)}")
    # loops: a backward branch to a label
    labels = {l: i for i, (l, o) in enumerate(insts) if l}
    loops = []
    for i, ln in enumerate(lines):
        pass
    idx = 0
    flat = []
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith(";"):
            continue
        if re.match(r"^\.LBB\S+:", t):
            flat.append(("L", t.split(":")[0], t))
        elif t.startswith("."):
            continue
        else:
            flat.append(("I", t.split()[0], t))
    pos = {name_: i for i, (k, name_, t) in enumerate(flat) if k == "L"}
    for i, (k, op, t) in enumerate(flat):
        if k == "I" and (op.startswith("s_cbranch") or op == "s_branch"):
            tgt = t.split()[-1]
            if tgt in pos and pos[tgt] < i:
                loops.append((pos[tgt], i))
    # innermost loops = loops containing no other loop
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    inner.sort(key=lambda lp: lp[0] - lp[1])
    for lo_, hi_ in inner[:3]:
        mix = collections.Counter(cls(op) for k, op, t in flat[lo_:hi_ + 1] if k == "I")
        print(f"  loop {flat[lo_][1]} ({hi_ - lo_} lines): " + " ".join(f"{k}={v}" for k, v in sorted(mix.items())))
    want = [a.split("=")[1] for a in sys.argv if a.startswith("--loop=")]
    for lo_, hi_ in sorted(set(loops), key=lambda lp: lp[0] - lp[1])[:6]:
        mix = collections.Counter(cls(op) for k, op, t in flat[lo_:hi_ + 1] if k == "I")
        print(f"  loop {flat[lo_][1]} ({hi_ - lo_} lines): " + " ".join(f"{k}={v}" for k, v in sorted(mix.items())))
        if dump and (flat[lo_][1] in want):
            for k, op, t in flat[lo_:hi_ + 1]:
                print("      " + t)
