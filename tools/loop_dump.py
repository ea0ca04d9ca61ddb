#!/usr/bin/env python3
"""Disassembly of a kernel's hot loop (the backward branch spanning the most MFMAs), MFMA runs collapsed:
usage: tools/loop_dump.py build/obj/gpe_wide.o 'w_bwd_map<128, 4, 1, 1, false, 0>' [full]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
obj, pat = sys.argv[1], sys.argv[2]
full = len(sys.argv) > 3
tmp = tempfile.mkdtemp()
fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                       f"--input={fat}", f"--output={co}"])
dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout.split("\n")
heads = [(i, re.match(r"^[0-9a-f]+ <(\S+)>:", l).group(1)) for i, l in enumerate(dis) if re.match(r"^[0-9a-f]+ <\S+>:", l)]
for n, (i, nm) in enumerate(heads):
    dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip().split("(")[0]
    if not dem.endswith(pat):
        continue
    end = heads[n + 1][0] if n + 1 < len(heads) else len(dis)
    ins = []
    for l in dis[i + 1:end]:
        mm = re.match(r"^\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if mm:
            ins.append((int(mm.group(2), 16), mm.group(1).split()[0], mm.group(1)))
    a2i = {a: k for k, (a, _, _) in enumerate(ins)}
    best = None
    for k, (a, op, txt) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            off = int(txt.split()[-1])
            if off >= 32768: off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in a2i:
                j = a2i[tgt]
                nm_ = sum(1 for x in ins[j:k + 1] if x[1].startswith("v_mfma"))
                if best is None or nm_ > best[0]: best = (nm_, j, k)
    _, j, k = best
    run = 0
    for a, op, txt in ins[j:k + 1]:
        if op.startswith("v_mfma") and not full:
            run += 1
            continue
        if run:
            print(f"        ... {run} x v_mfma")
            run = 0
        print("   ", txt)
    if run: print(f"        ... {run} x v_mfma")
