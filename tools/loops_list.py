#!/usr/bin/env python3
"""All backward-branch loops of a kernel with their MFMA / VALU / LDS / VMEM counts (to pick the real tile loop when an outer loop wraps it).
usage: tools/loops_list.py build/obj/gpe_engine.o 'f_backward_pipe<64, 4, 1, 1, 3, false>' [dump <index>]"""
import collections, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
obj, pat = sys.argv[1], sys.argv[2]
tmp = tempfile.mkdtemp()
fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"])
dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout.split("\n")
heads = [(i, re.match(r"^[0-9a-f]+ <(\S+)>:", l).group(1)) for i, l in enumerate(dis) if re.match(r"^[0-9a-f]+ <\S+>:", l)]
for n, (i, nm) in enumerate(heads):
    dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip().split("(")[0]
    if not dem.endswith(pat): continue
    end = heads[n + 1][0] if n + 1 < len(heads) else len(dis)
    ins = []
    for l in dis[i + 1:end]:
        mm = re.match(r"^\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if mm: ins.append((int(mm.group(2), 16), mm.group(1).split()[0], mm.group(1)))
    a2i = {a: k for k, (a, _, _) in enumerate(ins)}
    loops = []
    for k, (a, op, txt) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            off = int(txt.split()[-1])
            if off >= 32768: off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in a2i: loops.append((a2i[tgt], k))
    def cls(o):
        if o.startswith("v_mfma"): return "mfma"
        if o.startswith("v_pk_"): return "vpk"
        if o.startswith("v_"): return "valu"
        if o.startswith("ds_"): return "lds"
        if o.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
        if o == "s_barrier": return "barrier"
        return "other"
    print(dem, "total instructions", len(ins))
    for li, (j, k) in enumerate(loops):
        c = collections.Counter(cls(x[1]) for x in ins[j:k + 1])
        print("  loop %d: [%d..%d] %d instr: %s" % (li, j, k, k - j + 1, dict(c)))
    if len(sys.argv) > 4 and sys.argv[3] == "dump":
        j, k = loops[int(sys.argv[4])]
        c = collections.Counter(x[1] for x in ins[j:k + 1] if x[1].startswith("v_") and not x[1].startswith("v_mfma"))
        print(c.most_common(40))
        c2 = collections.Counter(x[1] for x in ins[j:k + 1] if x[1].startswith(("ds_", "global_", "buffer_")))
        print(c2.most_common())
