#!/usr/bin/env python3
"""Instruction mix (MFMA / packed VALU / VALU / SALU / LDS / VMEM) and register use of the device kernels in a hipcc object or .so
whose demangled name contains the given substring.   usage: tools/isa_mix.py build/obj/gpe_wide.o 'w_bwd_map<128, 4'"""
import collections, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
obj, pat = sys.argv[1], sys.argv[2]
tmp = tempfile.mkdtemp()
fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                       f"--input={fat}", f"--output={co}"])
notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
regs = {}
for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
    nm = re.search(r"\.name:\s+(\S+)", k).group(1)
    regs[nm] = (int(re.search(r"\.vgpr_count:\s+(\d+)", k).group(1)), int(re.search(r"\.vgpr_spill_count:\s+(\d+)", k).group(1)),
                int(re.search(r"\.sgpr_count:\s+(\d+)", k).group(1)))
dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout.split("\n")
heads = [(i, re.match(r"^[0-9a-f]+ <(\S+)>:", l).group(1)) for i, l in enumerate(dis) if re.match(r"^[0-9a-f]+ <\S+>:", l)]
for n, (i, nm) in enumerate(heads):
    if nm not in regs:
        continue
    dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip().split("(")[0]
    if pat not in dem:
        continue
    end = heads[n + 1][0] if n + 1 < len(heads) else len(dis)
    c = collections.Counter()
    for l in dis[i + 1:end]:
        t = l.strip()
        if not t or t.startswith("<") or t.endswith(":"):
            continue
        o = t.split()[0]
        if o.startswith("v_mfma"): c["mfma"] += 1
        elif o.startswith("v_pk_"): c["vpk"] += 1
        elif o.startswith("v_"): c["valu"] += 1
        elif o.startswith("ds_"): c["lds"] += 1
        elif o.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
        elif o.startswith("s_"): c["salu"] += 1
    v, sp, sg = regs[nm]
    print("%-52s vgpr %3d spill %2d sgpr %3d | mfma %4d valu %4d vpk %3d salu %4d lds %3d vmem %3d" %
          (dem[-52:], v, sp, sg, c["mfma"], c["valu"], c["vpk"], c["salu"], c["lds"], c["vmem"]))
