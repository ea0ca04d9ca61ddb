#!/usr/bin/env python3
"""Dynamic view of a kernel's hot loop: finds the backward branch that spans the most MFMAs in the disassembly of one kernel and prints
the instruction mix of that loop body (per iteration), with the VALU opcodes listed.
usage: tools/loop_mix.py build/obj/gpe_engine.o 'f_backward_coop<64, 4, 1, 1, 3>'"""
import collections, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
obj, pat = sys.argv[1], sys.argv[2]
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
    ins = []                                                   # (address, opcode, text)
    for l in dis[i + 1:end]:
        mm = re.match(r"^\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if mm:
            ins.append((int(mm.group(2), 16), mm.group(1).split()[0], mm.group(1)))
    addr_to_idx = {a: k for k, (a, _, _) in enumerate(ins)}
    best = None
    for k, (a, op, txt) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            off = int(txt.split()[-1])
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in addr_to_idx:
                j = addr_to_idx[tgt]
                nm_ = sum(1 for x in ins[j:k + 1] if x[1].startswith("v_mfma"))
                if best is None or nm_ > best[0]:
                    best = (nm_, j, k)
    if best is None:
        print(dem, ": no loop found"); continue
    _, j, k = best
    body = ins[j:k + 1]
    c = collections.Counter(x[1] for x in body)
    t = collections.Counter()
    for o, cnt in c.items():
        if o.startswith("v_mfma"): t["mfma"] += cnt
        elif o.startswith("v_pk_"): t["vpk"] += cnt
        elif o in ("v_exp_f32_e32", "v_rcp_f32_e32", "v_rsq_f32_e32", "v_log_f32_e32"): t["trans"] += cnt
        elif o.startswith("v_"): t["valu"] += cnt
        elif o.startswith("ds_"): t["lds"] += cnt
        elif o.startswith(("global_", "buffer_", "flat_", "scratch_")): t["vmem"] += cnt
        elif o == "s_nop": t["nop"] += cnt
        elif o == "s_barrier": t["barrier"] += cnt
        elif o.startswith("s_waitcnt"): t["wait"] += cnt
        elif o.startswith("s_"): t["salu"] += cnt
    cyc_m = 32 * t["mfma"]
    cyc_v = 4 * (t["valu"] + t["vpk"]) + 16 * t["trans"]
    print(dem)
    print("  loop body %d instructions: %s" % (len(body), dict(t)))
    print("  MFMA cycles %d, VALU cycles %d  ->  MFMA share if nothing else stalls: %.3f" % (cyc_m, cyc_v, cyc_m / max(1, cyc_m + cyc_v)))
    print("  VALU:", [x for x in c.most_common() if x[0].startswith("v_") and not x[0].startswith("v_mfma")][:24])
    print("  LDS/VMEM:", [x for x in c.most_common() if x[0].startswith(("ds_", "global_", "buffer_"))])
