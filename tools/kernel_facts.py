#!/usr/bin/env python3
"""ISA facts of the kernels inside a built liblig_hip.so: VGPRs / SGPRs / LDS / scratch from the code objects' metadata notes,
static instruction counts from the disassembly.   python tools/kernel_facts.py [lib.so] [name-substring ...]  -> JSON
(DESIGN.md section 4.1 quotes these numbers; tests/test_design_facts.py re-derives them from the library that was built)"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def kernel_facts(lib=None, want=()):
    lib = lib or os.path.join(ROOT, "ligero-prover_amd", "liblig_hip.so")
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(lib, os.path.join(tmp, "l.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "l.so"], cwd=tmp, check=True, capture_output=True)
        facts = {}
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            co = os.path.join(tmp, f)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            meta = {}
            for blk in notes.split("- .agpr_count:")[1:]:
                g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, blk)
                name = g("name")
                if not name:
                    continue
                meta[name.group(1)] = dict(vgprs=int(g("vgpr_count").group(1)), sgprs=int(g("sgpr_count").group(1)),
                                           lds_bytes=int(g("group_segment_fixed_size").group(1)), scratch_bytes=int(g("private_segment_fixed_size").group(1)))
            if not meta:
                continue
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            cur = None
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                if m:
                    cur = m.group(1) if m.group(1) in meta else None
                    if cur:
                        meta[cur].update(instructions=0, v_mad_u64_u32=0, s_barrier=0, v_mfma=0)
                    continue
                if cur is None:
                    continue
                t = ln.split()
                if len(t) < 1 or not re.match(r"^[a-z]", t[0]):
                    continue
                op = t[0]
                if op == "s_endpgm":
                    meta[cur]["instructions"] += 1
                    cur = None
                    continue
                if op.startswith("s_code_end") or op.startswith("s_nop") and False:
                    continue
                meta[cur]["instructions"] += 1
                if op == "v_mad_u64_u32":
                    meta[cur]["v_mad_u64_u32"] += 1
                elif op == "s_barrier":
                    meta[cur]["s_barrier"] += 1
                elif op.startswith("v_mfma"):
                    meta[cur]["v_mfma"] += 1
            facts.update(meta)
        names = demangle(list(facts))
        out = {names[k].replace("lig::", "").split("(")[0].replace("void ", ""): v for k, v in facts.items()}
        if want:
            out = {k: v for k, v in out.items() if any(w in k for w in want)}
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    lib = args.pop(0) if args and args[0].endswith(".so") else None
    print(json.dumps(kernel_facts(lib, args), indent=1, sort_keys=True))
