#!/usr/bin/env python3
"""profiles/pmc_valu_lds.json from the per-counter summaries tools/pmc_collect.sh leaves under gpurun_out/pmc/"""
import json
out = {"note": "rocprofv3 derived counters, one --pmc pass each (with --kernel-trace only; tools/pmc_collect.sh, tools/pmc_summary.py): "
               "stand-alone encode launches (python bench.py --workload encode --log2-constraints 23, 512-row launches through the public "
               "lig_encode_rows, i.e. reference-layout K3) and the stand-alone dense AES fill (tools/time_aes.py); percent, averaged over "
               "the launches of the largest grid", "kernels": {}}
for tag, ctrs in (("encode", ["VALUBusy", "LDSBankConflict", "MemUnitStalled"]), ("sampler", ["VALUBusy", "LDSBankConflict"])):
    for c in ctrs:
        d = json.load(open("gpurun_out/pmc/%s_%s.json" % (tag, c)))["kernels"]
        for k, v in d.items():
            if (tag == "encode" and k.startswith("k_encode")) or (tag == "sampler" and k.startswith("k_rng_fill_rows_dense<4>")):
                out["kernels"].setdefault(k, {})[c] = round(v["avg_at_largest_grid"], 2)
print(json.dumps(out, indent=1))
