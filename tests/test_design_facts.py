"""DESIGN.md is part of the boundary deliverable: what it says about the binary must be what the binary is.
Section 4.1's table (VGPRs, LDS, scratch, instruction and multiply counts per kernel) is re-derived from the code objects
inside the built liblig_hip.so (llvm-readelf notes + llvm-objdump), no table cell may exceed 300 characters, and the
library reads its environment in exactly one function (the knobs of the appendix)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _design():
    with open(os.path.join(ROOT, "DESIGN.md")) as f:
        return f.read()


def test_isa_table_of_design_md_matches_the_built_library():
    import design_isa_table as t
    import hip_lib
    mod = hip_lib.load()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    s = _design()
    have = s[s.index(t.BEGIN) + len(t.BEGIN):s.index(t.END)].strip()
    want = t.table(mod.LIB_PATH).strip()
    assert have == want, "DESIGN.md section 4.1 is stale: run `python tools/design_isa_table.py --write`"
    assert "| `k_encode_tiles<10, true>` | 158 |" in have or "k_encode_tiles<10, true>" in have
    assert "`v_mfma` in the whole library" in have and have.rstrip().endswith(": 0.")        # north star: no MFMA (integer field work)


def test_no_table_cell_of_design_md_is_longer_than_300_characters():
    for ln in _design().splitlines():
        if ln.startswith("|"):
            for cell in ln.strip().strip("|").split("|"):
                assert len(cell.strip()) <= 300, "cell of %d characters: %s..." % (len(cell.strip()), cell.strip()[:80])


def test_the_library_reads_its_environment_in_one_place_and_design_md_lists_every_knob():
    csrc = os.path.join(ROOT, "ligero-prover_amd", "csrc")
    sites = {}
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp")):
            with open(os.path.join(csrc, name)) as f:
                txt = f.read()
            n = len(re.findall(r"\bgetenv\s*\(", txt))
            if n:
                sites[name] = n
    assert list(sites) == ["lig_capi.hip"], sites
    with open(os.path.join(csrc, "lig_capi.hip")) as f:
        capi = f.read()
    body = capi[capi.index("const lig::Knobs& lig::knobs()"):capi.index("extern \"C\" {", capi.index("const lig::Knobs& lig::knobs()"))]
    names = set(re.findall(r'"(LIG_[A-Z0-9_]+)"', body))
    assert len(names) >= 20 and capi.count("getenv") == body.count("getenv")
    doc = _design()
    missing = [n for n in sorted(names) if n not in doc]
    assert not missing, "knobs not documented in DESIGN.md's appendix: %s" % missing


def test_measured_figures_of_design_md_are_the_committed_json():
    """every speed / traffic figure DESIGN.md states is a row of section 6's generated table (profiles/ file + JSON
    path); the block is regenerated from the committed files and must be what the document shows -- a figure cannot go stale against the
    JSON it cites.  Outside the block the document does not restate the headline or the dominant kernel's launch time."""
    import design_measured_table as t
    s = _design()
    have = s[s.index(t.BEGIN) + len(t.BEGIN):s.index(t.END)].strip()
    want = t.table().strip()
    assert have == want, "DESIGN.md section 6's measured table is stale: run `python tools/design_measured_table.py --write`"
    assert len([r for r in have.splitlines() if r.startswith("| ")]) >= 20
    outside = s[:s.index(t.BEGIN)] + s[s.index(t.END):]
    # the two figures that went stale in round 5 (headline, K2's in-run launch time) appear in the table only
    import json
    with open(os.path.join(ROOT, "profiles", t.BENCH)) as f:
        line = json.load(f)
    assert ("%.3f × 10⁹" % (line["value"] / 1e9)) not in outside
    assert ("%.1f µs" % (line["roofline"]["avg_launch_ms"] * 1e3)) not in outside
    assert "HIP events of `r05_bench_default.json`" not in s
