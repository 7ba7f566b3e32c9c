#!/bin/bash
# Offline ISA statistics of the raster kernels (no GPU needed): static VALU count, SGPR spill traffic
# (v_readlane / v_writelane), register counts.  K2 is VALU-issue bound (profiles/: SQ_ACTIVE_INST_VALU covers
# the whole kernel duration), so the instruction count of the hot blocks is the figure of merit.
# usage: tools/perf/isa_stats.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")/../../miniworld_amd/csrc"
OUT=${OUT:-/tmp/isa}
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -S --cuda-device-only mw_raster.hip -o $OUT/k2.s "$@" 2>/dev/null
python3 - "$OUT/k2.s" <<'PY'
import re, sys
src = open(sys.argv[1]).read().split("\n")
kern, cur = {}, None
for ln in src:
    m = re.match(r"^(mw_\w+):", ln)
    if m: cur = m.group(1); kern[cur] = []
    elif cur is not None:
        kern[cur].append(ln)
        if ln.strip() == "s_endpgm": cur = None
meta = {}
name = None
for ln in src:
    m = re.match(r"\s+\.name:\s+(\w+)", ln)
    if m: name = m.group(1); meta[name] = {}
    for key in ("sgpr_count", "vgpr_count", "sgpr_spill_count", "vgpr_spill_count"):
        m = re.match(r"\s+\." + key + r":\s+(\d+)", ln)
        if m and name: meta[name][key] = int(m.group(1))
for k, lines in kern.items():
    ins = [l.strip().split()[0] for l in lines if l.startswith("\t") and not l.strip().startswith((";", "."))]
    valu = sum(1 for i in ins if i.startswith("v_"))
    lanes = sum(1 for i in ins if i in ("v_readlane_b32", "v_writelane_b32"))
    salu = sum(1 for i in ins if i.startswith("s_"))
    print(f"{k:28s} insts {len(ins):5d} valu {valu:5d} salu {salu:5d} lane-spill {lanes:4d}  {meta.get(k)}")
PY
