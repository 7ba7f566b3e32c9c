#!/bin/bash
# Offline ISA of the quad kernel's trivial batch (no GPU needed): opcode histogram of mwq_probe_trivial
set -e
cd "$(dirname "$0")/../../miniworld_amd/csrc"
OUT=${OUT:-/tmp/isa}; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DMWQ_PROBE -S --cuda-device-only mw_rasterq.hip -o $OUT/k2q.s "$@" 2>/dev/null
python3 - "$OUT/k2q.s" <<'PY'
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
cur = None; ins = collections.defaultdict(list)
for ln in src:
    m = re.match(r"^(mw\w+):", ln)
    if m: cur = m.group(1)
    elif cur and ln.startswith("\t") and not ln.strip().startswith((";", ".")):
        ins[cur].append(ln.strip().split()[0])
        if ln.strip() == "s_endpgm": cur = None
FAST = {"v_fma_f32","v_fmac_f32","v_mul_f32","v_add_f32","v_sub_f32","v_mov_b32","v_and_b32","v_or_b32","v_xor_b32","v_add_u32","v_sub_u32","v_ashrrev_i32","v_subrev_u32","v_subrev_f32"}
for k, v in ins.items():
    valu = [i for i in v if i.startswith("v_")]
    fast = sum(1 for i in valu if re.sub(r"_e(32|64)$", "", i) in FAST)
    print(f"{k:24s} insts {len(v):5d} valu {len(valu):5d} (fast-class {fast}) salu {sum(1 for i in v if i.startswith('s_')):5d} ds {sum(1 for i in v if i.startswith('ds_')):4d} vmem {sum(1 for i in v if i.startswith('buffer_') or i.startswith('global_') or i.startswith('flat_')):4d}")
    if k == "mwq_probe_trivial":
        c = collections.Counter(valu)
        print("   ", ", ".join(f"{n} {o}" for o, n in c.most_common(60)))
PY
