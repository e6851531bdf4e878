#!/bin/bash
# Registers, spills and LDS of the kernels of one object file (default: the step kernels), read from the
# code object's metadata: tools/kernel_resources.sh [object] [name filter]
set -e
OBJ=${1:-roadsurf_amd/build/rs_kernels.o}
PAT=${2:-step_kernel}
TMP=$(mktemp -d)
trap 'rm -rf $TMP' EXIT
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $OBJ $TMP/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fat.bin --output=$TMP/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/dev.co > $TMP/notes.txt
python3 - "$TMP/notes.txt" "$PAT" <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
blocks = re.split(r"\n\s+- \.agpr_count:", txt)
rows = []
for b in blocks[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", b) or [None, "?"])[1]
    name = g("name")
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    if pat not in dem:
        continue
    dem = re.sub(r"\(rs::StepArgs\)|void |rs::", "", dem)
    rows.append((dem if dem else name, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
print(f"{'kernel':70s} {'vgpr':>5s} {'vspill':>6s} {'sgpr':>5s} {'sspill':>6s} {'lds':>6s} {'scratch':>7s}")
for r in sorted(rows):
    print(f"{r[0][:70]:70s} {r[1]:>5s} {r[2]:>6s} {r[3]:>5s} {r[4]:>6s} {r[5]:>6s} {r[6]:>7s}")
PY
