#!/bin/bash
# Registers, scratch, LDS and code size of the kernels in a built library: tools/kernel_resources.sh [lib] [name filter]
LIB=${1:-raytracetorch_amd/csrc/librtt_hip.so}; PAT=${2:-.}
B=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $LIB /dev/null 2>/dev/null || $B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $LIB $T/x.so
$B/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
$B/llvm-readelf --notes $T/dev.co > $T/notes.txt
$B/llvm-readelf -sW $T/dev.co | awk '$4=="FUNC"{print $3, $8}' > $T/sizes.txt
python3 - $T "$PAT" <<'PY'
import re, sys, subprocess
T, pat = sys.argv[1], sys.argv[2]
sizes = {l.split()[1]: int(l.split()[0]) for l in open(T + "/sizes.txt") if len(l.split()) == 2}
t = open(T + "/notes.txt").read()
for k in re.split(r"\n\s+- \.agpr_count", t)[1:]:
    name = re.search(r"\.name:\s+(\S+)", k).group(1)
    if not re.search(pat, name):
        continue
    g = lambda key: re.search(key + r":\s+(\d+)", k).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("(anonymous namespace)::", "").replace("float __vector(2)", "f2").split("(")[0].replace("void ", "")
    print(f"{dem[:60]:60s} vgpr {g('.vgpr_count'):>3s} sgpr {g('.sgpr_count'):>3s} scratch {g('.private_segment_fixed_size'):>4s} lds {g('.group_segment_fixed_size'):>6s} code {sizes.get(name, 0):>7d}")
PY
rm -rf $T
