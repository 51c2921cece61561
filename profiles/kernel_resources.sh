#!/bin/bash
# Static resources (VGPRs, SGPRs, spills, LDS, scratch) of every kernel in a .hip file: device-only compile
# of the file with the product's flags, then the code object's metadata notes.
#   profiles/kernel_resources.sh [mvosr_kernels.hip] [extra hipcc flags]
set -e
cd "$(dirname "$0")/../mvoscalerecovery_amd/csrc"
SRC=${1:-mvosr_kernels.hip}; shift || true
OUT=/tmp/$(basename "$SRC" .hip).co
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -c "$SRC" -o "$OUT" "$@"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$OUT" --output="$OUT.elf"
mv "$OUT.elf" "$OUT"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$OUT" | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    import subprocess
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(.*", "", name)
    print("%-70s vgpr %3s sgpr %3s spill %s/%s lds %6s scratch %5s" % (name[-70:], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
'
echo "code object: $OUT  (disassemble: /opt/rocm/lib/llvm/bin/llvm-objdump -d $OUT)"
