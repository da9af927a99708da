#!/bin/bash
# registers / spills / LDS / scratch of every kernel in an object of tramp_amd/csrc/build (from the code object's metadata)
#   tools/kernel_resources.sh ew_resident_pwl [name-filter]
o=/root/repo/tramp_amd/csrc/build/$1.o
d=$(mktemp -d)
cp "$o" "$d/x.o"
( cd "$d" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o >/dev/null 2>&1 )
co=$(ls "$d" | grep "hipv4" | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$d/$co" | grep -E "^\s+\.name:|\.vgpr_count|\.private_segment_fixed_size|\.vgpr_spill_count|\.group_segment_fixed_size" | paste - - - - - | sed 's/  */ /g' | grep "${2:-.}" | python3 -c "
import sys,subprocess,re
for l in sys.stdin:
    m=re.search(r'\.name: (\S+)',l)
    d=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()
    d=re.sub(r'\(.*','',d)
    r=dict(re.findall(r'\.(\w+): (\d+)',l))
    print(f\"{d[:100]:100s} vgpr {r.get('vgpr_count')} spill {r.get('vgpr_spill_count')} scratch {r.get('private_segment_fixed_size')} lds {r.get('group_segment_fixed_size')}\")
"
rm -rf "$d"
