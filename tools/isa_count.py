#!/usr/bin/env python3
"""Static instruction mix of the kernels in a device assembly file (hipcc --cuda-device-only -S ...): per kernel whose demangled
name matches the pattern, the number of VALU / SALU / LDS / global-memory / scratch instructions in the body, the registers and
the scratch bytes the compiler reports.  A static count: loops are counted once (the tile kernels' hot loop is the whole body).
    python tools/isa_count.py /tmp/ew_kernels.s 'ew_moments_tile<double, PwlOp<double, 2, false'
"""
import re
import subprocess
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    text = open(path).read()
    # kernels: "<mangled>:" ... ".end_amdhsa_kernel"
    names = re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, flags=re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for m, d in zip(names, dem):
        if pat not in d:
            continue
        i = text.index("\n" + m + ":")
        j = text.index(".amdhsa_kernel " + m, i)
        body = text[i:j]
        meta = text[j:text.index(".end_amdhsa_kernel", j)]
        c = dict(valu=0, salu=0, lds=0, gmem=0, scratch=0, trans=0, f64=0, waitcnt=0, other=0)
        for line in body.splitlines():
            t = line.strip().split()
            if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
                continue
            op = t[0]
            if op.startswith("v_"):
                c["valu"] += 1
                if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op):
                    c["trans"] += 1
                if "f64" in op:
                    c["f64"] += 1
            elif op.startswith("s_waitcnt"):
                c["waitcnt"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith("scratch_"):
                c["scratch"] += 1
            elif op.startswith(("global_", "buffer_", "flat_")):
                c["gmem"] += 1
            else:
                c["other"] += 1
        g = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\S+)", meta) or [None, "?"])[1]       # noqa: E731
        print(d[:150])
        print("   ", c, "vgpr", g("next_free_vgpr"), "accum_offset", g("accum_offset"), "sgpr", g("next_free_sgpr"),
              "scratch_bytes", g("private_segment_fixed_size"), "lds", g("group_segment_fixed_size"))


if __name__ == "__main__":
    main()
