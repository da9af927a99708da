TRAMP_EIG_VERBOSE=1 timeout 900 python tools/probes/probe_eigh.py 16384 50000 70000 2>&1 | grep -v "^\[tramp factorize\] \(gram\|long\)"
