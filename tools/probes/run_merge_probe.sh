export TRAMP_EIG_METHOD=tridiag TRAMP_EIG_VERBOSE=1
for kind in gram deficient clustered identity; do
  TRAMP_TD_MERGE_ABOVE=64 timeout 300 python tools/probes/probe_eigh.py --kind $kind 100 777 2048 4097 2>&1 | grep -v "^\[tramp fact" 
done
TRAMP_TD_MERGE_ABOVE=4096 timeout 300 python tools/probes/probe_eigh.py 8192 16384 2>&1 | tail -30
