export TRAMP_EIG_METHOD=tridiag TRAMP_EIG_VERBOSE=1
for kind in gram deficient identity; do
  TRAMP_TD_MERGE_ABOVE=64 timeout 300 python tools/probes/probe_eigh.py --kind $kind 100 777 4097 2>&1 | grep -v "^\[tramp" 
done
for pb in 256 512 1024 2048; do
echo "== back-transformation panel $pb"
TRAMP_TD_BT_PANEL=$pb timeout 300 python tools/probes/probe_eigh.py 8192 24576 2>&1 | grep "back-tr\|gram n"
done
