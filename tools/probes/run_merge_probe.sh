timeout 600 python -m pytest tests/test_gpu_factorize.py -x -q -m gpu 2>&1 | tail -n 2
TRAMP_TD_PROFILE=1 TRAMP_EIG_VERBOSE=1 timeout 300 python tools/probes/probe_eigh.py 32768 2>&1 | grep "synchronised"
for rep in 1 2; do
for lib in build_ab/libtramp_prev.so tramp_amd/libtramp_hip.so; do
echo "== $lib"
TRAMP_HIP_LIB=$PWD/$lib timeout 300 python tools/probes/probe_eigh.py 24576 32768 2>&1 | grep "gram n"
done
done
