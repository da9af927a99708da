# eigensolver of the set-up path, both methods on one box (profiles/r06/eigh_tridiag_vs_split.txt)
for method in split tridiag; do
echo "== TRAMP_EIG_METHOD=$method"
TRAMP_EIG_METHOD=$method timeout 600 python tools/probes/probe_eigh.py 2048 4096 8192 16384 24576 32768 2>&1 | grep "gram n"
done
