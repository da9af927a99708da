# own FFT engine against rocFFT at the shapes where it became the default in round 3 (forward + backward message pair)
for c in "conv 64 64" "conv 512 64" "conv 2000 64" "conv 8192 64" "grad 64 64" "grad 4096 64" "grad 1048576 64" "grad 1000000 64" "grad 128x128 64" "conv 128x128 64" "conv 256x256 64" "grad 32x32x32 64" "conv 4096x4096 4"; do
  for dt in f64 f32; do
    a=$(TRAMP_CONV_FFT=own python tools/bench_fftnd.py $dt 10 one $c | awk '{print $(NF-1)}')
    b=$(TRAMP_CONV_FFT=rocfft python tools/bench_fftnd.py $dt 10 one $c | awk '{print $(NF-1)}')
    echo "$c $dt: own $a ms, rocFFT $b ms"
  done
done
