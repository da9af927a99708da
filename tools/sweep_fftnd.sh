# launch-shape knobs of the N-D FFT passes on a few cases (run through gpurun)
for c in "f32 10 one conv 128x128x128 8" "f32 10 one grad 1024x1024 64" "f64 10 one conv 128x128x128 8"; do
for v in "" "TRAMP_FFT_AXIS_DIV=2" "TRAMP_FFT_AXIS_DIV=8" "TRAMP_FFT_AXIS_DIV=16" "TRAMP_FFT_BLOCK_THREADS=128" "TRAMP_FFT_BLOCK_THREADS=512" "TRAMP_CONV_FFT=rocfft"; do
  echo "== $c  $v  $(env $v python tools/bench_fftnd.py $c)"
done; done
