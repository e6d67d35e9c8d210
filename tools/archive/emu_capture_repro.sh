for a in "6000 2 4 1 0" "6000 2 4 1 1" "6000 4 4 1 0" "100000 2 8 0 0" "100000 4 8 0 0" "6000 2 4 0 0" "24000 2 4 1 0"; do
  python tools/scratch/emu_capture_repro.py $a > gpurun_out/repro.txt 2>&1; echo "args $a -> rc $? $(grep -c '^OK' gpurun_out/repro.txt)"
done
WARM=5 python tools/scratch/emu_capture_repro.py 6000 2 4 1 0 > gpurun_out/repro.txt 2>&1; echo "WARM=5 6000 2 4 1 0 -> rc $?"
