for k in 10 50; do for m in 0 4 16; do GF_SMP_PAD_FAMILY=$m python tools/ver67_time.py $k 10 1024 2>&1 | tail -1 | sed "s/^/pad_family=$m /"; done; done
python -m pytest tests/test_smp_gpu.py -x -q -m gpu -k "ver6 or ver7 or wiring or golden" 2>&1 | tail -5
