for r in 1 2 3; do for name in product splitv1; do L=$PWD/build/var/_ssfm_$name.so; [ $name = product ] && L=$PWD/opticomlib_amd/_ssfm_amd.so; echo -n "$name: "; SSFM_LIB=$L python3 tools/big_n_run.py 24 100; done; done
SSFM_LIB=$PWD/build/var/_ssfm_splitv1.so python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "two_to_the_24" 2>&1 | tail -1
