# register-round sets: parity of the encode path, then the sweep (SWT_EXTRA_FLAGS rebuilds on the box)
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "bpe and not train" 2>&1 | tail -5 && \
bash tools/gpu_enc_sweep.sh "-DSWT_REG_SETS=3" "-DSWT_REG_SETS=3 -DSWT_ENC_WAVES=7" "-DSWT_REG_SETS=2" "-DSWT_REG_SETS=1"
