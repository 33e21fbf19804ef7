export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "single_launch or edge_shapes or encode_word or golden or odd_vocab or fuzz or joined or duplicate" 2>&1 | tail -5 && \
timeout -k 10 300 python tools/gpu_single_call2.py 2>&1 | tail -9
