#!/bin/bash
set -o pipefail
for f in 1 0; do echo "== SWT_BPE_FUSED=$f"; SWT_BPE_FUSED=$f timeout -k 10 200 python tools/gpu_dbg_bpe.py 2>&1 | tail -30; done
